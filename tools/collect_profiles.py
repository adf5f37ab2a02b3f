#!/usr/bin/env python3
"""Turn the rocprofv3 output directories of one round into the files kept under profiles/.

  python tools/collect_profiles.py TAG STATS_DIR FETCH_DIR WRITE_DIR N_PROOFS_PMC

STATS_DIR  : rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline
FETCH_DIR  : rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0
WRITE_DIR  : same with --pmc WRITE_SIZE (separate pass, as MI355X_MICROARCH.md prescribes)
N_PROOFS_PMC: proofs of the bench shard in each PMC process (1 verified + warm-up + timed + 1 end-to-end + 2 profiled)

Counters are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM/rocprofv3 section): FETCH_SIZE reports half of a wide
coalesced streaming read; it applies to lde_block (contiguous 256 B per wave: checked against its byte model below),
not to the strided kernels (64-B segments) and not to WRITE_SIZE."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, stats_dir, fetch_dir, write_dir, n_proofs = sys.argv[1:6]
n_proofs = int(n_proofs)
# optional: trace elements that go through K1 per shard proof (bench line: roofline.alg_bytes_per_proof / 12).  With it the
# FETCH_SIZE correction of each pass is CALIBRATED on the pass's own byte model (P1 reads 4 B per element, P2 4 B, P3 8 B:
# every input byte is read once) as MI355X_MICROARCH.md asks for access patterns it has not calibrated itself
elements = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")


def short(name):
    return name.split("(")[0]


def pmc(d, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                tot[short(r["Kernel_Name"])] += float(r["Counter_Value"])
                cnt[short(r["Kernel_Name"])] += 1
    return tot, cnt


stats = glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(stats, os.path.join(ROOT, f"{tag}_bench_kernel_stats.csv"))
fetch, nf = pmc(fetch_dir, "FETCH_SIZE")
write, _ = pmc(write_dir, "WRITE_SIZE")
with open(os.path.join(ROOT, f"{tag}_pmc_fetch_write_by_kernel.csv"), "w") as f:
    f.write("kernel,dispatches,FETCH_SIZE_KiB_sum,WRITE_SIZE_KiB_sum\n")
    for k in sorted(fetch, key=lambda k: -(fetch[k] + write.get(k, 0))):
        f.write('"%s",%d,%.1f,%.1f\n' % (k, nf[k], fetch[k], write.get(k, 0.0)))
k1 = []
total = 0.0
# (round 3: the strided passes take four tile columns per thread, P2 two trace columns per workgroup; the few launches of the
#  older kernels on odd widths / other heights are added to their pass)
for names, label, corr in ((("void dvt::ntt_strided_v4_kernel<true>", "void dvt::ntt_strided_kernel<true>"), "P1 ntt_strided<inverse>", 1),
                           (("dvt::lde_block2_kernel", "dvt::lde_block_kernel"), "P2 lde_block", 2),
                           (("void dvt::ntt_strided_v4_kernel<false>", "void dvt::ntt_strided_kernel<false>"), "P3 ntt_strided<forward>", 1)):
    name = names[0]
    for extra_name in names[1:]:
        fetch[name] += fetch.get(extra_name, 0.0)
        write[name] = write.get(name, 0.0) + write.get(extra_name, 0.0)
        nf[name] += nf.get(extra_name, 0)
    rd = fetch[name] * 1024 / n_proofs
    wr = write[name] * 1024 / n_proofs
    if elements:
        model = elements * (8 if label.startswith("P3") else 4)
        corr = 2 if rd < 0.75 * model else 1       # raw counter at about half the bytes the pass must read: the gfx950 wide-read tally
    k1.append(dict(kernel=label, launches_per_proof=nf[name] / n_proofs, fetch_raw_bytes=rd, fetch_correction=corr, read_bytes=rd * corr, write_bytes=wr))
    total += rd * corr + wr
json.dump(dict(k1_hbm_bytes_per_proof=total, proofs_in_run=n_proofs,
               method="rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE over `bench.py --no-cpu-baseline "
                      "--steps 1 --warmup 0 --shards-per-gpu 4`; counters are KiB; FETCH_SIZE x2 on lde_block only (wide contiguous reads, gfx950), see tools/collect_profiles.py",
               kernels=k1), open(os.path.join(ROOT, f"{tag}_pmc_k1_traffic.json"), "w"), indent=1)
print("K1 HBM bytes per proof: %.3f GB" % (total / 1e9))
