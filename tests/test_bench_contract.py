"""CPU tests of the N > 1 bench path (world size 2, gloo) and of the bench line contract."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, time
sys.path.insert(0, %r)
from dvt_circuits_amd.dist_util import Ranks, whole_job_rate
r = Ranks(backend="gloo")
r.barrier()
t0 = time.perf_counter()
time.sleep(0.05 * (r.rank + 1))          # rank 1 is slower: the reported time must be its time
dt = r.max_over_ranks(time.perf_counter() - t0)
r.barrier()
mine = r.shard_of(5)
if r.rank == 0:
    print("RESULT", r.world, round(dt, 2) >= 0.1, mine, whole_job_rate(1000, r.world, 2, 1.0))
else:
    print("OTHER", mine)
r.close()
"""


EXCHANGE_WORKER = r"""
import os, sys, struct
import numpy as np
sys.path.insert(0, %r)
from dvt_circuits_amd import capi
from dvt_circuits_amd.dist_util import Ranks, HEADER_WORDS
r = Ranks(backend="gloo")
n = 5
rng = np.random.default_rng(7)
all_headers = rng.integers(0, 2013265921, size=(n, HEADER_WORDS), dtype=np.uint32)     # what a single process would hold
mine = r.shard_of(n)
got = r.exchange_headers(n, [all_headers[i] for i in mine])                              # each rank contributes only its own
assert got.shape == (n, HEADER_WORDS) and (got == all_headers).all(), "exchange_headers lost or reordered a header"
vk = struct.pack("<I", 0x314b5644) + b"rv32".ljust(16, b"\0") + struct.pack("<8I", *range(1, 9)) + struct.pack("<I", 3) \
     + struct.pack("<6I", 0, 6, 1, 16, 3, 7) + struct.pack("<2I", 1, 0x200800)
ch = capi.rv32_challenges(vk, got)
proofs = r.gather_proofs(n, [b"proof-%%d" %% i * (i + 1) for i in mine])
assert proofs == [b"proof-%%d" %% i * (i + 1) for i in range(n)]
print("CH", r.rank, " ".join(str(int(x)) for x in ch))
if r.rank == 0:
    print("SINGLE", " ".join(str(int(x)) for x in capi.rv32_challenges(vk, all_headers)))
r.close()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world_size_2_gloo_barrier_max_and_sharding():
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "RESULT 2 True [0, 2, 4] 4000.0" in outs[0][0]
    assert "OTHER [1, 3]" in outs[1][0]


def test_world_size_2_header_exchange_and_common_challenges():
    """the one exchange step of the multi-GPU path on CPU (gloo, world size 2): every rank contributes the headers of its
    own shards, gets all of them back in shard order, and derives from them the SAME LogUp challenges a single process
    derives (dvt_rv32_challenges is host-only); the proof gather returns the shard proofs in shard order"""
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", EXCHANGE_WORKER % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines = {ln.split()[0] + ln.split()[1] if ln.startswith("CH") else "SINGLE": ln.split()[2 if ln.startswith("CH") else 1:] for o in outs for ln in o[0].splitlines() if ln.startswith(("CH", "SINGLE"))}
    assert lines["CH0"] == lines["CH1"] == lines["SINGLE"] and len(lines["SINGLE"]) == 8


def test_bench_line_fields_of_committed_profile():
    """the committed bench line of the round carries every field of the contract"""
    line = json.load(open(os.path.join(ROOT, "profiles", "r2_bench_line.json")))
    assert "whole boundary call" in line["config"]["measured_unit"] and line["config"]["shards_per_step"] >= 32
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["vs_baseline"] is None and "workload" in line["config"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
