#!/usr/bin/env python3
"""Stage micro-benchmark (K1 LDE, K2/K3 Merkle) on one GPU with HIP-event timing.
Usage: python tools/bench_stages.py [log_n] [width] [iters]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from dvt_circuits_amd import capi

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
width = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n = 1 << log_n
p = capi.Prover()
src = torch.randint(0, 2013265921, (width * n,), dtype=torch.int32, device="cuda")
t_in = torch.empty_like(src)
t_out = torch.empty(width * 2 * n, dtype=torch.int32, device="cuda")
dg = torch.empty(((4 << log_n) - 1) * 8, dtype=torch.int32, device="cuda")
import time

torch.cuda.synchronize()
if True:
    for name in ("lde", "merkle"):
        times = []
        for it in range(iters + 2):
            t_in.copy_(src)
            torch.cuda.synchronize()
            t0 = time.perf_counter()           # the prover launches on its own stream: time launch -> dvt_sync
            if name == "lde":
                p.coset_lde(t_in, t_out, width, log_n, 0, )
            else:
                p.merkle_commit([(t_out, width, log_n + 1)], dg, )
            p.sync()
            if it >= 2:
                times.append((time.perf_counter() - t0) * 1e3)
        ms = sum(times) / len(times)
        if name == "lde":
            alg = 12 * n * width
            print(f"lde   n=2^{log_n} w={width}: {ms:.3f} ms  alg {alg/1e9:.3f} GB -> {alg/ms/1e6:.1f} GB/s (traffic model 36N: {36*n*width/ms/1e6:.1f} GB/s)")
        else:
            perms = 2 * n * ((width + 7) // 8 + 1)
            print(f"merkle rows=2^{log_n+1} w={width}: {ms:.3f} ms  {perms/ms/1e6:.2f} Gperm/s  read {8*n*width/ms/1e6:.1f} GB/s")
p.close()
