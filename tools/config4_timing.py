"""Config 4 (sha256-shaped proof: 29 823 wires, domain 2^15) timed the way bench.py's extras do -- one proof at a time with
a synchronisation and the proof's D2H copy (best of N) -- and as a queue of K proofs.  usage: python tools/config4_timing.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dg16_amd  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = dg16_amd.Context(0)
sha = bench.Workload(ctx, dev, 15, 0, 1, seed=4, curve="bn254", nv=29823, nc=29400, ni=2)
(A, B, C), _ = bench.oracle_prove(sha, bench.cpu_threads())
for _ in range(3):
    gp = bench.prove_once(ctx, sha)
ok = all(np.array_equal(x, y) for x, y in zip((A, B, C), bench.gpu_proof_affine("bn254", gp)))
ts = []
for _ in range(12):
    t0 = time.perf_counter()
    bench.prove_once(ctx, sha)
    ts.append(time.perf_counter() - t0)
K = 50
proof = torch.empty(sha.proof_bytes(), dtype=torch.uint8, device=dev)
t0 = time.perf_counter()
for _ in range(K):
    sha.qap()
    ctx.prove_dev(sha.pk, sha.a.data_ptr(), sha.b.data_ptr(), sha.c.data_ptr(), sha.w.data_ptr(), sha.rs, proof.data_ptr(),
                  scalars_mont=False)
t_host = (time.perf_counter() - t0) / K
for ch in range(3):
    ctx.sync(ch)
t_q = (time.perf_counter() - t0) / K
ok2 = all(np.array_equal(x, y) for x, y in zip((A, B, C), bench.gpu_proof_affine("bn254", proof.cpu().numpy())))
print("config4: one at a time best %.3f ms median %.3f ms; queue of %d: %.3f ms per proof (host enqueue %.3f ms); parity %s / %s"
      % (min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3, K, t_q * 1e3, t_host * 1e3, ok, ok2))
