"""Plain dg16_msm (fresh bases, device pointers) at small sizes: wall time per call, and -- under
rocprofv3 --kernel-trace -- the dispatch timeline of the last call (tools/rocprof_timeline.py).
usage: python tools/msm_small_probe.py curve group log_n[,log_n..] [reps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import dg16_amd  # noqa: E402

curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
group = int(sys.argv[2]) if len(sys.argv) > 2 else 1
logs = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "12").split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
FQB = 32 if curve == "bn254" else 48
FR_TOP = {"bn254": 0x30644E72E131A029, "bls12_381": 0x73EDA753299D7D48, "bls12_377": 0x12AB655E9A2CA556}[curve]
dev = torch.device("cuda:0")
ctx = dg16_amd.Context(0)
pb = 2 * FQB * group
for log_n in logs:
    n = 1 << log_n
    bases = torch.empty(n * pb, dtype=torch.uint8, device=dev)
    ctx.gen_bases_dev(curve, group, 5, n, bases.data_ptr())
    ctx.sync(0)
    lo = torch.randint(-2**63, 2**63 - 1, (n, 3), dtype=torch.int64, device=dev)
    hi = torch.randint(0, FR_TOP, (n, 1), dtype=torch.int64, device=dev)
    sc = torch.cat([lo, hi], dim=1).contiguous()
    out = torch.empty(3 * FQB * group, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        ctx.msm_dev(curve, group, bases.data_ptr(), sc.data_ptr(), n, out.data_ptr(), in_subgroup=True)
    ctx.sync(0)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.msm_dev(curve, group, bases.data_ptr(), sc.data_ptr(), n, out.data_ptr(), in_subgroup=True)
    ctx.sync(0)
    queued = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.msm_dev(curve, group, bases.data_ptr(), sc.data_ptr(), n, out.data_ptr(), in_subgroup=True)
        ctx.sync(0)
    single = (time.perf_counter() - t0) / reps * 1e3
    print("%s G%d 2^%d: %.3f ms per call queued, %.3f ms synchronised (event-timed call %.3f ms)"
          % (curve, group, log_n, queued, single, ctx.last_kernel_ms(0, 0)), flush=True)
