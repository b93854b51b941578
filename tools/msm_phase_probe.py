"""Per-kernel timing of ONE resident-table MSM run alone on the chip (every phase waits for the previous one: what each
kernel costs when nothing shares the GPU).  Run under rocprofv3 --kernel-trace --stats (tools/prof_run.sh).
usage: python tools/msm_phase_probe.py [group 1|2] [log_n] [reps]"""
import os
import sys

import torch

sys.path.insert(0, ".")
import dg16_amd  # noqa: E402

group = int(sys.argv[1]) if len(sys.argv) > 1 else 1
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
curve = os.environ.get("CURVE", "bn254")
FQB = 32 if curve == "bn254" else 48
FR_TOP = 0x30644E72E131A029 if curve == "bn254" else 0x73EDA753299D7D48
n = 1 << log_n
dev = torch.device("cuda:0")
ctx = dg16_amd.Context(0)
pb = 2 * FQB * group
bases = torch.empty(n * pb, dtype=torch.uint8, device=dev)
ctx.gen_bases_dev(curve, group, 5, n, bases.data_ptr())
ctx.sync(0)
lo = torch.randint(-2**63, 2**63 - 1, (n, 3), dtype=torch.int64, device=dev)
hi = torch.randint(0, FR_TOP, (n, 1), dtype=torch.int64, device=dev)
sc = torch.cat([lo, hi], dim=1).contiguous()
out = torch.empty(3 * FQB * group, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
hb = ctx.bases_upload(curve, group, bases.data_ptr(), n, device_ptrs=True)
for _ in range(reps + 1):
    ctx.msm_resident_dev(hb, sc.data_ptr(), n, out.data_ptr(), channel=0)
    ctx.sync(0)
    print("msm_resident G%d 2^%d: call %.3f ms, accumulate %.3f ms" % (group, log_n, ctx.last_kernel_ms(0, 0),
                                                                       ctx.last_kernel_ms(0, 1)))
hb.close()
