import sys, torch
sys.path.insert(0, ".")
import dg16_amd
ctx = dg16_amd.Context(0)
dev = torch.device("cuda:0")
variant = sys.argv[1]
N = 1 << 20
b2q = torch.empty(N * 128, dtype=torch.uint8, device=dev)
ctx.gen_bases_dev("bn254", 2, 2003, N, b2q.data_ptr())
lo = torch.randint(-2**63, 2**63 - 1, (N, 3), dtype=torch.int64, device=dev)
hi = torch.randint(0, 0x30644E72E131A029, (N, 1), dtype=torch.int64, device=dev)
w = torch.cat([lo, hi], dim=1).contiguous()
out = torch.empty(192, dtype=torch.uint8, device=dev)
ms = []
for _ in range(4):
    ctx.msm_dev("bn254", 2, b2q.data_ptr(), w.data_ptr(), N, out.data_ptr(), channel=0)
    if variant == "ctxsync":
        ctx.sync(0)
    else:
        torch.cuda.synchronize()
    ms.append(ctx.last_kernel_ms(0, 1))
print(variant, ["%.2f" % x for x in ms])
