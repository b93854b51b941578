"""Experiment: does the G2 bucket-accumulation time depend on n (2^20 vs 2^20 - 1), on the scalar generator
or on the group?"""
import sys, torch, numpy as np
sys.path.insert(0, ".")
import dg16_amd, bench
ctx = dg16_amd.Context(0)
dev = torch.device("cuda", 0)
N = 1 << 20
b2q = torch.empty(N * 128, dtype=torch.uint8, device=dev)
ctx.gen_bases_dev("bn254", 2, 2003, N, b2q.data_ptr())
gen = torch.Generator(device=dev); gen.manual_seed(20)
w_gen = bench.rand_fr(N, dev, gen)
lo = torch.randint(-2**63, 2**63 - 1, (N, 3), dtype=torch.int64, device=dev)
hi = torch.randint(0, 0x30644E72E131A029, (N, 1), dtype=torch.int64, device=dev)
w_def = torch.cat([lo, hi], dim=1).contiguous()
out = torch.empty(192, dtype=torch.uint8, device=dev)
for name, w in (("seeded-generator", w_gen), ("default-generator", w_def)):
    for n in (N, N - 1, N // 2):
        ms = []
        for _ in range(3):
            ctx.msm_dev("bn254", 2, b2q.data_ptr(), w.data_ptr(), n, out.data_ptr(), channel=0)
            ctx.sync(0)
            ms.append(ctx.last_kernel_ms(0, 1))
        print(name, "n", n, "G2 accumulate ms", ["%.2f" % x for x in ms])
print("hi limb stats:", int(w_gen[:, 3].max()), int(w_def[:, 3].max()), int(w_gen[:, 3].min()), int(w_def[:,3].min()))
print("lo limb negative fraction:", float((w_gen[:, 0] < 0).float().mean()), float((w_def[:, 0] < 0).float().mean()))
