import sys, os, torch, numpy as np
sys.path.insert(0, ".")
import dg16_amd, bench
from oracle import corc
ctx = dg16_amd.Context(0); dev = torch.device("cuda", 0)
CURVE="bn254"
log_s = int(sys.argv[1])
wl = bench.Workload(ctx, dev, log_s, 0, 1, seed=7)
m, nv, ni = wl.m, wl.nv, wl.ni
w = bench.to_host_u64(wl.w, 4)
proof = torch.empty(96 * 2 + 192, dtype=torch.uint8, device=dev)
ctx.prove_dev(wl.pk, wl.a.data_ptr(), wl.b.data_ptr(), wl.c.data_ptr(), wl.w.data_ptr(), wl.rs, proof.data_ptr(), scalars_mont=False)
for ch in range(3): ctx.sync(ch)
gp = proof.cpu().numpy().view(np.uint64)
gB = corc.jac_to_affine(CURVE, 2, gp[12:36])
b2q = bench.to_host_u64(wl.b2q, 16)
f2 = bench.to_host_u64(wl.fixed[192:], 16); beta2, delta2 = f2[0:1], f2[1:2]
s = int(sum(int(x) << (64 * i) for i, x in enumerate(wl.rs[1])))
add = lambda g, p, q: corc.point_add(CURVE, g, p, q); mul = lambda g, p, k: corc.point_mul(CURVE, g, p, k)
mB2 = corc.msm(CURVE, 2, b2q[1:], w[1:]); B = add(2, add(2, mB2, b2q[0:1]), add(2, beta2, mul(2, delta2, s)))
print("log", log_s, "TABLE_C", os.environ.get("DG16_MSM_TABLE_C"), "SEG", os.environ.get("DG16_MSM_SEG_LOG"), "B equal:", np.array_equal(B, gB))
