import sys, time, numpy as np
sys.path.insert(0, '.')
import dg16_amd
from oracle import corc
ctx = dg16_amd.Context(0)
for curve, group, n in (("bls12_381", 2, 1), ("bls12_381", 2, 100), ("bn254", 2, 1)):
    bases = corc.gen_points(curve, group, 2 + n, n)
    sc = corc.rand_field(curve, "fr", 9 + n, n, mont=False)
    t = time.time()
    print("start", curve, group, n, flush=True)
    jac = ctx.msm(curve, group, bases, sc)
    print("done %.3f s" % (time.time() - t), np.array_equal(corc.jac_to_affine(curve, group, jac), corc.msm(curve, group, bases, sc)), flush=True)
