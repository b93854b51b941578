import sys, numpy as np
sys.path.insert(0, ".")
import dg16_amd
from oracle import corc
ctx = dg16_amd.Context(0)
for group in (1, 2):
    for lg in (16, 17):
        n = 1 << lg
        bases = ctx.gen_bases("bn254", group, 3, n)
        sc = corc.rand_field("bn254", "fr", 5, n, mont=False)
        got = corc.jac_to_affine("bn254", group, ctx.msm("bn254", group, bases, sc))
        exp = corc.msm("bn254", group, bases, sc)
        print("plain G%d 2^%d equal:" % (group, lg), np.array_equal(got, exp))
