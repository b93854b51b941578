"""The limb-per-lane chains of csrc/lane29.h on the GPU, against the one-product-per-lane chains they replace
(msm_impl.h: dbl_wave29 / add_wave29, themselves pinned to the oracle by the MSM and prover suites): the row primitives
(DPP row shifts / rotations, v_permlane16/32_swap) lane by lane, 4096 products, and for every group the chains cover
(G1 and G2 of BN254, BLS12-381, BLS12-377) 1024 doublings, additions, p + p, p - p, identity operands and a chain of
48 doublings + 3 additions and the affine form (an inversion in lane form).  The program is tools/ubench/lane29_probe (built by csrc/Makefile).  End-to-end parity of the
kernels that USE the chains (Horner tail, scalar multiples, king's sums) is in test_gpu_msm / test_gpu_prover /
test_gpu_dist against the oracle."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tools", "ubench", "lane29_probe")


@pytest.mark.gpu
def test_lane_chains_match_the_wave_chains_on_every_group():
    assert os.path.exists(PROBE), "tools/ubench/lane29_probe is not built (make -C distributed-groth16_amd/csrc)"
    out = subprocess.run([PROBE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    text = out.stdout
    assert re.search(r"row primitives .*: 0 mismatches", text), text
    assert re.search(r"products: 0 of 4096 wrong; \(a - b\)\(2 b\): 0 wrong", text), text
    groups = re.findall(r"^(.+), of 1024 \(256 for the 14-limb G2\): doubling (\d+) wrong, addition (\d+), p \+ p (\d+), p - p (\d+), identity operands (\d+), "
                        r"chain of 48 doublings \+ 3 additions (\d+), affine form (\d+)$", text, re.M)
    assert [g[0] for g in groups] == ["BN254 G1", "BN254 G2", "BLS12-381 G1", "BLS12-377 G1", "BLS12-381 G2", "BLS12-377 G2"], text
    for g in groups:
        assert all(int(x) == 0 for x in g[1:]), g
    # the point of the exercise: a chain of 16 doublings + 1 addition at least twice as fast in every group
    speedups = [float(x) for x in re.findall(r"\((\d+\.\d+)x\)", text)]
    assert speedups and min(speedups) > 2.0, text


_ENV_CHECK = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
from oracle import corc
import dg16_amd
c = dg16_amd.Context(0)
for curve, group, n in (("bn254", 1, 1 << 10), ("bn254", 2, 1 << 12), ("bls12_377", 1, 1 << 13), ("bn254", 1, 1 << 17)):
    bases = c.gen_bases(curve, group, 5 + n, n)
    sc = corc.rand_field(curve, "fr", 6 + n, n, mont=False)
    got = corc.jac_to_affine(curve, group, c.msm(curve, group, bases, sc, in_subgroup=True))
    assert np.array_equal(got, corc.msm(curve, group, bases, sc)), (curve, group, n)
print("ok")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"DG16_NO_LANE_REDUCE": "1", "DG16_NO_LANE_TOP": "1"}, {"DG16_NO_LANE_TOP": "1"}, {}])
def test_bucket_reduction_is_the_same_through_the_row_and_top_kernels(env):
    """The lane-form reductions (msm_lane_reduce_kernel, msm_lane_top) can be switched off in one library: the row / top
    kernels they stand in for (still the path of large bucket sets inside proofs) must give the oracle's MSM as well."""
    import sys
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", _ENV_CHECK % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
