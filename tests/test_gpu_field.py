"""GPU parity: Montgomery field kernels and group arithmetic (through the C ABI) vs the oracle.
Bit-exact (integer arithmetic)."""

import numpy as np
import pytest

from oracle import corc
from oracle.pyref.fields import FQ, FR
from gpu_util import ctx

pytestmark = pytest.mark.gpu
ALL = ["bn254", "bls12_381", "bls12_377"]


@pytest.mark.parametrize("curve", ALL)
@pytest.mark.parametrize("kind", ["fq", "fr"])
def test_field_ops_bit_exact(curve, kind):
    F = (FQ if kind == "fq" else FR)[curve]
    n = 1 << 16
    A = corc.rand_field(curve, kind, 21, n)
    B = corc.rand_field(curve, kind, 22, n)
    edge = corc.ints_to_arr([0, F.R, F.p - 1, 1, F.to_mont(F.p - 1)], F.limbs64)
    A[:5] = edge
    B[:5] = edge[::-1]
    c = ctx()
    for op in ("add", "sub", "mul", "sqr", "neg", "from_mont"):
        got = c.field_op(curve, kind, op, A, B)
        assert np.array_equal(got, corc.field_op(curve, kind, op, A, B)), op
    canon = corc.field_op(curve, kind, "from_mont", A)
    assert np.array_equal(c.field_op(curve, kind, "to_mont", canon), A)
    assert np.array_equal(c.field_op(curve, kind, "inv", A[:256]), corc.field_op(curve, kind, "inv", A[:256]))


def test_field_mul_million_pairs_bn254():
    # >= 10^6 random pairs, bit-exact
    n = 1 << 20
    for kind in ("fq", "fr"):
        A = corc.rand_field("bn254", kind, 31, n)
        B = corc.rand_field("bn254", kind, 32, n)
        assert np.array_equal(ctx().field_op("bn254", kind, "mul", A, B),
                              corc.field_op("bn254", kind, "mul", A, B))


@pytest.mark.parametrize("curve,group", [("bn254", 1), ("bn254", 2), ("bls12_381", 1),
                                         ("bls12_381", 2), ("bls12_377", 1)])
def test_gen_bases_matches_oracle(curve, group):
    # exercises scalar_mul, madd, batch inversion and to_affine on the GPU
    n = 1000
    got = ctx().gen_bases(curve, group, 5, n)
    exp = corc.gen_points(curve, group, 5, n)
    assert np.array_equal(got, exp)


def test_empty_inputs():
    c = ctx()
    z = np.zeros((0, 4), dtype=np.uint64)
    assert c.field_op("bn254", "fr", "mul", z, z).shape == (0, 4)
    assert c.gen_bases("bn254", 1, 1, 0).shape == (0, 8)
