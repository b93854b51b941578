"""GPU parity: NTT / iNTT / coset transforms / h-polynomial vs the oracle (bit-exact), plus
size-independent properties at the BASELINE size 2^22 (round trip, linearity)."""

import numpy as np
import pytest

from oracle import corc
from oracle.pyref.fields import FR
from gpu_util import ctx

pytestmark = pytest.mark.gpu
ALL = ["bn254", "bls12_381", "bls12_377"]


# no skips: bn254 and bls12_381 (BASELINE configs 3 and 5) run every plan shape (1, 2 and 3 steps, odd and even
# splits); bls12_377 (the reference's dfft tests, dfft/mod.rs:277) runs one size per plan shape
# 21 and 22: the smallest three-pass plans (BASELINE config 3's size against the oracle, forward and inverse)
SWEEP = [(c, k) for c in ("bn254", "bls12_381") for k in (0, 1, 3, 9, 10, 11, 13, 18, 19, 20, 21, 22)] + \
        [("bls12_377", k) for k in (3, 10, 13, 19, 21)]


@pytest.mark.parametrize("curve,log_n", SWEEP)
def test_ntt_matches_oracle(curve, log_n):
    n = 1 << log_n
    X = corc.rand_field(curve, "fr", 3 + log_n, n)
    c = ctx()
    assert np.array_equal(c.ntt(curve, X), corc.ntt(curve, X))
    assert np.array_equal(c.ntt(curve, X, inverse=True), corc.ntt(curve, X, inverse=True))


@pytest.mark.parametrize("log_n", [4, 10, 12, 19, 21, 22])
def test_coset_ntt_matches_oracle(log_n):
    curve = "bn254"
    F = FR[curve]
    n = 1 << log_n
    X = corc.rand_field(curve, "fr", 77, n)
    off = corc.ints_to_arr([F.to_mont(F.generator)], 4)
    c = ctx()
    assert np.array_equal(c.ntt(curve, X, coset=off), corc.ntt(curve, X, coset=off))
    assert np.array_equal(c.ntt(curve, X, inverse=True, coset=off), corc.ntt(curve, X, inverse=True, coset=off))


def test_ntt_x_equals_i():
    # dist-primitives/examples/dfft_test.rs:20-23 uses x_i = i
    curve = "bn254"
    F = FR[curve]
    n = 1024
    X = corc.ints_to_arr([F.to_mont(i) for i in range(n)], 4)
    assert np.array_equal(ctx().ntt(curve, X), corc.ntt(curve, X))


@pytest.mark.parametrize("curve,log_m", [("bn254", 3), ("bn254", 10), ("bn254", 15), ("bn254", 20),
                                         ("bls12_381", 12), ("bls12_381", 16), ("bls12_381", 20), ("bn254", 21),
                                         ("bn254", 22)])
def test_h_poly_matches_oracle(curve, log_m):
    m = 1 << log_m
    a, b, c_ = (corc.rand_field(curve, "fr", 40 + i, m) for i in range(3))
    got = ctx().h_poly(curve, a, b, c_)
    assert np.array_equal(got, corc.h_poly(curve, a, b, c_))


@pytest.mark.parametrize("curve,log_n", [("bn254", 22), ("bls12_381", 22), ("bls12_381", 24)])
def test_ntt_large_properties(curve, log_n):
    # BASELINE config 3 (BN254, domain 2^22) and config 5's field at 2^22 / 2^24.  Round trip + linearity
    # (size-independent properties) and the oracle NTT at full size (a few seconds on the host cores).
    n = 1 << log_n
    c = ctx()
    X = corc.rand_field(curve, "fr", 3, n)
    Y = c.ntt(curve, X)
    assert np.array_equal(c.ntt(curve, Y, inverse=True), X)          # iNTT(NTT(x)) == x
    Z = corc.rand_field(curve, "fr", 4, n)
    lhs = c.ntt(curve, corc.field_op(curve, "fr", "add", X, Z))
    rhs = corc.field_op(curve, "fr", "add", Y, c.ntt(curve, Z))
    assert np.array_equal(lhs, rhs)                                  # linearity
    assert np.array_equal(Y, corc.ntt(curve, X))                     # oracle at full size
