"""Cross-checks the two independent restatements (C oracle vs pure-Python big ints) and the
reference's own relational assertions on the Python side:
  dfft/mod.rs:373,458,555 (d_ifft == domain.ifft, d_fft == domain.fft, round trip),
  dmsm/mod.rs:147-193, examples/dmsm_test.rs:62-64 (d_msm == clear msm),
  pss.rs:174-241 (pack/unpack, share-wise product),
  groth16/examples/sha256.rs:239,254 replaced by the known-trapdoor check (no pairing needed)."""

import random

import numpy as np
import pytest

from oracle import corc
from oracle.pyref.fields import FQ, FR
from oracle.pyref.curves import CURVES
from oracle.pyref.poly import Domain
from oracle.pyref.pss import PackedSharingParams
from oracle.pyref import dist, groth16 as G

ALL = ["bn254", "bls12_381", "bls12_377"]


def pt_to_ints(curve, group, arr):
    """(1, k) limb array (Montgomery) -> pyref affine point."""
    F = FQ[curve]
    nl = F.limbs64
    vals = [F.from_mont(v) for v in corc.arr_to_ints(np.asarray(arr).reshape(-1, nl))]
    if all(v == 0 for v in vals):
        return None
    if group == 1:
        return (vals[0], vals[1])
    return ((vals[0], vals[1]), (vals[2], vals[3]))


def ints_to_pt(curve, group, P):
    F = FQ[curve]
    nl = F.limbs64
    if P is None:
        return np.zeros((1, nl * 2 * group), dtype=np.uint64)
    flat = [P[0], P[1]] if group == 1 else [P[0][0], P[0][1], P[1][0], P[1][1]]
    return corc.ints_to_arr([F.to_mont(v) for v in flat], nl).reshape(1, -1)


@pytest.mark.parametrize("curve", ALL)
@pytest.mark.parametrize("kind", ["fq", "fr"])
def test_field_ops_c_vs_python(curve, kind):
    F = (FQ if kind == "fq" else FR)[curve]
    p, nl = F.p, F.limbs64
    rng = random.Random(11)
    a = [rng.randrange(p) for _ in range(64)] + [0, 1, p - 1, 0, p - 1]
    b = [rng.randrange(p) for _ in range(64)] + [0, p - 1, p - 1, 5, 1]
    A = corc.ints_to_arr([F.to_mont(x) for x in a], nl)
    B = corc.ints_to_arr([F.to_mont(x) for x in b], nl)
    dec = lambda arr: [F.from_mont(v) for v in corc.arr_to_ints(arr)]
    assert dec(corc.field_op(curve, kind, "add", A, B)) == [(x + y) % p for x, y in zip(a, b)]
    assert dec(corc.field_op(curve, kind, "sub", A, B)) == [(x - y) % p for x, y in zip(a, b)]
    assert dec(corc.field_op(curve, kind, "mul", A, B)) == [(x * y) % p for x, y in zip(a, b)]
    assert dec(corc.field_op(curve, kind, "neg", A)) == [(-x) % p for x in a]
    assert dec(corc.field_op(curve, kind, "inv", A)) == [pow(x, p - 2, p) for x in a]
    assert corc.arr_to_ints(corc.field_op(curve, kind, "from_mont", A)) == a
    assert np.array_equal(corc.field_op(curve, kind, "to_mont", corc.ints_to_arr(a, nl)), A)


@pytest.mark.parametrize("curve,group", [("bn254", 1), ("bn254", 2), ("bls12_381", 1),
                                         ("bls12_381", 2), ("bls12_377", 1), ("bls12_377", 2)])
def test_group_and_msm_c_vs_python(curve, group):
    C = CURVES[curve, "g%d" % group]
    r = FR[curve].p
    rng = random.Random(5 + group)
    g = corc.generator(curve, group)
    assert pt_to_ints(curve, group, g) == C.gen
    ks = [rng.randrange(1, r) for _ in range(6)]
    pts = [corc.point_mul(curve, group, g, k) for k in ks]
    for k, P in zip(ks, pts):
        assert pt_to_ints(curve, group, P) == C.mul(C.gen, k)
        assert corc.on_curve(curve, group, P)
    # add, doubling, inverse, identity
    assert pt_to_ints(curve, group, corc.point_add(curve, group, pts[0], pts[1])) == \
        C.mul(C.gen, (ks[0] + ks[1]) % r)
    assert pt_to_ints(curve, group, corc.point_add(curve, group, pts[0], pts[0])) == \
        C.mul(C.gen, 2 * ks[0] % r)
    neg = ints_to_pt(curve, group, C.neg(pt_to_ints(curve, group, pts[0])))
    assert pt_to_ints(curve, group, corc.point_add(curve, group, pts[0], neg)) is None
    # MSM: Pippenger (C) == naive (C) == definition (Python), with edge scalars and an identity base
    n = 40
    bases = np.concatenate([corc.point_mul(curve, group, g, rng.randrange(1, r)) for _ in range(n)])
    bases[7] = 0                      # identity point
    bases[9] = bases[8]               # duplicate point
    sc = [rng.randrange(r) for _ in range(n)]
    sc[0], sc[1], sc[2], sc[8], sc[9] = 0, 1, r - 1, 3, 3
    S = corc.ints_to_arr(sc, 4)
    got = corc.msm(curve, group, bases, S)
    assert np.array_equal(got, corc.msm(curve, group, bases, S, algo=1))
    exp = C.msm([pt_to_ints(curve, group, bases[i:i + 1]) for i in range(n)], sc)
    assert pt_to_ints(curve, group, got) == exp
    # Montgomery-form scalars give the same answer
    Sm = corc.field_op(curve, "fr", "to_mont", S)
    assert np.array_equal(got, corc.msm(curve, group, bases, Sm, scalars_mont=True))
    with pytest.raises(ValueError):
        corc.msm(curve, group, bases, S[:-1])


@pytest.mark.parametrize("curve", ALL)
def test_ntt_c_vs_python_and_definition(curve):
    F = FR[curve]
    rng = random.Random(3)
    for n in (1, 2, 8, 64, 1024):
        x = [rng.randrange(F.p) for _ in range(n)]
        X = corc.ints_to_arr([F.to_mont(v) for v in x], 4)
        dom = Domain(F, n)
        dec = lambda arr: [F.from_mont(v) for v in corc.arr_to_ints(arr)]
        assert dec(corc.ntt(curve, X)) == dom.fft(x)
        assert dec(corc.ntt(curve, X, inverse=True)) == dom.ifft(x)
        if n <= 64:
            assert dom.fft(x) == dom.fft_def(x)
        cd = dom.get_coset(F.generator)
        off = corc.ints_to_arr([F.to_mont(F.generator)], 4)
        assert dec(corc.ntt(curve, X, coset=off)) == cd.fft(x)
        assert dec(corc.ntt(curve, X, inverse=True, coset=off)) == cd.ifft(x)
    # x_i = i as in dist-primitives/examples/dfft_test.rs:20-23
    n = 256
    x = list(range(n))
    X = corc.ints_to_arr([F.to_mont(v) for v in x], 4)
    assert [F.from_mont(v) for v in corc.arr_to_ints(corc.ntt(curve, X))] == Domain(F, n).fft(x)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_h_poly_c_vs_python(curve):
    F = FR[curve]
    rng = random.Random(9)
    m = 64
    a, b, c = ([rng.randrange(F.p) for _ in range(m)] for _ in range(3))
    enc = lambda v: corc.ints_to_arr([F.to_mont(x) for x in v], 4)
    got = corc.h_poly(curve, enc(a), enc(b), enc(c))
    exp = G.witness_map_from_abc(a, b, c, Domain(F, m))
    assert [F.from_mont(v) for v in corc.arr_to_ints(got)] == exp


def test_gen_points_are_arithmetic_progression():
    curve, group = "bn254", 1
    P = corc.gen_points(curve, group, seed=2, n=2500)
    d = None
    for i in (0, 1, 1023, 1024, 2400):
        assert corc.on_curve(curve, group, P[i:i + 1])
    C = CURVES[curve, "g1"]
    p0, p1, p2 = (pt_to_ints(curve, 1, P[i:i + 1]) for i in (0, 1, 2))
    D = C.add(p1, C.neg(p0))
    assert C.add(p1, D) == p2
    pa, pb = (pt_to_ints(curve, 1, P[i:i + 1]) for i in (1023, 1024))   # across a chunk boundary
    assert C.add(pa, D) == pb


# ---- the reference's relational tests, on the Python restatement --------------------------------

@pytest.mark.parametrize("m", [8, 32])
def test_dfft_relations(m):
    F = FR["bls12_377"]                     # dfft/mod.rs:277 uses BLS12-377 Fr, L = 2
    pp = PackedSharingParams(F, 2)
    rng = random.Random(m)
    dom = Domain(F, m)
    x = [rng.randrange(F.p) for _ in range(m)]
    unpack_all = lambda res: [v for sh in dist.transpose(res) for v in pp.unpack(sh)]
    assert unpack_all(dist.d_ifft(dist.share_for_dfft(x, pp), False, 1, False, dom, pp)) == dom.ifft(x)
    assert unpack_all(dist.d_fft(dist.share_for_dfft(x, pp), False, 1, False, dom, pp)) == dom.fft(x)
    mid = dist.d_ifft(dist.share_for_dfft(x, pp), True, 1, False, dom, pp)
    assert unpack_all(dist.d_fft(mid, False, 1, False, dom, pp)) == x


def test_pss_relations():
    F = FR["bls12_377"]
    pp = PackedSharingParams(F, 4)          # pss.rs:163 uses L = 4
    rng = random.Random(1)
    s = [rng.randrange(F.p) for _ in range(pp.l)]
    assert pp.unpack(pp.pack_from_public(s)) == s
    a = [rng.randrange(F.p) for _ in range(pp.l)]
    b = [rng.randrange(F.p) for _ in range(pp.l)]
    prod = [x * y % F.p for x, y in zip(pp.pack_from_public(a), pp.pack_from_public(b))]
    assert pp.unpack2(prod) == [x * y % F.p for x, y in zip(a, b)]


def test_dmsm_relation_and_dpp():
    curve = "bls12_377"                     # dmsm/mod.rs:104 uses BLS12-377
    F = FR[curve]
    C = CURVES[curve, "g1"]
    pp = PackedSharingParams(F, 2)
    rng = random.Random(2)
    M = 8
    pts = [C.mul(C.gen, rng.randrange(1, F.p)) for _ in range(M)]
    sc = [rng.randrange(F.p) for _ in range(M)]
    pk = [pp.packexp_from_public(C, pts[i:i + 2]) for i in range(0, M, 2)]
    assert pp.unpackexp(C, pk[0], False) == pts[0:2]
    sk = [pp.pack_from_public(sc[i:i + 2]) for i in range(0, M, 2)]
    res = dist.d_msm(C, dist.transpose(pk), dist.transpose(sk), pp)
    assert all(r == C.msm(pts, sc) for r in res)
    # d_pp: prefix products of num/den (dpp/mod.rs)
    m = 8
    num = [rng.randrange(1, F.p) for _ in range(m)]
    den = [rng.randrange(1, F.p) for _ in range(m)]
    ns = dist.transpose(dist.pack_vec(num, pp))
    ds = dist.transpose(dist.pack_vec(den, pp))
    out = dist.d_pp(ns, ds, pp)
    got = [v for sh in dist.transpose(out) for v in pp.unpack(sh)]
    exp, run = [], 1
    for x, y in zip(num, den):
        run = run * x * F.inv(y) % F.p
        exp.append(run)
    assert got == exp


def test_groth16_trapdoor_and_mpc_equal_single_prover():
    F = FR["bn254"]
    r1cs, w = G.synthetic_r1cs(F, num_constraints=6, num_instance=2, num_witness=9, seed=4)
    assert G.is_satisfied(r1cs, w, F.p)
    rng = random.Random(7)
    td = tuple(rng.randrange(1, F.p) for _ in range(5))
    pk, sc = G.setup("bn254", r1cs, td)
    g1, g2 = CURVES["bn254", "g1"], CURVES["bn254", "g2"]
    for r, s in ((0, 0), (rng.randrange(F.p), rng.randrange(F.p))):
        A, B, Cc = G.create_proof("bn254", pk, r, s, r1cs, w)
        a, b, c = G.proof_scalars_from_trapdoor(r1cs, F, td, sc, r, s, w)
        assert A == g1.mul(g1.gen, a) and B == g2.mul(g2.gen, b) and Cc == g1.mul(g1.gen, c)
        assert G.verify_in_exponent(r1cs, F, td, sc, (a, b, c), w)
    # groth16/examples/sha256.rs: the 8-party proof equals the arkworks proof (r = s = 0)
    assert G.mpc_prove("bn254", pk, r1cs, w) == G.create_proof("bn254", pk, 0, 0, r1cs, w)
    # prove::A / B / C::compute with every term live (prove.rs:21-136): r, s != 0 and the clear points that make the
    # three compute calls the blinded single-prover proof
    r, s = rng.randrange(1, F.p), rng.randrange(1, F.p)
    fixed = dict(L=g1.add(pk["alpha_g1"], pk["a_query"][0]), N=pk["delta_g1"],
                 Z=g2.add(pk["beta_g2"], pk["b_g2_query"][0]), K=pk["delta_g2"],
                 M=g1.add(pk["beta_g1"], pk["b_g1_query"][0]))
    assert G.mpc_prove("bn254", pk, r1cs, w, r=r, s=s, **fixed) == G.create_proof("bn254", pk, r, s, r1cs, w)
    # ... and with unrelated points: the formulas themselves, d_msm replaced by the clear MSM it equals
    pts = {k: (g2 if k in "ZK" else g1).mul((g2 if k in "ZK" else g1).gen, rng.randrange(1, F.p)) for k in "LNZKM"}
    A, B, Cc = G.mpc_prove("bn254", pk, r1cs, w, r=r, s=s, **pts)
    wv = [x % F.p for x in w]
    h = G.witness_map_from_matrices(r1cs, w, F)
    assert A == g1.add(g1.add(pts["L"], g1.mul(pts["N"], r)), g1.msm(pk["a_query"][1:], wv[1:]))
    assert B == g2.add(g2.add(pts["Z"], g2.mul(pts["K"], s)), g2.msm(pk["b_g2_query"][1:], wv[1:]))
    exp = g1.add(g1.msm(pk["l_query"], wv[2:]), g1.msm(pk["h_query"], h))
    exp = g1.add(exp, g1.mul(A, s))
    exp = g1.add(exp, g1.mul(pts["M"], r))
    exp = g1.add(exp, g1.mul(g1.msm(pk["b_g1_query"][1:], wv[1:]), r))
    assert Cc == exp


@pytest.mark.parametrize("curve,m,n_ranks", [("bn254", 16, 2), ("bn254", 16, 4), ("bn254", 64, 8), ("bls12_381", 64, 4),
                                             ("bls12_377", 32, 2)])
def test_sharded_h_polynomial_restatement_equals_witness_map(curve, m, n_ranks):
    """oracle/pyref/hdist.py (what csrc/ntt.hip does on N ranks: cyclic rows, M-point transforms, two all-to-alls with
    the N-point cross-rank parts in between) against the single prover's witness_map (ark-circom qap.rs:64-91):
    rank rho must end with h[rho + N j]."""
    import random
    from oracle.pyref import groth16 as G, hdist
    from oracle.pyref.fields import FR
    from oracle.pyref.poly import Domain
    F = FR[curve]
    rng = random.Random(m + n_ranks)
    a, b, c = ([rng.randrange(F.p) for _ in range(m)] for _ in range(3))
    ref = G.witness_map_from_abc(a, b, c, Domain(F, m))
    got = hdist.h_poly_sharded(a, b, c, F, n_ranks)
    assert all(got[r] == ref[r::n_ranks] for r in range(n_ranks))


@pytest.mark.parametrize("m,n_ranks,inverse", [(16, 2, False), (16, 4, True), (64, 8, False), (64, 4, True)])
def test_sharded_ntt_restatement_equals_the_domain_transform(m, n_ranks, inverse):
    """oracle/pyref/hdist.py: ntt_sharded (one all-to-all, transposed output) against Radix2EvaluationDomain fft / ifft."""
    import random
    from oracle.pyref import hdist
    from oracle.pyref.fields import FR
    from oracle.pyref.poly import Domain
    F = FR["bn254"]
    rng = random.Random(m * n_ranks)
    x = [rng.randrange(F.p) for _ in range(m)]
    X = Domain(F, m).ifft(x) if inverse else Domain(F, m).fft(x)
    M, S = m // n_ranks, m // n_ranks // n_ranks
    got = hdist.ntt_sharded(x, F, n_ranks, inverse)
    for rho in range(n_ranks):
        assert got[rho] == [X[M * k1 + rho * S + j] for k1 in range(n_ranks) for j in range(S)]
