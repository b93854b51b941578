"""GPU parity of the dist-primitives mirror (PSS, d_fft/d_ifft, d_msm, d_pp, deg_red, ext_wit::h) against
the line-by-line restatement in oracle/pyref (which itself reproduces the reference's relational tests).
The protocol is deterministic, so every party's output SHARES must be bit-identical, not only the
unpacked values.  8 parties (l = 2) as host threads over the in-process LocalTestNet."""

import random

import numpy as np
import pytest

from oracle import corc
from oracle.pyref.fields import FQ, FR
from oracle.pyref.curves import CURVES
from oracle.pyref.poly import Domain
from oracle.pyref.pss import PackedSharingParams as RefPSS
from oracle.pyref import dist as R, groth16 as G

pytestmark = pytest.mark.gpu

_state = {}


def parties(curve, l=2):
    """n contexts + params + net, cached per (curve, l)."""
    key = (curve, l)
    if key not in _state:
        import dg16_amd
        from dg16_amd import dist as D
        n = 4 * l
        ctxs = [dg16_amd.Context(0) for _ in range(n)]
        pps = [D.PackedSharingParams(c, curve, l) for c in ctxs]
        _state[key] = (ctxs, pps, D.LocalTestNet(n), D)
    return _state[key]


def enc(F, vals):
    return corc.ints_to_arr([F.to_mont(v) for v in vals], 4)


def dec(F, arr):
    return [F.from_mont(v) for v in corc.arr_to_ints(np.asarray(arr).reshape(-1, 4))]


@pytest.mark.parametrize("curve,l", [("bls12_377", 2), ("bn254", 2), ("bls12_377", 4)])
def test_pss_pack_unpack(curve, l):
    F = FR[curve]
    ctxs, pps, net, D = parties(curve, l)
    ref = RefPSS(F, l)
    rng = random.Random(l)
    cnt = 50
    secrets = [[rng.randrange(F.p) for _ in range(l)] for _ in range(cnt)]
    got = pps[0].pack_from_public(np.concatenate([enc(F, s) for s in secrets]))
    exp = [ref.pack_from_public(s) for s in secrets]
    assert [dec(F, got[i]) for i in range(cnt)] == exp
    assert [dec(F, x) for x in pps[0].unpack(got)] == secrets
    # share-wise product then unpack2 (pss.rs:200-241)
    other = [[rng.randrange(F.p) for _ in range(l)] for _ in range(cnt)]
    prod = [[a * b % F.p for a, b in zip(ref.pack_from_public(s), ref.pack_from_public(o))]
            for s, o in zip(secrets, other)]
    got2 = pps[0].unpack2(np.concatenate([enc(F, p_) for p_ in prod]))
    assert [dec(F, x) for x in got2] == [[a * b % F.p for a, b in zip(s, o)] for s, o in zip(secrets, other)]


@pytest.mark.parametrize("log_m", [3, 6, 10])
@pytest.mark.parametrize("inverse", [False, True])
def test_d_fft_shares_bit_exact(log_m, inverse):
    curve = "bls12_377"                      # dfft/mod.rs:277
    F = FR[curve]
    ctxs, pps, net, D = parties(curve)
    ref = RefPSS(F, 2)
    m = 1 << log_m
    dom = Domain(F, m)
    rng = random.Random(log_m)
    x = [rng.randrange(F.p) for _ in range(m)]
    shares = R.share_for_dfft(x, ref)
    for rearrange, pad, degree2 in ((False, 1, False), (True, 2, False), (True, 1, False)):
        fn = R.d_ifft if inverse else R.d_fft
        exp = fn(shares, rearrange, pad, degree2, dom, ref)
        got = net.simulate_network_round(
            lambda i, h: D.d_fft(ctxs[i], pps[i], h, enc(F, shares[i]), log_m, rearrange, pad, degree2, inverse=inverse))
        assert [dec(F, g) for g in got] == exp, (rearrange, pad)
    # and the relation the reference asserts: unpacked result == domain.fft / ifft (dfft/mod.rs:373,458)
    plain = net.simulate_network_round(
        lambda i, h: D.d_fft(ctxs[i], pps[i], h, enc(F, shares[i]), log_m, False, 1, False, inverse=inverse))
    per_elem = np.stack(plain, axis=1)       # [m/l][n][4]
    vals = [v for row in pps[0].unpack(per_elem) for v in dec(F, row)]
    assert vals == (dom.ifft(x) if inverse else dom.fft(x))


def test_d_fft_degree2_and_size_mismatch():
    import dg16_amd
    curve = "bn254"
    F = FR[curve]
    ctxs, pps, net, D = parties(curve)
    ref = RefPSS(F, 2)
    m = 16
    dom = Domain(F, m)
    rng = random.Random(1)
    x = [rng.randrange(F.p) for _ in range(m)]
    y = [rng.randrange(F.p) for _ in range(m)]
    sx, sy = R.share_for_dfft(x, ref), R.share_for_dfft(y, ref)
    prod = [[a * b % F.p for a, b in zip(px, py)] for px, py in zip(sx, sy)]     # degree-2(t+l) shares
    exp = R.d_fft(prod, False, 1, True, dom, ref)
    got = net.simulate_network_round(
        lambda i, h: D.d_fft(ctxs[i], pps[i], h, enc(F, prod[i]), 4, False, 1, True))
    assert [dec(F, g) for g in got] == exp
    with pytest.raises(dg16_amd.Dg16Error):     # share.len() * l != dom.size()
        D.d_fft(ctxs[0], pps[0], net.party(0), enc(F, prod[0]), 5, False, 1, False)


@pytest.mark.parametrize("curve,group", [("bls12_377", 1), ("bls12_377", 2), ("bn254", 1), ("bn254", 2)])
def test_d_msm_equals_clear_msm(curve, group):
    # dist-primitives/examples/dmsm_test.rs:62-64 and dmsm/mod.rs:147-193
    F, Fq = FR[curve], FQ[curve]
    C = CURVES[curve, "g%d" % group]
    ctxs, pps, net, D = parties(curve)
    ref = RefPSS(F, 2)
    rng = random.Random(9)
    M = 32
    pts_arr = corc.gen_points(curve, group, 3, M)
    sc = [rng.randrange(F.p) for _ in range(M)]
    clear = corc.msm(curve, group, pts_arr, corc.ints_to_arr(sc, 4))
    # pack bases in the exponent on the GPU (packexp_from_public) and scalars with pack_from_public
    packed_bases = pps[0].packexp_from_public(group, pts_arr.reshape(M // 2, 2, -1))    # [M/l][n][..]
    packed_sc = pps[0].pack_from_public(enc(F, sc).reshape(M // 2, 2, 4))               # [M/l][n][4]
    got = net.simulate_network_round(
        lambda i, h: D.d_msm(ctxs[i], pps[i], h, group, packed_bases[:, i], packed_sc[:, i], in_subgroup=True))
    aff = [corc.jac_to_affine(curve, group, g) for g in got]
    assert all(np.array_equal(a, clear) for a in aff)
    # unpackexp(packexp(x)) == x (dmsm/mod.rs:127-145)
    back = pps[0].unpackexp(group, packed_bases[:4], False)
    assert np.array_equal(back.reshape(8, -1), pts_arr[:8])


@pytest.mark.parametrize("curve,group", [("bls12_377", 1), ("bn254", 2)])
def test_d_msm_resident_equals_d_msm(curve, group):
    """dg16_d_msm_resident: every party uploads its base shares once (what PackedProvingKeyShare holds,
    groth16/src/proving_key.rs:26-46) and runs d_msm over the table -- same point as d_msm and as the clear MSM,
    repeatedly with fresh scalars."""
    F = FR[curve]
    ctxs, pps, net, D = parties(curve)
    rng = random.Random(19)
    M = 64
    pts_arr = corc.gen_points(curve, group, 5, M)
    packed_bases = pps[0].packexp_from_public(group, pts_arr.reshape(M // 2, 2, -1))
    resident = [ctxs[i].bases_upload(curve, group, np.ascontiguousarray(packed_bases[:, i])) for i in range(len(ctxs))]
    assert resident[0].info()["n"] == M // 2
    for rep in range(2):
        sc = [rng.randrange(F.p) for _ in range(M)]
        clear = corc.msm(curve, group, pts_arr, corc.ints_to_arr(sc, 4))
        packed_sc = pps[0].pack_from_public(enc(F, sc).reshape(M // 2, 2, 4))
        got = net.simulate_network_round(
            lambda i, h: D.d_msm_resident(ctxs[i], pps[i], h, resident[i], packed_sc[:, i]))
        assert all(np.array_equal(corc.jac_to_affine(curve, group, g), clear) for g in got)
    import dg16_amd
    with pytest.raises(dg16_amd.Dg16Error):      # length mismatch, like VariableBaseMSM::msm's Err(min_len)
        D.d_msm_resident(ctxs[0], pps[0], net.party(0), resident[0], packed_sc[:-1, 0])
    for r in resident:
        r.close()


def test_dpoly_commit_is_the_kzg_commitment():
    """dpoly_commit (no reference module exists; thin wrapper over d_msm): commit(p) with an SRS of known tau must be
    p(tau) * G."""
    curve = "bn254"
    F = FR[curve]
    C = CURVES[curve, "g1"]
    ctxs, pps, net, D = parties(curve)
    rng = random.Random(21)
    M, tau = 16, 0xC0FFEE
    coeffs = [rng.randrange(F.p) for _ in range(M)]
    g = corc.generator(curve, 1)
    srs = np.concatenate([corc.point_mul(curve, 1, g, pow(tau, i, F.p)) for i in range(M)])
    packed_srs = pps[0].packexp_from_public(1, srs.reshape(M // 2, 2, -1))
    packed_c = pps[0].pack_from_public(enc(F, coeffs).reshape(M // 2, 2, 4))
    got = net.simulate_network_round(
        lambda i, h: D.dpoly_commit(ctxs[i], pps[i], h, packed_srs[:, i], packed_c[:, i], in_subgroup=True))
    p_tau = sum(c * pow(tau, i, F.p) for i, c in enumerate(coeffs)) % F.p
    exp = corc.point_mul(curve, 1, g, p_tau)
    assert all(np.array_equal(corc.jac_to_affine(curve, 1, x), exp) for x in got)


def test_d_pp_and_deg_red():
    curve = "bls12_377"
    F = FR[curve]
    ctxs, pps, net, D = parties(curve)
    ref = RefPSS(F, 2)
    rng = random.Random(4)
    m = 64
    num = [rng.randrange(1, F.p) for _ in range(m)]
    den = [rng.randrange(1, F.p) for _ in range(m)]
    ns = R.transpose(R.pack_vec(num, ref))
    ds = R.transpose(R.pack_vec(den, ref))
    exp = R.d_pp(ns, ds, ref)
    got = net.simulate_network_round(lambda i, h: D.d_pp(ctxs[i], pps[i], h, enc(F, ns[i]), enc(F, ds[i])))
    assert [dec(F, g) for g in got] == exp
    prod = [[a * b % F.p for a, b in zip(p_, q_)] for p_, q_ in zip(ns, ds)]
    exp = R.deg_red(prod, ref)
    got = net.simulate_network_round(lambda i, h: D.deg_red(ctxs[i], pps[i], h, enc(F, prod[i])))
    assert [dec(F, g) for g in got] == exp


@pytest.mark.parametrize("log_m", [3, 7])
def test_ext_wit_h_equals_witness_map(log_m):
    # groth16/src/ext_wit.rs:118-190: distributed h == CircomReduction::witness_map_from_matrices
    curve = "bn254"
    F = FR[curve]
    ctxs, pps, net, D = parties(curve)
    ref = RefPSS(F, 2)
    m = 1 << log_m
    rng = random.Random(log_m)
    a, b, c = ([rng.randrange(F.p) for _ in range(m)] for _ in range(3))
    dom = Domain(F, m)
    qs = G.qap_pss(a, b, c, ref)
    exp_shares = G.ext_wit_h(qs, dom, ref)
    got = net.simulate_network_round(
        lambda i, h: D.ext_wit_h(ctxs[i], pps[i], h, enc(F, qs[i][0]), enc(F, qs[i][1]), enc(F, qs[i][2]), log_m))
    assert [dec(F, g) for g in got] == exp_shares
    per_elem = np.stack(got, axis=1)
    vals = [v for row in pps[0].unpack(per_elem) for v in dec(F, row)]
    assert vals == G.witness_map_from_abc(a, b, c, dom)


def test_mpc_prove_equals_single_prover():
    """groth16/examples/sha256.rs: the 8-party proof (after adding a_query[0] + alpha etc. on the
    client, :208-212) equals the single-prover proof with r = s = 0; all on the GPU, checked against
    the big-int prover."""
    from dg16_amd import groth16_mpc as M
    from test_gpu_prover import enc_g1, enc_g2, dec_g1, dec_g2
    curve = "bn254"
    F, Fq = FR[curve], FQ[curve]
    ctxs, pps, net, D = parties(curve)
    r1cs, w = G.synthetic_r1cs(F, num_constraints=13, num_instance=2, num_witness=17, seed=3)
    rng = random.Random(5)
    td = tuple(rng.randrange(1, F.p) for _ in range(5))
    pk, _ = G.setup(curve, r1cs, td)
    a, b, c, dom = G.qap(r1cs, w, F)
    hpk = {k: enc_g1(Fq, pk[k]) for k in ("a_query", "b_g1_query", "h_query", "l_query")}
    hpk["b_g2_query"] = enc_g2(Fq, pk["b_g2_query"])
    crs = M.pack_from_arkworks_proving_key(pps[0], hpk)
    qs = M.qap_pss(pps[0], enc(F, a), enc(F, b), enc(F, c))
    a_sh = M.pack_from_witness(pps[0], enc(F, w[1:]))
    ax_sh = M.pack_from_witness(pps[0], enc(F, w[2:]))
    log_m = dom.size.bit_length() - 1
    res = net.simulate_network_round(
        lambda i, h: M.party_prove(ctxs[i], pps[i], h, crs[i], qs[i], a_sh[i], ax_sh[i], log_m))
    pi_a, pi_b, pi_c = res[0]
    g1, g2 = CURVES[curve, "g1"], CURVES[curve, "g2"]
    A = g1.add(dec_g1(Fq, corc.jac_to_affine(curve, 1, pi_a)), g1.add(pk["a_query"][0], pk["alpha_g1"]))
    B = g2.add(dec_g2(Fq, corc.jac_to_affine(curve, 2, pi_b)), g2.add(pk["b_g2_query"][0], pk["beta_g2"]))
    C = dec_g1(Fq, corc.jac_to_affine(curve, 1, pi_c))
    assert (A, B, C) == G.create_proof(curve, pk, 0, 0, r1cs, w)
    assert all(np.array_equal(r[0], res[0][0]) for r in res)      # same point on every party

    # prove::A / B / C::compute with every term live (prove.rs:21-136): r, s != 0, non-identity L, N, Z, K, M --
    # against the line-by-line restatement (oracle/pyref/groth16.py: prove_A / prove_B / prove_C over d_msm)
    r, sv = rng.randrange(1, F.p), rng.randrange(1, F.p)
    pts = {k: (g2 if k in "ZK" else g1).mul((g2 if k in "ZK" else g1).gen, rng.randrange(1, F.p)) for k in "LNZKM"}
    exp = G.mpc_prove(curve, pk, r1cs, w, r=r, s=sv, **pts)
    kw = {k: (enc_g2 if k in "ZK" else enc_g1)(Fq, [v]) for k, v in pts.items()}
    res = net.simulate_network_round(
        lambda i, h: M.party_prove(ctxs[i], pps[i], h, crs[i], qs[i], a_sh[i], ax_sh[i], log_m,
                                   r=enc(F, [r]), s=enc(F, [sv]), **kw))
    got = (dec_g1(Fq, corc.jac_to_affine(curve, 1, res[0][0])), dec_g2(Fq, corc.jac_to_affine(curve, 2, res[0][1])),
           dec_g1(Fq, corc.jac_to_affine(curve, 1, res[0][2])))
    assert got == exp
    assert all(all(np.array_equal(x, y) for x, y in zip(rr, res[0])) for rr in res)
    # ... and with the clear points that make the same three calls the blinded single-prover proof
    fixed = dict(L=g1.add(pk["alpha_g1"], pk["a_query"][0]), N=pk["delta_g1"],
                 Z=g2.add(pk["beta_g2"], pk["b_g2_query"][0]), K=pk["delta_g2"],
                 M=g1.add(pk["beta_g1"], pk["b_g1_query"][0]))
    kw = {k: (enc_g2 if k in "ZK" else enc_g1)(Fq, [v]) for k, v in fixed.items()}
    res = net.simulate_network_round(
        lambda i, h: M.party_prove(ctxs[i], pps[i], h, crs[i], qs[i], a_sh[i], ax_sh[i], log_m,
                                   r=enc(F, [r]), s=enc(F, [sv]), **kw))
    got = (dec_g1(Fq, corc.jac_to_affine(curve, 1, res[0][0])), dec_g2(Fq, corc.jac_to_affine(curve, 2, res[0][1])),
           dec_g1(Fq, corc.jac_to_affine(curve, 1, res[0][2])))
    assert got == G.create_proof(curve, pk, r, sv, r1cs, w)


def test_prove_compute_length_mismatch_is_an_error_on_every_party():
    """G::msm's Err(len) inside a d_msm of C::compute (dmsm/mod.rs:82, prove.rs:119-125) must come back as
    DG16_ERR_LENGTH_MISMATCH from the joined call, not hang the other two channels."""
    import dg16_amd
    from dg16_amd import groth16_mpc as M
    curve = "bn254"
    ctxs, pps, net, D = parties(curve)
    n = 8
    pts = corc.gen_points(curve, 1, 3, n)
    sc = corc.rand_field(curve, "fr", 3, n)
    Aj = np.zeros((1, 12), dtype=np.uint64)

    def run(i, h):
        with pytest.raises(dg16_amd.Dg16Error) as e:
            M.C(Aj, None, sc[:1], sc[:1], pps[i], pts, pts, pts, sc, sc[:-1], sc).compute(ctxs[i], h)
        return e.value.code

    codes = net.simulate_network_round(run)
    assert all(c in (1, 6) for c in codes) and 1 in codes      # LENGTH_MISMATCH where it arose; NET on peers it starved
    net.L.dg16_localnet_reset(net.h, 60)


# ---- the reference's own sizes (dist-primitives/examples/*.rs run m = 2^15, 8 parties, l = 2) ---------------------
@pytest.mark.parametrize("curve,log_M", [("bls12_377", 15), ("bn254", 16)])
def test_d_msm_reference_size(curve, log_M):
    """dist-primitives/examples/dmsm_test.rs:62-77 (scripts/dmsm_test.zsh): BLS12-377 G1, 2^15 points, 8 parties,
    l = 2 -> a 2^14-point local MSM per party, d_msm == clear MSM.  BASELINE config 1 names the same path on
    BN254 with 2^16 points."""
    F = FR[curve]
    ctxs, pps, net, D = parties(curve)
    M = 1 << log_M
    pts_arr = ctxs[0].gen_bases(curve, 1, 3, M)
    sc = corc.rand_field(curve, "fr", 17, M, mont=True)
    clear = corc.msm(curve, 1, pts_arr, sc, scalars_mont=True)
    packed_bases = pps[0].packexp_from_public(1, pts_arr.reshape(M // 2, 2, -1))    # [M/l][n][..]
    packed_sc = pps[0].pack_from_public(sc.reshape(M // 2, 2, 4))                   # [M/l][n][4]
    got = net.simulate_network_round(
        lambda i, h: D.d_msm(ctxs[i], pps[i], h, 1, np.ascontiguousarray(packed_bases[:, i]),
                             np.ascontiguousarray(packed_sc[:, i]), in_subgroup=True))
    assert all(np.array_equal(corc.jac_to_affine(curve, 1, g), clear) for g in got)


@pytest.mark.parametrize("inverse", [False, True])
def test_d_fft_reference_size(inverse):
    """m = 2^15 (the sha256 path's domain; dfft_test.rs runs 2^10): every party's shares equal the line-by-line
    restatement, and the unpacked result equals the C oracle's plain transform (dfft/mod.rs:373,458)."""
    curve = "bls12_377"
    F = FR[curve]
    ctxs, pps, net, D = parties(curve)
    ref = RefPSS(F, 2)
    log_m = 15
    m = 1 << log_m
    dom = Domain(F, m)
    rng = random.Random(15)
    x = [rng.randrange(F.p) for _ in range(m)]
    shares = R.share_for_dfft(x, ref)
    exp = (R.d_ifft if inverse else R.d_fft)(shares, False, 1, False, dom, ref)
    got = net.simulate_network_round(
        lambda i, h: D.d_fft(ctxs[i], pps[i], h, enc(F, shares[i]), log_m, False, 1, False, inverse=inverse))
    assert [dec(F, g) for g in got] == exp
    per_elem = np.stack(got, axis=1)
    vals = pps[0].unpack(per_elem).reshape(m, 4)
    assert np.array_equal(vals, corc.ntt(curve, enc(F, x), inverse=inverse))


def test_d_pp_reference_size():
    # dist-primitives/examples/dpp_test.rs:54-67: m = 2^15 (its debug_assert is a no-op in release: here it is checked)
    curve = "bls12_377"
    F = FR[curve]
    ctxs, pps, net, D = parties(curve)
    ref = RefPSS(F, 2)
    rng = random.Random(44)
    m = 1 << 15
    num = [rng.randrange(1, F.p) for _ in range(m)]
    den = [rng.randrange(1, F.p) for _ in range(m)]
    ns = R.transpose(R.pack_vec(num, ref))
    ds = R.transpose(R.pack_vec(den, ref))
    exp = R.d_pp(ns, ds, ref)
    got = net.simulate_network_round(lambda i, h: D.d_pp(ctxs[i], pps[i], h, enc(F, ns[i]), enc(F, ds[i])))
    assert [dec(F, g) for g in got] == exp


def test_ext_wit_h_reference_size():
    """groth16/src/ext_wit.rs:118-190 at the sha256 circuit's domain m = 2^15: every party's h shares equal the
    restatement's, and the unpacked h equals the single prover's witness map (C oracle)."""
    curve = "bn254"
    F = FR[curve]
    ctxs, pps, net, D = parties(curve)
    ref = RefPSS(F, 2)
    log_m = 15
    m = 1 << log_m
    a, b, c = (corc.rand_field(curve, "fr", 60 + i, m) for i in range(3))
    ai, bi, ci = (dec(F, v) for v in (a, b, c))
    dom = Domain(F, m)
    qs = G.qap_pss(ai, bi, ci, ref)
    exp_shares = G.ext_wit_h(qs, dom, ref)
    got = net.simulate_network_round(
        lambda i, h: D.ext_wit_h(ctxs[i], pps[i], h, enc(F, qs[i][0]), enc(F, qs[i][1]), enc(F, qs[i][2]), log_m))
    assert [dec(F, g) for g in got] == exp_shares
    vals = pps[0].unpack(np.stack(got, axis=1)).reshape(m, 4)
    assert np.array_equal(vals, corc.h_poly(curve, a, b, c))


def test_mpc_prove_reference_size():
    """groth16/examples/sha256.rs:26-95 (`dsha256`) at the sha256 circuit's shape: 29 823 wires, m = 2^15, 8 parties,
    r = s = 0.  The 8-party proof, completed like :208-212, equals the single prover's -- the C oracle's proof of the
    same statement (and therefore dg16_groth16_prove's, tests/test_gpu_prover.py)."""
    import torch
    import bench
    from dg16_amd import groth16_mpc as M
    curve = "bn254"
    ctxs, pps, net, D = parties(curve)
    dev = torch.device("cuda", 0)
    wl = bench.Workload(ctxs[0], dev, 15, 0, 1, seed=4, curve=curve, nv=29823, nc=29400, ni=2)
    host = lambda t, cols: bench.to_host_u64(t, cols)   # noqa: E731
    hpk = {"a_query": host(wl.aq, 8), "b_g1_query": host(wl.b1q, 8), "b_g2_query": host(wl.b2q, 16),
           "h_query": host(wl.hq, 8), "l_query": host(wl.lq, 8)}
    crs = M.pack_from_arkworks_proving_key(pps[0], hpk)
    a, b, c = (host(t, 4) for t in (wl.a, wl.b, wl.c))
    w = corc.field_op(curve, "fr", "to_mont", host(wl.w, 4))
    qs = M.qap_pss(pps[0], a, b, c)
    a_sh = M.pack_from_witness(pps[0], w[1:])
    ax_sh = M.pack_from_witness(pps[0], w[2:])
    res = net.simulate_network_round(
        lambda i, h: M.party_prove(ctxs[i], pps[i], h, crs[i], qs[i], a_sh[i], ax_sh[i], 15))
    pi_a, pi_b, pi_c = res[0]
    f1 = host(wl.fixed[:192], 8)
    f2 = host(wl.fixed[192:], 16)
    add = lambda g, p, q: corc.point_add(curve, g, p, q)   # noqa: E731
    aff = lambda g, j: corc.jac_to_affine(curve, g, j)     # noqa: E731
    A = add(1, aff(1, pi_a), add(1, hpk["a_query"][0:1], f1[0:1]))
    B = add(2, aff(2, pi_b), add(2, hpk["b_g2_query"][0:1], f2[0:1]))
    C = aff(1, pi_c)
    (eA, eB, eC), _ = bench.oracle_prove(wl, bench.cpu_threads(), 0, 0)
    assert np.array_equal(A, eA) and np.array_equal(B, eB) and np.array_equal(C, eC)

    # prove::A / B / C::compute as the reference defines them (prove.rs:21-46, 62-85, 106-136), every term live:
    # random r, s != 0 and L = alpha_g1 + a_query[0], N = delta_g1, Z = beta_g2 + b_g2_query[0], K = delta_g2,
    # M = beta_g1 + b_g1_query[0] -- the three native calls (C's three d_msm joined on channels 0 / 1 / 2 from three
    # host threads per party: 8 parties x 3 channels in flight) must give the BLINDED single-prover proof
    rng = random.Random(77)
    F = FR[curve]
    r, sv = rng.randrange(1, F.p), rng.randrange(1, F.p)
    L = add(1, f1[0:1], hpk["a_query"][0:1])
    N = f1[2:3]
    Z = add(2, f2[0:1], hpk["b_g2_query"][0:1])
    K = f2[1:2]
    Mp = add(1, f1[1:2], hpk["b_g1_query"][0:1])
    res = net.simulate_network_round(
        lambda i, h: M.party_prove(ctxs[i], pps[i], h, crs[i], qs[i], a_sh[i], ax_sh[i], 15, r=enc(F, [r]),
                                   s=enc(F, [sv]), L=L, N=N, Z=Z, K=K, M=Mp))
    (eA, eB, eC), _ = bench.oracle_prove(wl, bench.cpu_threads(), r, sv)
    assert np.array_equal(aff(1, res[0][0]), eA) and np.array_equal(aff(2, res[0][1]), eB)
    assert np.array_equal(aff(1, res[0][2]), eC)
    assert all(all(np.array_equal(x, y) for x, y in zip(rr, res[0])) for rr in res)

    # ... and with unrelated non-identity L, N, Z, K, M: the formulas of the restatement (oracle/pyref/groth16.py:
    # prove_A / prove_B / prove_C) evaluated with the C oracle, d_msm replaced by the clear MSM it equals
    Lr, Nr, Mr = (corc.gen_points(curve, 1, 900 + i, 1) for i in range(3))
    Zr, Kr = (corc.gen_points(curve, 2, 910 + i, 1) for i in range(2))
    res = net.simulate_network_round(
        lambda i, h: M.party_prove(ctxs[i], pps[i], h, crs[i], qs[i], a_sh[i], ax_sh[i], 15, r=enc(F, [r]),
                                   s=enc(F, [sv]), L=Lr, N=Nr, Z=Zr, K=Kr, M=Mr))
    mul = lambda g, p, k: corc.point_mul(curve, g, p, k)   # noqa: E731
    msm = lambda g, b, sc: corc.msm(curve, g, b, sc, scalars_mont=True)   # noqa: E731
    hv = corc.h_poly(curve, a.copy(), b.copy(), c.copy())
    eA = add(1, add(1, Lr, mul(1, Nr, r)), msm(1, hpk["a_query"][1:], w[1:]))
    eB = add(2, add(2, Zr, mul(2, Kr, sv)), msm(2, hpk["b_g2_query"][1:], w[1:]))
    eC = add(1, msm(1, hpk["l_query"], w[2:]), msm(1, hpk["h_query"], hv))
    eC = add(1, eC, mul(1, eA, sv))
    eC = add(1, eC, mul(1, Mr, r))
    eC = add(1, eC, mul(1, msm(1, hpk["b_g1_query"][1:], w[1:]), r))
    assert np.array_equal(aff(1, res[0][0]), eA) and np.array_equal(aff(2, res[0][1]), eB)
    assert np.array_equal(aff(1, res[0][2]), eC)
    wl.pk.close()
