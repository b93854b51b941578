"""GPU parity: MSM through the C ABI vs the oracle, compared in affine form (bit-exact)."""

import numpy as np
import pytest

from oracle import corc
from oracle.pyref.fields import FR
from gpu_util import ctx

pytestmark = pytest.mark.gpu

GROUPS = [("bn254", 1), ("bn254", 2), ("bls12_381", 1), ("bls12_381", 2), ("bls12_377", 1), ("bls12_377", 2)]


def gmsm(c, *a, **kw):
    """dg16_msm WITH DG16_F_BASES_IN_SUBGROUP (the Python default is the C ABI's: flag not set): the test points are
    multiples of the generator, and the split path is the one a prover's bases take; the tests of the unsplit path pass
    in_subgroup=False themselves."""
    kw.setdefault("in_subgroup", True)
    return c.msm(*a, **kw)


def check(curve, group, bases, scalars, **kw):
    jac = gmsm(ctx(), curve, group, bases, scalars, **kw)
    got = corc.jac_to_affine(curve, group, jac)
    exp = corc.msm(curve, group, bases, scalars, scalars_mont=kw.get("scalars_mont", False))
    assert np.array_equal(got, exp)
    return got


# every (group, size) pair is a real case: no skips.  bn254 G1 gets the odd sizes too; every other group runs
# n = 1, 100, 2^10 (direct atomic sort) and 2^14, 2^16 (W*n >= 2^18: LDS-partitioned sort, segment accumulation,
# giant-bucket work list; for the G2 groups the LDS-staged accumulator incl. BLS12-381's BLOCK = 128 instantiation)
SWEEP = [(c, g, n) for (c, g) in GROUPS for n in (1, 100, 1 << 10, 1 << 14, 1 << 16)] + \
        [("bn254", 1, 7), ("bn254", 1, 1 << 13)]


@pytest.mark.parametrize("curve,group,n", SWEEP)
def test_msm_matches_oracle(curve, group, n):
    if n >= 1 << 14:
        bases = ctx().gen_bases(curve, group, 2 + n, n)     # same (k0, k1) walk as the oracle's generator
    else:
        bases = corc.gen_points(curve, group, 2 + n, n)
    scalars = corc.rand_field(curve, "fr", 9 + n, n, mont=False)
    check(curve, group, bases, scalars)


@pytest.mark.parametrize("curve,group,n", [("bn254", 2, 1 << 14), ("bls12_381", 1, 1 << 14), ("bls12_381", 2, 1 << 13),
                                           ("bls12_377", 2, 1 << 10), ("bls12_377", 1, 33)])
def test_msm_without_the_subgroup_flag_is_the_unsplit_pippenger(curve, group, n):
    """dg16_msm WITHOUT DG16_F_BASES_IN_SUBGROUP (the C ABI's default): every group of cofactor != 1 runs the plain
    Pippenger over n points and all the windows -- the path that also serves MSMs too large for the split's index
    range -- and must equal the oracle's MSM and what the splitting path (flag set) returns."""
    bases = corc.gen_points(curve, group, 40 + n, n) if n < 1 << 14 else ctx().gen_bases(curve, group, 40 + n, n)
    sc = corc.rand_field(curve, "fr", 50 + n, n, mont=False)
    plain = check(curve, group, bases, sc, in_subgroup=False)
    split = check(curve, group, bases, sc, in_subgroup=True)
    assert np.array_equal(plain, split)


def _curve_points_outside_the_subgroup(curve, group, n, seed):
    """n points of E(Fq) (G1) / E'(Fq2) (G2) found by trying x = seed, seed + 1, ..: almost surely NOT in the order-r
    subgroup (cofactor ~2^125 for BLS12-381 G1, ~2^254 for BN254 G2) -- checked."""
    from oracle.pyref.curves import CURVES
    C = CURVES[curve, "g%d" % group]
    F, p = C.F, C.F.p
    assert p % 4 == 3
    pts, x = [], seed
    while len(pts) < n:
        x += 1
        if group == 1:
            rhs = (x * x * x + C.b) % p
            y = pow(rhs, (p + 1) // 4, p)
            if y * y % p != rhs:
                continue
            P = (x % p, y)
        else:
            xx = (x % p, 1)
            rhs = F.add(F.mul(F.sqr(xx), xx), C.b)
            # sqrt in Fq2 = Fq[u] / (u^2 + 1), p = 3 mod 4: candidate a^((q + 1) / 4)-style via the norm
            n0 = (rhs[0] * rhs[0] + rhs[1] * rhs[1]) % p
            s = pow(n0, (p + 1) // 4, p)
            if s * s % p != n0:
                continue
            y = None
            for sg in (s, (-s) % p):
                t = (rhs[0] + sg) * pow(2, p - 2, p) % p
                y0 = pow(t, (p + 1) // 4, p)
                if y0 * y0 % p != t or y0 == 0:
                    continue
                y1 = rhs[1] * pow(2 * y0, p - 2, p) % p
                if F.sqr((y0, y1)) == rhs:
                    y = (y0, y1)
                    break
            if y is None:
                continue
            P = (xx, y)
        assert C.on_curve(P)
        assert C.mul(P, C.order) is not None, "landed in the subgroup"
        pts.append(P)
    return C, pts


@pytest.mark.parametrize("curve,group", [("bls12_381", 1), ("bn254", 2)])
def test_msm_of_curve_points_outside_the_order_r_subgroup(curve, group):
    """VariableBaseMSM::msm is correct for ANY points of the curve; the endomorphism split is not (phi(P) = lambda P
    only in the order-r subgroup).  Without DG16_F_BASES_IN_SUBGROUP the library must return the group element the
    plain double-and-add gives for points of the curve that are NOT in the subgroup (what a caller gets from a decode
    with validate = 0)."""
    n = 40
    C, pts = _curve_points_outside_the_subgroup(curve, group, n, 1234)
    rng = np.random.default_rng(5)
    ks = [int.from_bytes(rng.bytes(31), "little") for _ in range(n)]
    exp = None
    for P, k in zip(pts, ks):
        exp = C.add(exp, C.mul(P, k))
    nl = 4 if curve == "bn254" else 6
    R = 1 << (64 * nl)
    p = C.F.p

    def limbs(v):
        v = v * R % p
        return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nl)]

    rows = []
    for P in pts:
        if group == 1:
            rows.append(limbs(P[0]) + limbs(P[1]))
        else:
            rows.append(limbs(P[0][0]) + limbs(P[0][1]) + limbs(P[1][0]) + limbs(P[1][1]))
    bases = np.array(rows, dtype=np.uint64)
    scal = np.array([[(k >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for k in ks], dtype=np.uint64)
    jac = ctx().msm(curve, group, bases, scal, in_subgroup=False)
    got = corc.jac_to_affine(curve, group, jac)
    want = np.array([limbs(exp[0]) + limbs(exp[1])] if group == 1 else
                    [limbs(exp[0][0]) + limbs(exp[0][1]) + limbs(exp[1][0]) + limbs(exp[1][1])], dtype=np.uint64)
    assert np.array_equal(got.reshape(-1), want.reshape(-1))


def test_gen_bases_equals_oracle_generator():
    # the large cases above take their bases from dg16_gen_bases: pin it to the oracle's walk bit for bit
    for curve, group in GROUPS:
        assert np.array_equal(ctx().gen_bases(curve, group, 77, 300), corc.gen_points(curve, group, 77, 300))


@pytest.mark.parametrize("group", [1, 2])
def test_msm_2_20_bls12_381(group):
    """BASELINE config 5's curve at the size where every large-MSM path engages (partitioned sort, c = 16,
    16 bucket windows, two-stage sums, Horner tail): bit-exact vs the C oracle + split linearity."""
    curve, n = "bls12_381", 1 << 20
    bases = ctx().gen_bases(curve, group, 5, n)
    scalars = corc.rand_field(curve, "fr", 6, n, mont=False)
    full = check(curve, group, bases, scalars)
    h = n // 2 + 12345
    a = corc.jac_to_affine(curve, group, gmsm(ctx(), curve, group, bases[:h], scalars[:h]))
    b = corc.jac_to_affine(curve, group, gmsm(ctx(), curve, group, bases[h:], scalars[h:]))
    assert np.array_equal(corc.point_add(curve, group, a, b), full)


def test_msm_edge_cases():
    curve, group = "bn254", 1
    r = FR[curve].p
    n = 600
    bases = corc.gen_points(curve, group, 4, n)
    scalars = corc.rand_field(curve, "fr", 5, n, mont=False)
    sc = corc.arr_to_ints(scalars)
    sc[0], sc[1], sc[2], sc[3] = 0, 1, r - 1, r - 2
    bases[10] = 0                 # identity point
    bases[11] = 0
    bases[21] = bases[20]         # duplicates with equal scalars (bucket doubling)
    sc[21] = sc[20]
    bases[31] = bases[30]
    sc[31] = r - sc[30]           # P and -P
    scalars = corc.ints_to_arr(sc, 4)
    check(curve, group, bases, scalars)
    # Montgomery-form scalars (arkworks memory)
    check(curve, group, bases, corc.field_op(curve, "fr", "to_mont", scalars), scalars_mont=True)
    # all scalars zero -> identity; all-identity bases -> identity
    z = np.zeros_like(scalars)
    assert not corc.jac_to_affine(curve, group, gmsm(ctx(), curve, group, bases, z)).any()
    assert not corc.jac_to_affine(curve, group, gmsm(ctx(), curve, group, np.zeros_like(bases), scalars)).any()


def test_msm_reference_degenerate_shape():
    # dist-primitives/src/dmsm/mod.rs:155-159: M copies of one point, scalars all 1 (BLS12-377)
    curve, group = "bls12_377", 1
    M = 256
    g = corc.generator(curve, group)
    bases = np.repeat(g, M, axis=0)
    scalars = corc.ints_to_arr([1] * M, 4)
    got = check(curve, group, bases, scalars)
    assert np.array_equal(got, corc.point_mul(curve, group, g, M))


@pytest.mark.parametrize("curve,group,n", [("bls12_381", 2, 1 << 12), ("bls12_377", 2, 1 << 11), ("bn254", 2, 1 << 12),
                                           ("bls12_381", 1, 1 << 12)])
def test_all_equal_points_at_a_size_with_giant_buckets(curve, group, n):
    """dmsm/mod.rs:155-159's shape (M copies of one point) with random scalars at a size where every window has giant
    buckets and every bucket's partial sums are multiples of ONE point: the finalize's additions keep hitting their
    doubling and identity branches.  This is the call on which round 5's form of the 14-limb G2 finalize aborted with an
    HSA aperture violation (a spill register inside the asm statements' register file: DESIGN.md section 7.2;
    tools/repro_abort.py) -- on both paths, unsplit and split."""
    one = corc.gen_points(curve, group, 7, 1)
    bases = np.repeat(one, n, axis=0)
    sc = corc.rand_field(curve, "fr", 60 + n, n, mont=False)
    plain = check(curve, group, bases, sc, in_subgroup=False)
    split = check(curve, group, bases, sc, in_subgroup=True)
    assert np.array_equal(plain, split)


def _doubling_chain(curve, group, seed, n):
    """local_dummy_crs (groth16/examples/local_groth_bench.rs:21-52): s[0] random, s[i] = 2 s[i-1]."""
    out = np.empty((n, corc.point_limbs(curve, group)), dtype=np.uint64)
    out[0] = corc.gen_points(curve, group, seed, 1)[0]
    for i in range(1, n):
        out[i] = corc.point_add(curve, group, out[i - 1:i], out[i - 1:i])[0]
    return out


def test_local_groth_bench_shape_bls12_377():
    """groth16/examples/local_groth_bench.rs:83-158 on its own curve (E = Bls12_377) at its own size m = 2^15
    (:153): iFFT(m) x 3, FFT(2m) x 3 (the vectors grow to the size of `constraint2`), h = p q - w, iFFT(2m), then the
    FIVE MSMs -- E::G1::msm(s, a),
    E::G2::msm(v, a) (:141, the only BLS12-377 G2 use of the reference), E::G1::msm(h, a), E::G1::msm(w, a),
    E::G1::msm(u, h_eval) -- with a_share = vec![rand; n] (ONE value repeated) and doubling-chain bases."""
    curve = "bls12_377"
    log_m = 15
    m = 1 << log_m
    c = ctx()
    p_eval = corc.field_op(curve, "fr", "to_mont", corc.ints_to_arr(list(range(m)), 4))
    want = corc.ntt(curve, p_eval.copy(), inverse=True)
    got = c.ntt(curve, p_eval.copy(), inverse=True)
    assert np.array_equal(got, want)
    wide = np.concatenate([want, np.zeros_like(want)])            # fft_in_place with the 2m domain resizes
    want2 = corc.ntt(curve, wide.copy())
    got2 = c.ntt(curve, wide.copy())
    assert np.array_equal(got2, want2)
    h_want = corc.field_op(curve, "fr", "sub", corc.field_op(curve, "fr", "mul", want2, want2), want2)
    h_got = c.field_op(curve, "fr", "sub", c.field_op(curve, "fr", "mul", got2, got2), got2)
    assert np.array_equal(h_got, h_want)
    h_want = corc.ntt(curve, h_want, inverse=True)
    h_got = c.ntt(curve, h_got, inverse=True)
    assert np.array_equal(h_got, h_want)
    a_share = np.repeat(corc.rand_field(curve, "fr", 8, 1, mont=True), m, axis=0)
    crs = {"s": (1, m), "v": (2, m), "h": (1, m), "w": (1, m), "u": (1, 2 * m)}
    shares = {}
    for i, (name, (group, n)) in enumerate(crs.items()):
        bases = _doubling_chain(curve, group, 40 + i, n)
        sc = h_got if name == "u" else a_share
        shares[name] = check(curve, group, bases, sc, scalars_mont=True)
    pi_c = corc.point_add(curve, 1, corc.point_add(curve, 1, shares["h"], shares["w"]), shares["u"])
    assert pi_c.any() and shares["v"].any()


def test_msm_length_mismatch_and_affine_out():
    import dg16_amd
    curve, group = "bn254", 1
    bases = corc.gen_points(curve, group, 1, 8)
    scalars = corc.rand_field(curve, "fr", 1, 8, mont=False)
    with pytest.raises(dg16_amd.Dg16Error) as e:
        gmsm(ctx(), curve, group, bases, scalars[:7])
    assert e.value.code == 1       # DG16_ERR_LENGTH_MISMATCH, like Err(usize) from G::msm
    aff = gmsm(ctx(), curve, group, bases, scalars, affine=True)
    assert np.array_equal(aff, corc.msm(curve, group, bases, scalars))
    # empty MSM is the identity
    e0 = gmsm(ctx(), curve, group, bases[:0], scalars[:0])
    assert not corc.jac_to_affine(curve, group, e0).any()


def test_msm_2_16_config1():
    # BASELINE configs[0]: BN254 G1, 2^16 points
    curve, group, n = "bn254", 1, 1 << 16
    bases = corc.gen_points(curve, group, 1, n)
    scalars = corc.rand_field(curve, "fr", 1, n, mont=False)
    check(curve, group, bases, scalars)


def test_msm_2_20_config2_and_linearity():
    # BASELINE configs[1]: BN254 G1, 2^20 points; oracle Pippenger needs a few seconds on all cores
    curve, group, n = "bn254", 1, 1 << 20
    bases = ctx().gen_bases(curve, group, 2, n)
    scalars = corc.rand_field(curve, "fr", 2, n, mont=False)
    full = check(curve, group, bases, scalars)
    # size-independent property: MSM(first half) + MSM(second half) == MSM(all)
    h = n // 2
    a = corc.jac_to_affine(curve, group, gmsm(ctx(), curve, group, bases[:h], scalars[:h]))
    b = corc.jac_to_affine(curve, group, gmsm(ctx(), curve, group, bases[h:], scalars[h:]))
    assert np.array_equal(corc.point_add(curve, group, a, b), full)


@pytest.mark.parametrize("kind", ["bits", "bytes", "same", "top_heavy"])
@pytest.mark.parametrize("group", [1, 2])
def test_msm_skewed_witness(kind, group):
    """Real circom witnesses are mostly bits and small integers, so whole waves hit one bucket of one window
    (and the top window of a c that does not divide 254 always does): the wave-aggregated histogram / rank
    path of the digit sort must give the same sum as the oracle."""
    curve, n = "bn254", (1 << 14) + 37
    r = FR[curve].p
    rng = np.random.default_rng(11)
    if kind == "bits":
        sc = [int(v) for v in rng.integers(0, 2, n)]
    elif kind == "bytes":
        sc = [int(v) for v in rng.integers(0, 256, n)]
    elif kind == "same":
        sc = [0x1234567 << 100] * n
    else:   # r - small: every window carries, the top window is constant
        sc = [r - 1 - int(v) for v in rng.integers(0, 4, n)]
    bases = corc.gen_points(curve, group, 6, n)
    check(curve, group, bases, corc.ints_to_arr(sc, 4))


@pytest.mark.parametrize("curve,group,log_n", [("bn254", 1, 0), ("bn254", 1, 7), ("bn254", 1, 16), ("bn254", 2, 14),
                                               ("bls12_381", 1, 15), ("bls12_381", 2, 12), ("bls12_377", 1, 13),
                                               ("bls12_377", 2, 12)])
def test_resident_msm_matches_oracle(curve, group, log_n):
    """dg16_bases_upload + dg16_msm_resident (window tables, one bucket set, no Horner tail) against the oracle and
    against dg16_msm, with fresh scalars per call, canonical and Montgomery scalars, identity bases inside."""
    import dg16_amd
    n = 1 << log_n
    c = ctx()
    bases = corc.gen_points(curve, group, 21 + log_n, n)
    if n >= 8:
        bases[3] = 0          # the identity (0, 0) as a base
    hb = c.bases_upload(curve, group, bases)
    info = hb.info()
    assert info["n"] == n and info["table_bytes"] >= bases.nbytes
    for seed, mont in ((1, False), (2, True)):
        sc = corc.rand_field(curve, "fr", seed, n, mont=mont)
        want = corc.msm(curve, group, bases, sc, scalars_mont=mont)
        got = c.msm_resident(hb, sc, scalars_mont=mont, affine=True)
        assert np.array_equal(got.reshape(-1), np.asarray(want).reshape(-1))
        plain = gmsm(c, curve, group, bases, sc, scalars_mont=mont, affine=True)
        assert np.array_equal(plain, got)
    if n > 1:
        with pytest.raises(dg16_amd.Dg16Error) as e:
            c.msm_resident(hb, sc[:-1])
        assert e.value.code == 1          # DG16_ERR_LENGTH_MISMATCH
    hb.close()


@pytest.mark.parametrize("curve,group,log_n,rows", [("bn254", 1, 14, 8), ("bn254", 2, 13, 5), ("bn254", 1, 16, 3),
                                                    ("bls12_381", 1, 13, 1), ("bls12_377", 2, 12, 4)])
def test_resident_msm_under_a_table_budget(curve, group, log_n, rows):
    """dg16_bases_upload under dg16_ctx_set_table_budget: a budget of `rows` table rows makes the table keep every
    k-th window's row; dg16_msm_resident then runs k bucket sets and a Horner tail.  Same point as the oracle's MSM;
    rows = 1 is the degenerate table (the bases themselves, one bucket set per window)."""
    import dg16_amd
    n = 1 << log_n
    c = dg16_amd.Context(0)
    bases = corc.gen_points(curve, group, 91 + log_n, n)
    bases[5] = 0
    sc = corc.rand_field(curve, "fr", 17, n, mont=False)
    want = corc.msm(curve, group, bases, sc)
    hb = c.bases_upload(curve, group, bases)
    full = hb.info()
    hb.close()
    c.set_table_budget(bases.nbytes * rows)
    hb = c.bases_upload(curve, group, bases)
    info = hb.info()
    assert info["table_bytes"] <= bases.nbytes * rows < full["table_bytes"]
    got = c.msm_resident(hb, sc, affine=True)
    assert np.array_equal(got.reshape(-1), np.asarray(want).reshape(-1))
    hb.close()
    c.close()


def test_resident_msm_2e22_rows_in_chunks():
    """A 2^22-point resident table has ONE bucket set of 2^18 buckets = 1024 rows: the size from which the bucket reduction
    runs msm_rowchunk_kernel + the top kernel's folded mode on a table's bucket set (256 chunk workgroups, k = 2).  Checked
    against the plain MSM of the same inputs (GLV: 2^23 points, 8 windows of 2^15 buckets -- the two paths share no
    reduction geometry) and against the oracle's MSM."""
    import torch
    curve, n = "bn254", 1 << 22
    c = ctx()
    dev = torch.device("cuda:0")
    bases = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    c.gen_bases_dev(curve, 1, 78, n, bases.data_ptr())
    c.sync(0)
    sc_h = corc.rand_field(curve, "fr", 6, n, mont=False)
    sc = torch.from_numpy(sc_h.view(np.int64)).to(dev)
    out = torch.empty((2, 64), dtype=torch.uint8, device=dev)
    hb = c.bases_upload(curve, 1, bases.data_ptr(), n=n, device_ptrs=True)
    c.msm_resident_dev(hb, sc.data_ptr(), n, out[0].data_ptr(), affine=True)
    c.msm_dev(curve, 1, bases.data_ptr(), sc.data_ptr(), n, out[1].data_ptr(), affine=True, in_subgroup=True)
    c.sync(0)
    o = out.cpu().numpy()
    assert np.array_equal(o[0], o[1]) and o[0].any()
    exp = corc.msm(curve, 1, bases.cpu().numpy().view(np.uint64).reshape(n, 8), sc_h, threads=32)
    assert np.array_equal(o[0].view(np.uint64).reshape(1, 8), exp)
    hb.close()


def test_resident_msm_2e20_equals_plain_msm():
    """BASELINE config 2's size through the resident path: same point as dg16_msm (itself oracle-checked above)."""
    import torch
    curve, n = "bn254", 1 << 20
    c = ctx()
    dev = torch.device("cuda:0")
    for group in (1, 2):
        pb = 64 * group
        bases = torch.empty(n * pb, dtype=torch.uint8, device=dev)
        c.gen_bases_dev(curve, group, 77, n, bases.data_ptr())
        c.sync(0)
        sc = torch.from_numpy(corc.rand_field(curve, "fr", 5, n, mont=False).view(np.int64)).to(dev)
        out = torch.empty((2, pb), dtype=torch.uint8, device=dev)
        hb = c.bases_upload(curve, group, bases.data_ptr(), n=n, device_ptrs=True)
        c.msm_resident_dev(hb, sc.data_ptr(), n, out[0].data_ptr(), affine=True)
        c.msm_dev(curve, group, bases.data_ptr(), sc.data_ptr(), n, out[1].data_ptr(), affine=True, in_subgroup=True)
        c.sync(0)
        o = out.cpu().numpy()
        assert np.array_equal(o[0], o[1]) and o[0].any()
        hb.close()


@pytest.mark.parametrize("curve,n", [("bn254", 1 << 13), ("bn254", (1 << 15) + 7), ("bn254", 1 << 16), ("bn254", 1 << 17),
                                      ("bls12_377", 1 << 14), ("bls12_381", 1 << 13), ("bls12_381", 1 << 15)])
def test_short_table_window_rule_and_its_plan_mirror(curve, n):
    """The window width of SHORT resident tables (csrc/msm_impl.h: msm_window_bits with the scalar width -- the width whose top
    window is not a handful of giant buckets: 15 for BN254 / BLS12-377 keys of 2^13 .. 2^16 points, lg + 1 for BLS12-381):
    the library's own report == the mirror `bench.py --dry-run` plans with, and the resident MSM at that width == dg16_msm
    (oracle-checked above) for G1."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    bits = {"bn254": 254, "bls12_381": 255, "bls12_377": 253}[curve]
    fqb = 32 if curve == "bn254" else 48
    c = ctx()
    dev = torch.device("cuda:0")
    bases = torch.empty(n * 2 * fqb, dtype=torch.uint8, device=dev)
    c.gen_bases_dev(curve, 1, 81, n, bases.data_ptr())
    c.sync(0)
    hb = c.bases_upload(curve, 1, bases.data_ptr(), n=n, device_ptrs=True)
    info = hb.info()
    assert info["window_bits"] == bench.table_window_bits(n, bits), (info, bench.table_window_bits(n, bits))
    if curve != "bls12_381":
        assert info["window_bits"] == (15 if n <= (3 << 15) else 16)
    sc = torch.from_numpy(corc.rand_field(curve, "fr", 9, n, mont=False).view(np.int64)).to(dev)
    out = torch.empty((2, 2 * fqb), dtype=torch.uint8, device=dev)
    c.msm_resident_dev(hb, sc.data_ptr(), n, out[0].data_ptr(), affine=True)
    c.msm_dev(curve, 1, bases.data_ptr(), sc.data_ptr(), n, out[1].data_ptr(), affine=True, in_subgroup=True)
    c.sync(0)
    o = out.cpu().numpy()
    assert np.array_equal(o[0], o[1]) and o[0].any()
    hb.close()


@pytest.mark.parametrize("curve,n", [("bn254", (1 << 17) + 3), ("bn254", 1 << 18), ("bls12_377", 1 << 17), ("bls12_381", 1 << 17),
                                      ("bls12_381", (1 << 18) - 5)])
def test_plain_g1_msm_at_the_sixteen_bit_window_sizes(curve, n):
    """Plain G1 MSMs of 2^17 .. 2^18 points run the split halves with c = 16 (eight windows of the 128-bit halves instead of
    log2(2 n) - 4 = 14 / 15: csrc/msm_impl.h, msm_run): against the oracle's MSM, through the host-pointer entry."""
    bases = ctx().gen_bases(curve, 1, 2 + n, n)
    scalars = corc.rand_field(curve, "fr", 9 + n, n, mont=False)
    jac = gmsm(ctx(), curve, 1, bases, scalars)
    got = corc.jac_to_affine(curve, 1, jac)
    exp = corc.msm(curve, 1, bases, scalars, threads=32)
    assert np.array_equal(got, exp)
