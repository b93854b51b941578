"""arkworks compressed key files (mpc-api/src/main.rs:154-171, :459-512).

CPU: `dg16_arkkey_layout` on key files written by an INDEPENDENT plain-Python implementation of the ark-serialize
encoding (distributed-groth16_amd/serialize.py, itself pinned by the reference's proof.bin) from the big-int oracle's
setup -- counts, offsets, truncation / trailing-byte errors.
GPU: the batched point codec against that implementation point by point (both directions, identity, both signs,
invalid encodings), a whole proving key file -> `read_proving_key` -> `dg16_pk_create` -> proof == the oracle's, and
2^16 points round trip."""

import random
import struct

import numpy as np
import pytest

import dg16_amd  # noqa: F401
from dg16_amd import arkkey, serialize as S
from oracle.pyref import groth16 as G
from oracle.pyref.fields import FQ, FR


def small_key(seed=5, nc=21, ni=3, nw=17):
    F = FR["bn254"]
    r1cs, w = G.synthetic_r1cs(F, num_constraints=nc, num_instance=ni, num_witness=nw, seed=seed)
    rng = random.Random(seed)
    td = tuple(rng.randrange(1, F.p) for _ in range(5))
    pk, sc = G.setup("bn254", r1cs, td)
    return r1cs, w, pk, sc["m"]


def py_key_bytes(pk, vk_only=False):
    """The file arkworks writes, built with the pure-Python point encoder."""
    g1 = lambda P: S.g1_to_bytes(P)                          # noqa: E731
    g2 = lambda P: S.g2_to_bytes(P)                          # noqa: E731
    vec = lambda f, pts: struct.pack("<Q", len(pts)) + b"".join(f(P) for P in pts)      # noqa: E731
    out = g1(pk["alpha_g1"]) + g2(pk["beta_g2"]) + g2(pk["gamma_g2"]) + g2(pk["delta_g2"]) + vec(g1, pk["gamma_abc_g1"])
    if not vk_only:
        out += (g1(pk["beta_g1"]) + g1(pk["delta_g1"]) + vec(g1, pk["a_query"]) + vec(g1, pk["b_g1_query"])
                + vec(g2, pk["b_g2_query"]) + vec(g1, pk["h_query"]) + vec(g1, pk["l_query"]))
    return out


def test_layout_of_python_written_key_files():
    r1cs, w, pk, m = small_key()
    raw = py_key_bytes(pk)
    lay = arkkey.layout(raw)
    nv = r1cs["num_instance"] + r1cs["num_witness"]
    assert (lay["n_ic"], lay["n_a"], lay["n_b1"], lay["n_b2"], lay["n_h"], lay["n_l"]) == \
        (r1cs["num_instance"], nv, nv, nv, len(pk["h_query"]), r1cs["num_witness"])
    assert lay["bytes"] == len(raw) and lay["off_alpha_g1"] == 0 and lay["off_beta_g2"] == 32
    assert lay["off_ic"] == 32 + 3 * 64 + 8 and lay["off_beta_g1"] == lay["off_ic"] + 32 * lay["n_ic"]
    assert raw[lay["off_a"] - 8:lay["off_a"]] == struct.pack("<Q", nv)
    vk = py_key_bytes(pk, vk_only=True)
    assert arkkey.layout(vk, True)["bytes"] == len(vk)
    for bad in (raw[:-1], raw + b"\0", raw[:lay["off_a"] - 8] + struct.pack("<Q", 1 << 40) + raw[lay["off_a"]:], vk):
        with pytest.raises(arkkey.ArkKeyError):
            arkkey.layout(bad)


# ---------------------------------------------------------------------------------------------- GPU
def _ctx():
    from gpu_util import ctx
    return ctx()


def enc_pts(group, pts):
    Fq = FQ["bn254"]
    from oracle import corc
    out = np.zeros((len(pts), 8 * group), dtype=np.uint64)
    for i, P in enumerate(pts):
        if P is None:
            continue
        co = [P[0], P[1]] if group == 1 else [P[0][0], P[0][1], P[1][0], P[1][1]]
        out[i] = corc.ints_to_arr([Fq.to_mont(v) for v in co], 4).reshape(-1)
    return out


@pytest.mark.gpu
def test_point_codec_matches_the_python_implementation():
    from oracle.pyref.curves import CURVES
    c = _ctx()
    rng = random.Random(3)
    for group, to_b in ((1, S.g1_to_bytes), (2, S.g2_to_bytes)):
        C = CURVES["bn254", "g%d" % group]
        pts = [C.mul(C.gen, rng.randrange(1, C.order)) for _ in range(40)]
        pts += [None, C.neg(pts[0]), pts[0]]                      # identity, both signs of one x
        arr = enc_pts(group, pts)
        want = b"".join(to_b(P) for P in pts)
        assert c.points_compress("bn254", group, arr) == want
        assert np.array_equal(c.points_decompress("bn254", group, want, validate=True), arr)
        # invalid encodings: x with no y on the curve, a coordinate >= q, both flags set
        cb = 32 * group
        x_bad = next(x for x in range(2, 200) if S._sqrt_fq((x ** 3 + 3) % S.Q) is None) if group == 1 else None
        bads = []
        if group == 1:
            bads.append(x_bad.to_bytes(32, "little"))
        bads.append((S.Q + 1).to_bytes(32, "little") + bytes(cb - 32))
        both = bytearray(cb)
        both[-1] = 0xC0
        bads.append(bytes(both))
        for bad in bads:
            with pytest.raises(dg16_amd.Dg16Error):
                c.points_decompress("bn254", group, want[:cb] + bad)
    # G2: a point on the twist outside the order-r subgroup passes Validate::No and fails Validate::Yes
    C2 = CURVES["bn254", "g2"]
    F2 = C2.F
    x = (1, 0)
    while True:
        y = S._sqrt_fq2(F2.add(F2.mul(F2.sqr(x), x), C2.b))
        if y is not None and C2.mul((x, y), C2.order) is not None:
            break
        x = (x[0] + 1, 0)
    raw = S._encode(S._G2, (x, y)) if hasattr(S, "_encode") else None
    if raw is not None:
        assert c.points_decompress("bn254", 2, raw, validate=False).any()
        with pytest.raises(dg16_amd.Dg16Error):
            c.points_decompress("bn254", 2, raw, validate=True)


@pytest.mark.gpu
def test_point_codec_bls12_377():
    """ark-bls12-377 uses the same default encoding (48 / 96 bytes): the group elements of the reference's d_msm tests
    on the wire.  Against the generic plain-Python encoder (tests/ark_points_py.py); Validate::Yes also checks G1 here
    (cofactor != 1)."""
    import ark_points_py as A
    from oracle import corc
    from oracle.pyref.curves import CURVES
    c = _ctx()
    Fq = FQ["bls12_377"]
    rng = random.Random(8)

    def arr_of(group, pts):
        out = np.zeros((len(pts), 12 * group), dtype=np.uint64)
        for k, P in enumerate(pts):
            if P is not None:
                co = [P[0], P[1]] if group == 1 else [P[0][0], P[0][1], P[1][0], P[1][1]]
                out[k] = corc.ints_to_arr([Fq.to_mont(v) for v in co], 6).reshape(-1)
        return out

    for group in (1, 2):
        C = CURVES["bls12_377", "g%d" % group]
        pts = [C.mul(C.gen, rng.randrange(1, C.order)) for _ in range(70)] + [C.mul(C.gen, k) for k in (1, 2, 3)]
        pts += [None, C.neg(pts[0]), pts[0]]
        arr = arr_of(group, pts)
        want = b"".join(A.encode("bls12_377", group, P) for P in pts)
        assert c.points_compress("bls12_377", group, arr) == want
        assert np.array_equal(c.points_decompress("bls12_377", group, want, validate=True), arr)
        cb = 48 * group
        bads = [(Fq.p + 1).to_bytes(48, "little") + bytes(cb - 48)]
        both = bytearray(cb)
        both[-1] = 0xC0
        bads.append(bytes(both))
        if group == 1:
            bads.append(A.x_off_curve("bls12_377").to_bytes(48, "little"))
        for bad in bads:
            with pytest.raises(dg16_amd.Dg16Error):
                c.points_decompress("bls12_377", group, want[:cb] + bad)
        outside = A.g1_point_outside_subgroup("bls12_377") if group == 1 else A.twist_point_outside_subgroup("bls12_377")
        raw = A.encode("bls12_377", group, outside)
        assert np.array_equal(c.points_decompress("bls12_377", group, raw, validate=False), arr_of(group, [outside]))
        with pytest.raises(dg16_amd.Dg16Error, match="subgroup"):
            c.points_decompress("bls12_377", group, raw, validate=True)


@pytest.mark.gpu
def test_point_codec_bls12_381_zcash_form():
    """BLS12-381 (BASELINE config 5's curve) in the encoding ark-bls12-381 0.4 uses -- zcash / IETF: big-endian x, G2 as
    x.c1 || x.c0, flags 0x80 compressed / 0x40 infinity / 0x20 larger y in the FIRST byte -- through the C ABI on the
    GPU: the generators give the published strings, random points the independent Python encoder's bytes, decode is
    the inverse; malformed encodings and points outside the order-r subgroup (Validate::Yes) are refused."""
    import ark_points_py as A
    from oracle import corc
    from oracle.pyref.curves import CURVES
    c = _ctx()
    Fq = FQ["bls12_381"]
    rng = random.Random(9)

    def arr_of(group, pts):
        out = np.zeros((len(pts), 12 * group), dtype=np.uint64)
        for k, P in enumerate(pts):
            if P is not None:
                co = [P[0], P[1]] if group == 1 else [P[0][0], P[0][1], P[1][0], P[1][1]]
                out[k] = corc.ints_to_arr([Fq.to_mont(v) for v in co], 6).reshape(-1)
        return out

    for group in (1, 2):
        C = CURVES["bls12_381", "g%d" % group]
        pts = [C.gen] + [C.mul(C.gen, rng.randrange(1, C.order)) for _ in range(70)] + [C.mul(C.gen, k) for k in (2, 3)]
        pts += [None, C.neg(pts[1]), pts[1]]
        arr = arr_of(group, pts)
        want = b"".join(A.encode_zcash(group, P) for P in pts)
        cb = 48 * group
        assert want[:cb] == (A.ZCASH_G1_GENERATOR if group == 1 else A.ZCASH_G2_GENERATOR)
        assert c.points_compress("bls12_381", group, arr) == want
        assert np.array_equal(c.points_decompress("bls12_381", group, want, validate=True), arr)
        unc = bytearray(want[:cb])
        unc[0] &= 0x7F                                                    # not flagged as compressed
        big = bytearray((Fq.p + 1).to_bytes(48, "big") + bytes(cb - 48))
        big[0] |= 0x80                                                    # coordinate not reduced
        bads = [bytes(unc), bytes(big), bytes([0xE0]) + bytes(cb - 1), bytes([0xC0]) + bytes(cb - 2) + b"\x01"]
        if group == 1:
            off = bytearray(A.x_off_curve("bls12_381").to_bytes(48, "big"))
            off[0] |= 0x80
            bads.append(bytes(off))
        for bad in bads:
            with pytest.raises(dg16_amd.Dg16Error):
                c.points_decompress("bls12_381", group, want[:cb] + bad)
        outside = A.g1_point_outside_subgroup("bls12_381") if group == 1 else A.twist_point_outside_subgroup("bls12_381")
        raw = A.encode_zcash(group, outside)
        assert np.array_equal(c.points_decompress("bls12_381", group, raw, validate=False), arr_of(group, [outside]))
        with pytest.raises(dg16_amd.Dg16Error, match="subgroup"):
            c.points_decompress("bls12_381", group, raw, validate=True)


@pytest.mark.gpu
def test_proving_key_file_to_proof():
    """key file (written by the Python encoder) -> read_proving_key (GPU square roots) -> dg16_pk_create -> prove ==
    the oracle's proof; write_proving_key(read(...)) gives the file back."""
    from oracle import corc
    from test_gpu_prover import dec_g1, dec_g2, enc_fr
    c = _ctx()
    F, Fq = FR["bn254"], FQ["bn254"]
    r1cs, w, pk, m = small_key(seed=11, nc=57, ni=3, nw=40)
    raw = py_key_bytes(pk)
    key = arkkey.read_proving_key(c, raw)
    assert [dec_g1(Fq, r) for r in key["a_query"]] == pk["a_query"]
    assert [dec_g2(Fq, r) for r in key["b_g2_query"]] == pk["b_g2_query"]
    assert arkkey.write_proving_key(c, key) == raw
    vk_raw = py_key_bytes(pk, vk_only=True)
    assert arkkey.write_verifying_key(c, arkkey.read_verifying_key(c, vk_raw, validate=True)) == vk_raw
    rk = arkkey.resident_key(c, key, r1cs["num_instance"], m)
    a, b, cc, dom = G.qap(r1cs, w, F)
    rng = random.Random(2)
    r, s = rng.randrange(1, F.p), rng.randrange(1, F.p)
    A, B, C = c.prove(rk, enc_fr(F, a), enc_fr(F, b), enc_fr(F, cc), enc_fr(F, w), enc_fr(F, [r]), enc_fr(F, [s]))
    proof = (dec_g1(Fq, corc.jac_to_affine("bn254", 1, A)), dec_g2(Fq, corc.jac_to_affine("bn254", 2, B)),
             dec_g1(Fq, corc.jac_to_affine("bn254", 1, C)))
    assert proof == G.create_proof("bn254", pk, r, s, r1cs, w)
    rk.close()


@pytest.mark.gpu
def test_point_codec_round_trip_2e16():
    import torch
    c = _ctx()
    n = 1 << 16
    dev = torch.device("cuda:0")
    for group in (1, 2):
        pts = torch.empty(n * 64 * group, dtype=torch.uint8, device=dev)
        c.gen_bases_dev("bn254", group, 91, n, pts.data_ptr())
        comp = torch.empty(n * 32 * group, dtype=torch.uint8, device=dev)
        back = torch.empty_like(pts)
        c.points_compress_dev("bn254", group, pts.data_ptr(), n, comp.data_ptr())
        c.points_decompress_dev("bn254", group, comp.data_ptr(), n, back.data_ptr(), validate=(group == 2))
        c.sync(0)
        assert torch.equal(pts, back)


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["bn254", "bls12_377", "bls12_381"])
def test_vec_fr_wire_form(curve):
    """`Vec<F>::serialize_compressed` as MpcSerNet sends it (dist-primitives/src/channel/mod.rs:14,49): u64 length ||
    canonical little-endian elements -- against the integers, both directions, plus the error cases."""
    from oracle import corc
    c = _ctx()
    F = FR[curve]
    rng = random.Random(8)
    vals = [0, 1, F.p - 1] + [rng.randrange(F.p) for _ in range(61)]
    mont = corc.ints_to_arr([F.to_mont(v) for v in vals], 4)
    want = struct.pack("<Q", len(vals)) + b"".join(v.to_bytes(32, "little") for v in vals)
    raw = c.wire_fr_encode(curve, mont)
    assert raw == want
    assert np.array_equal(c.wire_fr_decode(curve, raw), mont)
    assert c.wire_fr_encode(curve, mont[:0]) == struct.pack("<Q", 0) and c.wire_fr_decode(curve, struct.pack("<Q", 0)).shape[0] == 0
    for bad in (want[:-1], struct.pack("<Q", len(vals) + 1) + want[8:], want[:8] + F.p.to_bytes(32, "little") + want[40:]):
        with pytest.raises(dg16_amd.Dg16Error):
            c.wire_fr_decode(curve, bad)
