"""The product's field / curve headers (distributed-groth16_amd/csrc/{fp,fp2,ec}.h) instantiated with
the host compiler and compared with the oracle -- catches formula bugs without a GPU.  The gfx950
asm multiply is checked on the GPU (tests/test_gpu_field.py)."""

import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import corc
from oracle.pyref.fields import FQ, FR

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_arith", "host_arith.cpp")
SO = os.path.join(HERE, "host_arith", "libhost_arith.so")


@pytest.fixture(scope="module")
def ha():
    hdrs = [os.path.join(HERE, "..", "distributed-groth16_amd", "csrc", f)
            for f in ("fp.h", "fp2.h", "ec.h", "consts_gen.h", "fp29.h", "ec29.h", "codec_impl.h", "types.h", "glv.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(SO) < os.path.getmtime(p) for p in [SRC] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC])
    L = ctypes.CDLL(SO)
    vp, sz, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.ha_field_op.argtypes = [i, i, vp, vp, vp, sz]
    L.ha_point_op.argtypes = [i, i, i, vp, vp, vp, sz]
    L.ha_field_op29.argtypes = [i, i, vp, vp, vp, sz]
    L.ha_point_op29.argtypes = [i, i, i, vp, vp, vp, sz]
    L.ha_codec.argtypes = [i, i, i, i, vp, vp, sz, vp]
    L.ha_glv_split.argtypes = [i, vp, vp, vp, sz]
    L.ha_glv_split4.argtypes = [i, vp, vp, vp, vp, vp, sz]
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
@pytest.mark.parametrize("kind", ["fq", "fr"])
def test_field_ops(ha, curve, kind):
    F = (FQ if kind == "fq" else FR)[curve]
    nl = F.limbs64
    n = 200
    A = corc.rand_field(curve, kind, 1, n)
    B = corc.rand_field(curve, kind, 2, n)
    edge = corc.ints_to_arr([0, F.R, F.p - 1, 1, F.to_mont(F.p - 1)], nl)
    A[:5] = edge
    B[:5] = edge[::-1]
    for op in ("add", "sub", "mul", "sqr", "neg", "from_mont", "inv"):
        a = A if op != "inv" else A[:12]
        b = B if op != "inv" else B[:12]
        out = np.empty_like(a)
        assert ha.ha_field_op(corc.fid(curve, kind), corc.OPS[op], _p(a), _p(b), _p(out), len(a)) == 0
        exp = corc.field_op(curve, kind, op, a, b)
        assert np.array_equal(out, exp), op
    canon = corc.field_op(curve, kind, "from_mont", A)
    out = np.empty_like(A)
    ha.ha_field_op(corc.fid(curve, kind), corc.OPS["to_mont"], _p(canon), _p(canon), _p(out), n)
    assert np.array_equal(out, A)


@pytest.mark.parametrize("curve,group", [("bn254", 1), ("bn254", 2), ("bls12_381", 1),
                                         ("bls12_381", 2), ("bls12_377", 1), ("bls12_377", 2)])
@pytest.mark.parametrize("rr", [False, True])
def test_point_ops(ha, curve, group, rr):
    """rr = True: the same operations through the reduced-radix types of fp29.h / ec29.h (the representation of the
    bucket kernels: 29/28-bit limbs, lazy bounds, its own Montgomery radix), converted in and out."""
    cid = corc.CURVES[curve]
    point_op = ha.ha_point_op29 if rr else ha.ha_point_op
    rng = random.Random(3)
    r = FR[curve].p
    n = 24
    P = corc.gen_points(curve, group, 5, n)
    Q = corc.gen_points(curve, group, 6, n)
    Q[0] = P[0]            # doubling through the add path
    Q[1] = 0               # identity operand
    P[2] = 0
    P[3] = 0; Q[3] = 0

    def oracle_pairwise(f):
        return np.concatenate([f(P[i:i + 1], Q[i:i + 1]) for i in range(n)])

    out = np.empty_like(P)
    # general add and mixed add must both equal P + Q
    exp = oracle_pairwise(lambda a, b: corc.point_add(curve, group, a, b))
    for op in (0, 1):
        assert point_op(cid, group, op, _p(P), _p(Q), _p(out), n) == 0
        assert np.array_equal(out, exp), op
    # madd with negation: P - Q (index 0 gives the identity)
    negQ = np.concatenate([corc.point_mul(curve, group, Q[i:i + 1], r - 1) for i in range(n)])
    exp = np.concatenate([corc.point_add(curve, group, P[i:i + 1], negQ[i:i + 1]) for i in range(n)])
    assert point_op(cid, group, 2, _p(P), _p(Q), _p(out), n) == 0
    assert np.array_equal(out, exp)
    assert not out[0].any()
    # doubling
    exp = np.concatenate([corc.point_add(curve, group, P[i:i + 1], P[i:i + 1]) for i in range(n)])
    assert point_op(cid, group, 3, _p(P), _p(Q), _p(out), n) == 0
    assert np.array_equal(out, exp)
    # scalar multiplication by 256-bit integers (first 32 bytes of each Q slot hold k)
    K = Q.copy()
    ks = [rng.randrange(r) for _ in range(n)]
    ks[4], ks[5] = 0, 1
    K[:, :4] = corc.ints_to_arr(ks, 4)
    exp = np.concatenate([corc.point_mul(curve, group, P[i:i + 1], ks[i]) for i in range(n)])
    assert point_op(cid, group, 4, _p(P), _p(K), _p(out), n) == 0
    assert np.array_equal(out, exp)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
@pytest.mark.parametrize("kind", ["fq", "fr"])
def test_reduced_radix_field_ops(ha, curve, kind):
    """fp29.h: conversion in (x R32 -> x R), the operation on 29/28-bit limbs with lazy bounds, conversion out
    (canonical) must equal the oracle's result; op 8 runs a long lazy chain through reduce() and is_zero()."""
    F = (FQ if kind == "fq" else FR)[curve]
    nl = F.limbs64
    n = 300
    A = corc.rand_field(curve, kind, 11, n)
    B = corc.rand_field(curve, kind, 12, n)
    edge = corc.ints_to_arr([0, F.R, F.p - 1, 1, F.to_mont(F.p - 1), F.to_mont(1)], nl)
    A[:6] = edge
    B[:6] = edge[::-1]
    for op in ("add", "sub", "mul", "sqr", "neg"):
        out = np.empty_like(A)
        assert ha.ha_field_op29(corc.fid(curve, kind), corc.OPS[op], _p(A), _p(B), _p(out), n) == 0
        assert np.array_equal(out, corc.field_op(curve, kind, op, A, B)), op
    out = np.empty_like(A)
    assert ha.ha_field_op29(corc.fid(curve, kind), 8, _p(A), _p(B), _p(out), n) == 0
    f = lambda op, x, y=None: corc.field_op(curve, kind, op, x, y)   # noqa: E731
    exp = f("sub", f("add", f("mul", f("add", A, B), f("sub", A, B)), f("mul", A, B)), f("sqr", B))
    assert np.array_equal(out, exp)
    # op 9: (a + b)(a - b) - (2 b + a) a through mul_sub (fused for the 9-limb fields) and through mul_add4 with loose limbs
    out = np.empty_like(A)
    assert ha.ha_field_op29(corc.fid(curve, kind), 9, _p(A), _p(B), _p(out), n) == 0
    exp = f("sub", f("mul", f("add", A, B), f("sub", A, B)), f("mul", f("add", f("add", B, B), A), A))
    assert np.array_equal(out, exp)


# ---- arkworks compressed points: csrc/codec_impl.h on the host against the plain-Python encoder -------------------
def _affine_arr(curve, group, pts):
    Fq = FQ[curve]
    nl = Fq.limbs64
    out = np.zeros((len(pts), 2 * nl * group), dtype=np.uint64)
    for k, P in enumerate(pts):
        if P is None:
            continue
        co = [P[0], P[1]] if group == 1 else [P[0][0], P[0][1], P[1][0], P[1][1]]
        out[k] = corc.ints_to_arr([Fq.to_mont(v) for v in co], nl).reshape(-1)
    return out


def _codec(ha, curve, group, decode, validate, data, n, out_bytes):
    out = np.zeros(out_bytes, dtype=np.uint8)
    rc = np.zeros(n, dtype=np.int32)
    inp = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    assert ha.ha_codec({"bn254": 0, "bls12_381": 1, "bls12_377": 2}[curve], group, decode, validate, _p(inp), _p(out), n, _p(rc)) == 0
    return out, rc


@pytest.mark.parametrize("curve", ["bn254", "bls12_377"])
@pytest.mark.parametrize("group", [1, 2])
def test_point_codec_host(ha, curve, group):
    """encode == the Python encoder byte for byte; decode(encode(P)) == P (both signs, identity, small multiples whose
    square roots take different Tonelli-Shanks paths); x off the curve, unreduced coordinate, both flags, and a curve
    point outside the order-r subgroup (Validate::Yes only) are refused with the codec's codes."""
    import ark_points_py as A
    from oracle.pyref.curves import CURVES
    C = CURVES[curve, "g%d" % group]
    rng = random.Random(11 * group)
    n = 14 if curve == "bn254" else 8
    pts = [C.mul(C.gen, rng.randrange(1, C.order)) for _ in range(n)] + [C.mul(C.gen, k) for k in (1, 2, 3)]
    pts += [None, C.neg(pts[0]), pts[0]]
    fb = A.fbytes(curve)
    cb, pb = fb * group, 2 * fb * group
    arr = _affine_arr(curve, group, pts)
    want = b"".join(A.encode(curve, group, P) for P in pts)
    got, _ = _codec(ha, curve, group, 0, 0, arr.view(np.uint8).reshape(-1), len(pts), cb * len(pts))
    assert got.tobytes() == want
    back, rc = _codec(ha, curve, group, 1, 1, want, len(pts), pb * len(pts))
    assert not rc.any()
    assert np.array_equal(back.view(np.uint64).reshape(len(pts), -1), arr)
    q = C.F.p
    bads = [((q + 1).to_bytes(fb, "little") + bytes(cb - fb), 2)]
    both = bytearray(cb)
    both[-1] = 0xC0
    bads.append((bytes(both), 1))
    inf_x = bytearray(cb)
    inf_x[0], inf_x[-1] = 1, 0x40
    bads.append((bytes(inf_x), 1))                    # stricter than arkworks 0.4, documented in include/dg16.h
    if group == 1:
        bads.append((A.x_off_curve(curve).to_bytes(fb, "little"), 3))
    for raw, code in bads:
        _, rc = _codec(ha, curve, group, 1, 0, raw, 1, pb)
        assert rc[0] == code, (raw.hex(), rc[0], code)
    outside = A.twist_point_outside_subgroup(curve) if group == 2 else \
        (A.g1_point_outside_subgroup(curve) if curve != "bn254" else None)
    if outside is not None:
        raw = A.encode(curve, group, outside)
        dec, rc = _codec(ha, curve, group, 1, 0, raw, 1, pb)
        assert rc[0] == 0 and np.array_equal(dec.view(np.uint64).reshape(1, -1), _affine_arr(curve, group, [outside]))
        _, rc = _codec(ha, curve, group, 1, 1, raw, 1, pb)
        assert rc[0] == 4


@pytest.mark.parametrize("group", [1, 2])
def test_point_codec_host_bls12_381_zcash_form(ha, group):
    """BLS12-381 in the encoding ark-bls12-381 0.4 uses (zcash / IETF: big-endian, flags in the first byte).  The
    generators encode to the published strings (known answers from outside this repository), random points to the
    independent Python encoder's bytes; decode is the inverse (both signs, identity); an encoding without the
    "compressed" bit, both flags, a flagged infinity with x != 0, an unreduced x, an x off the curve and a curve point
    outside the order-r subgroup (Validate::Yes) are refused with the codec's codes."""
    import ark_points_py as A
    from oracle.pyref.curves import CURVES
    curve = "bls12_381"
    C = CURVES[curve, "g%d" % group]
    rng = random.Random(5 * group)
    pts = [C.gen] + [C.mul(C.gen, rng.randrange(1, C.order)) for _ in range(8)] + [C.mul(C.gen, k) for k in (2, 3)]
    pts += [None, C.neg(pts[1]), pts[1]]
    fb = 48
    cb, pb = fb * group, 2 * fb * group
    arr = _affine_arr(curve, group, pts)
    want = b"".join(A.encode_zcash(group, P) for P in pts)
    assert want[:cb] == (A.ZCASH_G1_GENERATOR if group == 1 else A.ZCASH_G2_GENERATOR)
    got, _ = _codec(ha, curve, group, 0, 0, arr.view(np.uint8).reshape(-1), len(pts), cb * len(pts))
    assert got.tobytes() == want
    back, rc = _codec(ha, curve, group, 1, 1, want, len(pts), pb * len(pts))
    assert not rc.any()
    assert np.array_equal(back.view(np.uint64).reshape(len(pts), -1), arr)
    q = C.F.p
    gen = bytearray(want[:cb])
    bads = []
    unc = bytearray(gen)
    unc[0] &= 0x7F
    bads.append((bytes(unc), 1))                                           # not flagged as compressed
    bads.append((bytes([0xE0]) + bytes(cb - 1), 1))                        # infinity + sign
    bads.append((bytes([0xC0]) + bytes(cb - 2) + b"\x01", 1))             # infinity with x != 0
    big = bytearray((q + 1).to_bytes(fb, "big") + bytes(cb - fb))
    big[0] |= 0x80
    bads.append((bytes(big), 2))                                           # coordinate not reduced (q + 1 < 2^381 fits under the flags)
    if group == 1:
        off = bytearray(A.x_off_curve(curve).to_bytes(fb, "big"))
        off[0] |= 0x80
        bads.append((bytes(off), 3))
    for raw, code in bads:
        _, rc = _codec(ha, curve, group, 1, 0, raw, 1, pb)
        assert rc[0] == code, (raw.hex(), rc[0], code)
    outside = A.twist_point_outside_subgroup(curve) if group == 2 else A.g1_point_outside_subgroup(curve)
    raw = A.encode_zcash(group, outside)
    dec, rc = _codec(ha, curve, group, 1, 0, raw, 1, pb)
    assert rc[0] == 0 and np.array_equal(dec.view(np.uint64).reshape(1, -1), _affine_arr(curve, group, [outside]))
    _, rc = _codec(ha, curve, group, 1, 1, raw, 1, pb)
    assert rc[0] == 4


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377", "bn254_g2"])
def test_glv_split_is_exact_and_short(ha, curve):
    """csrc/glv.h: k = k1 + k2 LAMBDA (mod r) for every scalar, the halves equal the same integer arithmetic done with
    Python integers on the constants of the header (division-free rounding included), and |k1|, |k2| < 2^127 -- the
    bound kGlvBits of msm_impl.h rests on -- over edge values, every 255-bit corner and 200 000 random scalars."""
    import re
    hdr = os.path.join(HERE, "..", "distributed-groth16_amd", "csrc", "consts_gen.h")
    text = open(hdr).read()
    blk = text[text.index("struct %s_glv_consts {" % curve):]
    blk = blk[:blk.index("\n};")]
    val = {}
    for m in re.finditer(r"static constexpr uint32_t (\w+)\[\d+\] = \{([^}]*)\}", blk):
        val[m.group(1)] = sum(int(x.strip().rstrip("u"), 16) << (32 * i) for i, x in enumerate(m.group(2).split(",")))
    sign = {m.group(1): -1 if m.group(2) == "true" else 1 for m in re.finditer(r"static constexpr bool (\w+)_NEG = (\w+);", blk)}
    a1, b1, a2, b2 = (sign[k] * val[k] for k in ("A1", "B1", "A2", "B2"))
    lam, r = val["LAMBDA"], FR[curve.split("_g2")[0]].p
    rng = random.Random(17)
    ks = [0, 1, 2, r - 1, r - 2, (r - 1) // 2, (r + 1) // 2, lam, r - lam, lam - 1, (1 << 254) - 1, (1 << 255) - 1, 1 << 254,
          val["B2"], val["B1"], r // 3, 2 * r // 3]
    ks += [rng.randrange(r) for _ in range(200000)] + [rng.randrange(1 << 255) for _ in range(20000)]
    K = corc.ints_to_arr(ks, 4)
    H1, H2 = np.zeros_like(K), np.zeros_like(K)
    assert ha.ha_glv_split({"bn254": 0, "bls12_381": 1, "bls12_377": 2, "bn254_g2": 3}[curve], _p(K), _p(H1), _p(H2), len(ks)) == 0
    g1, g2 = corc.arr_to_ints(H1), corc.arr_to_ints(H2)
    worst = 0
    for k, x1, x2 in zip(ks, g1, g2):
        c1, c2 = (val["G1"] * k + (1 << 255)) >> 256, (val["G2"] * k + (1 << 255)) >> 256
        k1, k2 = k - c1 * a1 - c2 * a2, -c1 * b1 - c2 * b2
        assert (k1 + k2 * lam - k) % r == 0
        dec = lambda x: -(x & ((1 << 255) - 1)) if x >> 255 else x      # noqa: E731
        assert (dec(x1), dec(x2)) == (k1, k2), hex(k)
        worst = max(worst, abs(k1), abs(k2))
    assert worst < 1 << 127, worst.bit_length()


@pytest.mark.parametrize("curve", ["bls12_381", "bls12_377"])
def test_glv_split4_is_exact_and_short(ha, curve):
    """csrc/glv.h: the four-dimensional split of G2 scalars -- k = sum_j k_j LAMBDA^j (mod r) for every scalar, the quarters
    equal the same integer arithmetic done with Python integers on the constants of the header, and |k_j| < 2^65 (the
    bound kGlv4Bits of msm_impl.h rests on)."""
    import re
    hdr = os.path.join(HERE, "..", "distributed-groth16_amd", "csrc", "consts_gen.h")
    text = open(hdr).read()
    blk = text[text.index("struct %s_g2_glv4_consts {" % curve):]
    blk = blk[:blk.index("\n};")]
    words = lambda body: [int(x.strip().rstrip("u"), 16) for x in body.split(",")]          # noqa: E731
    val = lambda ws: sum(v << (32 * i) for i, v in enumerate(ws))                             # noqa: E731
    lam = val(words(re.search(r"LAMBDA\[8\] = \{([^}]*)\}", blk).group(1)))
    Bm = [val(words(m)) for m in re.findall(r"\{([^{}]*)\}", re.search(r" B\[16\]\[3\] = \{(.*)\};", blk).group(1))]
    Bn = [x.strip() == "true" for x in re.search(r"B_NEG\[16\] = \{([^}]*)\}", blk).group(1).split(",")]
    Gm = [val(words(m)) for m in re.findall(r"\{([^{}]*)\}", re.search(r" G\[4\]\[7\] = \{(.*)\};", blk).group(1))]
    Cn = [x.strip() == "true" for x in re.search(r"C_NEG\[4\] = \{([^}]*)\}", blk).group(1).split(",")]
    B = [[(-Bm[4 * i + j] if Bn[4 * i + j] else Bm[4 * i + j]) for j in range(4)] for i in range(4)]
    r = FR[curve].p
    assert (lam ** 4 - lam ** 2 + 1) % r == 0
    rng = random.Random(23)
    ks = [0, 1, 2, r - 1, r - 2, (r - 1) // 2, lam, r - lam, (1 << 254) - 1, (1 << 255) - 1, 1 << 254, r // 3]
    ks += [rng.randrange(r) for _ in range(100000)] + [rng.randrange(1 << 255) for _ in range(10000)]
    K = corc.ints_to_arr(ks, 4)
    H = [np.zeros_like(K) for _ in range(4)]
    assert ha.ha_glv_split4({"bn254": 0, "bls12_381": 1, "bls12_377": 2}[curve], _p(K), *[_p(h) for h in H], len(ks)) == 0
    got = [corc.arr_to_ints(h) for h in H]
    dec = lambda x: -(x & ((1 << 255) - 1)) if x >> 255 else x      # noqa: E731
    worst = 0
    for n_, k in enumerate(ks):
        c = [(-1 if Cn[i] else 1) * ((Gm[i] * k + (1 << 255)) >> 256) for i in range(4)]
        q = [(k if j == 0 else 0) - sum(c[i] * B[i][j] for i in range(4)) for j in range(4)]
        assert sum(q[j] * lam ** j for j in range(4)) % r == k % r
        assert [dec(got[j][n_]) for j in range(4)] == q, hex(k)
        worst = max(worst, max(abs(x) for x in q))
    assert worst < 1 << 65, worst.bit_length()
