"""Test infrastructure: the arkworks DEFAULT short-Weierstrass compressed point encoding (ark-ec 0.4 `SWFlags`:
little-endian x, bit 7 of the last byte = "y is the larger of (y, -y)", bit 6 = infinity; Fq2 = c0 || c1, ordered on
(c1, c0)) in plain Python integers over the oracle's curve objects -- generic in the curve, so it also covers
BLS12-377 (ark-bls12-377), which distributed-groth16_amd/serialize.py (BN254 proof.bin only) does not.  Encoding
needs no square root; the decoder side of the tests is a round trip plus hand-made invalid encodings."""

from oracle.pyref.curves import CURVES


def fbytes(curve):
    return (CURVES[curve, "g1"].F.p.bit_length() + 7) // 8


def _neg_is_smaller(p, y):
    return y > (-y) % p


def encode(curve, group, P):
    C = CURVES[curve, "g%d" % group]
    p, fb = C.F.p, fbytes(curve)
    out = bytearray(fb * group)
    if P is None:
        out[-1] |= 0x40
        return bytes(out)
    x, y = P
    if group == 1:
        out[:] = x.to_bytes(fb, "little")
        neg = _neg_is_smaller(p, y)
    else:
        out[:fb] = x[0].to_bytes(fb, "little")
        out[fb:] = x[1].to_bytes(fb, "little")
        neg = _neg_is_smaller(p, y[1]) if y[1] != 0 else _neg_is_smaller(p, y[0])
    if neg:
        out[-1] |= 0x80
    return bytes(out)


def encode_zcash(group, P):
    """ark-bls12-381 0.4 (curves/util.rs) = the zcash / IETF BLS12-381 encoding: big-endian x (G2: x.c1 || x.c0), flag
    bits in the first byte: 0x80 compressed, 0x40 infinity, 0x20 y is the lexicographically larger of (y, -y)."""
    C = CURVES["bls12_381", "g%d" % group]
    p, fb = C.F.p, 48
    if P is None:
        return bytes([0xC0]) + bytes(fb * group - 1)
    x, y = P
    if group == 1:
        out = bytearray(x.to_bytes(fb, "big"))
        big = _neg_is_smaller(p, y)
    else:
        out = bytearray(x[1].to_bytes(fb, "big") + x[0].to_bytes(fb, "big"))
        big = _neg_is_smaller(p, y[1]) if y[1] != 0 else _neg_is_smaller(p, y[0])
    out[0] |= 0xA0 if big else 0x80
    return bytes(out)


# The generators in that encoding as every BLS12-381 implementation prints them (zcash "BLS12-381 for the rest of us" /
# draft-irtf-cfrg-pairing-friendly-curves, appendix "ZCash serialization format"): known answers from outside this repo.
ZCASH_G1_GENERATOR = bytes.fromhex(
    "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
ZCASH_G2_GENERATOR = bytes.fromhex(
    "93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
    "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")


def sqrt_fq(p, a):
    """Tonelli-Shanks with a brute-force non-residue (any p)."""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    s, t = 0, p - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    z = next(z for z in range(2, 1000) if pow(z, (p - 1) // 2, p) == p - 1)
    c, x, b, m = pow(z, t, p), pow(a, (t + 1) // 2, p), pow(a, t, p), s
    while b != 1:
        k, b2 = 0, b
        while b2 != 1:
            b2, k = b2 * b2 % p, k + 1
        w = pow(c, 1 << (m - k - 1), p)
        c, x, b, m = w * w % p, x * w % p, b * w * w % p, k
    return x


def sqrt_fq2(F2, a):
    """sqrt in Fq[u] / (u^2 - nr) through the norm; checked by squaring."""
    p, nr = F2.p, F2.nr
    a0, a1 = a
    if a1 == 0:
        r = sqrt_fq(p, a0)
        if r is not None:
            return (r, 0)
        r = sqrt_fq(p, a0 * pow(nr, p - 2, p) % p)
        return None if r is None else (0, r)
    n = sqrt_fq(p, (a0 * a0 - nr * a1 * a1) % p)
    if n is None:
        return None
    inv2 = (p + 1) // 2
    for d in ((a0 + n) * inv2 % p, (a0 - n) * inv2 % p):
        x0 = sqrt_fq(p, d)
        if x0:
            cand = (x0, a1 * pow(2 * x0, p - 2, p) % p)
            if F2.sqr(cand) == (a0 % p, a1 % p):
                return cand
    return None


def x_off_curve(curve):
    """smallest x >= 2 in Fq with x^3 + b a non-square (G1)."""
    C = CURVES[curve, "g1"]
    p = C.F.p
    return next(x for x in range(2, 500) if pow((x ** 3 + C.b) % p, (p - 1) // 2, p) == p - 1)


def twist_point_outside_subgroup(curve):
    C = CURVES[curve, "g2"]
    F2 = C.F
    x = (1, 0)
    while True:
        y = sqrt_fq2(F2, F2.add(F2.mul(F2.sqr(x), x), C.b))
        if y is not None and C.mul((x, y), C.order) is not None:
            return (x, y)
        x = (x[0] + 1, 0)


def g1_point_outside_subgroup(curve):
    C = CURVES[curve, "g1"]
    p = C.F.p
    x = 1
    while True:
        y = sqrt_fq(p, (x ** 3 + C.b) % p)
        if y is not None and C.mul((x, y), C.order) is not None:
            return (x, y)
        x += 1
