#!/usr/bin/env python3
"""Lane-level model of the limb-per-lane Montgomery product of csrc/lane29.h (round 6).

A field element of a 9-limb reduced-radix field (fp29.h: W = 29, R = 2^261) is ONE 32-bit register: lane i of a row of 16
lanes holds limb i (lanes 9..15 hold zero).  A product is then column-parallel: lane l owns column l of a b, the
Montgomery quotient m = T p' mod R and m p are column-parallel too (the NON-interleaved reduction: three half-size
products instead of a chain of nine dependent steps), and carries never ripple: every 64-bit column is cut into three
pieces (bits 0-28, 29-57, 58-63) that are added to the lanes above with row shifts, which leaves "loose" limbs
< 2^30 + 2^7 -- small enough for the next product's columns.  The exact carry out of the (zero mod R) low half is
recovered from three pieces of columns 6..8 (see `mont`).

This file is the executable statement of that algorithm: `mont` uses only operations that exist as one gfx950
instruction on a row (v_mad_u64_u32, v_and, v_alignbit, v_lshrrev, v_add_u32 with a DPP row shift, v_mov_b32_dpp
row_newbcast / row_shr / row_shl / row_ror), and `tests/test_lane29_model.py` checks it against big-integer arithmetic
at the extreme limb values the bounds below allow.  Test infrastructure; the product is csrc/lane29.h.
"""
import random

W, N, ROW = 29, 9, 16
MASK = (1 << W) - 1
R = 1 << (W * N)
M32 = (1 << 32) - 1
M64 = (1 << 64) - 1

BN254_Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617

# loose-limb bound the product accepts and produces (checked by the asserts in mont and by the test)
LOOSE = (1 << 30) + (1 << 8)
TIGHT = (1 << 29) + (1 << 8)      # after one renorm pass; a dual product needs one tight operand per pair


def limbs(x, n=N):
    return [(x >> (W * i)) & MASK for i in range(n)]


def value(v):
    return sum(int(x) << (W * i) for i, x in enumerate(v))


def to_row(x):
    return limbs(x) + [0] * (ROW - N)


# ---- row primitives (one instruction each) -----------------------------------------------------------------------------
def bc(v, i):          # v_mov_b32_dpp row_newbcast:i
    return [v[i]] * ROW


def shr(v, i):         # v_mov_b32_dpp row_shr:i bound_ctrl:0  -- lane l reads lane l - i, zero when out of the row
    return [v[l - i] if l - i >= 0 else 0 for l in range(ROW)]


def shl(v, i):         # row_shl:i bound_ctrl:0 -- lane l reads lane l + i
    return [v[l + i] if l + i < ROW else 0 for l in range(ROW)]


def ror(v, i):         # row_ror:i -- lane l reads lane (l - i) mod 16
    return [v[(l - i) % ROW] for l in range(ROW)]


def mad(a, b, c):      # v_mad_u64_u32, per lane
    out = []
    for x, y, z in zip(a, b, c):
        assert 0 <= x <= M32 and 0 <= y <= M32
        r = x * y + z
        assert r <= M64, "column overflow"
        out.append(r)
    return out


def pieces(c):         # 64-bit column -> bits 0..28, 29..57, 58..63 (v_and; v_alignbit + v_and; v_lshrrev on the high word)
    return [x & MASK for x in c], [(x >> W) & MASK for x in c], [x >> (2 * W) for x in c]


def add(*vs):
    out = [sum(t) for t in zip(*vs)]
    assert all(x <= M32 for x in out), "limb overflow"
    return out


class Consts:
    """Per-field constant registers of a row: p and p' = -p^-1 mod R, pre-shifted per multiplier limb."""

    def __init__(self, p):
        self.p = p
        pl = limbs(p)
        ppl = limbs((-pow(p, -1, R)) % R)
        # PS[i][l] = p_(l - i): column l of m_i p;  PP[i][l] = p'_(l - i) for l < 9 only (m is taken mod R)
        self.PS = [[pl[l - i] if 0 <= l - i < N else 0 for l in range(ROW)] for i in range(N)]
        self.PP = [[ppl[l - i] if 0 <= l - i < N and l < N else 0 for l in range(ROW)] for i in range(N)]
        self.P16 = [pl[8]] + [0] * (ROW - 1)            # column 16 = m_8 p_8, kept in lane 0 of a second accumulator
        self.LANE8 = [M32 if l == 8 else 0 for l in range(ROW)]
        self.LO7 = [l < 7 for l in range(ROW)]
        self.LIMB = [l < N for l in range(ROW)]


def mont(a, b, K, c=None, d=None):
    """(a b [+ c d]) / R mod p on rows: returns a row with loose limbs (< LOOSE), value < T / R + 2.2 p.
    Operation count (single product): 9 bcast + 9 shift + 10 mad | 6 | 9 bcast + 9 mad | 6 | 9 bcast + 10 mad | ~21."""
    zero = [0] * ROW
    main, hi = zero, zero
    for x, y in ((a, b),) + (((c, d),) if c is not None else ()):
        assert all(v <= LOOSE for v in x) and all(v <= (LOOSE if c is None else TIGHT) for v in y)
        assert all(v == 0 for v in x[N:]) and all(v == 0 for v in y[N:])
        for i in range(N):
            main = mad(bc(x, i), shr(y, i), main)       # column l += x_i y_(l - i)
        hi = mad(bc(x, 8), shl(y, 8), hi)               # column 16 = x_8 y_8 (lane 0 only)
    # B1: T mod R as loose limbs (pieces flowing out of column 8 land in lanes 9, 10 and are never read)
    p0, p1, p2 = pieces(main)
    t = add(p0, shr(p1, 1), shr(p2, 2))
    # B2: m = t p' mod R, columns 0..8
    mcol = zero
    for i in range(N):
        mcol = mad(bc(t, i), K.PP[i], mcol)
    # B3: loose limbs of m (any representative of m mod R serves)
    q0, q1, q2 = pieces(mcol)
    m = add(q0, shr(q1, 1), shr(q2, 2))
    # B4: T + m p
    for i in range(N):
        main = mad(bc(m, i), K.PS[i], main)
    hi = mad(bc(m, 8), K.P16, hi)
    # B5: columns 0..8 now sum to q R exactly; the carry into column 9 is the three-piece carry plus e in {0..3}
    p0, p1, p2 = pieces(main)
    L = add(p0, shr(p1, 1), shr(p2, 2))
    e = [((x + 2) >> W) & K.LANE8[l] for l, x in enumerate(L)]
    L = add(L, shr(e, 1))
    # B6: limbs 0..6 = columns 9..15 (rotate down), limbs 7, 8 from column 16 and the pieces that left the row
    h0, h1, h2 = pieces(hi)
    assert all(x == 0 for x in h2), "result does not fit 261 bits"
    top = add(shr(h0, 7), shr(h1, 8), shl(p1, 8), shl(p2, 7))
    rot = ror(L, 7)
    res = [(rot[l] if K.LO7[l] else top[l]) if K.LIMB[l] else 0 for l in range(ROW)]
    assert all(v <= LOOSE for v in res)
    return res


def renorm(v):
    """One parallel carry pass: limbs < 2^29 + (max >> 29) afterwards (and, lshr, add with row_shr:1)."""
    lo = [x & MASK for x in v]
    hi = [x >> W for x in v]
    hi[8] = 0 if False else hi[8]
    out = add(lo, shr(hi, 1))
    # the top limb keeps its own carry (a value below 2^261 has none to lose): fold lane 9 back
    out[8] = (v[8] & MASK) + (v[8] >> W << W) + (v[7] >> W)
    out[9] = 0
    return out


def self_test(p=BN254_Q, iters=2000, seed=1):
    rng = random.Random(seed)
    K = Consts(p)
    Rinv = pow(R, -1, p)
    worst = 0
    for it in range(iters):
        def rnd(top=LOOSE):
            mode = rng.randrange(4)
            if mode == 0:
                x = rng.randrange(7 * p)
                return to_row(x)
            if mode == 1:       # extreme loose limbs, small top limb (value < 2^261 must hold: top limb < 2^25)
                return [top] * 8 + [rng.randrange(1 << 24)] + [0] * 7
            if mode == 2:
                return [rng.choice((0, 1, MASK, top, top - 1)) for _ in range(8)] + [rng.randrange(1 << 24)] + [0] * 7
            return to_row(rng.choice((0, 1, p - 1, p, 7 * p - 1)))
        a, b = rnd(), rnd()
        if it % 3 == 0:
            b, c, d = rnd(TIGHT), rnd(), rnd(TIGHT)
            r = mont(a, b, K, c, d)
            T = value(a) * value(b) + value(c) * value(d)
        else:
            r = mont(a, b, K)
            T = value(a) * value(b)
        got = value(r)
        assert got % p == T * Rinv % p, (it, a, b)
        assert got < T // R + 3 * p
        worst = max(worst, max(r))
    return worst


if __name__ == "__main__":
    w = self_test()
    print("ok; largest output limb 2^%.3f" % (__import__("math").log2(w)))


# =========================================================================================================================
# Fourteen limbs of 28 bits (BLS12-377 / BLS12-381 Fq): an element is TWO registers on one row -- `lo` holds limbs 0..6
# on lanes 0..6, `hi` limbs 7..13 on lanes 0..6 -- so that nothing ever crosses a row: the product is three column sets
# C0 = lo lo (columns 0..12), C1 = lo hi + hi lo (columns 7..19), C2 = hi hi (columns 14..26), each on lanes 0..12, and the
# Montgomery reduction runs digit-serially over the two digits of base B = 2^196: m_j = (digit j) p'0 mod B,
# T += m_j p B^j, carry the (now zero mod B) digit out -- each step column-parallel on seven lanes.
W14, N14, H14 = 28, 14, 7
MASK14 = (1 << W14) - 1
B14 = 1 << (W14 * H14)
R14 = B14 * B14
BLS12_381_Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
BLS12_377_Q = 0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001
LOOSE14 = (1 << 29) + (1 << 10)
TIGHT14 = (1 << 28) + (1 << 4)


def limbs14(x):
    return [(x >> (W14 * i)) & MASK14 for i in range(N14)]


def to_rows14(x):
    l = limbs14(x)
    return (l[:7] + [0] * 9, l[7:] + [0] * 9)


def value14(e):
    lo, hi = e
    return sum(int(v) << (W14 * i) for i, v in enumerate(lo[:9])) + (sum(int(v) << (W14 * i) for i, v in enumerate(hi[:9])) << (W14 * 7))


def pieces14(c):
    return [x & MASK14 for x in c], [(x >> W14) & MASK14 for x in c], [x >> (2 * W14) for x in c]


def three14(c):
    p0, p1, p2 = pieces14(c)
    return add(p0, shr(p1, 1), shr(p2, 2))


class Consts14:
    def __init__(self, p):
        self.p = p
        pl = limbs14(p)
        pp0 = limbs14((-pow(p, -1, B14)) % B14)[:7]
        z = lambda f: [[f(l, i) for l in range(ROW)] for i in range(7)]
        self.PL = z(lambda l, i: pl[l - i] if 0 <= l - i < 7 else 0)
        self.PH = z(lambda l, i: pl[7 + l - i] if 0 <= l - i < 7 else 0)
        self.PP = z(lambda l, i: pp0[l - i] if 0 <= l - i < 7 and l < 7 else 0)
        self.LANE6 = [M32 if l == 6 else 0 for l in range(ROW)]
        self.ONE = [1] * ROW


def renorm14(e):
    lo, hi = e
    lo2 = add([x & MASK14 for x in lo], shr([x >> W14 for x in lo], 1))          # lane 7: the carry out of limb 6
    hi2 = add([x & MASK14 for x in hi], shr([x >> W14 for x in hi], 1), shl(lo2, 7))
    lo2 = [v if l < 7 else 0 for l, v in enumerate(lo2)]
    return lo2, hi2


def mont14(a, b, K):
    """a b / R mod p, R = 2^392; a: limbs <= LOOSE14, b: limbs <= TIGHT14 ... LOOSE14 (see the asserts), values < 2^392"""
    zero = [0] * ROW
    (al, ah), (bl, bh) = a, b
    for v in al + ah + bl + bh:
        assert v <= LOOSE14
    for r in (al, ah, bl, bh):
        assert all(v == 0 for v in r[7:])
    C0, C1, C2 = zero, zero, zero
    for i in range(7):
        C0 = mad(bc(al, i), shr(bl, i), C0)
        C1 = mad(bc(al, i), shr(bh, i), C1)
        C1 = mad(bc(ah, i), shr(bl, i), C1)
        C2 = mad(bc(ah, i), shr(bh, i), C2)

    def digit(Cl, Cm):
        t = three14(Cl)                                   # lanes 0..6: the digit, loose
        mc = zero
        for i in range(7):
            mc = mad(bc(t, i), K.PP[i], mc)
        m = three14(mc)                                   # lanes 0..6 (7, 8: spill, never read)
        for i in range(7):
            Cl = mad(bc(m, i), K.PL[i], Cl)
            Cm = mad(bc(m, i), K.PH[i], Cm)
        L = three14(Cl)
        e = [((x + 2) >> W14) & K.LANE6[l] for l, x in enumerate(L)]
        L = add(L, shr(e, 1))
        U = shl(L, 7)                                     # the digit is zero mod B: what is left starts at lane 7
        Cm = mad(U, K.ONE, Cm)
        return Cm
    C1 = digit(C0, C1)
    C2 = digit(C1, C2)
    res = three14(C2)
    assert all(v == 0 for v in res[14:]), "result does not fit 392 bits"
    lo = [v if l < 7 else 0 for l, v in enumerate(res)]
    hi = shl(res, 7)
    out = renorm14((lo, hi))
    assert all(v <= TIGHT14 for v in out[0] + out[1])
    return out


def self_test14(p, iters=1500, seed=2):
    rng = random.Random(seed)
    K = Consts14(p)
    Rinv = pow(R14, -1, p)
    top = (16 * p) >> (W14 * 13)
    for it in range(iters):
        def rnd():
            mode = rng.randrange(4)
            if mode == 0:
                return to_rows14(rng.randrange(7 * p))
            if mode == 1:
                return ([LOOSE14] * 7 + [0] * 9, [LOOSE14] * 6 + [rng.randrange(top)] + [0] * 9)
            if mode == 2:
                return ([rng.choice((0, 1, MASK14, LOOSE14)) for _ in range(7)] + [0] * 9,
                        [rng.choice((0, 1, MASK14, LOOSE14)) for _ in range(6)] + [rng.randrange(top)] + [0] * 9)
            return to_rows14(rng.choice((0, 1, p - 1, p, 7 * p - 1)))
        a, b = rnd(), rnd()
        r = mont14(a, b, K)
        T = value14(a) * value14(b)
        got = value14(r)
        assert got % p == T * Rinv % p, (it,)
        assert got < T // R14 + 3 * p
    return True


if __name__ == "__main__":
    for q in (BLS12_381_Q, BLS12_377_Q):
        self_test14(q)
    print("ok: fourteen-limb product (two registers per element)")
