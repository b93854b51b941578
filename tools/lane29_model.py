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
