"""The limb-per-lane arithmetic of csrc/lane29.h on the CPU: the lane-level model of its Montgomery products
(tools/lane29_model.py: only operations that exist as one gfx950 instruction on a row of 16 lanes) against big-integer
arithmetic at the extreme limb values, the value bounds of its group law (tools/lane_bounds.py) with the constants the
header uses, and the header's constants themselves.  The GPU side is tests/test_gpu_lane29.py."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import lane29_model as M      # noqa: E402
import lane_bounds as LB      # noqa: E402

HEADER = open(os.path.join(ROOT, "distributed-groth16_amd", "csrc", "lane29.h")).read()


def test_nine_limb_product_against_big_integers():
    worst = M.self_test(M.BN254_Q, iters=1500, seed=11)
    assert worst <= M.LOOSE
    # the scalar field of BN254 has nine limbs as well (not used by the chains; the algorithm does not care)
    M.self_test(M.BN254_R, iters=300, seed=12)


def test_fourteen_limb_product_against_big_integers():
    for q in (M.BLS12_381_Q, M.BLS12_377_Q):
        assert M.self_test14(q, iters=600, seed=13)


def test_carry_out_of_the_low_half_is_exact():
    """B5 of the header: the low columns sum to q R exactly and q is recovered from three pieces of columns 6..8; a product
    whose low half is all ones before the quotient is added is the case a rippling carry would need nine steps for"""
    K = M.Consts(M.BN254_Q)
    Rinv = pow(M.R, -1, M.BN254_Q)
    for a, b in ((M.R - 1, 1), (M.MASK, M.MASK), ((1 << 232) - 1, (1 << 29) - 1), (M.BN254_Q - 1, M.BN254_Q - 1)):
        a %= 7 * M.BN254_Q
        b %= 7 * M.BN254_Q
        got = M.value(M.mont(M.to_row(a), M.to_row(b), K))
        assert got % M.BN254_Q == a * b * Rinv % M.BN254_Q


def test_group_law_bounds_close_with_the_headers_constants():
    for name, (p, w, n) in LB.FIELDS.items():
        rp = (1 << (w * n)) / p
        for ext in (False, True):
            store = 2.04 if (ext and w * n - p.bit_length() < 9) else 7.0
            ks, kneg, nz = LB.LANE29[ext]
            law, acc = LB.closure(rp, ext, store, ks, kneg, LB.BETA[name])       # asserts that every K dominates its subtrahend
            assert law.zero_arg < nz and law.maxv < rp
            if not ext:
                assert acc[1] <= 7.0 and acc[2] <= 7.0 and acc[3] <= 7.0      # y, zz, zzz are stored as they are (exit_pt)


def test_header_constants_are_the_checked_ones():
    def law(ext):
        m = re.search(r"template <> struct LawK<%s> \{(.*?)\n\};" % ("true" if ext else "false"), HEADER, re.S)
        body = m.group(1)
        sub = [int(x) for x in re.search(r"SUB\[4\] = \{([^}]*)\}", body).group(1).split(",")]
        neg = int(re.search(r"NEG = (\d+)", body).group(1))
        zero = int(re.search(r"ZERO = (\d+)", body).group(1))
        return sub, neg, zero
    for ext in (False, True):
        ks, kneg, nz = LB.LANE29[ext]
        sub, neg, zero = law(ext)
        assert sub == [ks["K0"], ks["K1"], ks["K2"], ks["K3"]]
        assert neg == (kneg or 0) and zero == nz
    assert int(re.search(r"constexpr int kSpread = (\d+);", HEADER).group(1)) == 4
