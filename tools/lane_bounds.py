#!/usr/bin/env python3
"""Value bounds of the limb-per-lane group law (csrc/lane29.h), by fixed-point iteration.

Values live in [0, B p) and are never compared or conditionally subtracted between products; a Montgomery product of
values below A p and B p returns less than (A B p / R + 2.01) p (the quotient m is used with loose limbs: m < 2.0003 R),
a subtraction a - b adds K p with K > bound(b).  This script walks the formulas of dbl_pt / add_pt exactly as lane29.h
states them, starting from the storage bound of the inputs, until the bounds of a chain's running point stop growing,
and prints the multiples K the subtractions need and the number of multiples of p a zero test must compare against.
`tests/test_lane29_model.py` pins the constants of lane29.h against this output.

usage: python tools/lane_bounds.py
"""
import math

FIELDS = {
    # name: (p, limb bits, limbs)
    "bn254_fq": (21888242871839275222246405745257275088696311157297823662689037894645226208583, 29, 9),
    "bls12_381_fq": (0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab, 28, 14),
    "bls12_377_fq": (0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001, 28, 14),
}
# the quadratic extensions are Fq[u] / (u^2 + BETA): a product's c0 = a0 b0 + a1 (NEG BETA p - BETA b1)
BETA = {"bn254_fq": 1, "bls12_381_fq": 1, "bls12_377_fq": 5}


# the constants lane29.h uses: (kSubKs, negation multiple, multiples of p a zero test compares against)
LANE29 = {
    False: ({"K0": 4, "K1": 6, "K2": 9, "K3": 13}, None, 7),     # base fields (all three curves)
    True: ({"K0": 4, "K1": 8, "K2": 10, "K3": 14}, 19, 8),       # quadratic extension: the smallest set that closes for
}                                                                #  BN254 (R / p = 169, one negation multiple for every product)


class Law:
    def __init__(self, rp, ext, store, fixed=None, kneg_fixed=None, beta=1):
        self.rp, self.ext, self.store, self.beta = rp, ext, store, beta
        self.fixed, self.kneg_fixed = fixed, kneg_fixed
        self.need = {}          # name of a subtraction -> largest subtrahend bound seen
        self.kneg = 0.0         # largest bound of a negated product operand (quadratic extension)
        self.maxv = 0.0

    def see(self, v):
        self.maxv = max(self.maxv, v)
        return v

    def mul(self, a, b):
        """product bound; for the quadratic extension the SECOND operand's c1 is negated: c0 = a0 b0 + a1 (K - b1)"""
        if not self.ext:
            return self.see(a * b / self.rp + 2.01)
        k = math.floor(b) + 1
        self.kneg = max(self.kneg, b)
        if self.kneg_fixed is not None:
            assert b < self.kneg_fixed, "negation constant too small: %.2f" % b
            k = self.kneg_fixed
        return self.see(max((a * b + a * k * self.beta) / self.rp, 2 * a * b / self.rp) + 2.01)

    def sub(self, name, a, b):
        self.need[name] = max(self.need.get(name, 0.0), b)
        if self.fixed is not None:
            assert b < self.fixed[name], "%s = %d too small for a subtrahend below %.2f p" % (name, self.fixed[name], b)
            return self.see(a + self.fixed[name])
        return self.see(a + math.floor(self.need[name]) + 1)

    def dbl(self, P):
        X, Y, ZZ, ZZZ = P
        u = 2 * Y
        v = self.mul(u, u)
        xx = self.mul(X, X)
        m = 3 * xx
        w = self.mul(u, v)
        s = self.mul(X, v)
        mm = self.mul(m, m)
        zz3 = self.mul(ZZ, v)
        x3 = self.sub("K1", mm, 2 * s)
        sx = self.sub("K3", s, x3)
        t0 = self.mul(sx, m)
        t1 = self.mul(w, Y)
        zzz3 = self.mul(ZZZ, w)
        y3 = self.sub("K0", t0, t1)
        return (x3, y3, zz3, zzz3)

    def add(self, P, O):
        X1, Y1, ZZ1, ZZZ1 = P
        X2, Y2, ZZ2, ZZZ2 = O
        u1 = self.mul(X1, ZZ2)
        u2 = self.mul(X2, ZZ1)
        s1 = self.mul(Y1, ZZZ2)
        s2 = self.mul(Y2, ZZZ1)
        pd = self.sub("K0", u2, u1)
        rd = self.sub("K0", s2, s1)
        self.zero_arg = max(getattr(self, "zero_arg", 0.0), pd, rd)
        pp = self.mul(pd, pd)
        rr = self.mul(rd, rd)
        zzp = self.mul(ZZ1, ZZ2)
        zzzp = self.mul(ZZZ1, ZZZ2)
        ppp = self.mul(pd, pp)
        q = self.mul(u1, pp)
        zz3 = self.mul(zzp, pp)
        x3 = self.sub("K2", rr, ppp + 2 * q)
        qx = self.sub("K3", q, x3)
        t0 = self.mul(qx, rd)
        t1 = self.mul(s1, ppp)
        zzz3 = self.mul(zzzp, ppp)
        y3 = self.sub("K0", t0, t1)
        return (x3, y3, zz3, zzz3)

    def neg(self, P):
        X, Y, ZZ, ZZZ = P
        return (X, self.sub("K2", 0.0, Y), ZZ, ZZZ)


def closure(rp, ext, store, fixed=None, kneg_fixed=None, beta=1):
    law = Law(rp, ext, store, fixed, kneg_fixed, beta)
    S = (store,) * 4
    acc = S
    for _ in range(60):          # the K's only grow; iterate until nothing moves
        before = (dict(law.need), law.kneg, acc)
        outs = []
        for P in (S, acc):
            outs.append(law.dbl(P))
            for O in (S, acc, law.neg(S), law.neg(acc)):
                outs.append(law.add(P, O))
        acc = tuple(max(o[i] for o in outs + [acc]) for i in range(4))
        if before == (dict(law.need), law.kneg, acc):
            break
    return law, acc


def main():
    for name, (p, w, n) in FIELDS.items():
        rp = (1 << (w * n)) / p
        for ext in (False, True):
            # storage bounds of ec29.h: 7 p, or ~2.03 p for the extension of a field with little slack
            store = 2.04 if (ext and w * n - p.bit_length() < 9) else 7.0
            law, acc = closure(rp, ext, store, beta=BETA[name])
            ks = {k: math.floor(v) + 1 for k, v in sorted(law.need.items())}
            print("%-13s %s  R/p = %8.1f  running point < (%.2f, %.2f, %.2f, %.2f) p  K = %s  negation K = %d  "
                  "zero test argument < %.2f p  largest value %.1f p" %
                  (name, "Fq2" if ext else "Fq ", rp, *acc, ks, math.floor(law.kneg) + 1, law.zero_arg, law.maxv))
            assert law.maxv < rp, "a value would not fit below R"
            # ... and with the constants lane29.h uses for every field (kSubKs, kNegK, kZeroMultiples)
            ks, kneg, nz = LANE29[ext]
            law, acc = closure(rp, ext, store, ks, kneg, BETA[name])
            assert law.zero_arg < nz and law.maxv < rp
            print("   lane29.h: K = %s, negation K = %s: running point < (%.2f, %.2f, %.2f, %.2f) p, zero test argument "
                  "< %.2f p (%d multiples), largest value %.1f p" % (ks, kneg, *acc, law.zero_arg, nz, law.maxv))


if __name__ == "__main__":
    main()
