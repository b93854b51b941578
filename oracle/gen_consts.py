#!/usr/bin/env python3
"""Generates the field/curve constant headers from the big-int parameters in
oracle/pyref/fields.py + curves.py (which tests/test_oracle_kats.py pins against the
reference's in-tree known answers).

    python oracle/gen_consts.py            # rewrites both headers

Outputs (both committed):
    oracle/c/consts_gen.h                          64-bit limbs, for the CPU oracle (plain C)
    distributed-groth16_amd/csrc/consts_gen.h      32-bit limbs, for the gfx950 kernels

The two headers are generated independently of each other's consumers; the oracle never
includes the product header and vice versa.
"""

import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle.pyref.fields import FQ, FR  # noqa: E402
from oracle.pyref.curves import CURVES  # noqa: E402


def limbs(x, n, bits):
    mask = (1 << bits) - 1
    return [(x >> (bits * i)) & mask for i in range(n)]


def arr(x, n, bits):
    suffix = "ULL" if bits == 64 else "u"
    w = 16 if bits == 64 else 8
    return "{" + ", ".join("0x%0*x%s" % (w, v, suffix) for v in limbs(x, n, bits)) + "}"


def field_entries(F):
    p = F.p
    n64 = F.limbs64
    Rbits = 64 * n64
    R = (1 << Rbits) % p
    out = dict(P=p, R=R, R2=R * R % p,
               INV64=(-pow(p, -1, 1 << 64)) % (1 << 64),
               INV32=(-pow(p, -1, 1 << 32)) % (1 << 32))
    if F.generator:
        out["GEN"] = F.generator * R % p
        out["TWO_ADICITY"] = F.two_adicity
        out["TWO_ADIC_ROOT"] = F.two_adic_root * R % p
    return out


def glv_entries(cname):
    """The G1 endomorphism phi(x, y) = (BETA x, y) = LAMBDA (x, y) of a j = 0 curve and a reduced basis of the lattice
    {(a, b): a + b LAMBDA = 0 mod r}, oriented so that det = a1 b2 - a2 b1 = +r, b1 < 0 < b2 (then the rounded
    coordinates c1 = round(b2 k / r), c2 = round(-b1 k / r) of a scalar k are non-negative), and the multipliers
    G1 = round(2^256 b2 / r), G2 = round(2^256 |b1| / r) of the division-free rounding the device uses
    (csrc/glv.h).  (LAMBDA, BETA) is the pair with the shorter basis; phi(G) == LAMBDA G is checked here on the
    generator and again, independently, in tests/test_constants_independent.py."""
    import math
    C = CURVES[cname, "g1"]
    r, q = FR[cname].p, FQ[cname].p

    def cube_root_of_unity(p):
        for g in range(2, 100):
            w = pow(g, (p - 1) // 3, p)
            if w != 1:
                return w
        raise ValueError("no cube root of unity")

    lam0, beta0 = cube_root_of_unity(r), cube_root_of_unity(q)
    best = None
    for lam in (lam0, lam0 * lam0 % r):
        for beta in (beta0, beta0 * beta0 % q):
            if C.mul(C.gen, lam) != (beta * C.gen[0] % q, C.gen[1]):
                continue
            # extended Euclid on (r, lam): remainders r_i = s_i r + t_i lam, so (r_i, -t_i) is in the lattice
            seq, r0, r1, t0, t1 = [], r, lam, 0, 1
            while r1:
                qq = r0 // r1
                r0, r1, t0, t1 = r1, r0 - qq * r1, t1, t0 - qq * t1
                seq.append((r0, -t0))
            i = next(j for j, (ri, _) in enumerate(seq) if ri < math.isqrt(r))
            v1 = seq[i]
            v2 = min([seq[i - 1]] + ([seq[i + 1]] if i + 1 < len(seq) else []), key=lambda v: v[0] ** 2 + v[1] ** 2)
            norm = max(abs(x) for x in v1 + v2)
            if best is None or norm < best[0]:
                best = (norm, lam, beta, v1, v2)
    assert best, "no (lambda, beta) pair with phi(G) = lambda G"
    _, lam, beta, v1, v2 = best
    # orientation: b1 < 0 < b2 and det = +r
    if v1[1] > 0:
        v1 = (-v1[0], -v1[1])
    if v2[1] < 0:
        v2 = (-v2[0], -v2[1])
    if v1[0] * v2[1] - v2[0] * v1[1] < 0:
        v1, v2 = (-v2[0], -v2[1]), (-v1[0], -v1[1])
    (a1, b1), (a2, b2) = v1, v2
    assert b1 < 0 < b2 and a1 * b2 - a2 * b1 == r and (a1 + b1 * lam) % r == 0 and (a2 + b2 * lam) % r == 0
    g1 = ((b2 << 256) + r // 2) // r
    g2 = ((-b1 << 256) + r // 2) // r
    return dict(LAMBDA=lam, BETA=beta, A1=a1, B1=b1, A2=a2, B2=b2, G1=g1, G2=g2)


def glv_lattice(r, lam):
    """reduced basis of {(a, b): a + b lam = 0 mod r} with det = +r, b1 < 0 < b2, and the rounding multipliers"""
    import math
    seq, r0, r1, t0, t1 = [], r, lam, 0, 1
    while r1:
        qq = r0 // r1
        r0, r1, t0, t1 = r1, r0 - qq * r1, t1, t0 - qq * t1
        seq.append((r0, -t0))
    i = next(j for j, (ri, _) in enumerate(seq) if ri < math.isqrt(r))
    v1 = seq[i]
    v2 = min([seq[i - 1]] + ([seq[i + 1]] if i + 1 < len(seq) else []), key=lambda v: v[0] ** 2 + v[1] ** 2)
    if v1[1] > 0:
        v1 = (-v1[0], -v1[1])
    if v2[1] < 0:
        v2 = (-v2[0], -v2[1])
    if v1[0] * v2[1] - v2[0] * v1[1] < 0:
        v1, v2 = (-v2[0], -v2[1]), (-v1[0], -v1[1])
    (a1, b1), (a2, b2) = v1, v2
    assert b1 < 0 < b2 and a1 * b2 - a2 * b1 == r and (a1 + b1 * lam) % r == 0 and (a2 + b2 * lam) % r == 0
    return dict(LAMBDA=lam, A1=a1, B1=b1, A2=a2, B2=b2, G1=((b2 << 256) + r // 2) // r, G2=((-b1 << 256) + r // 2) // r)


def glv_g2_entries(cname):
    """BN254's G2: psi(x, y) = (GAMMA_X conj(x), GAMMA_Y conj(y)) -- untwist, Frobenius, twist: GAMMA_X = xi^((q-1)/3),
    GAMMA_Y = xi^((q-1)/2) or their inverses -- acts on the order-r subgroup as multiplication by LAMBDA = +-q mod r
    (~2^127 for a BN curve, so the two-dimensional lattice is balanced; a BLS12 curve has q = u mod r, 64 bits, and
    needs the four-dimensional form: not built).  The combination is found on the generator."""
    assert cname == "bn254"
    C = CURVES[cname, "g2"]
    F2, r = C.F, FR[cname].p
    q = F2.p

    def f2pow(a, e):
        res = (1, 0)
        while e:
            if e & 1:
                res = F2.mul(res, a)
            a = F2.mul(a, a)
            e >>= 1
        return res

    conj = lambda a: (a[0], (-a[1]) % q)   # noqa: E731
    xi = (9, 1)
    gx, gy = f2pow(xi, (q - 1) // 3), f2pow(xi, (q - 1) // 2)
    G = C.gen
    hits = []
    for cx in (gx, F2.inv(gx)):
        for cy in (gy, F2.inv(gy)):
            P = (F2.mul(cx, conj(G[0])), F2.mul(cy, conj(G[1])))
            for lam in (q % r, (-q) % r):
                if C.mul(G, lam) == P:
                    hits.append((cx, cy, lam))
    assert len(hits) == 1, hits
    cx, cy, lam = hits[0]
    e = glv_lattice(r, lam)
    e.update(GAMMA_X=cx, GAMMA_Y=cy)
    return e


def _lll(B):
    """integer LLL (delta = 3/4) on the rows of B, exact arithmetic: four rows, once per curve"""
    from fractions import Fraction
    n = len(B)
    B = [list(r) for r in B]

    def gs():
        Bs, mu = [], [[Fraction(0)] * n for _ in range(n)]
        for i in range(n):
            v = [Fraction(x) for x in B[i]]
            for j in range(i):
                mu[i][j] = sum(Fraction(B[i][k]) * Bs[j][k] for k in range(n)) / sum(x * x for x in Bs[j])
                v = [v[k] - mu[i][j] * Bs[j][k] for k in range(n)]
            Bs.append(v)
        return Bs, mu

    k = 1
    while k < n:
        Bs, mu = gs()
        for j in range(k - 1, -1, -1):
            q = round(mu[k][j])
            if q:
                B[k] = [B[k][i] - q * B[j][i] for i in range(n)]
                Bs, mu = gs()
        if sum(x * x for x in Bs[k]) >= (Fraction(3, 4) - mu[k][k - 1] ** 2) * sum(x * x for x in Bs[k - 1]):
            k += 1
        else:
            B[k], B[k - 1] = B[k - 1], B[k]
            k = max(k - 1, 1)
    return B


def _det(M):
    from fractions import Fraction
    n = len(M)
    M = [[Fraction(x) for x in r] for r in M]
    d = Fraction(1)
    for i in range(n):
        p = next((r for r in range(i, n) if M[r][i] != 0), None)
        if p is None:
            return 0
        if p != i:
            M[i], M[p] = M[p], M[i]
            d = -d
        d *= M[i][i]
        for r in range(i + 1, n):
            f = M[r][i] / M[i][i]
            M[r] = [M[r][c] - f * M[i][c] for c in range(n)]
    assert d.denominator == 1
    return int(d)


XI = {"bn254": (9, 1), "bls12_381": (1, 1), "bls12_377": (0, 1)}       # the twist's non-residue xi in Fq2


def glv4_g2_entries(cname):
    """G2 of a BN / BLS12 curve: psi(x, y) = (GAMMA_X conj(x), GAMMA_Y conj(y)) (untwist, Frobenius, twist) acts on the
    order-r subgroup as multiplication by LAMBDA = +-q mod r, a root of x^4 - x^2 + 1, so every scalar has a FOUR-
    dimensional split k = k0 + k1 LAMBDA + k2 LAMBDA^2 + k3 LAMBDA^3 with |k_j| < 2^64: an LLL-reduced basis B (rows)
    of {x: sum x_j LAMBDA^j = 0 mod r}, Babai rounding c_i = round(k A_i / det B) with A = the first row of adj(B), and
    k_j = [j = 0] k - sum_i c_i B[i][j] (csrc/glv.h: split4; the identity holds for any integers c_i).  Emitted:
    magnitudes + signs of B, the multipliers G_i = round(2^256 |A_i| / r) and the sign of each c_i."""
    C = CURVES[cname, "g2"]
    F2, r = C.F, FR[cname].p
    q = F2.p

    def f2pow(a, e):
        res = (1, 0)
        while e:
            if e & 1:
                res = F2.mul(res, a)
            a = F2.mul(a, a)
            e >>= 1
        return res

    conj = lambda a: (a[0], (-a[1]) % q)   # noqa: E731
    gx, gy = f2pow(XI[cname], (q - 1) // 3), f2pow(XI[cname], (q - 1) // 2)
    G = C.gen
    hits = []
    for cx in (gx, F2.inv(gx)):
        for cy in (gy, F2.inv(gy)):
            P = (F2.mul(cx, conj(G[0])), F2.mul(cy, conj(G[1])))
            for lam in (q % r, (-q) % r):
                if C.mul(G, lam) == P:
                    hits.append((cx, cy, lam))
    assert hits, "no psi on " + cname
    cx, cy, lam = hits[0]
    assert (lam ** 4 - lam ** 2 + 1) % r == 0
    B = _lll([[r, 0, 0, 0], [-lam % r, 1, 0, 0], [-(lam * lam) % r, 0, 1, 0], [-(lam ** 3) % r, 0, 0, 1]])
    for row in B:
        assert sum(x * lam ** j for j, x in enumerate(row)) % r == 0
    d = _det(B)
    assert abs(d) == r
    A = []
    for i in range(4):                       # (e0 B^-1)_i = cofactor(i, 0) / det
        minor = [[B[rr][c] for c in range(1, 4)] for rr in range(4) if rr != i]
        A.append((-1) ** i * _det(minor))
    assert all(sum(A[i] * B[i][j] for i in range(4)) == (d if j == 0 else 0) for j in range(4))
    return dict(GAMMA_X=cx, GAMMA_Y=cy, LAMBDA=lam, B=B,
                G=[((abs(a) << 256) + r // 2) // r for a in A], C_NEG=[(a < 0) != (d < 0) for a in A])


def emit_c64():
    o = ["/* GENERATED by oracle/gen_consts.py -- do not edit. 64-bit limbs, little endian;",
         "   field elements are in Montgomery form with R = 2^(64*limbs). */",
         "#ifndef ORACLE_CONSTS_GEN_H", "#define ORACLE_CONSTS_GEN_H", "#include <stdint.h>", ""]
    for kind, table in (("fq", FQ), ("fr", FR)):
        for cname, F in table.items():
            pre = "%s_%s_" % (cname, kind)
            n = F.limbs64
            e = field_entries(F)
            o.append("#define %sNL %d" % (pre.upper(), n))
            for k in ("P", "R", "R2", "GEN", "TWO_ADIC_ROOT"):
                if k in e:
                    o.append("static const uint64_t %s%s[%d] = %s;" % (pre, k, n, arr(e[k], n, 64)))
            o.append("static const uint64_t %sINV = 0x%016xULL;" % (pre, e["INV64"]))
            if "TWO_ADICITY" in e:
                o.append("#define %sTWO_ADICITY %d" % (pre.upper(), e["TWO_ADICITY"]))
            o.append("")
    for (cname, g), C in CURVES.items():
        F = FQ[cname]
        n = F.limbs64
        R = F.R
        pre = "%s_%s_" % (cname, g)
        if g == "g1":
            o.append("static const uint64_t %sB[%d] = %s;" % (pre, n, arr(C.b * R % F.p, n, 64)))
            o.append("static const uint64_t %sGX[%d] = %s;" % (pre, n, arr(C.gen[0] * R % F.p, n, 64)))
            o.append("static const uint64_t %sGY[%d] = %s;" % (pre, n, arr(C.gen[1] * R % F.p, n, 64)))
        else:
            for nm, v in (("B", C.b), ("GX", C.gen[0]), ("GY", C.gen[1])):
                for ci in (0, 1):
                    o.append("static const uint64_t %s%s_C%d[%d] = %s;" %
                             (pre, nm, ci, n, arr(v[ci] * R % F.p, n, 64)))
        o.append("")
    o.append("#endif")
    return "\n".join(o) + "\n"


def emit_hip32():
    o = ["// GENERATED by oracle/gen_consts.py -- do not edit. 32-bit limbs, little endian;",
         "// Montgomery form with R = 2^(32*NL) (== 2^(64*limbs64), the arkworks in-memory R).",
         "#pragma once", "#include <stdint.h>", "", "namespace dg16 {", ""]
    for kind, table in (("fq", FQ), ("fr", FR)):
        for cname, F in table.items():
            n = 2 * F.limbs64
            e = field_entries(F)
            sname = "%s_%s" % (cname, kind)
            o.append("struct %s_params {" % sname)
            o.append("  static constexpr int NL = %d;" % n)
            o.append("  static constexpr uint32_t INV = 0x%08xu;" % e["INV32"])
            for k in ("P", "R", "R2", "GEN", "TWO_ADIC_ROOT"):
                if k in e:
                    o.append("  static constexpr uint32_t %s[%d] = %s;" % (k, n, arr(e[k], n, 32)))
            if "TWO_ADICITY" in e:
                o.append("  static constexpr int TWO_ADICITY = %d;" % e["TWO_ADICITY"])
            o.append("};")
            o.append("")
    for (cname, g), C in CURVES.items():
        F = FQ[cname]
        n = 2 * F.limbs64
        R = F.R
        o.append("struct %s_%s_consts {" % (cname, g))
        if g == "g1":
            o.append("  static constexpr uint32_t B[%d] = %s;" % (n, arr(C.b * R % F.p, n, 32)))
            o.append("  static constexpr uint32_t GX[%d] = %s;" % (n, arr(C.gen[0] * R % F.p, n, 32)))
            o.append("  static constexpr uint32_t GY[%d] = %s;" % (n, arr(C.gen[1] * R % F.p, n, 32)))
        else:
            for nm, v in (("B", C.b), ("GX", C.gen[0]), ("GY", C.gen[1])):
                for ci in (0, 1):
                    o.append("  static constexpr uint32_t %s_C%d[%d] = %s;" %
                             (nm, ci, n, arr(v[ci] * R % F.p, n, 32)))
        o.append("};")
        o.append("")
    # GLV decomposition of G1 scalars (csrc/glv.h): magnitudes as 32-bit words, signs as flags
    for cname in FQ:
        e = glv_entries(cname)
        F = FQ[cname]
        n = 2 * F.limbs64
        o.append("struct %s_glv_consts {" % cname)
        o.append("  static constexpr uint32_t BETA[%d] = %s;   // arkworks Montgomery form" % (n, arr(e["BETA"] * F.R % F.p, n, 32)))
        o.append("  static constexpr uint32_t LAMBDA[8] = %s;   // plain integer" % arr(e["LAMBDA"], 8, 32))
        for k in ("A1", "B1", "A2", "B2"):
            assert abs(e[k]) < 1 << 160
            o.append("  static constexpr uint32_t %s[5] = %s;   // |%s|" % (k, arr(abs(e[k]), 5, 32), k.lower()))
            o.append("  static constexpr bool %s_NEG = %s;" % (k, "true" if e[k] < 0 else "false"))
        for k in ("G1", "G2"):
            assert e[k] < 1 << 160
            o.append("  static constexpr uint32_t %s[5] = %s;" % (k, arr(e[k], 5, 32)))
        o.append("};")
        o.append("")
    e = glv_g2_entries("bn254")
    F = FQ["bn254"]
    n = 2 * F.limbs64
    o.append("struct bn254_g2_glv_consts {")
    for nm in ("GAMMA_X", "GAMMA_Y"):
        for ci in (0, 1):
            o.append("  static constexpr uint32_t %s_C%d[%d] = %s;   // arkworks Montgomery form" %
                     (nm, ci, n, arr(e[nm][ci] * F.R % F.p, n, 32)))
    o.append("  static constexpr uint32_t LAMBDA[8] = %s;   // plain integer" % arr(e["LAMBDA"], 8, 32))
    for k in ("A1", "B1", "A2", "B2"):
        assert abs(e[k]) < 1 << 160
        o.append("  static constexpr uint32_t %s[5] = %s;   // |%s|" % (k, arr(abs(e[k]), 5, 32), k.lower()))
        o.append("  static constexpr bool %s_NEG = %s;" % (k, "true" if e[k] < 0 else "false"))
    for k in ("G1", "G2"):
        assert e[k] < 1 << 160
        o.append("  static constexpr uint32_t %s[5] = %s;" % (k, arr(e[k], 5, 32)))
    o.append("};")
    o.append("")
    for cname in ("bls12_381", "bls12_377"):      # (BN254's G2 keeps the two-dimensional form: measured faster there)
        e = glv4_g2_entries(cname)
        F = FQ[cname]
        n = 2 * F.limbs64
        o.append("struct %s_g2_glv4_consts {" % cname)
        for nm in ("GAMMA_X", "GAMMA_Y"):
            for ci in (0, 1):
                o.append("  static constexpr uint32_t %s_C%d[%d] = %s;   // arkworks Montgomery form" %
                         (nm, ci, n, arr(e[nm][ci] * F.R % F.p, n, 32)))
        o.append("  static constexpr uint32_t LAMBDA[8] = %s;   // plain integer" % arr(e["LAMBDA"], 8, 32))
        flat = [x for row in e["B"] for x in row]
        assert all(abs(x) < 1 << 96 for x in flat) and all(g < 1 << 224 for g in e["G"])
        o.append("  static constexpr uint32_t B[16][3] = {%s};   // |B[i][j]| at 4 i + j" % ", ".join(arr(abs(x), 3, 32) for x in flat))
        o.append("  static constexpr bool B_NEG[16] = {%s};" % ", ".join("true" if x < 0 else "false" for x in flat))
        o.append("  static constexpr uint32_t G[4][7] = {%s};" % ", ".join(arr(g, 7, 32) for g in e["G"]))
        o.append("  static constexpr bool C_NEG[4] = {%s};" % ", ".join("true" if x else "false" for x in e["C_NEG"]))
        o.append("};")
        o.append("")
    o.append("}  // namespace dg16")
    return "\n".join(o) + "\n"


def main():
    root = os.path.dirname(HERE)
    p1 = os.path.join(HERE, "c", "consts_gen.h")
    p2 = os.path.join(root, "distributed-groth16_amd", "csrc", "consts_gen.h")
    os.makedirs(os.path.dirname(p1), exist_ok=True)
    os.makedirs(os.path.dirname(p2), exist_ok=True)
    with open(p1, "w") as f:
        f.write(emit_c64())
    with open(p2, "w") as f:
        f.write(emit_hip32())
    print("wrote", p1, "and", p2)


if __name__ == "__main__":
    main()
