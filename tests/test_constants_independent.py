"""Every constant that oracle/gen_consts.py emits -- into BOTH consts_gen.h headers, the product's and the oracle's:
a wrong value would agree on both sides of every parity test -- recomputed here by other means, from nothing but the
curve-family parameter u (the BN / BLS12 polynomials), small integers, and long-hand arithmetic:

  P               q(u) / r(u) of the family                                  (not FQ / FR of oracle/pyref/fields.py)
  INV             Hensel lifting of -p^-1 mod 2^k, one bit at a time           (gen_consts: pow(p, -1, 2^k))
  R, R2           repeated doubling mod p                                      (gen_consts: (1 << bits) % p, R * R % p)
  GEN             smallest-looking multiplicative generator named by arkworks, checked to generate F_p^* against the
                  factorisation of p - 1 where it is known, and to be a non-residue everywhere
  TWO_ADIC_ROOT   square-and-multiply written out, order exactly 2^s
  B, GX, GY       decoded with R^-1 from the extended Euclidean algorithm: on the curve y^2 = x^3 + b with b from the
                  twist construction, and of order exactly r by a double-and-add ladder written here (affine, long-hand)

plus the constants fp29.h derives at COMPILE time (RR<P>: limb split of p, ONE, R2, FROM32, TO32, INV, R_WORDS,
R32SQ_OVER_R_WORDS), dumped through the host build of the header (tests/host_arith) and recomputed here."""

import ctypes
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIP_H = os.path.join(ROOT, "distributed-groth16_amd", "csrc", "consts_gen.h")
C_H = os.path.join(ROOT, "oracle", "c", "consts_gen.h")

U = {"bn254": 4965661367192848881, "bls12_381": -0xD201000000010000, "bls12_377": 0x8508C00000000001}
GENERATOR = {"bn254": 5, "bls12_381": 7, "bls12_377": 22}         # ark-bn254 / ark-bls12-381 / ark-bls12-377 Fr::GENERATOR
G1_B = {"bn254": 3, "bls12_381": 4, "bls12_377": 1}
NL64 = {"bn254": 4, "bls12_381": 6, "bls12_377": 6}


def family(curve):
    u = U[curve]
    if curve == "bn254":
        return (36 * u**4 + 36 * u**3 + 24 * u**2 + 6 * u + 1, 36 * u**4 + 36 * u**3 + 18 * u**2 + 6 * u + 1)
    r = u**4 - u**2 + 1
    return ((u - 1) ** 2 * r // 3 + u, r)


def doubling_pow2(bits, p):
    x = 1
    for _ in range(bits):
        x += x
        if x >= p:
            x -= p
    return x


def hensel_neg_inv(p, k):
    """-p^-1 mod 2^k, bit by bit: x p == -1 (mod 2^j) is extended to j + 1."""
    x = 1                                   # p odd: p * 1 == 1 == -1 (mod 2)
    for j in range(1, k):
        if (x * p + 1) >> j & 1:
            x |= 1 << j
    assert (x * p + 1) % (1 << k) == 0
    return x


def egcd_inv(a, p):
    r0, r1, s0, s1 = p, a % p, 0, 1
    while r1:
        q = r0 // r1
        r0, r1, s0, s1 = r1, r0 - q * r1, s1, s0 - q * s1
    assert r0 == 1
    return s0 % p


def sqmul(b, e, p):
    acc = 1
    for bit in bin(e)[2:]:
        acc = acc * acc % p
        if bit == "1":
            acc = acc * b % p
    return acc


# ---- header parsing -------------------------------------------------------------------------------------------------
def parse_hip(path):
    """{struct: {name: int}} from the 32-bit-limb header."""
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"struct (\w+) \{", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.match(r"\s*static constexpr uint32_t (\w+)\[(\d+)\] = \{(.*)\};", line)
        if m and cur is not None:
            limbs = [int(x.strip().rstrip("u"), 16) for x in m.group(3).split(",")]
            assert len(limbs) == int(m.group(2))
            cur[m.group(1)] = sum(v << (32 * i) for i, v in enumerate(limbs))
            continue
        m = re.match(r"\s*static constexpr (?:uint32_t|int) (\w+) = (\w+);", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2).rstrip("u"), 0)
    return out


def parse_c(path):
    out = {}
    for line in open(path):
        m = re.match(r"static const uint64_t (\w+)\[(\d+)\] = \{(.*)\};", line)
        if m:
            limbs = [int(x.strip().replace("ULL", ""), 16) for x in m.group(3).split(",")]
            out[m.group(1)] = sum(v << (64 * i) for i, v in enumerate(limbs))
            continue
        m = re.match(r"static const uint64_t (\w+) = (\w+);", line)
        if m:
            out[m.group(1)] = int(m.group(2).replace("ULL", ""), 16)
            continue
        m = re.match(r"#define (\w+) (\d+)$", line)
        if m:
            out[m.group(1)] = int(m.group(2))
    return out


HIP = parse_hip(HIP_H)
CH = parse_c(C_H)


# ---- long-hand affine curve arithmetic over Fp / Fp[u]/(u^2 + beta) ----------------------------------------------------
class Fp2:
    def __init__(self, p, beta):
        self.p, self.beta = p, beta

    def mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] - self.beta * a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def add(self, a, b): return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)
    def sub(self, a, b): return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)

    def inv(self, a):
        n = egcd_inv((a[0] * a[0] + self.beta * a[1] * a[1]) % self.p, self.p)
        return (a[0] * n % self.p, -a[1] * n % self.p)

    def const(self, k): return (k % self.p, 0)
    zero = (0, 0)


class Fp1:
    def __init__(self, p): self.p = p
    def mul(self, a, b): return a * b % self.p
    def add(self, a, b): return (a + b) % self.p
    def sub(self, a, b): return (a - b) % self.p
    def inv(self, a): return egcd_inv(a, self.p)
    def const(self, k): return k % self.p
    zero = 0


def affine_add(F, P, Q):
    if P is None: return Q
    if Q is None: return P
    if P[0] == Q[0]:
        if F.add(P[1], Q[1]) == F.zero:
            return None
        lam = F.mul(F.mul(F.const(3), F.mul(P[0], P[0])), F.inv(F.add(P[1], P[1])))
    else:
        lam = F.mul(F.sub(Q[1], P[1]), F.inv(F.sub(Q[0], P[0])))
    x = F.sub(F.sub(F.mul(lam, lam), P[0]), Q[0])
    return (x, F.sub(F.mul(lam, F.sub(P[0], x)), P[1]))


def ladder(F, P, k):
    acc = None
    for bit in bin(k)[2:]:
        acc = affine_add(F, acc, acc)
        if bit == "1":
            acc = affine_add(F, acc, P)
    return acc


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
@pytest.mark.parametrize("kind", ["fq", "fr"])
def test_field_constants_recomputed(curve, kind):
    q, r = family(curve)
    p = q if kind == "fq" else r
    nl = NL64[curve] if kind == "fq" else 4
    hs = HIP["%s_%s_params" % (curve, kind)]
    pre = "%s_%s_" % (curve, kind)
    R = doubling_pow2(64 * nl, p)
    R2 = doubling_pow2(128 * nl, p)
    assert hs["NL"] == 2 * nl and CH[pre.upper() + "NL"] == nl
    assert hs["P"] == p == CH[pre + "P"]
    assert hs["R"] == R == CH[pre + "R"]
    assert hs["R2"] == R2 == CH[pre + "R2"]
    assert hs["INV"] == hensel_neg_inv(p, 32) and CH[pre + "INV"] == hensel_neg_inv(p, 64)
    if kind == "fr":
        g = GENERATOR[curve]
        s, t = 0, p - 1
        while t % 2 == 0:
            s, t = s + 1, t // 2
        assert hs["TWO_ADICITY"] == s == CH[pre.upper() + "TWO_ADICITY"]
        assert hs["GEN"] == g * R % p == CH[pre + "GEN"]
        # a generator of the 2-Sylow subgroup: non-residue
        assert sqmul(g, (p - 1) // 2, p) == p - 1
        # ... and of every small odd-order part that trial division finds (multiplicative generator as far as checkable)
        for f in (3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47):
            if (p - 1) % f == 0:
                assert sqmul(g, (p - 1) // f, p) != 1, f
        root = sqmul(g, t, p)
        assert sqmul(root, 1 << (s - 1), p) == p - 1          # order exactly 2^s
        assert hs["TWO_ADIC_ROOT"] == root * R % p == CH[pre + "TWO_ADIC_ROOT"]


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
@pytest.mark.parametrize("group", [1, 2])
def test_curve_constants_recomputed(curve, group):
    q, r = family(curve)
    nl = NL64[curve]
    Rinv = egcd_inv(doubling_pow2(64 * nl, q), q)
    hs = HIP["%s_g%d_consts" % (curve, group)]
    pre = "%s_g%d_" % (curve, group)
    dec = lambda v: v * Rinv % q     # noqa: E731
    if group == 1:
        F = Fp1(q)
        b = G1_B[curve]
        for k in ("B", "GX", "GY"):
            assert hs[k] == CH[pre + k]
        assert dec(hs["B"]) == b
        G = (dec(hs["GX"]), dec(hs["GY"]))
    else:
        beta = 5 if curve == "bls12_377" else 1                  # Fq2 = Fq[u] / (u^2 + beta)
        assert sqmul((-beta) % q, (q - 1) // 2, q) == q - 1      # -beta is a non-residue: Fq2 is a field
        F = Fp2(q, beta)
        # twist: bn254 D-type b / (9 + u); bls12-381 M-type 4 (1 + u); bls12-377 D-type 1 / u
        if curve == "bn254":
            b = F.mul((3, 0), F.inv((9, 1)))
        elif curve == "bls12_381":
            b = (4, 4)
        else:
            b = F.inv((0, 1))
        for k in ("B", "GX", "GY"):
            for c in ("_C0", "_C1"):
                assert hs[k + c] == CH[pre + k + c]
        assert (dec(hs["B_C0"]), dec(hs["B_C1"])) == b
        G = ((dec(hs["GX_C0"]), dec(hs["GX_C1"])), (dec(hs["GY_C0"]), dec(hs["GY_C1"])))
    x, y = G
    assert F.mul(y, y) == F.add(F.mul(F.mul(x, x), x), b if group == 2 else F.const(b))
    assert ladder(F, G, r) is None and ladder(F, G, 1) == G
    # prime order: r is prime (Miller-Rabin with fixed bases), so G generates the order-r subgroup
    d, s = r - 1, 0
    while d % 2 == 0:
        d, s = d // 2, s + 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        xx = sqmul(a, d, r)
        if xx in (1, r - 1):
            continue
        for _ in range(s - 1):
            xx = xx * xx % r
            if xx == r - 1:
                break
        else:
            raise AssertionError("r is composite?")


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
def test_glv_constants_recomputed(curve):
    """<curve>_glv_consts (csrc/glv.h): BETA and LAMBDA are primitive cube roots of unity of Fq / Fr with
    (BETA x, y) = LAMBDA (x, y) on the generator (long-hand affine ladder, nothing of the oracle), (a1, b1), (a2, b2) lie
    in the lattice {a + b LAMBDA = 0 mod r}, have determinant +r and the orientation b1 < 0 < b2 glv::split assumes, are
    SHORT (every entry below 2^128: the halves' bound), and G1 / G2 are the nearest integers to 2^256 b2 / r, 2^256 |b1| / r."""
    q, r = family(curve)
    nl = NL64[curve]
    Rinv = egcd_inv(doubling_pow2(64 * nl, q), q)
    text = open(HIP_H).read()
    blk = text[text.index("struct %s_glv_consts {" % curve):]
    blk = blk[:blk.index("\n};")]
    val = {m.group(1): sum(int(x.strip().rstrip("u"), 16) << (32 * i) for i, x in enumerate(m.group(2).split(",")))
           for m in re.finditer(r"static constexpr uint32_t (\w+)\[\d+\] = \{([^}]*)\}", blk)}
    neg = {m.group(1): m.group(2) == "true" for m in re.finditer(r"static constexpr bool (\w+)_NEG = (\w+);", blk)}
    beta, lam = val["BETA"] * Rinv % q, val["LAMBDA"]
    assert beta != 1 and sqmul(beta, 3, q) == 1 and lam != 1 and sqmul(lam, 3, r) == 1 and 1 < lam < r
    g = HIP["%s_g1_consts" % curve]
    G = (g["GX"] * Rinv % q, g["GY"] * Rinv % q)
    assert ladder(Fp1(q), G, lam) == (beta * G[0] % q, G[1])
    a1, b1, a2, b2 = ((-val[k] if neg[k] else val[k]) for k in ("A1", "B1", "A2", "B2"))
    assert (a1 + b1 * lam) % r == 0 and (a2 + b2 * lam) % r == 0
    assert a1 * b2 - a2 * b1 == r and b1 < 0 < b2
    assert max(abs(v) for v in (a1, b1, a2, b2)) < 1 << 128
    for gk, b in (("G1", b2), ("G2", -b1)):
        assert abs(val[gk] * r - (b << 256)) * 2 <= r


def test_glv_constants_of_bn254_g2_recomputed():
    """bn254_g2_glv_consts: psi(x, y) = (GAMMA_X conj(x), GAMMA_Y conj(y)) is multiplication by LAMBDA on the generator of
    G2 (long-hand ladder over Fq2), LAMBDA = +-q mod r, GAMMA_X^3 and GAMMA_Y^2 are what untwist-Frobenius-twist makes them
    (xi^(q-1) up to inversion), and the lattice / multipliers satisfy what glv::split assumes."""
    q, r = family("bn254")
    Rinv = egcd_inv(doubling_pow2(256, q), q)
    text = open(HIP_H).read()
    blk = text[text.index("struct bn254_g2_glv_consts {"):]
    blk = blk[:blk.index("\n};")]
    val = {m.group(1): sum(int(x.strip().rstrip("u"), 16) << (32 * i) for i, x in enumerate(m.group(2).split(",")))
           for m in re.finditer(r"static constexpr uint32_t (\w+)\[\d+\] = \{([^}]*)\}", blk)}
    neg = {m.group(1): m.group(2) == "true" for m in re.finditer(r"static constexpr bool (\w+)_NEG = (\w+);", blk)}
    F = Fp2(q, 1)
    dec = lambda k: (val[k + "_C0"] * Rinv % q, val[k + "_C1"] * Rinv % q)     # noqa: E731
    gx, gy, lam = dec("GAMMA_X"), dec("GAMMA_Y"), val["LAMBDA"]
    assert lam in (q % r, (-q) % r)
    g = HIP["bn254_g2_consts"]
    one = lambda k: (g[k + "_C0"] * Rinv % q, g[k + "_C1"] * Rinv % q)         # noqa: E731
    G = (one("GX"), one("GY"))
    conj = lambda a: (a[0], (-a[1]) % q)                                        # noqa: E731
    assert ladder(F, G, lam) == (F.mul(gx, conj(G[0])), F.mul(gy, conj(G[1])))
    # xi^(q-1) = GAMMA_X^(+-3) = GAMMA_Y^(+-2), xi = 9 + u
    def f2pow(a, e):
        acc = (1, 0)
        for bit in bin(e)[2:]:
            acc = F.mul(acc, acc)
            if bit == "1":
                acc = F.mul(acc, a)
        return acc
    xq = f2pow((9, 1), q - 1)
    assert f2pow(gx, 3) in (xq, F.inv(xq)) and f2pow(gy, 2) in (xq, F.inv(xq))
    a1, b1, a2, b2 = ((-val[k] if neg[k] else val[k]) for k in ("A1", "B1", "A2", "B2"))
    assert (a1 + b1 * lam) % r == 0 and (a2 + b2 * lam) % r == 0 and a1 * b2 - a2 * b1 == r and b1 < 0 < b2
    assert max(abs(v) for v in (a1, b1, a2, b2)) < 1 << 128
    for gk, b in (("G1", b2), ("G2", -b1)):
        assert abs(val[gk] * r - (b << 256)) * 2 <= r


@pytest.mark.parametrize("curve", ["bls12_381", "bls12_377"])
def test_glv4_constants_of_g2_recomputed(curve):
    """<curve>_g2_glv4_consts: psi(x, y) = (GAMMA_X conj(x), GAMMA_Y conj(y)) is multiplication by LAMBDA on the generator of
    G2 (long-hand ladder over Fq2), LAMBDA = +-q mod r is a root of x^4 - x^2 + 1, every row of B lies in the lattice
    {x: sum x_j LAMBDA^j = 0 mod r}, |det B| = r (cofactor expansion), the entries are below 2^65, G_i is the nearest integer
    to 2^256 |A_i| / r for the first row A of adj(B), and C_NEG is the sign of A_i / det."""
    q, r = family(curve)
    Rinv = egcd_inv(doubling_pow2(64 * NL64[curve], q), q)
    text = open(HIP_H).read()
    blk = text[text.index("struct %s_g2_glv4_consts {" % curve):]
    blk = blk[:blk.index("\n};")]
    words = lambda body: sum(int(x.strip().rstrip("u"), 16) << (32 * i) for i, x in enumerate(body.split(",")))   # noqa: E731
    one = lambda name: words(re.search(r" %s\[\d+\] = \{([^}]*)\}" % name, blk).group(1))                          # noqa: E731
    beta = 5 if curve == "bls12_377" else 1
    F = Fp2(q, beta)
    gx = (one("GAMMA_X_C0") * Rinv % q, one("GAMMA_X_C1") * Rinv % q)
    gy = (one("GAMMA_Y_C0") * Rinv % q, one("GAMMA_Y_C1") * Rinv % q)
    lam = one("LAMBDA")
    assert lam in (q % r, (-q) % r) and (lam ** 4 - lam ** 2 + 1) % r == 0
    g = HIP["%s_g2_consts" % curve]
    G = tuple((g[k + "_C0"] * Rinv % q, g[k + "_C1"] * Rinv % q) for k in ("GX", "GY"))
    conj = lambda a: (a[0], (-a[1]) % q)                                                                            # noqa: E731
    assert ladder(F, G, lam) == (F.mul(gx, conj(G[0])), F.mul(gy, conj(G[1])))
    Bm = [words(m) for m in re.findall(r"\{([^{}]*)\}", re.search(r" B\[16\]\[3\] = \{(.*)\};", blk).group(1))]
    Bn = [x.strip() == "true" for x in re.search(r"B_NEG\[16\] = \{([^}]*)\}", blk).group(1).split(",")]
    Gm = [words(m) for m in re.findall(r"\{([^{}]*)\}", re.search(r" G\[4\]\[7\] = \{(.*)\};", blk).group(1))]
    Cn = [x.strip() == "true" for x in re.search(r"C_NEG\[4\] = \{([^}]*)\}", blk).group(1).split(",")]
    B = [[(-Bm[4 * i + j] if Bn[4 * i + j] else Bm[4 * i + j]) for j in range(4)] for i in range(4)]
    assert all(sum(x * lam ** j for j, x in enumerate(row)) % r == 0 for row in B)
    assert max(abs(x) for row in B for x in row) < 1 << 65

    def det3(m):
        return (m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
                + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]))
    A = [(-1) ** i * det3([[B[rr][c] for c in range(1, 4)] for rr in range(4) if rr != i]) for i in range(4)]
    d = sum(A[i] * B[i][0] for i in range(4))                         # cofactor expansion along the first column
    assert abs(d) == r
    for i in range(4):
        assert abs(Gm[i] * r - (abs(A[i]) << 256)) * 2 <= r
        assert A[i] == 0 or Cn[i] == ((A[i] < 0) != (d < 0))


# ---- the compile-time constants of fp29.h (RR<P>) -------------------------------------------------------------------------
SRC = os.path.join(HERE, "host_arith", "host_arith.cpp")
SO = os.path.join(HERE, "host_arith", "libhost_arith.so")


@pytest.fixture(scope="module")
def ha():
    hdrs = [os.path.join(ROOT, "distributed-groth16_amd", "csrc", f)
            for f in ("fp.h", "fp2.h", "ec.h", "consts_gen.h", "fp29.h", "ec29.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(SO) < os.path.getmtime(p) for p in [SRC] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC])
    L = ctypes.CDLL(SO)
    L.ha_rr_consts.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t]
    L.ha_rr_consts.restype = ctypes.c_int
    return L


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
@pytest.mark.parametrize("kind", ["fq", "fr"])
def test_reduced_radix_constants_recomputed(ha, curve, kind):
    q, r = family(curve)
    p = q if kind == "fq" else r
    nl32 = 2 * (NL64[curve] if kind == "fq" else 4)
    fid = {"bn254": 0, "bls12_381": 1, "bls12_377": 2}[curve] + (16 if kind == "fr" else 0)
    buf = (ctypes.c_uint32 * 256)()
    n = ha.ha_rr_consts(fid, buf, 256)
    assert n > 0
    v = list(buf[:n])
    W, N, slack, inv = v[0], v[1], v[2], v[3]
    bits = p.bit_length()
    assert W == (29 if bits <= 256 else 28)
    assert N == (bits + 5 + W - 1) // W and slack == W * N - bits and slack >= 6
    assert inv == hensel_neg_inv(p, W)
    off = 4

    def take(count, width):
        nonlocal off
        val = sum(x << (width * i) for i, x in enumerate(v[off:off + count]))
        assert all(x < (1 << width) for x in v[off:off + count - 1])
        off += count
        return val

    Rb = W * N
    assert take(N, W) == p                                       # PL
    assert take(N, W) == doubling_pow2(Rb, p)                    # ONE
    assert take(N, W) == doubling_pow2(2 * Rb, p)                # R2
    assert take(N, W) == doubling_pow2(2 * Rb - 32 * nl32, p)    # FROM32: x R32 -> x R
    assert take(N, W) == doubling_pow2(32 * nl32, p)             # TO32
    assert take(nl32 + 1, 32) == doubling_pow2(Rb, p)            # R_WORDS
    assert take(nl32 + 1, 32) == doubling_pow2(64 * nl32 - Rb, p)   # R32SQ_OVER_R_WORDS
    assert v[off] == p >> (W * (N - 1))                          # PTOP
