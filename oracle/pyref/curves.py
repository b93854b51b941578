"""TEST INFRASTRUCTURE ONLY -- short-Weierstrass groups G1 (over Fq) and G2 (over Fq2) for
BN254 / BLS12-381 / BLS12-377, plain Python ints.  Restates what the reference gets from
ark-ec 0.4 (`CurveGroup`, `AffineRepr`, `VariableBaseMSM::msm` at
/root/reference/dist-primitives/src/dmsm/mod.rs:82).  An MSM result is a unique group element,
so any correct algorithm is an exact oracle once results are compared in affine form.

Affine points are (x, y) tuples, the identity is None.  Fq2 elements are (c0, c1) = c0 + c1*u.
"""

from .fields import FQ, FR


class FqOps:
    def __init__(self, p):
        self.p = p
        self.zero = 0
        self.one = 1

    def add(self, a, b): return (a + b) % self.p
    def sub(self, a, b): return (a - b) % self.p
    def mul(self, a, b): return a * b % self.p
    def sqr(self, a): return a * a % self.p
    def neg(self, a): return (-a) % self.p
    def inv(self, a): return pow(a, self.p - 2, self.p)
    def is_zero(self, a): return a == 0
    def small(self, k): return k % self.p


class Fq2Ops:
    """Fq[u]/(u^2 - nr)."""

    def __init__(self, p, nr):
        self.p = p
        self.nr = nr % p
        self.zero = (0, 0)
        self.one = (1, 0)

    def add(self, a, b): return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)
    def sub(self, a, b): return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)

    def mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] + self.nr * a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def sqr(self, a): return self.mul(a, a)
    def neg(self, a): return ((-a[0]) % self.p, (-a[1]) % self.p)

    def inv(self, a):
        p = self.p
        n = (a[0] * a[0] - self.nr * a[1] * a[1]) % p
        ni = pow(n, p - 2, p)
        return (a[0] * ni % p, (-a[1]) * ni % p)

    def is_zero(self, a): return a[0] == 0 and a[1] == 0
    def small(self, k): return (k % self.p, 0)


class Curve:
    """y^2 = x^3 + b over the field described by `ops` (a = 0 for all three families)."""

    def __init__(self, name, ops, b, gen, order):
        self.name, self.F, self.b, self.gen, self.order = name, ops, b, gen, order

    # ---- affine helpers ---------------------------------------------------------------
    def on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.sqr(y) == F.add(F.mul(F.sqr(x), x), self.b)

    def neg(self, P):
        return None if P is None else (P[0], self.F.neg(P[1]))

    # ---- Jacobian arithmetic (X, Y, Z), identity has Z = 0 ------------------------------
    def to_jac(self, P):
        F = self.F
        return (F.one, F.one, F.zero) if P is None else (P[0], P[1], F.one)

    def to_affine(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def jdbl(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z):
            return J
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        D = F.sub(F.sub(F.sqr(F.add(X, B)), A), C)
        D = F.add(D, D)
        E = F.add(F.add(A, A), A)
        Fq = F.sqr(E)
        X3 = F.sub(Fq, F.add(D, D))
        C8 = F.add(C, C); C8 = F.add(C8, C8); C8 = F.add(C8, C8)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
        Z3 = F.mul(F.add(Y, Y), Z)
        return (X3, Y3, Z3)

    def jadd(self, P, Q):
        F = self.F
        if F.is_zero(P[2]):
            return Q
        if F.is_zero(Q[2]):
            return P
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Q
        Z1Z1 = F.sqr(Z1)
        Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(F.mul(Y1, Z2), Z2Z2)
        S2 = F.mul(F.mul(Y2, Z1), Z1Z1)
        if U1 == U2:
            if S1 == S2:
                return self.jdbl(P)
            return (F.one, F.one, F.zero)
        H = F.sub(U2, U1)
        Rr = F.sub(S2, S1)
        HH = F.sqr(H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.sqr(Rr), HHH), F.add(V, V))
        Y3 = F.sub(F.mul(Rr, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    # ---- affine API ---------------------------------------------------------------------
    def add(self, P, Q):
        return self.to_affine(self.jadd(self.to_jac(P), self.to_jac(Q)))

    def mul(self, P, k):
        """k*P for a non-negative integer k (NOT reduced mod the group order)."""
        return self.to_affine(self.jmul(self.to_jac(P), k))

    def jmul(self, J, k):
        F = self.F
        assert k >= 0
        acc = (F.one, F.one, F.zero)
        for bit in bin(k)[2:]:
            acc = self.jdbl(acc)
            if bit == "1":
                acc = self.jadd(acc, J)
        return acc

    def msm(self, bases, scalars):
        """Definition of VariableBaseMSM::msm: sum_i scalars[i] * bases[i]
        (dmsm/mod.rs:82).  arkworks returns Err(min_len) on a length mismatch."""
        if len(bases) != len(scalars):
            raise ValueError(min(len(bases), len(scalars)))
        acc = self.to_jac(None)
        for P, s in zip(bases, scalars):
            acc = self.jadd(acc, self.jmul(self.to_jac(P), s))
        return self.to_affine(acc)

    def msm_pippenger(self, bases, scalars, c=4):
        """Independent algorithm (unsigned c-bit buckets) used to cross-check `msm`."""
        if len(bases) != len(scalars):
            raise ValueError(min(len(bases), len(scalars)))
        nbits = max([s.bit_length() for s in scalars] + [1])
        total = self.to_jac(None)
        for w in reversed(range((nbits + c - 1) // c)):
            for _ in range(c):
                total = self.jdbl(total)
            buckets = [self.to_jac(None)] * (1 << c)
            for P, s in zip(bases, scalars):
                d = (s >> (w * c)) & ((1 << c) - 1)
                if d:
                    buckets[d] = self.jadd(buckets[d], self.to_jac(P))
            run = self.to_jac(None)
            acc = self.to_jac(None)
            for d in range((1 << c) - 1, 0, -1):
                run = self.jadd(run, buckets[d])
                acc = self.jadd(acc, run)
            total = self.jadd(total, acc)
        return self.to_affine(total)


def _mk():
    out = {}
    # ---------------- BN254 ----------------
    q = FQ["bn254"].p
    out["bn254", "g1"] = Curve("bn254_g1", FqOps(q), 3, (1, 2), FR["bn254"].p)
    f2 = Fq2Ops(q, -1)
    b2 = f2.mul((3, 0), f2.inv((9, 1)))
    g2 = (
        (10857046999023057135944570762232829481370756359578518086990519993285655852781,
         11559732032986387107991004021392285783925812861821192530917403151452391805634),
        (8495653923123431417604973247489272438418190587263600148770280649306958101930,
         4082367875863433681332203403145435568316851327593401208105741076214120093531),
    )
    out["bn254", "g2"] = Curve("bn254_g2", f2, b2, g2, FR["bn254"].p)
    # ---------------- BLS12-381 ----------------
    q = FQ["bls12_381"].p
    g1 = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
          0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)
    out["bls12_381", "g1"] = Curve("bls12_381_g1", FqOps(q), 4, g1, FR["bls12_381"].p)
    f2 = Fq2Ops(q, -1)
    g2 = (
        (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
         0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
        (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
         0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
    )
    out["bls12_381", "g2"] = Curve("bls12_381_g2", f2, (4, 4), g2, FR["bls12_381"].p)
    # ---------------- BLS12-377 (the reference's curve for its dmsm / dfft tests and local_groth_bench) -------
    q = FQ["bls12_377"].p
    g1 = (0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF,
          0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6)
    out["bls12_377", "g1"] = Curve("bls12_377_g1", FqOps(q), 1, g1, FR["bls12_377"].p)
    # G2 over Fq2 = Fq[u]/(u^2 + 5) (ark-bls12-377 Fq2Config::NONRESIDUE = -5), D-type twist y^2 = x^3 + 1/u
    # (groth16/examples/local_groth_bench.rs:141 runs E::G2::msm on this curve).  Generator = ark-bls12-377's
    # G2_GENERATOR_{X,Y}; on-curve / order r checked in tests/test_oracle_kats.py.
    f2 = Fq2Ops(q, -5)
    b2 = f2.inv((0, 1))
    g2 = (
        (233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294,
         140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118),
        (63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423,
         149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491),
    )
    out["bls12_377", "g2"] = Curve("bls12_377_g2", f2, b2, g2, FR["bls12_377"].p)
    return out


CURVES = _mk()
