"""TEST INFRASTRUCTURE ONLY -- reduced Tate pairing on BN254 / BLS12-381 with plain Python ints, and the
Groth16 verification equation built on it.  Nothing in the product path may import this.

Why it is here: the reference holds no golden MSM / NTT outputs, but it does hold complete
snarkjs-generated (proof, public inputs, verification key) triples -- ark-circom/test-vectors/{proof,public,
verification_key}.json and fixtures/million/{...}.json, copied as data under tests/golden/.  Accepting those
real proofs (and rejecting perturbed ones) pins this oracle's Fq / Fq2 / G1 / G2 arithmetic and its group
generators against vectors produced by an independent implementation; the same verifier then checks the
proofs our prover produces (e(A, B) = e(alpha, beta) e(IC, gamma) e(C, delta), the equation
`Groth16::verify_proof` evaluates at /root/reference/groth16/examples/sha256.rs:228-254).

Any non-degenerate bilinear map works for that equation, so this is the textbook reduced Tate pairing
t(P, Q) = f_{r,P}(psi(Q))^((q^12 - 1)/r): Miller loop over the bits of r with P in E(Fq) (vertical lines lie
in a proper subfield and die in the final exponentiation), Q untwisted into E(Fq12), and a plain
square-and-multiply final exponentiation -- no Frobenius constants, no curve-specific shortcuts to get wrong.
Fq12 = Fq2[w] / (w^6 - xi); elements are lists of six Fq2 pairs.
"""

from .curves import CURVES, Fq2Ops
from .fields import FQ, FR

# (xi, twist type): BN254 has a D-twist  (E': y^2 = x^3 + b/xi, psi(x, y) = (x w^2, y w^3)),
#                   BLS12-381 an M-twist (E': y^2 = x^3 + b*xi, psi(x, y) = (x / w^2, y / w^3))
_TOWER = {"bn254": ((9, 1), "D"), "bls12_381": ((1, 1), "M")}


class Fq12:
    def __init__(self, curve):
        self.q = FQ[curve].p
        self.F2 = Fq2Ops(self.q, -1)
        self.xi, self.twist = _TOWER[curve]
        self.one = [self.F2.one] + [self.F2.zero] * 5
        self.zero = [self.F2.zero] * 6

    def mul(self, a, b):
        F2 = self.F2
        t = [F2.zero] * 11
        for i, ai in enumerate(a):
            if ai == (0, 0):
                continue
            for j, bj in enumerate(b):
                if bj == (0, 0):
                    continue
                t[i + j] = F2.add(t[i + j], F2.mul(ai, bj))
        return [F2.add(t[i], F2.mul(t[i + 6], self.xi)) if i < 5 else t[i] for i in range(6)]

    def sub(self, a, b):
        return [self.F2.sub(x, y) for x, y in zip(a, b)]

    def scale(self, a, k):
        """a * k, k in Fq."""
        return [(x[0] * k % self.q, x[1] * k % self.q) for x in a]

    def pow(self, a, e):
        acc = self.one
        for bit in bin(e)[2:]:
            acc = self.mul(acc, acc)
            if bit == "1":
                acc = self.mul(acc, a)
        return acc

    def const(self, c):
        return [(c % self.q, 0)] + [self.F2.zero] * 5

    def untwist(self, Q):
        """psi: E'(Fq2) -> E(Fq12)."""
        F2 = self.F2
        x, y = Q
        X, Y = list(self.zero), list(self.zero)
        if self.twist == "D":
            X[2], Y[3] = x, y
        else:   # 1/w^2 = w^4 / xi, 1/w^3 = w^3 / xi
            xi_inv = F2.inv(self.xi)
            X[4], Y[3] = F2.mul(x, xi_inv), F2.mul(y, xi_inv)
        return X, Y


def miller_tate(curve, P, Q):
    """f_{r,P}(psi(Q)) without the final exponentiation; P in G1 (affine or None), Q in G2."""
    K = Fq12(curve)
    if P is None or Q is None:
        return K.one
    q, r = K.q, FR[curve].p
    X, Y = K.untwist(Q)
    xp, yp = P
    xt, yt = xp, yp
    f = K.one

    def line(lam, x0, y0):
        # l(psi(Q)) = Y - y0 - lam (X - x0)
        return K.sub(K.sub(Y, K.const(y0)), K.scale(K.sub(X, K.const(x0)), lam))

    bits = bin(r)[3:]
    for i, bit in enumerate(bits):
        lam = 3 * xt * xt * pow(2 * yt, q - 2, q) % q
        f = K.mul(K.mul(f, f), line(lam, xt, yt))
        x3 = (lam * lam - 2 * xt) % q
        yt = (lam * (xt - x3) - yt) % q
        xt = x3
        if bit == "1":
            if xt == xp:
                # T = -P only at the very last addition (T = (r-1)P): the line is vertical, in a subfield
                assert i == len(bits) - 1 and (yt + yp) % q == 0
                continue
            lam = (yt - yp) * pow(xt - xp, q - 2, q) % q
            f = K.mul(f, line(lam, xt, yt))
            x3 = (lam * lam - xt - xp) % q
            yt = (lam * (xt - x3) - yt) % q
            xt = x3
    return f


def final_exp(curve, f):
    K = Fq12(curve)
    return K.pow(f, (K.q ** 12 - 1) // FR[curve].p)


def pairing(curve, P, Q):
    return final_exp(curve, miller_tate(curve, P, Q))


def pairing_product_is_one(curve, pairs):
    """prod e(P_i, Q_i) == 1 with one shared final exponentiation."""
    K = Fq12(curve)
    f = K.one
    for P, Q in pairs:
        f = K.mul(f, miller_tate(curve, P, Q))
    return final_exp(curve, f) == K.one


def groth16_verify(curve, vk, public_inputs, proof):
    """vk = dict(alpha_g1, beta_g2, gamma_g2, delta_g2, ic=[G1...]); proof = (A, B, C) affine.
    e(A, B) == e(alpha, beta) * e(sum_i x_i IC_i, gamma) * e(C, delta)   (x_0 = 1)."""
    g1, g2 = CURVES[curve, "g1"], CURVES[curve, "g2"]
    A, B, C = proof
    if len(public_inputs) + 1 != len(vk["ic"]):
        raise ValueError("public input count does not match the verification key")
    for pt, grp in ((A, g1), (C, g1), (B, g2)):
        if not grp.on_curve(pt):
            return False
    acc = vk["ic"][0]
    for x, pt in zip(public_inputs, vk["ic"][1:]):
        acc = g1.add(acc, g1.mul(pt, x % FR[curve].p))
    return pairing_product_is_one(curve, [(A, B), (g1.neg(vk["alpha_g1"]), vk["beta_g2"]),
                                          (g1.neg(acc), vk["gamma_g2"]), (g1.neg(C), vk["delta_g2"])])


# ---- snarkjs JSON (decimal strings, projective with z = 1; G2 coordinates as [c0, c1]) --------------------
def _g1(j):
    x, y, z = (int(v) for v in j)
    return None if z == 0 else (x, y)


def _g2(j):
    (x0, x1), (y0, y1), (z0, z1) = ([int(v) for v in c] for c in j)
    return None if (z0, z1) == (0, 0) else ((x0, x1), (y0, y1))


def snarkjs_vk(j):
    assert j["protocol"] == "groth16" and j["curve"] == "bn128"
    return {"alpha_g1": _g1(j["vk_alpha_1"]), "beta_g2": _g2(j["vk_beta_2"]), "gamma_g2": _g2(j["vk_gamma_2"]),
            "delta_g2": _g2(j["vk_delta_2"]), "ic": [_g1(p) for p in j["IC"]]}


def snarkjs_proof(j):
    return _g1(j["pi_a"]), _g2(j["pi_b"]), _g1(j["pi_c"])
