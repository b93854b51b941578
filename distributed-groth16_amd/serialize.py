"""arkworks compressed (de)serialisation of BN254 Groth16 proofs -- the `proof.bin` the reference's service writes
and reads (`Proof::<Bn254>::serialize_compressed` / `deserialize_compressed`, mpc-api/src/main.rs:154-171,
zk-cli/test-circuits/sha256/proof.bin): A (G1, 32 B) || B (G2, 64 B) || C (G1, 32 B).

ark-ec short-Weierstrass flags live in the two top bits of the LAST byte of the little-endian x coordinate:
bit 7 = "y is the negative root" (y > -y; for Fq2 the comparison is lexicographic on (c1, c0)), bit 6 = point at
infinity (x = 0).  G2's x is c0 || c1.  Decoding validates like `Validate::Yes`: x < q, on the curve, in the
prime-order subgroup.

Two implementations: the native one of libdg16 (`dg16_proof_compress` / `dg16_proof_decompress`, csrc/serialize.hip,
on the Montgomery limbs the GPU prover writes -- `compress_gpu_proof` / `decompress_to_limbs` below) and this
module's plain-Python one on integers; the tests hold them against each other and against the reference's real
proof.bin.  Host-side format code on three points per proof; nothing here is on the proving path."""

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class SerializationError(ValueError):
    pass


# ---- Fq / Fq2 (u^2 = -1) ------------------------------------------------------------------------------------
def _sqrt_fq(a):
    y = pow(a, (Q + 1) // 4, Q)            # q = 3 mod 4
    return y if y * y % Q == a % Q else None


def _f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def _f2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % Q, Q - 2, Q)
    return (a[0] * n % Q, -a[1] * n % Q)


def _sqrt_fq2(a):
    a0, a1 = a
    if a1 == 0:
        r = _sqrt_fq(a0)
        if r is not None:
            return (r, 0)
        r = _sqrt_fq(-a0 % Q)
        return None if r is None else (0, r)
    n = _sqrt_fq((a0 * a0 + a1 * a1) % Q)
    if n is None:
        return None
    inv2 = (Q + 1) // 2
    for delta in ((a0 + n) * inv2 % Q, (a0 - n) * inv2 % Q):
        x0 = _sqrt_fq(delta)
        if x0 is not None and x0 != 0:
            x1 = a1 * pow(2 * x0 % Q, Q - 2, Q) % Q
            if _f2_mul((x0, x1), (x0, x1)) == (a0 % Q, a1 % Q):
                return (x0, x1)
    return None


_B1 = 3
_B2 = _f2_mul((3, 0), _f2_inv((9, 1)))     # twist: y^2 = x^3 + 3 / (9 + u)


class _G1:
    b = _B1
    zero, one = 0, 1
    mul = staticmethod(lambda a, b: a * b % Q)
    add = staticmethod(lambda a, b: (a + b) % Q)
    sub = staticmethod(lambda a, b: (a - b) % Q)
    inv = staticmethod(lambda a: pow(a, Q - 2, Q))
    neg = staticmethod(lambda a: -a % Q)
    sqrt = staticmethod(_sqrt_fq)
    is_neg = staticmethod(lambda y: y > (-y % Q))          # ark: y > -y  <=> YIsNegative
    nbytes = 32

    @staticmethod
    def to_bytes(x):
        return x.to_bytes(32, "little")

    @staticmethod
    def from_bytes(b):
        x = int.from_bytes(b, "little")
        if x >= Q:
            raise SerializationError("coordinate not reduced")
        return x


class _G2:
    b = _B2
    zero, one = (0, 0), (1, 0)
    mul = staticmethod(_f2_mul)
    add = staticmethod(lambda a, b: ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q))
    sub = staticmethod(lambda a, b: ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q))
    inv = staticmethod(_f2_inv)
    neg = staticmethod(lambda a: (-a[0] % Q, -a[1] % Q))
    sqrt = staticmethod(_sqrt_fq2)
    nbytes = 64

    @staticmethod
    def is_neg(y):                                          # QuadExtField ordering: c1 first, then c0
        n = (-y[0] % Q, -y[1] % Q)
        return (y[1], y[0]) > (n[1], n[0])

    @staticmethod
    def to_bytes(x):
        return x[0].to_bytes(32, "little") + x[1].to_bytes(32, "little")

    @staticmethod
    def from_bytes(b):
        c0, c1 = int.from_bytes(b[:32], "little"), int.from_bytes(b[32:], "little")
        if c0 >= Q or c1 >= Q:
            raise SerializationError("coordinate not reduced")
        return (c0, c1)


def _add(G, P, S):
    if P is None:
        return S
    if S is None:
        return P
    if P[0] == S[0]:
        if P[1] != S[1] or P[1] == G.zero:
            return None
        x2 = G.mul(P[0], P[0])
        lam = G.mul(G.add(G.add(x2, x2), x2), G.inv(G.add(P[1], P[1])))
    else:
        lam = G.mul(G.sub(S[1], P[1]), G.inv(G.sub(S[0], P[0])))
    x3 = G.sub(G.sub(G.mul(lam, lam), P[0]), S[0])
    return (x3, G.sub(G.mul(lam, G.sub(P[0], x3)), P[1]))


def _in_subgroup(G, P):
    acc = None
    for bit in bin(R)[2:]:
        acc = _add(G, acc, acc)
        if bit == "1":
            acc = _add(G, acc, P)
    return acc is None


def _encode(G, P):
    if P is None:
        out = bytearray(G.nbytes)
        out[-1] |= 0x40
        return bytes(out)
    out = bytearray(G.to_bytes(P[0]))
    if G.is_neg(P[1]):
        out[-1] |= 0x80
    return bytes(out)


def _decode(G, raw, check_subgroup=True):
    if len(raw) != G.nbytes:
        raise SerializationError("wrong length")
    b = bytearray(raw)
    neg, inf = bool(b[-1] & 0x80), bool(b[-1] & 0x40)
    b[-1] &= 0x3F
    if neg and inf:
        raise SerializationError("invalid flags")
    x = G.from_bytes(bytes(b))
    if inf:
        if x != G.zero:
            raise SerializationError("infinity flag with a non-zero x")
        return None
    y = G.sqrt(G.add(G.mul(G.mul(x, x), x), G.b))
    if y is None:
        raise SerializationError("x is not on the curve")
    if G.is_neg(y) != neg:
        y = G.neg(y)
    P = (x, y)
    if check_subgroup and G is _G2 and not _in_subgroup(G, P):      # G1 has cofactor 1
        raise SerializationError("point is not in the prime-order subgroup")
    return P


def proof_to_bytes(a, b, c):
    """Affine (x, y) of A and C, ((x0, x1), (y0, y1)) of B, or None for the identity -> 128 bytes."""
    return _encode(_G1, a) + _encode(_G2, b) + _encode(_G1, c)


def proof_from_bytes(raw):
    if len(raw) != 128:
        raise SerializationError("a compressed BN254 proof is 128 bytes")
    return _decode(_G1, raw[:32]), _decode(_G2, raw[32:96]), _decode(_G1, raw[96:])


def g1_to_bytes(p):
    return _encode(_G1, p)


def g2_to_bytes(p):
    return _encode(_G2, p)


def g1_from_bytes(raw):
    return _decode(_G1, raw)


def g2_from_bytes(raw):
    return _decode(_G2, raw)


# ---- native path (libdg16): straight from / to the limbs the GPU prover uses ----------------------------------
def compress_gpu_proof(proof_jacobian_u64):
    """12 x 4 uint64 limbs as written by `dg16_groth16_prove` (A, B, C Jacobian, Montgomery) -> 128 bytes."""
    import ctypes
    import numpy as np
    from . import lib as _lib
    L = _lib.load()
    src = np.ascontiguousarray(proof_jacobian_u64, dtype=np.uint64).reshape(-1)
    if src.size != 48:
        raise SerializationError("a BN254 proof is 12 field elements")
    out = ctypes.create_string_buffer(128)
    if L.dg16_proof_compress(0, src.ctypes.data_as(ctypes.c_void_p), out) != 0:
        raise SerializationError(L.dg16_serialize_error().decode())
    return out.raw


def decompress_to_limbs(raw, validate=True):
    """128 bytes -> 8 x 4 uint64 Montgomery limbs: A.x A.y | B.x0 B.x1 B.y0 B.y1 | C.x C.y (identity = zeros)."""
    import ctypes
    import numpy as np
    from . import lib as _lib
    L = _lib.load()
    if len(raw) != 128:
        raise SerializationError("a compressed BN254 proof is 128 bytes")
    out = np.zeros((8, 4), dtype=np.uint64)
    if L.dg16_proof_decompress(0, bytes(raw), 1 if validate else 0, out.ctypes.data_as(ctypes.c_void_p)) != 0:
        raise SerializationError(L.dg16_serialize_error().decode())
    return out
