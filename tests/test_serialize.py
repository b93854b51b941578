"""CPU: arkworks compressed proof encoding against the reference's real vector: the 128 bytes of
zk-cli/test-circuits/sha256/proof.bin (tests/golden/proof_bin_sha256.json) and the SAME proof printed in full
by the reference's CLI (zk-cli/README.md:82, decimal coordinates below).  Decoding must give exactly those
coordinates -- this pins the flag bits and the Fq2 sign convention -- and encoding them must give the file back."""

import json
import os

import pytest

import dg16_amd  # noqa: F401
from dg16_amd import serialize as S

FIX = os.path.join(os.path.dirname(__file__), "golden", "proof_bin_sha256.json")

# zk-cli/README.md:82
A = (498071793091850774982818679555333756485311255358673548654120302530182399203,
     1136016735329476510342430929598492059239405464600443570644253133992037045925)
B = ((16201123471192943709002890652913953088196139179085638192299840215083499178123,
      7661856193817045474354410892358379917961382296969612969639627312326949738602),
     (11727749940968147879747689210926573677381488818108937744933493916060120042583,
      1237642009120804369693292964353255765787557573751682535832640917644454968770))
C = (18296404575724821858336072059485956779860719965913072410314979178700813490716,
     17324311802124735498069101796568711680471649265468908505919699644485724510689)


def raw():
    with open(FIX) as f:
        return bytes.fromhex(json.load(f)["hex"])


def test_reference_proof_bin_decodes_to_the_printed_points():
    assert S.proof_from_bytes(raw()) == (A, B, C)


def test_printed_points_encode_to_the_reference_proof_bin():
    assert S.proof_to_bytes(A, B, C) == raw()


def test_negated_points_and_identity_round_trip():
    nA = (A[0], S.Q - A[1])
    nB = (B[0], ((S.Q - B[1][0]) % S.Q, (S.Q - B[1][1]) % S.Q))
    enc = S.proof_to_bytes(nA, nB, None)
    assert enc[:32] != raw()[:32] and enc[32:96] != raw()[32:96]      # only the sign flag differs
    assert bytes([enc[31] ^ 0x80]) == raw()[31:32]
    assert S.proof_from_bytes(enc) == (nA, nB, None)


def test_malformed_encodings_are_rejected():
    r = bytearray(raw())
    with pytest.raises(S.SerializationError):
        S.proof_from_bytes(bytes(r[:127]))
    bad = bytearray(r)
    bad[31] |= 0xC0                       # both flags
    with pytest.raises(S.SerializationError):
        S.proof_from_bytes(bytes(bad))
    bad = bytearray(r)
    bad[0:31] = b"\xff" * 31
    bad[31] = 0x3F                        # x >= q
    with pytest.raises(S.SerializationError):
        S.proof_from_bytes(bytes(bad))
    # an x with no y on the curve
    x = 1
    while S._sqrt_fq((x ** 3 + 3) % S.Q) is not None:
        x += 1
    with pytest.raises(S.SerializationError):
        S.g1_from_bytes(x.to_bytes(32, "little"))
    # a curve point of E'(Fq2) outside the order-r subgroup
    x = (1, 0)
    while True:
        y = S._sqrt_fq2(S._G2.add(S._f2_mul(S._f2_mul(x, x), x), S._B2))
        if y is not None and not S._in_subgroup(S._G2, (x, y)):
            break
        x = (x[0] + 1, 0)
    with pytest.raises(S.SerializationError):
        S.g2_from_bytes(S.g2_to_bytes((x, y)))


# ---- the native implementation of libdg16 (host code, runs without a GPU) ---------------------------------
RMONT = (1 << 256) % S.Q


def limbs(v):
    import numpy as np
    return np.frombuffer((v * RMONT % S.Q).to_bytes(32, "little"), dtype=np.uint64)


def unlimbs(row):
    return int.from_bytes(row.tobytes(), "little") * pow(RMONT, -1, S.Q) % S.Q


def test_native_decompress_gives_the_printed_points():
    out = S.decompress_to_limbs(raw())
    got = [unlimbs(r) for r in out]
    assert got == [A[0], A[1], B[0][0], B[0][1], B[1][0], B[1][1], C[0], C[1]]


def test_native_compress_of_jacobian_limbs_gives_the_reference_proof_bin():
    import numpy as np
    z1, z2, z3 = 0x1234567, (5, 7), 0xABCDEF987            # arbitrary non-trivial Jacobian denominators
    f2 = S._f2_mul
    za = [A[0] * z1 ** 2 % S.Q, A[1] * z1 ** 3 % S.Q, z1]
    zz2 = f2(z2, z2)
    zb = [f2(B[0], zz2), f2(B[1], f2(zz2, z2)), z2]
    zc = [C[0] * z3 ** 2 % S.Q, C[1] * z3 ** 3 % S.Q, z3]
    flat = za + [c for e in zb for c in e] + zc
    jac = np.stack([limbs(v) for v in flat])
    assert S.compress_gpu_proof(jac) == raw()
    # identity (z = 0) encodes as the infinity flag, and decodes back to zeros
    jac0 = jac.copy()
    jac0[2] = 0
    enc = S.compress_gpu_proof(jac0)
    assert enc[:32] == bytes(31) + b"\x40" and enc[32:] == raw()[32:]
    assert not S.decompress_to_limbs(enc)[:2].any()


def test_native_and_python_decoders_agree_on_malformed_input():
    r = bytearray(raw())
    bad_cases = []
    b = bytearray(r); b[31] |= 0xC0; bad_cases.append(bytes(b))
    b = bytearray(r); b[0:31] = b"\xff" * 31; b[31] = 0x3F; bad_cases.append(bytes(b))
    x = 1
    while S._sqrt_fq((x ** 3 + 3) % S.Q) is not None:
        x += 1
    bad_cases.append(x.to_bytes(32, "little") + bytes(r[32:]))
    x = (1, 0)
    while True:
        y = S._sqrt_fq2(S._G2.add(S._f2_mul(S._f2_mul(x, x), x), S._B2))
        if y is not None and not S._in_subgroup(S._G2, (x, y)):
            break
        x = (x[0] + 1, 0)
    off_subgroup = bytes(r[:32]) + S.g2_to_bytes((x, y)) + bytes(r[96:])
    bad_cases.append(off_subgroup)
    for case in bad_cases:
        with pytest.raises(S.SerializationError):
            S.proof_from_bytes(case)
        with pytest.raises(S.SerializationError):
            S.decompress_to_limbs(case)
    # without validation the off-subgroup point decodes (Validate::No)
    assert S.decompress_to_limbs(off_subgroup, validate=False).any()
