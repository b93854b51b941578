"""CPU: the oracle's pairing and Groth16 verifier, pinned against REAL vectors the reference ships:
snarkjs-generated (proof, public, verification key) triples copied as data into tests/golden/snarkjs_*.
Accepting an independently produced proof exercises the oracle's Fq, Fq2, G1, G2 and generator constants
end to end; the same verifier then accepts what the oracle prover (and, in tests/test_gpu_prover.py, the GPU
prover) produces from a key whose trapdoor it never sees."""

import json
import os

import pytest

from oracle.pyref import groth16 as g16
from oracle.pyref import pairing as pr
from oracle.pyref.curves import CURVES
from oracle.pyref.fields import FR

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(d):
    vk = pr.snarkjs_vk(json.load(open(os.path.join(GOLD, d, "verification_key.json"))))
    proof = pr.snarkjs_proof(json.load(open(os.path.join(GOLD, d, "proof.json"))))
    public = [int(x) for x in json.load(open(os.path.join(GOLD, d, "public.json")))]
    return vk, public, proof


def test_snarkjs_million_proof_is_accepted_and_perturbations_rejected():
    # /root/reference/fixtures/million/{proof,public,verification_key}.json
    vk, public, proof = load("snarkjs_million")
    assert pr.groth16_verify("bn254", vk, public, proof)
    assert not pr.groth16_verify("bn254", vk, [public[0] + 1], proof)
    g1 = CURVES["bn254", "g1"]
    A, B, C = proof
    assert not pr.groth16_verify("bn254", vk, public, (A, B, g1.add(C, g1.gen)))
    assert not pr.groth16_verify("bn254", vk, public, (g1.neg(A), B, C))


def test_snarkjs_vector_points_lie_in_the_prime_order_groups():
    # /root/reference/ark-circom/test-vectors/{proof,public,verification_key}.json: the verification key
    # belongs to test.zkey (zkey.rs:817-833, absent) and proof.json to another circuit (test-vectors/prove.sh),
    # so only membership is asserted -- and that the mismatched triple is rejected.
    r = FR["bn254"].p
    for d in ("snarkjs_test_vectors", "snarkjs_million"):
        vk, public, (A, B, C) = load(d)
        g1, g2 = CURVES["bn254", "g1"], CURVES["bn254", "g2"]
        for P in [A, C, vk["alpha_g1"]] + vk["ic"]:
            assert g1.on_curve(P) and g1.mul(P, r) is None
        for Q in [B, vk["beta_g2"], vk["gamma_g2"], vk["delta_g2"]]:
            assert g2.on_curve(Q) and g2.mul(Q, r) is None
    vk, public, proof = load("snarkjs_test_vectors")
    assert not pr.groth16_verify("bn254", vk, public, proof)
    # snarkjs' gamma is the G2 generator the oracle (and libdg16's gen_bases) uses
    assert vk["gamma_g2"] == CURVES["bn254", "g2"].gen


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_pairing_is_bilinear_and_non_degenerate(curve):
    g1, g2 = CURVES[curve, "g1"], CURVES[curve, "g2"]
    K = pr.Fq12(curve)
    e = pr.pairing(curve, g1.gen, g2.gen)
    assert e != K.one
    assert K.pow(e, FR[curve].p) == K.one
    a, b = 0x1234567, 0x7654321ABC
    assert pr.pairing(curve, g1.mul(g1.gen, a), g2.mul(g2.gen, b)) == K.pow(e, a * b)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_oracle_prover_output_verifies_under_the_pairing(curve):
    F = FR[curve]
    r1cs, w = g16.synthetic_r1cs(F, 13, 2, 11, seed=5)
    trapdoor = (11, 22, 33, 44, 987654321)
    pk, _ = g16.setup(curve, r1cs, trapdoor)
    vk = {"alpha_g1": pk["alpha_g1"], "beta_g2": pk["beta_g2"], "gamma_g2": pk["gamma_g2"],
          "delta_g2": pk["delta_g2"], "ic": pk["gamma_abc_g1"]}
    proof = g16.create_proof(curve, pk, 31337, 271828, r1cs, w)
    public = w[1:r1cs["num_instance"]]
    assert pr.groth16_verify(curve, vk, public, proof)
    assert not pr.groth16_verify(curve, vk, [(public[0] + 1) % F.p], proof)
