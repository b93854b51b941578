"""CPU: the native Groth16 verifier of libdg16 (`dg16_groth16_verify`, host code) against the reference's real
snarkjs proof triple (fixtures/million, copied under tests/golden/snarkjs_million) and against the oracle's
independent pairing verifier on proofs of the oracle prover."""

import json
import os
import random

import numpy as np
import pytest

import dg16_amd  # noqa: F401
from dg16_amd import verify as V
from dg16_amd.lib import Dg16Error
from oracle.pyref import groth16 as G
from oracle.pyref import pairing as PR
from oracle.pyref.fields import FQ, FR

GOLD = os.path.join(os.path.dirname(__file__), "golden")
Fq, Fr = FQ["bn254"], FR["bn254"]


def fq(v):
    return np.frombuffer(Fq.to_bytes(v, mont=True), dtype=np.uint64)


def g1(P):
    return np.zeros(8, dtype=np.uint64) if P is None else np.concatenate([fq(P[0]), fq(P[1])])


def g2(P):
    return np.zeros(16, dtype=np.uint64) if P is None else np.concatenate([fq(P[0][0]), fq(P[0][1]), fq(P[1][0]), fq(P[1][1])])


def scalars(vals):
    return np.stack([np.frombuffer((v % Fr.p).to_bytes(32, "little"), dtype=np.uint64) for v in vals]) if vals else \
        np.zeros((0, 4), dtype=np.uint64)


def native(vk, public, proof, **kw):
    A, B, C = proof
    return V.verify_proof(g1(vk["alpha_g1"]), g2(vk["beta_g2"]), g2(vk["gamma_g2"]), g2(vk["delta_g2"]),
                          np.stack([g1(P) for P in vk["ic"]]), scalars(public),
                          np.concatenate([g1(A), g2(B), g1(C)]), **kw)


def load(d):
    vk = PR.snarkjs_vk(json.load(open(os.path.join(GOLD, d, "verification_key.json"))))
    proof = PR.snarkjs_proof(json.load(open(os.path.join(GOLD, d, "proof.json"))))
    public = [int(x) for x in json.load(open(os.path.join(GOLD, d, "public.json")))]
    return vk, public, proof


def test_reference_snarkjs_proof_is_accepted_and_perturbations_rejected():
    vk, public, (A, B, C) = load("snarkjs_million")
    assert native(vk, public, (A, B, C))
    assert not native(vk, [public[0] + 1], (A, B, C))
    assert not native(vk, public, ((A[0], Fq.p - A[1]), B, C))
    assert not native(vk, public, (A, B, A))                       # C replaced by another curve point
    assert not native(vk, public, ((A[0], A[1] + 1), B, C))        # off the curve: a rejection, not an error
    # the other triple of the reference tree is a mismatched one (see tests/test_oracle_pairing.py)
    vk2, public2, proof2 = load("snarkjs_test_vectors")
    assert not native(vk2, public2, proof2)
    with pytest.raises(Dg16Error) as e:
        native(vk, public + [1], (A, B, C))
    assert e.value.code == 1                                        # LENGTH_MISMATCH <- MalformedVerifyingKey


def test_native_and_oracle_verifiers_agree_on_oracle_proofs():
    r1cs, w = G.synthetic_r1cs(Fr, 19, 3, 14, seed=2)
    rng = random.Random(7)
    td = tuple(rng.randrange(1, Fr.p) for _ in range(5))
    pk, _ = G.setup("bn254", r1cs, td)
    vk = {"alpha_g1": pk["alpha_g1"], "beta_g2": pk["beta_g2"], "gamma_g2": pk["gamma_g2"],
          "delta_g2": pk["delta_g2"], "ic": pk["gamma_abc_g1"]}
    proof = G.create_proof("bn254", pk, rng.randrange(1, Fr.p), rng.randrange(1, Fr.p), r1cs, w)
    public = w[1:r1cs["num_instance"]]
    assert PR.groth16_verify("bn254", vk, public, proof) and native(vk, public, proof)
    bad = [public[0], (public[1] + 5) % Fr.p]
    assert not PR.groth16_verify("bn254", vk, bad, proof) and not native(vk, bad, proof)
    # Montgomery-form public inputs
    mont = np.stack([np.frombuffer(Fr.to_bytes(v, mont=True), dtype=np.uint64) for v in public])
    A, B, C = proof
    assert V.verify_proof(g1(vk["alpha_g1"]), g2(vk["beta_g2"]), g2(vk["gamma_g2"]), g2(vk["delta_g2"]),
                          np.stack([g1(P) for P in vk["ic"]]), mont, np.concatenate([g1(A), g2(B), g1(C)]),
                          scalars_mont=True)


def test_verification_key_straight_from_a_zkey():
    """vk = the header / IC sections of a snarkjs-layout zkey, used as they lie in the file; proof = proof.bin-style
    bytes through the native decoder: the reference's verify path (zk-cli verify: key file + proof.bin + public)."""
    from dg16_amd import serialize as S
    from dg16_amd.zkey import ZKey
    from test_zkey_reader import small_key
    from zkey_writer import write_zkey
    r1cs, w, pk, m = small_key(seed=5, nc=23, ni=3, nw=12)
    z = ZKey(write_zkey(pk, r1cs, m))
    rng = random.Random(3)
    proof = G.create_proof("bn254", pk, rng.randrange(1, Fr.p), rng.randrange(1, Fr.p), r1cs, w)
    raw = S.proof_to_bytes(*proof)
    limbs = S.decompress_to_limbs(raw)
    public = scalars(w[1:r1cs["num_instance"]])
    assert V.verify_with_zkey(z, public, limbs)
    public[0, 0] ^= 1
    assert not V.verify_with_zkey(z, public, limbs)


def test_non_canonical_inputs_do_not_alias():
    """x, x + r and x + 2r all fit 256 bits: arkworks rejects a non-reduced field element at deserialisation, so
    must the C ABI (public-input aliasing otherwise).  Non-reduced or off-subgroup verifying-key material is an
    error; a proof coordinate + q is a rejection."""
    vk, public, (A, B, C) = load("snarkjs_million")
    raw = lambda v: np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint64)   # noqa: E731  (no reduction)
    args = [g1(vk["alpha_g1"]), g2(vk["beta_g2"]), g2(vk["gamma_g2"]), g2(vk["delta_g2"]),
            np.stack([g1(P) for P in vk["ic"]])]
    proof = np.concatenate([g1(A), g2(B), g1(C)])
    assert V.verify_proof(*args, np.stack([raw(public[0])]), proof)
    for k in (1, 2):
        if public[0] + k * Fr.p < 1 << 256:
            with pytest.raises(Dg16Error) as e:
                V.verify_proof(*args, np.stack([raw(public[0] + k * Fr.p)]), proof)
            assert e.value.code == 3                                  # BAD_ARG
    # proof coordinate x + q (same residue, non-reduced Montgomery limbs): rejected, not accepted
    bad = proof.copy()
    ax = int.from_bytes(bad[:4].tobytes(), "little") + Fq.p
    assert ax < 1 << 256
    bad[:4] = raw(ax)
    assert not V.verify_proof(*args, np.stack([raw(public[0])]), bad)
    # verifying key: IC point with a non-reduced coordinate, and a G2 point on the twist but outside the subgroup
    ic_bad = args[4].copy()
    ic_bad[0, :4] = raw(int.from_bytes(ic_bad[0, :4].tobytes(), "little") + Fq.p)
    with pytest.raises(Dg16Error):
        V.verify_proof(args[0], args[1], args[2], args[3], ic_bad, np.stack([raw(public[0])]), proof)
    from oracle.pyref.curves import CURVES
    from dg16_amd import serialize as S
    g2c = CURVES["bn254", "g2"]
    rng = random.Random(1)
    while True:                                                        # a point of E'(Fq2) that is not in G2
        x = (rng.randrange(Fq.p), rng.randrange(Fq.p))
        y = S._sqrt_fq2(g2c.F.add(g2c.F.mul(g2c.F.mul(x, x), x), g2c.b))
        if y is not None:
            break
    Q = (x, y)
    assert g2c.mul(Q, Fr.p) is not None                                # cofactor > 1: r * Q != identity
    with pytest.raises(Dg16Error):
        V.verify_proof(args[0], args[1], g2(Q), args[3], args[4], np.stack([raw(public[0])]), proof)
    assert not V.verify_proof(*args, np.stack([raw(public[0])]), np.concatenate([g1(A), g2(Q), g1(C)]))
