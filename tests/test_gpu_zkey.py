"""GPU: a snarkjs-layout `.zkey` flows through the product path -- `ZKey` (zero-copy queries) ->
`dg16_pk_create`, coefficient section -> `dg16_qap` -> `dg16_groth16_prove` -- and the proof is accepted by the
oracle's pairing verifier under the verification key read from the same file (the path of
ark-circom/src/zkey.rs:919-970, `verifies_with_zkey_matrices`, with our own key instead of the absent test.zkey)."""

import random

import numpy as np
import pytest

from oracle import corc
from oracle.pyref import groth16 as G
from oracle.pyref import pairing as PR
from oracle.pyref.fields import FQ, FR
from gpu_util import ctx
from test_gpu_prover import dec_g1, dec_g2, enc_fr
from test_zkey_reader import small_key
from zkey_writer import write_zkey

pytestmark = pytest.mark.gpu


def test_zkey_to_verified_proof():
    from dg16_amd.zkey import ZKey
    curve = "bn254"
    F, Fq = FR[curve], FQ[curve]
    r1cs, w, pk, m = small_key(seed=9, nc=57, ni=3, nw=40)
    z = ZKey(write_zkey(pk, r1cs, m))
    c_ = ctx()
    key = z.proving_key(c_)
    csr_a, csr_b = z.matrices(c_)
    W = enc_fr(F, w)
    a, b, c = c_.qap(curve, z.num_constraints, z.num_instance_variables, csr_a, csr_b, W)
    ea, eb, ec, dom = G.qap(r1cs, w, F)
    assert [F.from_mont(v) for v in corc.arr_to_ints(a)] == ea and dom.size == z.domain_size
    rng = random.Random(1)
    r, s = rng.randrange(1, F.p), rng.randrange(1, F.p)
    A, B, C = c_.prove(key, a, b, c, W, enc_fr(F, [r]), enc_fr(F, [s]))
    proof = (dec_g1(Fq, corc.jac_to_affine(curve, 1, A)), dec_g2(Fq, corc.jac_to_affine(curve, 2, B)),
             dec_g1(Fq, corc.jac_to_affine(curve, 1, C)))
    assert proof == G.create_proof(curve, pk, r, s, r1cs, w)
    vk = {"alpha_g1": dec_g1(Fq, z.alpha_g1), "beta_g2": dec_g2(Fq, z.beta_g2), "gamma_g2": dec_g2(Fq, z.gamma_g2),
          "delta_g2": dec_g2(Fq, z.delta_g2), "ic": [dec_g1(Fq, row) for row in z.ic]}
    assert PR.groth16_verify(curve, vk, w[1:z.num_instance_variables], proof)
    key.close()


def test_reference_g2_bytes_are_the_generator_on_the_gpu():
    """ark-circom/src/zkey.rs:443-456 (g2_buf): the 128 bytes snarkjs writes for G2.one, used AS THEY ARE as an MSM
    base: 1 * P comes back as the same bytes, k * P equals the oracle's k * G2 -- the Fq2 component order and x || y
    order of the G2 layout dg16_msm / dg16_pk_create consume is the reference's.  Same for g1_buf (:430-441)."""
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")))
    curve = "bn254"
    F, Fq = FR[curve], FQ[curve]
    c_ = ctx()
    for group, key, dec in ((1, "zkey_g1_buf", dec_g1), (2, "zkey_g2_buf", dec_g2)):
        raw = np.frombuffer(bytes(kat[key]), dtype=np.uint64).reshape(1, -1).copy()
        bases = np.repeat(raw, 3, axis=0)
        one = c_.msm(curve, group, bases[:1], corc.ints_to_arr([1], 4), affine=True)
        assert np.array_equal(one.reshape(-1), raw.reshape(-1))
        ks = [5, 7, F.p - 3]
        got = c_.msm(curve, group, bases, corc.ints_to_arr(ks, 4), affine=True)
        gen = corc.generator(curve, group)
        assert np.array_equal(np.asarray(gen).reshape(-1), raw.reshape(-1))
        want = corc.point_mul(curve, group, gen, sum(ks) % F.p)
        assert np.array_equal(got.reshape(-1), np.asarray(want).reshape(-1))
        assert dec(Fq, got) is not None
