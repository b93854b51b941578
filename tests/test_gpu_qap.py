"""GPU parity of dg16_qap (groth16/src/qap.rs:44-91) and the full R1CS -> proof path on the real circom
circuit of the reference tree (10 000 constraints), against the big-int restatement / C oracle."""

import os
import random

import numpy as np
import pytest

from oracle import corc
from oracle.pyref.fields import FQ, FR
from oracle.pyref import groth16 as G
from gpu_util import ctx
from test_r1cs_reader import FIX, witness_for_complex_circuit

pytestmark = pytest.mark.gpu


def enc(F, vals):
    return corc.ints_to_arr([F.to_mont(v) for v in vals], 4)


def dec(F, arr):
    return [F.from_mont(v) for v in corc.arr_to_ints(arr)]


def test_qap_synthetic_matches_bigint():
    F = FR["bn254"]
    r1cs, w = G.synthetic_r1cs(F, num_constraints=300, num_instance=3, num_witness=200, seed=8, nnz=4)
    a, b, c, dom = G.qap(r1cs, w, F)

    def csr(rows):
        ptr, col, val = [0], [], []
        for row in rows:
            for cf, idx in row:
                col.append(idx)
                val.append(cf)
            ptr.append(len(col))
        return np.asarray(ptr, dtype=np.uint32), np.asarray(col, dtype=np.uint32), enc(F, val)

    for mont in (True, False):
        wa = enc(F, w) if mont else corc.ints_to_arr([x % F.p for x in w], 4)
        ga, gb, gc = ctx().qap("bn254", 300, 3, csr(r1cs["a"]), csr(r1cs["b"]), wa, scalars_mont=mont)
        assert (dec(F, ga), dec(F, gb), dec(F, gc)) == (a, b, c)


def test_real_circuit_r1cs_to_proof():
    import dg16_amd  # noqa: F401
    from dg16_amd.r1cs import R1CS
    curve = "bn254"
    F, Fq = FR[curve], FQ[curve]
    r = R1CS.from_file(FIX)
    w = witness_for_complex_circuit(r, F.p)
    c_ = ctx()
    # coefficients arrive canonical (deserialize_uncompressed, r1cs_reader.rs:196-207): to Montgomery on the GPU
    csr_a = (r.csr[0][0], r.csr[0][1], c_.field_op(curve, "fr", "to_mont", r.csr[0][2]))
    csr_b = (r.csr[1][0], r.csr[1][1], c_.field_op(curve, "fr", "to_mont", r.csr[1][2]))
    W = enc(F, w)
    ga, gb, gc = c_.qap(curve, r.n_constraints, r.num_inputs, csr_a, csr_b, W)
    r1cs = dict(num_instance=r.num_inputs, num_witness=r.num_aux, num_constraints=r.n_constraints,
                a=r.rows(0), b=r.rows(1), c=r.rows(2))
    a, b, c, dom = G.qap(r1cs, w, F)
    assert (dec(F, ga), dec(F, gb), dec(F, gc)) == (a, b, c)
    m = dom.size
    assert m == 16384
    # h on the GPU == oracle h; and a proof with a synthetic key (like PackedProvingKeyShare::rand) equals the
    # oracle's proof of the same inputs
    h = c_.h_poly(curve, ga, gb, gc)
    assert np.array_equal(h, corc.h_poly(curve, ga, gb, gc))
    nv, ni = r.n_wires, r.num_inputs
    aq, b1q, b2q = corc.gen_points(curve, 1, 31, nv), corc.gen_points(curve, 1, 32, nv), corc.gen_points(curve, 2, 33, nv)
    hq, lq = corc.gen_points(curve, 1, 34, m), corc.gen_points(curve, 1, 35, nv - ni)
    f1, f2 = corc.gen_points(curve, 1, 36, 3), corc.gen_points(curve, 2, 37, 2)
    pk = c_.pk_create(curve, nv, ni, m, aq, b1q, b2q, hq, lq, np.concatenate([f1.reshape(-1), f2.reshape(-1)]))
    rng = random.Random(3)
    rr, ss = rng.randrange(1, F.p), rng.randrange(1, F.p)
    A, B, C = c_.prove(pk, ga, gb, gc, W, enc(F, [rr]), enc(F, [ss]))
    wc = corc.ints_to_arr([x % F.p for x in w], 4)
    hc = corc.field_op(curve, "fr", "from_mont", h)
    add = lambda g, p, q: corc.point_add(curve, g, p, q)
    mul = lambda g, p, k: corc.point_mul(curve, g, p, k)
    msm = lambda g, bases, sc: corc.msm(curve, g, bases, sc)
    eA = add(1, add(1, msm(1, aq[1:], wc[1:]), aq[0:1]), add(1, f1[0:1], mul(1, f1[2:3], rr)))
    eB1 = add(1, add(1, msm(1, b1q[1:], wc[1:]), b1q[0:1]), add(1, f1[1:2], mul(1, f1[2:3], ss)))
    eB = add(2, add(2, msm(2, b2q[1:], wc[1:]), b2q[0:1]), add(2, f2[0:1], mul(2, f2[1:2], ss)))
    eC = add(1, add(1, msm(1, lq, wc[ni:]), msm(1, hq, hc)),
             add(1, add(1, mul(1, eA, ss), mul(1, eB1, rr)), mul(1, f1[2:3], (F.p - rr * ss % F.p) % F.p)))
    assert np.array_equal(corc.jac_to_affine(curve, 1, A), eA)
    assert np.array_equal(corc.jac_to_affine(curve, 2, B), eB)
    assert np.array_equal(corc.jac_to_affine(curve, 1, C), eC)
    pk.close()
