"""Groth16 verification (BN254) through libdg16's native verifier (`dg16_groth16_verify`, csrc/verify.hip) -- the
counterpart of `Groth16::<Bn254>::verify_proof` as the reference calls it (groth16/examples/sha256.rs:228-254, the
verify endpoint of mpc-api/src/main.rs, zk-cli verify).  Points are affine x || y Montgomery limbs (uint64), the
layout of a zkey's header / IC section and of `serialize.decompress_to_limbs`."""

import ctypes

import numpy as np

from . import lib as _lib


def verify_proof(alpha_g1, beta_g2, gamma_g2, delta_g2, ic, public_inputs, proof_affine, scalars_mont=False):
    """ic: (n_public + 1) x 8 uint64; public_inputs: n_public x 4 uint64; proof_affine: 32 uint64
    (A.x A.y | B.x B.y | C.x C.y).  Returns True / False; raises on a malformed verification key."""
    L = _lib.load()
    arr = lambda x: np.ascontiguousarray(x, dtype=np.uint64)
    alpha_g1, beta_g2, gamma_g2, delta_g2 = arr(alpha_g1), arr(beta_g2), arr(gamma_g2), arr(delta_g2)
    ic = arr(ic).reshape(-1, 8)
    pub = arr(public_inputs).reshape(-1, 4)
    proof = arr(proof_affine).reshape(-1)
    if proof.size != 32:
        raise ValueError("a BN254 proof is 8 field elements in affine form")
    ok = ctypes.c_int(0)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    rc = L.dg16_groth16_verify(0, p(alpha_g1), p(beta_g2), p(gamma_g2), p(delta_g2), p(ic), ic.shape[0], p(pub),
                               pub.shape[0], p(proof), _lib.F_SCALARS_MONT if scalars_mont else 0, ctypes.byref(ok))
    if rc != 0:
        raise _lib.Dg16Error(rc, L.dg16_verify_error().decode())
    return bool(ok.value)


def verify_with_zkey(zkey, public_inputs, proof_affine, scalars_mont=False):
    """Verification key taken from a parsed `.zkey` (zkey.ZKey): alpha, beta, gamma, delta and IC as they lie in it."""
    return verify_proof(zkey.alpha_g1, zkey.beta_g2, zkey.gamma_g2, zkey.delta_g2, zkey.ic, public_inputs,
                        proof_affine, scalars_mont=scalars_mont)
