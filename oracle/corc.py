"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the C oracle (oracle/c/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module
(it is the checker / the CPU baseline, never the product).  All field elements cross this
boundary as numpy uint64 arrays of shape (n, limbs) (little-endian limbs), points as
(n, 2*limbs) [G1] or (n, 4*limbs) [G2: x.c0 x.c1 y.c0 y.c1], Montgomery form unless said
otherwise, identity = all zero.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "liboracle.so")

CURVES = {"bn254": 0, "bls12_381": 1, "bls12_377": 2}
FQ_LIMBS = {"bn254": 4, "bls12_381": 6, "bls12_377": 6}


def build(force=False):
    src = [os.path.join(_HERE, "c", f) for f in
           ("oracle.c", "fp_tmpl.h", "fp2_tmpl.h", "ec_tmpl.h", "ntt_tmpl.h", "consts_gen.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "c"), "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, sz, i, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64
        L.orc_field_op.argtypes = [i, i, vp, vp, vp, sz]
        L.orc_rand_field.argtypes = [i, u64, sz, i, vp]
        L.orc_generator.argtypes = [i, i, vp]
        L.orc_point_mul.argtypes = [i, i, vp, vp, vp]
        L.orc_point_add.argtypes = [i, i, vp, vp, vp]
        L.orc_jac_to_affine.argtypes = [i, i, vp, vp]
        L.orc_on_curve.argtypes = [i, i, vp]
        L.orc_gen_points.argtypes = [i, i, u64, sz, vp, i]
        L.orc_msm.argtypes = [i, i, vp, vp, sz, i, i, i, vp]
        L.orc_ntt.argtypes = [i, vp, ctypes.c_uint, i, vp, i]
        L.orc_h_poly.argtypes = [i, vp, vp, vp, ctypes.c_uint, vp, i]
        L.orc_root_of_unity.argtypes = [i, ctypes.c_uint, vp]
        L.orc_qap.argtypes = [i, sz, sz, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i]
        L.orc_affine_bytes.restype = sz
        L.orc_affine_bytes.argtypes = [i, i]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _chk(rc):
    if rc != 0:
        raise RuntimeError("oracle call failed: %d" % rc)


# ---- int <-> limb arrays ---------------------------------------------------------------------
def ints_to_arr(vals, limbs):
    out = np.zeros((len(vals), limbs), dtype=np.uint64)
    for k, v in enumerate(vals):
        for j in range(limbs):
            out[k, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def arr_to_ints(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(len(arr), -1)
    return [sum(int(x) << (64 * j) for j, x in enumerate(row)) for row in arr]


def fid(curve, kind):
    return CURVES[curve] + (16 if kind == "fr" else 0)


def limbs_of(curve, kind):
    return 4 if kind == "fr" else FQ_LIMBS[curve]


OPS = {"add": 0, "sub": 1, "mul": 2, "sqr": 3, "inv": 4, "to_mont": 5, "from_mont": 6, "neg": 7}


def field_op(curve, kind, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    _chk(lib().orc_field_op(fid(curve, kind), OPS[op], _p(a), _p(b), _p(out), a.shape[0]))
    return out


def rand_field(curve, kind, seed, n, mont=True):
    out = np.empty((n, limbs_of(curve, kind)), dtype=np.uint64)
    _chk(lib().orc_rand_field(fid(curve, kind), seed, n, int(mont), _p(out)))
    return out


def point_limbs(curve, group):
    return FQ_LIMBS[curve] * 2 * (2 if group == 2 else 1)


def generator(curve, group):
    out = np.zeros((1, point_limbs(curve, group)), dtype=np.uint64)
    _chk(lib().orc_generator(CURVES[curve], group, _p(out)))
    return out


def point_mul(curve, group, p, k):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    kk = ints_to_arr([k], 4)
    out = np.zeros_like(p)
    _chk(lib().orc_point_mul(CURVES[curve], group, _p(p), _p(kk), _p(out)))
    return out


def point_add(curve, group, p, q):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    q = np.ascontiguousarray(q, dtype=np.uint64)
    out = np.zeros_like(p)
    _chk(lib().orc_point_add(CURVES[curve], group, _p(p), _p(q), _p(out)))
    return out


def jac_to_affine(curve, group, jac):
    jac = np.ascontiguousarray(jac, dtype=np.uint64).reshape(1, -1)
    out = np.zeros((1, point_limbs(curve, group)), dtype=np.uint64)
    _chk(lib().orc_jac_to_affine(CURVES[curve], group, _p(jac), _p(out)))
    return out


def on_curve(curve, group, p):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    return lib().orc_on_curve(CURVES[curve], group, _p(p)) == 1


def gen_points(curve, group, seed, n, threads=0):
    out = np.zeros((n, point_limbs(curve, group)), dtype=np.uint64)
    _chk(lib().orc_gen_points(CURVES[curve], group, seed, n, _p(out), threads or os.cpu_count()))
    return out


def msm(curve, group, bases, scalars, scalars_mont=False, algo=0, threads=0):
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    if bases.shape[0] != scalars.shape[0]:
        raise ValueError(min(bases.shape[0], scalars.shape[0]))
    out = np.zeros((1, point_limbs(curve, group)), dtype=np.uint64)
    _chk(lib().orc_msm(CURVES[curve], group, _p(bases), _p(scalars), bases.shape[0],
                       int(scalars_mont), algo, threads, _p(out)))
    return out


def ntt(curve, data, inverse=False, coset=None, threads=0):
    """Returns a transformed copy (Montgomery in, Montgomery out)."""
    data = np.array(data, dtype=np.uint64, copy=True)
    n = data.shape[0]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    if coset is not None:
        coset = np.ascontiguousarray(coset, dtype=np.uint64)
    _chk(lib().orc_ntt(CURVES[curve], _p(data), log_n, int(inverse), _p(coset), threads))
    return data


def h_poly(curve, a, b, c, threads=0):
    a, b, c = (np.array(v, dtype=np.uint64, copy=True) for v in (a, b, c))
    m = a.shape[0]
    log_m = m.bit_length() - 1
    out = np.empty_like(a)
    _chk(lib().orc_h_poly(CURVES[curve], _p(a), _p(b), _p(c), log_m, _p(out), threads))
    return out


def qap(curve, nc, ni, m, csr_a, csr_b, w, threads=0):
    """qap::qap on CSR matrices (row_ptr uint32, col uint32, coeff Montgomery) and a Montgomery assignment."""
    ap, ac, av = (np.ascontiguousarray(x) for x in csr_a)
    bp, bc, bv = (np.ascontiguousarray(x) for x in csr_b)
    w = np.ascontiguousarray(w, dtype=np.uint64)
    out = [np.zeros((m, 4), dtype=np.uint64) for _ in range(3)]
    _chk(lib().orc_qap(CURVES[curve], nc, ni, m, _p(ap), _p(ac), _p(av), _p(bp), _p(bc), _p(bv), _p(w),
                       _p(out[0]), _p(out[1]), _p(out[2]), threads))
    return out


def root_of_unity(curve, log_n):
    out = np.zeros((1, 4), dtype=np.uint64)
    _chk(lib().orc_root_of_unity(CURVES[curve], log_n, _p(out)))
    return out
