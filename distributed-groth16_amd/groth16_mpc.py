"""Python mirror of the reference's `groth16` crate on packed shares (the n-party prover), composed
from the dist-primitives mirror (dist.py).  Names follow the reference:

    qap_pss                            QAP::pss                               groth16/src/qap.rs:143-187
    pack_from_arkworks_proving_key     PackedProvingKeyShare::...             groth16/src/proving_key.rs:35-110
    pack_from_witness                  examples/sha256.rs:97-121
    A / B / C .compute                 groth16/src/prove.rs:21-136
    party_prove                        `dsha256`                              groth16/examples/sha256.rs:26-95

All arithmetic runs on the GPU through the C ABI; this file only moves numpy arrays between calls.
"""

import numpy as np

from . import dist as D
from .lib import FQ_LIMBS64


def _bitrev_perm(n):
    bits = n.bit_length() - 1
    idx = np.arange(n)
    rev = np.zeros(n, dtype=np.int64)
    for b in range(bits):
        rev |= ((idx >> b) & 1) << (bits - 1 - b)
    return rev


def qap_pss(pp, a, b, c):
    """Per-party (a, b, c) share vectors: bit-reverse, stride-(m/l) l-tuples, pack (qap.rs:152-165)."""
    out = []
    for v in (a, b, c):
        v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
        m = v.shape[0]
        x = v[_bitrev_perm(m)]                                  # fft_in_place_rearrange
        tuples = x.reshape(pp.l, m // pp.l, 4).transpose(1, 0, 2)   # [i] = (x[i], x[i + m/l], ..)
        out.append(pp.pack_from_public(np.ascontiguousarray(tuples)))  # [m/l][n][4]
    return [tuple(np.ascontiguousarray(o[:, i]) for o in out) for i in range(pp.n)]


def pack_from_witness(pp, assignment):
    """examples/sha256.rs:97-121: l-chunks (zero padded) packed, one vector per party."""
    a = np.ascontiguousarray(assignment, dtype=np.uint64).reshape(-1, 4)
    pad = (-a.shape[0]) % pp.l
    if pad:
        a = np.concatenate([a, np.zeros((pad, 4), dtype=np.uint64)])
    packed = pp.pack_from_public(a.reshape(-1, pp.l, 4))
    return [np.ascontiguousarray(packed[:, i]) for i in range(pp.n)]


def _packexp_chunks(pp, group, pts):
    nl = pts.shape[1]
    pad = (-pts.shape[0]) % pp.l
    if pad:
        pts = np.concatenate([pts, np.zeros((pad, nl), dtype=np.uint64)])   # identity fill
    packed = pp.packexp_from_public(group, pts.reshape(-1, pp.l, nl))
    return [np.ascontiguousarray(packed[:, i]) for i in range(pp.n)]


def pack_from_arkworks_proving_key(pp, pk):
    """proving_key.rs:48-65: s <- a_query[1..], u <- h_query, w <- l_query, h <- b_g1_query[1..],
    v <- b_g2_query[1..]; returns one dict per party."""
    s = _packexp_chunks(pp, 1, pk["a_query"][1:])
    u = _packexp_chunks(pp, 1, pk["h_query"])
    w = _packexp_chunks(pp, 1, pk["l_query"])
    h = _packexp_chunks(pp, 1, pk["b_g1_query"][1:])
    v = _packexp_chunks(pp, 2, pk["b_g2_query"][1:])
    return [dict(s=s[i], u=u[i], w=w[i], h=h[i], v=v[i]) for i in range(pp.n)]


def party_prove(ctx, pp, net, crs_share, qap_share, a_share, ax_share, log_m):
    """One party of the reference's example (sha256.rs:26-95), r = s = 0 and default L, N, Z, K, M like
    there: returns (pi_a, pi_b, pi_c) Jacobian, identical on all parties."""
    h_share = D.ext_wit_h(ctx, pp, net, qap_share[0], qap_share[1], qap_share[2], log_m)     # :41
    pi_a = D.d_msm(ctx, pp, net, 1, crs_share["s"], a_share, sid=0)                              # A::compute
    pi_b = D.d_msm(ctx, pp, net, 2, crs_share["v"], a_share, sid=0)                              # B::compute
    w = D.d_msm(ctx, pp, net, 1, crs_share["w"], ax_share, sid=0)                                # C::compute
    u = D.d_msm(ctx, pp, net, 1, crs_share["u"], h_share, sid=1)
    return pi_a, pi_b, (w, u)
