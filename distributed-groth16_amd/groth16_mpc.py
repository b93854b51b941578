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
from .lib import FQ_LIMBS64, _ptr


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


def _pt(curve, group, p):
    """An affine point in the clear; None = arkworks' `Default` (the identity, stored as (0, 0))."""
    nl = FQ_LIMBS64[curve] * 2 * (2 if group == 2 else 1)
    if p is None:
        return np.zeros((1, nl), dtype=np.uint64)
    return np.ascontiguousarray(p, dtype=np.uint64).reshape(1, nl)


def _sc(x):
    return np.ascontiguousarray(x, dtype=np.uint64).reshape(1, 4)


def _shares(v, cols):
    return np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, cols)


class A:
    """prove::A (groth16/src/prove.rs:10-46): A = L + N * r + d_msm(S, a).  L, N: G1 affine in the clear (None =
    Default); r: one Montgomery scalar; S, a: this party's packed shares."""

    def __init__(self, L, N, r, pp, S, a):
        self.L, self.N, self.r, self.pp, self.S, self.a = L, N, r, pp, S, a

    def compute(self, ctx, net, sid=0):
        curve = self.pp.curve
        nl = FQ_LIMBS64[curve] * 2
        S, a = _shares(self.S, nl), _shares(self.a, 4)
        out = np.zeros((1, 3 * nl // 2), dtype=np.uint64)
        ctx._chk(ctx.L.dg16_prove_a(ctx.h, self.pp.h, net, _ptr(_pt(curve, 1, self.L)), _ptr(_pt(curve, 1, self.N)),
                                    _ptr(_sc(self.r)), _ptr(S), _ptr(a), S.shape[0], a.shape[0], 1 | 64, sid, _ptr(out)))
        return out


class B:
    """prove::B (prove.rs:50-85): B = Z + K * s + d_msm(V, a) in G2."""

    def __init__(self, Z, K, s, pp, V, a):
        self.Z, self.K, self.s, self.pp, self.V, self.a = Z, K, s, pp, V, a

    def compute(self, ctx, net, sid=0):
        curve = self.pp.curve
        nl = FQ_LIMBS64[curve] * 4
        V, a = _shares(self.V, nl), _shares(self.a, 4)
        out = np.zeros((1, 3 * nl // 2), dtype=np.uint64)
        ctx._chk(ctx.L.dg16_prove_b(ctx.h, self.pp.h, net, _ptr(_pt(curve, 2, self.Z)), _ptr(_pt(curve, 2, self.K)),
                                    _ptr(_sc(self.s)), _ptr(V), _ptr(a), V.shape[0], a.shape[0], 1 | 64, sid, _ptr(out)))
        return out


class C:
    """prove::C (prove.rs:89-136): C = d_msm(W, ax) + d_msm(U, h) + A * s + M * r + d_msm(H, a) * r, the three d_msm
    joined on channels 0 / 1 / 2.  A: the G1 Jacobian A::compute returned; M: G1 affine in the clear."""

    def __init__(self, A, M, s, r, pp, W, U, H, a, ax, h):
        self.A, self.M, self.s, self.r, self.pp = A, M, s, r, pp
        self.W, self.U, self.H, self.a, self.ax, self.h = W, U, H, a, ax, h

    def compute(self, ctx, net, serial_channels=False):
        """serial_channels: DG16_F_SERIAL_CHANNELS -- the three d_msm one after another (channels 0, 1, 2 in that order
        on every party) for a transport whose channels are not independent; lib.probe_channels decides it collectively."""
        curve = self.pp.curve
        nl = FQ_LIMBS64[curve] * 2
        W, U, H = (_shares(v, nl) for v in (self.W, self.U, self.H))
        a, ax, h = (_shares(v, 4) for v in (self.a, self.ax, self.h))
        Aj = np.ascontiguousarray(self.A, dtype=np.uint64).reshape(1, 3 * nl // 2)
        out = np.zeros((1, 3 * nl // 2), dtype=np.uint64)
        ctx._chk(ctx.L.dg16_prove_c(ctx.h, self.pp.h, net, _ptr(Aj), _ptr(_pt(curve, 1, self.M)), _ptr(_sc(self.s)),
                                    _ptr(_sc(self.r)), _ptr(W), _ptr(ax), W.shape[0], ax.shape[0], _ptr(U), _ptr(h),
                                    U.shape[0], h.shape[0], _ptr(H), _ptr(a), H.shape[0], a.shape[0],
                                    1 | 64 | (16 if serial_channels else 0), _ptr(out)))
        return out


def party_prove(ctx, pp, net, crs_share, qap_share, a_share, ax_share, log_m, r=None, s=None, L=None, N=None,
                Z=None, K=None, M=None, serial_channels=False):
    """One party of the reference's example (`dsha256`, groth16/examples/sha256.rs:26-95): h = ext_wit::h, then
    A::compute, B::compute, C::compute exactly as there.  The example passes r = s = 0 and Default points
    (:46-48, :60-62, :77-79) -- the defaults here; any other values go through the same native entry points.
    Returns (pi_a, pi_b, pi_c) Jacobian, identical on all parties."""
    zero = np.zeros((1, 4), dtype=np.uint64)
    r = zero if r is None else r
    s = zero if s is None else s
    h_share = D.ext_wit_h(ctx, pp, net, qap_share[0], qap_share[1], qap_share[2], log_m)     # :41
    pi_a = A(L, N, r, pp, crs_share["s"], a_share).compute(ctx, net, 0)                      # :45-57
    pi_b = B(Z, K, s, pp, crs_share["v"], a_share).compute(ctx, net, 0)                      # :59-71
    pi_c = C(pi_a, M, s, r, pp, crs_share["w"], crs_share["u"], crs_share["h"], a_share, ax_share,
             h_share).compute(ctx, net, serial_channels=serial_channels)                     # :73-92
    return pi_a, pi_b, pi_c
