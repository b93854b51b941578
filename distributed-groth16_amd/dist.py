"""Python mirror of the reference's dist-primitives / secret-sharing / mpc-net surface, bound to the
C ABI (include/dg16.h, csrc/dist.hip).  Names and argument meaning follow the reference:

    PackedSharingParams(ctx, curve, l)      secret-sharing/src/pss.rs:13-148
    LocalTestNet(n) / .party(i)             mpc-net/src/multi.rs:227-329
    d_fft / d_ifft / d_msm / d_pp / deg_red dist-primitives/src/{dfft,dmsm,dpp,utils}
    ext_wit_h                               groth16/src/ext_wit.rs:16-101

Every party is one caller with its own Context (the reference's parties are tokio tasks; here they
are host threads -- ctypes releases the GIL during the calls, so the rendezvous inside the library
works).  Host numpy arrays in and out; all compute on the GPU.
"""

import ctypes
import threading

import numpy as np

from .lib import CURVES, FQ_LIMBS64, Context, load, _ptr  # noqa: F401


class PackedSharingParams:
    def __init__(self, ctx, curve, l):
        self.ctx, self.curve, self.l, self.t, self.n = ctx, curve, l, l - 1, 4 * l
        h = ctypes.c_void_p()
        ctx._chk(ctx.L.dg16_pss_create(ctx.h, CURVES[curve], l, ctypes.byref(h)))
        self.h = h

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.L.dg16_pss_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _apply(self, which, arr, cols, rows):
        arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, cols, 4)
        out = np.zeros((arr.shape[0], rows, 4), dtype=np.uint64)
        self.ctx._chk(self.ctx.L.dg16_pss_apply(self.ctx.h, self.h, which, _ptr(arr), arr.shape[0], _ptr(out), 0, 0))
        return out

    def pack_from_public(self, secrets):       # [count][l] -> [count][n]
        return self._apply(0, secrets, self.l, self.n)

    def unpack(self, shares):                  # [count][n] -> [count][l]
        return self._apply(1, shares, self.n, self.l)

    def unpack2(self, shares):
        return self._apply(2, shares, self.n, self.l)

    def _apply_exp(self, group, which, pts, cols, rows):
        nl = FQ_LIMBS64[self.curve] * 2 * (2 if group == 2 else 1)
        pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, cols, nl)
        out = np.zeros((pts.shape[0], rows, nl), dtype=np.uint64)
        self.ctx._chk(self.ctx.L.dg16_pss_apply_exp(self.ctx.h, self.h, group, which, _ptr(pts), pts.shape[0],
                                                    _ptr(out), 0, 0))
        return out

    def packexp_from_public(self, group, secrets):
        return self._apply_exp(group, 0, secrets, self.l, self.n)

    def unpackexp(self, group, shares, degree2):
        return self._apply_exp(group, 2 if degree2 else 1, shares, self.n, self.l)


class LocalTestNet:
    """n parties in one process; party(i) is the handle the i-th caller passes to the d_* functions."""

    def __init__(self, n):
        self.L = load()
        h = ctypes.c_void_p()
        rc = self.L.dg16_localnet_create(n, ctypes.byref(h))
        if rc:
            raise RuntimeError("dg16_localnet_create failed: %d" % rc)
        self.h, self.n = h, n
        self.L.dg16_localnet_reset(h, 60)

    def party(self, i):
        return ctypes.c_void_p(self.L.dg16_localnet_party(self.h, i))

    def close(self):
        if self.h:
            self.L.dg16_localnet_destroy(self.h)
        self.h = None

    def simulate_network_round(self, fn):
        """Runs fn(party_id, net_handle) on n threads (LocalTestNet::simulate_network_round,
        mpc-net/src/multi.rs:289-316) and returns the results ordered by party id."""
        out, err = [None] * self.n, [None] * self.n

        def run(i):
            try:
                out[i] = fn(i, self.party(i))
            except BaseException as e:   # noqa: BLE001
                err[i] = e
                self.L.dg16_localnet_abort(self.h)   # peers fail fast instead of waiting for us

        th = [threading.Thread(target=run, args=(i,)) for i in range(self.n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if any(e is not None for e in err):
            self.L.dg16_localnet_reset(self.h, 0)
            raise next(e for e in err if e is not None)
        return out


def _fr(a):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)


def d_fft(ctx, pp, net, share, log_m, rearrange, pad, degree2, inverse=False, sid=0):
    share = _fr(share)
    out = np.zeros((pad * share.shape[0], 4), dtype=np.uint64)
    ctx._chk(ctx.L.dg16_d_fft(ctx.h, pp.h, net, _ptr(share), share.shape[0], log_m, int(rearrange), pad,
                              int(degree2), int(inverse), _ptr(out), 0, sid))
    return out


def d_ifft(ctx, pp, net, share, log_m, rearrange, pad, degree2, sid=0):
    return d_fft(ctx, pp, net, share, log_m, rearrange, pad, degree2, inverse=True, sid=sid)


def d_msm(ctx, pp, net, group, bases, scalars, scalars_mont=True, sid=0, in_subgroup=False):
    """in_subgroup: DG16_F_BASES_IN_SUBGROUP -- pass True when the base shares are combinations of CRS points (packed
    shares of a proving key are: in the order-r group), so that the local MSM may split its scalars; the default is the
    C ABI's (flag not set: the result VariableBaseMSM::msm gives for any points of the curve)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = _fr(scalars)
    nl = FQ_LIMBS64[pp.curve] * (2 if group == 2 else 1)
    out = np.zeros((1, 3 * nl), dtype=np.uint64)
    ctx._chk(ctx.L.dg16_d_msm(ctx.h, pp.h, net, group, _ptr(bases), _ptr(scalars), bases.shape[0], scalars.shape[0],
                              (1 if scalars_mont else 0) | (64 if in_subgroup else 0), sid, _ptr(out)))
    return out


def d_msm_resident(ctx, pp, net, resident_bases, scalars, scalars_mont=True, sid=0):
    """d_msm over base shares uploaded once (`Context.bases_upload`): what a party does per proof with its
    PackedProvingKeyShare (groth16/src/proving_key.rs:26-46)."""
    scalars = _fr(scalars)
    nl = FQ_LIMBS64[pp.curve] * (2 if resident_bases.group == 2 else 1)
    out = np.zeros((1, 3 * nl), dtype=np.uint64)
    ctx._chk(ctx.L.dg16_d_msm_resident(ctx.h, pp.h, net, resident_bases.h, _ptr(scalars), scalars.shape[0],
                                       1 if scalars_mont else 0, sid, _ptr(out)))
    return out


def dpoly_commit(ctx, pp, net, srs_shares, coeff_shares, scalars_mont=True, sid=0, in_subgroup=False):
    """KZG-style commitment to a polynomial held as packed coefficient shares: one `d_msm` of the coefficient
    shares against the packed-in-the-exponent SRS powers [tau^i]_1.  The north star names `dpoly_commit`, but the
    reference has no such module (only the launcher scripts/dpoly_commit_test.zsh:5-7; dist-primitives/src/lib.rs:2-6
    lists dfft, dmsm, dpp, utils, channel) -- there is no behaviour to match beyond d_msm's (dmsm/mod.rs:70-98)."""
    return d_msm(ctx, pp, net, 1, srs_shares, coeff_shares, scalars_mont=scalars_mont, sid=sid, in_subgroup=in_subgroup)


def deg_red(ctx, pp, net, px, sid=0):
    px = _fr(px)
    out = np.zeros_like(px)
    ctx._chk(ctx.L.dg16_deg_red(ctx.h, pp.h, net, _ptr(px), px.shape[0], _ptr(out), 0, sid))
    return out


def d_pp(ctx, pp, net, num, den, sid=0):
    num, den = _fr(num), _fr(den)
    out = np.zeros_like(num)
    ctx._chk(ctx.L.dg16_d_pp(ctx.h, pp.h, net, _ptr(num), _ptr(den), num.shape[0], _ptr(out), 0, sid))
    return out


def ext_wit_h(ctx, pp, net, a_share, b_share, c_share, log_m):
    a, b, c = _fr(a_share), _fr(b_share), _fr(c_share)
    out = np.zeros_like(a)
    ctx._chk(ctx.L.dg16_ext_wit_h(ctx.h, pp.h, net, _ptr(a), _ptr(b), _ptr(c), log_m, _ptr(out), 0))
    return out
