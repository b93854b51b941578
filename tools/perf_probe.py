"""Device-resident timing probe (run on the GPU box, optionally under rocprofv3 --kernel-trace --stats).
usage: python tools/perf_probe.py [msm|msm2|ntt|hpoly] [log_n] [reps]"""

import os
import sys
import time

import torch

sys.path.insert(0, ".")
import dg16_amd  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "msm"
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n = 1 << log_n
dev = torch.device("cuda:0")
ctx = dg16_amd.Context(0)
import os
if os.environ.get("DG16_ONE_STREAM"):
    for ch in range(3):
        ctx.set_stream(ch, torch.cuda.current_stream().cuda_stream)
elif what != "prove":
    ctx.set_stream(0, torch.cuda.current_stream().cuda_stream)
curve = os.environ.get("CURVE", "bn254")
FQB = 32 if curve == "bn254" else 48            # bytes per base-field element
FR_TOP = 0x30644E72E131A029 if curve == "bn254" else 0x73EDA753299D7D48


def rand_fr(n):
    # uniform over [0, r_top * 2^192) with r_top = top limb of the BN254 scalar modulus: canonical,
    # and statistically indistinguishable from uniform mod r for the bucket histogram
    lo = torch.randint(-2**63, 2**63 - 1, (n, 3), dtype=torch.int64, device=dev)
    hi = torch.randint(0, FR_TOP, (n, 1), dtype=torch.int64, device=dev)
    return torch.cat([lo, hi], dim=1).contiguous()


def witness_fr(n):
    """SKEW=bits|bytes: scalars like a real circom witness (mostly booleans / small integers)."""
    skew = os.environ.get("SKEW")
    if not skew:
        return rand_fr(n)
    hi = 2 if skew == "bits" else 256
    z = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    z[:, 0] = torch.randint(0, hi, (n,), dtype=torch.int64, device=dev)
    return z.contiguous()


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return min(ts), sum(ts) / len(ts)


if what in ("msm", "msm2"):
    group = 1 if what == "msm" else 2
    pb = 2 * FQB * group
    bases = torch.empty(n * pb, dtype=torch.uint8, device=dev)
    ctx.gen_bases_dev(curve, group, int(os.environ.get('SEED', '2')), n, bases.data_ptr())
    scal = witness_fr(n)
    out = torch.empty(3 * FQB * group, dtype=torch.uint8, device=dev)
    best, avg = timed(lambda: ctx.msm_dev(curve, group, bases.data_ptr(), scal.data_ptr(), n, out.data_ptr(), in_subgroup=True))
    print("msm G%d 2^%d: best %.3f ms avg %.3f ms  -> %.1f Mpts/s ; accumulate kernel %.3f ms" %
          (group, log_n, best * 1e3, avg * 1e3, n / best / 1e6, ctx.last_kernel_ms(0, 1)))
elif what == "ntt":
    x = rand_fr(n)
    best, avg = timed(lambda: ctx.ntt_dev(curve, x.data_ptr(), log_n))
    print("ntt 2^%d: best %.3f ms avg %.3f ms -> %.1f Melem/s, %.1f GB/s algorithmic" %
          (log_n, best * 1e3, avg * 1e3, n / best / 1e6, 64 * n / best / 1e9))
elif what == "prove":
    # synthetic proving key like PackedProvingKeyShare::rand (groth16/src/proving_key.rs:112-155)
    nv, ni, m = n - 7, 2, n
    def bases(group, cnt, seed):
        t = torch.empty(cnt * 2 * FQB * group, dtype=torch.uint8, device=dev)
        ctx.gen_bases_dev(curve, group, seed, cnt, t.data_ptr())
        return t
    aq, b1q, b2q, hq, lq = bases(1, nv, 11), bases(1, nv, 12), bases(2, nv, 13), bases(1, m, 14), bases(1, nv - ni, 15)
    f1, f2 = bases(1, 3, 16), bases(2, 2, 17)
    ctx.sync(0)
    fixed = torch.cat([f1, f2])
    torch.cuda.synchronize()
    pk = ctx.pk_create(curve, nv, ni, m, aq.data_ptr(), b1q.data_ptr(), b2q.data_ptr(), hq.data_ptr(), lq.data_ptr(),
                       fixed.data_ptr(), device_ptrs=True, shard=0, n_shards=int(os.environ.get("SHARDS", "1")))
    # SHARDS=N: time what ONE rank of an N-GPU job does (shard 0's key; the record is assembled alone, so the
    # output is not a proof -- timing only; the all-gather of N x 768 B is not included)
    a, b, c, w = rand_fr(m), rand_fr(m), rand_fr(m), witness_fr(nv)
    import numpy as np
    rs = np.array([[1, 2, 3, 4], [5, 6, 7, 8]], dtype=np.uint64)
    out = torch.empty(12 * FQB, dtype=torch.uint8, device=dev)
    rec = torch.empty(ctx.results_bytes(curve), dtype=torch.uint8, device=dev)
    def run():
        if os.environ.get("SHARDS", "1") == "1":
            ctx.prove_dev(pk, a.data_ptr(), b.data_ptr(), c.data_ptr(), w.data_ptr(), rs, out.data_ptr(), scalars_mont=False)
        else:
            ctx.groth16_msms_dev(pk, a.data_ptr(), b.data_ptr(), c.data_ptr(), w.data_ptr(), rs, rec.data_ptr(), scalars_mont=False)
            ctx.groth16_assemble_dev(pk, rec.data_ptr(), 1, rs, out.data_ptr(), scalars_mont=False)
        ctx.sync(0); ctx.sync(1); ctx.sync(2)
    best, avg = timed(run)
    print(curve, "groth16 prove m=2^%d: best %.3f ms avg %.3f ms -> %.2f M constraints/s" % (log_n, best * 1e3, avg * 1e3, (m - ni) / best / 1e6))
else:
    a, b, c = rand_fr(n), rand_fr(n), rand_fr(n)
    o = torch.empty_like(a)
    best, avg = timed(lambda: ctx.h_poly_dev(curve, a.data_ptr(), b.data_ptr(), c.data_ptr(), log_n, o.data_ptr()))
    print("h_poly 2^%d: best %.3f ms avg %.3f ms" % (log_n, best * 1e3, avg * 1e3))
