#!/usr/bin/env python3
"""bench.py -- Groth16 proving throughput on MI355X (BASELINE.json metric: constraints/sec, BN254,
2^20-constraint synthetic R1CS, at 1/2/4/8 GPUs).

    python bench.py --gpus N --steps K --warmup W [--curve bn254|bls12_381] [--log-m 20]
    python bench.py --curve bls12_381 --log-m 24 --shards-in-process 8 [--full-parity]   config 5's data path on one GPU
    python bench.py --gpus 8 --dry-run                                                   the plan of an N-GPU run, no GPU touched
    (N > 1 without WORLD_SIZE in the environment re-executes itself under
     python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A step = one complete Groth16 proof FROM THE MATRICES, the scope of the reference's
`create_proof_with_reduction_and_matrices` (groth16/examples/sha256.rs:159): R1CS x witness (dg16_qap:
a = A w, b = B w, c = a o b), h-polynomial (3 iNTT + 3 NTT of size m + pointwise), five MSMs (A, B1, L, H in G1,
B in G2; r, s != 0), A/B/C assembly -- with every input (CSR matrices, assignment, RESIDENT proving key =
window tables built once per key, outside the timed region, like any fixed CRS) already in HBM.
N > 1 is STRONG scaling of one proof (distributed-groth16_amd/parallel.py): each rank owns a 1/N slice of every
MSM and of the h-polynomial's transforms; the exchanges run inside libdg16 over RCCL.

One JSON line is printed by rank 0:
  roofline       the dominant kernel of the timed loop (G2 bucket accumulation, table mode), timed with HIP events
                 on the stream it runs on INSIDE the timed proofs: achieved = 160 B/point x points
                 per launch / duration, against the 8 TB/s HBM roof.  Small BY CONSTRUCTION (Pippenger does ~28 x W
                 field multiplications per 160 bytes): the binding roof is integer VALU, `valu_roofline`, priced
                 against BOTH the measured rate of this library's multiply and the v_mad_u64_u32 issue bound.
  cpu_baseline   the oracle ("port": arkworks-structured CPU restatement, NOT arkworks) proving THE TIMED INSTANCE on
                 this box's host cores (all cores, bounded to 32 OpenMP threads; plus a 1-thread run on a smaller
                 sample) -- its proof is compared with the GPU's proof of the timed loop (live parity gate).
  roofline_g1    the same for B1's G1 bucket accumulation (four launches of that kernel per proof: more total time than
                 the G2 launch), with valu_roofline_g1.
  host_pointer_step  the step with the assignment coming from host memory: one statement at a time, and as a
                 double-buffered queue (pipelined_ms_per_step).
  extras         ntt_2^22 (BASELINE config 3), plain / resident MSM points/s for G1 and G2 (config 2), the
                 sha256-shaped prove (config 4), key table bytes and build time.
"""

import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CURVE = "bn254"                    # default curve (kept as a module constant for tools/ that import bench)
FR_TOP = {"bn254": 0x30644E72E131A029, "bls12_381": 0x73EDA753299D7D48}   # top 64-bit limb of the scalar modulus
FR_MOD = {"bn254": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
          "bls12_381": 52435875175126190479447740508185965837690552500527637822603658699938581184513}
FQ_BYTES = {"bn254": 32, "bls12_381": 48}
SCALAR_BITS = {"bn254": 254, "bls12_381": 255}
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md
MAD_ISSUE_T = 34.4                 # T lane-op/s of v_mad_u64_u32, chip-wide (profiles/r1_ubench_instr_rate.txt)


def calibrate(device):
    """In-run calibration (distributed-groth16_amd/libdg16_calib.so, csrc/calib.hip): the chip-wide issue rate of
    v_mad_u64_u32 at 8 and at 2 waves per SIMD and the shader clock the chip holds under that load, measured in THIS
    process right before the timed loop (< 0.2 s).  `peak` of the valu rooflines is the 8-wave figure of this run;
    box_factor = that / MAD_ISSUE_T (the round-1 constant), so two boxes of the pool that differ in ms_per_step can be
    told apart from two builds that differ.  Returns None if the library is not built (older trees)."""
    import ctypes
    path = os.path.join(ROOT, "distributed-groth16_amd", "libdg16_calib.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.dg16_calib_mad_rate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    out = {}
    t0 = time.perf_counter()
    for wps in (8, 2):
        buf = (ctypes.c_double * 6)()
        rc = lib.dg16_calib_mad_rate(device, wps, 12288, buf)      # ~3 ms per run: the length of the kernels it prices
        if rc != 0:
            return None
        out[wps] = tuple(buf)
    cus = int(out[8][3])
    return {"mad_issue_T_lane_ops_per_s": out[8][0], "mad_issue_T_fastest_slowest_of_5": [out[8][4], out[8][5]],
            "mad_issue_T_at_2_waves_per_simd": out[2][0],
            "sclk_under_mad_load_mhz": out[8][1], "sclk_under_mad_load_mhz_at_2_waves_per_simd": out[2][1],
            "issue_bound_at_that_clock_T": 16.0 * 4 * cus * out[8][1] * 1e6 / 1e12,
            "kernel_ms": out[8][2], "box_factor": out[8][0] / MAD_ISSUE_T, "reference_T": MAD_ISSUE_T,
            "seconds": time.perf_counter() - t0,
            "source": "in-run: csrc/calib.hip (16 independent v_mad_u64_u32 x 12288 rounds per lane, 8 waves per SIMD on every "
                      "CU, a warm-up then the median of five ~3-ms runs; clock = s_memtime / s_memrealtime x 100 MHz over the "
                      "kernel; issue_bound = 16 lanes per SIMD-cycle x 4 SIMDs x CUs x that clock)"}


def valu_constants(curve):
    """(v_mad_u64_u32 per base-field product, measured chip-wide G products/s) of the reduced-radix product
    (csrc/fp29.h), from the committed micro-benchmark output -- measured constants live under profiles/, not in the
    library's ABI.  profiles/r3c_valu_constants.json = tools/ubench/fe_rate on an MI355X with the product as the
    library ships it (explicit v_mad_u64_u32 chains, csrc/fp29_asm_gen.h: 174.9 G/s for the 9-limb fields; r4a_*: the
    same benchmark in round 4's first GPU call, 177.1); r3_valu_constants.json = the same benchmark on the C++-compiled
    product it replaced (169.3).  An auxiliary figure of
    the line, NOT its peak -- the peak is the issue rate of the instruction itself (MAD_ISSUE_T), which no rewrite of the
    product can move."""
    for name in ("r4a_valu_constants.json", "r3c_valu_constants.json", "r3_valu_constants.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                d = json.load(f)[curve + "_fq"]
            return d["mads_per_product"], d["product_G_per_s"], "profiles/" + name
    # round-2 figure (profiles/r2_ubench_montmul29_rate.txt: E rows, best of the occupancy sweep); BN254 only
    return {"bn254": 162, "bls12_381": 392}[curve], {"bn254": 166.02, "bls12_381": 34.4e3 / 392}[curve], \
        "profiles/r2_ubench_montmul29_rate.txt (bls12_381: mad issue bound, unmeasured)"


def add_mads(curve):
    """v_mad_u64_u32 of one XYZZ mixed addition as the kernels execute it (csrc/ec29.h, fp29.h; N limbs: a product is
    2 N^2, a square N (N + 1) / 2 + N^2, a Montgomery reduction alone N^2).  G1: 8 products + 2 squares; G2 over Fq2:
    8 products of 3 N^2... = 3 base products each + 2 squares of 2.  For the 9-limb fields Y3 = R (Q - X3) - PPP Y1 is
    fused (mul_sub): one reduction less in G1, two in G2.  -> (G1, G2, one base-field product)"""
    n = {"bn254": 9, "bls12_381": 14}[curve]
    prod, sqr, red = 2 * n * n, n * (n + 1) // 2 + n * n, n * n
    fused = n <= 9
    g1 = 8 * prod + 2 * sqr - (red if fused else 0)
    g2 = 8 * 3 * prod + 2 * 2 * prod - (2 * red if fused else 0)
    return g1, g2, prod


def proof_products(curve, info, log_m, nc, nnz=3, world=1):
    """v_mad_u64_u32 lane-operations one proof must perform with the shipped algorithms: one XYZZ mixed addition per
    bucket entry (add_mads), six radix-2 transforms (one product per butterfly = log2(m) / 2 per element) + the three
    w_2m^i shifts + the pointwise a b - c, and the sparse R1CS x witness products."""
    bits = SCALAR_BITS[curve]
    win = lambda c: (bits + 1 + c - 1) // c       # noqa: E731
    m = 1 << log_m
    g1_add, g2_add, prod = add_mads(curve)
    g1 = float(g1_add) * (3 * info["n_ab"] * win(info["c_ab"]) + info["n_h"] * win(info["c_h"]))
    g2 = float(g2_add) * info["n_ab"] * win(info["c_ab"])
    pf = {"bn254": 162, "bls12_381": 162}[curve]       # scalar fields: 9 limbs for both curves
    ntt = pf * m * (6 * log_m / 2.0 + 3 + 1) / world      # `info` describes this rank's key shard: its share of the rest
    qap = pf * 2.0 * nnz * nc / world
    return {"g1_msm": g1, "g2_msm": g2, "ntt": ntt, "r1cs_x_witness": qap, "total": g1 + g2 + ntt + qap,
            "unit": "v_mad_u64_u32 lane-operations", "g1_add": g1_add, "g2_add": g2_add}


def rand_fr(n, dev, gen, curve=CURVE):
    """Uniform canonical scalars in [0, r_top * 2^192) (statistically uniform mod r for histograms)."""
    lo = torch.randint(-2**63, 2**63 - 1, (n, 3), dtype=torch.int64, device=dev, generator=gen)
    hi = torch.randint(0, FR_TOP[curve], (n, 1), dtype=torch.int64, device=dev, generator=gen)
    return torch.cat([lo, hi], dim=1).contiguous()


class Workload:
    """Synthetic proving key (like PackedProvingKeyShare::rand, groth16/src/proving_key.rs:112-155) and a synthetic
    R1CS instance: `nc` constraints with `nnz` nonzeros per row of A and of B, `nv` wires of which `ni` are instance
    variables, domain m = 2^log_m >= nc + ni.  Defaults: nc = m - 2, nv = m (the headline shape).  The prover's work
    does not depend on the R1CS being satisfied (qap.rs:44-91 sets c = a o b), so A, B are uniform sparse rows."""

    def __init__(self, ctx, dev, log_m, rank=0, world=1, seed=20, curve=CURVE, nv=None, nc=None, ni=2, nnz=3,
                 h_sharded=None, share=None):
        """share: another Workload of the same shape and seed -- its bases, matrices and assignment are reused (the N
        shard keys of one proving key held in ONE process: sharded_prove_in_process); only the key shard and this
        rank's a, b, c rows are new."""
        import dg16_amd  # noqa: F401
        self.ctx, self.dev, self.curve = ctx, dev, curve
        self.m = 1 << log_m
        self.log_m = log_m
        self.ni = ni
        self.nv = self.m if nv is None else nv
        self.nc = self.m - ni if nc is None else nc
        assert self.nc + ni <= self.m and ni <= self.nv
        m, nv, nc = self.m, self.nv, self.nc
        fqb = FQ_BYTES[curve]

        def bases(group, cnt, s):
            t = torch.empty(cnt * 2 * fqb * group, dtype=torch.uint8, device=dev)
            ctx.gen_bases_dev(curve, group, seed * 100 + s, cnt, t.data_ptr())
            return t

        if share is not None:
            assert (share.curve, share.m, share.nv, share.nc, share.ni) == (curve, m, nv, nc, ni)
            self.aq, self.b1q, self.b2q, self.hq, self.lq, self.fixed = (share.aq, share.b1q, share.b2q, share.hq,
                                                                         share.lq, share.fixed)
        else:
            self.aq, self.b1q, self.b2q = bases(1, nv, 1), bases(1, nv, 2), bases(2, nv, 3)
            self.hq, self.lq = bases(1, m, 4), bases(1, nv - ni, 5)
            f1, f2 = bases(1, 3, 6), bases(2, 2, 7)
            ctx.sync(0)          # the generators ran on the library's stream: finish before torch touches them
            self.fixed = torch.cat([f1, f2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # N > 1 (2 / 4 / 8 ranks): the h-polynomial is sharded -- this rank evaluates only its cyclic rows of a, b, c
        # and owns the h bases h_query[rank + world * j] (parallel.py); otherwise whole vectors, contiguous slices
        from dg16_amd.parallel import h_is_sharded
        self.rank, self.world = rank, world
        self.h_sharded = (world > 1 and h_is_sharded(m, world)) if h_sharded is None else bool(h_sharded)
        self.pk = ctx.pk_create(curve, nv, ni, m, self.aq.data_ptr(), self.b1q.data_ptr(), self.b2q.data_ptr(),
                                self.hq.data_ptr(), self.lq.data_ptr(), self.fixed.data_ptr(), device_ptrs=True,
                                shard=rank, n_shards=world, h_cyclic=self.h_sharded)
        self.pk_build_s = time.perf_counter() - t0
        if share is not None:
            self.row_ptr, self.a_col, self.b_col, self.a_val, self.b_val, self.w = (
                share.row_ptr, share.a_col, share.b_col, share.a_val, share.b_val, share.w)
        else:
            gen = torch.Generator(device=dev)
            gen.manual_seed(seed)
            # CSR matrices A, B (Montgomery coefficients) and the full assignment (canonical integers, w[0] = 1)
            self.row_ptr = (torch.arange(nc + 1, dtype=torch.int64, device=dev) * nnz).to(torch.int32)
            self.a_col = torch.randint(0, nv, (nc * nnz,), dtype=torch.int32, device=dev, generator=gen)
            self.b_col = torch.randint(0, nv, (nc * nnz,), dtype=torch.int32, device=dev, generator=gen)
            self.a_val = rand_fr(nc * nnz, dev, gen, curve)
            self.b_val = rand_fr(nc * nnz, dev, gen, curve)
            self.w = rand_fr(nv, dev, gen, curve)
            self.w[0] = 0
            self.w[0, 0] = 1
        self.a = torch.empty((m // world if self.h_sharded else m, 4), dtype=torch.int64, device=dev)
        self.b = torch.empty_like(self.a)
        self.c = torch.empty_like(self.a)
        # r, s: canonical, DENSE (a prover draws them uniformly, prove.rs / sha256.rs:159 `Fr::rand`; until round 5 this was
        # 3 + 2^64 + 4 * 2^128 + 2^192 and its like, whose four set bits made the double-and-add chains s A', r B1' of the
        # proof nearly addition-free -- half the work a real proof's chains do); below every curve's modulus
        self.rs = np.array([[0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB, 0x0123456789ABCDEF],
                            [0xD6E8FEB86659FD93, 0xCA5A826395121157, 0xC2B2AE3D27D4EB4F, 0x0FEDCBA987654321]],
                           dtype=np.uint64)
        torch.cuda.synchronize()   # torch's stream -> the library's stream
        self.qap()
        ctx.sync(0)

    def qap(self):
        """a, b, c <- R1CS x witness on the GPU (stream-ordered on channel 0, like the proof that follows)."""
        if self.h_sharded:
            self.ctx.qap_rows_dev(self.curve, self.nc, self.ni, self.nv, self.log_m, self.row_ptr.data_ptr(),
                                  self.a_col.data_ptr(), self.a_val.data_ptr(), self.row_ptr.data_ptr(),
                                  self.b_col.data_ptr(), self.b_val.data_ptr(), self.w.data_ptr(), self.rank,
                                  self.world, self.a.data_ptr(), self.b.data_ptr(), self.c.data_ptr(),
                                  scalars_mont=False)
            return
        self.ctx.qap_dev(self.curve, self.nc, self.ni, self.nv, self.log_m, self.row_ptr.data_ptr(),
                         self.a_col.data_ptr(), self.a_val.data_ptr(), self.row_ptr.data_ptr(), self.b_col.data_ptr(),
                         self.b_val.data_ptr(), self.w.data_ptr(), self.a.data_ptr(), self.b.data_ptr(),
                         self.c.data_ptr(), scalars_mont=False)

    def proof_bytes(self):
        return 12 * FQ_BYTES[self.curve]


def to_host_u64(t, cols):
    return t.view(torch.uint8).cpu().numpy().view(np.uint64).reshape(-1, cols)


def oracle_prove(wl, threads, r=None, s=None):
    """The oracle ("port") proves the instance of `wl` from the matrices on the host cores.
    Returns (A, B, C) affine arrays and the seconds spent in qap + h-polynomial + the five MSMs."""
    from oracle import corc
    curve, m, nv, ni, nc = wl.curve, wl.m, wl.nv, wl.ni, wl.nc
    l1, l2 = FQ_BYTES[curve] // 4, FQ_BYTES[curve] // 2     # u64 limbs of a G1 / G2 affine point
    aq, b1q, b2q = to_host_u64(wl.aq, l1), to_host_u64(wl.b1q, l1), to_host_u64(wl.b2q, l2)
    hq, lq = to_host_u64(wl.hq, l1), to_host_u64(wl.lq, l1)
    g1b = 3 * 2 * FQ_BYTES[curve]
    f1 = to_host_u64(wl.fixed[:g1b], l1)
    f2 = to_host_u64(wl.fixed[g1b:], l2)
    alpha, beta1, delta1 = f1[0:1], f1[1:2], f1[2:3]
    beta2, delta2 = f2[0:1], f2[1:2]
    w = to_host_u64(wl.w, 4)
    rp = wl.row_ptr.cpu().numpy().view(np.uint32)
    csr_a = (rp, wl.a_col.cpu().numpy().view(np.uint32), to_host_u64(wl.a_val, 4))
    csr_b = (rp, wl.b_col.cpu().numpy().view(np.uint32), to_host_u64(wl.b_val, 4))
    w_mont = corc.field_op(curve, "fr", "to_mont", w)       # arkworks holds the assignment in Montgomery form
    if r is None:
        r = int(sum(int(x) << (64 * i) for i, x in enumerate(wl.rs[0])))
        s = int(sum(int(x) << (64 * i) for i, x in enumerate(wl.rs[1])))
    t0 = time.perf_counter()
    a, b, c = corc.qap(curve, nc, ni, m, csr_a, csr_b, w_mont, threads=threads)
    h = corc.h_poly(curve, a, b, c, threads=threads)
    h_canon = corc.field_op(curve, "fr", "from_mont", h)
    msm = lambda g, bases, sc: corc.msm(curve, g, bases, sc, threads=threads)   # noqa: E731
    mA = msm(1, aq[1:], w[1:])
    mB1 = msm(1, b1q[1:], w[1:])
    mB2 = msm(2, b2q[1:], w[1:])
    mL = msm(1, lq, w[ni:])
    mH = msm(1, hq, h_canon)
    t_cpu = time.perf_counter() - t0
    add = lambda g, p, q: corc.point_add(curve, g, p, q)    # noqa: E731
    mul = lambda g, p, k: corc.point_mul(curve, g, p, k)    # noqa: E731
    R = FR_MOD[curve]
    A = add(1, add(1, mA, aq[0:1]), add(1, alpha, mul(1, delta1, r)))
    B1 = add(1, add(1, mB1, b1q[0:1]), add(1, beta1, mul(1, delta1, s)))
    B = add(2, add(2, mB2, b2q[0:1]), add(2, beta2, mul(2, delta2, s)))
    C = add(1, add(1, mL, mH), add(1, add(1, mul(1, A, s), mul(1, B1, r)), mul(1, delta1, (R - r * s % R) % R)))
    return (A, B, C), t_cpu


def gpu_proof_affine(curve, proof_u8):
    from oracle import corc
    nl = FQ_BYTES[curve] // 8
    gp = np.asarray(proof_u8).view(np.uint64)
    return (corc.jac_to_affine(curve, 1, gp[:3 * nl]), corc.jac_to_affine(curve, 2, gp[3 * nl:9 * nl]),
            corc.jac_to_affine(curve, 1, gp[9 * nl:]))


def prove_once(ctx, wl, rs=None):
    """qap + prove of `wl` on the GPU; returns the proof bytes (host)."""
    proof = torch.empty(wl.proof_bytes(), dtype=torch.uint8, device=wl.dev)
    wl.qap()
    ctx.prove_dev(wl.pk, wl.a.data_ptr(), wl.b.data_ptr(), wl.c.data_ptr(), wl.w.data_ptr(),
                  wl.rs if rs is None else rs, proof.data_ptr(), scalars_mont=False)
    for ch in range(3):
        ctx.sync(ch)
    return proof.cpu().numpy()



def _all_to_all_in_process(bufs):
    """bufs[r]: rank r's send buffer [peer][...] -> receive buffers [src][...] (what the all-to-all over xGMI moves)."""
    n = len(bufs)
    chunks = [b.view(n, -1) for b in bufs]
    return [torch.stack([chunks[src][dst] for src in range(n)]).contiguous().view(-1) for dst in range(n)]


class ShardedInProcess:
    """BASELINE config 5's data path with all N ranks played by ONE process on one GPU: N shard keys of one proving
    key (DG16_F_H_CYCLIC: 1/N of every MSM's bases as window tables, the h bases h_query[rank + N j]), cyclic QAP rows
    (dg16_qap_rows), the three stages of the sharded h-polynomial with the two all-to-alls as device transposes,
    dg16_groth16_msms_h per shard, the N records concatenated as the all-gather would, dg16_groth16_assemble.  Same
    entry points, same kernels, same shard sizes as one process per GPU -- only the wire is a device copy.  Bases,
    matrices and assignment are generated once and shared by the N shard Workloads.
    (local_groth_bench.rs:83-158 is the reference shape: FFTs + five MSMs on a BLS curve.)"""

    def __init__(self, ctx, dev, curve, log_m, world, seed=33):
        from dg16_amd.parallel import h_is_sharded
        assert h_is_sharded(1 << log_m, world), "power-of-two rank count and m >= N^2"
        self.ctx, self.dev, self.curve, self.log_m, self.world = ctx, dev, curve, log_m, world
        first = Workload(ctx, dev, log_m, 0, world, seed=seed, curve=curve)
        self.shards = [first] + [Workload(ctx, dev, log_m, k, world, seed=seed, curve=curve, share=first)
                                 for k in range(1, world)]
        self.M = (1 << log_m) // world
        assert all(wl.h_sharded and wl.pk.info()["n_h"] == self.M for wl in self.shards)
        i64 = dict(dtype=torch.int64, device=dev)
        self.send = [torch.empty(3 * self.M * 4, **i64) for _ in range(world)]
        self.h = [torch.empty(self.M * 4, **i64) for _ in range(world)]
        self.recs = [torch.empty(ctx.results_bytes(curve), dtype=torch.uint8, device=dev) for _ in range(world)]
        self.proof = torch.empty(first.proof_bytes(), dtype=torch.uint8, device=dev)

    def table_bytes(self):
        return sum(wl.pk.info()["table_bytes"] for wl in self.shards)

    def _sync(self):
        for ch in range(3):
            self.ctx.sync(ch)
        torch.cuda.synchronize()

    def prove(self):
        """One proof; returns (proof tensor, seconds per rank [list]): the time each rank's kernels took between the
        exchanges (qap_rows + stage 0, stage 1, stage 2 + the five partial MSMs), the exchanges themselves excluded."""
        c, curve, log_m, world = self.ctx, self.curve, self.log_m, self.world
        per_rank = [0.0] * world

        def timed(r, fn):
            self._sync()
            t0 = time.perf_counter()
            fn()
            self._sync()
            per_rank[r] += time.perf_counter() - t0

        for r, wl in enumerate(self.shards):
            def stage0(wl=wl, r=r):
                wl.qap()
                c.h_poly_dist_stage_dev(curve, log_m, r, world, 0, [wl.a.data_ptr(), wl.b.data_ptr(), wl.c.data_ptr()],
                                        self.send[r].data_ptr())
            timed(r, stage0)
        recv = _all_to_all_in_process(self.send)
        torch.cuda.synchronize()
        for r in range(world):
            timed(r, lambda r=r: c.h_poly_dist_stage_dev(curve, log_m, r, world, 1, [recv[r].data_ptr()],
                                                         self.send[r].data_ptr()))
        recv = _all_to_all_in_process(self.send)
        torch.cuda.synchronize()
        for r, wl in enumerate(self.shards):
            def stage2(wl=wl, r=r):
                c.h_poly_dist_stage_dev(curve, log_m, r, world, 2, [recv[r].data_ptr()], self.h[r].data_ptr())
                c.groth16_msms_h_dev(wl.pk, self.h[r].data_ptr(), wl.w.data_ptr(), wl.rs, self.recs[r].data_ptr(),
                                     scalars_mont=False)
            timed(r, stage2)
        gathered = torch.cat(self.recs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c.groth16_assemble_dev(self.shards[0].pk, gathered.data_ptr(), world, self.shards[0].rs, self.proof.data_ptr(),
                               scalars_mont=False)
        self._sync()
        assemble = time.perf_counter() - t0
        return self.proof, per_rank, assemble

    def close(self):
        for wl in self.shards:
            wl.pk.close()
        self.shards = []


def sharded_in_process_line(ctx, dev, curve, log_m, world, steps, parity=True):
    """`bench.py --shards-in-process N`: the sharded proof of a 2^log_m instance over N shard keys on one GPU, proved
    `steps` times; the oracle proves the same instance (parity gate).  The figure is the slowest rank's compute per
    proof (what an N-GPU node would run between its exchanges) -- NOT an N-GPU throughput: no byte crosses a link."""
    t0 = time.perf_counter()
    sp = ShardedInProcess(ctx, dev, curve, log_m, world)
    build_s = time.perf_counter() - t0
    sp.prove()
    worst, asm = [], []
    for _ in range(steps):
        proof, per_rank, assemble = sp.prove()
        worst.append(max(per_rank))
        asm.append(assemble)
    wl = sp.shards[0]
    res = {"mode": "sharded proof, %d shard keys in one process on one GPU (exchanges = device copies)" % world,
           "curve": curve, "log_domain": log_m, "shards": world,
           "slowest_rank_compute_ms": sum(worst) / len(worst) * 1e3, "assemble_ms": sum(asm) / len(asm) * 1e3,
           "all_ranks_compute_ms_last": [t * 1e3 for t in per_rank],
           "key_table_bytes_all_shards": sp.table_bytes(), "keys_build_s": build_s, "steps": steps,
           "exchange_bytes_per_rank": {"all_to_all_x2": 3 * 32 * sp.M, "all_gather_record": ctx.results_bytes(curve)}}
    if parity:
        (A, B, C), t_cpu = oracle_prove(wl, cpu_threads())
        gA, gB, gC = gpu_proof_affine(curve, proof.cpu().numpy())
        ok = bool(np.array_equal(A, gA) and np.array_equal(B, gB) and np.array_equal(C, gC))
        res["parity_check"] = "pass (the oracle proved the same 2^%d instance: %.1f s on %d threads)" % (
            log_m, t_cpu, cpu_threads()) if ok else "FAIL"
    sp.close()
    return res



def table_window_bits(n, bits=254):
    """csrc/msm_impl.h: msm_window_bits(n, table = true, scalar bits)."""
    lg = max(n, 1).bit_length() - 1
    if n > (3 << lg) // 2:
        lg += 1
    c = lg - 3
    if c < 16:
        c = min(lg + 1, 16)
        if lg >= 13:
            def top(w):
                return bits + 1 - ((bits + w) // w - 1) * w
            if top(15) >= top(16):
                c = 15 if (top(15) > top(16) or lg <= 16) else 16
    return max(4, min(20, c))


def dry_run_plan(curve, log_m, world, steps, warmup):
    """`bench.py --gpus N --dry-run`: what an N-GPU run of this command WILL do, without touching a GPU -- per-rank key
    shard (points per MSM, window bits, table bytes: the formulas of csrc/prover_impl.h), the bytes each exchange
    moves, and the exact command lines for the headline and for BASELINE config 5.  For the first run on a multi-GPU
    node: nothing here has ever crossed a link."""
    from dg16_amd.parallel import shard_bounds, h_is_sharded
    m = 1 << log_m
    nv, ni = m, 2
    fq = FQ_BYTES[curve]
    p1, p2 = 2 * fq, 4 * fq
    bits = SCALAR_BITS[curve]
    ranks = []
    for r in range(world):
        lo, hi = shard_bounds(nv - 1, r, world)
        n_ab = hi - lo
        n_h = m // world
        c_ab, c_h = table_window_bits(n_ab + 3, bits), table_window_bits(n_h, bits)
        w_ab, w_h = (bits + c_ab) // c_ab, (bits + c_h) // c_h
        ranks.append({"rank": r, "n_ab": n_ab + 3, "n_h": n_h, "window_bits": {"ab": c_ab, "h": c_h},
                      "windows": {"ab": w_ab, "h": w_h},
                      "table_bytes": w_ab * (n_ab + 3) * (3 * p1 + p2) + w_h * n_h * p1})
    sharded = world > 1 and h_is_sharded(m, world)
    rec = (6 * 3 * fq + 3 * 2 * fq)           # A', B1', L, H, s A', r B1' (G1 Jacobian) + B' (G2 Jacobian)
    exch = {"h_polynomial": ("two all-to-alls of %d bytes per rank (3 vectors x 32 B x m / N), %d bytes to each peer"
                             % (3 * 32 * m // world, 3 * 32 * m // world // world)) if sharded else
            "replicated on every rank (N not in {2, 4, 8} or m < N^2): no exchange",
            "results": "one all-gather of %d-byte records (%d bytes received per rank)" % (rec, rec * world)} if world > 1 \
        else {}
    me = os.path.basename(__file__)
    launch = ("python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port 29500 "
              % world) if world > 1 else "python "
    return {"dry_run": True, "curve": curve, "log_domain": log_m, "n_gpus": world, "steps": steps, "warmup": warmup,
            "h_polynomial_sharded": sharded, "ranks": ranks,
            "table_bytes_per_rank_max": max(r["table_bytes"] for r in ranks),
            "hbm_per_gpu_bytes": 288 * 10**9, "exchanges_per_proof": exch,
            "transport": "native RCCL communicator of libdg16 (dlopen of librccl); every rank falls back to "
                         "torch.distributed together if it cannot be bound or a rank cannot join; config.rccl_ranks on "
                         "the line = what the communicator itself reports, asserted == n_gpus",
            "commands": {
                "this_run": "%s%s --gpus %d --steps %d --warmup %d --curve %s --log-m %d" % (
                    launch, me, world, steps, warmup, curve, log_m),
                "headline_bn254_2e20": "python %s --gpus %d --steps 20 --warmup 3" % (me, world),
                "config5_bls12_381_2e24": "python %s --gpus 8 --curve bls12_381 --log-m 24 --steps 3 --warmup 1 "
                                          "--no-replicas [--full-parity]" % me,
                "config5_data_path_on_one_gpu": "python %s --curve bls12_381 --log-m 24 --shards-in-process 8 --steps 2 "
                                                "--full-parity" % me}}


def cpu_threads():
    # >32 OpenMP threads only adds fork/join overhead here (measured on the EPYC 9575F GPU box:
    # NTT 2^16 takes 5 ms at 32 threads and 2 s at 256)
    return min(os.cpu_count() or 1, 32)


def cpu_baseline_and_parity(ctx, dev, log_s, curve=CURVE, wl=None, gpu_proof=None, **shape):
    """Oracle ("port") prove + comparison with the GPU proof of the same instance.  With `wl` / `gpu_proof` given
    the instance is the caller's (bench: the timed one); otherwise a fresh 2^log_s instance (seed 7) is proved
    on both sides (smoke, tests)."""
    own = wl is None
    if own:
        wl = Workload(ctx, dev, log_s, 0, 1, seed=7, curve=curve, **shape)
    if gpu_proof is None:
        gpu_proof = prove_once(ctx, wl)
    threads = cpu_threads()
    (A, B, C), t_cpu = oracle_prove(wl, threads)
    gA, gB, gC = gpu_proof_affine(curve, gpu_proof)
    ok = bool(np.array_equal(A, gA) and np.array_equal(B, gB) and np.array_equal(C, gC))
    res = {"value": wl.nc / t_cpu, "unit": "constraints/s", "cores": threads, "kind": "port",
           "sample": "one proof of %s (R1CS x witness + h-poly + 5 MSMs; C oracle, OpenMP, Pippenger parallel over "
                     "<=19 windows like arkworks): %.2f s" % ("the timed instance" if not own else
                                                              "a 2^%d instance of the same workload" % wl.log_m, t_cpu)}
    if own:
        wl.pk.close()
    return res, ok


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def time_call(ctx, fn, channel, reps=3):
    """Median (whole call ms, dominant-kernel ms) by the library's HIP events on the channel's stream."""
    out = []
    for _ in range(reps + 1):
        fn()
        ctx.sync(channel)
        out.append((ctx.last_kernel_ms(channel, 0), ctx.last_kernel_ms(channel, 1)))
    out = sorted(out[1:])
    return out[len(out) // 2]


def extras(ctx, dev, wl, curve, res):
    """Config 2 / 3 / 4 lines and key disclosure (N = 1, rank 0 only)."""
    fqb = FQ_BYTES[curve]
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    # BASELINE config 3: Fr radix-2 NTT, domain 2^22
    x = rand_fr(1 << 22, dev, gen, curve)
    torch.cuda.synchronize()
    call_ms, k_ms = time_call(ctx, lambda: ctx.ntt_dev(curve, x.data_ptr(), 22), 0)
    res["ntt_2^22"] = {"ms": k_ms, "algorithmic_GBps": 64.0 * (1 << 22) / (k_ms * 1e-3) / 1e9,
                       "hbm_frac": 64.0 * (1 << 22) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "field": curve + " Fr"}
    del x
    # BASELINE config 2: plain MSM over fresh bases (dg16_msm: sort + accumulate + reductions + Horner tail) and the
    # resident form (dg16_bases_upload once, then dg16_msm_resident: window tables, no tail)
    # Every timed call is checked: the result of the LAST timed call against the oracle's MSM of the same inputs, and
    # the oracle ("port": C, OpenMP, Pippenger parallel over windows like arkworks) is timed beside it.
    from oracle import corc
    n = wl.nv - 1
    msm = {}
    w_host = to_host_u64(wl.w, 4)[1:]
    msm_ok = True
    for g, bases in ((1, wl.aq), (2, wl.b2q)):
        pb = 2 * fqb * g
        bases_host = to_host_u64(bases, fqb // 8 * 2 * g)[1:]
        t0 = time.perf_counter()
        want = corc.msm(curve, g, bases_host, w_host, threads=cpu_threads())
        t_cpu = time.perf_counter() - t0
        msm["g%d_cpu_port_pts_per_s" % g] = n / t_cpu
        msm["g%d_cpu_port_cores" % g] = cpu_threads()
        out = torch.empty(3 * fqb * g, dtype=torch.uint8, device=dev)
        call_ms, acc_ms = time_call(ctx, lambda: ctx.msm_dev(curve, g, bases.data_ptr() + pb, wl.w.data_ptr() + 32, n,
                                                              out.data_ptr(), channel=1,
                                                              in_subgroup=True), 1)
        msm["g%d_plain_pts_per_s" % g] = n / (call_ms * 1e-3)
        msm["g%d_plain_ms" % g] = call_ms
        ok = bool(np.array_equal(corc.jac_to_affine(curve, g, out.cpu().numpy().view(np.uint64)), want))
        if hasattr(ctx, "bases_upload"):
            hb = ctx.bases_upload(curve, g, bases.data_ptr() + pb, n, device_ptrs=True)
            out.zero_()
            torch.cuda.synchronize()
            call_ms, acc_ms = time_call(ctx, lambda: ctx.msm_resident_dev(hb, wl.w.data_ptr() + 32, n, out.data_ptr(),
                                                                          channel=1), 1)
            msm["g%d_resident_pts_per_s" % g] = n / (call_ms * 1e-3)
            msm["g%d_resident_ms" % g] = call_ms
            ok = ok and bool(np.array_equal(corc.jac_to_affine(curve, g, out.cpu().numpy().view(np.uint64)), want))
            hb.close()
        msm["g%d_parity" % g] = "pass (the timed calls' results == the oracle's MSM)" if ok else "FAIL"
        msm_ok = msm_ok and ok
    msm["n"] = n
    res["msm_pts_per_s"] = msm
    res["msm_cpu_port"] = msm_cpu_port(curve, wl, w_host)
    res["msm_sweep"] = msm_sweep(ctx, dev)
    try:
        res["dmsm_sweep"] = dmsm_sweep(dev)
    except Exception as e:      # noqa: BLE001 -- a secondary figure must not take the line with it
        res["dmsm_sweep"] = {"error": repr(e)}
    if not msm_ok or "FAIL" in json.dumps(res["msm_sweep"]) or "FAIL" in json.dumps(res["dmsm_sweep"]):
        print(json.dumps(res))
        raise SystemExit("a timed MSM differs from the oracle's")
    # A proof WITHOUT window tables (dg16_ctx_set_table_budget below one row: the key keeps the plain bases, every MSM is
    # the fresh-bases Pippenger with its bucket sets per window and its Horner tail) -- what the resident tables buy,
    # and what a key that does not fit HBM costs; same instance, must be the same proof
    user_budget = getattr(ctx, "table_budget", 0)
    try:
        ctx.set_table_budget(1)
        cold = Workload(ctx, dev, wl.log_m, 0, 1, curve=curve, share=wl, h_sharded=False)
        ctx.set_table_budget(user_budget)
        want = gpu_proof_affine(curve, prove_once(ctx, wl))
        got = prove_once(ctx, cold)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            got = prove_once(ctx, cold)
            ts.append(time.perf_counter() - t0)
        ok = all(np.array_equal(x, y) for x, y in zip(gpu_proof_affine(curve, got), want))
        cinfo = cold.pk.info()
        res["tableless_proof"] = {"ms_per_proof": min(ts) * 1e3, "constraints_per_s": wl.nc / min(ts),
                                  "key_bytes": cinfo["table_bytes"], "key_table_stride": cinfo["table_stride"],
                                  "key_build_s": cold.pk_build_s,
                                  "parity_check": "pass (== the proof of the resident-table key)" if ok else "FAIL",
                                  "note": "one row per table = the plain bases (table budget 1 byte); each step includes "
                                          "the device-to-host copy of the proof"}
        cold.pk.close()
        del cold
        if not ok:
            raise SystemExit("the table-less proof differs from the resident-table proof")
    finally:
        ctx.set_table_budget(user_budget)
    # BASELINE config 4: sha256-shaped prove (29 823 wires, 2 instance variables, domain 2^15; the real r1cs is a
    # missing blob of the reference tree: SURVEY.md section 0), parity against the oracle, r = s = 0 and random
    if curve == "bn254":
        sha = Workload(ctx, dev, 15, 0, 1, seed=4, curve=curve, nv=29823, nc=29400, ni=2)
        ok = True
        for rs in (np.zeros((2, 4), dtype=np.uint64), sha.rs):
            gp = prove_once(ctx, sha, rs)
            r = int(sum(int(v) << (64 * i) for i, v in enumerate(rs[0])))
            s = int(sum(int(v) << (64 * i) for i, v in enumerate(rs[1])))
            (A, B, C), t_cpu = oracle_prove(sha, cpu_threads(), r, s)
            gA, gB, gC = gpu_proof_affine(curve, gp)
            ok = ok and bool(np.array_equal(A, gA) and np.array_equal(B, gB) and np.array_equal(C, gC))
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            prove_once(ctx, sha)
            ts.append(time.perf_counter() - t0)
        best = min(ts[1:])
        res["config4_sha256_shaped"] = {"ms_per_proof": best * 1e3, "constraints_per_s": sha.nc / best,
                                        "cpu_port_ms": t_cpu * 1e3, "parity_check": "pass" if ok else "FAIL",
                                        "shape": "29823 wires, 29400 constraints, 2 instance variables, domain 2^15, "
                                                 "3 nonzeros per row; r = s = 0 and r, s != 0 both checked"}
        sha.pk.close()
        if not ok:
            raise SystemExit("sha256-shaped proof differs from the oracle's")
        # BASELINE config 5's curve on one GPU at the headline size: BLS12-381, 2^20 - 2 constraints, timed, with the
        # parity gate on THE TIMED instance (the 2^24 size of config 5 is `--curve bls12_381 --log-m 24`: 126 GB of
        # window tables on one GPU, its 8-GPU form is the driver's to run)
        res["config5_bls12_381_2e20"] = timed_prove_with_parity(ctx, dev, "bls12_381", 20, steps=5)


def msm_cpu_port(curve, wl, w_host):
    """BASELINE config 1 (G1 MSM of 2^16 random points and scalars, single-process CPU: dmsm_test.rs:49-50 calls the
    plain G::msm next to d_msm) and the same at 2^20, on the oracle's MSM ("port"), one thread and all the threads
    the port uses -- the CPU figures the GPU pts/s of this line stand beside."""
    from oracle import corc
    fqb = FQ_BYTES[curve]
    bases = to_host_u64(wl.aq, fqb // 8 * 2)[1:]
    out = {"kind": "port", "group": "%s G1" % curve, "cpu_model": cpu_model()}
    for log_n in (16, 20):
        n = min(1 << log_n, bases.shape[0])
        for threads in (1, cpu_threads()):
            t0 = time.perf_counter()
            corc.msm(curve, 1, bases[:n], w_host[:n], threads=threads)
            dt = time.perf_counter() - t0
            out["2^%d_%d_threads" % (log_n, threads)] = {"n": n, "seconds": dt, "pts_per_s": n / dt}
    return out


def msm_sweep(ctx, dev):
    """dist-primitives/examples/msm_bench.rs:24-29: G::msm of 2^10 .. 2^19 random BLS12-377 G1 points and scalars (the
    reference times nothing itself: it is run under `time`).  Here: dg16_msm on fresh device-resident bases (whole call
    by the library's HIP events, median of 3), the oracle's MSM on the host cores beside it, and the GPU result
    compared with the oracle's at every size."""
    from oracle import corc
    curve, fqb = "bls12_377", 48
    nmax = 1 << 19
    gen = torch.Generator(device=dev)
    gen.manual_seed(1019)
    lo = torch.randint(-2**63, 2**63 - 1, (nmax, 3), dtype=torch.int64, device=dev, generator=gen)
    hi = torch.randint(0, 0x12AB655E9A2CA556, (nmax, 1), dtype=torch.int64, device=dev, generator=gen)  # < top limb of r
    scal = torch.cat([lo, hi], dim=1).contiguous()
    bases = torch.empty(nmax * 2 * fqb, dtype=torch.uint8, device=dev)
    ctx.gen_bases_dev(curve, 1, 377, nmax, bases.data_ptr())
    ctx.sync(0)
    torch.cuda.synchronize()
    bh, sh = to_host_u64(bases, 12), to_host_u64(scal, 4)
    out = torch.empty(3 * fqb, dtype=torch.uint8, device=dev)
    rows = []
    for log_n in range(10, 20):
        n = 1 << log_n
        call_ms, _ = time_call(ctx, lambda: ctx.msm_dev(curve, 1, bases.data_ptr(), scal.data_ptr(), n, out.data_ptr(),
                                                        channel=1, in_subgroup=True), 1)
        t0 = time.perf_counter()
        want = corc.msm(curve, 1, bh[:n], sh[:n], threads=cpu_threads())
        t_cpu = time.perf_counter() - t0
        ok = bool(np.array_equal(corc.jac_to_affine(curve, 1, out.cpu().numpy().view(np.uint64)), want))
        rows.append({"log_n": log_n, "gpu_ms": call_ms, "gpu_pts_per_s": n / (call_ms * 1e-3), "cpu_port_ms": t_cpu * 1e3,
                     "cpu_port_pts_per_s": n / t_cpu, "parity": "pass" if ok else "FAIL"})
    return {"curve": "bls12_377 G1 (msm_bench.rs)", "cpu_port_cores": cpu_threads(), "rows": rows}


def dmsm_sweep(dev, budget_s=25.0):
    """dist-primitives/examples/dmsm_bench.rs:40-53: d_msm over BLS12-377 G1 for domains 2^10 .. 2^19, packed sharing
    with l = 2 (PackedSharingParams::new(2): n = 8 parties -- the example's local testnet of 4 cannot feed an 8-share
    unpack, dmsm_test.rs:62-77 runs the same call with 8), every party's bases and scalars `dom.size()` long.  Here: the
    8 parties are host threads with a context each on THIS GPU (dist.LocalTestNet = mpc-net's LocalTestNet), shares in
    host memory as in the example; a row = wall time of one network round (all parties' local MSMs, gather to the king,
    unpack / pack in the exponent, scatter), best of 3.  The shares are PACKINGS of 2 x dom.size() clear points and
    scalars, so every round is checked: each party's result == the oracle's clear MSM (dmsm/mod.rs:147-193).  Sizes stop
    early when the sweep has used `budget_s` seconds (the default bench run must stay within minutes)."""
    import dg16_amd
    from dg16_amd import dist as D
    from oracle import corc
    curve, l = "bls12_377", 2
    n = 4 * l
    ctxs = [dg16_amd.Context(dev.index or 0) for _ in range(n)]
    pps = [D.PackedSharingParams(c, curve, l) for c in ctxs]
    net = D.LocalTestNet(n)
    rows, t_start = [], time.perf_counter()
    try:
        for log_d in range(10, 20):
            if time.perf_counter() - t_start > budget_s:
                break
            d = 1 << log_d
            M = d * l
            pts = ctxs[0].gen_bases(curve, 1, 1000 + log_d, M)
            sc = corc.rand_field(curve, "fr", 2000 + log_d, M, mont=True)
            t0 = time.perf_counter()
            clear = corc.msm(curve, 1, pts, sc, scalars_mont=True, threads=cpu_threads())
            t_cpu = time.perf_counter() - t0
            pb = pps[0].packexp_from_public(1, pts.reshape(d, l, -1))          # [d][n][..]
            ps = pps[0].pack_from_public(sc.reshape(d, l, 4))                  # [d][n][4]
            shares = [(np.ascontiguousarray(pb[:, i]), np.ascontiguousarray(ps[:, i])) for i in range(n)]
            del pb, ps
            best, got = None, None
            for rep in range(4):       # (the first round of a size is a warm-up: workspaces of eight fresh contexts grow in it)
                t0 = time.perf_counter()
                got = net.simulate_network_round(
                    lambda i, h: D.d_msm(ctxs[i], pps[i], h, 1, shares[i][0], shares[i][1], in_subgroup=True))
                dt = time.perf_counter() - t0
                if rep:
                    best = dt if best is None else min(best, dt)
            ok = all(np.array_equal(corc.jac_to_affine(curve, 1, g), clear) for g in got)
            rows.append({"log_domain": log_d, "points_per_party": d, "round_ms": best * 1e3,
                         "party_pts_per_s": n * d / best, "clear_msm_cpu_port_ms": t_cpu * 1e3,
                         "parity": "pass" if ok else "FAIL"})
            if not ok:
                break
    finally:
        net.close()
        for p_ in pps:
            p_.close()
        for c in ctxs:
            c.close()
    return {"shape": "d_msm, BLS12-377 G1, l = 2, 8 parties as host threads on one GPU (dmsm_bench.rs:40-53), host-memory "
                     "shares; round_ms = one simulate_network_round, best of 3 after a warm-up round",
            "cpu_port_cores": cpu_threads(), "rows": rows}


def host_pointer_figure(ctx, wl, prover, steps):
    """The reference entry point takes host memory (create_proof_with_reduction_and_matrices: matrices and key fixed per
    circuit, the full assignment new per proof, groth16/examples/sha256.rs:159).  Two secondary figures:
      ms_per_step            one statement at a time: H2D copy of the assignment (pinned source), R1CS x witness, proof,
                             D2H copy of the proof, a host synchronisation after every step;
      pipelined_ms_per_step  a QUEUE of statements (what mpc-api/src/main.rs:393 serves, one job after another): two
                             device buffers for the assignment -- the H2D copy of statement k + 1 runs on a copy stream
                             under the proof of statement k, ordered by events, no host synchronisation inside the
                             queue; the proofs' D2H copies ride on the same copy stream behind each proof."""
    w_host = wl.w.cpu().pin_memory()
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps + 1):
        t0 = time.perf_counter()
        wl.w.copy_(w_host, non_blocking=True)
        torch.cuda.synchronize()                 # torch's stream -> the library's stream
        wl.qap()
        proof = prover.prove(wl.a, wl.b, wl.c, wl.w, wl.rs, scalars_mont=False)
        for ch in range(3):
            ctx.sync(ch)
        proof.cpu()
        ts.append(time.perf_counter() - t0)
    dt = sum(ts[1:]) / steps
    out = {"ms_per_step": dt * 1e3, "constraints_per_s": wl.nc / dt, "steps": steps,
           "includes": "H2D copy of the %d-byte full assignment from pinned host memory, R1CS x witness, the proof, D2H "
                       "copy of the %d-byte proof; matrices and proving key resident (fixed per circuit)"
                       % (w_host.numel() * 8, wl.proof_bytes())}
    try:
        out.update(host_pointer_pipelined(ctx, wl, prover, steps, w_host, proof.cpu()))
    except Exception as e:      # noqa: BLE001 -- a secondary figure must not take the line with it
        out["pipelined_error"] = repr(e)
    return out


def host_pointer_pipelined(ctx, wl, prover, steps, w_host, want_proof_host):
    """The double-buffered queue behind host_pointer_step.pipelined_ms_per_step.  The library's three channels run on
    torch streams for the duration (dg16_set_stream), so that torch events order the copy stream against them."""
    dev = wl.dev
    chans = [torch.cuda.Stream(device=dev) for _ in range(3)]
    copy = torch.cuda.Stream(device=dev)
    bufs = [wl.w, torch.empty_like(wl.w)]
    keep_w = wl.w
    copied = [torch.cuda.Event() for _ in range(2)]
    freed = [[torch.cuda.Event() for _ in range(3)] for _ in range(2)]     # per buffer: last proof done on channel c
    for ch in range(3):
        ctx.sync(ch)
    torch.cuda.synchronize()
    for ch in range(3):
        ctx.set_stream(ch, chans[ch].cuda_stream)
    try:
        def run(n):
            host_proofs = [torch.empty(wl.proof_bytes(), dtype=torch.uint8).pin_memory() for _ in range(n)]
            used = [False, False]
            with torch.cuda.stream(copy):
                bufs[0].copy_(w_host, non_blocking=True)
                copied[0].record(copy)
            t0 = time.perf_counter()
            for k in range(n):
                cur, oth = k % 2, (k + 1) % 2
                if k + 1 < n:
                    with torch.cuda.stream(copy):
                        if used[oth]:
                            for e in freed[oth]:
                                copy.wait_event(e)           # the proof that last read this buffer is complete
                        bufs[oth].copy_(w_host, non_blocking=True)
                        copied[oth].record(copy)
                for st in chans:
                    st.wait_event(copied[cur])
                wl.w = bufs[cur]
                wl.qap()
                proof = prover.prove(wl.a, wl.b, wl.c, wl.w, wl.rs, scalars_mont=False)
                for c_, st in enumerate(chans):
                    freed[cur][c_].record(st)
                used[cur] = True
                with torch.cuda.stream(copy):
                    for e in freed[cur]:
                        copy.wait_event(e)
                    host_proofs[k].copy_(proof, non_blocking=True)
            copy.synchronize()
            for st in chans:
                st.synchronize()
            return (time.perf_counter() - t0) / n, host_proofs

        run(2)
        dt, proofs = run(steps)
        same = all(np.array_equal(x, y) for x, y in zip(gpu_proof_affine(wl.curve, proofs[-1].numpy()),
                                                       gpu_proof_affine(wl.curve, want_proof_host.numpy())))
        return {"pipelined_ms_per_step": dt * 1e3, "pipelined_constraints_per_s": wl.nc / dt,
                "pipelined_last_proof_equals_synchronous": bool(same),
                "pipelined": "queue of %d statements, assignment double-buffered: H2D of statement k + 1 on a copy stream "
                             "under the proof of statement k (events, no host synchronisation inside the queue)" % steps}
    finally:
        for st in chans:
            st.synchronize()
        for ch in range(3):
            ctx.set_stream(ch, None)
        wl.w = keep_w


def pmc_traffic(curve):
    """(G2 traffic, G1 traffic, source) per launch from the committed summary of the separate rocprofv3 --pmc passes over
    2^20 proofs of `curve` (tools/evidence_run.sh, tools/pmc_json.py); (None, None, None) if there is none."""
    for name in {"bls12_381": ("r6_pmc_bls12_381_accumulate.json",)}.get(curve, ()):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                pj = json.load(f)
            return (pj["traffic_bytes_per_launch"], pj.get("g1_traffic_bytes_per_launch"),
                    "profiles/%s (%s)" % (name, pj.get("correction", "")))
    return None, None, None


def timed_prove_with_parity(ctx, dev, curve, log_m, steps):
    wl5 = Workload(ctx, dev, log_m, 0, 1, seed=21, curve=curve)
    prove_once(ctx, wl5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gp = prove_once(ctx, wl5)
    dt = (time.perf_counter() - t0) / steps
    g2_ms, g1_ms = ctx.last_kernel_ms(2, 1), ctx.last_kernel_ms(1, 1)      # the accumulations of the last timed proof
    (A, B, C), t_cpu = oracle_prove(wl5, cpu_threads())
    gA, gB, gC = gpu_proof_affine(curve, gp)
    ok = bool(np.array_equal(A, gA) and np.array_equal(B, gB) and np.array_equal(C, gC))
    info = wl5.pk.info()
    out = {"ms_per_proof": dt * 1e3, "constraints_per_s": wl5.nc / dt, "steps": steps, "cpu_port_s": t_cpu,
           "cpu_port_cores": cpu_threads(), "parity_check": "pass (the timed instance)" if ok else "FAIL",
           "key_table_bytes": info["table_bytes"], "key_table_build_s": wl5.pk_build_s,
           "workload": "%s Groth16 prove from the matrices, 2^%d - 2 constraints, 2^%d wires, r, s != 0; each step "
                       "includes the device-to-host copy of the proof" % (curve.upper(), log_m, log_m)}
    if curve == "bls12_381":
        # the dominant kernels of THIS curve against both roofs (the 14-limb forms: 10 584 / 3 542 v_mad_u64_u32 per G2 / G1
        # mixed addition, 224 / 128 algorithmic bytes per point), traffic from the committed PMC passes when log_m = 20
        g1_mads, g2_mads, _ = add_mads(curve)
        n_pts = info["n_ab"]
        nwin = (SCALAR_BITS[curve] + 1 + info["c_ab"] - 1) // info["c_ab"]
        tr2, tr1, src = pmc_traffic(curve) if log_m == 20 else (None, None, None)
        for key, ms, alg, mads, tr, kern in (("roofline", g2_ms, 224.0, g2_mads, tr2, "msm_accumulate_steps_kernel<Fp2<bls12_381_fq>>"),
                                             ("roofline_g1", g1_ms, 128.0, g1_mads, tr1, "msm_accumulate_kernel<Fp<bls12_381_fq>>")):
            if not ms:
                continue
            ach = alg * n_pts / (ms * 1e-3) / 1e9
            out[key] = {"bound": "hbm", "kernel": kern, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": tr, "traffic_source": src, "kernel_ms": ms,
                        "points_per_launch": n_pts,
                        "mad_T_lane_ops_per_s": float(mads) * n_pts * nwin / (ms * 1e-3) / 1e12}
    wl5.pk.close()
    del wl5
    torch.cuda.empty_cache()
    if not ok:
        raise SystemExit("%s 2^%d proof differs from the oracle's" % (curve, log_m))
    return out


def self_launch(n):
    """Re-executes this command line under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1,
    a free port) and returns its exit status.  The children see WORLD_SIZE and take the normal path."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this driver
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--curve", default=CURVE, choices=sorted(FR_TOP))
    ap.add_argument("--log-m", type=int, default=20)
    ap.add_argument("--cpu-sample-log", type=int, default=20,
                    help="log2 size of the CPU baseline / parity instance when the timed one is larger")
    ap.add_argument("--full-parity", action="store_true",
                    help="prove the TIMED instance with the oracle whatever its size (2^24: minutes of host time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--table-budget-gb", type=float, default=0.0,
                    help="HBM budget of the key's window tables (dg16_ctx_set_table_budget); 0 = one row per window")
    ap.add_argument("--no-replicas", action="store_true",
                    help="N > 1: skip the secondary figure (N independent proofs, one whole key per GPU)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N = 1: do not pass DG16_F_OVERLAP_TAIL (each proof's last bucket reduction and assembly then "
                         "finish on channel 0 before the next proof's first kernel, as in rounds 1-3)")
    ap.add_argument("--dist-overlap", action="store_true",
                    help="N > 1: pass DG16_F_OVERLAP_TAIL to dg16_groth16_prove_dist (the queue of sharded proofs overlaps each "
                         "proof's last reduction, all-gather and assembly with the next proof's first stage)")
    ap.add_argument("--shards-in-process", type=int, default=0, metavar="N",
                    help="N = 1 only: prove the instance over N shard keys (2, 4, 8) held by THIS process -- config 5's "
                         "data path (cyclic h shards, dg16_qap_rows, three h stages, N records, assembly) with device "
                         "copies for the exchanges -- against the oracle's proof of the same instance, and print that line")
    ap.add_argument("--dry-run", action="store_true",
                    help="print the plan of this command (per-rank key shard and table bytes, exchange sizes, command "
                         "lines) as JSON and exit -- no GPU is touched")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "torch", "python"],
                    help="N > 1: native RCCL communicator of libdg16 (default), torch.distributed under the native "
                         "pipeline, or the Python-driven protocol")
    args = ap.parse_args()
    curve = args.curve
    if args.full_parity:
        args.cpu_sample_log = max(args.cpu_sample_log, args.log_m)

    if args.dry_run:
        print(json.dumps(dry_run_plan(curve, args.log_m, args.gpus, args.steps, args.warmup), indent=1))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one process per GPU) and relay their output
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libdg16 has no CPU path")
    # DG16_BENCH_SINGLE_DEVICE=1: every rank on cuda:0 with a gloo process group -- the N > 1 flow of this file on a
    # one-GPU box (RCCL refuses two ranks on one device; use --transport torch or python with it).  A test hook: such a
    # run measures nothing.
    single_dev = os.environ.get("DG16_BENCH_SINGLE_DEVICE") == "1"
    if single_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single_dev:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    import dg16_amd
    from dg16_amd.parallel import make_prover

    ctx = dg16_amd.Context(local_rank)
    if args.table_budget_gb:
        ctx.set_table_budget(int(args.table_budget_gb * 1e9))
    if args.shards_in_process:
        if world != 1:
            raise SystemExit("--shards-in-process plays all ranks in one process: use it with --gpus 1")
        res = sharded_in_process_line(ctx, dev, curve, args.log_m, args.shards_in_process, args.steps,
                                      parity=not args.no_cpu_baseline and
                                      (args.log_m <= max(args.cpu_sample_log, 22) or args.full_parity))
        print(json.dumps(res))
        if res.get("parity_check", "pass").startswith("FAIL"):
            raise SystemExit("sharded in-process proof differs from the oracle proof")
        return
    wl = Workload(ctx, dev, args.log_m, rank, world, curve=curve)
    # N > 1, --transport rccl: if librccl cannot be bound or a rank cannot join, ALL ranks fall back to
    # torch.distributed together (make_prover decides collectively); config.parallelism says which transport ran
    prover = make_prover(ctx, wl.pk, curve, dist, rank, world, transport=args.transport)

    # N = 1: the timed region is a QUEUE of proofs on one context, so each is issued with DG16_F_OVERLAP_TAIL -- its
    # last (H) bucket reduction, assembly and copy-out run on channel 2's stream under the next proof's R1CS x witness
    # and h-polynomial.  `single_proof_ms` of the line is the same call followed by a synchronisation every time.
    # N > 1: the same flag on dg16_groth16_prove_dist (round 6: H's reduction, the all-gather and the assembly under the next
    # proof's first stage) -- only with --dist-overlap: the all-gather then shares a communicator with the all-to-alls from
    # another stream, which no run between two devices has exercised yet (per-rank work on one GPU, loopback comm: 3.0 ->
    # 2.6 ms at 8 shards, tools/shard_timing.py with DG16_OVERLAP=1)
    overlap = (world == 1 or args.dist_overlap) and not args.no_overlap and hasattr(prover, "overlap_tail")
    if overlap:
        prover.overlap_tail = True

    def step():
        wl.qap()
        return prover.prove(wl.a, wl.b, wl.c, wl.w, wl.rs, scalars_mont=False)

    def full_sync():
        for ch in range(3):
            ctx.sync(ch)
        torch.cuda.synchronize()

    # time to first proof: what a prover that has just loaded a key waits for -- the window tables (pk_build_s, in
    # Workload) plus the first proof on a cold context (workspace allocations, twiddle tables, code objects)
    t_first = time.perf_counter()
    step()
    full_sync()
    first_proof_s = time.perf_counter() - t_first
    for _ in range(max(0, args.warmup - 1)):
        step()
    full_sync()
    # the roof of THIS box in THIS run (every rank measures its own GPU; rank 0's figures go on the line)
    calib = calibrate(local_rank)
    mad_peak = calib["mad_issue_T_lane_ops_per_s"] if calib else MAD_ISSUE_T
    for _ in range(1 if calib else 0):     # (one more untimed step: the calibration kernel evicted the proof's working set)
        step()
    full_sync()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    queued = []
    for _ in range(args.steps):
        proof = step()
        queued.append(proof)
    full_sync()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if single_dev else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = wl.nc * args.steps / elapsed
    single_ms = None
    if world == 1:
        ts = []
        for _ in range(min(args.steps, 5)):
            t1 = time.perf_counter()
            proof = step()
            full_sync()
            ts.append(time.perf_counter() - t1)
        single_ms = sum(ts) / len(ts) * 1e3
        # same inputs, same r and s: every queued proof must be the bytes of the synchronised one (which the oracle checks)
        # (compared as affine points: the order of the additions inside a bucket is not fixed, so the Jacobian
        # representatives of the same point may differ from run to run)
        want = gpu_proof_affine(curve, proof.cpu().numpy())
        bad = [i for i, q in enumerate(queued)
               if not all(np.array_equal(x, y) for x, y in zip(gpu_proof_affine(curve, q.cpu().numpy()), want))]
        if bad:
            raise SystemExit("queued proofs %s differ from the synchronised proof" % bad)
    del queued

    # ---- dominant kernel: the G2 bucket accumulation of the LAST TIMED PROOF (HIP events recorded around it on
    # the stream it ran on: dg16_last_kernel_ms, channel 2 = G2 accumulation, channel 1 = B1's G1 accumulation) ----
    g2_acc_ms = ctx.last_kernel_ms(2, 1)
    g1_acc_ms = ctx.last_kernel_ms(1, 1)            # B1's accumulation (A's runs beside B's G2 finalize)
    # ... and the shader clock the chip held UNDER each of them, measured by the kernels themselves (ClkProbe, msm_impl.h):
    # 16 lanes per SIMD-cycle x 4 SIMDs x CUs x that clock is the issue bound no DVFS state can move
    g2_mhz = ctx.last_kernel_mhz(2) if hasattr(ctx, "last_kernel_mhz") else 0.0
    g1_mhz = ctx.last_kernel_mhz(1) if hasattr(ctx, "last_kernel_mhz") else 0.0
    n_cus = ctx.device_info()[1]

    def cycle_view(mhz, mads_t, mads_per_add, slots_per_add):
        if not mhz:
            return None
        bound = 16.0 * 4 * n_cus * mhz * 1e6 / 1e12
        out = {"sclk_under_kernel_mhz": mhz, "issue_bound_at_kernel_clock_T": bound, "mads_frac_of_issue_bound": mads_t / bound}
        if slots_per_add:
            out["valu_slots_per_add"] = slots_per_add
            out["all_valu_frac_of_issue_bound"] = mads_t * slots_per_add / mads_per_add / bound
        return out
    info = wl.pk.info()
    n_g2 = info["n_ab"]                              # points of this rank's A / B1 / B launches (slice + 2 delta slots)
    nwin = (SCALAR_BITS[curve] + 1 + info["c_ab"] - 1) // info["c_ab"]
    g2_alg = {"bn254": 160.0, "bls12_381": 224.0}[curve]
    g1_alg = {"bn254": 96.0, "bls12_381": 128.0}[curve]          # G1 affine point + 32-byte scalar
    alg_bytes = g2_alg * n_g2
    achieved = alg_bytes / (g2_acc_ms * 1e-3) / 1e9 if g2_acc_ms else 0.0
    # one XYZZ mixed addition per nonzero digit, counted in the instruction that bounds it (add_mads)
    g1_add_mads, g2_add_mads, mul_cost = add_mads(curve)
    mads = float(g2_add_mads) * n_g2 * nwin
    valu_t = mads / (g2_acc_ms * 1e-3) / 1e12 if g2_acc_ms else 0.0       # T v_mad_u64_u32 lane-op/s
    _, mul_rate_g, mul_src = valu_constants(curve)   # measured G products/s of the C++-compiled product (auxiliary)
    prods = proof_products(curve, info, args.log_m, wl.nc, world=world)
    whole_t = prods["total"] / (ms_per_step * 1e-3) / 1e12     # per GPU: `prods` is this rank's share
    g2_kernel = ("msm_accumulate_lds_kernel<Fp2<%s_fq>>" if curve == "bn254" else "msm_accumulate_steps_kernel<Fp2<%s_fq>>") % curve

    # HBM traffic of that kernel: PMC counters cannot be read from inside the process; the committed summary of
    # the separate rocprofv3 --pmc passes over the same launch (same curve, group, size) supplies it.
    traffic, traffic_src, traffic_g1 = None, None, None
    if curve == "bls12_381" and args.log_m == 20 and world == 1:
        traffic, traffic_g1, traffic_src = pmc_traffic("bls12_381")
    for name in ("r6_pmc_g2_accumulate.json", "r5_pmc_g2_accumulate.json", "r4v_pmc_g2_accumulate.json", "r4p_pmc_g2_accumulate.json", "r4l_pmc_g2_accumulate.json", "r4f_pmc_g2_accumulate.json", "r4a_pmc_g2_accumulate.json", "r3_pmc_g2_accumulate.json", "r2_pmc_g2_accumulate.json",
                 "r1_pmc_g2_accumulate.json"):
        pmc_path = os.path.join(ROOT, "profiles", name)
        if args.log_m == 20 and curve == "bn254" and world == 1 and os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pj = json.load(f)
            traffic = pj["traffic_bytes_per_launch"]
            traffic_g1 = pj.get("g1_traffic_bytes_per_launch")
            traffic_src = "profiles/%s (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes)" % name
            break

    # The peak is an instruction issue rate: a fraction above 1 can only mean a wrong count or a wrong clock.  The line is
    # still printed (a timed run must not vanish in an accounting check), with the fact on it.
    valu_over = valu_t > mad_peak * 1.0001 or whole_t > mad_peak
    peak_src = ("in-run calibration (calibration.mad_issue_T_lane_ops_per_s of this line)" if calib else
                "profiles/r1_ubench_instr_rate.txt (libdg16_calib.so not built)")
    rccl_ranks = prover.rccl_ranks() if hasattr(prover, "rccl_ranks") and world > 1 else None
    if rccl_ranks is not None and rccl_ranks != world:
        raise SystemExit("the RCCL communicator reports %s ranks, the job has %d" % (rccl_ranks, world))
    res = {
        "metric": "groth16_constraints_per_sec",
        "value": value,
        "unit": "constraints/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "single_proof_ms": single_ms,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": "%s Groth16 prove from the matrices, synthetic R1CS with 2^%d - 2 constraints (3 nonzeros "
                               "per row of A and B), 2^%d wires, 2 instance variables; r, s != 0 (R1CS x witness + "
                               "6 NTTs of 2^%d + 4 G1 MSMs + 1 G2 MSM); proving key RESIDENT as window tables "
                               "(built once per key outside the timed region)"
                               % (curve.upper(), args.log_m, args.log_m, args.log_m),
                   "curve": curve, "log_domain": args.log_m,
                   "parallelism": prover.describe(), "rccl_ranks": rccl_ranks,
                   "queue": ("DG16_F_OVERLAP_TAIL: K proofs queued on one context, a proof's last bucket reduction + "
                             "assembly run under the next proof's first kernels" if overlap else
                             "K proofs queued on one context, each complete on channel 0 before the next starts"),
                   "key_table_bytes": info["table_bytes"], "key_table_stride": info["table_stride"],
                   "key_table_build_s": wl.pk_build_s, "first_proof_s": first_proof_s,
                   "time_to_first_proof_s": wl.pk_build_s + first_proof_s,
                   "key_window_bits": {"ab": info["c_ab"], "l": info["c_l"], "h": info["c_h"]}},
        "roofline": {"bound": "hbm", "kernel": g2_kernel + " (G2 bucket accumulation, table mode, inside the "
                     "timed proofs)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel_ms": g2_acc_ms, "points_per_launch": n_g2,
                     "note": "%d B/point algorithmic (each point and scalar once); Pippenger gathers every point once "
                             "per window, which is what the PMC traffic shows -- the kernel is integer-VALU-bound "
                             "(see valu_roofline)" % int(g2_alg)},
        "calibration": calib,
        "valu_roofline": {"unit": "T v_mad_u64_u32 lane-op/s", "achieved": valu_t, "peak": mad_peak,
                          "frac": valu_t / mad_peak, "peak_source": peak_src,
                          "frac_vs_round1_constant": valu_t / MAD_ISSUE_T,
                          "cycle_view": cycle_view(g2_mhz, valu_t, g2_add_mads, {"bn254": 5900}.get(curve)),
                          "mads_per_g1_add": g1_add_mads, "mads_per_g2_add": g2_add_mads, "mads_per_product": mul_cost,
                          "product_equivalents_G_per_s": valu_t * 1e3 / mul_cost,
                          "measured_product_rate_G_per_s": mul_rate_g, "measured_product_rate_source": mul_src,
                          "whole_proof_mads": prods, "whole_proof_achieved": whole_t,
                          "whole_proof_valu_frac": whole_t / mad_peak,
                          "exceeds_peak": valu_over,
                          "note": "%d v_mad_u64_u32 per G2 mixed add x %d points x %d windows / kernel time against the "
                                  "chip-wide issue rate of that instruction (%.1f T lane-op/s: the bound no rewrite of the "
                                  "product can move).  product_equivalents = achieved / %d mads; measured_product_rate = "
                                  "tools/ubench/fe_rate (a dependent chain of the library's own product per lane). "
                                  "whole_proof_* = every v_mad_u64_u32 a proof must issue (per GPU) / ms_per_step: the "
                                  "headroom of the whole pipeline, not of one kernel"
                                  % (g2_add_mads, n_g2, nwin, mad_peak, mul_cost)},
        "g1_accumulate_ms": g1_acc_ms,
        # the G1 accumulation: four launches per proof (A, B1, L, H), more TOTAL time than the G2 launch and further from
        # the issue roof -- on the line next to the G2 figures since round 5
        "roofline_g1": {"bound": "hbm", "kernel": "msm_accumulate_kernel<Fp<%s_fq>> (B1's G1 bucket accumulation, table mode, "
                        "inside the timed proofs; B1, L, H are launches of the same kernel)" % curve,
                        "achieved": (g1_alg * n_g2 / (g1_acc_ms * 1e-3) / 1e9) if g1_acc_ms else 0.0, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": (g1_alg * n_g2 / (g1_acc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if g1_acc_ms else 0.0,
                        "traffic": traffic_g1, "traffic_source": traffic_src,
                        "kernel_ms": g1_acc_ms, "points_per_launch": n_g2, "launches_per_proof": 4,
                        "note": "%d B/point algorithmic (point + scalar once)" % int(g1_alg)},
        "valu_roofline_g1": {"unit": "T v_mad_u64_u32 lane-op/s",
                             "achieved": (float(g1_add_mads) * n_g2 * nwin / (g1_acc_ms * 1e-3) / 1e12) if g1_acc_ms else 0.0,
                             "peak": mad_peak, "peak_source": peak_src,
                             "frac": (float(g1_add_mads) * n_g2 * nwin / (g1_acc_ms * 1e-3) / 1e12 / mad_peak) if g1_acc_ms else 0.0,
                             "mads_per_g1_add": g1_add_mads,
                             "valu_slots_per_g1_add": {"bn254": 2089}.get(curve),
                             "cycle_view": cycle_view(g1_mhz, (float(g1_add_mads) * n_g2 * nwin / (g1_acc_ms * 1e-3) / 1e12) if g1_acc_ms else 0.0,
                                                      g1_add_mads, {"bn254": 2089}.get(curve)),
                             "note": "every integer VALU instruction of the loop (mads, column shifts / masks, m[k] products) "
                                     "issues at about the same ~35 T lane-op/s (profiles/r5c_ubench_instr_rate.txt): the "
                                     "loop's 2089 issue slots per addition, not its 1467 mads, are what the time buys -- "
                                     "SQ counters (profiles/r5a_pmc_sq_bn254.md): VALU busy 74 % of the cycles at the nominal "
                                     "2.4 GHz = 83 % at the ~2.13 GHz the chip holds under this kernel "
                                     "(profiles/r5c_clock_under_kernels.txt)"},
        "host": {"cpu_model": cpu_model(), "nproc": os.cpu_count()},
    }
    if world > 1 and not args.no_replicas:
        # Secondary figure, labelled as such: N INDEPENDENT proofs, one whole resident key and one whole proof per
        # GPU, no exchange at all (what a proving service with many statements in flight would run).  Weak scaling.
        rwl = Workload(ctx, dev, args.log_m, 0, 1, curve=curve, seed=20 + rank)
        rprover = make_prover(ctx, rwl.pk, curve, None, 0, 1)

        def rstep():
            rwl.qap()
            return rprover.prove(rwl.a, rwl.b, rwl.c, rwl.w, rwl.rs, scalars_mont=False)

        for _ in range(max(1, args.warmup)):
            rstep()
        full_sync()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rstep()
        full_sync()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if single_dev else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rel = float(t.item())
        res["replicas"] = {"mode": "replicas: %d independent proofs in flight, one per GPU, no collective" % world,
                           "value": world * rwl.nc * args.steps / rel, "unit": "constraints/s", "scaling": "weak",
                           "ms_per_step": rel / args.steps * 1e3, "key_table_bytes_per_gpu": rwl.pk.info()["table_bytes"]}
        rwl.pk.close()
    if rank == 0 and world == 1:
        res["host_pointer_step"] = host_pointer_figure(ctx, wl, prover, min(args.steps, 10))
    if rank == 0 and world == 1 and not args.no_extras:
        extras(ctx, dev, wl, curve, res)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if args.log_m <= args.cpu_sample_log:
            cb, ok = cpu_baseline_and_parity(ctx, dev, args.log_m, curve, wl=wl, gpu_proof=proof.cpu().numpy())
        else:       # e.g. 2^24: prove a bounded 2^cpu_sample_log instance on both sides instead
            cb, ok = cpu_baseline_and_parity(ctx, dev, args.cpu_sample_log, curve)
        # 1-thread run on a smaller sample
        small = Workload(ctx, dev, min(args.log_m, 16), 0, 1, seed=9, curve=curve)
        _, t1 = oracle_prove(small, 1)
        cb["one_thread"] = {"value": small.nc / t1, "unit": "constraints/s", "cores": 1,
                            "sample": "one proof of a 2^%d instance: %.2f s" % (small.log_m, t1)}
        small.pk.close()
        cb["cpu_model"] = cpu_model()
        res["cpu_baseline"] = cb
        res["parity_check"] = "pass" if ok else "FAIL"
        if not ok:
            print(json.dumps(res))
            raise SystemExit("GPU proof differs from the oracle proof")
    if rank == 0 and world > 1 and not args.no_cpu_baseline and args.log_m <= args.cpu_sample_log:
        # live parity gate of the distributed proof: the oracle proves the timed instance on rank 0's host cores
        # (checker only: cpu_baseline is reported at N = 1)
        _, ok = cpu_baseline_and_parity(ctx, dev, args.log_m, curve, wl=wl, gpu_proof=proof.cpu().numpy())
        res["parity_check"] = "pass" if ok else "FAIL"
        if not ok:
            print(json.dumps(res))
            raise SystemExit("distributed GPU proof differs from the oracle proof")
    if rank == 0:
        print(json.dumps(res))
    prover.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
