"""GPU parity: dg16_groth16_prove vs the restated single prover (oracle/pyref/groth16.py), on a
satisfied synthetic R1CS with a known-trapdoor setup -- the proof must equal the big-int prover's
and satisfy the pairing equation in the exponent.  Bit-exact after normalising to affine."""

import random

import numpy as np
import pytest

from oracle import corc
from oracle.pyref.fields import FQ, FR
from oracle.pyref.curves import CURVES
from oracle.pyref import groth16 as G
from oracle.pyref import pairing as PR
from gpu_util import ctx

pytestmark = pytest.mark.gpu


def enc_fr(F, vals):
    return corc.ints_to_arr([F.to_mont(v) for v in vals], 4)


def enc_g1(Fq, pts):
    out = np.zeros((len(pts), 2 * Fq.limbs64), dtype=np.uint64)
    for i, P in enumerate(pts):
        if P is not None:
            out[i] = corc.ints_to_arr([Fq.to_mont(P[0]), Fq.to_mont(P[1])], Fq.limbs64).reshape(-1)
    return out


def enc_g2(Fq, pts):
    out = np.zeros((len(pts), 4 * Fq.limbs64), dtype=np.uint64)
    for i, P in enumerate(pts):
        if P is not None:
            flat = [P[0][0], P[0][1], P[1][0], P[1][1]]
            out[i] = corc.ints_to_arr([Fq.to_mont(v) for v in flat], Fq.limbs64).reshape(-1)
    return out


def dec_g1(Fq, arr):
    v = [Fq.from_mont(x) for x in corc.arr_to_ints(np.asarray(arr).reshape(-1, Fq.limbs64))]
    return None if all(x == 0 for x in v) else (v[0], v[1])


def dec_g2(Fq, arr):
    v = [Fq.from_mont(x) for x in corc.arr_to_ints(np.asarray(arr).reshape(-1, Fq.limbs64))]
    return None if all(x == 0 for x in v) else ((v[0], v[1]), (v[2], v[3]))


@pytest.mark.parametrize("curve,nc,nw", [("bn254", 28, 40), ("bn254", 60, 30), ("bls12_381", 13, 20)])
def test_prove_matches_bigint_prover(curve, nc, nw):
    F, Fq = FR[curve], FQ[curve]
    ni = 2
    r1cs, w = G.synthetic_r1cs(F, num_constraints=nc, num_instance=ni, num_witness=nw, seed=nc)
    assert G.is_satisfied(r1cs, w, F.p)
    rng = random.Random(5)
    td = tuple(rng.randrange(1, F.p) for _ in range(5))
    pk, sc = G.setup(curve, r1cs, td)
    a, b, c, dom = G.qap(r1cs, w, F)
    m = dom.size
    c_ = ctx()
    fixed = np.concatenate([enc_g1(Fq, [pk["alpha_g1"], pk["beta_g1"], pk["delta_g1"]]).reshape(-1),
                            enc_g2(Fq, [pk["beta_g2"], pk["delta_g2"]]).reshape(-1)])
    dpk = c_.pk_create(curve, ni + nw, ni, m, enc_g1(Fq, pk["a_query"]), enc_g1(Fq, pk["b_g1_query"]),
                       enc_g2(Fq, pk["b_g2_query"]), enc_g1(Fq, pk["h_query"]), enc_g1(Fq, pk["l_query"]), fixed)
    g1, g2 = CURVES[curve, "g1"], CURVES[curve, "g2"]
    for r, s in ((0, 0), (rng.randrange(1, F.p), rng.randrange(1, F.p))):
        for mont in (True, False):
            enc = (lambda v: enc_fr(F, v)) if mont else (lambda v: corc.ints_to_arr([x % F.p for x in v], 4))
            A, B, C = c_.prove(dpk, enc_fr(F, a), enc_fr(F, b), enc_fr(F, c), enc(w), enc([r]), enc([s]),
                               scalars_mont=mont)
            gA = dec_g1(Fq, corc.jac_to_affine(curve, 1, A))
            gB = dec_g2(Fq, corc.jac_to_affine(curve, 2, B))
            gC = dec_g1(Fq, corc.jac_to_affine(curve, 1, C))
            eA, eB, eC = G.create_proof(curve, pk, r, s, r1cs, w)
            assert (gA, gB, gC) == (eA, eB, eC)
        sa, sb, scc = G.proof_scalars_from_trapdoor(r1cs, F, td, sc, r, s, w)
        assert gA == g1.mul(g1.gen, sa) and gB == g2.mul(g2.gen, sb) and gC == g1.mul(g1.gen, scc)
        assert G.verify_in_exponent(r1cs, F, td, sc, (sa, sb, scc), w)
    # ... and the GPU's last (blinded) proof is accepted by the pairing verifier that accepts the reference's
    # snarkjs proof (tests/test_oracle_pairing.py): e(A, B) = e(alpha, beta) e(IC(x), gamma) e(C, delta)
    vk = {"alpha_g1": pk["alpha_g1"], "beta_g2": pk["beta_g2"], "gamma_g2": pk["gamma_g2"],
          "delta_g2": pk["delta_g2"], "ic": pk["gamma_abc_g1"]}
    assert PR.groth16_verify(curve, vk, w[1:ni], (gA, gB, gC))
    assert not PR.groth16_verify(curve, vk, [(w[1] + 1) % F.p], (gA, gB, gC))
    dpk.close()


def test_sharded_prove_equals_unsharded():
    """The multi-GPU split run as N shards on one GPU: per-shard MSM records, concatenated as the
    all-gather would, assembled -> the same proof as the unsharded key."""
    curve = "bn254"
    F, Fq = FR[curve], FQ[curve]
    ni, nc, nw = 2, 45, 50
    r1cs, w = G.synthetic_r1cs(F, num_constraints=nc, num_instance=ni, num_witness=nw, seed=11)
    rng = random.Random(8)
    td = tuple(rng.randrange(1, F.p) for _ in range(5))
    pk, _ = G.setup(curve, r1cs, td)
    a, b, c, dom = G.qap(r1cs, w, F)
    c_ = ctx()
    fixed = np.concatenate([enc_g1(Fq, [pk["alpha_g1"], pk["beta_g1"], pk["delta_g1"]]).reshape(-1),
                            enc_g2(Fq, [pk["beta_g2"], pk["delta_g2"]]).reshape(-1)])
    args = (curve, ni + nw, ni, dom.size, enc_g1(Fq, pk["a_query"]), enc_g1(Fq, pk["b_g1_query"]),
            enc_g2(Fq, pk["b_g2_query"]), enc_g1(Fq, pk["h_query"]), enc_g1(Fq, pk["l_query"]), fixed)
    r, s = rng.randrange(1, F.p), rng.randrange(1, F.p)
    exp = G.create_proof(curve, pk, r, s, r1cs, w)
    for world in (1, 2, 3, 8):
        recs = []
        keys = [c_.pk_create(*args, shard=k, n_shards=world) for k in range(world)]
        for k in range(world):
            recs.append(c_.groth16_msms(keys[k], enc_fr(F, a), enc_fr(F, b), enc_fr(F, c), enc_fr(F, w),
                                        enc_fr(F, [r]), enc_fr(F, [s])))
        A, B, C = c_.groth16_assemble(keys[0], np.concatenate(recs), enc_fr(F, [r]), enc_fr(F, [s]))
        got = (dec_g1(Fq, corc.jac_to_affine(curve, 1, A)), dec_g2(Fq, corc.jac_to_affine(curve, 2, B)),
               dec_g1(Fq, corc.jac_to_affine(curve, 1, C)))
        assert got == exp, world
        for k in keys:
            k.close()


@pytest.mark.parametrize("curve,log_m", [("bn254", 6), ("bn254", 9), ("bn254", 12), ("bn254", 14), ("bn254", 16),
                                         ("bls12_381", 12), ("bls12_381", 14), ("bls12_381", 16)])
def test_prove_medium_sizes_vs_oracle(curve, log_m):
    """Resident-table prover at sizes where every phase is multi-block (scan, segments, two-stage sums): the
    proof of a synthetic 2^log_m instance FROM THE MATRICES (dg16_qap + dg16_groth16_prove) must equal the C
    oracle's proof of the same inputs.  BLS12-381 = BASELINE config 5's curve (384-bit base field: the 12-limb
    multiply, the BLOCK = 128 G2 accumulation kernel)."""
    import torch
    import bench
    cb, ok = bench.cpu_baseline_and_parity(ctx(), torch.device("cuda", 0), log_m, curve)
    assert ok
    # the accumulation kernels of that proof measured the clock they ran under (ClkProbe: dg16_last_kernel_ms, which = 2):
    # a plausible shader clock for the G2 launch (channel 2) and A's G1 launch (channel 1)
    for ch in (2, 1):
        mhz = ctx().last_kernel_mhz(ch)
        assert 500.0 < mhz < 3000.0, (ch, mhz)


@pytest.mark.parametrize("curve,log_m,frac", [("bn254", 14, 0.5), ("bn254", 14, 0.26), ("bn254", 12, 0.01),
                                              ("bls12_381", 12, 0.4)])
def test_prove_under_a_table_budget(curve, log_m, frac):
    """dg16_ctx_set_table_budget: a key whose full window tables exceed the budget keeps every k-th table row and proves
    with k bucket sets + a Horner tail.  Same proof as the oracle's (and therefore as the unbudgeted key's), the
    tables fit the budget, and the stride is what the budget forces (frac = 0.01: down to the plain bases)."""
    import torch
    import bench
    import dg16_amd
    c = dg16_amd.Context(0)               # a context of its own: the budget is a context setting
    dev = torch.device("cuda", 0)
    full = bench.Workload(c, dev, log_m, 0, 1, seed=31, curve=curve)
    fi = full.pk.info()
    assert fi["table_stride"] == 1
    full.pk.close()
    budget = int(fi["table_bytes"] * frac)
    c.set_table_budget(budget)
    wl = bench.Workload(c, dev, log_m, 0, 1, seed=31, curve=curve)
    info = wl.pk.info()
    bits = bench.SCALAR_BITS[curve]
    nwin = (bits + 1 + info["c_ab"] - 1) // info["c_ab"]
    assert info["table_stride"] > 1
    # the floor is one row (the bases themselves): the budget is met unless it is below that
    assert info["table_bytes"] <= budget or info["table_stride"] >= min(nwin, (bits + 1 + info["c_h"] - 1) // info["c_h"])
    assert info["table_bytes"] < fi["table_bytes"]
    if frac == 0.5:
        assert info["table_stride"] in (2, 3)       # 17 windows: 9 rows are 53 % of the full table, 6 rows fit
    gp = bench.prove_once(c, wl)
    (A, B, C), _ = bench.oracle_prove(wl, bench.cpu_threads())
    gA, gB, gC = bench.gpu_proof_affine(curve, gp)
    assert np.array_equal(A, gA) and np.array_equal(B, gB) and np.array_equal(C, gC)
    wl.pk.close()
    c.set_table_budget(0)
    c.close()


@pytest.mark.parametrize("rs_zero", [True, False])
def test_prove_sha256_shaped_config4(rs_zero):
    """BASELINE config 4.  The reference's fixtures/sha256/sha256.r1cs is a missing blob, so
    the instance is sha256-SHAPED: 29 823 wires, 2 instance variables, domain 2^15 (groth16/examples/sha256.rs runs
    exactly this shape), r = s = 0 as at sha256.rs:152-153 and random r, s."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    wl = bench.Workload(ctx(), dev, 15, 0, 1, seed=4, curve="bn254", nv=29823, nc=29400, ni=2)
    rs = np.zeros((2, 4), dtype=np.uint64) if rs_zero else wl.rs
    gp = bench.prove_once(ctx(), wl, rs)
    r = int(sum(int(v) << (64 * i) for i, v in enumerate(rs[0])))
    s = int(sum(int(v) << (64 * i) for i, v in enumerate(rs[1])))
    (A, B, C), _ = bench.oracle_prove(wl, bench.cpu_threads(), r, s)
    gA, gB, gC = bench.gpu_proof_affine("bn254", gp)
    assert np.array_equal(A, gA) and np.array_equal(B, gB) and np.array_equal(C, gC)
    wl.pk.close()


@pytest.mark.parametrize("curve,log_m,world", [("bls12_381", 12, 2), ("bls12_381", 14, 8), ("bn254", 16, 8),
                                               ("bls12_381", 16, 1)])
def test_sharded_prove_large_vs_oracle(curve, log_m, world):
    """N shard keys on one GPU (what N ranks hold), records concatenated as the all-gather would, assembled: the
    proof must equal the C oracle's, at sizes where the shards run the large-MSM paths.  Replicated h-polynomial,
    contiguous h slices (the form for rank counts the sharded h-polynomial does not take); the sharded form is
    tests/test_gpu_hdist.py::test_sharded_h_prove_all_ranks_in_process_vs_oracle."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    c_ = ctx()
    shards = [bench.Workload(c_, dev, log_m, k, world, seed=31, curve=curve, h_sharded=False) for k in range(world)]
    recs = []
    for wl in shards:
        rec = torch.empty(c_.results_bytes(curve), dtype=torch.uint8, device=dev)
        wl.qap()
        c_.groth16_msms_dev(wl.pk, wl.a.data_ptr(), wl.b.data_ptr(), wl.c.data_ptr(), wl.w.data_ptr(), wl.rs,
                            rec.data_ptr(), scalars_mont=False)
        for ch in range(3):
            c_.sync(ch)
        recs.append(rec)
    gathered = torch.cat(recs)
    proof = torch.empty(shards[0].proof_bytes(), dtype=torch.uint8, device=dev)
    c_.groth16_assemble_dev(shards[0].pk, gathered.data_ptr(), world, shards[0].rs, proof.data_ptr(), scalars_mont=False)
    c_.sync(0)
    (A, B, C), _ = bench.oracle_prove(shards[0], bench.cpu_threads())
    gA, gB, gC = bench.gpu_proof_affine(curve, proof.cpu().numpy())
    assert np.array_equal(A, gA) and np.array_equal(B, gB) and np.array_equal(C, gC)
    for wl in shards:
        wl.pk.close()


@pytest.mark.parametrize("curve,log_m", [("bn254", 14), ("bn254", 17), ("bls12_381", 12)])
def test_queue_of_proofs_with_overlapped_tail(curve, log_m):
    """DG16_F_OVERLAP_TAIL: five proofs of different witnesses and different (r, s) queued on one context with NO host
    synchronisation in between -- each proof's last bucket reduction and assembly run on channel 2's stream while the next
    proof's R1CS x witness, h-polynomial and first accumulations run on channel 0's -- plus an unrelated MSM on channel 0
    in the middle of the queue (which reuses channel-0 workspace and must order itself behind the tail).  Every queued
    proof must equal the proof of the plain, synchronised call on the same inputs; the first and the last of those are
    checked against the oracle."""
    import torch
    import bench
    c = ctx()
    dev = torch.device("cuda", 0)
    wl = bench.Workload(c, dev, log_m, 0, 1, seed=31, curve=curve)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    K = 5
    ws, rss = [], []
    for k in range(K):
        w = bench.rand_fr(wl.nv, dev, gen, curve)
        w[0] = 0
        w[0, 0] = 1
        ws.append(w)
        rss.append(np.array([[3 + k, 1, 4, 1], [5, 9 + k, 2, 6]], dtype=np.uint64))
    # an unrelated MSM (fresh bases, channel 0)
    n_msm = 1 << 10
    fqb = bench.FQ_BYTES[curve]
    bases = torch.empty(n_msm * 2 * fqb, dtype=torch.uint8, device=dev)
    c.gen_bases_dev(curve, 1, 77, n_msm, bases.data_ptr())
    c.sync(0)
    sc = bench.rand_fr(n_msm, dev, gen, curve)
    msm_ref = torch.empty(3 * fqb, dtype=torch.uint8, device=dev)
    msm_out = torch.empty_like(msm_ref)
    torch.cuda.synchronize()
    c.msm_dev(curve, 1, bases.data_ptr(), sc.data_ptr(), n_msm, msm_ref.data_ptr())
    c.sync(0)

    refs = []
    for k in range(K):
        wl.w = ws[k]
        refs.append(bench.prove_once(c, wl, rss[k]))
    for k in (0, K - 1):
        wl.w = ws[k]
        r = int(sum(int(x) << (64 * i) for i, x in enumerate(rss[k][0])))
        s = int(sum(int(x) << (64 * i) for i, x in enumerate(rss[k][1])))
        (A, B, C), _ = bench.oracle_prove(wl, bench.cpu_threads(), r, s)
        gA, gB, gC = bench.gpu_proof_affine(curve, refs[k])
        assert np.array_equal(A, gA) and np.array_equal(B, gB) and np.array_equal(C, gC), "plain proof %d vs oracle" % k

    outs = [torch.zeros(wl.proof_bytes(), dtype=torch.uint8, device=dev) for _ in range(K)]
    torch.cuda.synchronize()
    for k in range(K):
        wl.w = ws[k]
        wl.qap()
        c.prove_dev(wl.pk, wl.a.data_ptr(), wl.b.data_ptr(), wl.c.data_ptr(), wl.w.data_ptr(), rss[k],
                    outs[k].data_ptr(), scalars_mont=False, overlap_tail=True)
        if k == 2:
            c.msm_dev(curve, 1, bases.data_ptr(), sc.data_ptr(), n_msm, msm_out.data_ptr())
    for ch in range(3):
        c.sync(ch)
    nl = fqb // 8
    assert np.array_equal(corc.jac_to_affine(curve, 1, msm_out.cpu().numpy().view(np.uint64)[:3 * nl]),
                          corc.jac_to_affine(curve, 1, msm_ref.cpu().numpy().view(np.uint64)[:3 * nl])), \
        "the MSM issued in the middle of the queue"
    for k in range(K):
        got = bench.gpu_proof_affine(curve, outs[k].cpu().numpy())
        want = bench.gpu_proof_affine(curve, refs[k])
        assert all(np.array_equal(g, w_) for g, w_ in zip(got, want)), "queued proof %d" % k
    wl.pk.close()
