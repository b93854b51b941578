"""GPU parity of the one-process-per-GPU path, on the one GPU of the test box:

  * the sharded h-polynomial (csrc/ntt.hip: h_poly_dist_stage) -- all N ranks are played by this process, stage by
    stage, with the two all-to-alls done as tensor transposes; every rank's slice must equal the C oracle's
    h[rank + N j] (and, at 2^20, the unsharded GPU h-polynomial, itself oracle-checked in test_gpu_ntt.py);
  * dg16_qap_rows against the oracle's QAP vectors, row subset by row subset;
  * the native RCCL communicator at world size 1 (binding, communicator, grouped send / recv to self, all-gather) and
    dg16_groth16_prove_dist through it;
  * MpcNet's send_to / recv_from on the in-process LocalNet.
The multi-process form (2 and 4 ranks sharing the GPU, gloo transport) is tests/test_gpu_two_rank.py."""

import ctypes
import threading

import numpy as np
import pytest
import torch

from oracle import corc
from gpu_util import ctx

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev_t(arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int64)).to(DEV)


def all_to_all(bufs):
    """bufs[r]: rank r's send buffer as [peer][...]; returns the receive buffers [src][...]."""
    n = len(bufs)
    chunks = [b.view(n, -1) for b in bufs]
    return [torch.stack([chunks[src][dst] for src in range(n)]).contiguous().view(-1) for dst in range(n)]


def sharded_h(c, curve, a, b, cc, n_ranks):
    """All ranks in this process.  a, b, cc: host arrays (m x 4, Montgomery).  Returns [rank] -> host array."""
    m = a.shape[0]
    log_m = m.bit_length() - 1
    M = m // n_ranks
    rows = [[dev_t(v[r::n_ranks]) for v in (a, b, cc)] for r in range(n_ranks)]
    send = [torch.empty(3 * M * 4, dtype=torch.int64, device=DEV) for _ in range(n_ranks)]
    for r in range(n_ranks):
        c.h_poly_dist_stage_dev(curve, log_m, r, n_ranks, 0, [t.data_ptr() for t in rows[r]], send[r].data_ptr())
    c.sync(0)
    recv = all_to_all(send)
    torch.cuda.synchronize()
    send2 = [torch.empty_like(t) for t in recv]
    for r in range(n_ranks):
        c.h_poly_dist_stage_dev(curve, log_m, r, n_ranks, 1, [recv[r].data_ptr()], send2[r].data_ptr())
    c.sync(0)
    recv2 = all_to_all(send2)
    torch.cuda.synchronize()
    out = [torch.empty(M * 4, dtype=torch.int64, device=DEV) for _ in range(n_ranks)]
    for r in range(n_ranks):
        c.h_poly_dist_stage_dev(curve, log_m, r, n_ranks, 2, [recv2[r].data_ptr()], out[r].data_ptr())
    c.sync(0)
    return [t.cpu().numpy().view(np.uint64).reshape(M, 4) for t in out]


@pytest.mark.parametrize("curve,log_m,n_ranks", [("bn254", 2, 2), ("bn254", 6, 2), ("bn254", 6, 4), ("bn254", 6, 8),
                                                 ("bn254", 12, 4), ("bn254", 15, 8), ("bls12_381", 10, 2),
                                                 ("bls12_381", 13, 8), ("bls12_377", 11, 4), ("bn254", 18, 2)])
def test_sharded_h_poly_equals_oracle(curve, log_m, n_ranks):
    m = 1 << log_m
    a, b, cc = (corc.rand_field(curve, "fr", 11 * i + log_m, m) for i in (1, 2, 3))
    ref = corc.h_poly(curve, a, b, cc)
    got = sharded_h(ctx(), curve, a, b, cc, n_ranks)
    for r in range(n_ranks):
        assert np.array_equal(got[r], ref[r::n_ranks]), "rank %d" % r


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_sharded_h_poly_at_2e20_over_8_ranks_equals_unsharded(curve):
    # BASELINE's headline domain: the per-rank transforms are 2^17-point (two-step plans), S = 2^14
    m = 1 << 20
    a, b, cc = (corc.rand_field(curve, "fr", 5 + i, m) for i in (1, 2, 3))
    c = ctx()
    ref = c.h_poly(curve, a, b, cc)
    got = sharded_h(c, curve, a, b, cc, 8)
    for r in range(8):
        assert np.array_equal(got[r], ref[r::8]), "rank %d" % r


@pytest.mark.parametrize("curve,log_m,world", [("bn254", 12, 2), ("bls12_381", 14, 8), ("bn254", 16, 4)])
def test_sharded_h_prove_all_ranks_in_process_vs_oracle(curve, log_m, world):
    """What N ranks do, on one GPU: DG16_F_H_CYCLIC key shards, cyclic QAP rows (dg16_qap_rows), the three stages of
    the sharded h-polynomial with the exchanges as tensor transposes, dg16_groth16_msms_h, records concatenated as
    the all-gather would, assembly -- the proof must equal the C oracle's."""
    import bench
    dev = torch.device(DEV)
    c = ctx()
    shards = [bench.Workload(c, dev, log_m, k, world, seed=33, curve=curve) for k in range(world)]
    assert all(wl.h_sharded and wl.pk.info()["n_h"] == (1 << log_m) // world for wl in shards)
    M = (1 << log_m) // world
    for wl in shards:
        wl.qap()
    c.sync(0)
    send = [torch.empty(3 * M * 4, dtype=torch.int64, device=DEV) for _ in range(world)]
    for r, wl in enumerate(shards):
        c.h_poly_dist_stage_dev(curve, log_m, r, world, 0, [wl.a.data_ptr(), wl.b.data_ptr(), wl.c.data_ptr()],
                                send[r].data_ptr())
    c.sync(0)
    recv = all_to_all(send)
    torch.cuda.synchronize()
    for r in range(world):
        c.h_poly_dist_stage_dev(curve, log_m, r, world, 1, [recv[r].data_ptr()], send[r].data_ptr())
    c.sync(0)
    recv = all_to_all(send)
    torch.cuda.synchronize()
    recs = []
    for r, wl in enumerate(shards):
        h = torch.empty(M * 4, dtype=torch.int64, device=DEV)
        c.h_poly_dist_stage_dev(curve, log_m, r, world, 2, [recv[r].data_ptr()], h.data_ptr())
        rec = torch.empty(c.results_bytes(curve), dtype=torch.uint8, device=DEV)
        c.groth16_msms_h_dev(wl.pk, h.data_ptr(), wl.w.data_ptr(), wl.rs, rec.data_ptr(), scalars_mont=False)
        for ch in range(3):
            c.sync(ch)
        recs.append(rec)
    gathered = torch.cat(recs)
    proof = torch.empty(shards[0].proof_bytes(), dtype=torch.uint8, device=DEV)
    c.groth16_assemble_dev(shards[0].pk, gathered.data_ptr(), world, shards[0].rs, proof.data_ptr(), scalars_mont=False)
    c.sync(0)
    (A, B, C), _ = bench.oracle_prove(shards[0], bench.cpu_threads())
    gA, gB, gC = bench.gpu_proof_affine(curve, proof.cpu().numpy())
    assert np.array_equal(A, gA) and np.array_equal(B, gB) and np.array_equal(C, gC)
    # a cyclic shard refuses the replicated entry point instead of using the wrong h slice
    import dg16_amd
    with pytest.raises(dg16_amd.Dg16Error):
        wl = shards[0]
        c.groth16_msms_dev(wl.pk, wl.a.data_ptr(), wl.b.data_ptr(), wl.c.data_ptr(), wl.w.data_ptr(), wl.rs,
                           recs[0].data_ptr(), scalars_mont=False)
    for wl in shards:
        wl.pk.close()


@pytest.mark.parametrize("curve,log_m,world", [("bls12_381", 20, 8), ("bn254", 20, 8), ("bls12_381", 22, 8)])
def test_config5_sharded_proof_at_size_vs_oracle(curve, log_m, world):
    """BASELINE config 5 on ITS OWN data path at sizes where every shard runs the large-MSM paths (2^17 / 2^19 points
    per shard: partitioned sort, 16-bit table windows, giant buckets of the top window): eight DG16_F_H_CYCLIC shard
    keys of one key in this process, dg16_qap_rows, the three stages of the sharded h-polynomial, dg16_groth16_msms_h
    per shard, eight records, assembly (bench.ShardedInProcess) == the C oracle's proof of the same instance.
    (local_groth_bench.rs:83-158: FFTs + five MSMs on a BLS curve.)  The 2^24 x 8 form:
    `python bench.py --curve bls12_381 --log-m 24 --shards-in-process 8 --full-parity` (profiles/)."""
    import bench
    dev = torch.device(DEV)
    sp = bench.ShardedInProcess(ctx(), dev, curve, log_m, world)
    proof, per_rank, _ = sp.prove()
    assert len(per_rank) == world
    (A, B, C), _ = bench.oracle_prove(sp.shards[0], bench.cpu_threads())
    gA, gB, gC = bench.gpu_proof_affine(curve, proof.cpu().numpy())
    sp.close()
    torch.cuda.empty_cache()
    assert np.array_equal(A, gA) and np.array_equal(B, gB) and np.array_equal(C, gC)


@pytest.mark.parametrize("curve,log_n,n_ranks,inverse", [("bn254", 4, 2, False), ("bn254", 6, 8, True), ("bn254", 12, 4, False),
                                                         ("bls12_381", 13, 8, True), ("bls12_377", 10, 2, False),
                                                         ("bn254", 20, 8, False), ("bn254", 20, 8, True)])
def test_sharded_ntt_one_all_to_all_equals_oracle(curve, log_n, n_ranks, inverse):
    """dg16_ntt_dist_stage: all ranks in this process, the all-to-all as a tensor transpose; rank sigma must end with
    out[k1 S + j] = X[M k1 + sigma S + j] of the oracle's (i)NTT -- the transposed layout of a four-step transform."""
    m = 1 << log_n
    M, S = m // n_ranks, m // n_ranks // n_ranks
    x = corc.rand_field(curve, "fr", 17 + log_n, m)
    X = corc.ntt(curve, x, inverse=inverse)
    c = ctx()
    send = []
    for r in range(n_ranks):
        src = dev_t(x[r::n_ranks])
        dst = torch.empty(M * 4, dtype=torch.int64, device=DEV)
        c.ntt_dist_stage_dev(curve, log_n, r, n_ranks, inverse, 0, src.data_ptr(), dst.data_ptr())
        c.sync(0)
        send.append(dst)
    recv = all_to_all(send)
    torch.cuda.synchronize()
    for r in range(n_ranks):
        out = torch.empty(M * 4, dtype=torch.int64, device=DEV)
        c.ntt_dist_stage_dev(curve, log_n, r, n_ranks, inverse, 1, recv[r].data_ptr(), out.data_ptr())
        c.sync(0)
        got = out.cpu().numpy().view(np.uint64).reshape(n_ranks, S, 4)
        want = X.reshape(n_ranks, n_ranks, S, 4)[:, r]          # [k1][sigma][j] -> sigma = r
        assert np.array_equal(got, want), "rank %d" % r


def test_ntt_dist_through_rccl_world_one_is_the_plain_transform():
    from dg16_amd import lib
    c = ctx()
    comm = lib.RcclComm(c, lib.rccl_unique_id(), 1, 0)
    x = corc.rand_field("bn254", "fr", 5, 1 << 12)
    src = dev_t(x)
    dst = torch.empty_like(src)
    for inverse in (False, True):
        c.ntt_dist_dev("bn254", comm, src.data_ptr(), dst.data_ptr(), 12, inverse=inverse)
        c.sync(0)
        assert np.array_equal(dst.cpu().numpy().view(np.uint64).reshape(-1, 4), corc.ntt("bn254", x, inverse=inverse))
    comm.close()


def test_sharded_h_poly_argument_checks():
    import dg16_amd
    c = ctx()
    t = torch.zeros(3 * 64 * 4, dtype=torch.int64, device=DEV)
    for kw in (dict(log_m=6, rank=0, n_ranks=3), dict(log_m=3, rank=0, n_ranks=4), dict(log_m=6, rank=2, n_ranks=2),
               dict(log_m=8, rank=0, n_ranks=16)):
        with pytest.raises(dg16_amd.Dg16Error):
            c.h_poly_dist_stage_dev("bn254", kw["log_m"], kw["rank"], kw["n_ranks"], 1, [t.data_ptr()], t.data_ptr())


@pytest.mark.parametrize("stride", [1, 2, 8])
def test_qap_rows_are_the_strided_rows_of_qap(stride):
    from oracle.pyref.fields import FR
    from oracle.pyref import groth16 as G
    curve = "bn254"
    F = FR[curve]
    nc, ni = 200, 3
    r1cs, w = G.synthetic_r1cs(F, num_constraints=nc, num_instance=ni, num_witness=90, seed=8)

    def csr(rows):
        ptr, col, val = [0], [], []
        for row in rows:
            for cf, idx in row:
                col.append(idx)
                val.append(cf)
            ptr.append(len(col))
        return (np.asarray(ptr, dtype=np.uint32), np.asarray(col, dtype=np.uint32),
                corc.ints_to_arr([F.to_mont(v) for v in val], 4))

    csr_a, csr_b = csr(r1cs["a"]), csr(r1cs["b"])
    c = ctx()
    wm = corc.ints_to_arr([F.to_mont(x % F.p) for x in w], 4)
    full = c.qap(curve, nc, ni, csr_a, csr_b, wm)
    log_m = full[0].shape[0].bit_length() - 1
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)      # noqa: E731
    ap, ac, av = d(csr_a[0].view(np.int32)), d(csr_a[1].view(np.int32)), dev_t(csr_a[2])
    bp, bc, bv = d(csr_b[0].view(np.int32)), d(csr_b[1].view(np.int32)), dev_t(csr_b[2])
    wd = dev_t(wm)
    rows = (1 << log_m) // stride
    for start in range(stride):
        outs = [torch.empty(rows * 4, dtype=torch.int64, device=DEV) for _ in range(3)]
        c.qap_rows_dev(curve, nc, ni, len(w), log_m, ap.data_ptr(), ac.data_ptr(), av.data_ptr(), bp.data_ptr(),
                       bc.data_ptr(), bv.data_ptr(), wd.data_ptr(), start, stride, *[o.data_ptr() for o in outs])
        c.sync(0)
        for o, f in zip(outs, full):
            assert np.array_equal(o.cpu().numpy().view(np.uint64).reshape(rows, 4), f[start::stride])


def test_native_rccl_comm_world_size_one():
    """One rank: the library binds librccl, forms a communicator and runs its collectives against itself."""
    from dg16_amd import lib
    c = ctx()
    comm = lib.RcclComm(c, lib.rccl_unique_id(), 1, 0)
    vt = ctypes.cast(comm.comm_ptr, ctypes.POINTER(lib.CommStruct)).contents
    assert vt.n_ranks(vt.self) == 1 and vt.rank(vt.self) == 0
    assert comm.ranks() == (1, 0)            # what ncclCommCount / ncclCommUserRank report (bench.py's rccl_ranks)
    src = torch.arange(4096, dtype=torch.int64, device=DEV)
    dst = torch.zeros_like(src)
    st = torch.cuda.current_stream().cuda_stream
    assert vt.all_to_all(vt.self, src.data_ptr(), dst.data_ptr(), src.numel() * 8, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    dst.zero_()
    assert vt.all_gather(vt.self, src.data_ptr(), src.numel() * 8, dst.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    # the same communicator behind the MpcNet vtable: gather / scatter of the king with itself
    net = ctypes.cast(comm.net_ptr, ctypes.POINTER(lib.NetStruct)).contents
    assert net.n_parties(net.self) == 1 and net.party_id(net.self) == 0 and net.is_init(net.self) == 1
    dst.zero_()
    assert net.gather_to_king(net.self, 0, src.data_ptr(), src.numel() * 8, dst.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    comm.close()


@pytest.mark.parametrize("no_split", [False, True])
def test_rccl_net_one_communicator_per_channel_world_one(no_split, monkeypatch):
    """dg16_rccl_net holds one communicator per MultiplexedStreamID (mpc-net/src/lib.rs:29-33), made by ncclCommSplit or
    -- DG16_RCCL_NO_SPLIT=1, the branch for a librccl without it -- joined through two ids broadcast over the first.
    Three host threads drive the three channels at once (what dg16_prove_c does, prove.rs:113-125); a fourth channel
    is refused."""
    from dg16_amd import lib
    monkeypatch.setenv("DG16_RCCL_NO_SPLIT", "1" if no_split else "0")
    c = ctx()
    comm = lib.RcclComm(c, lib.rccl_unique_id(), 1, 0)
    assert c.L.dg16_rccl_channels_split(comm.h) == (0 if no_split else 1)
    assert comm.ranks() == (1, 0)                       # all three communicators agree
    net = ctypes.cast(comm.net_ptr, ctypes.POINTER(lib.NetStruct)).contents
    assert net.is_init(net.self) == 1
    srcs = [torch.full((4096,), 100 + ch, dtype=torch.int64, device=DEV) for ch in range(3)]
    mids = [torch.zeros(4096, dtype=torch.int64, device=DEV) for _ in range(3)]
    dsts = [torch.zeros(4096, dtype=torch.int64, device=DEV) for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    torch.cuda.synchronize()
    rcs = [None] * 3

    def drive(ch):
        st = streams[ch].cuda_stream
        out = []
        for _ in range(8):
            out.append(net.gather_to_king(net.self, ch, srcs[ch].data_ptr(), 4096 * 8, mids[ch].data_ptr(), st))
            out.append(net.scatter_from_king(net.self, ch, mids[ch].data_ptr(), 4096 * 8, dsts[ch].data_ptr(), st))
        streams[ch].synchronize()
        rcs[ch] = out

    ts = [threading.Thread(target=drive, args=(ch,)) for ch in (2, 0, 1)]
    [t.start() for t in ts]
    [t.join(120) for t in ts]
    assert rcs == [[0] * 16] * 3
    for ch in range(3):
        assert torch.equal(dsts[ch], srcs[ch])
    st = torch.cuda.current_stream().cuda_stream
    assert net.gather_to_king(net.self, 3, srcs[0].data_ptr(), 64, mids[0].data_ptr(), st) == 3      # DG16_ERR_BAD_ARG
    assert net.send_to(net.self, 0, 1, srcs[0].data_ptr(), 64, st) == 3                              # peer == me
    comm.close()


def test_prove_dist_through_rccl_world_one_equals_prove():
    import bench
    from dg16_amd import lib
    from dg16_amd.parallel import NativeProver
    c = ctx()
    wl = bench.Workload(c, torch.device(DEV), 10, 0, 1, seed=31)
    ref = bench.prove_once(c, wl)
    comm = lib.RcclComm(c, lib.rccl_unique_id(), 1, 0)
    p = NativeProver(c, wl.pk, "bn254", comm, 0, 1)
    wl.qap()
    got = p.prove(wl.a, wl.b, wl.c, wl.w, wl.rs, scalars_mont=False)
    for ch in range(3):
        c.sync(ch)
    # (Jacobian coordinates are not canonical: compare the points)
    for x, y in zip(bench.gpu_proof_affine("bn254", got.cpu().numpy()), bench.gpu_proof_affine("bn254", ref)):
        assert np.array_equal(x, y)
    p.close()
    wl.pk.close()


@pytest.mark.parametrize("curve,world", [("bn254", 8), ("bls12_381", 4)])
def test_overlapped_queue_of_sharded_proofs_equals_the_synchronised_calls(curve, world):
    """DG16_F_OVERLAP_TAIL on dg16_groth16_prove_dist (round 6): rank 0's shard of a `world`-way proof, driven through a
    STREAM-ORDERED loopback dg16_comm (the exchanges are device copies on the stream the library hands over, like RCCL's;
    the records of the other ranks are copies of rank 0's, so the proof is not a proof -- but it is a deterministic function
    of everything the call reads).  A queue of proofs with different (r, s) and witnesses, no host synchronisation between
    them, H's reduction + all-gather + assembly of proof k on channel 2's stream under the first stage of proof k + 1, must
    give exactly the points the same calls give one at a time: a record, a gathered buffer or the digit-sort metadata of h
    read after the next proof overwrote it would show here.  (Real exchanges: tools/two_rank_check.py, several processes.)"""
    import bench
    from dg16_amd import lib
    from dg16_amd.parallel import NativeProver
    c = ctx()
    dev = torch.device(DEV)

    class Loopback(lib.TorchComm):
        def __init__(self, n):
            self.torch, self.device, self.n_ranks, self.rank, self.errors = torch, dev, n, 0, []
            self._cb = (lib._COMM_N(lambda _s: n), lib._COMM_N(lambda _s: 0), lib._COMM_GATHER(self._all_gather),
                        lib._COMM_A2A(self._all_to_all))
            self.struct = lib.CommStruct(None, *self._cb)
            self.comm_ptr = ctypes.cast(ctypes.pointer(self.struct), ctypes.c_void_p)

        def _on(self, stream, fn):
            try:
                with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)):
                    fn()
                return 0
            except Exception as e:      # noqa: BLE001
                self.errors.append(repr(e))
                return 6

        def _all_gather(self, _s, send, nbytes, recv, stream):
            return self._on(stream, lambda: self._tensor(recv, nbytes * self.n_ranks).view(self.n_ranks, -1).copy_(
                self._tensor(send, nbytes).view(1, -1).expand(self.n_ranks, -1)))

        def _all_to_all(self, _s, send, recv, per_peer, stream):
            return self._on(stream, lambda: self._tensor(recv, per_peer * self.n_ranks).copy_(
                self._tensor(send, per_peer * self.n_ranks)))

    wl = bench.Workload(c, dev, 14, 0, world, seed=41, curve=curve)
    comm = Loopback(world)
    p = NativeProver(c, wl.pk, curve, comm, 0, world)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    ws = [wl.w] + [bench.rand_fr(wl.nv, dev, gen, curve) for _ in range(3)]
    for w in ws[1:]:
        w[0] = 0
        w[0, 0] = 1
    rss = [np.array([[7 + k, 11, 13 * k + 1, 1], [5, 9 + k, 2, 3 + k]], dtype=np.uint64) for k in range(4)]

    def run(overlap):
        p.overlap_tail = overlap
        outs = []
        for w, rs in zip(ws, rss):
            wl.w = w
            wl.qap()
            outs.append(p.prove(wl.a, wl.b, wl.c, wl.w, rs, scalars_mont=False))
            if not overlap:
                for ch in range(3):
                    c.sync(ch)
        for ch in range(3):
            c.sync(ch)
        return [bench.gpu_proof_affine(curve, o.cpu().numpy()) for o in outs]

    alone = run(False)
    for _ in range(2):
        queued = run(True)
        for q, a in zip(queued, alone):
            assert all(np.array_equal(x, y) for x, y in zip(q, a))
    assert not all(np.array_equal(x, y) for x, y in zip(alone[0], alone[1]))
    assert not comm.errors
    p.overlap_tail = False
    wl.pk.close()


def test_localnet_send_to_recv_from():
    """MpcNet::send_to / recv_from (mpc-net/src/lib.rs:48-58) on the in-process net: a ring of three parties, and a
    length mismatch that fails on both sides."""
    from dg16_amd import lib
    from dg16_amd.dist import LocalTestNet
    net = LocalTestNet(3)
    bufs = [torch.full((256,), 10 + i, dtype=torch.int64, device=DEV) for i in range(3)]
    got = [torch.zeros(256, dtype=torch.int64, device=DEV) for _ in range(3)]
    torch.cuda.synchronize()
    res = [None] * 3

    def party(i):
        vt = ctypes.cast(net.party(i), ctypes.POINTER(lib.NetStruct)).contents
        nxt, prv = (i + 1) % 3, (i + 2) % 3
        if i == 0:      # break the ring's symmetry: party 0 sends first, the others receive first
            a = vt.send_to(vt.self, nxt, 1, bufs[i].data_ptr(), 2048, None)
            b = vt.recv_from(vt.self, prv, 1, got[i].data_ptr(), 2048, None)
        else:
            b = vt.recv_from(vt.self, prv, 1, got[i].data_ptr(), 2048, None)
            a = vt.send_to(vt.self, nxt, 1, bufs[i].data_ptr(), 2048, None)
        res[i] = (a, b)

    ts = [threading.Thread(target=party, args=(i,)) for i in range(3)]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert res == [(0, 0)] * 3
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(got[i], bufs[(i + 2) % 3])
    # unequal lengths: DG16_ERR_NET (6) on both sides
    out = [None, None]

    def snd():
        vt = ctypes.cast(net.party(0), ctypes.POINTER(lib.NetStruct)).contents
        out[0] = vt.send_to(vt.self, 1, 0, bufs[0].data_ptr(), 1024, None)

    def rcv():
        vt = ctypes.cast(net.party(1), ctypes.POINTER(lib.NetStruct)).contents
        out[1] = vt.recv_from(vt.self, 0, 0, got[1].data_ptr(), 2048, None)

    ts = [threading.Thread(target=snd), threading.Thread(target=rcv)]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert out == [6, 6]
    net.close()
