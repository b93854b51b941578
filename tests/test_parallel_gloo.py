"""N > 1 path on CPU: world_size-2 gloo runs of DistributedProver with an oracle-backed engine standing
in for the GPU (test infrastructure only).  Checks the shard bounds, the sharded h-polynomial's exchange layout
(cyclic rows, two all-to-alls, cyclic h bases), the single all-gather and the assembly: the distributed proof must
equal the single-prover proof of the big-int restatement -- with the replicated and with the sharded h-polynomial."""

import os
import random
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleEngine:
    """Stands in for GpuEngine: same record layout as csrc/prover_impl.h (A', B1', L, H, s*A', r*B1' as G1
    Jacobian, then B' as G2 Jacobian, Montgomery limbs; shard 0 folds the fixed points in), computed with the
    CPU oracle on this rank's slices."""

    def __init__(self, curve, pk, r1cs_dims, shard, n_shards, h_cyclic=False):
        from oracle import corc
        from dg16_amd.parallel import shard_bounds, l_bounds
        self.corc, self.curve, self.pk = corc, curve, pk
        self.nv, self.ni, self.m = r1cs_dims
        self.shard, self.n_shards, self.h_cyclic = shard, n_shards, h_cyclic
        self.ab = shard_bounds(self.nv - 1, shard, n_shards)
        self.lb = l_bounds(self.nv, self.ni, shard, n_shards)
        self.hb = shard_bounds(self.m, shard, n_shards)
        self.first = shard == 0
        self.last = shard + 1 == n_shards
        self.rec = (6 * 3 * 4 + 3 * 8) * 8

    def _jac(self, group, aff):
        nl = 4 * (2 if group == 2 else 1)
        out = np.zeros(3 * nl, dtype=np.uint64)
        if aff.any():
            out[:2 * nl] = aff.reshape(-1)
            one = self.corc.field_op(self.curve, "fq", "to_mont", self.corc.ints_to_arr([1], 4)).reshape(-1)
            out[2 * nl:2 * nl + 4] = one
        return out

    # ---- sharded h-polynomial: the stages of oracle/pyref/hdist.py behind GpuEngine's interface ----
    def _ints(self, t):
        arr = t.numpy().view(np.uint64).reshape(-1, 4)
        return self.corc.arr_to_ints(self.corc.field_op(self.curve, "fr", "from_mont", arr))

    def _tensor(self, vals):
        from oracle.pyref.fields import FR
        F = FR[self.curve]
        return torch.from_numpy(self.corc.ints_to_arr([F.to_mont(v) for v in vals], 4).view(np.uint8).reshape(-1).copy())

    def h_stage(self, stage, inputs, rank, world):
        from oracle.pyref import hdist
        from oracle.pyref.fields import FR
        F, m = FR[self.curve], self.m
        M, S = m // world, m // world // world
        if stage == 0:      # rows -> [peer][vector][S]
            Y = [hdist.stage0(self._ints(v), F, m, world) for v in inputs]
            return self._tensor([Y[v][p * S + j] for p in range(world) for v in range(3) for j in range(S)])
        buf = self._ints(inputs[0])
        piece = lambda p, v: buf[(p * 3 + v) * S:(p * 3 + v + 1) * S]      # noqa: E731
        if stage == 1:
            U = [hdist.stage1([piece(p, v) for p in range(world)], F, m, world, rank) for v in range(3)]
            return self._tensor([U[v][q][j] for q in range(world) for v in range(3) for j in range(S)])
        W = [[x for p in range(world) for x in piece(p, v)] for v in range(3)]
        return self._tensor(hdist.stage2(W[0], W[1], W[2], F, m, world))

    def empty_like_bytes(self, t, times=1):
        return torch.empty(t.numel() * times, dtype=torch.uint8)

    def partial_h(self, h, w, rs_host, scalars_mont):
        return self.partial(None, None, None, w, rs_host, scalars_mont, h_shard=h)

    def partial(self, a, b, c, w, rs_host, scalars_mont, h_shard=None):
        corc, cv = self.corc, self.curve
        R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
        r_full = int(sum(int(x) << (64 * i) for i, x in enumerate(rs_host[0])))
        s_full = int(sum(int(x) << (64 * i) for i, x in enumerate(rs_host[1])))
        r, s = (r_full, s_full) if self.last else (0, 0)     # the delta pairs ride on the last shard
        w = w.numpy()
        if h_shard is None:
            h = corc.field_op(cv, "fr", "from_mont", corc.h_poly(cv, a.numpy(), b.numpy(), c.numpy()))
        else:
            h = corc.field_op(cv, "fr", "from_mont", h_shard.numpy().view(np.uint64).reshape(-1, 4))
        pk = self.pk
        sc = lambda v: corc.ints_to_arr([v], 4)
        add = lambda grp, p, q: corc.point_add(cv, grp, p, q)
        lo, hi = self.ab
        A = corc.msm(cv, 1, np.concatenate([pk["a_query"][1:][lo:hi], pk["delta_g1"]]), np.concatenate([w[1:][lo:hi], sc(r)]))
        B1 = corc.msm(cv, 1, np.concatenate([pk["b_g1_query"][1:][lo:hi], pk["delta_g1"]]), np.concatenate([w[1:][lo:hi], sc(s)]))
        B2 = corc.msm(cv, 2, np.concatenate([pk["b_g2_query"][1:][lo:hi], pk["delta_g2"]]), np.concatenate([w[1:][lo:hi], sc(s)]))
        if self.first:                                        # shard 0 folds the fixed points in
            A = add(1, add(1, A, pk["alpha_g1"]), pk["a_query"][0:1])
            B1 = add(1, add(1, B1, pk["beta_g1"]), pk["b_g1_query"][0:1])
            B2 = add(2, add(2, B2, pk["beta_g2"]), pk["b_g2_query"][0:1])
        if r_full == 0:
            B1 = np.zeros_like(B1)
        lo, hi = self.lb
        L = corc.msm(cv, 1, np.concatenate([pk["l_query"][lo:hi], pk["delta_g1"]]),
                     np.concatenate([w[self.ni:][lo:hi], sc((R - r * s % R) % R)]))
        lo, hi = self.hb
        if h_shard is not None:      # cyclic bases h_query[shard + n_shards * j] against this rank's h
            H = corc.msm(cv, 1, np.ascontiguousarray(pk["h_query"][self.shard::self.n_shards][:len(h)]), h)
        else:
            H = corc.msm(cv, 1, pk["h_query"][lo:hi], h[lo:hi])
        sA = corc.point_mul(cv, 1, A, s_full)
        rB1 = corc.point_mul(cv, 1, B1, r_full)
        rec = np.concatenate([self._jac(1, A), self._jac(1, B1), self._jac(1, L), self._jac(1, H), self._jac(1, sA),
                              self._jac(1, rB1), self._jac(2, B2)])
        return torch.from_numpy(rec.view(np.uint8).copy())

    def empty_gather(self, n):
        return torch.empty(n * self.rec, dtype=torch.uint8)

    def assemble(self, gathered, n_shards, rs_host, scalars_mont):
        corc, cv = self.corc, self.curve
        g = gathered.numpy().view(np.uint64).reshape(n_shards, -1)

        def total(group, off, nl):
            acc = np.zeros((1, 2 * nl), dtype=np.uint64)
            for k in range(n_shards):
                acc = corc.point_add(cv, group, acc, corc.jac_to_affine(cv, group, g[k, off:off + 3 * nl]))
            return acc
        A, B1, L, H, sA, rB1 = (total(1, 12 * i, 4) for i in range(6))
        B2 = total(2, 72, 8)
        add = lambda grp, p, q: corc.point_add(cv, grp, p, q)
        return A, B2, add(1, add(1, L, H), add(1, sA, rB1))


def _worker(rank, world, port, q, sharded_h=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dg16_amd  # noqa: F401
    from dg16_amd.parallel import DistributedProver
    from oracle.pyref.fields import FQ, FR
    from oracle.pyref import groth16 as G
    from oracle import corc
    from test_gpu_prover import enc_fr, enc_g1, enc_g2, dec_g1, dec_g2
    curve = "bn254"
    F, Fq = FR[curve], FQ[curve]
    r1cs, w = G.synthetic_r1cs(F, num_constraints=13, num_instance=2, num_witness=17, seed=3)
    rng = random.Random(5)
    td = tuple(rng.randrange(1, F.p) for _ in range(5))
    pk, _ = G.setup(curve, r1cs, td)
    a, b, c, dom = G.qap(r1cs, w, F)
    hpk = {k: enc_g1(Fq, v if isinstance(v, list) else [v]) for k, v in pk.items()
           if k in ("a_query", "b_g1_query", "h_query", "l_query", "alpha_g1", "beta_g1", "delta_g1")}
    hpk.update({k: enc_g2(Fq, v if isinstance(v, list) else [v]) for k, v in pk.items()
                if k in ("b_g2_query", "beta_g2", "delta_g2")})
    eng = OracleEngine(curve, hpk, (len(w), 2, dom.size), rank, world, h_cyclic=sharded_h)
    prover = DistributedProver(eng, dist, rank, world, sharded_h=sharded_h)
    if sharded_h:                    # every rank evaluates only its cyclic rows (dg16_qap_rows on the GPU)
        from dg16_amd.parallel import h_is_sharded
        assert h_is_sharded(dom.size, world)
        a, b, c = a[rank::world], b[rank::world], c[rank::world]
    r, s = rng.randrange(1, F.p), rng.randrange(1, F.p)
    rs = corc.ints_to_arr([r, s], 4)
    t = lambda v: torch.from_numpy(enc_fr(F, v).view(np.int64))
    wc = torch.from_numpy(corc.ints_to_arr([x % F.p for x in w], 4))
    gA, gB, gC = prover.prove(t(a), t(b), t(c), wc, rs, scalars_mont=False)
    exp = G.create_proof(curve, pk, r, s, r1cs, w)
    ok = (dec_g1(Fq, gA), dec_g2(Fq, gB), dec_g1(Fq, gC)) == exp
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("sharded_h", [False, True])
def test_two_rank_gloo_proof_equals_single_prover(sharded_h):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, sharded_h)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_bounds_cover_range():
    from dg16_amd.parallel import shard_bounds, l_bounds
    for nv, ni in ((20, 3), (1 << 20, 2), (9, 9), (10, 1)):
        for world in (1, 2, 3, 8):
            pieces = [l_bounds(nv, ni, k, world) for k in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == nv - ni
            assert all(pieces[i][1] == pieces[i + 1][0] for i in range(world - 1))
    for n in (0, 1, 7, 1048575):
        for world in (1, 2, 3, 8):
            pieces = [shard_bounds(n, k, world) for k in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == n
            assert all(pieces[i][1] == pieces[i + 1][0] for i in range(world - 1))


def test_make_prover_picks_the_driver_by_rank_count():
    """Host logic of parallel.make_prover (no GPU: dummy context / key): one rank -> the plain prover; 2 / 4 / 8 ranks
    with m >= N^2 -> the native pipeline over the chosen transport; other rank counts (or a domain smaller than N^2) ->
    the Python-driven protocol with the replicated h-polynomial."""
    from dg16_amd import parallel as P

    class Ctx:
        device = 0

        @staticmethod
        def results_bytes(curve):
            return 768

    class Pk:
        def __init__(self, m):
            self.domain_size = m

    class Dist:
        @staticmethod
        def get_backend():
            return "gloo"

    assert P.h_is_sharded(1 << 20, 8) and P.h_is_sharded(64, 8) and not P.h_is_sharded(32, 8)
    assert not P.h_is_sharded(1 << 20, 3) and not P.h_is_sharded(1 << 20, 16) and not P.h_is_sharded(1 << 20, 1)
    one = P.make_prover(Ctx(), Pk(1 << 10), "bn254", None, 0, 1)
    assert isinstance(one, P.NativeProver) and one.describe() == "single GPU"
    three = P.make_prover(Ctx(), Pk(1 << 10), "bn254", Dist(), 1, 3, transport="rccl")
    assert isinstance(three, P.DistributedProver) and not three.sharded_h
    tiny = P.make_prover(Ctx(), Pk(32), "bn254", Dist(), 0, 8, transport="rccl")
    assert isinstance(tiny, P.DistributedProver) and not tiny.sharded_h
    py = P.make_prover(Ctx(), Pk(1 << 10), "bn254", Dist(), 0, 4, transport="python")
    assert isinstance(py, P.DistributedProver) and py.sharded_h and "sharded h-polynomial" in py.describe()


def _fallback_worker(rank, world, port, q, failure):
    """One rank of the transport decision: `failure` = 'id' (rank 0 cannot make the RCCL id), 'join' (rank 1 fails
    inside the join, promptly), 'load1' (librccl is missing on rank 1 only) or 'samedev' (both ranks sit on one device) --
    the last two are preconditions the ranks compare BEFORE anyone enters the blocking join.  Every rank must come out with the SAME transport, and a collective issued afterwards must
    still match up across the ranks (the round-2 code left rank 0 on torch and rank 1 inside the broadcast)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        from dg16_amd import lib, parallel as P

        class Ctx:
            device = 0 if failure == "samedev" else rank      # (no torch device here: the index is the identity)

        class Pk:
            domain_size = 1 << 10

        class FakeRccl:
            def __init__(self, ctx, uid, n, r):
                # a precondition failure must keep EVERY rank out of the (blocking) join
                assert failure not in ("load1", "samedev"), "ncclCommInitRank entered although a precondition failed"
                if failure == "join" and r == 1:
                    raise lib.Dg16Error(6, "ncclCommInitRank: unhandled system error (simulated)")
                self.closed = False

            def describe(self):
                return "native RCCL (fake)"

            def close(self):
                self.closed = True

        def fake_id():
            if failure == "id" or (failure == "load1" and rank == 1):
                raise lib.Dg16Error(7, "librccl not found (simulated)")
            return b"\0" * 128

        lib.rccl_unique_id, lib.RcclComm = fake_id, FakeRccl
        prover = P.make_prover(Ctx(), Pk(), "bn254", dist, rank, world, transport="rccl")
        kind = type(prover.comm).__name__
        t = torch.tensor([rank + 1])
        dist.all_reduce(t)                     # the ranks are still in step
        q.put((rank, kind, int(t.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("failure", ["id", "join", "load1", "samedev", "none"])
def test_rccl_fallback_is_a_collective_decision(failure):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q, failure)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    want = "FakeRccl" if failure == "none" else "TorchComm"
    assert res == [(0, want, 3), (1, want, 3)]


def _torchnet_worker(rank, world, port, q):
    """One party of three concurrent 'protocols', one per channel, each a gather to the king followed by a scatter from
    it (the shape of d_msm's exchange, dmsm/mod.rs:88-97), driven from three host threads whose per-channel delays are
    ordered differently on every party -- all payloads have the same size, so a cross-matched channel would not fail,
    it would deliver another channel's bytes."""
    import ctypes
    import threading
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        from dg16_amd import lib
        rng = random.Random(1000 + rank)
        lock = threading.Lock()

        def before(channel, op):
            with lock:
                d = rng.uniform(0.0, 0.02) + 0.03 * ((channel + rank) % 3)     # the slow channel differs per party
            time.sleep(d)

        net = lib.TorchNet(dist, torch.device("cpu"), world, rank, before=before)
        vt = net.struct
        nbytes = 96
        ok = [True] * 3

        def protocol(c):
            for it in range(4):
                send = np.full(nbytes, 16 * c + rank + 64 * it, dtype=np.uint8)
                gathered = np.zeros(nbytes * world, dtype=np.uint8)
                rc = vt.gather_to_king(None, c, send.ctypes.data, nbytes, gathered.ctypes.data if rank == 0 else None,
                                       None)
                if rank == 0:
                    want = np.concatenate([np.full(nbytes, 16 * c + p + 64 * it, dtype=np.uint8) for p in range(world)])
                    ok[c] &= rc == 0 and np.array_equal(gathered, want)
                    back = np.concatenate([np.full(nbytes, 200 - 16 * c - p - it, dtype=np.uint8) for p in range(world)])
                else:
                    ok[c] &= rc == 0
                    back = None
                got = np.zeros(nbytes, dtype=np.uint8)
                rc = vt.scatter_from_king(None, c, back.ctypes.data if rank == 0 else None, nbytes, got.ctypes.data, None)
                ok[c] &= rc == 0 and bool((got == 200 - 16 * c - rank - it).all())
            # MpcNet's required pair on the same channel: a ring
            nxt, prv = (rank + 1) % world, (rank - 1) % world
            mine = np.full(nbytes, 7 * c + rank, dtype=np.uint8)
            got = np.zeros(nbytes, dtype=np.uint8)
            if rank == 0:
                a = vt.send_to(None, nxt, c, mine.ctypes.data, nbytes, None)
                b = vt.recv_from(None, prv, c, got.ctypes.data, nbytes, None)
            else:
                b = vt.recv_from(None, prv, c, got.ctypes.data, nbytes, None)
                a = vt.send_to(None, nxt, c, mine.ctypes.data, nbytes, None)
            ok[c] &= a == 0 and b == 0 and bool((got == 7 * c + prv).all())

        order = [(rank + i) % 3 for i in range(3)]            # thread start order differs per party as well
        ths = [threading.Thread(target=protocol, args=(c,)) for c in order]
        for t in ths:
            t.start()
        for t in ths:
            t.join(100)
        bad = vt.gather_to_king(None, 3, None, 0, None, None)      # not a MultiplexedStreamID
        q.put((rank, all(ok), net.errors[:-1], bad))
    finally:
        dist.destroy_process_group()


def test_torchnet_channels_are_independent():
    """lib.TorchNet (the caller-side MpcNet vtable over torch.distributed): one process group per MultiplexedStreamID
    (mpc-net/src/lib.rs:29-33).  Three parties x three channels, all in flight at once, each party completing its
    channels in a different order: every payload arrives on its own channel."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_torchnet_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True, [], 6) for r in range(world)]


def _probe_worker(rank, world, port, q, stall):
    """One party of the channel watchdog (lib.probe_channels).  stall: the party whose channel 1 sleeps past the soft
    deadline on first use (a transport whose channels share a pipe looks like this from every party), or None."""
    import threading
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        from dg16_amd import lib
        first = {"seen": False}
        lock = threading.Lock()

        def before(channel, op):
            if stall == rank and channel == 1:
                with lock:
                    hit, first["seen"] = not first["seen"], True
                if hit:
                    time.sleep(2.5)

        net = lib.TorchNet(dist, torch.device("cpu"), world, rank, before=before)
        mode = lib.probe_channels(net.struct, world, rank, dist, soft_s=1.0, hard_s=60.0)
        # the net is usable afterwards: a gather + scatter on every channel in the decided mode's order
        ok = True
        for c in range(3):
            send = np.full(16, 10 * c + rank, dtype=np.uint8)
            gathered = np.zeros(16 * world, dtype=np.uint8)
            rc = net.struct.gather_to_king(None, c, send.ctypes.data, 16, gathered.ctypes.data if rank == 0 else None, None)
            ok &= rc == 0 and (rank != 0 or bool((gathered.reshape(world, 16)[:, 0] == 10 * c + np.arange(world)).all()))
        q.put((rank, mode, ok, net.errors))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("stall", [None, 1])
def test_channel_watchdog_lands_every_party_on_the_serial_form(stall):
    """The three-communicator form of prove::C (dg16_prove_c: three host threads, one per MultiplexedStreamID) is only
    safe on a transport whose channels make progress independently.  lib.probe_channels exercises exactly that access
    pattern on tiny payloads before the first proof and decides COLLECTIVELY: with healthy channels every party gets
    "joined"; when ONE party's channel 1 stalls past the soft deadline, EVERY party -- also those whose own probe was
    in time -- lands on "serial" (DG16_F_SERIAL_CHANNELS), and the transport is intact afterwards."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_probe_worker, args=(r, world, port, q, stall)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    want = "joined" if stall is None else "serial"
    assert res == [(r, want, True, []) for r in range(world)]


def test_bench_dry_run_plan_matches_the_keys_the_gpu_built():
    """`bench.py --gpus N --dry-run` touches no GPU and prints the plan of the run: per-rank shard sizes, window bits and
    table bytes by the formulas of csrc/prover_impl.h / msm_impl.h.  Pinned against what the library itself reported for
    keys it built on the GPU (profiles/): BN254 2^20 one key 6 039 807 360 B (r5z_bench_line.json), BLS12-381 2^20 over 8
    shards 9 663 853 056 B, 2^24 over 8 shards 144 955 311 840 B (r5g_*_full_parity.json)."""
    import json
    import subprocess
    bench = os.path.join(ROOT, "bench.py")

    def plan(*args):
        out = subprocess.run([sys.executable, bench, "--dry-run"] + list(args), capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout)

    d = plan()
    assert d["dry_run"] and d["n_gpus"] == 1 and d["ranks"][0]["table_bytes"] == 6039807360 and d["exchanges_per_proof"] == {}
    d = plan("--gpus", "8", "--curve", "bls12_381", "--log-m", "20")
    assert sum(r["table_bytes"] for r in d["ranks"]) == 9663853056 and d["h_polynomial_sharded"]
    d = plan("--gpus", "8", "--curve", "bls12_381", "--log-m", "24")
    assert sum(r["table_bytes"] for r in d["ranks"]) == 144955311840
    assert d["table_bytes_per_rank_max"] < d["hbm_per_gpu_bytes"]
    assert "--nproc-per-node 8" in d["commands"]["this_run"] and "--master-addr 127.0.0.1" in d["commands"]["this_run"]
    d = plan("--gpus", "3")            # not a rank count the sharded h-polynomial takes: replicated
    assert not d["h_polynomial_sharded"] and "replicated" in d["exchanges_per_proof"]["h_polynomial"]
