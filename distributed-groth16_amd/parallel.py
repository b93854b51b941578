"""One-process-per-GPU Groth16 proving: the king/client split of the reference's mpc-net
(gather to king -> combine -> scatter, dist-primitives/src/dmsm/mod.rs:88-97, dfft/mod.rs:185-256,
mpc-net/src/lib.rs:61-140) re-mapped to the GPUs of one MI355X node.

Every rank holds 1/N of each MSM's bases (`Context.pk_create(..., shard, n_shards, h_cyclic=True)`) and proves in
three exchanges:

  * the h-polynomial is SHARDED (N a power of two, m >= N^2): rank rho evaluates only its cyclic rows
    a[N j + rho] of the QAP vectors, transforms them with M = m / N-point NTTs and meets the other ranks in two
    all-to-alls of 3 * 32 * M bytes (csrc/ntt.hip: h_poly_dist_launch; restated on integers in
    oracle/pyref/hdist.py); it ends with h[rho + N j], the scalars of ITS h bases h_query[rho + N j];
  * the five partial MSMs follow, and ONE all-gather moves N records of 768 bytes (A, B1, L, H, s*A, r*B1 in G1 and
    B in G2; the serial scalar multiples are taken BEFORE the exchange, by linearity) -- RCCL has no user-defined
    reduction, so the "all-reduce of bucket sums" is all-gather + local add; every rank assembles (A, B, C).

Two drivers of the same protocol:

  NativeProver        one call, `dg16_groth16_prove_dist`: kernels and exchanges are enqueued by libdg16 on its own
                      streams through a dg16_comm (RcclComm = native RCCL send/recv over xGMI; TorchComm =
                      torch.distributed), no host synchronisation between them.  This is what bench.py times.
  DistributedProver   the protocol spelled out in Python over an injected engine and torch.distributed collectives:
                      `GpuEngine` (libdg16 stage by stage) or an oracle-backed engine on CPU
                      (tests/test_parallel_gloo.py, gloo, world size 2).  With a rank count that is not a power of
                      two (or m < N^2) it falls back to the replicated h-polynomial and contiguous h slices.

Neither has a CPU path of its own: the product engines call libdg16 and fail without it.
"""

import numpy as np


def shard_bounds(n, shard, n_shards):
    """Slice [lo, hi) of a length-n range owned by `shard` -- must match PkDev::slice in
    csrc/prover.hip."""
    return n * shard // n_shards, n * (shard + 1) // n_shards


def l_bounds(num_vars, num_inputs, shard, n_shards):
    """Range of l_query owned by `shard`: L rides on the digit sort of A / B1 / B, so its range is the shard's range of
    w[1..] minus the public-input positions (PkDev::l_lo / l_hi in csrc/prover_impl.h)."""
    lo, hi = shard_bounds(num_vars - 1, shard, n_shards)
    k = num_inputs - 1
    return max(lo, k) - k, max(hi, k) - k


def h_is_sharded(m, world):
    """The sharded h-polynomial needs a power-of-two rank count (2, 4, 8) and m >= world^2 -- the condition
    csrc/ntt.hip (h_poly_dist_stage_typed) and DG16_F_H_CYCLIC enforce."""
    return world in (2, 4, 8) and world * world <= m


class GpuEngine:
    """Local compute on one GPU through libdg16 (device tensors via torch, plumbing only)."""

    def __init__(self, ctx, pk, curve):
        import torch
        self.torch = torch
        self.ctx, self.pk, self.curve = ctx, pk, curve
        self.device = torch.device("cuda", ctx.device)
        self.rec = ctx.results_bytes(curve)
        self.log_m = pk.domain_size.bit_length() - 1

    def _bytes(self, n):
        return self.torch.empty(n, dtype=self.torch.uint8, device=self.device)

    def partial(self, a, b, c, w, rs_host, scalars_mont):
        """a, b, c, w: device tensors (whole vectors).  Returns this shard's results record as a uint8 device tensor."""
        out = self._bytes(self.rec)
        self.ctx.groth16_msms_dev(self.pk, a.data_ptr(), b.data_ptr(), c.data_ptr(), w.data_ptr(), rs_host,
                                  out.data_ptr(), scalars_mont=scalars_mont)
        return out

    # sharded h-polynomial, stage by stage (exchange buffers: 3 * m / world elements of 32 bytes)
    def h_stage(self, stage, inputs, rank, world):
        m_rows = self.pk.domain_size // world
        out = self._bytes((m_rows if stage == 2 else 3 * m_rows) * 32)
        self.ctx.h_poly_dist_stage_dev(self.curve, self.log_m, rank, world, stage, [t.data_ptr() for t in inputs],
                                       out.data_ptr())
        return out

    def partial_h(self, h, w, rs_host, scalars_mont):
        out = self._bytes(self.rec)
        self.ctx.groth16_msms_h_dev(self.pk, h.data_ptr(), w.data_ptr(), rs_host, out.data_ptr(),
                                    scalars_mont=scalars_mont)
        return out

    def before_collective(self):
        """The payload is produced on libdg16's stream; torch.distributed orders only against torch's current
        stream -- finish it before handing it over."""
        self.ctx.sync(0)

    def after_collective(self):
        """...and the received data must have landed before libdg16's stream reads it."""
        self.torch.cuda.current_stream(self.device).synchronize()

    def assemble(self, gathered, n_shards, rs_host, scalars_mont):
        nl = 4 if self.curve == "bn254" else 6
        proof = self._bytes(12 * nl * 8)
        self.ctx.groth16_assemble_dev(self.pk, gathered.data_ptr(), n_shards, rs_host, proof.data_ptr(),
                                      scalars_mont=scalars_mont)
        return proof

    def empty_like_bytes(self, t, times=1):
        return self._bytes(t.numel() * times)


class DistributedProver:
    """The protocol in Python; prove() on every rank returns the same (A, B, C) record."""

    def __init__(self, engine, dist=None, rank=0, world=1, sharded_h=False):
        self.engine, self.dist, self.rank, self.world = engine, dist, rank, world
        self.sharded_h = sharded_h and world > 1

    def describe(self):
        return "msm-shard x%d%s + all-gather (torch.distributed, Python-driven)" % (
            self.world, " + sharded h-polynomial (2 all-to-alls)" if self.sharded_h else "")

    def close(self):
        pass

    def _collective(self, fn):
        if hasattr(self.engine, "before_collective"):
            self.engine.before_collective()
        out = fn()
        if hasattr(self.engine, "after_collective"):
            self.engine.after_collective()
        return out

    def _all_to_all(self, send):
        """chunk p of `send` goes to rank p; chunk p of the result came from rank p."""
        recv = self.engine.empty_like_bytes(send)
        if self.dist.get_backend() == "gloo":          # gloo has no all-to-all: gather, keep this rank's column
            n = self.world
            allbuf = self.engine.empty_like_bytes(send, n).cpu()
            self._collective(lambda: self.dist.all_gather_into_tensor(allbuf, send.cpu()))
            recv.copy_(allbuf.view(n, n, -1)[:, self.rank, :].reshape(-1))
        else:
            self._collective(lambda: self.dist.all_to_all_single(recv, send))
        return recv

    def prove(self, a, b, c, w, rs_host, scalars_mont=True):
        """a, b, c: the whole QAP vectors -- or, with sharded_h, this rank's cyclic rows a[world * j + rank]."""
        eng = self.engine
        if self.world == 1:
            return eng.assemble(eng.partial(a, b, c, w, rs_host, scalars_mont), 1, rs_host, scalars_mont)
        if self.sharded_h:
            send1 = eng.h_stage(0, [a, b, c], self.rank, self.world)
            send2 = eng.h_stage(1, [self._all_to_all(send1)], self.rank, self.world)
            h = eng.h_stage(2, [self._all_to_all(send2)], self.rank, self.world)
            part = eng.partial_h(h, w, rs_host, scalars_mont)
        else:
            part = eng.partial(a, b, c, w, rs_host, scalars_mont)
        gathered = eng.empty_like_bytes(part, self.world)
        if self.dist.get_backend() == "gloo" and gathered.is_cuda:
            host = gathered.cpu()
            self._collective(lambda: self.dist.all_gather_into_tensor(host, part.cpu()))
            gathered.copy_(host)
        else:
            # N records of 768 B (BN254)
            self._collective(lambda: self.dist.all_gather_into_tensor(gathered, part))
        return eng.assemble(gathered, self.world, rs_host, scalars_mont)


class NativeProver:
    """dg16_groth16_prove_dist: the whole distributed proof enqueued by libdg16 (kernels and exchanges), through
    `comm` (lib.RcclComm or lib.TorchComm)."""

    def __init__(self, ctx, pk, curve, comm, rank=0, world=1):
        import torch
        self.torch, self.ctx, self.pk, self.curve, self.comm = torch, ctx, pk, curve, comm
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", ctx.device)
        self.nl = 4 if curve == "bn254" else 6
        # a queue of proofs: DG16_F_OVERLAP_TAIL (the proof is then complete on channel 2's stream) -- one GPU, and since
        # round 6 the sharded proof too (H's reduction, the all-gather and the assembly under the next proof's first stage)
        self.overlap_tail = False

    def describe(self):
        if self.world == 1:
            return "single GPU"
        return "msm-shard x%d + sharded h-polynomial (2 all-to-alls) + all-gather of the records, %s" % (
            self.world, self.comm.describe())

    def rccl_ranks(self):
        """Rank count the native RCCL communicator reports; None when another transport carries the exchanges."""
        return self.comm.ranks()[0] if hasattr(self.comm, "ranks") else None

    def close(self):
        if self.comm is not None:
            self.comm.close()

    def prove(self, a, b, c, w, rs_host, scalars_mont=True):
        proof = self.torch.empty(12 * self.nl * 8, dtype=self.torch.uint8, device=self.device)
        if self.world == 1 and self.overlap_tail:
            self.ctx.prove_dev(self.pk, a.data_ptr(), b.data_ptr(), c.data_ptr(), w.data_ptr(), rs_host, proof.data_ptr(),
                               scalars_mont=scalars_mont, overlap_tail=True)
            return proof
        self.ctx.prove_dist_dev(self.pk, self.comm, a.data_ptr(), b.data_ptr(), c.data_ptr(), w.data_ptr(), rs_host,
                                proof.data_ptr(), scalars_mont=scalars_mont, overlap_tail=self.overlap_tail)
        return proof


def rccl_precondition(ctx, rank):
    """(host, device identity, error or None) of this rank: what the ranks compare before they form a communicator."""
    import socket
    from . import lib
    err = None
    try:
        lib.rccl_unique_id()          # binds librccl in this process (the id itself is discarded)
    except lib.Dg16Error as e:
        err = "rank %d: %s" % (rank, e)
    ident = ctx.device
    try:
        import torch
        p = torch.cuda.get_device_properties(ctx.device)
        ident = str(getattr(p, "uuid", None) or (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", ctx.device),
                                                 getattr(p, "pci_device_id", 0)))
    except Exception:                 # no torch device view (CPU tests): the index is all there is
        pass
    return socket.gethostname(), ident, err


def make_prover(ctx, pk, curve, dist, rank, world, transport="rccl"):
    """The prover bench.py and the tools drive.  N = 1 is the plain resident-key prover behind the same interface.
    transport: "rccl" (native communicator; the unique id travels through torch.distributed's object broadcast),
    "torch" (TorchComm under the native pipeline) or "python" (DistributedProver + GpuEngine).  A rank count the
    sharded h-polynomial does not take (not 2 / 4 / 8, or m < N^2) is served by the Python driver with the
    replicated h-polynomial."""
    from . import lib
    if world == 1:
        return NativeProver(ctx, pk, curve, None, 0, 1)
    if not h_is_sharded(pk.domain_size, world) or transport == "python":
        return DistributedProver(GpuEngine(ctx, pk, curve), dist, rank, world,
                                 sharded_h=h_is_sharded(pk.domain_size, world))
    import torch
    comm = None
    if transport == "rccl":
        # The choice of transport is COLLECTIVE: every rank issues the same broadcast and the same all-reduce whatever
        # happens locally, and all ranks fall back together.  (Rank 0 raising before the broadcast while the others
        # wait in it, or one rank failing ncclCommInitRank alone, would leave the ranks issuing different collectives.)
        box = [None]
        if rank == 0:
            try:
                box[0] = lib.rccl_unique_id()
            except lib.Dg16Error as e:
                box[0] = "error: %s" % e
        dist.broadcast_object_list(box, src=0)
        why = box[0] if isinstance(box[0], str) else None
        # Preconditions are checked TOGETHER before anyone enters ncclCommInitRank: that call blocks until every rank has
        # joined, so a rank that failed it alone (librccl missing there, two ranks on one device) would leave its peers
        # inside it while it moved on to the all-reduce below.  Failures that only show inside ncclCommInitRank on some
        # ranks (a link going down mid-join) are not recoverable here: the join itself has no timeout.
        pre = [None] * world
        dist.all_gather_object(pre, rccl_precondition(ctx, rank))
        if why is None:
            bad = [p[2] for p in pre if p[2]]
            ids = [p[:2] for p in pre]
            if bad:
                why = bad[0]
            elif len(set(ids)) != world:
                why = "two ranks share one device (%s): RCCL needs one GPU per rank" % (ids,)
        if why is None:
            try:
                comm = lib.RcclComm(ctx, box[0], world, rank)
            except lib.Dg16Error as e:
                why = "rank %d: %s" % (rank, e)
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32,
                          device="cpu" if dist.get_backend() == "gloo" else torch.device("cuda", ctx.device))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm is not None:
                comm.close()
                comm = None
            import sys
            print("dg16: native RCCL transport unavailable (%s); every rank uses torch.distributed" %
                  (why or "a peer failed to join"), file=sys.stderr)
    if comm is None:
        comm = lib.TorchComm(dist, torch.device("cuda", ctx.device), world, rank)
    return NativeProver(ctx, pk, curve, comm, rank, world)
