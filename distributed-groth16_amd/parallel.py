"""One-process-per-GPU Groth16 proving: the king/client split of the reference's mpc-net
(gather to king -> combine -> scatter, dist-primitives/src/dmsm/mod.rs:88-97, mpc-net/src/lib.rs:61-140)
re-mapped to the GPUs of one MI355X node.

Every rank holds a contiguous 1/N slice of each MSM's bases (`Context.pk_create(..., shard, n_shards)`),
computes the h-polynomial (replicated: 6 NTTs, ~5 % of a proof) and its five partial MSMs, then ONE
collective moves N x 768 bytes: an all-gather of the per-rank records (A, B1, L, H, s*A, r*B1 in G1 and B in G2;
the serial scalar multiples s*A, r*B1 are taken BEFORE the exchange, by linearity) (RCCL has no user-defined
reduction, so an elliptic-curve "all-reduce" is all-gather + local add), after which every rank adds
the N partial points per MSM and assembles (A, B, C).  The exchange is latency-bound (microseconds over
xGMI); link bandwidth is irrelevant at this size, so there is exactly one collective per proof.

The local engine and the collective are injected so that the sharding / exchange logic can be tested
on CPU with the gloo backend and an oracle-backed engine (tests/test_parallel_gloo.py); the product
engine is `GpuEngine` (libdg16 on device tensors) and has no CPU path.
"""

import numpy as np


def shard_bounds(n, shard, n_shards):
    """Slice [lo, hi) of a length-n range owned by `shard` -- must match PkDev::slice in
    csrc/prover.hip."""
    return n * shard // n_shards, n * (shard + 1) // n_shards


class GpuEngine:
    """Local compute on one GPU through libdg16 (device tensors via torch, plumbing only)."""

    def __init__(self, ctx, pk, curve):
        import torch
        self.torch = torch
        self.ctx, self.pk, self.curve = ctx, pk, curve
        self.device = torch.device("cuda", ctx.device)
        self.rec = ctx.results_bytes(curve)

    def partial(self, a, b, c, w, rs_host, scalars_mont):
        """a, b, c, w: device tensors.  Returns this shard's results record as a uint8 device tensor."""
        out = self.torch.empty(self.rec, dtype=self.torch.uint8, device=self.device)
        self.ctx.groth16_msms_dev(self.pk, a.data_ptr(), b.data_ptr(), c.data_ptr(), w.data_ptr(), rs_host,
                                  out.data_ptr(), scalars_mont=scalars_mont)
        return out

    def before_collective(self):
        """The record is produced on libdg16's stream; RCCL (torch.distributed) orders only against torch's
        current stream -- finish the record before handing it over (one host sync, ~10 us)."""
        self.ctx.sync(0)

    def after_collective(self):
        """...and the gathered records must have landed before libdg16's stream reads them."""
        self.torch.cuda.current_stream(self.device).synchronize()

    def assemble(self, gathered, n_shards, rs_host, scalars_mont):
        nl = 4 if self.curve == "bn254" else 6
        proof = self.torch.empty(12 * nl * 8, dtype=self.torch.uint8, device=self.device)
        self.ctx.groth16_assemble_dev(self.pk, gathered.data_ptr(), n_shards, rs_host, proof.data_ptr(),
                                      scalars_mont=scalars_mont)
        return proof

    def empty_gather(self, n_shards):
        return self.torch.empty(n_shards * self.rec, dtype=self.torch.uint8, device=self.device)


class DistributedProver:
    """prove() on every rank returns the same (A, B, C) record."""

    def __init__(self, engine, dist=None, rank=0, world=1):
        self.engine, self.dist, self.rank, self.world = engine, dist, rank, world

    def describe(self):
        return "msm-shard x%d + all-gather (torch.distributed)" % self.world

    def close(self):
        pass

    def prove(self, a, b, c, w, rs_host, scalars_mont=True):
        part = self.engine.partial(a, b, c, w, rs_host, scalars_mont)
        if self.world == 1:
            gathered = part
        else:
            gathered = self.engine.empty_gather(self.world)
            if hasattr(self.engine, "before_collective"):
                self.engine.before_collective()
            # the only data-path collective of a proof: N records of 768 B (BN254) over RCCL / xGMI
            self.dist.all_gather_into_tensor(gathered, part)
            if hasattr(self.engine, "after_collective"):
                self.engine.after_collective()
        return self.engine.assemble(gathered, self.world, rs_host, scalars_mont)


def make_prover(ctx, pk, curve, dist, rank, world):
    """The prover bench.py and the tools drive: N = 1 is the plain resident-key prover behind the same interface."""
    return DistributedProver(GpuEngine(ctx, pk, curve), dist, rank, world)
