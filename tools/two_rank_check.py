"""Two (or N) processes sharing ONE GPU prove one statement -- through the native pipeline
(dg16_groth16_prove_dist under a torch.distributed-backed dg16_comm) or the Python-driven protocol
(DistributedProver + GpuEngine) -- exchanging over gloo (RCCL refuses two ranks on one device), and compare with
the unsharded proof.  This is the multi-GPU path of bench.py with everything but the RCCL transport real: sharded
h-polynomial (cyclic rows, two all-to-alls), cyclic h bases, MSM slices, all-gather, assembly.
Launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
            tools/two_rank_check.py [log_m] [torch|python]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dg16_amd  # noqa: E402
from dg16_amd.parallel import make_prover  # noqa: E402
import bench  # noqa: E402

log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 12
transport = sys.argv[2] if len(sys.argv) > 2 else "torch"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend=os.environ.get("DG16_DIST_BACKEND", "gloo"))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = dg16_amd.Context(0)
wl = bench.Workload(ctx, dev, log_m, rank, world)          # same seed on every rank -> same statement
prover = make_prover(ctx, wl.pk, bench.CURVE, dist, rank, world, transport=transport)
proofs = []
for _ in range(3):                                          # repeated: buffers warm, events reused
    wl.qap()
    proof = prover.prove(wl.a, wl.b, wl.c, wl.w, wl.rs, scalars_mont=False)
    for ch in range(3):
        ctx.sync(ch)
    proofs.append(proof.cpu().numpy().copy())
# unsharded proof of the same statement on this rank
one = bench.Workload(ctx, dev, log_m, 0, 1)
ref = torch.empty_like(proof)
ctx.prove_dev(one.pk, one.a.data_ptr(), one.b.data_ptr(), one.c.data_ptr(), one.w.data_ptr(), one.rs, ref.data_ptr(),
              scalars_mont=False)
for ch in range(3):
    ctx.sync(ch)
ref = ref.cpu().numpy()


def affine(p):
    """Jacobian records differ by representation between shard counts; compare in affine via the C ABI."""
    g1 = ctx.to_affine(bench.CURVE, 1, np.concatenate([p[:96], p[288:384]]).view(np.uint64).reshape(2, 12))
    g2 = ctx.to_affine(bench.CURVE, 2, p[96:288].view(np.uint64).reshape(1, 24))
    return np.concatenate([np.asarray(g1).reshape(-1), np.asarray(g2).reshape(-1)])


ok = all(np.array_equal(affine(p), affine(ref)) for p in proofs)
# Round 6: a QUEUE of sharded proofs with DG16_F_OVERLAP_TAIL (H's reduction, the all-gather and the assembly of proof k on
# channel 2's stream under the first stage of proof k + 1), different (r, s) per proof so that a record or a gathered buffer
# read too late shows: every queued proof == the same statement proved alone with a synchronisation behind it
if transport == "torch" and hasattr(prover, "overlap_tail"):
    rss = [np.array([[7 + k, 11, 13 * k + 1, 1], [5, 9 + k, 2, 3 + k]], dtype=np.uint64) for k in range(4)]
    alone = []
    for rs in rss:
        wl.qap()
        pr = prover.prove(wl.a, wl.b, wl.c, wl.w, rs, scalars_mont=False)
        for ch in range(3):
            ctx.sync(ch)
        alone.append(pr.cpu().numpy().copy())
    prover.overlap_tail = True
    queued = []
    for rs in rss:
        wl.qap()
        queued.append(prover.prove(wl.a, wl.b, wl.c, wl.w, rs, scalars_mont=False))
    for ch in range(3):
        ctx.sync(ch)
    prover.overlap_tail = False
    ok = ok and all(np.array_equal(affine(q.cpu().numpy()), affine(a_)) for q, a_ in zip(queued, alone))
    ok = ok and not np.array_equal(affine(alone[0]), affine(alone[1]))
flag = torch.tensor([1 if ok else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print("TWO_RANK_CHECK", "PASS" if int(flag.item()) == 1 else "FAIL", "world", world, "log_m", log_m, "|",
          prover.describe(), "| comm errors:", getattr(getattr(prover, "comm", None), "errors", []))
dist.destroy_process_group()
sys.exit(0 if int(flag.item()) == 1 else 1)
