"""Per-rank work of an N-way distributed proof, timed on ONE GPU: rank 0's shard of the key and rows, with a
loopback dg16_comm that stands in for the other ranks (the all-to-alls and the all-gather become device copies of the
same size, so the RESULT is meaningless -- only the time is read).  This is the compute a rank does between its
exchanges; link time is not in it.  usage: python tools/shard_timing.py [log_m] [steps] [curve] [worlds, e.g. 1,2,4,8]"""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dg16_amd  # noqa: E402
from dg16_amd import lib  # noqa: E402
from dg16_amd.parallel import NativeProver  # noqa: E402
import bench  # noqa: E402

log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
curve = sys.argv[3] if len(sys.argv) > 3 else "bn254"
worlds = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 4, 8]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = dg16_amd.Context(0)


class LoopbackComm(lib.TorchComm):
    def __init__(self, n):
        self.torch, self.device, self.n_ranks, self.rank, self.errors = torch, dev, n, 0, []
        self._cb = (lib._COMM_N(lambda _s: n), lib._COMM_N(lambda _s: 0), lib._COMM_GATHER(self._all_gather),
                    lib._COMM_A2A(self._all_to_all))
        self.struct = lib.CommStruct(None, *self._cb)
        self.comm_ptr = ctypes.cast(ctypes.pointer(self.struct), ctypes.c_void_p)

    def describe(self):
        return "loopback"

    def _copy(self, stream, fn):
        try:
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)):
                fn()
            return 0
        except Exception as e:
            self.errors.append(repr(e))
            return 6

    def _all_gather(self, _s, send, nbytes, recv, stream):
        return self._copy(stream, lambda: self._tensor(recv, nbytes * self.n_ranks).view(self.n_ranks, -1).copy_(
            self._tensor(send, nbytes).view(1, -1).expand(self.n_ranks, -1)))

    def _all_to_all(self, _s, send, recv, per_peer, stream):
        return self._copy(stream, lambda: self._tensor(recv, per_peer * self.n_ranks).copy_(
            self._tensor(send, per_peer * self.n_ranks)))


out = {}
for world in worlds:
    wl = bench.Workload(ctx, dev, log_m, 0, world, curve=curve)
    comm = LoopbackComm(world) if world > 1 else None
    p = NativeProver(ctx, wl.pk, curve, comm, 0, world)
    # DG16_OVERLAP=1: the queue of proofs with DG16_F_OVERLAP_TAIL (one GPU: round 4; the sharded proof: round 6)
    p.overlap_tail = os.environ.get("DG16_OVERLAP") == "1"

    def step():
        wl.qap()
        return p.prove(wl.a, wl.b, wl.c, wl.w, wl.rs, scalars_mont=False)

    def sync():
        for ch in range(3):
            ctx.sync(ch)
        torch.cuda.synchronize()

    for _ in range(5):
        step()
    sync()
    # settle: the previous world's key (gigabytes of tables) was just freed and this one built -- one call of this tool
    # printed 4 ms of host time per proof for whichever world came third or fourth (profiles/r5f_shard_timing.txt,
    # r5g_shard_timing_a.txt), 0.4 ms when the same world ran alone
    import gc
    gc.collect()
    torch.cuda.synchronize()
    time.sleep(0.5)
    for _ in range(2):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    host_ms = (time.perf_counter() - t0) / steps * 1e3      # time the host spends enqueuing one proof
    sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out["x%d" % world] = round(ms, 3)
    print("world %d%s: %.3f ms per proof on rank 0, host enqueue %.3f ms (n_ab %d, n_h %d)%s" % (
        world, " overlap" if p.overlap_tail else "", ms, host_ms, wl.pk.info()["n_ab"], wl.pk.info()["n_h"], "  comm errors: %s" % comm.errors if comm and comm.errors else ""),
        flush=True)
    wl.pk.close()
    del wl
    torch.cuda.empty_cache()
print(json.dumps({"per_rank_ms": out, "log_m": log_m, "curve": curve}))
