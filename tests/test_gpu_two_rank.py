"""GPU: the N > 1 path with the real engine.  Two processes share the one GPU of the test box, each owns a
shard key, computes its records with libdg16, exchanges them through torch.distributed (gloo: RCCL refuses
two ranks on one device) and assembles; every rank's proof must equal the unsharded proof.  Everything of
bench.py's multi-GPU path except the RCCL transport itself is exercised."""

import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# (2, 4 ranks: sharded h-polynomial, native pipeline under a gloo-backed dg16_comm and the Python-driven protocol;
#  3 ranks: replicated h-polynomial, contiguous slices)
@pytest.mark.parametrize("world,log_m,transport", [(2, 12, "torch"), (4, 12, "torch"), (2, 10, "python"),
                                                   (3, 14, "python")])
def test_sharded_proof_across_processes(world, log_m, transport):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tools", "two_rank_check.py"), str(log_m), transport]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "TWO_RANK_CHECK PASS" in out.stdout


def test_bench_flow_with_two_ranks_on_one_device():
    """bench.py's own N > 1 flow (sharded workload, native distributed prove, barriers, max over ranks, the live parity
    gate on rank 0) with both ranks on cuda:0 and a gloo process group (DG16_BENCH_SINGLE_DEVICE: RCCL refuses two ranks
    on one device) -- everything the 8-GPU run does except the RCCL wire."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-m", "14",
           "--transport", "torch"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DG16_BENCH_SINGLE_DEVICE="1")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["parity_check"] == "pass" and d["scaling"] == "strong"
    assert "sharded h-polynomial" in d["config"]["parallelism"]
