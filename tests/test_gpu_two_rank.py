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


@pytest.mark.parametrize("transport", ["torch", "rccl"])
def test_bench_flow_with_two_ranks_on_one_device(transport):
    """bench.py's own N > 1 flow launched PLAIN -- `python bench.py --gpus 2`, no torch.distributed.run in front: the
    script re-executes itself as the launcher of two ranks -- with both ranks on cuda:0 and a gloo process group
    (DG16_BENCH_SINGLE_DEVICE: RCCL refuses two ranks on one device): sharded workload, native distributed prove,
    barriers, max over ranks, the live parity gate on rank 0, the replicas figure.  transport = rccl asks for the native
    communicator, which cannot form here: the ranks must notice TOGETHER and all fall back to torch.distributed
    (parallel.make_prover's collective decision) instead of issuing mismatched collectives."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-m", "14",
           "--transport", transport]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DG16_BENCH_SINGLE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["parity_check"] == "pass" and d["scaling"] == "strong"
    assert "sharded h-polynomial" in d["config"]["parallelism"]
    assert "torch.distributed" in d["config"]["parallelism"] and d["config"]["rccl_ranks"] is None
    assert d["replicas"]["scaling"] == "weak" and d["replicas"]["value"] > 0
    assert d["valu_roofline"]["frac"] <= 1.0 and d["valu_roofline"]["whole_proof_valu_frac"] <= 1.0
    if transport == "rccl":
        assert "native RCCL transport unavailable" in out.stderr


@pytest.mark.parametrize("parties,l,mode", [(4, 1, "joined"), (4, 1, "serial"), (8, 2, "joined")])
def test_mpc_parties_as_processes_with_adversarial_channel_order(parties, l, mode):
    """ext_wit::h, prove::A / B / C::compute with ONE PROCESS PER PARTY over a multi-process dg16_net (lib.TorchNet: one
    gloo process group per MultiplexedStreamID) whose channels are delayed in a different order on every party, while
    dg16_prove_c issues its three d_msm from three host threads (prove.rs:113-125): the proof elements equal the
    oracle's.  'serial' = DG16_F_SERIAL_CHANNELS (the three d_msm one after another, for a transport with one pipe)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(parties),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "mpc_rank_check.py"), str(l), "6", "40"] + (["serial"] if mode == "serial" else [])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "MPC_RANK_CHECK PASS" in out.stdout
