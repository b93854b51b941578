"""One PROCESS per party: ext_wit::h, prove::A, prove::B and prove::C (groth16/examples/sha256.rs:26-95 -- `dsha256`)
over a multi-process MpcNet whose channels complete in an adversarial order, against the oracle.

The parties share the one GPU of the test box and exchange over torch.distributed/gloo through lib.TorchNet -- one
process group per MultiplexedStreamID (mpc-net/src/lib.rs:29-33).  Every exchange is preceded by a delay whose
per-channel ordering differs from party to party, and prove::C's three d_msm (prove.rs:113-125) are issued from three
host threads by dg16_prove_c, so the three channels of a party finish in an order no two parties share.  All three
payloads are one Jacobian G1 point of the same size: a transport that matched them by issue order would swap w, u and h
silently -- and h is the term that gets multiplied by r (prove.rs:132) -- so the check below is on the values:
C == msm(W, ax) + msm(U, h) + s A + r M + r msm(H, a) with the clear MSMs of the C oracle.

With DG16_MPC_TRANSPORT=rccl and one GPU per process the same script drives the native three-communicator transport
(dg16_rccl_net); on one GPU RCCL refuses to form, hence gloo here.

Launch: python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 --master-port P \
            tests/mpc_rank_check.py [l] [log_m] [n_points] [serial]
(test infrastructure: imports the oracle)"""
import os
import random
import sys
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dg16_amd  # noqa: E402
from dg16_amd import dist as D, groth16_mpc as M, lib  # noqa: E402
from oracle import corc  # noqa: E402

l = int(sys.argv[1]) if len(sys.argv) > 1 else 1
log_m = int(sys.argv[2]) if len(sys.argv) > 2 else 6
npts = int(sys.argv[3]) if len(sys.argv) > 3 else 50
serial = len(sys.argv) > 4 and sys.argv[4] == "serial"
curve = os.environ.get("DG16_MPC_CURVE", "bn254")
transport = os.environ.get("DG16_MPC_TRANSPORT", "torch")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert world == 4 * l, "PackedSharingParams::new(l) has n = 4 l parties (secret-sharing/src/pss.rs:34-38)"
dist.init_process_group(backend=os.environ.get("DG16_DIST_BACKEND", "gloo"))
one_gpu = transport != "rccl"
device_index = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", rank))
torch.cuda.set_device(device_index)
dev = torch.device("cuda", device_index)
ctx = dg16_amd.Context(device_index)
pp = D.PackedSharingParams(ctx, curve, l)

rng = random.Random(4242 + 17 * rank)
lock = threading.Lock()
trace = []


def before(channel, op):
    with lock:
        d = rng.uniform(0.0, 0.01) + 0.025 * ((channel + rank) % 3)      # this party's slow channel: (2 - rank) % 3
        trace.append(channel)
    time.sleep(d)


if transport == "rccl":
    box = [lib.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    rc = lib.RcclComm(ctx, box[0], world, rank)
    net_ptr, net = rc.net_ptr, rc
else:
    net = lib.TorchNet(dist, dev, world, rank, before=before)
    net_ptr = net.net_ptr
    # the channel watchdog on a GPU net: device payloads, one stream per channel (lib.probe_channels; the CPU form runs in
    # tests/test_parallel_gloo.py) -- this transport's channels are independent groups, so every party must get "joined"
    probed = lib.probe_channels(net.struct, world, rank, dist, soft_s=20.0, hard_s=60.0, device=dev)
    assert probed == "joined", probed

# the statement: the same on every party (seeded), each keeps its own shares
m = 1 << log_m
fr = lambda seed, n: corc.rand_field(curve, "fr", seed, n)      # noqa: E731  (Montgomery form)
a_v, b_v, c_v = (fr(300 + i, m) for i in range(3))
qap = M.qap_pss(pp, a_v, b_v, c_v)[rank]
wa, wax = fr(310, npts), fr(311, npts - 1)                        # a = w[1..], ax = w[2..]-like vectors
S, H, W = (corc.gen_points(curve, 1, 320 + i, n) for i, n in enumerate((npts, npts, npts - 1)))
U = corc.gen_points(curve, 1, 323, m)
V = corc.gen_points(curve, 2, 324, npts)
Lp, Np, Mp = (corc.gen_points(curve, 1, 330 + i, 1) for i in range(3))
Zp, Kp = (corc.gen_points(curve, 2, 335 + i, 1) for i in range(2))
r_i, s_i = 0x1234567 + 5, 0x7654321 + 9
r, s = (corc.field_op(curve, "fr", "to_mont", corc.ints_to_arr([x], 4)) for x in (r_i, s_i))
pk = lambda g, pts: M._packexp_chunks(pp, g, pts)[rank]           # noqa: E731
crs = dict(s=pk(1, S), h=pk(1, H), w=pk(1, W), u=pk(1, U), v=pk(2, V))
a_sh, ax_sh = M.pack_from_witness(pp, wa)[rank], M.pack_from_witness(pp, wax)[rank]

h_share = D.ext_wit_h(ctx, pp, net_ptr, qap[0], qap[1], qap[2], log_m)
pi_a = M.A(Lp, Np, r, pp, crs["s"], a_sh).compute(ctx, net_ptr, 1)          # sid = 1: not only channel 0
pi_b = M.B(Zp, Kp, s, pp, crs["v"], a_sh).compute(ctx, net_ptr, 2)
cc = M.C(pi_a, Mp, s, r, pp, crs["w"], crs["u"], crs["h"], a_sh, ax_sh, h_share)
if serial:      # DG16_F_SERIAL_CHANNELS (16): the three d_msm in the order 0, 1, 2 on the calling thread
    nl = lib.FQ_LIMBS64[curve] * 2
    out = np.zeros((1, 3 * nl // 2), dtype=np.uint64)
    P = lib._ptr
    sh = lambda v, cols: np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, cols)   # noqa: E731
    Wv, Uv, Hv = (sh(crs[k], nl) for k in ("w", "u", "h"))
    av, axv, hv = (sh(v, 4) for v in (a_sh, ax_sh, h_share))
    Aj = np.ascontiguousarray(pi_a, dtype=np.uint64).reshape(1, -1)
    ctx._chk(ctx.L.dg16_prove_c(ctx.h, pp.h, net_ptr, P(Aj), P(Mp), P(s), P(r), P(Wv), P(axv), Wv.shape[0],
                                axv.shape[0], P(Uv), P(hv), Uv.shape[0], hv.shape[0], P(Hv), P(av), Hv.shape[0],
                                av.shape[0], 1 | 16, P(out)))
    pi_c = out
else:
    pi_c = cc.compute(ctx, net_ptr)

# every party holds the same three points; the king checks them against the oracle
mine = [np.asarray(x).tobytes() for x in (pi_a, pi_b, pi_c)] + [np.asarray(h_share).tobytes()]
everyone = [None] * world
dist.all_gather_object(everyone, mine)
same = all(e[:3] == everyone[0][:3] for e in everyone)
ok = same
if rank == 0:
    add = lambda g, p, q: corc.point_add(curve, g, p, q)            # noqa: E731
    mul = lambda g, p, k: corc.point_mul(curve, g, p, k)            # noqa: E731
    msm = lambda g, b, sc: corc.msm(curve, g, b, sc, scalars_mont=True)   # noqa: E731
    aff = lambda g, j: corc.jac_to_affine(curve, g, j)              # noqa: E731
    h_shares = np.stack([np.frombuffer(e[3], dtype=np.uint64).reshape(-1, 4) for e in everyone], axis=1)
    h_clear = pp.unpack(h_shares).reshape(m, 4)
    # every party's h SHARES == the line-by-line restatement of ext_wit::h (oracle/pyref/groth16.py:216-236) ...
    from oracle.pyref.fields import FR
    from oracle.pyref.pss import PackedSharingParams as RefPSS
    from oracle.pyref.poly import Domain
    from oracle.pyref import groth16 as G
    F = FR[curve]
    ints = lambda arr: [F.from_mont(v) for v in corc.arr_to_ints(np.asarray(arr).reshape(-1, 4))]   # noqa: E731
    ref = RefPSS(F, l)
    exp_sh = G.ext_wit_h(G.qap_pss(ints(a_v), ints(b_v), ints(c_v), ref), Domain(F, m), ref)
    ok &= all(ints(np.frombuffer(e[3], dtype=np.uint64)) == exp_sh[i] for i, e in enumerate(everyone))
    # ... and, for l = 2 (t = 1), the h-polynomial itself: `s1.swap(i, i * l + t)` (ext_wit.rs:74-76) picks the odd
    # positions of the 2m evaluations -- the coset -- only then; with l = 1 the swap is the identity and the protocol's
    # output is the product on the first m points of the 2m domain, on the reference as on this library
    if l == 2:
        ok &= np.array_equal(h_clear, corc.h_poly(curve, a_v.copy(), b_v.copy(), c_v.copy()))
    eA = add(1, add(1, Lp, mul(1, Np, r_i)), msm(1, S, wa))
    eB = add(2, add(2, Zp, mul(2, Kp, s_i)), msm(2, V, wa))
    eC = add(1, msm(1, W, wax), msm(1, U, h_clear))
    eC = add(1, add(1, eC, mul(1, eA, s_i)), mul(1, Mp, r_i))
    eC = add(1, eC, mul(1, msm(1, H, wa), r_i))
    ok &= np.array_equal(aff(1, pi_a), eA) and np.array_equal(aff(2, pi_b), eB) and np.array_equal(aff(1, pi_c), eC)
flag = torch.tensor([1 if ok else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
# how the three channels of prove::C finished on each party (the last three exchanges per channel are its scatters)
orders = [None] * world
dist.all_gather_object(orders, "".join(str(c) for c in trace[-6:]))
if rank == 0:
    print("MPC_RANK_CHECK", "PASS" if int(flag.item()) == 1 else "FAIL", "parties", world, "l", l, "log_m", log_m,
          "| transport:", net.describe(), "| serial" if serial else "| joined", "| channel order of prove::C's exchanges per party:",
          orders, "| errors:", getattr(net, "errors", []))
dist.destroy_process_group()
sys.exit(0 if int(flag.item()) == 1 else 1)
