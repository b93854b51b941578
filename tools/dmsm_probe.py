"""d_msm rounds of bench.dmsm_sweep (dmsm_bench.rs: 8 parties as host threads on one GPU) for a short budget -- under
rocprofv3 --kernel-trace the dispatch timeline of the last round shows how the parties' MSMs share the device.
usage: python tools/dmsm_probe.py [budget_s]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
res = bench.dmsm_sweep(dev, budget_s=float(sys.argv[1]) if len(sys.argv) > 1 else 4.0)
for r in res["rows"]:
    print(r["log_domain"], round(r["round_ms"], 2), r["parity"])
