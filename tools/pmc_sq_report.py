"""SQ counter summary per kernel from one or more `rocprofv3 --kernel-trace --pmc ... --output-format csv` passes
(each pass = its own directory; counters of different passes are joined by kernel name, averages per launch).
usage: python tools/pmc_sq_report.py out.md "<title / command>" pass_dir [pass_dir ...]

Columns (shares of the launch's wave-cycles, SQ counts quad-cycles):
  wait_any     parked on s_waitcnt / barrier          (SQ_WAIT_ANY / SQ_WAVE_CYCLES)
  wait_inst    ready but not issued: pipe busy / hazard (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)
  active_valu  issuing VALU                             (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)
  resident     average waves per SIMD = SQ_WAVE_CYCLES x 4 / (duration x CLK x 1024 SIMDs)
  valu_busy    fraction of SIMD cycles issuing VALU = SQ_ACTIVE_INST_VALU x 4 / (duration x CLK x 1024)
  valu/wave-cy VALU wave-instructions per wave quad-cycle (SQ_INSTS_VALU / SQ_WAVE_CYCLES)
CLK = 2.4 GHz (the guide's peak clock; profiled passes run somewhat lower, so `resident` and `valu_busy` are lower
bounds)."""
import csv
import glob
import os
import sys
from collections import defaultdict

CLK = 2.4e9
SIMDS = 1024


def short(name):
    n = name.split("(")[0]
    for a, b in (("dg16::", ""), ("void ", "")):
        n = n.replace(a, b)
    # keep the template argument that tells the groups apart
    if "<" in name:
        arg = name[name.index("<") + 1:]
        tag = ""
        if "Fp2" in arg.split(",")[0]:
            tag = " (G2)"
        for c in ("bn254", "bls12_381", "bls12_377"):
            if c in arg.split(">")[0] or c in arg[:80]:
                tag = " " + c + tag
                break
        n = n.split("<")[0] + tag
    return n


def main():
    out, title, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    ctr = defaultdict(lambda: defaultdict(float))      # kernel -> counter -> sum
    launches = defaultdict(lambda: defaultdict(set))   # kernel -> counter -> dispatch ids
    dur = defaultdict(list)                            # kernel -> durations (ns) under PMC
    meta = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                c = r["Counter_Name"]
                ctr[k][c] += float(r["Counter_Value"])
                launches[k][c].add(r["Dispatch_Id"])
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                    meta[k] = (r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")), r.get("LDS_Block_Size", ""),
                               r.get("Scratch_Size", ""))
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if "Start_Timestamp" in r:
                    dur[k + "@trace"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    for k in ctr:
        def avg(c):
            n = len(launches[k][c])
            return ctr[k][c] / n if n else None
        wc = avg("SQ_WAVE_CYCLES")
        if not wc:
            continue
        ds = dur.get(k) or dur.get(k + "@trace") or []
        us = sum(ds) / len(ds) / 1e3 if ds else None
        n_l = max(len(v) for v in launches[k].values())
        def share(c):
            v = avg(c)
            return "%.1f%%" % (100.0 * v / wc) if v is not None else "-"
        av = avg("SQ_ACTIVE_INST_VALU")
        iv = avg("SQ_INSTS_VALU")
        im = avg("SQ_INSTS_VMEM")
        busy = avg("SQ_BUSY_CYCLES")
        res = wc * 4 / (us * 1e-6 * CLK * SIMDS) if us else None
        vb = av * 4 / (us * 1e-6 * CLK * SIMDS) if (us and av is not None) else None
        extra = []
        for c in sorted(ctr[k]):
            if c not in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY",
                         "SQ_INSTS_VALU", "SQ_INSTS_VMEM", "SQ_BUSY_CYCLES", "SQ_WAVES"):
                extra.append("%s %.3g" % (c, avg(c)))
        rows.append((-(us or 0) * n_l, "| `%s` | %d | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            k, n_l, "%.1f" % us if us else "-", meta.get(k, ("", "", ""))[0], meta.get(k, ("", "", ""))[1],
            meta.get(k, ("", "", ""))[2], "%.3g" % avg("SQ_WAVES") if avg("SQ_WAVES") else "-", share("SQ_WAIT_ANY"),
            share("SQ_WAIT_INST_ANY"), share("SQ_ACTIVE_INST_ANY"), share("SQ_ACTIVE_INST_VALU"),
            "%.3g" % iv if iv else "-", "%.3g" % im if im else "-",
            "%.2f" % res if res else "-", "%.0f%%" % (100 * vb) if vb is not None else "-",
            "; ".join(extra) + ((" SQ_BUSY_CYCLES %.3g" % busy) if busy else ""))))
    rows.sort()
    with open(out, "w") as f:
        f.write("# SQ counters per kernel -- %s\n\n" % title)
        f.write(__doc__.split("Columns", 1)[1].join(["Columns", ""]) + "\n")
        f.write("| kernel | launches | avg us (under PMC) | VGPRs | LDS B | scratch B | waves | wait_any | wait_inst | "
                "active_any | active_valu | VALU wave-instr | VMEM wave-instr | resident waves/SIMD | valu_busy | other |\n")
        f.write("|" + "---|" * 16 + "\n")
        for _, r in rows:
            f.write(r + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
