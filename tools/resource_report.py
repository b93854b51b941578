"""Parse hipcc's -Rpass-analysis=kernel-resource-usage reports (csrc/*.usage.txt, written by the Makefile) into
{demangled kernel name: {vgprs, agprs, sgprs, scratch, occupancy, lds}}.
usage: python tools/resource_report.py [out.md]"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "distributed-groth16_amd", "csrc")
CXXFILT = "c++filt"
KEYS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
        "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds", "VGPRs Spill": "vgpr_spill"}


def load():
    out = {}
    for path in sorted(glob.glob(os.path.join(CSRC, "*.usage.txt"))):
        cur = None
        for line in open(path, errors="replace"):
            # ("file:line:col: remark: key: value" from a plain compile, "remark: file:line:col: key: value" under -save-temps)
            m = re.search(r"remark:\s+(?:\S+:\d+:\d+:\s+)?(.*?): (\S+) \[-Rpass-analysis", line)
            if not m:
                continue
            key, val = m.group(1).strip(), m.group(2)
            if key == "Function Name":
                cur = out.setdefault(val, {"file": os.path.basename(path)})
            elif cur is not None and key in KEYS:
                cur[KEYS[key]] = int(val)
    if not out:
        return {}
    names = list(out)
    dem = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return {d: out[n] for n, d in zip(names, dem)}


def short(name):
    name = name.replace("dg16::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    return name.replace("Fp<bn254_fq_params>", "bn254_fq").replace("Fp<bn254_fr_params>", "bn254_fr") \
               .replace("Fp<bls12_381_fq_params>", "bls12_381_fq").replace("Fp<bls12_381_fr_params>", "bls12_381_fr") \
               .replace("Fp<bls12_377_fq_params>", "bls12_377_fq").replace("Fp<bls12_377_fr_params>", "bls12_377_fr")


if __name__ == "__main__":
    rep = load()
    lines = ["| kernel | VGPRs | AGPRs | scratch B/lane | LDS B/block | occupancy waves/SIMD |", "|---|---|---|---|---|---|"]
    for name in sorted(rep, key=short):
        r = rep[name]
        if "occupancy" not in r:
            continue
        lines.append("| `%s` | %d | %d | %d | %d | %d |" % (short(name), r.get("vgprs", 0), r.get("agprs", 0),
                                                       r.get("scratch", 0), r.get("lds", 0), r["occupancy"]))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("# hipcc kernel resource usage (gfx950, -Rpass-analysis=kernel-resource-usage), end of round 1\n\n" + text + "\n")
