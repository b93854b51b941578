#!/usr/bin/env python3
"""What the gfx950 code of a built object looks like, block by block -- without a GPU and without recompiling:
the device bundle of a csrc/*.o is extracted (llvm-objdump --offloading, in a temporary directory), disassembled, and
every kernel is cut into basic blocks at its branches and branch targets.  Per block: instructions, v_mad_u64_u32,
scratch / LDS / global accesses, s_nop.  A spill INSIDE a hot block (the G2 bucket kernel once moved 31.7 GB per launch
that way) shows here and nowhere in hipcc's per-kernel resource report, which only gives the scratch size.

usage: python tools/isa_report.py distributed-groth16_amd/csrc/msm_bn254_g1.o [kernel-name substring] [min block size]
import: kernels(path) -> {mangled name: [block dict]}   (tests/test_kernel_isa.py pins the hot blocks)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BRANCH = re.compile(r"^(s_cbranch_\w+|s_branch)\s+(\d+)")
ADDR = re.compile(r"//\s*([0-9A-Fa-f]+):")


def disassemble(obj_path):
    """-> text of llvm-objdump -d over the gfx950 bundle of a host object (or over a code object given directly)."""
    tmp = tempfile.mkdtemp(prefix="isa_report_")
    try:
        local = os.path.join(tmp, os.path.basename(obj_path))
        shutil.copy(obj_path, local)
        subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, text=True, check=True)
        bundles = [f for f in os.listdir(tmp) if "amdgcn" in f]
        target = os.path.join(tmp, bundles[0]) if bundles else local
        return subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", target], capture_output=True, text=True,
                              check=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def instructions(obj_path):
    """-> {mangled name: [(address, instruction text)]} of every function in the object's gfx950 bundle"""
    text = disassemble(obj_path)
    out, cur = {}, None
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body = line.strip()
        am = ADDR.search(body)
        if am:
            cur.append((int(am.group(1), 16), body.split("//")[0].strip()))
    return {k: v for k, v in out.items() if v}


def kernels(obj_path):
    text = disassemble(obj_path)
    out, cur, name = {}, None, None
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            name = m.group(1)
            cur = out.setdefault(name, [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body = line.strip()
        am = ADDR.search(body)
        if not am:
            continue
        cur.append((int(am.group(1), 16), body.split("//")[0].strip()))
    return {k: blocks(v) for k, v in out.items() if v}


def blocks(ins):
    """ins: [(address, text)] of one function -> list of block dicts in address order."""
    starts = {ins[0][0]}
    for i, (addr, text) in enumerate(ins):
        m = BRANCH.match(text)
        ends_block = bool(m) or text.startswith(("s_setpc_b64", "s_swappc_b64", "s_endpgm"))
        if m:
            off = int(m.group(2))
            if off >= 0x8000:
                off -= 0x10000
            starts.add(addr + 4 + 4 * off)
        if ends_block and i + 1 < len(ins):
            starts.add(ins[i + 1][0])
    res, cur = [], None
    for addr, text in ins:
        if addr in starts or cur is None:
            cur = {"addr": addr, "instr": 0, "mads": 0, "scratch": 0, "lds": 0, "vmem": 0, "nops": 0, "lshl_add_u64": 0,
                   "valu": 0, "salu": 0}
            res.append(cur)
        op = text.split()[0]
        cur["instr"] += 1
        cur["mads"] += op == "v_mad_u64_u32"
        cur["scratch"] += op.startswith("scratch_")
        cur["lds"] += op.startswith("ds_")
        cur["vmem"] += op.startswith(("global_", "buffer_", "flat_"))
        cur["nops"] += op == "s_nop"
        cur["lshl_add_u64"] += op == "v_lshl_add_u64"
        cur["valu"] += op.startswith("v_")
        cur["salu"] += op.startswith("s_")
    return res


def summary(blks):
    keys = ("instr", "mads", "scratch", "lds", "vmem", "nops", "lshl_add_u64", "valu", "salu")
    return {k: sum(b[k] for b in blks) for k in keys}


if __name__ == "__main__":
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    min_size = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    for name, blks in kernels(path).items():
        if pat not in name:
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        s = summary(blks)
        print("%s\n  %d blocks; %s" % (dem[:160], len(blks), ", ".join("%s %d" % kv for kv in s.items())))
        for b in blks:
            if b["instr"] >= min_size or b["scratch"]:
                print("    block @%06x: instr %5d  mads %5d  valu %5d  salu %4d  nops %3d  scratch %3d  lds %3d  vmem %2d"
                      % (b["addr"], b["instr"], b["mads"], b["valu"], b["salu"], b["nops"], b["scratch"], b["lds"],
                         b["vmem"]))
