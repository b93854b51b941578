#!/usr/bin/env python3
"""Build-time check of the accumulation-register file of msm_impl.h (acc_set / acc_get: field elements parked in
a[144 .. 255] by asm statements the compiler cannot see into).  Reads the assembly hipcc emits for the device
(-save-temps: inline asm statements are bracketed by ;;#ASMSTART / ;;#ASMEND there -- a disassembly cannot tell the
compiler's own v_accvgpr_write from acc_set's) and fails if, in any kernel whose asm statements touch the file, an
instruction OF THE COMPILER'S OWN names an accumulation register inside it.  That collision was round 5's
HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.  csrc/Makefile runs this on every msm_group object.

usage: check_agpr_file.py file.s [report.txt]      exit status 1 on a collision
(report: one line per kernel that names accumulation registers -- kernel, asm range, highest register of the compiler's
own; tests/test_kernel_isa.py reads it)
import: collisions(path, base) -> {kernel: sorted registers}, usage(path) -> {kernel: (asm regs, compiler regs)}"""
import re
import sys

_REG = re.compile(r"\ba(\d+)\b")
_IDX = re.compile(r"\ba\[(0x[0-9a-fA-F]+|\d+)\]")
_RANGE = re.compile(r"\ba\[(\d+):(\d+)\]")


def usage(path):
    out, fn, in_asm = {}, None, False
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            fn = m.group(1)
            continue
        if ";;#ASMSTART" in line:
            in_asm = True
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            continue
        t = line.strip()
        if fn is None or not t or t[0] in ";.":
            continue
        t = t.split(";")[0]
        regs = [int(x) for x in _REG.findall(t)] + [int(x, 0) for x in _IDX.findall(t)]
        for lo, hi in _RANGE.findall(t):
            regs += list(range(int(lo), int(hi) + 1))
        if regs:
            d = out.setdefault(fn, (set(), set()))
            d[0 if in_asm else 1].update(regs)
    return out


def collisions(path, base=144):
    bad = {}
    for fn, (asm, cc) in usage(path).items():
        if any(r >= base for r in asm):
            hit = sorted(r for r in cc if r >= base)
            if hit:
                bad[fn] = hit
    return bad


if __name__ == "__main__":
    base = 144
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            for fn, (asm, cc) in sorted(usage(sys.argv[1]).items()):
                f.write("%s asm %s compiler_max %s\n" % (fn, ("%d-%d" % (min(asm), max(asm))) if asm else "-",
                                                       max(cc) if cc else "-"))
    bad = collisions(sys.argv[1], base)
    for fn, regs in bad.items():
        print("check_agpr_file: %s: the compiler allocated a[%d..%d] (%d registers) inside the asm statements' file a[%d..255]"
              % (fn, regs[0], regs[-1], len(regs), base), file=sys.stderr)
    sys.exit(1 if bad else 0)
