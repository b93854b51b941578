#!/usr/bin/env python3
"""Writes distributed-groth16_amd/csrc/fp29_asm_gen.h: the device form of the reduced-radix Montgomery products of
fp29.h (rr::mont_inl / mont_sqr_inl), one explicit instruction sequence per limb shape.

Why instructions.  Given `acc += (uint64_t)x * y`, hipcc starts EVERY column of the product as a v_mad_u64_u32 chain of
its own from zero and joins it to the shifted carry of the column before with a 64-bit addition: 16 v_lshl_add_u64 on
top of the 162 v_mad_u64_u32 of a 9-limb product (7 % of its instructions) and up to 17 live 64-bit accumulators.
LLVM's reassociation ranks the carry last by construction, so no C++ spelling changes that.  Here a column is ONE
accumulator seeded with the carry:
    column k:  [k <= N: acc += m[k-1] p[0]; acc >>= W]      (the end of column k - 1: its low W bits are zero by now)
               acc += sum a[i] b[k-i] (+ c[i] d[k-i] ...)  +  sum_{i<k} m[i] p[k-i]
    then, in C++:  m[k] = (lo(acc) INV) & MASK  (k < N)   or   r[k-N] = lo(acc) & MASK; acc >>= W  (k >= N)
One asm statement per column (the compiler's hazard recognizer puts an s_nop between an asm statement and a reader of
its result, so statements are as large as possible: ~1 s_nop per column).  The limbs of p are SGPR operands (VOP3 takes
one scalar source), everything else VGPRs.  No instruction in a statement reads a register another one of the same
statement writes except the accumulator (v_mad_u64_u32 -> v_mad_u64_u32 / v_lshrrev_b64 on the same pair needs no wait
state on gfx950: hipcc itself emits them back to back).

(Measured and removed in round 4: every product as ONE statement with the accumulator and the m[k] in fixed registers,
so that no s_nop is needed inside it -- 10.89-10.98 ms per 2^20 BN254 proof against 10.76-10.79 for this form, same box,
same call: the fixed registers cost the allocator more than the ~170 s_nop per addition cost the scalar issue port;
profiles/r4a_ab_variants.md.)

Kinds: `mul` a b, `dual` a b + c d, `quad` a b + c d + e f + g h (one reduction each), `sqr` a^2 with the doubled
operand.  Shapes: (N, W) = (9, 29) for the 254 / 255 / 253-bit fields, (14, 28) for the 377 / 381-bit ones.

The host keeps the plain loops of fp29.h (tests/host_arith runs them against the oracle); the DEVICE text written here
is checked on the CPU by tests/test_fp29_asm_isa.py, which compiles it for gfx950, reads the instructions back from
the assembly hipcc emits and executes them on Python integers against a big-number Montgomery product.

usage: python tools/gen_fp29_asm.py            (rewrites the header; tests/test_fp29_asm_isa.py checks it is current)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "distributed-groth16_amd", "csrc", "fp29_asm_gen.h")
SHAPES = [(9, 29), (14, 28)]
PAIR_NAMES = [("a", "b"), ("c", "d"), ("e", "f"), ("g", "h")]


def column_terms(kind, n, k):
    """[(x expr, y expr)] of the operand products of column k (VGPR x VGPR)."""
    lo, hi = max(0, k - n + 1), min(k, n - 1)
    terms = []
    if kind == "sqr":
        for i in range(lo, hi + 1):
            if 2 * i < k:
                terms.append(("a2[%d]" % i, "a[%d]" % (k - i)))
        if k % 2 == 0:
            terms.append(("a[%d]" % (k // 2), "a[%d]" % (k // 2)))
    else:
        npair = {"mul": 1, "dual": 2, "quad": 4}[kind]
        for i in range(lo, hi + 1):
            for x, y in PAIR_NAMES[:npair]:
                terms.append(("%s[%d]" % (x, i), "%s[%d]" % (y, k - i)))
    return terms


def reduction_terms(n, k):
    """[(m index, p index)] of the m[i] p[k-i] products of column k that do not involve m[k]."""
    lo = max(0, k - n + 1)
    hi = min(k - 1, n - 1)
    return [(i, k - i) for i in range(lo, hi + 1)]


class Block:
    """One asm statement: instructions over numbered operands."""

    def __init__(self, seed_zero):
        self.ops = []          # (constraint, expr)
        self.index = {}
        self.lines = []
        self.seed_zero = seed_zero
        self.acc = self.operand("=v" if seed_zero else "+v", "acc")

    def operand(self, constraint, expr):
        key = (constraint, expr)
        if key not in self.index:
            self.index[key] = len(self.ops)
            self.ops.append(key)
        return "%%%d" % self.index[key]

    def mad(self, x, y, y_scalar=False):
        xo = self.operand("v", x)
        yo = self.operand("s" if y_scalar else "v", y)
        addend = "0" if (self.seed_zero and not self.lines) else self.acc
        self.lines.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (self.acc, xo, yo, addend))

    def shr(self, w):
        self.lines.append("v_lshrrev_b64 %s, %d, %s" % (self.acc, w, self.acc))

    def emit(self, out, indent="  "):
        outs = [(c, e) for c, e in self.ops if c in ("=v", "+v")]
        ins = [(c, e) for c, e in self.ops if c not in ("=v", "+v")]
        assert self.ops[:1] == outs and len(outs) == 1
        text = "\\n\\t".join(self.lines)
        # operands are numbered outputs first, then inputs, in self.ops order (acc is operand 0)
        out.append('%sasm("%s"' % (indent, text))
        out.append('%s    : "%s"(acc)' % (indent, outs[0][0]))
        out.append("%s    : %s" % (indent, ", ".join('"%s"(%s)' % (c, e) for c, e in ins)))
        out.append('%s    : "vcc");' % indent)


def gen_function(kind, n, w):
    npair = {"mul": 1, "dual": 2, "quad": 4, "sqr": 1}[kind]
    args = ", ".join("const uint32_t* %s" % nm for pr in PAIR_NAMES[:npair] for nm in pr) if kind != "sqr" else "const uint32_t* a"
    out = []
    out.append("template <class P>")
    out.append("__device__ __forceinline__ void mont_asm_%s_%d(uint32_t* __restrict__ r, %s) {" % (kind, n, args))
    out.append("  using T = RR<P>;")
    out.append('  static_assert(T::N == %d && T::W == %d, "limb shape of this instruction sequence");' % (n, w))
    if kind == "sqr":
        out.append("  uint32_t a2[%d];" % n)
        out.append("#pragma unroll")
        out.append("  for (int i = 0; i < %d; i++) a2[i] = a[i] << 1;" % n)
    out.append("  uint64_t acc;")
    out.append("  uint32_t %s;" % ", ".join("m%d" % i for i in range(n)))
    for k in range(2 * n - 1):
        blk = Block(seed_zero=(k == 0))
        if 1 <= k <= n:
            blk.mad("m%d" % (k - 1), "T::PL.v[0]", y_scalar=True)     # ends column k - 1: its low W bits are now zero
            blk.shr(w)
        for x, y in column_terms(kind, n, k):
            blk.mad(x, y)
        for i, j in reduction_terms(n, k):
            blk.mad("m%d" % i, "T::PL.v[%d]" % j, y_scalar=True)
        out.append("  // column %d" % k)
        blk.emit(out)
        if k < n:
            out.append("  m%d = ((uint32_t)acc * T::INV) & T::MASK;" % k)
        else:
            # (the shift of a column that yields a result limb stays in C++: inside the next statement it would tie the
            # accumulator to one register pair and cost a v_mov_b64 whenever the compiler reads the limb later)
            out.append("  r[%d] = (uint32_t)acc & T::MASK;" % (k - n))
            out.append("  acc >>= %d;" % w)
    out.append("  r[%d] = (uint32_t)acc;" % (n - 1))
    out.append("}")
    return out


def generate():
    out = []
    out.append("// GENERATED by tools/gen_fp29_asm.py -- do not edit (tests/test_fp29_asm_isa.py checks that it is current).")
    out.append("// Device form of rr::mont_inl / mont_sqr_inl (fp29.h): one v_mad_u64_u32 chain per column, seeded with the carry of")
    out.append("// the column before -- see the generator for why this is written as instructions.")
    out.append("#pragma once")
    out.append("namespace dg16 {")
    out.append("namespace rr {")
    for n, w in SHAPES:
        for kind in ("mul", "dual", "quad", "sqr"):
            out.extend(gen_function(kind, n, w))
            out.append("")
    out.append("}  // namespace rr")
    out.append("}  // namespace dg16")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    text = generate()
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        sys.exit(0 if open(OUT).read() == text else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote %s (%d lines)" % (OUT, text.count("\n")))
