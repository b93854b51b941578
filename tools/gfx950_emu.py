#!/usr/bin/env python3
"""A small gfx950 (CDNA4) instruction emulator for ONE lane of ONE workgroup, over the assembly text hipcc emits
(`hipcc -S --cuda-device-only`): enough of the ISA to run the library's lane-independent kernels (the bucket
accumulations: every lane walks its own segment, nothing crosses lanes) on the CPU, against the oracle, when no GPU is
at hand.  Test infrastructure (tests/test_kernel_emulation.py); never part of the product.

Model: 256 VGPRs and ~106 SGPRs of Python integers, EXEC / VCC / SGPR-pair masks of which only bit 0 (lane 0) means
anything, SCC, a flat dictionary of dwords for global memory (kernarg segment, buffers, the constant arrays the
translation unit defines: parsed from the .s), one for LDS, one for scratch.  Control flow over the labels of the
function, including hipcc's long-branch sequence (s_getpc_b64 / s_add_u32 label-.Lpost_getpc / s_addc_u32 /
s_setpc_b64).  Floating point (hipcc's integer division goes through v_rcp_iflag_f32) in IEEE single precision via
struct; the reciprocal is the correctly rounded one, the hardware's is within 1 ulp of it, and the division sequences
hipcc emits correct either.  An opcode the emulator does not know raises: nothing is guessed.
"""
import re
import struct

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF
PRIVATE_BASE = 0x0000_5000_0000_0000      # flat addresses in [base, base + 4 GiB) are this lane's scratch
SHARED_BASE = 0x0000_5100_0000_0000


def f32(bits):
    return struct.unpack("<f", struct.pack("<I", bits & M32))[0]


def bits_of(x):
    try:
        return struct.unpack("<I", struct.pack("<f", x))[0]
    except OverflowError:
        return 0x7F800000 if x > 0 else 0xFF800000


def s32(x):
    x &= M32
    return x - (1 << 32) if x & 0x80000000 else x


def s64(x):
    x &= M64
    return x - (1 << 64) if x & (1 << 63) else x


CODE_BASE = 0x0000_6000_0000_0000            # "address" of instruction i = CODE_BASE + 4 i (function calls, returns)


class Program:
    """The code of a .s file -- every function, one instruction list, labels and function entry points -- with one
    kernel chosen as the entry, and the data symbols (constant arrays) of the file."""

    def __init__(self, text, kernel_substr):
        lines = text.split("\n")
        self.ins, self.labels, self.funcs = [], {}, {}
        in_func = False
        for l in lines:
            m = re.match(r"^([A-Za-z_]\w*):\s*(;.*)?$", l)
            if m and not in_func:
                in_func, name = True, m.group(1)
                self.funcs[name] = len(self.ins)
                continue
            if not in_func:
                continue
            if l.startswith(".Lfunc_end"):
                in_func = False
                continue
            m = re.match(r"^(\.L\w+):", l)
            if m:
                self.labels[m.group(1)] = len(self.ins)
            elif l.startswith("\t"):
                body = l.split(";")[0].strip()
                if body and not body.startswith("."):
                    self.ins.append(body)
        # a "function" without instructions is a data symbol that happened to parse as one
        self.funcs = {k: v for k, v in self.funcs.items() if k.startswith("_Z") or k.startswith("probe_")}
        hits = [k for k in self.funcs if kernel_substr in k]
        assert hits, "kernel %r not found" % kernel_substr
        self.name = hits[0]
        self.entry = self.funcs[self.name]
        # data symbols (constant arrays): "sym:\n\t.long 1\n\t.long 2 ..." until the next non-data line
        self.data = {}
        cur = None
        for l in lines:
            m = re.match(r"^(_Z[\w.]+):\s*(;.*)?$", l)
            if m:
                cur = self.data.setdefault(m.group(1), [])
                continue
            if cur is not None:
                t = l.strip()
                m = re.match(r"^\.long\s+(\S+)", t)
                if m:
                    cur.append(int(m.group(1), 0) & M32)
                elif re.match(r"^\.zero\s+(\d+)", t):
                    cur.extend([0] * (int(t.split()[1]) // 4))
                elif t.startswith(".quad"):
                    v = int(t.split()[1], 0)
                    cur.extend([v & M32, (v >> 32) & M32])
                elif t == "" or t.startswith((";", ".size", ".type", ".p2align", ".weak", ".section", ".globl", ".protected",
                                              ".hidden", ".set", ".text")):
                    if t.startswith((".section", ".text", ".type")):
                        cur = None
                else:
                    cur = None
        self.data = {k: v for k, v in self.data.items() if v}


class Lane:
    def __init__(self, prog):
        self.p = prog
        self.v = [0] * 512
        self.a = [0] * 512           # accumulation registers (hipcc spills VGPRs into them)
        self.s = [0] * 128
        self.vcc = 0
        self.exec = 1
        self.scc = 0
        self.mem = {}           # global: byte address (multiple of 4) -> dword
        self.lds = {}
        self.scratch = {}
        self.other_lanes = {}
        self.clamp = False
        self.tmp = {}               # pre-selected SDWA operands
        self.mods = {}
        self.bitop3 = None
        self.count = 0
        self.hist = {}              # opcode -> times executed
        self.lane_no = 0            # this lane's bit in EXEC / VCC / SGPR-pair masks
        self.sym_addr = {}
        base = 0x7000_0000_0000
        for name, words in prog.data.items():
            self.sym_addr[name] = base
            for i, w in enumerate(words):
                self.mem[base + 4 * i] = w
            base += (4 * len(words) + 255) // 256 * 256 + 256

    # ---- operands ------------------------------------------------------------------------------------------------
    def rd(self, tok, width=32):
        tok = tok.strip()
        m = re.fullmatch(r"a\[(0x[0-9a-fA-F]+|\d+)\]", tok)     # a[0x90]: one register, index syntax (inline asm "n" operand)
        if m:
            tok = "a%d" % int(m.group(1), 0)
        if tok in self.tmp:
            return self.tmp[tok]
        neg = False
        if tok.startswith("-") and not re.match(r"^-\d", tok):
            neg, tok = True, tok[1:]
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            return sum(self.v[lo + i] << (32 * i) for i in range(hi - lo + 1))
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            return sum(self.s[lo + i] << (32 * i) for i in range(hi - lo + 1))
        if re.fullmatch(r"v\d+", tok):
            return self.v[int(tok[1:])]
        if re.fullmatch(r"s\d+", tok):
            return self.s[int(tok[1:])]
        m = re.fullmatch(r"a\[(\d+):(\d+)\]", tok)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            return sum(self.a[lo + i] << (32 * i) for i in range(hi - lo + 1))
        if re.fullmatch(r"a\d+", tok):
            return self.a[int(tok[1:])]
        if tok == "vcc":
            return self.vcc
        if tok == "exec":
            return self.exec
        if tok in ("vcc_lo", "exec_lo"):
            return (self.vcc if tok[0] == "v" else self.exec) & M32
        if tok in ("vcc_hi", "exec_hi"):
            return ((self.vcc if tok[0] == "v" else self.exec) >> 32) & M32
        if tok == "scc":
            return self.scc
        if tok in ("src_private_base", "src_shared_base"):     # apertures of flat addressing (64-bit: the window's base)
            return PRIVATE_BASE if "private" in tok else SHARED_BASE
        if re.fullmatch(r"-?(0x[0-9a-fA-F]+|\d+)", tok):
            v = int(tok, 0)
            return v & (M64 if width == 64 else M32)
        if re.fullmatch(r"-?\d+\.\d+", tok):                 # inline float constant
            return bits_of(float(tok))
        raise AssertionError("operand %r" % tok)

    def wr(self, tok, val):
        tok = tok.strip()
        m = re.fullmatch(r"a\[(0x[0-9a-fA-F]+|\d+)\]", tok)
        if m:
            tok = "a%d" % int(m.group(1), 0)
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            for i in range(hi - lo + 1):
                self.v[lo + i] = (val >> (32 * i)) & M32
            return
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            for i in range(hi - lo + 1):
                self.s[lo + i] = (val >> (32 * i)) & M32
            return
        m = re.fullmatch(r"a\[(\d+):(\d+)\]", tok)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            for i in range(hi - lo + 1):
                self.a[lo + i] = (val >> (32 * i)) & M32
            return
        if re.fullmatch(r"v\d+", tok):
            self.v[int(tok[1:])] = val & M32
        elif re.fullmatch(r"a\d+", tok):
            self.a[int(tok[1:])] = val & M32
        elif re.fullmatch(r"s\d+", tok):
            self.s[int(tok[1:])] = val & M32
        elif tok == "vcc":
            self.vcc = val & M64
        elif tok == "exec":
            self.exec = val & M64
        elif tok == "vcc_lo":
            self.vcc = (self.vcc & ~M32) | (val & M32)
        elif tok == "vcc_hi":
            self.vcc = (self.vcc & M32) | ((val & M32) << 32)
        elif tok == "exec_lo":
            self.exec = (self.exec & ~M32) | (val & M32)
        elif tok == "exec_hi":
            self.exec = (self.exec & M32) | ((val & M32) << 32)
        else:
            raise AssertionError("destination %r" % tok)

    def mbit(self, mask):
        return (mask >> self.lane_no) & 1

    def wr_mbit(self, tok, bit):
        """this lane's bit of a mask destination (vcc or an SGPR pair); the other lanes' bits stay"""
        cur = self.rd(tok, 64) if tok.startswith("s[") else (self.vcc if tok == "vcc" else None)
        assert cur is not None, "mask destination %r" % tok
        self.wr(tok, (cur & ~(1 << self.lane_no)) | ((bit & 1) << self.lane_no))

    def flat_space(self, op, addr):
        if op.startswith("flat_") and (addr >> 32) == (PRIVATE_BASE >> 32):
            return self.scratch, addr & M32
        if op.startswith("flat_") and (addr >> 32) == (SHARED_BASE >> 32):
            return self.lds, addr & M32
        return self.mem, addr

    def load(self, space, addr, n):
        out = 0
        for i in range(n):
            a = addr + 4 * i
            assert a % 4 == 0 and a in space, "load from unmapped %s address 0x%x" % (
                "LDS" if space is self.lds else "scratch" if space is self.scratch else "global", a)
            out |= space[a] << (32 * i)
        return out

    def store(self, space, addr, val, n):
        for i in range(n):
            assert (addr + 4 * i) % 4 == 0
            space[addr + 4 * i] = (val >> (32 * i)) & M32

    # ---- execution -----------------------------------------------------------------------------------------------
    _parsed = None

    def parse(self, pc):
        """instruction pc -> (op, operands, modifiers, text); cached per program"""
        cache = self.p.__dict__.setdefault("_parse_cache", {})
        if pc in cache:
            return cache[pc]
        text = self.p.ins[pc]
        op, _, rest = text.partition(" ")
        mods = {}
        for key in ("offset0", "offset1", "offset"):
            m = re.search(r"\s%s:(-?\d+)" % key, rest)      # (global / scratch offsets are signed)
            if m:
                mods[key] = int(m.group(1))
                rest = rest[:m.start()] + rest[m.end():]
        rest = re.sub(r"\s(glc|slc|nt|sc0|sc1)\b", "", rest)
        m = re.search(r"\sbitop3:(0x[0-9a-fA-F]+|\d+)", rest)
        mods["bitop3"] = int(m.group(1), 0) if m else None
        if m:
            rest = rest[:m.start()] + rest[m.end():]
        for key in ("dst_sel", "dst_unused", "src0_sel", "src1_sel"):          # SDWA
            m = re.search(r"\s%s:(\w+)" % key, rest)
            if m:
                mods[key] = m.group(1)
                rest = rest[:m.start()] + rest[m.end():]
        mods["clamp"] = bool(re.search(r"\sclamp\b", rest))
        rest = re.sub(r"\sclamp\b", "", rest)
        a = [x.strip() for x in rest.split(",")] if rest.strip() else []
        cache[pc] = (op, a, mods, text)
        return cache[pc]

    def run(self, limit=5_000_000):
        """one lane, alone: lane 0 of its wave"""
        pc = self.p.entry
        while pc is not None:
            assert pc < len(self.p.ins), "ran off the end"
            op, a, mods, text = self.parse(pc)
            self.count += 1
            self.hist[op] = self.hist.get(op, 0) + 1
            assert self.count < limit, "instruction limit"
            pc = self.exec_one(pc, op, a, mods, text, self.exec & 1)

    def exec_one(self, pc, op, a, mods, text, live):
        """executes instruction pc for this lane (scalar state included) -> next pc, None at s_endpgm"""
        ins, labels = self.p.ins, self.p.labels
        nxt = pc + 1
        self.bitop3, self.clamp, self.mods = mods["bitop3"], mods["clamp"], mods

        # -------- program flow
        if op == "s_endpgm":
            return None
        if op in ("s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_sleep", "s_waitcnt_depctr", "s_setreg_imm32_b32"):
            return nxt
        if op == "s_branch":
            return labels[a[0]]
        if op.startswith("s_cbranch_"):
            cond = {"s_cbranch_scc0": self.scc == 0, "s_cbranch_scc1": self.scc == 1,
                    "s_cbranch_execz": live == 0, "s_cbranch_execnz": live == 1,
                    "s_cbranch_vccz": (self.vcc & 1) == 0, "s_cbranch_vccnz": (self.vcc & 1) == 1}[op]
            return labels[a[0]] if cond else nxt
        if op == "s_getpc_b64":
            # long branch (.. s_add_u32 label-.Lpost_getpc ..) or the address of a data symbol (sym@rel32@lo+4)
            add = ins[pc + 1]
            m = re.search(r"\((\.LBB\w+)-\.Lpost_getpc\d+\)", add)
            if m:
                assert ins[pc + 3].startswith("s_setpc_b64")
                return labels[m.group(1)]
            m = re.search(r"([\w.]+)@rel32@lo\+(\d+)", add)          # sym@rel32@lo+4 = the symbol itself, +36 = sym + 32
            assert m, "s_getpc_b64 followed by %r" % add
            if m.group(1) not in self.sym_addr:               # a function of the file
                assert m.group(1) in self.p.funcs, "unknown symbol %r" % m.group(1)
                self.sym_addr[m.group(1)] = CODE_BASE + 4 * self.p.funcs[m.group(1)]
            self.wr(a[0], self.sym_addr[m.group(1)] + int(m.group(2)) - 4)
            assert ins[pc + 2].startswith("s_addc_u32")
            return pc + 3

        if op == "s_swappc_b64":                 # call: return address into a[0], jump to the function at a[1]
            target = self.rd(a[1], 64)
            assert (target - CODE_BASE) % 4 == 0 and 0 <= (target - CODE_BASE) // 4 < len(ins), "call to 0x%x" % target
            self.wr(a[0], CODE_BASE + 4 * nxt)
            return (target - CODE_BASE) // 4
        if op == "s_setpc_b64":                  # return (the long-branch use of s_setpc_b64 is taken at its s_getpc_b64)
            target = self.rd(a[0], 64)
            assert (target - CODE_BASE) % 4 == 0, "s_setpc_b64 to 0x%x" % target
            return (target - CODE_BASE) // 4

        # -------- scalar ALU
        if op == "s_mov_b32":
            self.wr(a[0], self.rd(a[1]))
        elif op == "s_mov_b64":
            v = self.rd(a[1], 64)
            self.wr(a[0], M64 if a[1] == "-1" else v)
        elif op == "s_movk_i32":
            v = int(a[1], 0) & 0xFFFF
            self.wr(a[0], v - 0x10000 if v & 0x8000 else v)
        elif op == "s_mulk_i32":
            v = int(a[1], 0) & 0xFFFF
            self.wr(a[0], s32(self.rd(a[0])) * (v - 0x10000 if v & 0x8000 else v))
        elif op == "s_addk_i32":
            v = int(a[1], 0) & 0xFFFF
            v = v - 0x10000 if v & 0x8000 else v
            r = s32(self.rd(a[0])) + v
            self.scc = int(not (-2**31 <= r < 2**31))
            self.wr(a[0], r & M32)
        elif op == "s_brev_b32":
            self.wr(a[0], int("{:032b}".format(self.rd(a[1]) & M32)[::-1], 2))
        elif op in ("s_add_u32", "s_add_i32"):
            r = (self.rd(a[1]) & M32) + (self.rd(a[2]) & M32)
            self.wr(a[0], r)
            self.scc = (r >> 32) & 1 if op == "s_add_u32" else int(not (-2**31 <= s32(self.rd(a[1])) + s32(self.rd(a[2])) < 2**31))
        elif op == "s_addc_u32":
            r = (self.rd(a[1]) & M32) + (self.rd(a[2]) & M32) + self.scc
            self.wr(a[0], r)
            self.scc = (r >> 32) & 1
        elif op in ("s_sub_u32", "s_sub_i32"):
            x, y = self.rd(a[1]) & M32, self.rd(a[2]) & M32
            self.wr(a[0], x - y)
            self.scc = int(y > x) if op == "s_sub_u32" else int(not (-2**31 <= s32(x) - s32(y) < 2**31))
        elif op == "s_subb_u32":
            x, y = self.rd(a[1]) & M32, (self.rd(a[2]) & M32) + self.scc
            self.wr(a[0], x - y)
            self.scc = int(y > x)
        elif op == "s_mul_i32":
            self.wr(a[0], (self.rd(a[1]) & M32) * (self.rd(a[2]) & M32))
        elif op == "s_mul_hi_u32":
            self.wr(a[0], ((self.rd(a[1]) & M32) * (self.rd(a[2]) & M32)) >> 32)
        elif op == "s_mul_hi_i32":
            self.wr(a[0], ((s32(self.rd(a[1])) * s32(self.rd(a[2]))) >> 32) & M32)
        elif op in ("s_lshl_b32", "s_lshr_b32", "s_ashr_i32"):
            x, sh = self.rd(a[1]) & M32, self.rd(a[2]) & 31
            r = (x << sh) & M32 if op == "s_lshl_b32" else (x >> sh) if op == "s_lshr_b32" else (s32(x) >> sh) & M32
            self.wr(a[0], r)
            self.scc = int(r != 0)
        elif op == "s_ashr_i64":
            x, sh = self.rd(a[1], 64) & M64, self.rd(a[2]) & 63
            r = ((x - (1 << 64) if x >> 63 else x) >> sh) & M64
            self.wr(a[0], r)
            self.scc = int(r != 0)
        elif op in ("s_lshl_b64", "s_lshr_b64"):
            x, sh = self.rd(a[1], 64) & M64, self.rd(a[2]) & 63
            r = (x << sh) & M64 if op == "s_lshl_b64" else x >> sh
            self.wr(a[0], r)
            self.scc = int(r != 0)
        elif op in ("s_and_b32", "s_or_b32", "s_xor_b32", "s_andn2_b32", "s_and_b64", "s_or_b64", "s_xor_b64",
                    "s_andn2_b64", "s_orn2_b64"):
            w = 64 if op.endswith("b64") else 32
            mask = M64 if w == 64 else M32
            x = M64 if a[1] == "-1" else self.rd(a[1], w)
            y = M64 if a[2] == "-1" else self.rd(a[2], w)
            x &= mask
            y &= mask
            kind = op[2:-4]
            r = {"and": x & y, "or": x | y, "xor": x ^ y, "andn2": x & ~y, "orn2": x | ~y}[kind] & mask
            self.wr(a[0], r)
            self.scc = int(r != 0)
        elif op == "s_not_b64":
            r = ~self.rd(a[1], 64) & M64
            self.wr(a[0], r)
            self.scc = int(r != 0)
        elif op in ("s_bitcmp0_b32", "s_bitcmp1_b32"):
            bit = (self.rd(a[0]) >> (self.rd(a[1]) & 31)) & 1
            self.scc = int(bit == (1 if op == "s_bitcmp1_b32" else 0))
        elif op == "s_not_b32":
            r = ~self.rd(a[1]) & M32
            self.wr(a[0], r)
            self.scc = int(r != 0)
        elif op in ("s_and_saveexec_b64", "s_or_saveexec_b64", "s_andn2_saveexec_b64", "s_xor_saveexec_b64"):
            src = M64 if a[1] == "-1" else self.rd(a[1], 64)
            old = self.exec
            kind = op[2:-13]
            new = {"and": src & old, "or": src | old, "andn2": src & ~old, "xor": src ^ old}[kind] & M64
            self.wr(a[0], old)
            self.exec = new
            self.scc = int(new != 0)
        elif op in ("s_ff1_i32_b64", "s_ff1_i32_b32"):
            x = self.rd(a[1], 64 if op.endswith("b64") else 32)
            self.wr(a[0], (x & -x).bit_length() - 1 if x else M32)
        elif op in ("s_bcnt1_i32_b64", "s_bcnt1_i32_b32"):
            r = bin(self.rd(a[1], 64 if op.endswith("b64") else 32)).count("1")
            self.wr(a[0], r)
            self.scc = int(r != 0)
        elif op in ("s_flbit_i32_b32", "s_flbit_i32_b64"):
            wd = 64 if op.endswith("b64") else 32
            x = self.rd(a[1], wd)
            self.wr(a[0], wd - x.bit_length() if x else M32)
        elif op in ("s_min_u32", "s_max_u32"):
            x, y = self.rd(a[1]) & M32, self.rd(a[2]) & M32
            r = min(x, y) if "min" in op else max(x, y)
            self.wr(a[0], r)
            self.scc = int(r == x)
        elif op in ("s_bfe_u32",):
            x, ctl = self.rd(a[1]) & M32, self.rd(a[2])
            r = (x >> (ctl & 31)) & ((1 << ((ctl >> 16) & 0x7F)) - 1)
            self.wr(a[0], r)
            self.scc = int(r != 0)
        elif op == "s_cselect_b32":
            self.wr(a[0], self.rd(a[1]) if self.scc else self.rd(a[2]))
        elif op == "s_cselect_b64":
            self.wr(a[0], self.rd(a[1], 64) if self.scc else self.rd(a[2], 64))
        elif op.startswith("s_cmpk_"):           # compare with a 16-bit immediate (sign-extended for i32, zero-extended for u32)
            kind, ty = op[7:].rsplit("_", 1)
            x, k = self.rd(a[0]), int(a[1], 0) & 0xFFFF
            if ty == "i32":
                x, y = s32(x), (k - 0x10000 if k & 0x8000 else k)
            else:
                x, y = x & M32, k
            self.scc = int({"eq": x == y, "lg": x != y, "gt": x > y, "ge": x >= y, "lt": x < y, "le": x <= y}[kind])
        elif op.startswith("s_cmp_"):
            kind, ty = op[6:].rsplit("_", 1)
            x, y = self.rd(a[0]), self.rd(a[1])
            if ty == "i32":
                x, y = s32(x), s32(y)
            else:
                x, y = x & M32, y & M32
            self.scc = int({"eq": x == y, "lg": x != y, "gt": x > y, "ge": x >= y, "lt": x < y, "le": x <= y}[kind])
        elif op.startswith("s_load_dword"):
            n = {"": 1, "x2": 2, "x4": 4, "x8": 8, "x16": 16}[op[len("s_load_dword"):]]
            off = self.rd(a[2]) if len(a) > 2 else 0
            self.wr(a[0] if n > 1 else a[0], self.load(self.mem, self.rd(a[1], 64) + off + mods.get("offset", 0), n))

        # -------- vector memory / LDS
        elif op.startswith(("global_load_dword", "flat_load_dword")):
            if live:
                n = {"": 1, "x2": 2, "x3": 3, "x4": 4}[op.split("_dword")[1]]
                if len(a) >= 3 and a[2] != "off":
                    addr = self.rd(a[2], 64) + (self.rd(a[1]) & M32)
                else:
                    addr = self.rd(a[1], 64)
                space, addr = self.flat_space(op, addr + mods.get("offset", 0))
                self.wr(a[0], self.load(space, addr, n))
        elif op.startswith(("global_store_dword", "flat_store_dword")):
            if live:
                n = {"": 1, "x2": 2, "x3": 3, "x4": 4}[op.split("_dword")[1]]
                if len(a) >= 3 and a[2] != "off":
                    addr = self.rd(a[2], 64) + (self.rd(a[0]) & M32)
                else:
                    addr = self.rd(a[0], 64)
                space, addr = self.flat_space(op, addr + mods.get("offset", 0))
                self.store(space, addr, self.rd(a[1]), n)
        elif op.startswith("scratch_load_dword"):
            if live:
                n = {"": 1, "x2": 2, "x3": 3, "x4": 4}[op[len("scratch_load_dword"):]]
                base = (0 if a[1] == "off" else self.rd(a[1])) + (0 if a[2] == "off" else self.rd(a[2]))
                self.wr(a[0], self.load(self.scratch, base + mods.get("offset", 0), n))
        elif op.startswith("scratch_store_dword"):
            if live:
                n = {"": 1, "x2": 2, "x3": 3, "x4": 4}[op[len("scratch_store_dword"):]]
                base = (0 if a[0] == "off" else self.rd(a[0])) + (0 if a[2] == "off" else self.rd(a[2]))
                self.store(self.scratch, base + mods.get("offset", 0), self.rd(a[1]), n)
        elif op in ("ds_read_b32", "ds_read_b64", "ds_read_b128"):
            if live:
                n = {"b32": 1, "b64": 2, "b128": 4}[op[8:]]
                self.wr(a[0], self.load(self.lds, (self.rd(a[1]) & M32) + mods.get("offset", 0), n))
        elif op in ("ds_write_b32", "ds_write_b64", "ds_write_b128"):
            if live:
                n = {"b32": 1, "b64": 2, "b128": 4}[op[9:]]
                self.store(self.lds, (self.rd(a[0]) & M32) + mods.get("offset", 0), self.rd(a[1]), n)
        elif op in ("ds_read_u16", "ds_write_b16"):
            if live:
                if op == "ds_read_u16":
                    addr = (self.rd(a[1]) & M32) + mods.get("offset", 0)
                    self.wr(a[0], (self.lds.get(addr & ~3, 0) >> (8 * (addr & 2))) & 0xFFFF)
                else:
                    addr = (self.rd(a[0]) & M32) + mods.get("offset", 0)
                    sh = 8 * (addr & 2)
                    self.lds[addr & ~3] = (self.lds.get(addr & ~3, 0) & ~(0xFFFF << sh)) | ((self.rd(a[1]) & 0xFFFF) << sh)
        elif op in ("ds_read2_b64", "ds_read2st64_b64"):
            if live:
                unit = 8 * (64 if "st64" in op else 1)
                base = self.rd(a[1]) & M32
                lo = self.load(self.lds, base + unit * mods.get("offset0", 0), 2)
                hi = self.load(self.lds, base + unit * mods.get("offset1", 0), 2)
                self.wr(a[0], lo | (hi << 64))
        elif op in ("ds_write2_b64", "ds_write2st64_b64"):
            if live:
                unit = 8 * (64 if "st64" in op else 1)
                base = self.rd(a[0]) & M32
                self.store(self.lds, base + unit * mods.get("offset0", 0), self.rd(a[1]), 2)
                self.store(self.lds, base + unit * mods.get("offset1", 0), self.rd(a[2]), 2)
        elif op in ("ds_read_b96", "ds_write_b96"):
            if live:
                if op == "ds_read_b96":
                    self.wr(a[0], self.load(self.lds, (self.rd(a[1]) & M32) + mods.get("offset", 0), 3))
                else:
                    self.store(self.lds, (self.rd(a[0]) & M32) + mods.get("offset", 0), self.rd(a[1]), 3)
        elif op in ("ds_read2_b32", "ds_read2st64_b32"):
            if live:
                unit = 4 * (64 if "st64" in op else 1)
                base = self.rd(a[1]) & M32
                lo = self.load(self.lds, base + unit * mods.get("offset0", 0), 1)
                hi = self.load(self.lds, base + unit * mods.get("offset1", 0), 1)
                self.wr(a[0], lo | (hi << 32))
        elif op in ("ds_write2_b32", "ds_write2st64_b32"):
            if live:
                unit = 4 * (64 if "st64" in op else 1)
                base = self.rd(a[0]) & M32
                self.store(self.lds, base + unit * mods.get("offset0", 0), self.rd(a[1]), 1)
                self.store(self.lds, base + unit * mods.get("offset1", 0), self.rd(a[2]), 1)

        elif re.fullmatch(r"ds_(add|sub|max|min|or|and)(_rtn)?_(u32|b32)", op):
            if live:                                # LDS atomics; a cell nobody wrote yet reads as 0 (the other lanes' job)
                kind, rtn = re.fullmatch(r"ds_(\w+?)(_rtn)?_(u32|b32)", op).group(1, 2)
                dst = a[0] if rtn else None
                addr_tok, val_tok = (a[1], a[2]) if rtn else (a[0], a[1])
                addr = (self.rd(addr_tok) & M32) + mods.get("offset", 0)
                old = self.lds.get(addr, 0)
                x = self.rd(val_tok) & M32
                new = {"add": old + x, "sub": old - x, "max": max(old, x), "min": min(old, x), "or": old | x,
                       "and": old & x}[kind] & M32
                self.lds[addr] = new
                if dst:
                    self.wr(dst, old)

        # -------- vector ALU (skipped when the lane is masked off)
        elif op == "v_readfirstlane_b32":
            self.wr(a[0], self.rd(a[1]))
        elif op == "v_writelane_b32":           # hipcc spills SGPRs into the lanes of a VGPR: lanes other than 0 live here
            vreg, lane_no = int(a[0][1:]), self.rd(a[2]) & 63
            self.other_lanes[(vreg, lane_no)] = self.rd(a[1]) & M32
            if lane_no == 0:
                self.v[vreg] = self.rd(a[1]) & M32
        elif op == "v_readlane_b32":
            vreg, lane_no = int(a[1][1:]), self.rd(a[2]) & 63
            self.wr(a[0], self.v[vreg] if lane_no == 0 else self.other_lanes[(vreg, lane_no)])
        elif op.startswith("v_"):
            if live:
                self.valu(op, a, text)
        else:
            raise AssertionError("the emulator does not know %r: %s" % (op, text))
        return nxt

    def valu(self, op, a, text):
        rd, wr = self.rd, self.wr
        if op.endswith("_sdwa"):                # sub-dword operand selection, then the plain operation
            def sel(v, how):
                v &= M32
                if how.startswith("BYTE_"):
                    return (v >> (8 * int(how[5]))) & 0xFF
                if how.startswith("WORD_"):
                    return (v >> (16 * int(how[5]))) & 0xFFFF
                assert how == "DWORD", how
                return v
            assert self.mods.get("dst_sel", "DWORD") == "DWORD", text
            self.tmp = {"__s0": sel(rd(a[1]), self.mods.get("src0_sel", "DWORD"))}
            args = [a[0], "__s0"]
            if len(a) > 2:
                self.tmp["__s1"] = sel(rd(a[2]), self.mods.get("src1_sel", "DWORD"))
                args.append("__s1")
            try:
                self.valu(op[:-5], args + a[3:], text)
            finally:
                self.tmp = {}
            return
        base = re.sub(r"_e(32|64)$", "", op)
        if base in ("v_mov_b32", "v_mov_b64"):
            wr(a[0], rd(a[1], 64 if base.endswith("64") else 32))
        elif base == "v_add_u32":
            r = (rd(a[1]) & M32) + (rd(a[2]) & M32)
            wr(a[0], min(r, M32) if self.clamp else r)
        elif base == "v_sub_u32":
            r = (rd(a[1]) & M32) - (rd(a[2]) & M32)
            wr(a[0], max(r, 0) if self.clamp else r)
        elif base == "v_subrev_u32":
            wr(a[0], rd(a[2]) - rd(a[1]))
        elif base in ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32"):
            # vdst, carry-out, src0, src1[, carry-in]
            x, y = rd(a[2]) & M32, rd(a[3]) & M32
            cin = self.mbit(rd(a[4], 64)) if len(a) > 4 else 0
            if "rev" in base:
                x, y = y, x
            if base.startswith("v_add"):
                r = x + y + cin
                cout = r >> 32
            else:
                r = x - y - cin
                cout = int(y + cin > x)
            wr(a[0], r)
            self.wr_mbit(a[1], cout)
        elif base == "v_add3_u32":
            wr(a[0], rd(a[1]) + rd(a[2]) + rd(a[3]))
        elif base == "v_lshl_add_u32":
            wr(a[0], ((rd(a[1]) & M32) << (rd(a[2]) & 31)) + rd(a[3]))
        elif base == "v_add_lshl_u32":
            wr(a[0], ((rd(a[1]) + rd(a[2])) & M32) << (rd(a[3]) & 31))
        elif base == "v_lshl_or_b32":
            wr(a[0], (((rd(a[1]) & M32) << (rd(a[2]) & 31)) & M32) | rd(a[3]))
        elif base == "v_lshl_add_u64":
            wr(a[0], (((rd(a[1], 64) & M64) << (rd(a[2]) & 7)) + rd(a[3], 64)) & M64)
        elif base in ("v_and_b32", "v_or_b32", "v_xor_b32"):
            x, y = rd(a[1]), rd(a[2])
            wr(a[0], x & y if "and" in base else x | y if "_or_" in base else x ^ y)
        elif base == "v_or3_b32":
            wr(a[0], rd(a[1]) | rd(a[2]) | rd(a[3]))
        elif base == "v_and_or_b32":
            wr(a[0], (rd(a[1]) & rd(a[2])) | rd(a[3]))
        elif base == "v_xad_u32":
            wr(a[0], ((rd(a[1]) ^ rd(a[2])) & M32) + rd(a[3]))
        elif base == "v_bfi_b32":               # (s0 & s1) | (~s0 & s2)
            m_, x, y = rd(a[1]), rd(a[2]), rd(a[3])
            wr(a[0], (m_ & x) | (~m_ & y))
        elif base in ("v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32"):     # bits of the mask BELOW this lane, low / high half
            below = (1 << self.lane_no) - 1
            half = (below & M32) if "lo" in base else (below >> 32)
            wr(a[0], bin(rd(a[1]) & M32 & half).count("1") + rd(a[2]))
        elif base == "v_bcnt_u32_b32":
            wr(a[0], bin(rd(a[1]) & M32).count("1") + rd(a[2]))
        elif base == "v_bitop3_b32":            # truth table over (s0, s1, s2) = (0xF0, 0xCC, 0xAA): bit (s0 << 2 | s1 << 1 | s2)
            x, y, z, tt = rd(a[1]) & M32, rd(a[2]) & M32, rd(a[3]) & M32, self.bitop3
            r = 0
            for idx in range(8):
                if (tt >> idx) & 1:
                    r |= (x if idx & 4 else ~x) & (y if idx & 2 else ~y) & (z if idx & 1 else ~z)
            wr(a[0], r & M32)
        elif base in ("v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_accvgpr_mov_b32"):
            wr(a[0], rd(a[1]))
        elif base == "v_bfrev_b32":
            wr(a[0], int("{:032b}".format(rd(a[1]) & M32)[::-1], 2))
        elif base == "v_not_b32":
            wr(a[0], ~rd(a[1]))
        elif base == "v_lshlrev_b32":
            wr(a[0], (rd(a[2]) & M32) << (rd(a[1]) & 31))
        elif base == "v_lshrrev_b32":
            wr(a[0], (rd(a[2]) & M32) >> (rd(a[1]) & 31))
        elif base == "v_ashrrev_i32":
            wr(a[0], s32(rd(a[2])) >> (rd(a[1]) & 31))
        elif base == "v_lshlrev_b64":
            wr(a[0], ((rd(a[2], 64) & M64) << (rd(a[1]) & 63)) & M64)
        elif base == "v_lshrrev_b64":
            wr(a[0], (rd(a[2], 64) & M64) >> (rd(a[1]) & 63))
        elif base == "v_ashrrev_i64":
            wr(a[0], (s64(rd(a[2], 64)) >> (rd(a[1]) & 63)) & M64)
        elif base == "v_alignbit_b32":
            wr(a[0], (((rd(a[1]) & M32) << 32) | (rd(a[2]) & M32)) >> (rd(a[3]) & 31))
        elif base == "v_bfe_u32":
            wr(a[0], ((rd(a[1]) & M32) >> (rd(a[2]) & 31)) & ((1 << (rd(a[3]) & 31)) - 1))
        elif base == "v_mul_lo_u32":
            wr(a[0], (rd(a[1]) & M32) * (rd(a[2]) & M32))
        elif base == "v_mul_hi_u32":
            wr(a[0], ((rd(a[1]) & M32) * (rd(a[2]) & M32)) >> 32)
        elif base == "v_mul_i32_i24":
            x, y = rd(a[1]) & 0xFFFFFF, rd(a[2]) & 0xFFFFFF
            x = x - (1 << 24) if x & 0x800000 else x
            y = y - (1 << 24) if y & 0x800000 else y
            wr(a[0], x * y)
        elif base == "v_mul_hi_i32_i24":
            x, y = rd(a[1]) & 0xFFFFFF, rd(a[2]) & 0xFFFFFF
            x = x - (1 << 24) if x & 0x800000 else x
            y = y - (1 << 24) if y & 0x800000 else y
            wr(a[0], (x * y) >> 32)
        elif base == "v_mul_hi_u32_u24":
            wr(a[0], ((rd(a[1]) & 0xFFFFFF) * (rd(a[2]) & 0xFFFFFF)) >> 32)
        elif base == "v_mul_u32_u24":
            wr(a[0], (rd(a[1]) & 0xFFFFFF) * (rd(a[2]) & 0xFFFFFF))
        elif base == "v_mad_u32_u24":
            wr(a[0], (rd(a[1]) & 0xFFFFFF) * (rd(a[2]) & 0xFFFFFF) + rd(a[3]))
        elif base == "v_mad_u64_u32":
            wr(a[0], ((rd(a[2]) & M32) * (rd(a[3]) & M32) + rd(a[4], 64)) & M64)       # a[1]: carry-out, never read
        elif base == "v_mad_i64_i32":
            wr(a[0], (s32(rd(a[2])) * s32(rd(a[3])) + s64(rd(a[4], 64))) & M64)
        elif base in ("v_min_u32", "v_max_u32"):
            x, y = rd(a[1]) & M32, rd(a[2]) & M32
            wr(a[0], min(x, y) if "min" in base else max(x, y))
        elif base == "v_cndmask_b32":
            sel = self.mbit(rd(a[3], 64) if len(a) > 3 else self.vcc)
            wr(a[0], rd(a[2]) if sel else rd(a[1]))
        elif base.startswith("v_cmp_"):                      # v_cmp_<op>_<type> dst (vcc | s[a:b]), src0, src1
            kind, ty = base[6:].rsplit("_", 1)
            assert len(a) == 3, text
            w = 64 if ty.endswith("64") else 32
            xv, yv = rd(a[1], w), rd(a[2], w)
            if ty.startswith("i"):
                xv, yv = (s64(xv), s64(yv)) if w == 64 else (s32(xv), s32(yv))
            else:
                mask = M64 if w == 64 else M32
                xv, yv = xv & mask, yv & mask
            res = {"eq": xv == yv, "ne": xv != yv, "lg": xv != yv, "gt": xv > yv, "ge": xv >= yv, "lt": xv < yv,
                   "le": xv <= yv}[kind]
            self.wr_mbit(a[0], int(res))
        elif base == "v_cvt_f32_u32":
            wr(a[0], bits_of(float(rd(a[1]) & M32)))
        elif base == "v_cvt_u32_f32":
            x = f32(rd(a[1]))
            wr(a[0], 0 if x != x or x <= 0 else min(int(x), M32))
        elif base in ("v_rcp_f32", "v_rcp_iflag_f32"):
            x = f32(rd(a[1]))
            wr(a[0], bits_of(struct.unpack("<f", struct.pack("<f", 1.0 / x))[0]) if x != 0 else 0x7F800000)
        elif base == "v_mul_f32":
            wr(a[0], bits_of(f32(rd(a[1])) * f32(rd(a[2]))))
        elif base == "v_trunc_f32":
            x = f32(rd(a[1]))
            wr(a[0], bits_of(float(int(x))))
        elif base == "v_fmamk_f32":           # d = s0 * K + s1
            wr(a[0], bits_of(f32(rd(a[1])) * f32(rd(a[2])) + f32(rd(a[3]))))
        elif base == "v_fmaak_f32":           # d = s0 * s1 + K
            wr(a[0], bits_of(f32(rd(a[1])) * f32(rd(a[2])) + f32(rd(a[3]))))
        elif base == "v_fmac_f32":
            wr(a[0], bits_of(f32(rd(a[1])) * f32(rd(a[2])) + f32(rd(a[0]))))
        elif base == "v_fma_f32":
            wr(a[0], bits_of(f32(rd(a[1])) * f32(rd(a[2])) + f32(rd(a[3]))))
        elif base == "v_perm_b32":
            src = ((rd(a[1]) & M32) << 32) | (rd(a[2]) & M32)
            sel = rd(a[3]) & M32
            out = 0
            for i in range(4):
                c = (sel >> (8 * i)) & 0xFF
                assert c < 8 or c == 0x0C, "v_perm_b32 selector %x" % c
                byte = 0 if c == 0x0C else (src >> (8 * c)) & 0xFF
                out |= byte << (8 * i)
            wr(a[0], out)
        else:
            raise AssertionError("the emulator does not know %r: %s" % (op, text))


class Workgroup:
    """A workgroup of 64-lane waves in lockstep per wave, waves interleaved at s_barrier: for kernels whose lanes talk
    through LDS, ballots and barriers (the in-workgroup bucket tree).  Every lane is a Lane; the lanes of a wave share
    the SGPR file (one list object), VCC / EXEC / SCC live in the wave's scalar context, LDS and global memory are
    shared by all.  Vector and memory instructions run lane by lane under EXEC; everything else once per wave."""

    MASK_WRITERS = ("v_cmp_", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_subrev_co", "v_subbrev_co")

    def __init__(self, prog, n_threads, wg_id=(0, 0), kernarg_addr=0):
        assert n_threads % 64 == 0
        self.p = prog
        self.mem, self.lds = {}, {}
        self.waves = []
        first = Lane(prog)                                  # places the constant arrays of the file in memory
        self.mem.update(first.mem)
        self.sym_addr = first.sym_addr
        for w in range(n_threads // 64):
            sc = Lane(prog)                                 # the wave's scalar context
            sc.mem, sc.lds, sc.sym_addr = self.mem, self.lds, self.sym_addr
            sc.exec = M64
            sc.s[0], sc.s[1] = kernarg_addr & M32, kernarg_addr >> 32
            sc.s[2], sc.s[3] = wg_id
            lanes = []
            for l in range(64):
                ln = Lane(prog)
                ln.mem, ln.lds, ln.sym_addr = self.mem, self.lds, self.sym_addr
                ln.s = sc.s
                ln.lane_no = l
                ln.v[0] = 64 * w + l                        # work-item id x
                lanes.append(ln)
            self.waves.append({"sc": sc, "lanes": lanes, "pc": prog.entry, "state": "run"})
        self.count = 0

    def step(self, wv):
        sc, lanes, pc = wv["sc"], wv["lanes"], wv["pc"]
        op, a, mods, text = sc.parse(pc)
        self.count += 1
        if op == "s_barrier":
            wv["pc"], wv["state"] = pc + 1, "barrier"
            return
        if op in ("s_cbranch_execz", "s_cbranch_execnz", "s_cbranch_vccz", "s_cbranch_vccnz"):
            cond = {"s_cbranch_execz": sc.exec == 0, "s_cbranch_execnz": sc.exec != 0, "s_cbranch_vccz": sc.vcc == 0,
                    "s_cbranch_vccnz": sc.vcc != 0}[op]
            wv["pc"] = self.p.labels[a[0]] if cond else pc + 1
            return
        if op == "v_readfirstlane_b32":
            src = next((l for l in range(64) if (sc.exec >> l) & 1), 0)
            sc.wr(a[0], lanes[src].rd(a[1]))
        elif op == "v_readlane_b32":
            sc.wr(a[0], lanes[sc.rd(a[2]) & 63].rd(a[1]))
        elif op == "v_writelane_b32":
            lanes[sc.rd(a[2]) & 63].wr(a[0], sc.rd(a[1]))
        elif op.endswith("_dpp"):       # src0 through a row-of-16 data path: row_newbcast / row_shr / row_shl / row_ror : N
            m = re.search(r"(row_newbcast|row_shr|row_shl|row_ror):(\d+)", text)
            assert m and "row_mask:0xf" in text and "bank_mask:0xf" in text, text
            kind, n = m.group(1), int(m.group(2))
            zero_fill = "bound_ctrl:0" in text or "bound_ctrl:1" in text      # (both spellings set BOUND_CTRL: 0 for a lane out of the row)
            args = a[:-1] + [a[-1].split()[0]]
            vals = [ln.rd(args[1]) & M32 for ln in lanes]

            def source(l):
                r = l & 15
                if kind == "row_newbcast":
                    return (l & ~15) + n
                if kind == "row_ror":
                    return (l & ~15) + ((r - n) & 15)
                q = r - n if kind == "row_shr" else r + n
                return (l & ~15) + q if 0 <= q < 16 else None
            for l in range(64):
                if (sc.exec >> l) & 1:
                    src = source(l)
                    if src is None and not zero_fill:
                        continue                                # the lane is disabled: its destination keeps its value
                    ln = lanes[l]
                    ln.bitop3, ln.clamp, ln.mods = mods["bitop3"], mods["clamp"], mods
                    ln.tmp = {"__s0": 0 if src is None else vals[src]}
                    try:
                        ln.valu(op[:-4], [args[0], "__s0"] + args[2:], text)
                    finally:
                        ln.tmp = {}
        elif op in ("v_permlane16_swap_b32_e32", "v_permlane32_swap_b32_e32", "v_permlane16_swap_b32", "v_permlane32_swap_b32"):
            # gfx950: odd rows of vdst <-> even rows of vsrc (16), upper half of vdst <-> lower half of vsrc (32)
            assert sc.exec == M64, "permlane swap under a partial EXEC"
            d = [ln.rd(a[0]) & M32 for ln in lanes]
            v = [ln.rd(a[1]) & M32 for ln in lanes]
            step = 16 if "permlane16" in op else 32
            for l in range(64):
                if (l // step) & 1:
                    lanes[l].wr(a[0], v[l - step])
                    lanes[l - step].wr(a[1], d[l])
        elif op == "ds_bpermute_b32":                       # dst[l] = data[(addr[l] / 4) % 64], all reads before any write
            vals = [ln.rd(a[2]) & M32 for ln in lanes]
            off = int(str(mods.get("offset", 0)), 0)
            picks = [vals[(((ln.rd(a[1]) + off) & M32) >> 2) & 63] for ln in lanes]
            for l in range(64):
                if (sc.exec >> l) & 1:
                    lanes[l].wr(a[0], picks[l])
        elif op.startswith(("v_", "ds_", "global_", "flat_", "scratch_")):
            # an instruction that writes a lane mask (compare, carry-out): every lane must see the ORIGINAL value of
            # the destination registers (v_cmp_lt_u32 s[0:1], s0, v73 reads s0), inactive lanes end up as 0
            dst = None
            if op.startswith(self.MASK_WRITERS):
                dst = a[0] if op.startswith("v_cmp_") else a[1]
                orig = sc.vcc if dst == "vcc" else sc.rd(dst, 64)
                acc = 0
            for l in range(64):
                if (sc.exec >> l) & 1:
                    ln = lanes[l]
                    if dst is not None:
                        if dst == "vcc":
                            sc.vcc = orig
                        else:
                            sc.wr(dst, orig)
                    ln.vcc, ln.scc, ln.exec = sc.vcc, sc.scc, sc.exec
                    nxt = ln.exec_one(pc, op, a, mods, text, 1)
                    assert nxt == pc + 1
                    sc.vcc = ln.vcc
                    if dst is not None:
                        acc |= (((sc.vcc if dst == "vcc" else sc.rd(dst, 64)) >> l) & 1) << l
            if dst is not None:
                if dst == "vcc":
                    sc.vcc = acc
                else:
                    sc.wr(dst, acc)
        else:
            nxt = sc.exec_one(pc, op, a, mods, text, int(sc.exec != 0))
            if nxt is None:
                wv["state"] = "done"
                return
            wv["pc"] = nxt
            return
        wv["pc"] = pc + 1

    def run(self, limit=50_000_000):
        while True:
            running = [w for w in self.waves if w["state"] == "run"]
            if not running:
                waiting = [w for w in self.waves if w["state"] == "barrier"]
                if not waiting:
                    return
                for w in waiting:                           # every wave that is still alive has arrived
                    w["state"] = "run"
                continue
            for w in running:
                while w["state"] == "run":
                    self.step(w)
                    assert self.count < limit, "instruction limit"
