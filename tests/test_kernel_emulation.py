"""CPU: the bucket-accumulation KERNELS as hipcc compiled them for gfx950, executed for one lane on the CPU
(tools/gfx950_emu.py) against the oracle's curve arithmetic -- a pre-flight of the device code when no GPU is at hand.

What it covers that no host test can: the kernel bodies themselves (segment lookup, the entry / point gathers with the
next point fetched ahead, the LDS-staged accumulator of G2, the mixed addition with its fused Y3, the zero test, the
doubling and identity branches, the write of the bucket), on the exact instructions the library ships, including the
field products that exist only as gfx950 instruction sequences (csrc/fp29_asm_gen.h).  One lane, one bucket, a handful
of entries: plain and negated points, the same point twice (doubling branch), P then -P (identity in the middle), a
single entry.  The emulator was first run on the kernel of the commit before the instruction-sequence products -- a
binary the GPU tests had passed on hardware -- and reproduced the oracle's sums there.

The translation units are compiled to assembly once per source state (a cache directory under /tmp keyed by the hash of
csrc/): ~1.5 minutes of hipcc on the first run, seconds afterwards."""

import hashlib
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gfx950_emu as E  # noqa: E402

CSRC = os.path.join(ROOT, "distributed-groth16_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
CURVE_ID = {"bn254": 0, "bls12_381": 1, "bls12_377": 2}
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")


def source_key():
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "dg16.h"), "rb").read())
    h.update(open(os.path.join(ROOT, STEPS_PROBE), "rb").read())
    h.update(open(os.path.join(ROOT, FINALIZE_PROBE), "rb").read())
    h.update(open(os.path.join(ROOT, ASSEMBLE_PROBE), "rb").read())
    return h.hexdigest()[:16]


STEPS_PROBE = "tests/isa/steps_probe.hip"            # msm_accumulate_steps_kernel alone (seconds of hipcc)
FINALIZE_PROBE = "tests/isa/finalize_probe.hip"      # msm_finalize_lds_kernel of a 14-limb G2 alone (seconds of hipcc)
ASSEMBLE_PROBE = "tests/isa/assemble_probe.hip"      # prover_assemble_kernel alone (seconds of hipcc)
CASES = [
    ("bn254", 2, "msm_accumulate_lds_kernel"),        # LDS-staged accumulator, four-product Y3, next point fetched ahead
    ("bn254", 1, "msm_accumulate_kernel"),            # fused Y3, one-compare zero test; the bucket tree degenerates (one lane)
    ("bls12_381", 1, "msm_accumulate_kernel"),        # 14 limbs: the other limb shape of the instruction sequences
    # the 14-limb G2 accumulation: three product sites visited by a step loop, temporaries in accumulation registers
    ("bls12_381", 2, "msm_accumulate_steps_kernel"), ("bls12_377", 2, "msm_accumulate_steps_kernel"),
]
if os.environ.get("DG16_EMU_ALL"):                    # +2 minutes of hipcc on a cold cache
    CASES += [("bls12_377", 1, "msm_accumulate_kernel")]


def case_source(kernel):
    return STEPS_PROBE if kernel == "msm_accumulate_steps_kernel" else "msm_group.hip"


def assembly(curve, group, source="msm_group.hip"):
    """`source` of (curve, group) as gfx950 assembly text.  All translation units the tests need are compiled together
    (one hipcc each, in parallel) the first time any is asked for, once per source state; safe under pytest-xdist.
    msm_reduce.hip is built as the Makefile builds it (G2: out-of-line field products)."""
    cache = os.path.join("/tmp", "dg16_emu_cache", source_key())
    os.makedirs(cache, exist_ok=True)
    path = lambda c, g, f: os.path.join(cache, "%s_%s_g%d.s" % (os.path.basename(f).split(".")[0], c, g))     # noqa: E731
    units = [(c, g, case_source(k)) for c, g, k in CASES] + [("bn254", 1, "msm_reduce.hip"),
                                                             ("bls12_381", 2, FINALIZE_PROBE),
                                                             ("bn254", 1, ASSEMBLE_PROBE)]
    if (curve, group, source) not in units:
        units.append((curve, group, source))
    jobs = []
    for c, g, f in units:
        out = path(c, g, f)
        if os.path.exists(out):
            continue
        try:
            fd = os.open(out + ".lock", os.O_CREAT | os.O_EXCL | os.O_WRONLY)
        except FileExistsError:
            continue                                   # another worker compiles this one
        os.close(fd)
        flags = ["-DDG_CURVE=%d" % CURVE_ID[c], "-DDG_GROUP=%d" % g, "-DDG_NAME=%s_g%d" % (c, g)]
        if f == "msm_reduce.hip" and g == 2:
            flags.append("-DDG29_OUTLINE_MUL")
        src = os.path.join(ROOT, f) if "/" in f else os.path.join(CSRC, f)
        proc = subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC] + flags +
                                ["--cuda-device-only", "-S", src, "-o", out + ".tmp"],
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        jobs.append((proc, out))
    for proc, out in jobs:
        log = proc.communicate()[0]
        try:
            assert proc.returncode == 0, log[-2000:]
            os.replace(out + ".tmp", out)
        finally:
            os.unlink(out + ".lock")
    out = path(curve, group, source)
    for _ in range(900):                               # compiled by another worker
        if os.path.exists(out):
            break
        time.sleep(1)
    assert os.path.exists(out), "assembly of %s %s g%d was not produced" % (source, curve, group)
    return open(out).read()


def limb_shape(p):
    bits = p.bit_length()
    w = 29 if bits <= 256 else 28
    return (bits + 5 + w - 1) // w, w


def run_bucket(text, kernel, curve, group, spec):
    """One lane of workgroup (0, 0): bucket 3 of a 16-bucket window holds the entries `spec` = [(point index, negate)].
    -> (affine sum the kernel wrote or None for the identity, instructions executed)"""
    from oracle.pyref.curves import CURVES
    C = CURVES[curve, "g%d" % group]
    Fq_p = C.F.p
    n_limbs, w = limb_shape(Fq_p)
    nw = (Fq_p.bit_length() + 31) // 32            # packed words per base-field element
    R = 1 << (w * n_limbs)
    ext = group == 2
    prog = E.Program(text, kernel)
    lane = E.Lane(prog)
    npts = 6
    pts = [C.mul(C.gen, i + 1) for i in range(npts)]

    def words(v):
        return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(nw)]

    TAB, OFFS, CNTS, SOFF, STOT, ENT, SSUM, BUCK, KARG = (0x100000 * k for k in range(1, 10))
    pw = 2 * nw * (2 if ext else 1)                # words per table point
    for i, P in enumerate(pts):
        comps = [P[0][0], P[0][1], P[1][0], P[1][1]] if ext else [P[0], P[1]]
        wlist = []
        for c in comps:
            wlist += words(c * R % Fq_p)
        for k, v in enumerate(wlist):
            lane.mem[TAB + 4 * pw * i + 4 * k] = v
    log_nb, bucket = 4, 3
    for b in range(16):
        lane.mem[OFFS + 4 * b] = 0
        lane.mem[CNTS + 4 * b] = len(spec) if b == bucket else 0
        lane.mem[SOFF + 4 * b] = 0 if b <= bucket else 1
    lane.mem[STOT] = 1
    for j, (idx, neg) in enumerate(spec):
        lane.mem[ENT + 4 * j] = idx | (0x80000000 if neg else 0)
    karg = [0] * 36                                                # (.. + the null ClkProbe pointer at 0x88)

    def put64(off, v):
        karg[off // 4], karg[off // 4 + 1] = v & 0xFFFFFFFF, v >> 32

    for k in range(4):                                             # MsmBases: four instance pointers
        put64(8 * k, TAB)
    put64(0x20, npts)                                              # n
    for k, v in enumerate([5, 1, log_nb, 4, 64, 1, 1, 1]):         # MsmGeom: c, nwin, log_nb, seg_log, seg_cap, bw, table, rows
        karg[0x28 // 4 + k] = v
    put64(0x48, npts)                                              # region
    for k, base in enumerate([OFFS, CNTS, SOFF, STOT, ENT, SSUM, BUCK]):
        put64(0x50 + 8 * k, base)
    for k, v in enumerate(karg):
        lane.mem[KARG + 4 * k] = v
    lane.s[0], lane.s[1] = KARG & 0xFFFFFFFF, KARG >> 32          # kernarg segment pointer (the only user SGPR pair)
    lane.s[2], lane.s[3] = 0, 0                                    # workgroup id x, y
    lane.v[0] = 0                                                  # work-item id
    lane.run()
    run_bucket.last_hist = dict(lane.hist)
    ncoord = n_limbs * (2 if ext else 1)
    out = [lane.mem.get(BUCK + 4 * 4 * ncoord * bucket + 4 * k) for k in range(4 * ncoord)]
    assert all(v is not None for v in out), "the bucket was not written"

    def fe(ws):
        return sum(v << (w * i) for i, v in enumerate(ws)) % Fq_p

    F = C.F
    if ext:
        co = [(fe(out[ncoord * c:ncoord * c + n_limbs]), fe(out[ncoord * c + n_limbs:ncoord * (c + 1)])) for c in range(4)]
        zero = (0, 0)
    else:
        co = [fe(out[ncoord * c:ncoord * (c + 1)]) for c in range(4)]
        zero = 0
    if co[2] == zero:
        return None, pts, lane.count
    inv = F.inv if ext else (lambda v: pow(v, Fq_p - 2, Fq_p))
    mul = F.mul if ext else (lambda x, y: x * y % Fq_p)
    return (mul(co[0], inv(co[2])), mul(co[1], inv(co[3]))), pts, lane.count


SPECS = [
    [(1, False), (2, True), (0, False), (4, False), (5, True)],
    [(1, False), (1, False), (3, True)],              # the same point twice: the doubling branch
    [(2, False), (2, True), (4, False)],              # P then -P: the accumulator passes through the identity
    [(3, True)],
    [(0, False), (0, True)],                          # the bucket sums to the identity
]


@pytest.mark.parametrize("curve,group,kernel", CASES)
def test_accumulation_kernel_on_the_cpu(curve, group, kernel):
    from oracle.pyref.curves import CURVES
    C = CURVES[curve, "g%d" % group]
    text = assembly(curve, group, case_source(kernel))
    for spec in SPECS:
        got, pts, count = run_bucket(text, kernel, curve, group, spec)
        exp = None
        for idx, neg in spec:
            exp = C.add(exp, C.neg(pts[idx]) if neg else pts[idx])
        assert got == exp, (curve, group, spec)
        assert count > 300


def test_g1_accumulation_workgroup_with_the_bucket_tree():
    """The whole 256-lane workgroup of the BN254 G1 accumulation (four waves in lockstep, barriers, LDS, ballots: the
    Workgroup emulator): a bucket of 40 entries is cut into three segments held by lanes 0-2, whose partial sums the
    in-workgroup bucket tree must add before lane 0 writes the bucket; a second bucket of 5 entries sits in lane 3.
    Both buckets == the oracle's sums."""
    import random
    from oracle.pyref.curves import CURVES
    curve, group = "bn254", 1
    C = CURVES[curve, "g1"]
    p = C.F.p
    n_limbs, w = limb_shape(p)
    R = 1 << (w * n_limbs)
    prog = E.Program(assembly(curve, group), "msm_accumulate_kernel")
    TAB, OFFS, CNTS, SOFF, STOT, ENT, SSUM, BUCK, KARG = (0x100000 * k for k in range(1, 10))
    wg = E.Workgroup(prog, 256, wg_id=(0, 0), kernarg_addr=KARG)
    mem = wg.mem
    npts = 12
    pts = [C.mul(C.gen, 3 * i + 1) for i in range(npts)]
    for i, P in enumerate(pts):
        for c, coord in enumerate(P):
            v = coord * R % p
            for k in range(8):
                mem[TAB + 64 * i + 32 * c + 4 * k] = (v >> (32 * k)) & 0xFFFFFFFF
    rng = random.Random(3)
    spec = {3: [(rng.randrange(npts), rng.random() < 0.5) for _ in range(40)],
            7: [(rng.randrange(npts), rng.random() < 0.5) for _ in range(5)]}
    log_nb, seg_log = 4, 4
    seg_running, ent_running = 0, 0
    for b in range(16):
        cnt = len(spec.get(b, []))
        mem[CNTS + 4 * b] = cnt
        mem[OFFS + 4 * b] = ent_running
        mem[SOFF + 4 * b] = seg_running
        for j, (idx, neg) in enumerate(spec.get(b, [])):
            mem[ENT + 4 * (ent_running + j)] = idx | (0x80000000 if neg else 0)
        ent_running += cnt
        seg_running += (cnt + (1 << seg_log) - 1) >> seg_log
    mem[STOT] = seg_running
    assert seg_running == 4
    karg = [0] * 36                                                # (.. + the null ClkProbe pointer at 0x88)

    def put64(off, v):
        karg[off // 4], karg[off // 4 + 1] = v & 0xFFFFFFFF, v >> 32

    for k in range(4):
        put64(8 * k, TAB)
    put64(0x20, npts)
    for k, v in enumerate([5, 1, log_nb, seg_log, 64, 1, 1, 1]):
        karg[0x28 // 4 + k] = v
    put64(0x48, 64)                                                # region: room for the 45 entries
    for k, base in enumerate([OFFS, CNTS, SOFF, STOT, ENT, SSUM, BUCK]):
        put64(0x50 + 8 * k, base)
    for k, v in enumerate(karg):
        mem[KARG + 4 * k] = v
    wg.run()
    for b, entries in spec.items():
        out = [mem.get(BUCK + 144 * b + 4 * k) for k in range(36)]
        assert all(v is not None for v in out), "bucket %d was not written" % b
        co = [sum(v << (w * i) for i, v in enumerate(out[9 * c:9 * c + 9])) % p for c in range(4)]
        exp = None
        for idx, neg in entries:
            exp = C.add(exp, C.neg(pts[idx]) if neg else pts[idx])
        got = None if co[2] == 0 else (co[0] * pow(co[2], p - 2, p) % p, co[1] * pow(co[3], p - 2, p) % p)
        assert got == exp, "bucket %d" % b


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_g2_finalize_workgroup(curve):
    """(bls12_381: the 14-limb kernel, BLOCK = 128, compiled alone from tests/isa/finalize_probe.hip.)
    msm_finalize_lds_kernel<Fp2<bn254>, 256> with two lanes per bucket (the throughput finalize behind the G2
    accumulation: each lane adds its share of the bucket's partial sums into an accumulator in LDS columns --
    XYZZ29::add_into with the four-product Y3, ONE addition site for the serial partials and the tree partners --, then
    one tree step across the two lanes behind a barrier) on the Workgroup emulator:
    buckets with 5, 2, 1 and 0 partials, partials in XYZZ form with Z != 1 -> the buckets the oracle's sums predict.
    Round 6: also a bucket whose two partials are THE SAME group element in two representations (the addition's doubling
    branch: what the all-equal-points shape of dmsm/mod.rs:155-159 does to every bucket, and where round 5's form of the
    14-limb kernel used a clobbered spill register as an address) and one whose partials are P and -P (the identity)."""
    import random
    from oracle.pyref.curves import CURVES
    C = CURVES[curve, "g2"]
    F2 = C.F
    p = F2.p
    n_limbs, w = limb_shape(p)
    R = 1 << (w * n_limbs)
    block = 256 if curve == "bn254" else 128
    xb = 4 * 2 * n_limbs * 4                                       # bytes of an XYZZ29 over Fq2
    if curve == "bn254":
        prog = E.Program(assembly("bn254", 2), "msm_finalize_lds_kernelINS_3Fp2INS_2FpINS_15bn254_fq_paramsEEEEELi256EE")
    else:
        prog = E.Program(assembly(curve, 2, FINALIZE_PROBE),
                         "msm_finalize_lds_kernelINS_3Fp2INS_2FpINS_19bls12_381_fq_paramsEEEEELi128EE")
    CNTS, SOFF, SSUM, BUCK, GCNT, GLIST, KARG = (0x100000 * k for k in range(1, 8))
    wg = E.Workgroup(prog, block, wg_id=(0, 0), kernarg_addr=KARG)
    mem = wg.mem
    rng = random.Random(11)
    log_nb, seg_log = 4, 4
    parts = {2: 5, 5: 2, 7: 2, 9: 1, 11: 2, 12: 0}            # bucket -> number of partial sums (segments)
    same, opposite = 7, 11                                    # ... whose two partials are P, P / P, -P
    expect, slot = {}, 0

    def limbs(v):
        v = v * R % p
        return [(v >> (w * i)) & ((1 << w) - 1) for i in range(n_limbs - 1)] + [v >> (w * (n_limbs - 1))]

    for b in range(16):
        k = parts.get(b, 0)
        mem[CNTS + 4 * b] = 16 * k                              # k full segments
        mem[SOFF + 4 * b] = slot
        total = None
        for s in range(k):
            if s == 0 or b not in (same, opposite):
                P = C.mul(C.gen, rng.randrange(1, 1000))
            elif b == opposite:
                P = C.neg(P)
            z = (rng.randrange(1, p), rng.randrange(p))
            zz = F2.sqr(z)
            zzz = F2.mul(zz, z)
            coords = [F2.mul(P[0], zz), F2.mul(P[1], zzz), zz, zzz]
            words = []
            for c in coords:
                words += limbs(c[0]) + limbs(c[1])
            for i, v in enumerate(words):
                mem[SSUM + xb * (slot + s) + 4 * i] = v
            total = C.add(total, P)
        expect[b] = total
        slot += k
    for i in range(4):
        mem[GCNT + 4 * i] = 0
    karg = [0] * 27
    for k, v in enumerate([5, 1, log_nb, seg_log, 64, 1, 1, 1]):    # MsmGeom: c, nwin, log_nb, seg_log, seg_cap, bw, table, rows
        karg[k] = v

    def put64(off, v):
        karg[off // 4], karg[off // 4 + 1] = v & 0xFFFFFFFF, v >> 32

    put64(0x20, 64)                                                # region
    put64(0x28, 16)                                                # total buckets
    karg[0x30 // 4] = 0                                            # wg_log: one slot per segment (no tree in the G2 accumulation)
    karg[0x34 // 4] = 1                                            # lpb_log: two lanes per bucket
    for k, base in enumerate([CNTS, SOFF, SSUM, BUCK, GCNT, GLIST]):
        put64(0x38 + 8 * k, base)
    karg[0x68 // 4] = 16                                           # giant_cap
    for k, v in enumerate(karg):
        mem[KARG + 4 * k] = v
    wg.run()
    for b, k in parts.items():
        if k == 1:
            continue                                               # a one-segment bucket is written by the accumulation itself
        out = [mem.get(BUCK + xb * b + 4 * i) for i in range(8 * n_limbs)]
        assert all(v is not None for v in out), "bucket %d was not written" % b

        def fe(ws):
            return sum(v << (w * i) for i, v in enumerate(ws)) % p

        nn = n_limbs
        co = [(fe(out[2 * nn * c:2 * nn * c + nn]), fe(out[2 * nn * c + nn:2 * nn * (c + 1)])) for c in range(4)]
        got = None if co[2] == (0, 0) else (F2.mul(co[0], F2.inv(co[2])), F2.mul(co[1], F2.inv(co[3])))
        assert got == expect[b], "bucket %d" % b


@pytest.mark.parametrize("group,chunked", [(1, False), (1, True)] + ([(2, False), (2, True)] if os.environ.get("DG16_EMU_ALL") else []))
def test_bucket_reduction_workgroups(group, chunked):
    """The bucket reduction of a window on the Workgroup emulator: msm_row_kernel (one 256-lane workgroup per row of 256
    buckets: suffix scan + tree over LDS, XYZZ29::add_mem) for both rows of a 512-bucket window, then msm_top_kernel
    (512 lanes: sum W and sum r R on its two halves, doublings) -> the window sum (internal form) == sum_b (b + 1) B_b
    from the oracle.  A dozen buckets are occupied, in XYZZ form with Z != 1.
    chunked: msm_rowchunk_kernel instead of the rows (the plain MSM's path: every lane owns two consecutive buckets,
    serial running sums, one suffix scan, two trees side by side) and the top kernel in its folded mode.
    G2 (DG16_EMU_ALL=1): the same kernels with the field products behind calls (DG29_OUTLINE_MUL, as the Makefile builds)."""
    import random
    from oracle.pyref.curves import CURVES
    C = CURVES["bn254", "g%d" % group]
    ext = group == 2
    p = C.F.p
    n_limbs, w = limb_shape(p)
    R = 1 << (w * n_limbs)
    text = assembly("bn254", group, "msm_reduce.hip")
    BUCK, ROWW, ROWR, FOLD, WSUM, KARG = (0x100000 * k for k in range(1, 7))
    rng = random.Random(5)
    log_nb = 9
    ncomp = 2 if ext else 1
    pt_words = 4 * 9 * ncomp
    occupied = sorted(rng.sample(range(512), 10) + [0, 511])
    mem0 = {}
    expect = None
    for b in range(512):
        for i in range(pt_words):
            mem0[BUCK + 4 * pt_words * b + 4 * i] = 0               # all-zero limbs: the identity
    F = C.F
    mul = F.mul if ext else (lambda x, y: x * y % p)
    for b in occupied:
        P = C.mul(C.gen, rng.randrange(1, 10**6))
        z = (rng.randrange(1, p), rng.randrange(p)) if ext else rng.randrange(1, p)
        zz = mul(z, z)
        zzz = mul(zz, z)
        comps = []
        for val in (mul(P[0], zz), mul(P[1], zzz), zz, zzz):
            comps += list(val) if ext else [val]
        for c, val in enumerate(comps):
            v = val * R % p
            for i in range(9):
                mem0[BUCK + 4 * pt_words * b + 36 * c + 4 * i] = (v >> (29 * i)) & ((1 << 29) - 1) if i < 8 else v >> 232
        expect = C.add(expect, C.mul(P, b + 1))
    geom = [10, 1, log_nb, 4, 64, 1, 1, 1]
    # ---- rows
    if chunked:
        wg = E.Workgroup(E.Program(text, "msm_rowchunk_kernel"), 256, wg_id=(0, 0), kernarg_addr=KARG)
        wg.mem.update(mem0)
        karg = [log_nb, 1, BUCK, 0, FOLD, 0]                        # log_nb, k_log, buckets, fold
        for k, v in enumerate(karg):
            wg.mem[KARG + 4 * k] = v
        wg.run()
        for a, v in wg.mem.items():
            if FOLD <= a < FOLD + 0x100000:
                mem0[a] = v
    for row in range(0 if chunked else 2):
        wg = E.Workgroup(E.Program(text, "msm_row_kernel"), 256, wg_id=(row, 0), kernarg_addr=KARG)
        wg.mem.update(mem0)
        karg = [0] * 18
        for k, v in enumerate(geom):
            karg[k] = v
        karg[0x20 // 4] = 64                                        # region (unused here)
        karg[0x28 // 4], karg[0x2c // 4] = 8, 1                     # RowGeom: row_log, rows_log
        for k, base in enumerate([BUCK, ROWW, ROWR]):
            karg[0x30 // 4 + 2 * k], karg[0x30 // 4 + 2 * k + 1] = base, 0
        for k, v in enumerate(karg):
            wg.mem[KARG + 4 * k] = v
        wg.run()
        for a, v in wg.mem.items():
            if ROWW <= a < ROWW + 0x200000:
                mem0[a] = v
    # ---- top
    wg = E.Workgroup(E.Program(text, "msm_top_kernel"), 512, wg_id=(0, 0), kernarg_addr=KARG)
    wg.mem.update(mem0)
    # TopGeom: folded, per_log, lanes, final_log, rows_log (+ padding), then row_w, row_r, fold, window_sums
    karg = ([1, 8, 1, 1, 1, 0] if chunked else [0, 0, 2, 8, 1, 0]) + [0] * 8
    for k, base in enumerate([ROWW, ROWR, FOLD, WSUM]):
        karg[6 + 2 * k] = base
    for k, v in enumerate(karg):
        wg.mem[KARG + 4 * k] = v
    wg.run()
    nout = 4 * 9 * ncomp
    out = [wg.mem.get(WSUM + 4 * i) for i in range(nout)]
    assert all(v is not None for v in out), "the window sum was not written"
    r_inv = pow(R, p - 2, p)
    vals = [sum(v << (29 * i) for i, v in enumerate(out[9 * c:9 * c + 9])) * r_inv % p for c in range(4 * ncomp)]
    if ext:
        co = [(vals[2 * c], vals[2 * c + 1]) for c in range(4)]
        got = (F.mul(co[0], F.inv(co[2])), F.mul(co[1], F.inv(co[3])))
    else:
        got = (vals[0] * pow(vals[2], p - 2, p) % p, vals[1] * pow(vals[3], p - 2, p) % p)
    assert got == expect


def test_horner_tail_on_one_wave():
    """msm_tail_kernel (G1 of BN254) on the Workgroup emulator: one wave runs the Horner chain over the window sums with
    the wave-cooperative operations on the reduced-radix types (msm_impl.h: dbl_wave29 / add_wave29 -- a dependency level
    is one product per lane, the slots are joined with v_mov_b32_dpp row_newbcast): 2^c (2^c S_2 + S_1) + S_0 == the oracle's, for window
    sums in XYZZ form with Z != 1, one of them the identity's neighbour case S_1 = S_2 (the addition's doubling branch)."""
    import random
    from oracle.pyref.curves import CURVES
    C = CURVES["bn254", "g1"]
    p = C.F.p
    n_limbs, w = limb_shape(p)
    R = 1 << (w * n_limbs)
    text = assembly("bn254", 1, "msm_group.hip")      # (instantiated with the accumulation: inline products for every group)
    WSUM, OUT, KARG = 0x100000, 0x200000, 0x300000
    rng = random.Random(11)
    c_bits, bw = 1, 3
    P2 = C.mul(C.gen, rng.randrange(1, 10**6))
    pts = [C.mul(C.gen, rng.randrange(1, 10**6)), C.add(P2, P2), P2]      # S_0, S_1 = 2 S_2, S_2: 2 S_2 + S_1 doubles
    mem0 = {}
    for k, P in enumerate(pts):
        z = rng.randrange(1, p)
        zz = z * z % p
        zzz = zz * z % p
        for cidx, val in enumerate((P[0] * zz % p, P[1] * zzz % p, zz, zzz)):
            v = val * R % p
            for i in range(9):
                mem0[WSUM + 4 * (36 * k + 9 * cidx + i)] = (v >> (29 * i)) & ((1 << 29) - 1) if i < 8 else v >> 232
    expect = None
    for P in reversed(pts):
        for _ in range(c_bits):
            expect = C.add(expect, expect) if expect is not None else None
        expect = C.add(expect, P)
    wg = E.Workgroup(E.Program(text, "msm_tail_kernel"), 64, wg_id=(0, 0), kernarg_addr=KARG)
    wg.mem.update(mem0)
    karg = [WSUM, 0] + [c_bits, bw, 1, 4, 64, bw, 0, 1, 64, 0] + [0, 0] + [OUT, 0]   # ptr | MsmGeom (+ region) | affine, pad | ptr
    for k, v in enumerate(karg):
        wg.mem[KARG + 4 * k] = v
    wg.run()
    out = [wg.mem.get(OUT + 4 * i) for i in range(24)]
    assert all(v is not None for v in out), "the result was not written"
    r32_inv = pow(1 << 256, p - 2, p)
    X, Y, Z = (sum(v << (32 * i) for i, v in enumerate(out[8 * k:8 * k + 8])) * r32_inv % p for k in range(3))
    zi = pow(Z, p - 2, p)
    assert (X * zi * zi % p, Y * zi * zi * zi % p) == expect


@pytest.mark.parametrize("curve,n_shards", [("bn254", 1), ("bn254", 3)] +
                         ([("bn254", 8), ("bls12_381", 2)] if os.environ.get("DG16_EMU_ALL") else []))
def test_proof_assembly_workgroup(curve, n_shards):
    """prover_assemble_kernel (prover_impl.h; compiled alone from tests/isa/assemble_probe.hip) on the Workgroup emulator:
    seven waves in two workgroups sum the gathered records of n_shards shards per slot -- L, H, s A, r B1 | A and the
    two halves of B (G2) -- on the reduced-radix wave-cooperative operations with the products behind a call, one
    barrier, then B's other half and C = L + H + s A + r B1.  Records are Jacobian points in arkworks form with Z != 1; one record slot holds the
    identity, one pair of slots the same point (the addition's doubling branch).  (A, B, C) == the oracle's sums."""
    import random
    from oracle.pyref.curves import CURVES
    G1, G2 = CURVES[curve, "g1"], CURVES[curve, "g2"]
    F2 = G2.F
    p = G1.F.p
    nw = (p.bit_length() + 31) // 32
    R32 = 1 << (32 * nw)
    g1j, g2j = 3 * nw * 4, 6 * nw * 4
    rec_bytes = 6 * g1j + g2j
    text = assembly(curve, 1, ASSEMBLE_PROBE)
    GATH, OUTA, OUTB, OUTC, KARG = (0x100000 * k for k in range(1, 6))
    mem = {}
    rng = random.Random(5 + n_shards)

    def words(v):
        v = v * R32 % p
        return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(nw)]

    def put_g1(addr, P):
        if P is None:
            ws = words(1) + words(1) + [0] * nw
        else:
            z = rng.randrange(1, p)
            ws = words(P[0] * z * z % p) + words(P[1] * z * z * z % p) + words(z)
        for i, v in enumerate(ws):
            mem[addr + 4 * i] = v

    def put_g2(addr, P):
        z = (rng.randrange(1, p), rng.randrange(p))
        zz = F2.sqr(z)
        co = [F2.mul(P[0], zz), F2.mul(P[1], F2.mul(zz, z)), z]
        ws = []
        for c in co:
            ws += words(c[0]) + words(c[1])
        for i, v in enumerate(ws):
            mem[addr + 4 * i] = v

    sums = [None] * 7
    same = G1.mul(G1.gen, 77)
    for k in range(n_shards):
        for slot in range(6):
            P = G1.mul(G1.gen, rng.randrange(1, 10 ** 6))
            if slot == 3 and k == 0:
                P = None                                            # H of shard 0: the identity (Z = 0)
            if slot in (4, 5) and k == 0:
                P = same                                            # s A == r B1 on shard 0: C's last addition may double
            put_g1(GATH + k * rec_bytes + slot * g1j, P)
            sums[slot] = G1.add(sums[slot], P) if P is not None else sums[slot]
        P2 = G2.mul(G2.gen, rng.randrange(1, 10 ** 6))
        put_g2(GATH + k * rec_bytes + 6 * g1j, P2)
        sums[6] = G2.add(sums[6], P2)
    exp_c = None
    for slot in (2, 3, 4, 5):
        exp_c = G1.add(exp_c, sums[slot]) if sums[slot] is not None else exp_c
    karg = []
    for v in (GATH, n_shards, rec_bytes, OUTA, OUTB, OUTC):
        karg += [v & 0xFFFFFFFF, v >> 32]
    for k, v in enumerate(karg):
        mem[KARG + 4 * k] = v
    prog = E.Program(text, "prover_assemble_kernel")
    for block in (0, 1):                                            # block 0: C's four chains; block 1: A and B's halves
        wg = E.Workgroup(prog, 256, wg_id=(block, 0), kernarg_addr=KARG)
        wg.mem.update(mem)
        wg.run()
        mem.update({a: v for a, v in wg.mem.items() if OUTA <= a < KARG})
    r32_inv = pow(R32, p - 2, p)

    def fe(addr, k):
        out = [mem.get(addr + 4 * (nw * k + i)) for i in range(nw)]
        assert all(v is not None for v in out), "the proof element was not written"
        v = sum(x << (32 * i) for i, x in enumerate(out))
        assert v < p, "not canonical"
        return v * r32_inv % p

    def g1_out(addr):
        X, Y, Z = fe(addr, 0), fe(addr, 1), fe(addr, 2)
        zi = pow(Z, p - 2, p)
        return (X * zi * zi % p, Y * zi * zi * zi % p)

    assert g1_out(OUTA) == sums[0]
    assert g1_out(OUTC) == exp_c
    X, Y, Z = ((fe(OUTB, 2 * k), fe(OUTB, 2 * k + 1)) for k in range(3))
    zi = F2.inv(Z)
    zi2 = F2.sqr(zi)
    assert (F2.mul(X, zi2), F2.mul(Y, F2.mul(zi2, zi))) == sums[6]
