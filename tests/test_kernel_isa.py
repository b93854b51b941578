"""CPU: the hot blocks of the accumulation kernels, read from the objects `make` built (tools/isa_report.py: device
bundle -> llvm-objdump -> basic blocks).  hipcc's resource report (tests/test_kernel_resources.py) gives a kernel's
scratch SIZE; whether a spill sits on the path every lane runs sixteen times per segment, or in the doubling branch no
lane takes, is only visible in the code.  The blocks are identified by their v_mad_u64_u32 count, which the formulas fix:
a mixed addition is [U2, S2 = two products] then [everything else], N limbs -> product 2 N^2, square N (N + 1) / 2 +
N^2, reduction alone N^2 (bench.add_mads has the same arithmetic for the roofline)."""

import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_report  # noqa: E402

CSRC = os.path.join(ROOT, "distributed-groth16_amd", "csrc")
pytestmark = pytest.mark.skipif(not (os.path.exists(isa_report.OBJDUMP) and os.path.exists(os.path.join(CSRC, "msm_bn254_g1.o"))),
                                reason="needs llvm-objdump and the built objects (make -C distributed-groth16_amd/csrc)")


def kernel(obj, needle):
    ks = isa_report.kernels(os.path.join(CSRC, obj))
    hits = [v for k, v in ks.items() if needle in k]
    assert len(hits) == 1, (obj, needle, [k for k in ks if needle in k])
    return hits[0]


def block_with(blks, mads):
    hits = [b for b in blks if b["mads"] == mads]
    assert hits, "no block with %d v_mad_u64_u32 (the formulas or the compiler's block layout changed): %s" % (
        mads, sorted({b["mads"] for b in blks if b["mads"] > 100}))
    return hits


def counts(n, fused):
    prod, sqr, red = 2 * n * n, n * (n + 1) // 2 + n * n, n * n
    g1 = 8 * prod + 2 * sqr - (red if fused else 0)
    g2 = 8 * 3 * prod + 2 * 2 * prod - (2 * red if fused else 0)
    return prod, g1, g2


def test_bn254_g1_accumulation_hot_path():
    prod, g1, _ = counts(9, True)
    assert g1 == 1467
    blks = kernel("msm_bn254_g1.o", "msm_accumulate_kernel")
    head, rest = block_with(blks, 2 * prod)[0], block_with(blks, g1 - 2 * prod)[0]
    for b in (head, rest):
        assert b["scratch"] == 0 and b["lds"] == 0 and b["lshl_add_u64"] == 0, b
    # 1467 mads in ~2090 instructions: the products as chains, the fused Y3, the one-compare zero test (round 3's
    # kernel before them: ~2400 with 1548 mads)
    assert head["instr"] + rest["instr"] <= 2200
    # the tree's full addition on LDS columns (12 products + 2 squares - 1 reduction = 2115 mads: [U1, U2, S1, S2] then
    # the rest) does not spill either
    sqr, red = 9 * 10 // 2 + 81, 81
    full = 12 * prod + 2 * sqr - red
    assert full == 2115
    for b in block_with(blks, 4 * prod) + block_with(blks, full - 4 * prod):
        assert b["scratch"] == 0, b


def test_bn254_g2_accumulation_hot_path():
    prod, _, g2 = counts(9, True)
    assert g2 == 4374
    blks = kernel("msm_bn254_g2.o", "msm_accumulate_lds_kernel")
    head, rest = block_with(blks, 2 * 3 * prod)[0], block_with(blks, g2 - 2 * 3 * prod)[0]
    for b in (head, rest):
        assert b["scratch"] == 0 and b["vmem"] == 0, b
    assert head["instr"] + rest["instr"] <= 6100           # (6525 with 4536 mads before the chains and the fused Y3)
    assert sum(b["scratch"] for b in blks) == 0


@pytest.mark.parametrize("curve", ["bls12_381", "bls12_377"])
def test_48_byte_g1_accumulation_hot_path(curve):
    prod, g1, _ = counts(14, True)          # (round 4: the fused Y3 for the 14-limb fields too, at two waves per SIMD)
    blks = kernel("msm_%s_g1.o" % curve, "msm_accumulate_kernel")
    head, rest = block_with(blks, 2 * prod)[0], block_with(blks, g1 - 2 * prod)[0]
    for b in (head, rest):
        assert b["scratch"] == 0 and b["lshl_add_u64"] == 0, b
    assert sum(b["scratch"] for b in blks) == 0


def test_ntt_step_has_no_scratch_and_no_column_joins():
    for needle in ("bn254_fr", "bls12_381_fr"):
        blks = kernel("ntt.o", "ntt_step_kernelINS_2FpINS_%d%s" % (len(needle) + 7, needle))
        s = isa_report.summary(blks)
        assert s["scratch"] == 0
        # v_lshl_add_u64 is now address arithmetic only: a product no longer joins its columns with 64-bit additions
        # (443 in this kernel before the chains)
        assert s["lshl_add_u64"] <= 260 and s["mads"] >= 17 * 162   # (229-240 with the direct twiddle tables: two more address chains)


@pytest.mark.parametrize("curve", ["bls12_381", "bls12_377"])
def test_14_limb_g2_accumulation_is_a_step_loop_over_three_product_sites(curve):
    """msm_accumulate_steps_kernel: ONE product site (2 dual products = 6 N^2 mads), ONE squaring site (4 N^2) and ONE
    fused Y3 site (2 x 5 N^2) inside the step loop -- not eleven inlined products (round 4: a 100-KB loop against a
    64-KB instruction cache).  The loop must fit that cache, and the temporaries' accumulation registers
    (a[kAccFileBase ..]: fixed numbers in asm statements, invisible to the compiler) must not collide with AGPRs
    hipcc allocates itself."""
    import re
    n = 14
    blks = kernel("msm_%s_g2.o" % curve, "msm_accumulate_steps_kernel")
    mul, sqr = block_with(blks, 6 * n * n), block_with(blks, 4 * n * n)
    # (BLS12-377, u^2 = -5: the multiplications by 5 of the fused site are 13 more v_mad_u64_u32)
    fused = [b for b in blks if 10 * n * n <= b["mads"] <= 10 * n * n + 2 * n]
    assert len(mul) == 1 and len(sqr) == 1 and len(fused) == 1, "a site was cloned"
    assert sum(b["scratch"] for b in blks) == 0
    # everything in front of the doubling branch (the one big cold block: dbl_affine, 3 products + 4 squares + the fused Y3)
    cold = max(blks, key=lambda b: b["instr"])
    assert cold["mads"] > 6000 and cold["addr"] > fused[0]["addr"]
    first = min(b["addr"] for b in blks)
    assert cold["addr"] - first <= 62 * 1024, "the step loop no longer fits the instruction cache: %d bytes" % (cold["addr"] - first)
    # The temporaries' register file.  Round 5 classified AGPR references in the DISASSEMBLY (any v_accvgpr_write / read
    # of a[>= 144] was taken for acc_set / acc_get) and so could not see hipcc's own spills into the file -- the bug behind
    # round 5's HSA aperture violation.  The build now checks the compiler's ASSEMBLY, where asm
    # statements are bracketed (tools/check_agpr_file.py, run by csrc/Makefile on every msm_group object before the
    # object takes its place) and leaves the report read here.
    rep = os.path.join(CSRC, "msm_%s_g2.agpr.txt" % curve)
    assert os.path.exists(rep), "csrc/Makefile did not leave %s: the AGPR-file check is not part of the build" % rep
    rows = [ln.split() for ln in open(rep) if "msm_accumulate_steps_kernel" in ln]
    assert len(rows) == 1, rows
    _, _, asm_range, _, cc_max = rows[0]
    assert asm_range == "144-255", asm_range                   # four slots of 2 x 14 limbs
    assert cc_max == "-" or int(cc_max) < 144, "hipcc allocated a[%s] inside the temporaries' file" % cc_max


def test_agpr_file_checker_tells_the_compilers_registers_from_the_asm_statements(tmp_path):
    """tools/check_agpr_file.py on hand-made assembly: a spill of the compiler's own into a[144 ..] is a collision only in a
    kernel whose asm statements use the file, whatever the operand spelling; asm-bracketed accesses never are."""
    import check_agpr_file as chk
    ok = tmp_path / "ok.s"
    ok.write_text("""
_Zkernel_a:
	v_accvgpr_write_b32 a12, v3
	;;#ASMSTART
	v_accvgpr_write_b32 a150, v6
	;;#ASMEND
	;;#ASMSTART
	v_accvgpr_read_b32 v7, a[0xff]
	;;#ASMEND
	v_accvgpr_read_b32 v3, a143
_Zkernel_b:
	v_accvgpr_write_b32 a200, v1            ; no asm statement touches the file here: the compiler may
""")
    assert chk.collisions(str(ok)) == {}
    bad = tmp_path / "bad.s"
    bad.write_text("""
_Zkernel_a:
	;;#ASMSTART
	v_accvgpr_write_b32 a[0x90], v65
	;;#ASMEND
	v_add_u32_e32 v6, 0x2a00, v4
	v_accvgpr_write_b32 a154, v6
	v_accvgpr_read_b32 v188, a154
	v_mfma_f32_32x32x2_f32 a[150:165], v0, v1, a[150:165]
""")
    hit = chk.collisions(str(bad))
    assert list(hit) == ["_Zkernel_a"] and hit["_Zkernel_a"][0] == 150 and 154 in hit["_Zkernel_a"] and hit["_Zkernel_a"][-1] == 165


def test_the_makefile_cannot_build_an_msm_object_without_the_agpr_check():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    rule = mk[mk.index("define PAIR_RULES"):mk.index("endef")]
    group = rule[rule.index("msm_$(3).o:"):rule.index("msm_red_$(3).o:")]
    lines = [ln.strip() for ln in group.splitlines()[1:] if ln.strip()]
    assert "-save-temps=obj" in lines[1] and "-o .st_$(3)/msm_$(3).o" in lines[1]      # compiled aside ...
    assert lines[2].startswith("python3 $$(AGPR_CHECK) .st_$(3)/")                      # ... checked ...
    assert lines[3].startswith("mv .st_$(3)/msm_$(3).o $$@")                            # ... and only then put in place


def test_round5_register_declaration_is_caught_by_the_build_check(tmp_path):
    """The negative control of DESIGN.md section 7.2, without a GPU: the 14-limb G2 finalize compiled the way round 5
    declared its accumulation-register file (-DDG16_ACC_CLOBBER_R5: two registers named in one clobber list, nothing on the
    writes) puts hipcc's own spills inside a[144..255] -- tools/check_agpr_file.py, which the Makefile runs on every
    msm_group object, must refuse it -- and the shipped declaration (every write names its register) must pass."""
    import subprocess
    import check_agpr_file as chk
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("needs hipcc")
    probe = os.path.join(ROOT, "tests", "isa", "finalize_probe.hip")
    outs = {}
    procs = []
    for name, extra in (("r5", ["-DDG16_ACC_CLOBBER_R5"]), ("shipped", [])):
        outs[name] = str(tmp_path / (name + ".s"))
        procs.append(subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-DDG_CURVE=1",
                                       "-DDG_GROUP=2", "-DDG_NAME=bls12_381_g2", "--cuda-device-only", "-S", probe, "-o",
                                       outs[name]] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for pr in procs:
        log = pr.communicate()[0]
        assert pr.returncode == 0, log[-2000:]
    bad = chk.collisions(outs["r5"])
    assert len(bad) == 1 and "msm_finalize_lds_kernel" in list(bad)[0], bad
    regs = list(bad.values())[0]
    assert regs[0] == 144 and len(regs) >= 8, regs            # round 6 found a[144..159] there
    assert chk.collisions(outs["shipped"]) == {}
