"""CPU: compile-time resource figures of the hot kernels, from the reports hipcc writes during `make`
(csrc/*.usage.txt, -Rpass-analysis=kernel-resource-usage).  A regression here is a performance bug that no parity
test sees: the G2 bucket kernel once spilled ~200 dwords per lane (31.7 GB of HBM traffic per launch instead of
2.7 GB), and an out-of-line call in the hot loop once kept the loaded point in scratch memory."""

import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import resource_report  # noqa: E402

REP = {resource_report.short(k): v for k, v in resource_report.load().items()}
pytestmark = pytest.mark.skipif(not REP, reason="no csrc/*.usage.txt: run `make -C distributed-groth16_amd/csrc` first")


def find(prefix):
    hits = {k: v for k, v in REP.items() if k.replace(" ", "").startswith(prefix.replace(" ", ""))}
    assert hits, "kernel %s not in the build" % prefix
    return hits


def test_g1_bucket_accumulation_keeps_four_waves_with_the_bucket_tree():
    """The in-workgroup bucket tree (a full XYZZ addition on LDS columns) must not cost the accumulation LOOP its
    occupancy: 128 VGPRs = four waves per SIMD for BN254; the only scratch is the call frame of the out-of-line
    doubling (XYZZ29::dbl_cold: two 144-byte points + the return address), on the equal-operands branch."""
    (r,) = find("msm_accumulate_kernel<bn254_fq,256>").values()
    assert r["vgprs"] <= 128 and r["agprs"] == 0 and r["occupancy"] >= 4
    assert r["scratch"] <= 2 * 144 + 32
    assert r["lds"] == 4 * 9 * 256 * 4 + 256 * 2 + 5 * 4      # partial columns + compaction list + wave counters
    # 48-byte fields: no tree (measured slower), no scratch; round 4: the fused Y3 (one reduction less per addition) at
    # TWO waves per SIMD -- 178 VGPRs -- measured 4 % ahead of the unfused loop at three (DG16_ACC48_WAVES)
    for k, r in find("msm_accumulate_kernel<bls12_").items():
        assert r["occupancy"] >= 2 and r["vgprs"] <= 192 and r["scratch"] == 0 and r["lds"] == 0, k


def test_g2_bucket_accumulation_stages_through_lds_without_spilling():
    (r,) = find("msm_accumulate_lds_kernel<Fp2<bn254_fq>,256,0>").values()
    assert r["lds"] == 4 * 18 * 256 * 4 and r["occupancy"] == 2   # 4 coordinates x 2 x 9 limbs x 256 lanes: 2 blocks / CU
    assert r["scratch"] == 0 and r["agprs"] == 0
    # the 14-limb G2 accumulation: a step loop over three product sites with its four temporaries in accumulation
    # registers a[144..255] (declared clobbered: AGPRs = 256), one wave per SIMD, no scratch
    for curve in ("bls12_381", "bls12_377"):
        (r,) = find("msm_accumulate_steps_kernel<Fp2<%s_fq>,128>" % curve).values()
        assert r["scratch"] == 0 and r["agprs"] == 256 and r["vgprs"] <= 256 and r["lds"] == 4 * 28 * 128 * 4
    # the throughput finalize of G2 (two lanes per bucket, add_into with the four-product Y3): two waves per SIMD, and
    # since the products are chains neither scratch (BN254: was 16 B) nor a full AGPR file + scratch (BLS12-377)
    for k, r in find("msm_finalize_lds_kernel<Fp2<bn254_fq>,256>").items():
        assert r["occupancy"] == 2 and r["scratch"] == 0 and r["agprs"] == 0, k
    # 14-limb G2: the addition is a step loop again (xyzz_add_into_steps; round 6) with its temporaries in a[144..255]: the
    # kernel is allocated the whole file, hipcc's own spills stay below it (csrc/Makefile checks the assembly) and what does
    # not fit there is a few words of scratch -- round 5's form had NO scratch because sixteen of the compiler's spills sat
    # inside the file (the aperture violation)
    for k, r in find("msm_finalize_lds_kernel<Fp2<bls12_").items():
        assert r["scratch"] <= 96 and r["agprs"] == 256 and r["vgprs"] <= 256, (k, r)


def test_proof_assembly_keeps_its_chains_in_registers():
    """prover_assemble_kernel (the exposed tail of every proof: seven chains of wave-cooperative additions in two
    workgroups of four waves): no scratch beyond the call frame of its one out-of-line product per field -- as ONE
    workgroup of seven waves the 14-limb curves had 256 registers per wave and spilled 536 B per lane."""
    for k, r in find("prover_assemble_kernel<").items():
        assert r["scratch"] <= 16 and r["lds"] <= 2048, k


def test_ntt_and_sort_kernels_are_register_and_lds_only():
    for k, r in find("ntt_step_kernel<").items():
        # 1024 tile elements of 9 limbs + 512 staged twiddles of 8 packed words: three workgroups per CU
        assert r["scratch"] == 0 and r["lds"] == 1024 * 9 * 4 + 512 * 8 * 4 and r["occupancy"] >= 3, k
        assert r["vgprs"] <= 80, k          # 72 with one accumulator per product (94-101 with hipcc's 17 column chains)
    for name in ("msm_part_hist_kernel<", "msm_part_scatter_kernel<", "msm_part_count_kernel<", "msm_part_place_kernel<",
                 "msm_digits_kernel<", "msm_scatter_kernel<"):
        for k, r in find(name).items():
            assert r["scratch"] == 0 and r["vgprs"] <= 64, k
    (r,) = find("msm_part_place_kernel<").values()
    assert r["lds"] == 3 * 4096 * 4                          # bin counts, bucket ranks, bucket destinations


def test_lane_form_reduction_kernels_fit_their_workgroups():
    """msm_lane_reduce_kernel runs 2^RB waves per workgroup -- 16 for the nine-limb base field, 8 for the others -- so
    its registers must leave that many waves on one compute unit (4 per SIMD at <= 128 VGPRs, 2 at <= 256), without
    scratch: every step of its chains is a dependent product, a spill would sit on all of them.  The one-wave form and the
    Horner tails of the nine-limb groups likewise stay in registers."""
    (r,) = find("msm_lane_reduce_kernel<bn254_fq,4>").values()
    assert r["vgprs"] <= 128 and r["scratch"] == 0 and r["agprs"] == 0
    for name in ("msm_lane_reduce_kernel<Fp2<bn254_fq>,3>", "msm_lane_reduce_kernel<bls12_381_fq,3>",
                 "msm_lane_reduce_kernel<bls12_377_fq,3>", "msm_lane_reduce_kernel<Fp2<bls12_381_fq>,3>",
                 "msm_lane_reduce_kernel<Fp2<bls12_377_fq>,3>"):
        (r,) = find(name).values()
        assert r["vgprs"] <= 256 and r["scratch"] == 0, (name, r)
    for k, r in find("msm_lane_reduce_serial_kernel<").items():
        assert r["scratch"] == 0 and r["vgprs"] <= 256, (k, r)
    (r,) = find("msm_tail_kernel<bn254_fq>").values()
    assert r["scratch"] == 0 and r["vgprs"] <= 128
    # the 14-limb G2 tail was 256 VGPRs + ~140 AGPRs + 160 B of scratch in the one-product-per-lane form
    for curve in ("bls12_381", "bls12_377"):
        (r,) = find("msm_tail_kernel<Fp2<%s_fq>>" % curve).values()
        assert r["vgprs"] <= 200 and r["agprs"] == 0 and r["scratch"] <= 128, r
