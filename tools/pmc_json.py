"""pmc_raw.txt (tools/pmc_sum.py lines of the FETCH_SIZE and WRITE_SIZE passes) -> the JSON bench.py reads
`roofline.traffic` from.  usage: python tools/pmc_json.py pmc_raw.txt <tag> [curve: bn254 | bls12_381]"""
import json
import sys

raw = {}
for ln in open(sys.argv[1]):
    f = ln.split()
    if len(f) == 6 and f[2] == "launches":
        raw.setdefault(f[1], {})[f[0]] = {"launches": int(f[3]), "KB_per_launch": float(f[5])}
curve = sys.argv[3] if len(sys.argv) > 3 else "bn254"
if curve != "bn254":
    # BLS12-381 (config 5's curve): the G2 accumulation is msm_accumulate_steps_kernel (192-byte points: twelve 16-byte
    # loads per lane, wide requests -> the guide's x2), the G1 accumulation gathers 96-byte points.  Calibration of the G1
    # width against the bytes the kernel MUST gather is printed next to the raw figures (`g1_expected_gather_bytes`).
    g2s = raw.get("msm_accumulate_steps_kernel", {})
    g1s = raw.get("msm_accumulate_kernel", {})
    f2, w2 = g2s.get("FETCH_SIZE", {}).get("KB_per_launch", 0.0), g2s.get("WRITE_SIZE", {}).get("KB_per_launch", 0.0)
    f1, w1 = g1s.get("FETCH_SIZE", {}).get("KB_per_launch", 0.0), g1s.get("WRITE_SIZE", {}).get("KB_per_launch", 0.0)
    n_pts, nwin = 1048578, 15            # 2^20 + 2 points, c = 17: ceil(256 / 17) = 16 digits, the top one almost always zero
    print(json.dumps({
        "tag": sys.argv[2], "curve": curve,
        "kernel": "msm_accumulate_steps_kernel<Fp2<bls12_381_fq>>",
        "workload": "BLS12-381 Groth16 proof at 2^20 (tools/shard_timing.py 20 3 bls12_381 1): the G2 bucket accumulation "
                    "in table mode (c = 17, 1 048 578 points)",
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- "
                   "python tools/shard_timing.py 20 3 bls12_381 1; tools/pmc_sum.py (tools/evidence_run.sh)",
        "correction": "gfx950: FETCH_SIZE x2 for the G2 kernel's 192-byte point gathers (wide requests tallied at 64 B, the "
                      "guide's HBM section); the G1 kernel's 96-byte gathers: see g1_expected_gather_bytes next to the raw "
                      "figure; WRITE_SIZE as is; KB = 1024 B",
        "FETCH_SIZE_KB_raw_per_launch": f2, "WRITE_SIZE_KB_raw_per_launch": w2,
        "traffic_bytes_per_launch": (2 * f2 + w2) * 1024,
        "expected_gather_bytes": (192 + 4) * n_pts * nwin,
        "launches": g2s.get("FETCH_SIZE", {}).get("launches", 0),
        "g1_kernel": "msm_accumulate_kernel<Fp<bls12_381_fq>> (A / B1 / L / H: four launches per proof)",
        "g1_FETCH_SIZE_KB_raw_per_launch": f1, "g1_WRITE_SIZE_KB_raw_per_launch": w1,
        "g1_expected_gather_bytes": (96 + 4) * n_pts * nwin,
        # x2 iff the raw figure is below what the kernel must gather (then the requests were tallied at half their size)
        "g1_fetch_factor": 2 if f1 * 1024 < 0.8 * (96 + 4) * n_pts * nwin else 1,
        "g1_traffic_bytes_per_launch": ((2 if f1 * 1024 < 0.8 * (96 + 4) * n_pts * nwin else 1) * f1 + w1) * 1024,
        "other_kernels_raw": raw,
    }, indent=1))
    sys.exit(0)
g2 = raw.get("msm_accumulate_lds_kernel", {})
fetch = g2.get("FETCH_SIZE", {}).get("KB_per_launch", 0.0)
write = g2.get("WRITE_SIZE", {}).get("KB_per_launch", 0.0)
print(json.dumps({
    "tag": sys.argv[2] if len(sys.argv) > 2 else "",
    "kernel": "msm_accumulate_lds_kernel<Fp2<bn254_fq>>",
    "workload": "BN254 Groth16 proof at 2^20 (tools/shard_timing.py 20 3 bn254 1): the G2 bucket accumulation in table "
                "mode (c = 17, 15 windows, 1 048 578 points, balanced segments of <= 16 entries)",
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- "
               "python tools/shard_timing.py 20 3 bn254 1; tools/pmc_sum.py (tools/evidence_run.sh)",
    "correction": "gfx950: FETCH_SIZE x2 for this kernel's 128-byte point gathers (the guide: requests are tallied at "
                  "64 B); WRITE_SIZE taken as is; KB = 1024 B",
    "FETCH_SIZE_KB_raw_per_launch": fetch,
    "WRITE_SIZE_KB_raw_per_launch": write,
    "traffic_bytes_per_launch": (2 * fetch + write) * 1024,
    "launches": g2.get("FETCH_SIZE", {}).get("launches", 0),
    # G1 accumulation (msm_accumulate_kernel): 64-byte point gathers.  Calibration of the counter for THIS width: the
    # kernel must gather 64 B x 15 windows x 1 048 578 points = 1.007 GB of points + 63 MB of entry indices per launch;
    # raw FETCH_SIZE reads 1.14-1.17 GB -- the requests are tallied at their true 64 bytes, NO x2 (the x2 of the guide
    # applies to the 128-byte gathers of the G2 kernel: raw 1.12 GB for 2.01 GB of points)
    "g1_kernel": "msm_accumulate_kernel<Fp<bn254_fq>> (A / B1 / L / H: four launches per proof)",
    "g1_FETCH_SIZE_KB_raw_per_launch": raw.get("msm_accumulate_kernel", {}).get("FETCH_SIZE", {}).get("KB_per_launch", 0.0),
    "g1_WRITE_SIZE_KB_raw_per_launch": raw.get("msm_accumulate_kernel", {}).get("WRITE_SIZE", {}).get("KB_per_launch", 0.0),
    "g1_traffic_bytes_per_launch": (raw.get("msm_accumulate_kernel", {}).get("FETCH_SIZE", {}).get("KB_per_launch", 0.0) +
                                    raw.get("msm_accumulate_kernel", {}).get("WRITE_SIZE", {}).get("KB_per_launch", 0.0)) * 1024,
    "other_kernels_raw": {k: v for k, v in raw.items() if k != "msm_accumulate_lds_kernel"},
}, indent=1))
