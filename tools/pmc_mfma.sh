#!/bin/bash
# MFMA utilisation + effective clock of the encoder kernels (rocprofv3 PMC, own pass, kernel-trace only).
# util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * SQ_BUSY_CU_CYCLES)   ; clock ~ GRBM_GUI_ACTIVE / 8 / kernel time
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_mfma${PMC_TAG:-}
mkdir -p $OUT
cd /tmp
timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o a -- python3 $R/bench.py --steps 2 --warmup 1 --legs encode > $OUT/a.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, glob, collections, json
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = None
        for k in ("k_gemm_bf16_nt<6>", "k_gemm_bf16_nt<3>", "k_gemm_bf16_nt<2>", "k_gemm_bf16_nt<1>", "k_attn_resident64", "k_attn_varlen_causal", "k_attn_stream"):
            if k in n: key = k
        if key: cnt[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$OUT/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for k in ("k_gemm_bf16_nt<6>", "k_gemm_bf16_nt<3>", "k_gemm_bf16_nt<2>", "k_gemm_bf16_nt<1>", "k_attn_resident64", "k_attn_varlen_causal", "k_attn_stream"):
            if k in n: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
out = {}
for k, c in cnt.items():
    a = {x: sum(v) / len(v) for x, v in c.items()}
    t = sum(dur[k]) / max(len(dur[k]), 1)
    e = {"avg_kernel_s": t, **{x: round(v) for x, v in a.items()}}
    if "GRBM_GUI_ACTIVE" in a and t > 0: e["effective_clock_GHz"] = round(a["GRBM_GUI_ACTIVE"] / 8 / t / 1e9, 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "SQ_BUSY_CU_CYCLES" in a and a["SQ_BUSY_CU_CYCLES"] > 0:
        e["mfma_busy_over_cu_busy_x4simd"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * a["SQ_BUSY_CU_CYCLES"]), 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "GRBM_GUI_ACTIVE" in a and a["GRBM_GUI_ACTIVE"] > 0:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
        e["mfma_util_vs_clock"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * a["GRBM_GUI_ACTIVE"] / 8), 4)
    out[k] = e
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/a
