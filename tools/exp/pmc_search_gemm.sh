#!/bin/bash
# Why does the search's GEMM K loop take 51 us per 256 x 256 x 2048 tile where the encoder's takes 37?  rocprofv3 PMC passes (each its own run,
# kernel-trace only) over tools/bench_search.py QS=1000 (k_gemm_bf16_nt<5> = EPI_EMIT, <7> = EPI_SAMPLE) and over the encode leg of bench.py
# (k_gemm_bf16_nt<2> = gate-up, <3> = QKV + RoPE): MFMA busy share + effective clock, LDS bank conflicts / LDS wait, vector-memory waits.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_sgemm
rm -rf $OUT; mkdir -p $OUT
cd /tmp
SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"
      "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES")
i=0
for S in "${SETS[@]}"; do
  QS=1000 timeout 300 rocprofv3 --kernel-trace --pmc $S --output-format csv -d $OUT/s$i -o a -- python3 $R/tools/bench_search.py > $OUT/s$i.log 2>&1 || echo "search pass $i failed"
  timeout 300 rocprofv3 --kernel-trace --pmc $S --output-format csv -d $OUT/e$i -o a -- python3 $R/bench.py --steps 2 --warmup 1 --legs encode > $OUT/e$i.log 2>&1 || echo "encode pass $i failed"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections, json
keys = ("k_gemm_bf16_nt<5>", "k_gemm_bf16_nt<7>", "k_gemm_bf16_nt<2>", "k_gemm_bf16_nt<3>", "k_gemm_bf16_nt<6>")
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in keys:
            if k in r["Kernel_Name"]: cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in keys:
            if k in r["Kernel_Name"]: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
out = {}
for k in keys:
    if k not in cnt: continue
    a = {x: sum(v) / len(v) for x, v in cnt[k].items()}
    t = sum(dur[k]) / max(len(dur[k]), 1)
    e = {"avg_kernel_ms": round(t * 1e3, 4)}
    cu = a.get("SQ_BUSY_CU_CYCLES", 0)
    if a.get("GRBM_GUI_ACTIVE") and t > 0: e["effective_clock_GHz"] = round(a["GRBM_GUI_ACTIVE"] / 8 / t / 1e9, 3)
    if cu:
        for name, c, div in (("mfma_busy_share", "SQ_VALU_MFMA_BUSY_CYCLES", 4), ("lds_idx_active_share", "SQ_LDS_IDX_ACTIVE", 1), ("lds_bank_conflict_share", "SQ_LDS_BANK_CONFLICT", 1)):
            if c in a: e[name] = round(a[c] / (div * cu), 4)
    for c in ("SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM"):
        if c in a: e[c] = round(a[c])
    if "SQ_WAVE_CYCLES" in a:
        for c in ("SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"):
            if c in a and a["SQ_WAVE_CYCLES"]: e[c + "_per_wave_cycle"] = round(a[c] / a["SQ_WAVE_CYCLES"], 4)
    out[k] = e
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/s? $OUT/e?
