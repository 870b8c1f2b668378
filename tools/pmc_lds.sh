#!/bin/bash
# LDS behaviour of the encoder kernels (rocprofv3 PMC, own pass): bank-conflict cycles, LDS-array active cycles, LDS instructions.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_lds
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 500 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/a -o a -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sparse > $OUT/a.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, glob, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = None
        for k in ("k_gemm_bf16_nt<3", "k_gemm_bf16_nt<2", "k_gemm_bf16_nt<1", "k_attn_resident64", "k_flat_ip_scores_split", "k_refine_topk"):
            if k in n: key = k
        if key: cnt[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in cnt.items():
    a = {x: sum(v) / len(v) for x, v in c.items()}
    gui = a.get("GRBM_GUI_ACTIVE", 0)
    print(k, {x: int(v) for x, v in a.items()})
    if a.get("SQ_LDS_IDX_ACTIVE"):
        print("   bank-conflict share of LDS-array cycles: %.3f ; LDS-array busy per CU vs kernel cycles: %.3f" % (
            a.get("SQ_LDS_BANK_CONFLICT", 0) / a["SQ_LDS_IDX_ACTIVE"], a["SQ_LDS_IDX_ACTIVE"] / 256 / (gui / 8) if gui else -1))
PY
