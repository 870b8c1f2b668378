#!/bin/bash
# HBM traffic of the dominant kernels via rocprofv3 PMC counters, collected exactly as MI355X_MICROARCH.md prescribes:
# separate passes for FETCH_SIZE and WRITE_SIZE, --kernel-trace only, program directly after `--`.
# FETCH_SIZE on gfx950 reports 1/2 of the bytes of wide coalesced streaming reads -> doubled in the summary script.
# Two workloads: the encode leg of bench.py (GEMM / attention kernels) and ONE search shape (Q=100 over 1M x 2048, tools/bench_search.py)
# so that the per-launch averages of the search kernels are not a mix of query counts.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_r06
mkdir -p $OUT/enc $OUT/srch
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  sub=$( [ $C = FETCH_SIZE ] && echo fetch || echo write )
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/enc/$sub -o $sub -- python3 $R/bench.py --steps 2 --warmup 1 --legs encode,sparse > $OUT/enc_$sub.log 2>&1 || echo "encode $sub pass failed"
  QS=100 timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/srch/$sub -o $sub -- python3 $R/tools/bench_search.py > $OUT/srch_$sub.log 2>&1 || echo "search $sub pass failed"
done
python3 $R/tools/pmc_summary.py $OUT/enc > $OUT/enc_summary.json
python3 $R/tools/pmc_summary.py $OUT/srch > $OUT/srch_summary.json
python3 - <<PY
import json
a = json.load(open("$OUT/enc_summary.json")); b = json.load(open("$OUT/srch_summary.json"))
keep = ("k_flat_ip", "k_filter_xreg", "k_sample_threshold", "k_refine", "k_topk_select", "k_rescore", "k_shard_rows")
out = {k: v for k, v in a.items() if not k.startswith(keep)}
out.update({k: v for k, v in b.items() if k.startswith(keep)})
out["_workloads"] = {"encode": "bench.py --steps 2 --warmup 1 --legs encode,sparse (llama3.2-1b dims, 256 x 512 tokens, fp32 residual stream: one shape per kernel name)", "search": "tools/bench_search.py QS=100, 1M x 2048, k=100"}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k in ("k_gemm_bf16_nt<2>", "k_filter_xreg<emit>", "k_filter_xreg<scores>", "k_sample_threshold", "k_refine_band", "k_refine_merge"):
    if k in out: print(k, round(out[k]["hbm_bytes_per_launch"] / 1e6, 1), "MB per launch")
PY
rm -rf $OUT/enc $OUT/srch   # (raw per-dispatch counter files: tens of MB; gpurun merges back at most 64 MiB)
