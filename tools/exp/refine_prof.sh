#!/bin/bash
# per-kernel times of the search chain per query count (one rocprofv3 run per Q)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/refine; mkdir -p $OUT
cd /tmp
for Q in ${QSS:-32 64 100 128}; do
  rm -rf $OUT/p
  QS=$Q timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o t -- python3 $R/tools/bench_search.py > $OUT/p.log 2>&1
  grep "Q=" $OUT/p.log
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/p/t_kernel_stats.csv")))
for r in rows:
    n=r["Name"]
    if n.startswith(("void k_filter_xreg","k_refine","k_sample","k_pack","void k_flat_ip","k_topk","k_rescore","k_split","k_round")) or "fill" in n.lower() or "memset" in n.lower():
        print("   Q=$Q", n[:56].ljust(56), r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done
