#!/bin/bash
# per-kernel times of the search chain (one rocprofv3 run per shape).  CFGS="rows,k,queries[,dim] ..."
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/bigk; mkdir -p $OUT
cd /tmp
for cfg in ${CFGS:-100000,1000,1000 1000000,1000,1000 1000000,100,1000}; do
  IFS=, read n k q d <<< "$cfg"
  rm -rf $OUT/p
  N=$n K=$k QS=$q D=${d:-2048} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o t -- python3 $R/tools/bench_search.py > $OUT/p.log 2>&1
  echo "== N=$n K=$k Q=$q D=${d:-2048}"; grep "Q=" $OUT/p.log
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/p/t_kernel_stats.csv")))
for r in rows:
    n=r["Name"]
    if "at::" in n or "elementwise" in n or "copyBuffer" in n or "k_shard_rows" in n: continue
    print("   ", n[:70].ljust(70), r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done
