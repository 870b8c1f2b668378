#!/bin/bash
# Per-tile time of the search's GEMM passes on shards small enough that each launch is ONE round of tiles (N = 32768: 64 + 64 tiles x 4
# query n-tiles = 256 workgroups per launch) and sits in the Infinity Cache after the first pass: separates the K loop's rate from HBM latency.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for N in ${NS:-32768 65536 131072 1000000}; do
  rm -rf /tmp/pp
  N=$N QS=1000 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python3 $R/tools/bench_search.py 2>&1 | grep "Q="
  python3 - <<PY
import csv
for r in csv.DictReader(open("/tmp/pp/t_kernel_stats.csv")):
    if "k_gemm_bf16_nt" in r["Name"]:
        print("   N=$N", r["Name"][:28], "calls", r["Calls"], "avg %.1f us  min %.1f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
