#!/bin/bash
# search timings after a filter-kernel change: default mode, the BASELINE shapes
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
echo "== 1M x 2048 (default mode)"; QS=1,16,32,64,100,128,256,1000 timeout -k 10 300 python3 $R/tools/bench_search.py 2>&1 | grep "Q="
echo "== 1M x 2048 rows layout"; LAYOUT=rows QS=1,100 timeout -k 10 300 python3 $R/tools/bench_search.py 2>&1 | grep "Q="
echo "== 10M x 256"; N=10000000 D=256 QS=1,100,1000 timeout -k 10 300 python3 $R/tools/bench_search.py 2>&1 | grep "Q="
echo "== 1M x 4096"; D=4096 QS=1,100 timeout -k 10 300 python3 $R/tools/bench_search.py 2>&1 | grep "Q="
