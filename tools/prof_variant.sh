#!/bin/bash
# filter-kernel time of a dev variant library: tools/prof_variant.sh <variant .so>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_variant
rm -rf $OUT; mkdir -p $OUT
cd /tmp
export LRX_LIB_DEV_VARIANT=$1
QS=100 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 $R/tools/bench_search.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/s_kernel_stats.csv")):
    if 'scores_split<7, 1' in r['Name']: print("$1".split('/')[-1], "filter kernel avg %.1f us" % (float(r['AverageNs'])/1e3))
PY
