#!/bin/bash
# Round-end evidence: default bench line + rocprofv3 --kernel-trace --stats of the same command (program directly after `--`).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp
timeout 600 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
tail -c 3000 $OUT/bench.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/prof.log 2>&1 || echo "profile failed"
find $OUT/prof -name "*kernel_stats*.csv" | head -3
python3 $R/tools/check_trace_clean.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1)
