#!/bin/bash
# Package power + shader clock (rocm-smi) sampled while a GEMM-only loop runs: tools/power_probe.sh [variant.so]
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -n "$1" ] && export LRX_LIB_DEV_VARIANT=$R/$1
LOOPS=400 timeout 120 python3 $R/tools/bench_gemm_loop.py > /tmp/pp.log 2>&1 &
PID=$!
for i in $(seq 1 40); do
  sleep 1
  kill -0 $PID 2>/dev/null || break
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Package Power|sclk" | sed 's/.*: //' | tr "\n" " "; echo
done
wait $PID
tail -2 /tmp/pp.log
