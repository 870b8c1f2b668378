#!/bin/bash
# (the LRX_* switches below exist only in a -DLRX_DEV_KNOBS build of the library: tools/dev_lib.sh builds it and exports LRX_LIB_DEV_VARIANT)
. "$(dirname "$0")/dev_lib.sh"
# Package power + shader clock (rocm-smi, one sample per second) while a GEMM-only loop runs, one block per variant:
#   tools/power_probe.sh            -> the round-4 table: gate-up kernel (random / constant / zero operands), block->tile group sizes with
#                                      more and less fabric traffic, the plain-store epilogue, the vendor GEMM on the same shape
#   VARIANTS="KIND=lrx;DATA=randn ..." tools/power_probe.sh   -> custom list (semicolon-separated env assignments per variant)
R=${GRAFT_REPO_ROOT:-/root/repo}
VARIANTS=${VARIANTS:-"KIND=lrx;DATA=randn KIND=lrx;DATA=randn;LRX_GEMM_GM=1 KIND=lrx;DATA=randn;LRX_GEMM_GM=16 KIND=lrx;DATA=randn;LRX_GEMM_GM=64 KIND=lrx_store;DATA=randn KIND=vendor;DATA=randn KIND=lrx;DATA=const KIND=lrx;DATA=zeros KIND=vendor;DATA=zeros"}
echo "# rocm-smi --showmaxpower: $(rocm-smi --showmaxpower 2>&1 | grep -E 'Max' | sed 's/.*: //' | tr '\n' ' ')"
for v in $VARIANTS; do
  echo "== $v"
  ( export $(echo $v | tr ';' ' '); LOOPS=${LOOPS:-300} REPS=3 timeout 120 python3 $R/tools/bench_gemm_loop.py > /tmp/pp.log 2>&1 ) &
  PID=$!
  sleep 4
  for i in $(seq 1 6); do
    kill -0 $PID 2>/dev/null || break
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Package Power|sclk" | sed 's/.*: //' | tr "\n" " "; echo
    sleep 1
  done
  wait $PID
  tail -2 /tmp/pp.log
done
