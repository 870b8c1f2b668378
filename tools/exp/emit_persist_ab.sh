#!/bin/bash
# RETIRED (round 6): the switch this script drove is gone from the library -- its result is in profiles/r04_*.txt and is now a constant of plan_chunk / launch_scores.
# (the LRX_* switches below exist only in a -DLRX_DEV_KNOBS build of the library: tools/dev_lib.sh builds it and exports LRX_LIB_DEV_VARIANT)
. "$(dirname "$0")/../dev_lib.sh"
# A/B: persistent main pass (one workgroup per CU walking its blocks) vs one workgroup per 128-row block, on the per-rank shard sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "125000 2048" "250000 2048" "500000 2048" "1250000 256"; do
  set -- $shape
  for bpc in 0 1000 0 1000; do
    echo "N=$1 D=$2 persist_min_bpc=$bpc: $(LRX_EMIT_PERSIST_MIN_BPC=$bpc N=$1 D=$2 QS=100 K=100 python3 $R/tools/bench_search.py | grep 'Q=')"
  done
done
