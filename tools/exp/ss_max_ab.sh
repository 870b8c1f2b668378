#!/bin/bash
# RETIRED (round 6): the switch this script drove is gone from the library -- its result is in profiles/r04_*.txt and is now a constant of plan_chunk / launch_scores.
# (the LRX_* switches below exist only in a -DLRX_DEV_KNOBS build of the library: tools/dev_lib.sh builds it and exports LRX_LIB_DEV_VARIANT)
. "$(dirname "$0")/../dev_lib.sh"
# larger sample strides now that the candidate lists hold 64 Ki entries (LRX_SS_MAX caps the rule; 32 is the shipped cap)
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "10000000 256" "1000000 4096" "1250000 4096" "1000000 2048"; do
  set -- $shape
  for m in 32 64 128 32 64; do
    echo "N=$1 D=$2 ss_max=$m: $(LRX_SS_MAX=$m N=$1 D=$2 QS=100 K=100 python3 $R/tools/bench_search.py | grep 'Q=')"
  done
done
