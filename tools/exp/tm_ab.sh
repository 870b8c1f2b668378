#!/bin/bash
# Round 5: selection + dynamically claimed main pass in one launch (k_filter_tm) against the three-launch chain (LRX_SEARCH_TM=0), same box
set -u
for shape in "125000 2048 100" "250000 2048 100" "62500 4096 100" "125000 1024 100" "125000 2048 10"; do
  set -- $shape
  for t in 1 0; do
    echo "== N=$1 D=$2 K=$3 LRX_SEARCH_TM=$t"
    N=$1 D=$2 K=$3 QS=${QS:-32,64,100,128} LRX_SEARCH_TM=$t timeout -k 10 120 python tools/bench_search.py 2>&1 | grep "Q="
  done
done
