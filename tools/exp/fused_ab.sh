#!/bin/bash
# (the LRX_* switches below exist only in a -DLRX_DEV_KNOBS build of the library: tools/dev_lib.sh builds it and exports LRX_LIB_DEV_VARIANT)
. "$(dirname "$0")/../dev_lib.sh"
# Round 5: the fused filter launch (k_filter_fused: sample + selection + main pass) against the three-launch chain, same box, one process per
# (shape, mode).  usage: tools/exp/fused_ab.sh > gpurun_out/r05_fused_ab.txt
set -u
for shape in "125000 2048 100" "250000 2048 100" "1000000 2048 100" "1250000 256 100" "1250000 4096 100" "1000000 4096 100" "100000 2048 1000" "1000000 2048 1000"; do
  set -- $shape
  for f in 1 0; do
    echo "== N=$1 D=$2 K=$3 LRX_SEARCH_FUSED=$f"
    N=$1 D=$2 K=$3 QS=${QS:-1,32,100,128} LRX_SEARCH_FUSED=$f timeout -k 10 300 python tools/bench_search.py 2>&1 | grep "Q="
  done
done
