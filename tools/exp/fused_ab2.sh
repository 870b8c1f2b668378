#!/bin/bash
# (the LRX_* switches below exist only in a -DLRX_DEV_KNOBS build of the library: tools/dev_lib.sh builds it and exports LRX_LIB_DEV_VARIANT)
. "$(dirname "$0")/../dev_lib.sh"
# dev: the fused launch against the three-launch chain on the per-rank shard, with the phase timeline of the fused kernel
for v in "1" "0"; do
  echo "== 125000x2048 fused=$v"
  N=125000 D=2048 K=100 QS=1,100 LRX_SEARCH_FUSED=$v timeout -k 10 120 python tools/bench_search.py 2>&1 | grep "Q="
  N=125000 D=2048 LRX_SEARCH_FUSED=$v timeout -k 10 120 python tools/exp/fused_debug.py 2>&1 | grep "ids equal"
done
for q in 100 1; do echo "== timeline Q=$q"; Q=$q LRX_FUSED_PHASES=135 timeout -k 10 120 python tools/exp/fused_timeline.py 2>&1 | grep -v amdgpu; done
