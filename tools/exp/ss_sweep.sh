#!/bin/bash
# (the LRX_* switches below exist only in a -DLRX_DEV_KNOBS build of the library: tools/dev_lib.sh builds it and exports LRX_LIB_DEV_VARIANT)
. "$(dirname "$0")/../dev_lib.sh"
# sample stride (LRX_SS_FORCE sets it; "rule" = the library's own choice) against shard size, k and query count: where the emission of k*ss hits per query starts to cost more than a
# larger sample.  CFGS="rows,k,queries ..." SSS="2 4 8 ..."
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in ${CFGS:-100000,1000,256 100000,1000,100 1000000,1000,256 1000000,1000,100 125000,100,100 1000000,100,256}; do
  IFS=, read n k q <<< "$c"
  for ss in ${SSS:-2 4 8 12 20 32}; do
    echo -n "N=$n K=$k Q=$q ss=$ss  "
    if [ "$ss" = rule ]; then unset LRX_SS_FORCE; else export LRX_SS_FORCE=$ss; fi
    N=$n K=$k QS=$q D=${D:-2048} timeout -k 10 200 python3 $R/tools/bench_search.py 2>/dev/null | cut -c1-20
  done
done
