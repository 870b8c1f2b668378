#!/bin/bash
# Same-box A/B of the wide query chunks of the bounded search (round 6): chunks of 256 (rounds 2-5), 512 and 1024 queries per pass over the
# shadow, on a -DLRX_DEV_KNOBS build (LRX_SEARCH_WIDE_MAX); shapes: the 1M x 2048 headline index at k = 100, the reference's evaluation point
# (100 k x 2048, k = 1000, eval/call_evaluate_mteb.sh:8-10), the 8B width.  Usage: tools/exp/wide_ab.sh <dev library> > profiles/r06_wide_ab.txt
set -e
. "$(dirname "$0")/../dev_lib.sh"
LIBV=${1:-$LRX_LIB_DEV_VARIANT}
for W in 256 512 1024; do
  echo "== LRX_SEARCH_WIDE_MAX=$W  1M x 2048, k = 100"
  LRX_LIB_DEV_VARIANT=$LIBV LRX_SEARCH_WIDE_MAX=$W QS=300,512,1000,2000 python tools/bench_search.py
  echo "== LRX_SEARCH_WIDE_MAX=$W  100k x 2048, k = 1000"
  LRX_LIB_DEV_VARIANT=$LIBV LRX_SEARCH_WIDE_MAX=$W N=100000 K=1000 QS=1000 python tools/bench_search.py
  echo "== LRX_SEARCH_WIDE_MAX=$W  1M x 4096, k = 100"
  LRX_LIB_DEV_VARIANT=$LIBV LRX_SEARCH_WIDE_MAX=$W D=4096 QS=1000 python tools/bench_search.py
done
