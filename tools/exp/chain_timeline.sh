#!/bin/bash
# Timeline of one search chain per shape (rocprofv3 --kernel-trace of tools/search_chain_timeline.py).  CFGS="N,D,Q,K,EXCHANGE ..."
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/chain; mkdir -p $OUT
cd /tmp
for cfg in ${CFGS:-125000,2048,100,100,0 125000,2048,100,100,1 1250000,256,100,100,0 1000000,2048,100,100,0 1000000,2048,100,1000,0 100000,2048,100,1000,0}; do
  IFS=, read n d q k x <<< "$cfg"
  rm -rf $OUT/p
  N=$n D=$d Q=$q K=$k EXCHANGE=$x timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p -o t -- python3 $R/tools/search_chain_timeline.py run > $OUT/p.log 2>&1
  echo "== $(grep 'ms per search' $OUT/p.log)"
  python3 $R/tools/search_chain_timeline.py show $(find $OUT/p -name "*kernel_trace.csv" | head -1)
done
