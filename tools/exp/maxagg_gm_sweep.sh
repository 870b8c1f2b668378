#!/bin/bash
# (the LRX_* switches below exist only in a -DLRX_DEV_KNOBS build of the library: tools/dev_lib.sh builds it and exports LRX_LIB_DEV_VARIANT)
. "$(dirname "$0")/../dev_lib.sh"
# tile-group size of the LM-head max-aggregation GEMM (EPI_MAXAGG, M = 131 072, N = 128 256, K = 2048) -- m-tiles per group
# of the block -> tile map; one process per value (the value is read once).  usage: tools/exp/maxagg_gm_sweep.sh > gpurun_out/maxagg_gm.txt
for gm in 2 4 6 8 12 16 32 64; do
  echo "== LRX_MAXAGG_GM=$gm"
  LRX_MAXAGG_GM=$gm python tools/bench_sparse.py --iters 4 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('max_aggregate_ms','max_aggregate_tflops','docs_per_s_dense_plus_sparse')})"
done
