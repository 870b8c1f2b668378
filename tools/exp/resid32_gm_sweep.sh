#!/bin/bash
# (the LRX_* switches below exist only in a -DLRX_DEV_KNOBS build of the library: tools/dev_lib.sh builds it and exports LRX_LIB_DEV_VARIANT)
. "$(dirname "$0")/../dev_lib.sh"
# Round 5: tile-group size of the residual GEMMs on the fp32 stream (EPI_RESID32: 2.5 x the epilogue bytes of the bf16 one) -- the round-4 sweep
# chose 6 (K <= 4096) / 4 on the bf16 stream.  One process per value (LRX_GEMM_GM is read once).
for gm in 2 3 4 6 8 12 16; do
  echo "== LRX_GEMM_GM=$gm"
  LRX_GEMM_GM=$gm VARIANTS=1 timeout -k 10 200 python tools/bench_gemm.py 2>&1 | grep -E " o | down "
done
