#!/bin/bash
# (the LRX_* switches below exist only in a -DLRX_DEV_KNOBS build of the library: tools/dev_lib.sh builds it and exports LRX_LIB_DEV_VARIANT)
. "$(dirname "$0")/../dev_lib.sh"
# m-tiles per group of the GEMM's block -> tile map (LRX_GEMM_GM) on the four projection shapes of the headline model
R=${GRAFT_REPO_ROOT:-/root/repo}
for gm in 0 1 2 3 4 6 8 12 16 24 32 0; do
  echo "GM=$gm: $(LRX_GEMM_GM=$gm python3 $R/tools/bench_gemm.py 2>&1 | grep -E 'qkv|^o |down' | awk '{print $1, $7, $8, $9}' | tr '\n' ' ')"
done
