#!/bin/bash
# (the LRX_* switches below exist only in a -DLRX_DEV_KNOBS build of the library: tools/dev_lib.sh builds it and exports LRX_LIB_DEV_VARIANT)
. "$(dirname "$0")/../dev_lib.sh"
# block -> tile group size (LRX_GEMM_GM) on the fused QKV + RoPE projection and on the bf16 / fp32-stream residual GEMMs, 1B and 8B shapes
R=${GRAFT_REPO_ROOT:-/root/repo}
for gm in 0 2 4 6 8 12 16 0; do
  echo "== GM=$gm"; LRX_GEMM_GM=$gm VARIANTS=1 python3 $R/tools/bench_gemm.py 2>&1 | grep -E "^(1B|8B)" | sed 's/^/   /'
done
