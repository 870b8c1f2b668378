#!/bin/bash
# K-loop ablation of k_gemm_bf16_nt (diagnostic variant builds; results are garbage, only the time matters):
#   tools/gemm_ablate.sh            -- builds liblrx_abl<mask>.so for each mask HERE (no GPU needed), or times them on the GPU box
# masks: 1 no LDS-DMA in the loop, 2 no fragment ds_reads, 4 no barriers, 8 no counted waits; 15 = MFMA only
R=${GRAFT_REPO_ROOT:-/root/repo}
MASKS="${MASKS:-0 1 2 4 3 12 13 15}"
if [ "$1" = "build" ]; then
  # the ablation / trace variants live in tools/exp/gemm_diagnostics.patch, not in the product kernel: build from a patched copy
  D=$R/lightretriever_amd/build/diag_csrc; rm -rf $D; mkdir -p $D; cp $R/lightretriever_amd/csrc/* $D/
  patch -s $D/lrx_gemm.hip < $R/tools/exp/gemm_diagnostics.patch || exit 1
  sed -i 's#"../../include/lrx.h"#"'$R'/include/lrx.h"#' $D/lrx_common.h
  export LRX_CSRC_DIR=$D
  for m in $MASKS; do python3 -m lightretriever_amd.build -DGEMM_ABL=$m --out=$R/lightretriever_amd/build/liblrx_abl$m.so > /dev/null || exit 1; done
  exit 0
fi
for m in $MASKS; do
  echo "== GEMM_ABL=$m"
  LRX_LIB_DEV_VARIANT=$R/lightretriever_amd/build/liblrx_abl$m.so timeout 120 python3 $R/tools/bench_gemm.py 2>&1 | grep -E "gate_up|down|qkv"
done
