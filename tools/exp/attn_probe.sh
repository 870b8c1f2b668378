#!/bin/bash
# Phase ablation of k_attn_varlen_causal (profiles/r02_attn_ablation.txt): diagnostic variants live in tools/exp/attn_diagnostics.patch,
# not in the product kernel.   tools/exp/attn_probe.sh build   -- here (no GPU needed);   tools/exp/attn_probe.sh   -- on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}
MASKS="${MASKS:-1 2 4 7 15 32 47}"
if [ "$1" = "build" ]; then
  D=$R/lightretriever_amd/build/diag_csrc; rm -rf $D; mkdir -p $D; cp $R/lightretriever_amd/csrc/* $D/
  patch -s $D/lrx_attn.hip < $R/tools/exp/attn_diagnostics.patch || exit 1
  sed -i 's#"../../include/lrx.h"#"'$R'/include/lrx.h"#' $D/lrx_common.h
  for m in $MASKS; do LRX_CSRC_DIR=$D python3 -m lightretriever_amd.build -DATTN_DIAG=$m --out=$R/lightretriever_amd/build/liblrx_ad$m.so > /dev/null || exit 1; done
  exit 0
fi
echo "== product"; SHAPES=${SHAPES:-32-8-128} python3 $R/tools/bench_attn.py 2>&1 | grep "^attn"
for m in $MASKS; do
  echo "== ATTN_DIAG=$m (1 no QK mfma, 2 no softmax VALU, 4 no PV mfma, 8 no K/V DMA, 32 no O stores)"
  LRX_LIB_DEV_VARIANT=$R/lightretriever_amd/build/liblrx_ad$m.so SHAPES=${SHAPES:-32-8-128} python3 $R/tools/bench_attn.py 2>&1 | grep "^attn"
done
