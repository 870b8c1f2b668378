R=${GRAFT_REPO_ROOT:-/root/repo}
for v in 0 31 63 95 127 32 64; do
  echo "== ATTN_DIAG=$v (1 no QK mfma, 2 no softmax VALU, 4 no PV mfma, 8 no DMA, 16 no LDS fragment reads)"
  if [ $v = 0 ]; then SHAPES=32-8-128 python3 $R/tools/bench_attn.py 2>&1 | grep attn; else LRX_LIB_DEV_VARIANT=$R/lightretriever_amd/build/liblrx_ad$v.so SHAPES=32-8-128 python3 $R/tools/bench_attn.py 2>&1 | grep attn; fi
done
