#!/bin/bash
# same-box A/B of GEMM variant libraries: tools/gemm_ab.sh <variant.so> [<variant2.so> ...]   (product library first, ABAB order)
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for lib in product "$@"; do
    echo "== $lib (rep $rep)"
    if [ "$lib" = product ]; then unset LRX_LIB_DEV_VARIANT; else export LRX_LIB_DEV_VARIANT=$R/$lib; fi
    timeout 120 python3 $R/tools/bench_gemm.py 2>&1 | grep -E "gate_up|down|qkv|^o "
  done
done
