#!/bin/bash
# One workgroup per tile against one persistent workgroup per CU (LRX_GEMM_PERSIST=1; =2: two per CU slot) on the encoder's GEMM shapes.
R=${GRAFT_REPO_ROOT:-/root/repo}
for p in 0 1 0 1; do
  echo "== LRX_GEMM_PERSIST=$p"
  LRX_GEMM_PERSIST=$p python3 $R/tools/bench_gemm.py 2>&1 | grep "TF/s"
  LRX_GEMM_PERSIST=$p VARIANTS=1 python3 $R/tools/bench_gemm.py 2>&1 | grep "TF/s"
done
