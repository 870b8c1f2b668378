#!/bin/bash
# gate-up GEMM: one workgroup per tile against the pipelined persistent variant (LRX_GEMM_PIPE=1: next tile's first K-tile requested before the epilogue)
R=${GRAFT_REPO_ROOT:-/root/repo}
for p in 0 1 0 1; do
  echo "== LRX_GEMM_PIPE=$p"
  LRX_GEMM_PIPE=$p python3 $R/tools/bench_gemm.py 2>&1 | grep "gate_up"
  LRX_GEMM_PIPE=$p H=4096 I=14336 QKV=6144 M=65536 python3 $R/tools/bench_gemm.py 2>&1 | grep "gate_up"
done
