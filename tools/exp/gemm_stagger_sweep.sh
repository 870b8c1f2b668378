#!/bin/bash
# Round 5: first-round start offsets of the residual GEMMs (four groups of CUs, step LRX_GEMM_STAGGER_US) so that the epilogues of a round do not
# all hit HBM in the same window.  One process per value.
for us in 0 5 10 15 20 30 45; do
  echo "== LRX_GEMM_STAGGER_US=$us"
  LRX_GEMM_STAGGER_US=$us VARIANTS=1 timeout -k 10 200 python tools/bench_gemm.py 2>&1 | grep -E " o | down "
done
