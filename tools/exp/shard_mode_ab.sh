#!/bin/bash
# A/B of the two filters of the bounded search on the per-rank shards of the 8-GPU configurations: MODE=1 score-matrix
# filter (one full pass that stores every score + exact select), MODE=2 score-free filter (sample -> threshold -> emitting main pass).
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "125000 2048" "250000 2048" "1250000 256" "1250000 4096"; do
  set -- $shape
  for mode in 1 2 1 2; do
    echo "N=$1 D=$2 MODE=$mode: $(N=$1 D=$2 MODE=$mode QS=100 K=100 python3 $R/tools/bench_search.py | grep 'Q=')"
  done
done
