#!/bin/bash
# idle cost of the gated six-product launch by grid size (CUs x LRX_GATED_GRID_X), and the cost of a search that DOES take the fallback
R=${GRAFT_REPO_ROOT:-/root/repo}
for x in 8 2 1 8 2 1; do
  echo "gated grid = CUs x $x: $(LRX_GATED_GRID_X=$x CFGS=125000,2048,100,100,0 bash $R/tools/exp/chain_timeline.sh 2>&1 | grep -E 'ms per search|k_flat_ip_scores_split' | tr '\n' ' ')"
done
