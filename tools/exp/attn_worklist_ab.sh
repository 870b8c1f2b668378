#!/bin/bash
# Work-list launch against the walker launch (tools/bench_attn.py) over head layouts and sequence lengths; on the GPU box:
#   bash tools/exp/attn_worklist_ab.sh > gpurun_out/attn_worklist_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "256 512 32-8-128,24-8-128,28-4-128,12-2-128,64-8-128" "32 2048 32-8-128,32-8-64,14-2-64,12-2-64" "8 8192 32-8-128,32-8-64"; do
  set -- $cfg
  echo "== B=$1 S=$2"
  B=$1 S=$2 SHAPES=$3 REPS=8 python3 $R/tools/bench_attn.py 2>&1 | grep "^attn"
  B=$1 S=$2 SHAPES=$3 REPS=8 WALKER=1 python3 $R/tools/bench_attn.py 2>&1 | grep "^attn"
done
