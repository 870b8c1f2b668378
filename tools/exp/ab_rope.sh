R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
for v in 1 0; do
  echo "== LRX_ROPE_FP32_TABLE=$v"
  LRX_ROPE_FP32_TABLE=$v python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-search --no-sparse 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'docs/s gemm_store', d['roofline']['per_class_ms_per_step']['gemm_store'])"
done; done
