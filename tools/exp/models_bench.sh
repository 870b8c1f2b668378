R=${GRAFT_REPO_ROOT:-/root/repo}
for m in llama3.2-1b qwen2.5-1.5b llama3.2-3b; do
  python3 $R/bench.py --model $m --steps 3 --warmup 1 --no-cpu-baseline --no-search --no-sparse --no-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], 'docs/s', d['roofline']['end_to_end_tflops'], 'TF/s e2e', d['roofline']['per_class_ms_per_step'])"
done
for m in qwen2.5-7b llama3.1-8b; do
  python3 $R/bench.py --model $m --batch-docs 128 --steps 3 --warmup 1 --no-cpu-baseline --no-search --no-sparse --no-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], 'docs/s', d['roofline']['end_to_end_tflops'], 'TF/s e2e', d['roofline']['per_class_ms_per_step'])"
done
# the two small deep backbones of the default list need their own line when LRX_ALL=1 asks for all six
if [ -n "$LRX_ALL" ]; then
  python3 $R/bench.py --model qwen2.5-3b --steps 3 --warmup 1 --no-cpu-baseline --no-search --no-sparse --no-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qwen2.5-3b', d['value'], 'docs/s', d['roofline']['end_to_end_tflops'], 'TF/s e2e', d['roofline']['per_class_ms_per_step'])"
fi
