# same-box A/B of the fp32 stream's GEMM operands (bf16 vs fp16): the 1B headline step and the 8B step
R=${GRAFT_REPO_ROOT:-/root/repo}
for ops in bf16 fp16_qkv fp16 bf16 fp16_qkv fp16; do
  LRX_BENCH_OPERANDS=$ops python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-search --no-sparse --no-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('llama3.2-1b', '$ops', d['value'], 'docs/s', d['config']['stream_mode'], d['dtype'], d['roofline']['per_class_ms_per_step'])"
done
for ops in bf16 fp16_qkv fp16; do
  LRX_BENCH_OPERANDS=$ops python3 $R/bench.py --legs configs --config-legs config2_encode_llama31_8b --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['configs']['config2_encode_llama31_8b']; print('llama3.1-8b', c['operands'], c['docs_per_s'], 'docs/s, gate-up', c['roofline']['achieved'], 'TFLOP/s')"
done
