R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 700 python3 tools/parity_margin.py --presets llama32_1b,llama31_8b,qwen25_7b --docs 64 --out gpurun_out/r6_f16_parity.jsonl > gpurun_out/r6_f16_parity.log 2>&1 || { tail -30 gpurun_out/r6_f16_parity.log; exit 1; }
python3 - <<'PY'
import json
for l in open('gpurun_out/r6_f16_parity.jsonl'):
    d=json.loads(l); a=d['lrx_vs_fp32']; m=d['lrx_vs_fp32_mrl']
    print(d['preset'], d['stream'], 'sat', d['fp16_saturations'], 'lrx p50 %.2e max %.2e | mrl max %.2e | hf16 max %.2e'%(a['p50'],a['max'],m['max'],d['hfbf16_vs_fp32']['max']))
PY
