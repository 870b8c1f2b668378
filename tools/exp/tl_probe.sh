# round-4 probe: margins of the trained-like and the Gaussian profile, 64 documents x 3 weight seeds (tools/parity_margin.py)
set -e
P="python tools/parity_margin.py --docs 64 --out gpurun_out/r04_tl_probe.jsonl"
$P --presets llama32_1b,qwen25_1_5b,llama31_8b,qwen25_7b --seeds 0,1,2 > gpurun_out/r04_tl_probe.log 2>&1
$P --presets llama31_8b,qwen25_7b --seeds 0,1,2 --profile gaussian >> gpurun_out/r04_tl_probe.log 2>&1
$P --presets llama32_1b,qwen25_1_5b,llama32_3b,qwen25_3b --seeds 0 --profile gaussian >> gpurun_out/r04_tl_probe.log 2>&1
