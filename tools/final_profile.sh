#!/bin/bash
# Round-end evidence: the default bench line, then ONE rocprofv3 --kernel-trace --stats run PER BENCH LEG (program directly after `--`), so
# that every `frac` of the line can be recomputed from a kernel-stats file that holds that leg's launches only (one
# file for the whole bench mixed the 1B and 8B shapes of k_gemm_bf16_nt<2> under one average).  Output: gpurun_out/final/
#   bench.json                      the default `python bench.py` line
#   leg_<name>_kernel_stats.csv     per leg: encode, search, sparse, and each entry of `configs`
#   trace_clean.txt                 tools/check_trace_clean.py on a run of the two headline legs
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp
timeout 900 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
tail -c 1500 $OUT/bench.json
prof() {   # name, bench arguments ...
  local name=$1; shift
  rm -rf $OUT/prof_$name
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o t -- python3 $R/bench.py --steps 4 --warmup 2 "$@" > $OUT/prof_$name.log 2>&1 || echo "profile of $name failed"
  cp $(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1) $OUT/leg_${name}_kernel_stats.csv 2>/dev/null || true
  echo "leg $name: $(wc -l < $OUT/leg_${name}_kernel_stats.csv 2>/dev/null) kernels"
}
prof headline --legs encode,search
python3 $R/tools/check_trace_clean.py $(find $OUT/prof_headline -name "*kernel_trace.csv" | head -1) | tee $OUT/trace_clean.txt
prof encode --legs encode
prof search --legs search
prof sparse --legs sparse
for leg in ${LEGS:-top_k_1000 search_per_shard_8way search_clustered config2_search_1Mx4096 config4_search_10Mx256 config3_per_rank_shard_1250kx4096 config4_per_rank_shard_1250kx256 config2_encode_llama31_8b ragged_encode_llama32_1b n1_embedding_bag_build}; do
  prof cfg_$leg --legs configs --config-legs $leg
done
# the legs that only exist over a communicator (BASELINE configs[3] / configs[4] + the 8B encoder), as ONE rank of an 8-GPU run sees them: a forced
# one-rank RCCL group over 10M / 8 rows (the environment is exported BEFORE rocprofv3: the program itself follows `--`)
export LRX_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
prof sharded --legs sharded --sharded-rows 1250000
tail -c 4000 $OUT/prof_sharded.log | grep -o '{"partial_run.*' > $OUT/sharded_line.json || true
unset LRX_BENCH_FORCE_DIST MASTER_ADDR MASTER_PORT RANK WORLD_SIZE LOCAL_RANK
rm -rf $OUT/prof_*/   # (the traces are tens of MB; the stats files and logs stay)
ls $OUT
