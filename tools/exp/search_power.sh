#!/bin/bash
# Package power + shader clock (rocm-smi, one sample per second) while (a) the search's f16 GEMM passes (Q = 1000 over 1M x 2048) and (b) the
# encoder's bf16 gate-up GEMM loop back to back: does the f16 MFMA run at a lower clock under the 1400 W cap than the bf16 one?
R=${GRAFT_REPO_ROOT:-/root/repo}
probe() {
  "$@" > /tmp/pp.log 2>&1 &
  PID=$!
  sleep ${WARM:-9}
  for i in $(seq 1 6); do
    kill -0 $PID 2>/dev/null || break
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Package Power|sclk" | sed 's/.*: //' | tr "\n" " "; echo
    sleep 1
  done
  wait $PID
  tail -1 /tmp/pp.log
}
echo "== search, Q = 1000 over 1M x 2048 (k_gemm_bf16_nt<EPI_SAMPLE / EPI_EMIT>, v_mfma_f32_16x16x32_f16)"
LOOPS=4000 probe python3 $R/tools/exp/search_loop.py
echo "== encoder gate-up loop (k_gemm_bf16_nt<EPI_SWIGLU>, v_mfma_f32_16x16x32_bf16)"
KIND=lrx DATA=randn LOOPS=2500 REPS=1 WARM=5 probe python3 $R/tools/bench_gemm_loop.py
