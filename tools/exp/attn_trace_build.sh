#!/bin/bash
# Diagnostic build of liblrx with wall-clock stamps in k_attn_resident64 (workgroup 0, every wave): run HERE (no GPU needed), then on the GPU box
#   LRX_LIB_DEV_VARIANT=$GRAFT_REPO_ROOT/lightretriever_amd/build/liblrx_atrace.so python3 tools/exp/attn_trace.py
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/lightretriever_amd/build/exp_csrc; rm -rf $D; mkdir -p $D; cp $R/lightretriever_amd/csrc/* $D/
sed -i 's#"../../include/lrx.h"#"'$R'/include/lrx.h"#' $D/lrx_common.h
python3 - "$D/lrx_attn.hip" <<'EOF'
import sys
p = sys.argv[1]
s = open(p).read()
s = s.replace('#include "lrx_common.h"', '''#include "lrx_common.h"
__device__ long long g_attn_trace[16 * 512 * 2];
__device__ int g_attn_trace_n[16];
extern "C" int lrx_debug_read_attn_trace(void* dst, size_t bytes, void* cnt) {
  if (hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_attn_trace_n), 64) != hipSuccess) return 1;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_trace), bytes) == hipSuccess ? 0 : 1;
}
// stamps of workgroup 0: the event count lives in a register (a stamp is two fire-and-forget stores), wall clock = s_memrealtime (100 MHz)
#define A_TRACE(tag)                                                                       \\
  do {                                                                                     \\
    if (blockIdx.x == 0 && lane == 0 && tr_n < 512) {                                      \\
      g_attn_trace[(wave * 512 + tr_n) * 2] = wall_clock64(); g_attn_trace[(wave * 512 + tr_n) * 2 + 1] = (tag); g_attn_trace_n[wave] = tr_n + 1; \\
    }                                                                                      \\
    ++tr_n;                                                                                \\
  } while (0)
''', 1)
i = s.index('k_attn_resident64(const __bf16* __restrict__ qkv'); j = s.index('// Suffix-over-shared-prefix attention')
k = s[i:j]
def before(anchor, text):
    global k
    assert anchor in k, anchor
    k = k.replace(anchor, text + anchor, 1)
k = k.replace('  for (int pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {', '  int tr_n = 0;\n  for (int pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {', 1)
k = k.replace('  if (len <= 0) continue;\n', '  if (len <= 0) continue;\n  A_TRACE(1000);\n', 1)
before('  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // K/V staged', '  A_TRACE(2000);\n')
k = k.replace('  while (t < n_tasks) {\n', '  A_TRACE(3000);\n  while (t < n_tasks) {\n    A_TRACE(4000 + (nsub - 1 - t / grp));\n', 1)
before("    // ---- normalise and store this task's rows", '    A_TRACE(6000);\n')
before('    t = tn;\n  }', '    A_TRACE(7000);\n')
before('  __syncthreads();     // every wave is done with this pair', '  A_TRACE(8000);\n')
open(p, 'w').write(s[:i] + k + s[j:])
EOF
LRX_CSRC_DIR=$D python3 -m lightretriever_amd.build --out=$R/lightretriever_amd/build/liblrx_atrace.so > /dev/null || exit 1
echo built $R/lightretriever_amd/build/liblrx_atrace.so
