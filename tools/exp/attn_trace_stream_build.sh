#!/bin/bash
# Diagnostic build of liblrx with wall-clock stamps in k_attn_stream (run the probe with LRX_ATTN_STREAM=1) (workgroup 0, every wave): run HERE, then on the GPU box
#   LRX_LIB_DEV_VARIANT=$GRAFT_REPO_ROOT/lightretriever_amd/build/liblrx_atrace.so python3 tools/exp/attn_trace_tiled.py
# Tags: 1000+qt item start, 1100 next item known, 100 tile step start, 200 tile requested, 300 computed, 400 tile landed, 500 barrier passed,
# 1200 item's tiles done, 1300 output rows in the staging block, 1400 read back, 2000 item's output stores issued.
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
#define A_TRACE(tag)                                                                       \\
  do {                                                                                     \\
    if (blockIdx.x == 0 && lane == 0 && tr_n < 512) {                                      \\
      g_attn_trace[(wave * 512 + tr_n) * 2] = wall_clock64(); g_attn_trace[(wave * 512 + tr_n) * 2 + 1] = (tag); g_attn_trace_n[wave] = tr_n + 1; \\
    }                                                                                      \\
    ++tr_n;                                                                                \\
  } while (0)
''', 1)
i = s.index('k_attn_stream(const __bf16* __restrict__ qkv'); j = s.index('// 64 q rows per wave.')
k = s[i:j]
def rep(a, b):
    global k
    assert a in k, a
    k = k.replace(a, b, 1)
rep('  int ci = 0;\n', '  int tr_n = 0;\n  int ci = 0;\n')
rep('  while (ic[1] != 0) {\n  ++ci;\n  const i32x4 inext = lst[ci];\n', '  while (ic[1] != 0) {\n  A_TRACE(1000 + (ic[2] & 0xffff));\n  ++ci;\n  const i32x4 inext = lst[ci];\n  A_TRACE(1100);\n')
rep('    const bool more = stage_next();', '    A_TRACE(100);\n    const bool more = stage_next();')
rep('    if (active) {\n      if (first_half || !two) {', '    A_TRACE(200);\n    if (active) {\n      if (first_half || !two) {')
rep('    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");\n    else', '    A_TRACE(300);\n    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");\n    else')
rep('    __builtin_amdgcn_s_barrier();\n    cur = cur == NST - 1 ? 0 : cur + 1;\n', '    A_TRACE(400);\n    __builtin_amdgcn_s_barrier();\n    A_TRACE(500);\n    cur = cur == NST - 1 ? 0 : cur + 1;\n')
rep('  if (inext[1] != 0) read_q(qf);\n  if (active) {', '  A_TRACE(1200);\n  if (inext[1] != 0) read_q(qf);\n  if (active) {')
rep('    const __amdgpu_buffer_rsrc_t orsrc', '    A_TRACE(1400);\n    const __amdgpu_buffer_rsrc_t orsrc')
rep('  ic = inext;\n  }  // items', '  A_TRACE(2000);\n  ic = inext;\n  }  // items')
rep('        f32x16 s0_ = qk_product(sK, qf);\n        rescale(softmax(s0_, kt * 64 == q0, pf));\n        wait_v(vf);\n        pv(vf, pf);\n', '        f32x16 s0_ = qk_product(sK, qf);\n        A_TRACE(210);\n        rescale(softmax(s0_, kt * 64 == q0, pf));\n        A_TRACE(220);\n        wait_v(vf);\n        pv(vf, pf);\n        A_TRACE(230);\n')
rep('          f32x16 s1_ = qk_product(sK + 32 * G::ROW_BYTES, qf);\n          rescale(softmax(s1_, kt * 64 + 32 == q0, pf));\n          wait_v(vf);\n          pv(vf, pf);\n', '          f32x16 s1_ = qk_product(sK + 32 * G::ROW_BYTES, qf);\n          A_TRACE(240);\n          rescale(softmax(s1_, kt * 64 + 32 == q0, pf));\n          A_TRACE(250);\n          wait_v(vf);\n          pv(vf, pf);\n          A_TRACE(260);\n')
open(p, 'w').write(s[:i] + k + s[j:])
EOF
[ $? -eq 0 ] || exit 1
set -e
LRX_CSRC_DIR=$D python3 -m lightretriever_amd.build --out=$R/lightretriever_amd/build/liblrx_atrace.so > /dev/null || exit 1
echo built $R/lightretriever_amd/build/liblrx_atrace.so
