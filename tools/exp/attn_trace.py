#!/usr/bin/env python3
"""Per-wave timeline of workgroup 0 of k_attn_resident64 (diagnostic build with A_TRACE stamps; LRX_LIB_DEV_VARIANT=<that .so>).
Tags: 1000 pair start, 2000 staging issued, 3000 staged + barrier, 4000+i task start (q block i), 6000 last softmax
done, 7000 stores issued, 8000 wave done with the pair."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lightretriever_amd import ops, _lib

lib = _lib.lib()
lib.lrx_debug_read_attn_trace.restype = C.c_int
lib.lrx_debug_read_attn_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
B, S, nq, nkv, d = 256, 512, 32, 8, 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * S, (nq + 2 * nkv) * d, generator=g, device="cuda").to(torch.float16)
cu = (torch.arange(B + 1, device="cuda") * S).to(torch.int32)
ops.attn_varlen_causal(qkv, cu, S, nq, nkv, d)
torch.cuda.synchronize()
buf = np.zeros(16 * 512 * 2, np.int64); cnt = np.zeros(16, np.int32)
assert lib.lrx_debug_read_attn_trace(buf.ctypes.data, buf.nbytes, cnt.ctypes.data) == 0
t = buf.reshape(16, 512, 2)
base = min(t[w, 0, 0] for w in range(16))
MHZ = float(os.environ.get("CLK_MHZ", 100.0))      # clock64 = s_memtime: constant 100 MHz on gfx9
for w in (0, 1, 5, 15):
    ev = t[w, :cnt[w]]
    print("wave", w, "events", cnt[w])
    line = []
    for (c, tag) in ev[:60]:
        line.append("%d@%.2f" % (tag, (c - base) / MHZ))
    print("   ", " ".join(line))
# aggregate over all waves: time per phase
tot = {}
for w in range(16):
    ev = t[w, :cnt[w]]
    for a, b in zip(ev[:-1], ev[1:]):
        key = (int(a[1]) // 1000, int(b[1]) // 1000)
        tot.setdefault(key, []).append((b[0] - a[0]) / MHZ)
for key in sorted(tot):
    v = tot[key]
    print("phase %s -> %s: n=%d mean %.2f us  sum/wave %.1f us" % (key[0], key[1], len(v), sum(v) / len(v), sum(v) / 16))
# task time against its sub-tile count
import collections
per = collections.defaultdict(list)
for w in range(16):
    ev = t[w, :cnt[w]]
    for a, b, c_ in zip(ev[:-2], ev[1:-1], ev[2:]):
        if 4000 <= a[1] < 5000 and b[1] == 6000:
            per[int(a[1]) - 4000].append((b[0] - a[0]) / MHZ)
for i in sorted(per):
    print("q block %2d (%2d sub-tiles): compute %.2f us -> %.3f us per sub-tile" % (i, i + 1, np.mean(per[i]), np.mean(per[i]) / (i + 1)))
