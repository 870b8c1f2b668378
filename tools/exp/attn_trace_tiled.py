import ctypes as C, os, sys, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from lightretriever_amd import ops, _lib
lib = _lib.lib()
lib.lrx_debug_read_attn_trace.restype = C.c_int
lib.lrx_debug_read_attn_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
B, S, nq, nkv, d = 256, 512, 32, 8, 128
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * S, (nq + 2 * nkv) * d, generator=g, device="cuda").to(torch.float16)
cu = (torch.arange(B + 1, device="cuda") * S).to(torch.int32)
ops.attn_varlen_causal(qkv, cu, S, nq, nkv, d); torch.cuda.synchronize()
buf = np.zeros(16 * 512 * 2, np.int64); cnt = np.zeros(16, np.int32)
assert lib.lrx_debug_read_attn_trace(buf.ctypes.data, buf.nbytes, cnt.ctypes.data) == 0
t = buf.reshape(16, 512, 2)
for w in (0, 4):
    ev = t[w, :min(cnt[w], 512)]
    print("wave", w, "events", cnt[w], " ".join("%d@%.2f" % (tag, (c - ev[0][0]) / 100.0) for c, tag in ev[:70]))
tot = collections.defaultdict(list)
for w in range(1):
    ev = t[w, :min(cnt[w], 512)]
    for a, b in zip(ev[:-1], ev[1:]):
        ka = int(a[1]) if a[1] < 1000 or a[1] in (1100, 1200, 1300, 1400, 2000) else 1000
        kb = int(b[1]) if b[1] < 1000 or b[1] in (1100, 1200, 1300, 1400, 2000) else 1000
        tot[(ka, kb)].append((b[0] - a[0]) / 100.0)
for key in sorted(tot): print("phase %s -> %s: n=%d mean %.2f us" % (key[0], key[1], len(tot[key]), np.mean(tot[key])))
