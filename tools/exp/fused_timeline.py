"""Dev: per-workgroup phase timeline of one fused filter launch (LRX_FUSED_PHASES must include bit 7 = 128).
Needs a -DLRX_DEV_KNOBS build of the library (`. tools/dev_lib.sh` builds it and exports LRX_LIB_DEV_VARIANT): the shipping liblrx.so reads no
environment variable."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lightretriever_amd import FlatIPIndex, _lib
N, D, Q, K = int(os.environ.get("N", 125000)), int(os.environ.get("D", 2048)), int(os.environ.get("Q", 100)), int(os.environ.get("K", 100))
g = torch.Generator(device="cuda").manual_seed(3)
idx = FlatIPIndex(D, capacity=N)
slot = idx.append_slot(N)
for s in range(0, N, 65536):
    e = min(s + 65536, N)
    slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
idx.commit(N)
q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
for _ in range(5):
    idx.search(q, K)
torch.cuda.synchronize()
ts = np.zeros(256 * 8, dtype=np.uint64)
_lib.check(_lib.lib().lrx_probe_fused_timestamps(ts.ctypes.data_as(C.c_void_p), ts.size))
t = ts.reshape(256, 8).astype(np.float64) / 100.0      # us
t0 = t[:, 0].min()
def col(i):
    c = t[:, i].copy(); c[c == 0] = np.nan; return c - t0
names = ["start", "S done", "sel start (after wait for samples)", "T done", "first main K loop done", "thresholds seen", "end"]
for i, n in enumerate(names):
    c = col(i)
    print("%-40s n=%3d  min %7.1f  median %7.1f  max %7.1f us" % (n, int(np.sum(~np.isnan(c))), np.nanmin(c), np.nanmedian(c), np.nanmax(c)))
sel = ~np.isnan(col(2))
print("selection workgroups: %d; their T phase (sel start -> T done): median %.1f max %.1f us; wait for samples median %.1f us" % (
    sel.sum(), np.nanmedian((col(3) - col(2))[sel]), np.nanmax((col(3) - col(2))[sel]), np.nanmedian((col(2) - col(1))[sel])))
print("threshold wait (K loop done -> seen): median %.1f max %.1f us" % (np.nanmedian(col(5) - col(4)), np.nanmax(col(5) - col(4))))
print("main phase after thresholds (seen -> end): median %.1f max %.1f" % (np.nanmedian(col(6) - col(5)), np.nanmax(col(6) - col(5))))
