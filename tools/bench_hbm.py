"""Practical HBM read ceiling on this box: time simple streaming reads of an 8 GB fp32 buffer with library kernels."""
import torch, time
x = torch.randn(1_000_000, 2048, device="cuda")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = x.numel() * 4
for name, fn in [("sum", lambda: x.sum()), ("abs().max", lambda: x.abs().max()), ("sum(dim=1)", lambda: x.sum(dim=1)), ("sum(dim=0)", lambda: x.sum(dim=0)),
                 ("matvec fp32", lambda: x @ x[0]), ("copy (r+w)", lambda: x.clone())]:
    ms = t(fn)
    print(f"{name:14s} {ms:.3f} ms  {B/ms/1e6*(2 if 'copy' in name else 1):.0f} GB/s")
