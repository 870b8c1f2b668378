"""Is the search GEMM's rate (1.18-1.35 PFLOP/s at M = 1M rows, N = 1000 queries, K = 2048) a property of its instantiation or of the SHAPE?
The encoder's bf16 kernel (plain-store epilogue) on M x N x K for a few N at K = 2048: same tile, same K loop, operands from HBM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lightretriever_amd import _lib
lib = _lib.lib()
M, K = int(os.environ.get("M", 1_000_000)), int(os.environ.get("K", 2048))
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.randn(M, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
for N in [int(x) for x in os.environ.get("NS", "256,1024,4096,16384").split(",")]:
    B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    fn = lambda: _lib.check(lib.lrx_gemm_bf16_nt(_lib.ptr(A), _lib.ptr(B), _lib.ptr(C), None, None, M, N, K, 0, _lib.current_stream()))
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tiles = -(-M // 256) * -(-N // 256)
    print("M=%d N=%5d K=%d: %.3f ms = %.1f TFLOP/s; %.1f us per tile round (%d tiles)" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9, ms * 1e3 / (tiles / 256), tiles), flush=True)
    del B, C
