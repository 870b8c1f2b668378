#!/usr/bin/env python3
"""Back-to-back gate-up-shaped GEMMs (M=131072, N=16384, K=2048) for LOOPS launches: a steady load for tools/power_probe.sh.
env: KIND = lrx (k_gemm_bf16_nt<EPI_SWIGLU>, default) | lrx_store (plain-store epilogue) | vendor (torch.matmul -> hipBLASLt);
     DATA = randn (default) | zeros | const (all operands 1.0: no bit toggles in the MFMA inputs);
     SHAPE = gate_up (default) | qkv | o | down;  LRX_GEMM_GM = m-tiles per group of the block -> tile map (library dev switch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightretriever_amd import ops

M = 131072
N, K = {"gate_up": (16384, 2048), "qkv": (3072, 2048), "o": (2048, 2048), "down": (2048, 8192)}[os.environ.get("SHAPE", "gate_up")]
kind, data = os.environ.get("KIND", "lrx"), os.environ.get("DATA", "randn")
g = torch.Generator(device="cuda").manual_seed(0)
if data == "randn":
    A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
else:
    v = 0.0 if data == "zeros" else 1.0
    A = torch.full((M, K), v, device="cuda", dtype=torch.bfloat16)
    B = torch.full((N, K), v, device="cuda", dtype=torch.bfloat16)
out = torch.empty(M, N // 2 if kind == "lrx" else N, dtype=torch.bfloat16, device="cuda")
if kind == "lrx":
    fn = lambda: ops.gemm_bf16_nt(A, B, epilogue=2, out=out)
elif kind == "lrx_store":
    fn = lambda: ops.gemm_bf16_nt(A, B, epilogue=0, out=out)
else:
    Bt = B.t()
    fn = lambda: torch.matmul(A, Bt, out=out)
for _ in range(3):
    fn()
torch.cuda.synchronize()
loops = int(os.environ.get("LOOPS", 400))
for rep in range(int(os.environ.get("REPS", 4))):
    t0 = time.perf_counter()
    for _ in range(loops):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {loops} launches in {dt:.2f} s = {2.0 * M * N * K * loops / dt / 1e12:.1f} TFLOP/s", flush=True)
