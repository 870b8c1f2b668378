#!/usr/bin/env python3
"""Per-CU timeline of one GEMM launch (diagnostic build from a csrc copy patched with tools/exp/gemm_diagnostics.patch -- see
tools/gemm_ablate.sh for the recipe --: LRX_CSRC_DIR=<copy> python lightretriever_amd/build.py -DGEMM_TRACE --out=.../liblrx_trace.so,
then LRX_LIB_DEV_VARIANT=.../liblrx_trace.so python tools/gemm_timeline.py).  Each workgroup stamps start / main loop done / end and
its hardware id; this script reports, per epilogue class, tile time, epilogue time, the gap between consecutive workgroups on the
same CU, and how many CUs are inside their epilogue at the same moment."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lightretriever_amd import ops, _lib

def main():
    lib = _lib.lib()
    lib.lrx_debug_read_gemm_trace.restype = C.c_int
    lib.lrx_debug_read_gemm_trace.argtypes = [C.c_void_p, C.c_size_t]
    M = 131072
    g = torch.Generator(device="cuda").manual_seed(0)
    from lightretriever_amd.encoder import EncoderConfig, rope_tables
    cos, sin = rope_tables(EncoderConfig.llama32_1b(512))
    cos, sin = cos.cuda(), sin.cuda()
    pos = (torch.arange(M, device="cuda") % 512).to(torch.int32)
    for name, N, K, epi in [("gate_up", 16384, 2048, 2), ("o", 2048, 2048, 1), ("down", 2048, 8192, 1), ("qkv", 3072, 2048, 0), ("qkv_rope", 3072, 2048, 3)]:
        A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
        B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
        out = torch.empty(M, N // 2 if epi == 2 else N, dtype=torch.bfloat16, device="cuda")
        for _ in range(2):
            if epi == 3:
                ops.gemm_qkv_rope(A, B, pos, cos, sin, 32, 8, 64)
            else:
                ops.gemm_bf16_nt(A, B, resid=out if epi == 1 else None, epilogue=epi, out=out)
        torch.cuda.synchronize()
        ntiles = (M // 256) * ((N + 255) // 256)
        n = min(ntiles, 65536)
        buf = np.zeros(8 * 65536, np.int64)
        assert lib.lrx_debug_read_gemm_trace(buf.ctypes.data, buf.nbytes) == 0
        t = buf.reshape(-1, 8)[:n]
        t0, t1, t2, hw = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
        base = t0.min()
        us = lambda x: x / 100.0                                  # wall_clock64: 100 MHz
        cu = ((hw >> 32) << 16) | ((hw & 0xffffffff) >> 8 & 0xf) | (((hw & 0xffffffff) >> 13 & 0x7) << 4) | (((hw & 0xffffffff) >> 16 & 0xf) << 8)
        tile, epi_t = us(t2 - t0), us(t2 - t1)
        gaps = []
        for c in np.unique(cu):
            idx = np.flatnonzero(cu == c)
            o = idx[np.argsort(t0[idx])]
            gaps.append(us(t0[o][1:] - t2[o][:-1]))
        gaps = np.concatenate(gaps) if gaps else np.zeros(1)
        # concurrency of epilogues: for each workgroup, how many others are in their epilogue at its epilogue midpoint
        mid = (t1 + t2) // 2
        order = np.argsort(mid)
        sample = order[:: max(1, n // 2000)]
        conc = [(int(((t1 <= m) & (t2 >= m)).sum())) for m in mid[sample]]
        span = us(t2.max() - base)
        t4, t5, t6 = t[:, 4], t[:, 5], t[:, 6]
        print(f"{name:8s} phases (median us): prologue (first K-tile landed) {np.median(us(t4-t0)):.2f} | K loop {np.median(us(t1-t4)):.2f} | acc->LDS staging {np.median(us(t5-t1)):.2f} "
              f"| LDS->global issue {np.median(us(t6-t5)):.2f} | store drain {np.median(us(t2-t6)):.2f}")
        print(f"{name:8s} tiles={ntiles} (traced {n}) distinct CUs={len(np.unique(cu))} launch span {span:.0f} us | tile {np.median(tile):.1f} us (p10 {np.percentile(tile,10):.1f}, p90 {np.percentile(tile,90):.1f}) "
              f"| main loop {np.median(us(t1-t0)):.1f} | epilogue {np.median(epi_t):.2f} (p90 {np.percentile(epi_t,90):.2f}) | gap between workgroups on a CU {np.median(gaps):.2f} us (p90 {np.percentile(gaps,90):.2f}) "
              f"| CUs in epilogue at once: median {int(np.median(conc))}, p90 {int(np.percentile(conc,90))}", flush=True)

if __name__ == "__main__":
    main()
