"""A/B of the tiled attention kernels: the walker kernel (no work list) against the work-list kernel (include/lrx.h, ABI 7).
  python3 tools/exp/attn_ab.py save <old|new>  -> writes outputs to gpurun_out/attn_ab_<tag>.pt and prints timings (old = no work list)
  python3 tools/exp/attn_ab.py cmp A B        -> compares two saved runs bit for bit
Cases: the 8B-class geometry at 256 x 512 tokens (the VERDICT's figure), ragged lengths, GQA groups 1/2/3/6/8 at d = 128, d = 64 tiled, last-tile mode."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch


def cases():
    g = torch.Generator().manual_seed(1)
    yield "8b_256x512", 32, 8, 128, [512] * 256, False
    yield "8b_ragged", 32, 8, 128, [int(x) for x in torch.randint(1, 513, (300,), generator=g)], False
    yield "8b_last_tile", 32, 8, 128, [int(x) for x in torch.randint(1, 513, (300,), generator=g)], True
    yield "d128_g1", 8, 8, 128, [int(x) for x in torch.randint(1, 300, (64,), generator=g)], False
    yield "d128_g2", 8, 4, 128, [int(x) for x in torch.randint(1, 700, (64,), generator=g)], False
    yield "d128_g3", 12, 4, 128, [int(x) for x in torch.randint(1, 300, (64,), generator=g)], False
    yield "d128_g6", 12, 2, 128, [int(x) for x in torch.randint(1, 300, (64,), generator=g)], False
    yield "d128_g8", 16, 2, 128, [int(x) for x in torch.randint(1, 300, (64,), generator=g)], False
    yield "d64_g4_long", 32, 8, 64, [int(x) for x in torch.randint(1, 1500, (40,), generator=g)], False
    yield "d64_g7", 14, 2, 64, [int(x) for x in torch.randint(1, 900, (40,), generator=g)], False
    yield "tiny", 32, 8, 128, [1, 2, 63, 64, 65], False


def main():
    from lightretriever_amd import ops
    mode = sys.argv[1]
    if mode == "cmp":
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        bad = 0
        for k in a:
            same = torch.equal(a[k], b[k])
            d = (a[k].float() - b[k].float()).abs().max().item()
            print("%-16s %s max|diff| %.3g  nan %d/%d" % (k, "identical" if same else "DIFFERENT", d, int(torch.isnan(a[k].float()).sum()), int(torch.isnan(b[k].float()).sum())))
            bad += not same
        sys.exit(1 if bad else 0)
    tag = sys.argv[2]
    outs = {}
    for name, nq, nkv, d, lens, last in cases():
        gg = torch.Generator(device="cuda").manual_seed(7)
        T = sum(lens)
        qkv = torch.randn(T, (nq + 2 * nkv) * d, generator=gg, device="cuda").to(torch.float16)
        cu = torch.tensor([0] + lens, dtype=torch.int64).cumsum(0).to(torch.int32).cuda()
        wl = False if tag == "old" else None
        o = ops.attn_varlen_causal(qkv, cu, max(lens), nq, nkv, d, last_tile_only=last, work_list=wl)
        torch.cuda.synchronize()
        if last:   # only the last q tile of each sequence is defined: keep the last rows
            idx = (cu[1:] - 1).long()
            o = o[idx]
        outs[name] = o.cpu()
        if name == "8b_256x512":
            if tag != "old":
                wl = ops.attn_work_list(cu, T, 512, nq, nkv, d)          # as the encoder does: one list per batch, every layer's launch reads it
            for _ in range(3):
                ops.attn_varlen_causal(qkv, cu, 512, nq, nkv, d, work_list=wl)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.attn_varlen_causal(qkv, cu, 512, nq, nkv, d, work_list=wl)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            fl = 4.0 * 128 * 32 * 256 * (512 * 513 / 2)
            print("%s %s: %.4f ms/launch  %.0f TFLOP/s causal" % (tag, name, ms, fl / ms / 1e9), flush=True)
            if tag != "old":
                e0.record()
                for _ in range(20):
                    ops.attn_work_list(cu, T, 512, nq, nkv, d)
                e1.record(); torch.cuda.synchronize()
                print("   work list build: %.4f ms" % (e0.elapsed_time(e1) / 20), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    torch.save(outs, "gpurun_out/attn_ab_%s.pt" % tag)


main()
