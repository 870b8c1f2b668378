"""Text -> embeddings through the reference-shaped boundary (B2 `encode_corpus`, SURVEY 8a-1 + 8b), host side included:
format_text + tokenizer (the synthetic byte-level BPE of tests/golden/tok; no released tokenizer offline) + packed collation +
H2D of the ids + the HIP encoder writing into its output rows.  Reports the tokeniser-only rate, the packed-ids GPU rate and
the end-to-end rate, so a host-bound pipeline shows up as a gap between the last two.

  python tools/bench_text_e2e.py [--docs 4096] [--batch 256] [--model llama3.2-1b]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--model", default="llama3.2-1b")
    ap.add_argument("--sparse", action="store_true")
    ap.add_argument("--words", type=int, default=400, help="words per document (400 -> truncated at 512 tokens; 40 -> short documents)")
    ap.add_argument("--max-batch-tokens", type=int, default=131072, help="token budget of encode_corpus's batch merging (0 = off)")
    args = ap.parse_args()
    from transformers import PreTrainedTokenizerFast
    from lightretriever_amd import EncoderConfig, LrxEncoder
    from lightretriever_amd.modeling import EncodeCollator, LrxExactSearchModel, LrxHybridModel
    tok = PreTrainedTokenizerFast.from_pretrained(os.path.join(ROOT, "tests", "golden", "tok"))
    cfg = {"llama3.2-1b": EncoderConfig.llama32_1b, "llama3.2-3b": EncoderConfig.llama32_3b, "llama3.1-8b": EncoderConfig.llama31_8b,
           "qwen2.5-1.5b": EncoderConfig.qwen25_1_5b, "qwen2.5-3b": EncoderConfig.qwen25_3b, "qwen2.5-7b": EncoderConfig.qwen25_7b}[args.model]()
    enc = LrxEncoder.random_init(cfg, seed=0)
    hm = LrxHybridModel(enc, normalize=True, pad_token_id=tok.pad_token_id, encode_sparse=args.sparse) if args.sparse else \
        LrxHybridModel(enc, normalize=True, pad_token_id=tok.pad_token_id)
    model = LrxExactSearchModel(model=hm, tokenizer=tok, q_max_len=512, p_max_len=512, max_batch_tokens=args.max_batch_tokens)

    rng = np.random.default_rng(0)
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))
    words = ["".join(rng.choice(letters, size=rng.integers(2, 10))) for _ in range(5000)]
    docs = [{"title": " ".join(rng.choice(words, size=6)), "text": " ".join(rng.choice(words, size=max(1, int(rng.integers(args.words // 2, args.words + 1)))))} for _ in range(args.docs)]

    coll = EncodeCollator(tok, encode_is_query=False, p_max_len=512)
    b0 = coll(docs[:args.batch])
    lens = np.diff(b0["cu_seqlens"].numpy())
    print("tokens/doc after truncation: mean %.0f, max %d" % (lens.mean(), lens.max()))
    t0 = time.perf_counter()
    packed = [coll(docs[s:s + args.batch]) for s in range(0, args.docs, args.batch)]
    t_tok = time.perf_counter() - t0
    print("host only (format + tokenise + pack): %.0f docs/s on %d host threads" % (args.docs / t_tok, os.cpu_count()))

    out = torch.empty(args.docs, cfg.hidden_size, dtype=torch.float32, device="cuda")
    for _ in range(2):                      # GPU only, packed ids already collated
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, p in enumerate(packed):
            hm.encode_passage(p, out=out[i * args.batch:(i + 1) * args.batch])
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0
    print("device only (packed ids -> rows): %.0f docs/s" % (args.docs / t_gpu))

    for _ in range(2):                      # end to end through encode_corpus
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = model.encode_corpus(docs, batch_size=args.batch, out=out)
        torch.cuda.synchronize()
        t_e2e = time.perf_counter() - t0
    assert torch.isfinite(res["dense_reps"]).all()
    print("end to end (texts -> rows, encode_corpus): %.0f docs/s  (host %.2f s, device %.2f s, end to end %.2f s)"
          % (args.docs / t_e2e, t_tok, t_gpu, t_e2e))


if __name__ == "__main__":
    main()
