"""Where does the HOST time of HybridSearch.search go?  A small encoder (so that the GPU side is short), a 50 k-document corpus in two chunks,
1000 queries, top_k = 1000 -- the reference's evaluation shape (eval/call_evaluate_mteb.sh:8-10) -- under cProfile."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from transformers import PreTrainedTokenizerFast
from lightretriever_amd import EncoderConfig, LrxEncoder
from lightretriever_amd.modeling import LrxExactSearchModel, LrxHybridModel
from lightretriever_amd.retriever import HybridSearch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tok = PreTrainedTokenizerFast.from_pretrained(os.path.join(ROOT, "tests", "golden", "tok"))
cfg = EncoderConfig(vocab_size=max(512, len(tok)), hidden_size=256, num_layers=2, num_q_heads=4, num_kv_heads=2, head_dim=64, intermediate_size=512,
                    rope_type="default", max_positions=128)
enc = LrxEncoder.random_init(cfg, seed=0)
model = LrxExactSearchModel(model=LrxHybridModel(enc, normalize=True, pad_token_id=tok.pad_token_id), tokenizer=tok, q_max_len=32, p_max_len=64,
                            eval_batch_size_embedding_bag=512)
model.query_prompt = "query: "
rng = np.random.default_rng(0)
words = ["".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), size=rng.integers(2, 9))) for _ in range(3000)]
N, Q = int(os.environ.get("N", 50000)), int(os.environ.get("Q", 1000))
corpus = {"d%d" % i: {"title": "", "text": " ".join(rng.choice(words, size=int(rng.integers(5, 30))))} for i in range(N)}
queries = {"q%d" % i: " ".join(rng.choice(words, size=int(rng.integers(3, 9)))) for i in range(Q)}
hs = HybridSearch(model, batch_size=256, corpus_chunk_size=25000)
hs.search(dict(list(corpus.items())[:2000]), dict(list(queries.items())[:50]), top_k=100)      # warm-up (EmbeddingBag table, workspaces)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = hs.search(corpus, queries, top_k=1000)
pr.disable()
print("search(): %.2f s for %d documents x %d queries, top-1000 (%d result dicts)" % (time.perf_counter() - t0, N, Q, len(res)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
