#!/usr/bin/env python3
"""VERDICT r1 item 3 check on a rocprofv3 kernel trace of bench.py: inside the HEADLINE legs -- from the first encode launch (k_embedding_gather)
to the last search launch (k_refine_merge / k_merge_topk) of the timed search leg -- no at::native:: kernel may run: index maintenance,
query embedding, search and exchange are all liblrx kernels.  The headline legs end where the first kernel of a later leg appears: the
HIP-graph probe's query copy, the LM-head GEMM of the sparse leg (k_gemm_bf16_nt<4>) or the synthetic indexes of the extra single-GPU
legs (their fills are torch kernels by design).  usage: python tools/check_trace_clean.py <..._kernel_trace.csv>"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
first = next(i for i, n in enumerate(names) if "k_embedding_gather" in n)
# the timed search leg is the first run of >= 20 consecutive searches (k_embedding_bag + search chain) after the encode steps; everything up
# to the first at::native kernel or sparse-leg GEMM that follows the first search launch belongs to the headline legs
first_search = next(i for i, n in enumerate(names) if i > first and "k_sample_threshold" in n)
end = next((i for i, n in enumerate(names) if i > first_search and ("at::" in n or "k_gemm_bf16_nt<4>" in n)), len(names))
last = max(i for i, n in enumerate(names[:end]) if "k_refine_merge" in n or "k_merge_topk" in n)
bad = collections.Counter(n[:90] for n in names[first:last + 1] if "at::" in n)
n_search = sum(1 for n in names[first:last + 1] if "k_sample_threshold" in n)
print("kernels between the first encode launch and the last launch of the headline search legs:", last - first + 1, "(%d searches)" % n_search,
      "| at::native among them:", sum(bad.values()))
for k, v in bad.most_common():
    print("  ", v, k)
sys.exit(1 if bad or n_search < 20 else 0)
