#!/usr/bin/env python3
"""VERDICT r1 item 3 check on a rocprofv3 kernel trace of bench.py: between the first encode launch (k_embedding_gather) and the last
search launch (k_refine_merge / k_merge_topk) no at::native:: kernel may run -- index maintenance, query embedding, search and
exchange are all liblrx kernels.  usage: python tools/check_trace_clean.py <..._kernel_trace.csv>"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
first = next(i for i, n in enumerate(names) if "k_embedding_gather" in n)
last = max(i for i, n in enumerate(names) if "k_refine_merge" in n or "k_merge_topk" in n)
bad = collections.Counter(n[:90] for n in names[first:last + 1] if "at::" in n)
print("kernels between first encode and last search launch:", last - first + 1, "| at::native among them:", sum(bad.values()))
for k, v in bad.most_common():
    print("  ", v, k)
sys.exit(1 if bad else 0)
