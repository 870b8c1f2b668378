#!/usr/bin/env python3
"""Check on a rocprofv3 kernel trace of bench.py: inside the HEADLINE legs -- between the two marker kernels bench.py launches,
k_trace_marker<0> (before the first encode step, warm-up included) and k_trace_marker<1> (after the last pass of the timed search leg) --
no kernel may run that is not liblrx's own: index maintenance, query embedding, search and the exchange's pack / merge are all liblrx
kernels (the RCCL kernels of a multi-rank exchange are the one allowed exception).  Every at::native (or any other foreign) kernel in
that window is counted, wherever it sits.  usage: python tools/check_trace_clean.py <..._kernel_trace.csv> [min_searches=20]"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
try:
    first = next(i for i, n in enumerate(names) if "k_trace_marker<0>" in n)
    last = next(i for i, n in enumerate(names) if i > first and "k_trace_marker<1>" in n)
except StopIteration:
    print("markers k_trace_marker<0> / <1> not found in the trace")
    sys.exit(2)
window = names[first + 1:last]
own = lambda n: (n[5:] if n.startswith("void ") else n).startswith("k_") or "nccl" in n.lower() or "rccl" in n.lower()
# the HIP runtime's own blit kernels (hipMemcpyAsync: bench.py moves its timing scalars with them) are not compute kernels of anybody: listed, not failed
blits = [n for n in window if n.startswith("__amd_rocclr_")]
bad = collections.Counter(n[:90] for n in window if not own(n) and not n.startswith("__amd_rocclr_"))
n_search = sum(1 for n in window if "k_sample_threshold" in n)
n_encode = sum(1 for n in window if "k_embedding_gather" in n or "k_embed_stream32" in n)
print("kernels between the markers of the headline legs:", len(window), "(%d encode steps, %d searches)" % (n_encode, n_search),
      "| foreign (at::native ...) among them:", sum(bad.values()), "| runtime memcpy blits:", len(blits))
for k, v in bad.most_common():
    print("  ", v, k)
min_search = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sys.exit(1 if bad or n_search < min_search else 0)
