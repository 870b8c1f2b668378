#!/usr/bin/env python3
"""What does a dependent kernel launch cost on this box, whatever the kernel does?  N back-to-back launches of liblrx's empty marker kernel
(one wave) on one stream between two events; the same through a captured HIP graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lightretriever_amd import _lib
lib = _lib.lib()
st = _lib.current_stream()
def run(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        lib.lrx_trace_marker(3, st)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
run(200)
print("empty kernel, eager, back to back: %.2f us per launch (2000 launches)" % run(2000))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    lib.lrx_trace_marker(3, _lib.current_stream())
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(100):
            lib.lrx_trace_marker(3, _lib.current_stream())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(20):
    g.replay()
e1.record()
torch.cuda.synchronize()
print("empty kernel, graph of 100, replayed 20 times: %.2f us per launch" % (1e3 * e0.elapsed_time(e1) / 2000))
