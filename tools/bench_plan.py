#!/usr/bin/env python3
"""Per-event plan build cost (sort + CSR + work list), excluded from the K1 figure but amortised over
the 14 / 13 aggregations of a forward."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hierarchicalgnn_amd as H
from hierarchicalgnn_amd import synth

x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N = 120_000
H.GraphPlan(graph[1].clone(), N)  # warm-up: module load, allocator
torch.cuda.synchronize()
for validate in (True, False):
    ts = []
    for _ in range(10):
        idx = graph[1].clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        H.GraphPlan(idx, N, validate=validate)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"plan build (M={graph.shape[1]}, N={N}, validate={validate}): median {ts[5]:.3f} ms  min {ts[0]:.3f} ms (host wall, synced)")
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
idx = graph[1].clone()
s.record()
H.GraphPlan(idx, N, validate=False)
e.record()
torch.cuda.synchronize()
print(f"plan build GPU time (events): {s.elapsed_time(e):.3f} ms")
