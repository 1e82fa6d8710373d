#!/usr/bin/env python3
"""K1 on the full-pileup shape of BASELINE config 5 (N=480,000 hits, E=4,000,000 -> M=8,000,000
directed rows, latent 256) on ONE GPU, next to the headline shape."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hierarchicalgnn_amd as H
from hierarchicalgnn_amd import synth

out = {}
for name, n, e in (("1GeV", 120_000, 1_000_000), ("full_pileup", 480_000, 4_000_000)):
    x, ei = synth.trackml_event(n, e)
    graph = synth.directed(ei).cuda()
    M, L = graph.shape[1], 256
    src = torch.randn(M, L, device="cuda")
    plan = H.get_plan(graph[1], n)
    for _ in range(3):
        H.scatter_add(src, graph[1], dim_size=n, plan=plan)
    ts = []
    for _ in range(20):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        H.scatter_add(src, graph[1], dim_size=n, plan=plan)
        t.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(t))
    ts.sort()
    ms = ts[len(ts) // 2]
    b = 4 * L * M + 4 * M + 4 * L * n
    out[name] = {"N": n, "M": M, "ms": ms, "GBps": b / ms / 1e6, "frac_of_8TBps": b / ms / 1e6 / 8000,
                 "edges_per_s": M / ms * 1e3, "degree": synth.degree_stats(graph[1].cpu(), n)}
    del src, graph, plan
    H.clear_plan_cache()
print(json.dumps(out, indent=1))
