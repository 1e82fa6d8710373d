#!/usr/bin/env python3
"""kNN graph rebuild (K9) at the BASELINE hierarchy shape: bipartite (120k hits -> ~10k centres, K=5) and
super graph (~10k -> ~10k, K=10) in emb_dim = 8."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd.ops import knn_radius

torch.manual_seed(0)
emb = torch.nn.functional.normalize(torch.randn(120_000, 8, device="cuda"))
means = torch.nn.functional.normalize(torch.randn(10_000, 8, device="cuda"))


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


res = {"bipartite_120k_x_10k_K5_ms": timeit(lambda: knn_radius(emb, means, 5, 2.0)),
       "super_10k_x_10k_K10_ms": timeit(lambda: knn_radius(means, means, 10, 2.0))}
print(json.dumps(res, indent=1))
