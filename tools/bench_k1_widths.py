#!/usr/bin/env python3
"""K1 scatter_add at every BASELINE latent width (32/128/256/512) on the headline event."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hierarchicalgnn_amd as H
from hierarchicalgnn_amd import synth

x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
plan = H.get_plan(graph[1], N)
out = {}
from hierarchicalgnn_amd import _lib
cases = [(L, 0) for L in (32, 64, 128, 256, 512)]   # (the grouped narrow-row variant of round 2 was removed: slower)
for L, grouped in cases:
    src = torch.randn(M, L, device="cuda")
    for _ in range(3):
        H.scatter_add(src, graph[1], dim_size=N, plan=plan)
    ts = []
    for _ in range(20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        H.scatter_add(src, graph[1], dim_size=N, plan=plan)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    t = ts[len(ts) // 2]
    b = 4 * L * M + 4 * M + 4 * L * N
    out[f"L{L}" + ("_grouped" if grouped and L <= 128 else "")] = {"ms": t, "alg_bytes": b, "GBps": b / t / 1e6, "frac_of_8TBps": b / t / 1e6 / 8000,
                    "edges_per_s": M / t * 1e3}
    del src
# the destination-sorted layout the model blocks run in (streaming reads): narrow rows
order = torch.argsort(graph[1], stable=True)
g_sorted = graph[:, order].contiguous()
plan_s = H.get_plan(g_sorted[1], N)
for L in (32, 64, 128, 256):
    src = torch.randn(M, L, device="cuda")
    for _ in range(3):
        H.scatter_add(src, g_sorted[1], dim_size=N, plan=plan_s)
    ts = []
    for _ in range(20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        H.scatter_add(src, g_sorted[1], dim_size=N, plan=plan_s)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    t = ts[len(ts) // 2]
    b = 4 * L * M + 4 * L * N
    out[f"L{L}_sorted_layout"] = {"ms": t, "alg_bytes": b, "GBps": b / t / 1e6, "frac_of_8TBps": b / t / 1e6 / 8000,
                                  "edges_per_s": M / t * 1e3}
    del src
# BASELINE config 4 dtype: latent=512 bf16 (a 1-KiB row again)
src = torch.randn(M, 512, device="cuda").bfloat16()
for _ in range(3):
    H.scatter_add(src, graph[1], dim_size=N, plan=plan)
ts = []
for _ in range(20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    H.scatter_add(src, graph[1], dim_size=N, plan=plan)
    e.record()
    torch.cuda.synchronize()
    ts.append(s.elapsed_time(e))
ts.sort()
t = ts[len(ts) // 2]
b = 2 * 512 * M + 4 * M + 2 * 512 * N
out["L512_bf16"] = {"ms": t, "alg_bytes": b, "GBps": b / t / 1e6, "frac_of_8TBps": b / t / 1e6 / 8000,
                    "edges_per_s": M / t * 1e3}
print(json.dumps(out, indent=1))
