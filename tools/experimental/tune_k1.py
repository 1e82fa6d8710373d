#!/usr/bin/env python3
"""A/B the K1 segmented-reduce launch knobs in ONE process, interleaved rounds
(cdna_hip_programming.md 5.4 rule 24).  Usage: python tools/tune_k1.py [L]"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hierarchicalgnn_amd as H
from hierarchicalgnn_amd import _lib, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
src = torch.randn(M, L, device="cuda")
plan = H.get_plan(graph[1], N)
alg = 4 * L * M + 4 * M + 4 * L * N


def setopt(**kw):
    for k, v in kw.items():
        _lib.check(lib.hgnn_set_option(k.encode(), int(v)))


variants = [dict(nt_loads=nt, seg_unroll=u, seg_wpb=w, seg_xcd=x)
            for nt, u, w, x in itertools.product((1, 0), (2, 4, 8, 16), (4, 8, 16), (0, 1))
            if not (w != 4 and u == 2)]
times = {i: [] for i in range(len(variants))}
ref = None
for rnd in range(6):
    for i, v in enumerate(variants):
        setopt(**v)
        out = H.scatter_add(src, graph[1], dim=0, dim_size=N, plan=plan)
        if ref is None:
            ref = out.clone()
        elif rnd == 0:
            assert torch.equal(out, ref), v  # every variant sums in the same order
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            H.scatter_add(src, graph[1], dim=0, dim_size=N, plan=plan)
        e.record()
        torch.cuda.synchronize()
        times[i].append(s.elapsed_time(e) / 10)
rows = []
for i, v in enumerate(variants):
    t = sorted(times[i])
    med, mn = t[len(t) // 2], t[0]
    rows.append((med, mn, v))
rows.sort(key=lambda r: r[0])
print(f"L={L} M={M} N={N} algorithmic bytes={alg}")
for med, mn, v in rows:
    print(f"median {med*1e3:7.1f} us  min {mn*1e3:7.1f} us  {alg/med/1e6:7.0f} GB/s  {v}")
