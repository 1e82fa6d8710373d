#!/usr/bin/env python3
"""A/B the row-gather kernel (K6 / backward of K1) store policy in one process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hierarchicalgnn_amd as H
from hierarchicalgnn_amd import _lib, synth
from hierarchicalgnn_amd.ops import _gather_rows, _spread_rows

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
table = torch.randn(N, L, device="cuda")
plan = H.get_plan(graph[1], N)
bytes_ = 4 * L * M + 4 * M + 4 * L * N
res = {}
for rnd in range(5):
    for nt in (0, 1):
        _lib.check(lib.hgnn_set_option(b"nt_stores", nt))
        _gather_rows(table, plan.dst32, M)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            _gather_rows(table, plan.dst32, M)
        e.record()
        torch.cuda.synchronize()
        res.setdefault(nt, []).append(s.elapsed_time(e) / 10)
for nt, t in res.items():
    t.sort()
    print(f"nt_stores={nt}: median {t[len(t)//2]*1e3:.1f} us  {bytes_/t[len(t)//2]/1e6:.0f} GB/s")
ref = _gather_rows(table, plan.dst32, M)
for nt in (0, 1):
    _lib.check(lib.hgnn_set_option(b"nt_stores", nt))
    out = _spread_rows(plan, table)
    assert torch.equal(out, ref)
    ts = []
    for rnd in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            _spread_rows(plan, table)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 10)
    ts.sort()
    print(f"spread_rows nt_stores={nt}: median {ts[2]*1e3:.1f} us  {bytes_/ts[2]/1e6:.0f} GB/s")
_lib.check(lib.hgnn_set_option(b"nt_stores", 0))
# reference point: a plain device copy of the same number of bytes
a = torch.empty(M, L, device="cuda")
b = torch.randn(M, L, device="cuda")
a.copy_(b)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    a.copy_(b)
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / 10
print(f"torch copy_ [M,L]: {t*1e3:.1f} us  (read+write {2*4*L*M/t/1e6:.0f} GB/s)")
s.record()
for _ in range(10):
    a.fill_(1.0)
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / 10
print(f"torch fill_ [M,L]: {t*1e3:.1f} us  (write {4*L*M/t/1e6:.0f} GB/s)")
