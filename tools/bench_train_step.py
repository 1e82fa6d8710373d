#!/usr/bin/env python3
"""Secondary figures at the BASELINE shape: (a) one InteractionGNNCell forward+backward under
reentrant checkpointing (the way the reference trains), (b) one HierarchicalGNNCell inference
forward with the synthetic hierarchy (S=10k, B=600k, Q~200k).  Usage: bench_train_step.py [L]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hierarchicalgnn_amd as H
from hierarchicalgnn_amd import fused, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
hp = dict(latent=L, hidden=2 * L, nb_edge_layer=2, nb_node_layer=3, layernorm=True, hidden_activation="GELU")


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
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


res = {"L": L, "N": N, "M": M}
cell = H.InteractionGNNCell(hp).cuda()
nodes = torch.randn(N, L, device="cuda", requires_grad=True)
edges = torch.randn(M, L, device="cuda", requires_grad=True)


def train_step():
    on, oe = cell(nodes, edges, graph)
    (on.sum() + oe.sum()).backward()
    cell.zero_grad(set_to_none=True)
    nodes.grad = None
    edges.grad = None


for ckpt in (True, False):
    cell._ckpt = ckpt
    for name, fwd, trn in (("fused_fwd+fused_train", True, True), ("fused_fwd+library_bwd", True, False),
                           ("library_only", False, False)):  # noqa
        fused.set_enabled(fwd, train=trn)
        res[f"ignn_cell_fwd_bwd_{'checkpointed' if ckpt else 'no_checkpoint'}_{name}_ms"] = timeit(train_step, 3, 1)
        res[f"peak_mem_GB_{'ckpt' if ckpt else 'nockpt'}_{name}"] = torch.cuda.max_memory_allocated() / 2**30
        torch.cuda.reset_peak_memory_stats()
cell._ckpt = True
fused.set_enabled(True)

S = 10_000
bg, bw = synth.bipartite_assignment(N, S, 5)
sg, sw = synth.super_graph(S, 10)
bg, bw, sg, sw = bg.cuda(), bw.cuda(), sg.cuda(), sw.cuda()
hcell = H.HierarchicalGNNCell(hp).cuda()
sn = torch.randn(S, L, device="cuda")
se = torch.randn(sg.shape[1], L, device="cuda")
res["B"] = int(bg.shape[1])
res["Q"] = int(sg.shape[1])
with torch.no_grad():
    n0, e0 = nodes.detach(), edges.detach()
    for name, on in (("fused", True), ("library", False)):
        fused.set_enabled(on)
        res[f"hgnn_cell_forward_{name}_ms"] = timeit(lambda: hcell(n0, e0, sn, se, graph, bg, bw, sg, sw), 3, 1)
    fused.set_enabled(True)
print(json.dumps(res, indent=1))
