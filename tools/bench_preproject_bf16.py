#!/usr/bin/env python3
"""bf16 feature-split edge MLP with / without pre-projected node segments (hgnn_mlp_desc.n_pre), A/B in one process;
the projected arm includes its two N-row projection GEMMs and the weight slicing."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import fused, make_mlp, mlp, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
torch.manual_seed(0)
net = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
nodes = torch.randn(N, L, device="cuda").bfloat16()
edges = torch.randn(M, L, device="cuda").bfloat16()
seg = [(nodes, graph[0]), (nodes, graph[1]), (edges, None)]
out, res = {}, {}
with torch.no_grad():
    for name, flag in [("direct", False), ("projected", True), ("direct_again", False), ("projected_again", True)]:
        fused._preproject_bf16 = flag
        for _ in range(3):
            res[flag] = mlp.concat_mlp(net, seg, skip=edges)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            mlp.concat_mlp(net, seg, skip=edges)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        ts.sort()
        out[name + "_ms"] = round(ts[len(ts) // 2], 4)
fused._preproject_bf16 = None
d = (res[True].float() - res[False].float()).abs().max() / res[False].float().abs().max()
out["projected_vs_direct_max_rel"] = float(d)
out["L"] = L
print(json.dumps(out))
