#!/usr/bin/env python3
"""Run only the bf16 fused edge MLP a few times (for rocprofv3 passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import make_mlp, mlp, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
net = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
nodes = torch.randn(120_000, L, device="cuda").bfloat16()
edges = torch.randn(graph.shape[1], L, device="cuda").bfloat16()
with torch.no_grad():
    for _ in range(3):
        out = mlp.concat_mlp(net, [(nodes, graph[0]), (nodes, graph[1]), (edges, None)], skip=edges)
torch.cuda.synchronize()
print("ok", out.dtype)
