#!/usr/bin/env python3
"""A few launches of the feature-split bf16 edge MLP at the BASELINE shape (rocprofv3 target).  Usage: run_mlp_split_once.py [L]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import make_mlp, mlp, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
net = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
nodes = torch.randn(N, L, device="cuda").bfloat16()
edges = torch.randn(M, L, device="cuda").bfloat16()
seg = [(nodes, graph[0]), (nodes, graph[1]), (edges, None)]
with torch.no_grad():
    for _ in range(4):
        mlp.concat_mlp(net, seg, skip=edges)
torch.cuda.synchronize()
