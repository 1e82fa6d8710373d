#!/usr/bin/env python3
"""A few launches of the fp32 edge update on the split-bf16 kernel (k_mlp_f32_split3) at the BASELINE shape, on the
destination-sorted layout the model blocks run in (rocprofv3 target).  Usage: run_mlp_split3_once.py [L]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import fused, make_mlp, mlp, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
x, ei = synth.trackml_event()
graph = synth.directed(ei)
graph = graph[:, torch.argsort(graph[1], stable=True)].contiguous().cuda()
N, M = 120_000, graph.shape[1]
net = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
nodes = torch.randn(N, L, device="cuda")
edges = torch.randn(M, L, device="cuda")
seg = [(nodes, graph[0]), (nodes, graph[1]), (edges, None)]
fused.set_fp32_split3(True)
with torch.no_grad():
    for _ in range(4):
        mlp.concat_mlp(net, seg, skip=edges)
torch.cuda.synchronize()
