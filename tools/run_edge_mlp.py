#!/usr/bin/env python3
"""Run only the fused edge update (for rocprofv3 passes). Usage: run_edge_mlp.py [L] [reps] [M_edges]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hierarchicalgnn_amd as H
from hierarchicalgnn_amd import synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
E = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
torch.manual_seed(0)
x, ei = synth.trackml_event(120_000, E)
graph = synth.directed(ei).cuda()
nodes = torch.randn(120_000, L, device="cuda")
edges = torch.randn(graph.shape[1], L, device="cuda")
hp = dict(latent=L, hidden=2 * L, nb_edge_layer=2, nb_node_layer=3, layernorm=True, hidden_activation="GELU")
cell = H.InteractionGNNCell(hp).cuda()
with torch.no_grad():
    for _ in range(reps):
        out = cell._edge_update(nodes, edges, graph)
        out2 = cell._node_update(nodes, edges, graph)
torch.cuda.synchronize()
print("ok", out.shape, float(out.abs().mean()))
