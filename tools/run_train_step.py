#!/usr/bin/env python3
"""A few checkpointed InteractionGNNCell training steps (for rocprofv3). Usage: run_train_step.py [L] [train_fused 0/1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hierarchicalgnn_amd as H
from hierarchicalgnn_amd import fused, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fused.set_enabled(True, train=bool(int(sys.argv[2])) if len(sys.argv) > 2 else True)
torch.manual_seed(0)
x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
hp = dict(latent=L, hidden=2 * L, nb_edge_layer=2, nb_node_layer=3, layernorm=True, hidden_activation="GELU")
cell = H.InteractionGNNCell(hp).cuda()
nodes = torch.randn(120_000, L, device="cuda", requires_grad=True)
edges = torch.randn(graph.shape[1], L, device="cuda", requires_grad=True)
for _ in range(3):
    on, oe = cell(nodes, edges, graph)
    (on.sum() + oe.sum()).backward()
torch.cuda.synchronize()
print("ok")
