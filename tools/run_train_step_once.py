#!/usr/bin/env python3
"""A few InteractionGNNCell training steps (fused forward with dumps + hand-written backward, no
checkpointing) at the BASELINE shape: rocprofv3 target.  Usage: run_train_step_once.py [L] [ckpt]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hierarchicalgnn_amd as H
from hierarchicalgnn_amd import fused, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ckpt = len(sys.argv) > 2 and sys.argv[2] == "ckpt"
torch.manual_seed(0)
x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
hp = dict(latent=L, hidden=2 * L, nb_edge_layer=2, nb_node_layer=3, layernorm=True, hidden_activation="GELU",
          checkpointing=ckpt)
cell = H.InteractionGNNCell(hp).cuda()
nodes = torch.randn(N, L, device="cuda", requires_grad=True)
edges = torch.randn(M, L, device="cuda", requires_grad=True)
fused.set_enabled(True, train=True)
for _ in range(4):
    on, oe = cell(nodes, edges, graph)
    (on.sum() + oe.sum()).backward()
    cell.zero_grad(set_to_none=True)
    nodes.grad = None
    edges.grad = None
torch.cuda.synchronize()
