#!/usr/bin/env python3
"""BASELINE config 1 shape (2k hits / 12k edges, latent 32) on the GPU: eager vs captured HIP graph."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import synth
from hierarchicalgnn_amd.models import EC_InteractionGNN, GraphedInference

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hp = dict(spatial_channels=3, latent=L, hidden=2 * L, n_interaction_graph_iters=14, nb_node_layer=3,
          nb_edge_layer=2, output_layers=3, hidden_output_activation="GELU", hidden_activation="GELU",
          layernorm=True, share_weight=False)
torch.manual_seed(0)
model = EC_InteractionGNN(hp).cuda().eval()
x, ei = synth.trackml_event(2000, 12000, seed=1)
x, ei = x.cuda(), ei.cuda()


def wall(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    eager = wall(lambda: model(x, ei))
g = GraphedInference(model, x, ei)
graphed = wall(lambda: g())
print(json.dumps({"latent": L, "hits": 2000, "edges": 12000, "eager_ms": eager, "hip_graph_replay_ms": graphed}, indent=1))
