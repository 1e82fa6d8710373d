#!/usr/bin/env python3
"""Where the bf16 feature-split edge MLP spends its time: hgnn_set_option("mlp_ablate", bits) A/B in one process
(results are WRONG with any bit set; only the time matters).  bits: 1 weights from chunk 0 only, 2 no LayerNorm /
activation, 4 only the first input panel loaded, 8 no per-panel barriers, 16 LDS operand reads from chunk 0 only."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import _lib, make_mlp, mlp, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
torch.manual_seed(0)
net = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
nodes = torch.randn(N, L, device="cuda").bfloat16()
edges = torch.randn(M, L, device="cuda").bfloat16()
seg = [(nodes, graph[0]), (nodes, graph[1]), (edges, None)]
lib = _lib.load()
out = {}
with torch.no_grad():
    for bits in (0, 2, 1, 4, 8, 16, 3, 7, 31, 0):
        lib.hgnn_set_option(b"mlp_ablate", bits)
        for _ in range(3):
            mlp.concat_mlp(net, seg, skip=edges)
        torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            mlp.concat_mlp(net, seg, skip=edges)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        ts.sort()
        out[f"ablate_{bits}" + ("_again" if f"ablate_{bits}" in out else "")] = round(ts[len(ts) // 2], 4)
lib.hgnn_set_option(b"mlp_ablate", 0)
print(json.dumps(out))
