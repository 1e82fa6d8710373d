#!/usr/bin/env python3
"""Price the parts of the fused edge MLP with its diagnostic ablation switches (one process,
interleaved rounds).  Ablated variants compute WRONG results; only their time is meaningful."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hierarchicalgnn_amd as H
from hierarchicalgnn_amd import _lib, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
torch.manual_seed(0)
x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
nodes = torch.randn(120_000, L, device="cuda")
edges = torch.randn(graph.shape[1], L, device="cuda")
hp = dict(latent=L, hidden=2 * L, nb_edge_layer=2, nb_node_layer=3, layernorm=True, hidden_activation="GELU")
cell = H.InteractionGNNCell(hp).cuda()
flop = 2 * (3 * L * 2 * L + 2 * L * L) * graph.shape[1]
variants = [("real", 0), ("no LN/act", 1), ("no weight DMA", 2), ("no LN/act, no DMA", 3), ("no barriers", 4)]  # 0 = real kernel; -(1000+bits): ablations (wrong results)
times = {v: [] for _, v in variants}
ref = None
with torch.no_grad():
    for rnd in range(4):
        for _, v in variants:
            _lib.check(lib.hgnn_set_option(b"mlp_ablate", v))
            out = cell._edge_update(nodes, edges, graph)
            if ref is None:
                ref = out.clone()
            if v == 0:
                assert torch.equal(out, ref)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                cell._edge_update(nodes, edges, graph)
            e.record()
            torch.cuda.synchronize()
            times[v].append(s.elapsed_time(e) / 3)
_lib.check(lib.hgnn_set_option(b"mlp_ablate", 0))
for name, v in variants:
    t = sorted(times[v])
    print(f"{name:22s}: median {t[len(t)//2]:.3f} ms  min {t[0]:.3f} ms  {flop/t[len(t)//2]/1e9:.1f} TF/s")
