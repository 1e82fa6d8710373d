#!/usr/bin/env python3
"""bf16 feature-split edge MLP alone (M = 2M rows): time, TFLOP/s, error vs the fp32 kernel.  Usage: time_split.py L [L ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import make_mlp, mlp, synth

x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
out = {}
for L in [int(a) for a in sys.argv[1:]] or [256]:
    torch.manual_seed(0)
    net = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
    nodes = torch.randn(N, L, device="cuda").bfloat16()
    edges = torch.randn(M, L, device="cuda").bfloat16()
    seg = [(nodes, graph[0]), (nodes, graph[1]), (edges, None)]
    flop = 2 * (3 * L * 2 * L + 2 * L * L) * M
    with torch.no_grad():
        for _ in range(3):
            a = mlp.concat_mlp(net, seg, skip=edges)
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
        t = ts[len(ts) // 2]
        x32 = torch.cat([nodes[graph[0][:20000]].float(), nodes[graph[1][:20000]].float(), edges[:20000].float()], 1)
        ref = net(x32) + edges[:20000].float()
        err = float((a[:20000].float() - ref).abs().max() / ref.abs().max())
    out[f"L{L}"] = {"ms": round(t, 4), "tflops": round(flop / t / 1e9, 1), "frac_of_2.5PF": round(flop / t / 1e9 / 2500, 3),
                    "rel_err_vs_fp32": err}
    del net, nodes, edges, a
print(json.dumps(out))
