#!/usr/bin/env python3
"""hgnn_mlp_forward_f32_split3 (fp32 MLP, split-bf16 GEMMs, opt-in) against the exact fp32 fused kernel and an fp64
torch evaluation on small ragged cases, then the edge-update time of both at M = 2M (A/B in one process)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import fused, make_mlp, mlp, synth

res = {}
for L, layers, nseg, M in [(256, 2, 3, 64), (256, 2, 3, 333), (128, 2, 3, 1000), (256, 3, 2, 200), (128, 3, 3, 77),
                           (256, 2, 1, 130), (256, 2, 3, 40000), (128, 2, 3, 50001)]:
    torch.manual_seed(L + layers * 7 + nseg)
    out_act = "Tanh" if layers == 2 else "GELU"
    net = make_mlp(nseg * L, 2 * L, L, layers, layer_norm=True, output_activation=out_act, hidden_activation="GELU").cuda()
    for p in net.parameters():
        if p.dim() == 1:
            p.data.add_(0.2 * torch.randn_like(p))
    n_tab = max(8, M // 20)
    table = torch.randn(n_tab, L, device="cuda")
    i0 = torch.randint(0, n_tab, (M,), device="cuda")
    i1 = torch.randint(0, n_tab, (M,), device="cuda")
    direct = torch.randn(M, L, device="cuda")
    segs = [(table, i0), (table, i1), (direct, None)][3 - nseg:]
    with torch.no_grad():
        fused.set_fp32_split3(False)
        exact = fused.fused_concat_mlp(net, segs, direct)
        fused.set_fp32_split3(True)
        n0 = fused.stats.get("split3_calls", 0)
        fast = fused.fused_concat_mlp(net, segs, direct)
        assert fused.stats.get("split3_calls", 0) == n0 + 1
        fused.set_fp32_split3(False)
        x = torch.cat([t.double() if i is None else t.double()[i] for t, i in segs], dim=1)
        ref = net.double()(x) + direct.double()
        net.float()
    sc = float(ref.abs().max())
    res[f"L{L}_n{layers}_s{nseg}_M{M}"] = {"split3_vs_fp64": float((fast.double() - ref).abs().max()) / sc,
                                          "exact_vs_fp64": float((exact.double() - ref).abs().max()) / sc}
print(json.dumps(res, indent=1), flush=True)

x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
tim = {}
for L in (256, 128):
    torch.manual_seed(0)
    net = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
    nodes = torch.randn(N, L, device="cuda")
    edges = torch.randn(M, L, device="cuda")
    seg = [(nodes, graph[0]), (nodes, graph[1]), (edges, None)]
    for name, flag in [("exact", False), ("split3", True), ("exact_again", False), ("split3_again", True)]:
        fused.set_fp32_split3(flag)
        with torch.no_grad():
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
        tim[f"L{L}_{name}_ms"] = round(ts[len(ts) // 2], 4)
    del nodes, edges
fused.set_fp32_split3(False)
print(json.dumps(tim))
