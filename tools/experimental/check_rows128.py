#!/usr/bin/env python3
"""hgnn_mlp_forward_bf16_rows128 against the feature-split kernel and an fp32 torch evaluation (small ragged cases),
then the edge-update time of both kernels at M = 2M (A/B in one process)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import _lib, fused, make_mlp, mlp, synth

lib = _lib.load()
L = 256
res = {}


def run(net, segs, skip, rows128):
    lib.hgnn_set_option(b"mlp_rows128", 1 if rows128 else 0)
    with torch.no_grad():
        return fused.fused_concat_mlp(net, segs, skip)


for layers, nseg, M in [(2, 3, 128), (2, 3, 1), (2, 3, 333), (3, 2, 200), (2, 1, 64), (2, 3, 70000), (3, 3, 33000)]:
    torch.manual_seed(layers * 7 + nseg)
    out_act = "Tanh" if layers == 2 else "GELU"
    net = make_mlp(nseg * L, 2 * L, L, layers, layer_norm=True, output_activation=out_act, hidden_activation="GELU").cuda()
    for p in net.parameters():
        if p.dim() == 1:
            p.data.add_(0.2 * torch.randn_like(p))
    table = torch.randn(97, L, device="cuda").bfloat16()
    i0 = torch.randint(0, 97, (M,), device="cuda")
    i1 = torch.randint(0, 97, (M,), device="cuda")
    direct = torch.randn(M, L, device="cuda").bfloat16()
    segs = [(table, i0), (table, i1), (direct, None)][3 - nseg:]
    a = run(net, segs, direct, True).float()
    b = run(net, segs, direct, False).float()
    x = torch.cat([t.float() if i is None else t.float()[i] for t, i in segs], dim=1)
    ref = x
    mods = list(net)
    with torch.no_grad():
        for m in mods:
            if isinstance(m, torch.nn.Linear):
                ref = torch.nn.functional.linear(ref, m.weight.bfloat16().float(), m.bias)
            else:
                ref = m(ref)
        ref = ref + direct.float()
    sc = float(ref.abs().max())
    res[f"L{layers}_s{nseg}_M{M}"] = {"rows128_vs_fp32": float((a - ref).abs().max()) / sc,
                                      "split_vs_fp32": float((b - ref).abs().max()) / sc,
                                      "rows128_vs_split": float((a - b).abs().max()) / sc}
print(json.dumps(res, indent=1), flush=True)

x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
torch.manual_seed(0)
net = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
nodes = torch.randn(N, L, device="cuda").bfloat16()
edges = torch.randn(M, L, device="cuda").bfloat16()
seg = [(nodes, graph[0]), (nodes, graph[1]), (edges, None)]
tim = {}
abl = [int(v) for v in sys.argv[1:]] or [0]
for name, r128, bits in [("split", 0, 0)] + [(f"rows128_ablate{b}", 1, b) for b in abl] + [("split_again", 0, 0), ("rows128_again", 1, 0)]:
    lib.hgnn_set_option(b"mlp_rows128", r128)
    lib.hgnn_set_option(b"mlp_ablate", bits)
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
    tim[name] = round(ts[len(ts) // 2], 4)
lib.hgnn_set_option(b"mlp_ablate", 0)
lib.hgnn_set_option(b"mlp_rows128", 1)
tim["note"] = "ms per edge update incl. the per-call weight re-layout; 2.10 TFLOP reference-equivalent"
print(json.dumps(tim))
