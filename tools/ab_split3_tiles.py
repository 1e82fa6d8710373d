#!/usr/bin/env python3
"""A/B of the three tile configurations of hgnn_mlp_forward_f32_split3 for K -> 512 -> 256 (the latent-256 edge update):
hgnn_set_option("mlp_split3_rows128", v) with v = 0: 64-row tiles, 8 waves, full hidden planes (one workgroup per CU);
1: 128-row tiles, 8 waves, K-half hidden planes; 2: 64-row tiles, 4 waves, K-half planes, two workgroups per CU.
Error vs fp64 on a ragged case, then the time of the 2M-row edge update (preprojection included, as the cell runs it)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HGNN_EXPERIMENTAL", "1")   # option value 2 (the experimental two-workgroup tile) is refused without it
import torch
from hierarchicalgnn_amd import _lib, fused, make_mlp, mlp, synth

lib = _lib.load()
names = {0: "rows64_8waves", 1: "rows128_8waves_khalf", 2: "rows64_4waves_khalf_x2"}
fused.set_fp32_split3(True)
res = {}
L = 256
for M in (66001, 80000):
    torch.manual_seed(M)
    net = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
    for p in net.parameters():
        if p.dim() == 1:
            p.data.add_(0.2 * torch.randn_like(p))
    n_tab = M // 17
    table = torch.randn(n_tab, L, device="cuda")
    i0 = torch.randint(0, n_tab, (M,), device="cuda")
    i1 = torch.sort(torch.randint(0, n_tab, (M,), device="cuda")).values
    direct = torch.randn(M, L, device="cuda")
    segs = [(table, i0), (table, i1), (direct, None)]
    with torch.no_grad():
        outs = {}
        for v in names:
            _lib.check(lib.hgnn_set_option(b"mlp_split3_rows128", v))
            outs[v] = fused.fused_concat_mlp(net, segs, direct)
        x = torch.cat([t.double() if i is None else t.double()[i] for t, i in segs], dim=1)
        ref = net.double()(x) + direct.double()
        net.float()
    sc = float(ref.abs().max())
    res[f"M{M}"] = {names[v]: float((outs[v].double() - ref).abs().max()) / sc for v in names}
print(json.dumps(res, indent=1), flush=True)

x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
torch.manual_seed(0)
net = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
nodes = torch.randn(N, L, device="cuda")
edges = torch.randn(M, L, device="cuda")
seg = [(nodes, graph[0]), (nodes, graph[1]), (edges, None)]
tim = {}
for rep in range(2):
    for v in names:
        _lib.check(lib.hgnn_set_option(b"mlp_split3_rows128", v))
        with torch.no_grad():
            for _ in range(3):
                mlp.concat_mlp(net, seg, skip=edges)
            torch.cuda.synchronize()
            ts = []
            for _ in range(8):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                mlp.concat_mlp(net, seg, skip=edges)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
        tim[f"{names[v]}_ms_run{rep}"] = round(sorted(ts)[len(ts) // 2], 4)
_lib.check(lib.hgnn_set_option(b"mlp_split3_rows128", 1))
print(json.dumps(tim))
