#!/usr/bin/env python3
"""Ablations of the feature-split bf16 MLP kernel (DIAGNOSTIC: the ablated runs compute wrong results).
Usage: tune_mlp_split.py [L]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import _lib, make_mlp, mlp, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
net = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
nodes = torch.randn(N, L, device="cuda").bfloat16()
edges = torch.randn(M, L, device="cuda").bfloat16()
seg = [(nodes, graph[0]), (nodes, graph[1]), (edges, None)]
seg_nogather = [(edges, None), (edges, None), (edges, None)]
flop = 2 * (3 * L * 2 * L + 2 * L * L) * M
lib = _lib.load()


def timeit(fn, reps=5):
    fn()
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


res = {"L": L, "M": M}
try:
    with torch.no_grad():
        for var in (0, 2, 0, 2):
            lib.hgnn_set_option(b"mlp_split_variant", var)
            res.setdefault(f"variant{var}_ms", []).append(timeit(lambda: mlp.concat_mlp(net, seg, skip=edges)))
        lib.hgnn_set_option(b"mlp_split_variant", int(os.environ.get("SPLIT_VARIANT", "-1")))
        if os.environ.get("SPLIT_AB_ONLY"):
            print(json.dumps(res, indent=1)); raise SystemExit(0)
        for bits, name in ((0, "full"), (1, "weights_L1_resident"), (2, "no_layernorm_act"), (3, "no_epilogue_weights_L1"),
                           (4, "one_input_panel"), (7, "no_epilogue_weights_L1_one_input_panel"),
                           (8, "no_panel_barriers"), (31, "mfma_and_chunk0_operands_only")):
            lib.hgnn_set_option(b"mlp_ablate", bits)
            t = timeit(lambda: mlp.concat_mlp(net, seg, skip=edges))
            res[name + "_ms"] = t
        lib.hgnn_set_option(b"mlp_ablate", 0)
        res["no_gather_ms"] = timeit(lambda: mlp.concat_mlp(net, seg_nogather, skip=edges))
finally:
    lib.hgnn_set_option(b"mlp_ablate", 0)
    lib.hgnn_set_option(b"mlp_split_variant", -1)
res["full_tflops"] = flop / res["full_ms"] / 1e9
print(json.dumps(res, indent=1))
