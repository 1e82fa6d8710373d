#!/usr/bin/env python3
"""BASELINE config 3 shape: the tensor arithmetic of BC-HGNN-GMM (latent=256, 6 + 6 cells) on the
synthetic event with a synthetic hierarchy (S=10k clusters from phi-wedges; kNN graphs rebuilt by
the HIP kNN kernel).  Inference forward, fused vs library.  Usage: bench_bc_forward.py [L] [bf16]
(`bench_bc_forward.py 512 bf16` = BASELINE config 4: latent 512, bf16 latent rows, kNN graphs rebuilt per forward)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import fused, synth
from hierarchicalgnn_amd.models import BC_MessagePassing

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BF16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
torch.manual_seed(1236)
hp = dict(spatial_channels=3, latent=L, hidden=2 * L, emb_dim=8, n_interaction_graph_iters=6,
          n_hierarchical_graph_iters=6, nb_node_layer=3, nb_edge_layer=2, output_layers=3,
          hidden_output_activation="Tanh", hidden_activation="GELU", layernorm=True, share_weight=False,
          bipartitegraph_sparsity=5, supergraph_sparsity=10, min_cluster_size=3, cluster_granularity=5,
          feature_dtype="bf16" if BF16 else "fp32")
model = BC_MessagePassing(hp).cuda().eval()
model.hgnn_block.super_graph_construction.knn_radius.fill_(2.0)
model.hgnn_block.bipartite_graph_construction.knn_radius.fill_(2.0)
# HGNN_EVENT=full_pileup: the BASELINE config 5 event (480k hits / 4M edges -> 8M directed rows) on ONE GPU
if os.environ.get("HGNN_EVENT") == "full_pileup":
    x, ei = synth.trackml_event(480_000, 4_000_000)
else:
    x, ei = synth.trackml_event()
x, ei = x.cuda(), ei.cuda()
S = 10_000
# stand-in for the (host-side, out-of-scope) GMM + connected-components clustering: phi-z cells
clusters = ((x[:, 1] + 1) * 0.5 * 100).long().clamp(0, 99) * 100 + ((x[:, 2] + 1) * 0.5 * 100).long().clamp(0, 99)
_, clusters = torch.unique(clusters, return_inverse=True)
res = {"model": "BC-HGNN-GMM message passing", "latent": L, "feature_dtype": hp["feature_dtype"], "N": x.shape[0], "E": ei.shape[1],
       "clusters": int(clusters.max()) + 1, "params": sum(p.numel() for p in model.parameters())}


def forward():
    directed, emb, nodes, edges, order = model.embed(x, ei)
    means, bg, bw, sg, sw, _ = model.hgnn_block.hierarchy_from_clusters(emb, clusters)
    nodes, sn, _, _ = model.hgnn_block(nodes, edges, directed, means, bg, bw, sg, sw)
    return model.score(nodes, sn, bg), bg.shape[1], sg.shape[1]


with torch.no_grad():
    for name, on in (("fused", True), ("library", False))[:1 if os.environ.get("BC_FUSED_ONLY") else 2]:
        fused.set_enabled(on)
        for _ in range(2):
            out = forward()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = forward()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        res[f"forward_{name}_ms"] = ts[1]
    fused.set_enabled(True)
res["bipartite_edges"], res["super_edges"] = int(out[1]), int(out[2])
print(json.dumps(res, indent=1))
