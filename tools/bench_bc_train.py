#!/usr/bin/env python3
"""One TRAINING step of the BC-HGNN-GMM message passing (BASELINE configs 3 / 4: HGNN_GMM.yaml, 6 + 6 cells) on the
synthetic event: IGNN block -> hierarchy (synthetic phi-z clusters, kNN graphs + differentiable attention weights
rebuilt by the HIP kNN kernel) -> K5 pooling -> HGNN cells -> bipartite head, BCE-style loss, backward through
every HIP op (reference-style reentrant checkpointing).  Usage: bench_bc_train.py [L] [bf16] [nockpt]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import fused, synth
from hierarchicalgnn_amd.models import BC_MessagePassing

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BF16 = "bf16" in sys.argv[2:]
ckpt = "nockpt" not in sys.argv[2:]
torch.manual_seed(1236)
hp = dict(spatial_channels=3, latent=L, hidden="ratio", hidden_ratio=2, emb_dim=8, n_interaction_graph_iters=6,
          n_hierarchical_graph_iters=6, nb_node_layer=3, nb_edge_layer=2, output_layers=3,
          hidden_output_activation="Tanh", hidden_activation="GELU", layernorm=True, share_weight=False,
          bipartitegraph_sparsity=5, supergraph_sparsity=10, min_cluster_size=3, cluster_granularity=5,
          checkpointing=ckpt)
if BF16:
    hp["feature_dtype"] = "bf16"
model = BC_MessagePassing(hp).cuda().train()
model.hgnn_block.super_graph_construction.knn_radius.fill_(2.0)
model.hgnn_block.bipartite_graph_construction.knn_radius.fill_(2.0)
x, ei = synth.trackml_event()
x, ei = x.cuda(), ei.cuda()
clusters = ((x[:, 1] + 1) * 0.5 * 100).long().clamp(0, 99) * 100 + ((x[:, 2] + 1) * 0.5 * 100).long().clamp(0, 99)
_, clusters = torch.unique(clusters, return_inverse=True)
n_clusters = int(clusters.max()) + 1
res = {"model": "BC-HGNN-GMM message passing, training step", "latent": L, "feature_dtype": "bf16" if BF16 else "fp32",
       "checkpointing": ckpt, "N": x.shape[0], "E": ei.shape[1], "clusters": n_clusters,
       "params": sum(p.numel() for p in model.parameters())}


def step():
    directed, emb, nodes, edges, _ = model.embed(x.clone(), ei)
    means, bg, bw, sg, sw, _ = model.hgnn_block.hierarchy_from_clusters(emb, clusters, n_clusters)
    n_out, sn_out, _, _ = model.hgnn_block(nodes, edges, directed, means, bg, bw, sg, sw)
    scores = model.score(n_out, sn_out, bg)
    target = (torch.arange(scores.shape[0], device=scores.device) % 3 == 0).float()
    loss = torch.nn.functional.binary_cross_entropy(scores, target)
    loss.backward()
    model.zero_grad(set_to_none=True)
    return float(loss.detach())


for name, on in (("hip", True), ("library", False))[:1 if os.environ.get("TRAIN_HIP_ONLY") else 2]:
    fused.set_enabled(on)
    step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        loss = step()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    res[f"train_step_{name}_ms"] = ts[1]
    res[f"loss_{name}"] = loss
    res[f"peak_mem_GB_{name}"] = torch.cuda.max_memory_allocated() / 2**30
fused.set_enabled(True)
print(json.dumps(res, indent=1))
