"""Drop-in ``DynamicGraphConstruction`` (reference Modules/gnn_utils.py:171-218) and
``find_neighbors`` (Modules/utils.py:228-239): the per-forward kNN graph rebuild and the
attention weights of the hierarchical model.

Same constructor (``weighting_function`` name, ``hparams``), buffers (``knn_radius``) and
sub-module (``weight_normalization`` = ``BatchNorm1d(1)``), hence the same ``state_dict``
keys.  The neighbour search runs in the exact brute-force HIP kernel
(``hgnn_knn_radius_f32``) instead of the un-vendored ``frnn`` grid search; the per-edge
dot products use ``hgnn_edge_dot_f32`` (no gathered copies), differentiable w.r.t. both
embeddings.  ``symmetrize`` restates cugraph's (union of both directions, duplicates
removed); cugraph does not document an edge order, ours is sorted by (src, dst).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .ops import edge_dot, knn_radius


def find_neighbors(embedding1, embedding2, r_max=1.0, k_max=10, return_dist2=False):
    # a tensor radius (the knn_radius buffer) is read by the kernel itself: no .item() host read
    r = r_max if torch.is_tensor(r_max) and r_max.is_cuda else (float(r_max.item()) if torch.is_tensor(r_max)
                                                                 else float(r_max))
    return knn_radius(embedding1, embedding2, k_max, r, return_dist2=return_dist2)


def symmetrize(src: torch.Tensor, dst: torch.Tensor, n: int):
    key = torch.cat([src * n + dst, dst * n + src])
    key = torch.unique(key)            # sorted
    return torch.div(key, n, rounding_mode="floor"), key % n


class DynamicGraphConstruction(nn.Module):
    def __init__(self, weighting_function, hparams):
        super().__init__()
        self.hparams = hparams
        self.weight_normalization = nn.BatchNorm1d(1)
        self.weighting_function = getattr(torch, weighting_function)
        self.register_buffer("knn_radius", torch.ones(1), persistent=True)

    def build_graph(self, src_embeddings, dst_embeddings, sym=False, k=10):
        """gnn_utils.py:193-205 (no gradients): kNN within the tracked radius, optional symmetrisation,
        radius EMA in training mode"""
        with torch.no_grad():
            idxs, d2 = find_neighbors(src_embeddings, dst_embeddings, r_max=self.knn_radius, k_max=k,
                                      return_dist2=True)
            positive = idxs >= 0
            ind = torch.arange(idxs.shape[0], device=idxs.device).unsqueeze(1).expand(idxs.shape)
            if sym:
                s, d = symmetrize(ind[positive], idxs[positive], max(src_embeddings.shape[0], dst_embeddings.shape[0]))
                graph = torch.stack([s, d], dim=0)
            else:
                graph = torch.stack([ind[positive], idxs[positive]], dim=0)
            if self.training and graph.shape[1] > 0:
                maximum_dist = d2.max().clamp(min=0).sqrt()      # padding slots hold -1: max over the valid ones
                self.knn_radius = 0.9 * self.knn_radius + 0.11 * maximum_dist
        return graph

    def edge_weights(self, src_embeddings, dst_embeddings, graph, norm=False, logits=False):
        """gnn_utils.py:208-216: dot product -> BatchNorm1d(1) -> weighting function (-> /mean)"""
        likelihood = edge_dot(src_embeddings, graph[0], dst_embeddings, graph[1])
        edge_weights_logits = self.weight_normalization(likelihood.unsqueeze(1)).squeeze()
        edge_weights = self.weighting_function(edge_weights_logits)
        if norm:
            edge_weights = edge_weights / edge_weights.mean()
        edge_weights = edge_weights.unsqueeze(1)
        if logits:
            return edge_weights, edge_weights_logits
        return edge_weights

    def forward(self, src_embeddings, dst_embeddings, sym=False, norm=False, k=10, logits=False):
        graph = self.build_graph(src_embeddings, dst_embeddings, sym=sym, k=k)
        out = self.edge_weights(src_embeddings, dst_embeddings, graph, norm=norm, logits=logits)
        if logits:
            return graph, out[0], out[1]
        return graph, out
