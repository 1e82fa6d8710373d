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


def batch_norm_1(bn: nn.BatchNorm1d, x: torch.Tensor, stat_reduce=None) -> torch.Tensor:
    """``bn(x.unsqueeze(1)).squeeze()`` for the one-channel BatchNorm of gnn_utils.py:179,209.  With ``stat_reduce``
    (a differentiable sum over the shards of ONE event, partition.allreduce_supernode_sums) and the module in
    training mode the batch statistics are those of the WHOLE event: every rank contributes (sum, sum of squares,
    count) of its own edges -- a synchronised BatchNorm; running statistics are updated exactly as nn.BatchNorm1d
    does (momentum, unbiased variance).  Gradients flow through the reduced statistics: the backward of the sum over
    ranks is again a sum over ranks, which is what makes d loss / d x_i see the other shards' loss terms."""
    if stat_reduce is None or not bn.training:
        return bn(x.unsqueeze(1)).squeeze(1)
    xd = x.double()
    stats = stat_reduce(torch.stack([xd.sum(), (xd * xd).sum(), xd.new_tensor(float(x.numel()))]))
    n = stats[2]
    mean = stats[0] / n
    var = (stats[1] / n - mean * mean).clamp(min=0)                 # biased, as the normalisation uses
    with torch.no_grad():
        if bn.track_running_stats and bn.running_mean is not None:
            bn.num_batches_tracked += 1
            m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - m).add_(m * mean.to(bn.running_mean.dtype))
            bn.running_var.mul_(1 - m).add_(m * (var * n / (n - 1).clamp(min=1)).to(bn.running_var.dtype))
    y = ((xd - mean) * torch.rsqrt(var + bn.eps)).to(x.dtype)
    if bn.affine:
        y = y * bn.weight[0] + bn.bias[0]
    return y


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

    def edge_weights(self, src_embeddings, dst_embeddings, graph, norm=False, logits=False, stat_reduce=None):
        """gnn_utils.py:208-216: dot product -> BatchNorm1d(1) -> weighting function (-> /mean).
        ``stat_reduce``: the edges are one shard of an event (partition.py): synchronised batch statistics"""
        likelihood = edge_dot(src_embeddings, graph[0], dst_embeddings, graph[1])
        edge_weights_logits = batch_norm_1(self.weight_normalization, likelihood, stat_reduce)
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
