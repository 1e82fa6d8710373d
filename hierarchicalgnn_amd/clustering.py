"""GPU-side restatement of the hierarchy decision of ``HierarchicalGNNBlock`` (reference
BipartiteClassification/Models/HGNN_GMM.py:162-234):

    likelihood = atanh(<emb[g0], emb[g1]>)          (:188-189)
    2-component 1-D Gaussian mixture on it          (:192, sklearn GaussianMixture(2).fit on the HOST)
    cut where the right component is r times likelier than the left   (:162-170, scipy fsolve on the HOST)
    EMA of the cut in `score_cut`                   (:195-208)
    connected components of the edges above the cut (:212-221, cugraph)
    clusters of >= min_cluster_size hits, relabelled consecutively    (:172-181)

The reference moves 2M likelihoods to the CPU and runs sklearn / scipy there.  Here every step is device
code (csrc/cluster.hip through the C ABI: EM with the convergence test on the device, the root solve and
the score_cut EMA in a one-thread kernel, lock-free union-find components; the per-edge dot product in
``hgnn_edge_dot_f32``) and the whole decision performs exactly ONE host read: the number of clusters,
which fixes the shapes of everything built afterwards (``stats["host_reads"]`` counts them; a second one
happens only on the reference's own fallback path, when the cut leaves at most 3 clusters).
sklearn's fit starts from a random k-means initialisation (``random_state=None``), so the reference's own
result varies from run to run: parity here is "same mixture up to EM tolerance from a deterministic start"
and "same partition for the same cut" (tests), not bit-equality.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from .ops import edge_dot

stats = {"host_reads": 0}


def _host_int(t: torch.Tensor) -> int:
    """the one deliberate device->host read of the decision (whitelisted under torch.cuda sync debug mode)"""
    stats["host_reads"] += 1
    mode = torch.cuda.get_sync_debug_mode()
    torch.cuda.set_sync_debug_mode(0)
    try:
        return int(t.item())
    finally:
        torch.cuda.set_sync_debug_mode(mode)


def gmm2_state(v: torch.Tensor, max_iter: int = 100, tol: float = 1e-3, reg_covar: float = 1e-6) -> torch.Tensor:
    """``hgnn_gmm2_fit_f32``: the float64[16] device state {w0, w1, mu0, mu1, var0, var1, previous lower
    bound, converged, EM passes, min, max, c0, c1, cut, lower bound, -} of the 2-component mixture of v"""
    if not v.is_cuda:
        raise RuntimeError("gmm2_state needs a HIP device tensor: hierarchicalgnn_amd has no CPU path")
    v = v.detach().float().reshape(-1).contiguous()
    if v.numel() == 0:
        raise RuntimeError("gmm2_state: empty input")
    dev = v.device
    state = torch.empty(_lib.GMM_STATE, dtype=torch.float64, device=dev)
    partials = torch.empty(_lib.GMM_BLOCKS * 8, dtype=torch.float64, device=dev)
    ticket = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().hgnn_gmm2_fit_f32(_lib.ptr(v), int(v.numel()), int(max_iter), float(tol),
                                                 float(reg_covar), _lib.ptr(state), _lib.ptr(partials),
                                                 _lib.ptr(ticket), _lib.current_stream(dev)), "hgnn_gmm2_fit_f32")
    return state


def fit_gmm2_1d(v: torch.Tensor, max_iter: int = 100, tol: float = 1e-3, reg_covar: float = 1e-6):
    """EM for a 2-component 1-D Gaussian mixture (sklearn GaussianMixture(2) semantics: tol on the
    per-sample lower bound, reg_covar added to the variances).  Deterministic start: 2-means from the
    data extremes.  Returns (weights[2], means[2], vars[2]) as device tensors; no host synchronisation."""
    st = gmm2_state(v, max_iter, tol, reg_covar)
    return st[0:2], st[2:4], st[4:6]


def solve_cut(w, mu, var, granularity: float, x0: Optional[float] = None) -> float:
    """root of sigmoid(r) * P(left | x) - sigmoid(-r) * P(right | x)  (HGNN_GMM.py:162-170), by
    bisection between the two means (the reference uses fsolve from `x0`; inside the bracket the
    function is monotone, so both find the same root when it exists)."""
    w, mu, var = [t.double().cpu() for t in (w, mu, var)]
    left, right = (0, 1) if mu[0] <= mu[1] else (1, 0)
    sr, sl = 1 / (1 + math.exp(-granularity)), 1 / (1 + math.exp(granularity))

    def post(x):
        lp = [math.log(float(w[k])) - 0.5 * ((x - float(mu[k])) ** 2 / float(var[k]) + math.log(2 * math.pi * float(var[k])))
              for k in (0, 1)]
        m = max(lp)
        p = [math.exp(t - m) for t in lp]
        s = p[0] + p[1]
        return p[0] / s, p[1] / s

    def f(x):
        p = post(x)
        return sr * p[left] - sl * p[right]

    lo, hi = float(mu[left]), float(mu[right])
    if f(lo) * f(hi) > 0:                                # no sign change: the reference's fsolve would not converge either
        return 0.5 * (lo + hi)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if f(lo) * f(mid) <= 0:
            hi = mid
        else:
            lo = mid
    return 0.5 * (lo + hi)


def _cc(src: torch.Tensor, dst: torch.Tensor, n: int, score: Optional[torch.Tensor] = None,
        cut: Optional[torch.Tensor] = None):
    """``hgnn_cc_labels``: (labels int32[n], present int32[n]) of the edges with score >= cut"""
    if not src.is_cuda:
        raise RuntimeError("connected_components needs HIP device tensors: hierarchicalgnn_amd has no CPU path")
    dev = src.device
    labels = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    present = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    s64, d64 = src.contiguous(), dst.contiguous()
    if s64.dtype != torch.int64 or d64.dtype != torch.int64:
        raise RuntimeError("connected_components: int64 vertex ids expected (PyG edge_index rows)")
    sc = score.detach().float().contiguous() if score is not None else None
    ct = cut.detach().float().contiguous() if cut is not None else None
    with torch.cuda.device(dev):
        _lib.check(_lib.load().hgnn_cc_labels(_lib.ptr(s64), _lib.ptr(d64), int(s64.numel()), int(n),
                                              _lib.ptr(sc), _lib.ptr(ct) if sc is not None else None,
                                              _lib.ptr(labels), _lib.ptr(present), _lib.current_stream(dev)),
                   "hgnn_cc_labels")
    return labels[:n], present[:n]


def connected_components(src: torch.Tensor, dst: torch.Tensor, n: int) -> torch.Tensor:
    """weakly connected components (lock-free union-find, csrc/cluster.hip).  Returns labels[n]: the
    smallest vertex id of the component; vertices without an edge keep their own id.  Exact for any graph
    (no sweep cap), deterministic, no host synchronisation."""
    return _cc(src, dst, n)[0].long()


def _cluster_labels(src, dst, n: int, min_cluster_size: int, score=None, cut=None):
    """HGNN_GMM.py:172-181 on top of the components, all on the device: returns (clusters[n] with -1 for
    hits that appear in no kept edge or whose component has fewer than `min_cluster_size` hits, and the
    rest numbered 0..C-1 in increasing order of the component's smallest hit id; C as a 0-d device tensor)"""
    labels, present = _cc(src, dst, n, score, cut)
    lab = labels.long()
    counts = torch.zeros(n, dtype=torch.int32, device=src.device).index_add_(0, lab, present)
    keep_root = counts >= int(min_cluster_size)              # true only at roots (counts live at roots)
    new_id = torch.cumsum(keep_root.to(torch.int64), dim=0) - 1
    keep = (present > 0) & keep_root[lab]
    clusters = torch.where(keep, new_id[lab], torch.full_like(lab, -1))
    return clusters, keep_root.sum()


def cluster_labels(src, dst, n: int, min_cluster_size: int) -> torch.Tensor:
    """cluster id per hit (-1 = none) of the graph (src, dst); see ``_cluster_labels``"""
    if src.numel() == 0:
        return torch.full((n,), -1, dtype=torch.long, device=src.device)
    return _cluster_labels(src, dst, n, min_cluster_size)[0]


@torch.no_grad()
def gmm_edge_clustering(embeddings, graph, score_cut: torch.Tensor, hparams, training: bool,
                        return_count: bool = False):
    """``HierarchicalGNNBlock.clustering`` (HGNN_GMM.py:184-234): cluster id of every hit (-1 = none).
    ``score_cut`` is the block's persistent device buffer (EMA of the cut, momentum 0.95), updated in place
    by the device.  ONE host read (the cluster count; also returned with ``return_count``)."""
    n = embeddings.shape[0]
    dev = embeddings.device
    if score_cut.device != dev or score_cut.dtype != torch.float32 or score_cut.numel() != 1:
        raise RuntimeError("gmm_edge_clustering: score_cut must be a float32[1] buffer on the embeddings' device")
    min_size = int(hparams["min_cluster_size"])
    if graph.shape[1] == 0:
        clusters = torch.full((n,), -1, dtype=torch.long, device=dev)
        return (clusters, 0) if return_count else clusters
    likelihood = edge_dot(embeddings.detach(), graph[0], embeddings.detach(), graph[1])
    likelihood = torch.atanh(torch.clamp(likelihood, min=-1 + 1e-7, max=1 - 1e-7))
    state = gmm2_state(likelihood)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().hgnn_gmm2_cut_f32(_lib.ptr(state), float(hparams.get("cluster_granularity", 0)),
                                                 1 if training else 0, 0.95, _lib.ptr(score_cut),
                                                 _lib.current_stream(dev)), "hgnn_gmm2_cut_f32")
    clusters, count = _cluster_labels(graph[0], graph[1], n, min_size, likelihood, score_cut)
    c = _host_int(count)
    if c <= 3:                                           # HGNN_GMM.py:222-232 (clusters.max() <= 2): uncut graph
        clusters, count = _cluster_labels(graph[0], graph[1], n, min_size)
        c = _host_int(count)
    return (clusters, c) if return_count else clusters


class GMMEdgeClustering(nn.Module):
    """stand-alone module form with its own ``score_cut`` buffer"""

    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        self.register_buffer("score_cut", torch.tensor([float("inf")]))

    def forward(self, embeddings: torch.Tensor, graph: torch.Tensor) -> torch.Tensor:
        return gmm_edge_clustering(embeddings, graph, self.score_cut, self.hparams, self.training)
