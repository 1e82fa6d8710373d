"""GPU-side restatement of the hierarchy decision of ``HierarchicalGNNBlock`` (reference
BipartiteClassification/Models/HGNN_GMM.py:162-234):

    likelihood = atanh(<emb[g0], emb[g1]>)          (:188-189)
    2-component 1-D Gaussian mixture on it          (:192, sklearn GaussianMixture(2).fit on the HOST)
    cut where the right component is r times likelier than the left   (:162-170, scipy fsolve on the HOST)
    EMA of the cut in `score_cut`                   (:195-208)
    connected components of the edges above the cut (:212-221, cugraph)
    clusters of >= min_cluster_size hits, relabelled consecutively    (:172-181)

The reference moves 2M likelihoods to the CPU and runs sklearn / scipy there; here everything stays
on the GPU (EM and label propagation as elementwise / scatter-min tensor programs, the per-edge dot
product in ``hgnn_edge_dot_f32``) except ONE read of six mixture parameters for the scalar root
solve.  sklearn's fit starts from a random k-means initialisation (``random_state=None``), so the
reference's own result varies from run to run: parity here is "same mixture up to EM tolerance from
a deterministic start" and "same partition for the same cut" (tests), not bit-equality.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .ops import edge_dot


def fit_gmm2_1d(v: torch.Tensor, max_iter: int = 100, tol: float = 1e-3, reg_covar: float = 1e-6):
    """EM for a 2-component 1-D Gaussian mixture (sklearn GaussianMixture(2) semantics: tol on the
    per-sample lower bound, reg_covar added to the variances).  Deterministic start: 2-means from the
    data extremes.  Returns (weights[2], means[2], vars[2]) as device tensors."""
    v = v.detach().float().reshape(-1)
    c = torch.stack([v.min(), v.max()])
    for _ in range(8):                                   # Lloyd iterations on the line
        hard = (v - c[0]).abs() > (v - c[1]).abs()
        n1 = hard.sum().clamp(min=1)
        n0 = (~hard).sum().clamp(min=1)
        c = torch.stack([(v * (~hard)).sum() / n0, (v * hard).sum() / n1])
    resp1 = ((v - c[0]).abs() > (v - c[1]).abs()).float()
    resp = torch.stack([1 - resp1, resp1])               # [2, M]
    prev = None
    w = mu = var = None
    for _ in range(max_iter):
        nk = resp.sum(dim=1) + 10 * torch.finfo(torch.float32).eps
        w = nk / v.numel()
        mu = (resp * v).sum(dim=1) / nk
        var = (resp * (v - mu[:, None]) ** 2).sum(dim=1) / nk + reg_covar
        logp = -0.5 * ((v - mu[:, None]) ** 2 / var[:, None] + torch.log(2 * math.pi * var)[:, None]) \
            + torch.log(w)[:, None]
        norm = torch.logsumexp(logp, dim=0)
        resp = torch.exp(logp - norm)
        lower = float(norm.mean())                       # one scalar read per iteration
        if prev is not None and abs(lower - prev) < tol:
            break
        prev = lower
    return w, mu, var


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


def connected_components(src: torch.Tensor, dst: torch.Tensor, n: int) -> torch.Tensor:
    """weakly connected components by min-label propagation with pointer jumping.  Returns labels[n]
    (the smallest vertex id of the component); vertices without an edge keep their own id."""
    labels = torch.arange(n, device=src.device)
    if src.numel() == 0:
        return labels
    for _ in range(64):
        m = torch.minimum(labels[src], labels[dst])
        new = labels.clone()
        new.scatter_reduce_(0, src, m, reduce="amin")
        new.scatter_reduce_(0, dst, m, reduce="amin")
        new = new[new]                                   # pointer jumping
        new = new[new]
        if torch.equal(new, labels):
            break
        labels = new
    return labels


def cluster_labels(src, dst, n: int, min_cluster_size: int) -> torch.Tensor:
    """HGNN_GMM.py:172-181 on top of the components: hits that appear in no kept edge, or whose
    component has fewer than `min_cluster_size` hits, get -1; the rest are numbered 0..C-1."""
    clusters = torch.full((n,), -1, dtype=torch.long, device=src.device)
    if src.numel() == 0:
        return clusters
    labels = connected_components(src, dst, n)
    present = torch.zeros(n, dtype=torch.bool, device=src.device)
    present[src] = True
    present[dst] = True
    counts = torch.bincount(labels[present], minlength=n)
    keep = present & (counts[labels] >= min_cluster_size)
    if bool(keep.any()):
        clusters[keep] = torch.unique(labels[keep], return_inverse=True)[1]
    return clusters


@torch.no_grad()
def gmm_edge_clustering(embeddings, graph, score_cut: torch.Tensor, hparams, training: bool) -> torch.Tensor:
    """``HierarchicalGNNBlock.clustering`` (HGNN_GMM.py:184-234): cluster id of every hit (-1 = none).
    ``score_cut`` is the block's persistent buffer (EMA of the cut, momentum 0.95), updated in place."""
    n = embeddings.shape[0]
    likelihood = edge_dot(embeddings.detach(), graph[0], embeddings.detach(), graph[1])
    likelihood = torch.atanh(torch.clamp(likelihood, min=-1 + 1e-7, max=1 - 1e-7))
    w, mu, var = fit_gmm2_1d(likelihood)
    mu_host = mu.cpu()
    lo, hi = float(mu_host.min()), float(mu_host.max())
    if math.isinf(float(score_cut)):
        score_cut.fill_(0.5 * (lo + hi))
    r = float(hparams.get("cluster_granularity", 0))
    cut = solve_cut(w, mu, var, r, float(score_cut))
    if training and lo < cut < hi:
        score_cut.mul_(0.95).add_(0.05 * cut)
    mask = likelihood >= score_cut.to(likelihood.device)
    clusters = cluster_labels(graph[0][mask], graph[1][mask], n, int(hparams["min_cluster_size"]))
    if int(clusters.max()) <= 2:                         # HGNN_GMM.py:222-232: fall back to the uncut graph
        clusters = cluster_labels(graph[0], graph[1], n, int(hparams["min_cluster_size"]))
    return clusters


class GMMEdgeClustering(nn.Module):
    """stand-alone module form with its own ``score_cut`` buffer"""

    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        self.register_buffer("score_cut", torch.tensor([float("inf")]))

    def forward(self, embeddings: torch.Tensor, graph: torch.Tensor) -> torch.Tensor:
        return gmm_edge_clustering(embeddings, graph, self.score_cut, self.hparams, self.training)
