"""Synthetic TrackML-1GeV-shaped events (no dataset is reachable offline).

Shape contract (SURVEY.md section 8d / BASELINE.md section 4): ``N`` hits on 10
concentric barrel layers with per-hit ``x = (r, phi, z)`` scaled to O(1);
``E`` undirected candidate edges joining hits on adjacent layers inside a
(delta-phi) window, mean directed in-degree 2E/N ~ 16.7 with a tail; the
columns of ``edge_index`` are SHUFFLED (stored TrackML graphs are not sorted by
destination).  The model doubles the graph to ``M = 2E`` directed rows
(reference EdgeClassifier/Models/IN.py:122).  Seeds: 1234 topology, 1235
features, 1236 weights.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

N_LAYERS = 10


def trackml_event(n_hits: int = 120_000, n_edges: int = 1_000_000, seed: int = 1234,
                  hub_fraction: float = 0.01, hub_boost: int = 6) -> Tuple[torch.Tensor, torch.Tensor]:
    """returns (x[N,3] float32, edge_index[2,E] int64) on the CPU"""
    g = torch.Generator().manual_seed(seed)
    layer = torch.randint(0, N_LAYERS, (n_hits,), generator=g)
    phi = torch.rand(n_hits, generator=g) * 2 - 1
    z = torch.randn(n_hits, generator=g).clamp_(-3, 3) / 3
    x = torch.stack([(layer.float() + 1) / N_LAYERS, phi, z], dim=1).contiguous()

    n_src_hits = int((layer < N_LAYERS - 1).sum())
    mean_fanout = n_edges / max(n_src_hits, 1) * 1.15  # oversample, trimmed below
    srcs, dsts = [], []
    for l in range(N_LAYERS - 1):
        a = torch.nonzero(layer == l).squeeze(1)
        b = torch.nonzero(layer == l + 1).squeeze(1)
        if a.numel() == 0 or b.numel() == 0:
            continue
        b = b[torch.argsort(phi[b])]
        pos = torch.searchsorted(phi[b].contiguous(), phi[a].contiguous())
        # per-hit fan-out: Poisson-ish body plus a few "hub" hits (dense jets) for the degree tail
        fan = torch.poisson(torch.full((a.numel(),), mean_fanout), generator=g).long().clamp_(min=1)
        hub = torch.rand(a.numel(), generator=g) < hub_fraction
        fan = torch.where(hub, fan * hub_boost, fan).clamp_(max=b.numel())
        rep = torch.repeat_interleave(torch.arange(a.numel()), fan)
        start = torch.cumsum(fan, 0) - fan
        k = torch.arange(rep.numel()) - start[rep]                 # 0..fan-1 within each hit
        off = k - fan[rep] // 2                                    # window centred on the phi match
        partner = b[(pos[rep] + off) % b.numel()]
        srcs.append(a[rep])
        dsts.append(partner)
    src = torch.cat(srcs)
    dst = torch.cat(dsts)
    perm = torch.randperm(src.numel(), generator=g)
    if src.numel() >= n_edges:
        perm = perm[:n_edges]
    else:  # pad by repeating random edges (keeps shape exact)
        extra = torch.randint(0, src.numel(), (n_edges - src.numel(),), generator=g)
        perm = torch.cat([perm, extra])
    # random orientation so that both rows carry hubs, then the shuffle above is the column order
    flip = torch.rand(perm.numel(), generator=g) < 0.5
    s, d = src[perm], dst[perm]
    edge_index = torch.stack([torch.where(flip, d, s), torch.where(flip, s, d)]).contiguous()
    return x, edge_index


def directed(edge_index: torch.Tensor) -> torch.Tensor:
    """the doubling the model applies (IN.py:122 / HGNN_GMM.py:328)"""
    return torch.cat([edge_index, edge_index.flip(0)], dim=1)


def bipartite_assignment(n_hits: int, n_super: int = 10_000, k: int = 5, seed: int = 1234):
    """HGNN extras: B = N*k bipartite edges hit -> supernode with cluster-size skew,
    weights exp(N(0,1))/mean; returns (bipartite_graph[2,B] int64, weights[B,1] float32)"""
    g = torch.Generator().manual_seed(seed + 7)
    # heavy-tailed cluster popularity (Zipf-like) => supernode fan-in skew
    pop = 1.0 / torch.arange(1, n_super + 1, dtype=torch.float64) ** 0.7
    pop = pop[torch.randperm(n_super, generator=g)]
    sn = torch.multinomial(pop, n_hits * k, replacement=True, generator=g)
    hit = torch.arange(n_hits).repeat_interleave(k)
    w = torch.exp(torch.randn(n_hits * k, 1, generator=g))
    w = (w / w.mean()).float()
    shuffle = torch.randperm(n_hits * k, generator=g)
    return torch.stack([hit[shuffle], sn[shuffle]]).contiguous(), w[shuffle].contiguous()


def super_graph(n_super: int = 10_000, k: int = 10, seed: int = 1234):
    g = torch.Generator().manual_seed(seed + 11)
    s0 = torch.arange(n_super).repeat_interleave(k)
    s1 = torch.randint(0, n_super, (n_super * k,), generator=g)
    sg = torch.unique(torch.stack([torch.cat([s0, s1]), torch.cat([s1, s0])]), dim=1)
    sg = sg[:, torch.randperm(sg.shape[1], generator=g)].contiguous()
    w = torch.sigmoid(torch.randn(sg.shape[1], 1, generator=g))
    return sg, (w / w.mean()).float().contiguous()


def degree_stats(index: torch.Tensor, n: int):
    deg = torch.bincount(index, minlength=n)
    return dict(mean=float(deg.float().mean()), max=int(deg.max()), median=float(deg.float().median()),
                zero=int((deg == 0).sum()))
