#!/usr/bin/env python3
"""Timing of the hierarchy decision (GMM cut + connected components, HGNN_GMM.py:184-234) on the BASELINE event
shape: N = 120k hits, M = 2M directed edges, emb_dim = 8.  Prints one JSON object."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from hierarchicalgnn_amd import clustering, synth
    from hierarchicalgnn_amd.clustering import GMMEdgeClustering, _cluster_labels, gmm2_state
    dev = torch.device("cuda")
    n, e = 120_000, 1_000_000
    g = torch.Generator().manual_seed(1)
    # track-like event: 12k tracks of 10 hits; a third of the candidate edges join hits of one track
    # (likely edges), the rest join random hits (unlikely): the mixture the reference's GMM cut separates
    n_tracks, per = 12_000, 10
    tid = torch.arange(n_tracks).repeat_interleave(per)
    centers = torch.nn.functional.normalize(torch.randn(n_tracks, 8, generator=g))
    emb = torch.nn.functional.normalize(centers[tid] + 0.05 * torch.randn(n, 8, generator=g)).to(dev)
    i = torch.arange(n)
    true_e = torch.cat([torch.stack([i[:-k], i[k:]])[:, tid[:-k] == tid[k:]] for k in (1, 2, 3)], dim=1)
    n_fake = e - true_e.shape[1]
    fake_e = torch.stack([torch.randint(0, n, (n_fake,), generator=g), torch.randint(0, n, (n_fake,), generator=g)])
    ei = torch.cat([true_e, fake_e], dim=1)[:, torch.randperm(e, generator=g)]
    graph = synth.directed(ei).to(dev)
    m = GMMEdgeClustering(dict(min_cluster_size=3, cluster_granularity=5)).to(dev).train()

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2]

    r0 = clustering.stats["host_reads"]
    total = timed(lambda: m(emb, graph))
    reads = (clustering.stats["host_reads"] - r0) / 11
    lik = torch.atanh(torch.clamp(clustering.edge_dot(emb, graph[0], emb, graph[1]), -1 + 1e-7, 1 - 1e-7))
    t_gmm = timed(lambda: gmm2_state(lik))
    st = gmm2_state(lik).cpu()
    t_cc = timed(lambda: _cluster_labels(graph[0], graph[1], n, 3, lik, m.score_cut))
    t_cc_all = timed(lambda: _cluster_labels(graph[0], graph[1], n, 3))
    clusters = m(emb, graph)
    # the reference's way: likelihoods to the host + sklearn + scipy (its cugraph part replaced by scipy)
    cpu_ms = None
    try:
        from sklearn.mixture import GaussianMixture
        t0 = time.perf_counter()
        GaussianMixture(2).fit(lik.cpu().numpy().reshape(-1, 1))
        cpu_ms = (time.perf_counter() - t0) * 1e3
    except Exception:
        pass
    print(json.dumps({
        "shape": {"hits": n, "directed_edges": int(graph.shape[1]), "emb_dim": 8},
        "clustering_total_ms": total, "host_reads_per_call": reads,
        "gmm_fit_ms": t_gmm, "em_passes": float(st[8]), "converged": float(st[7]),
        "components_and_relabel_ms_cut_graph": t_cc, "components_and_relabel_ms_all_edges": t_cc_all,
        "clusters": int(clusters.max()) + 1,
        "reference_style_sklearn_gmm_fit_on_host_ms": cpu_ms,
    }))


if __name__ == "__main__":
    main()
