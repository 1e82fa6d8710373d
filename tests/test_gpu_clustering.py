"""GPU hierarchy decision: GMM cut + connected components, against CPU restatements
(sklearn GaussianMixture / scipy connected_components -- the libraries the reference itself calls)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _same_partition(a, b):
    """two labelings describe the same partition (ids may differ); -1 must match exactly"""
    a, b = np.asarray(a), np.asarray(b)
    if not np.array_equal(a < 0, b < 0):
        return False
    m = a >= 0
    fwd, bwd = {}, {}
    for x, y in zip(a[m].tolist(), b[m].tolist()):
        if fwd.setdefault(x, y) != y or bwd.setdefault(y, x) != x:
            return False
    return True


def test_connected_components_and_cluster_filter_vs_scipy():
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components as cc
    from hierarchicalgnn_amd.clustering import cluster_labels
    g = torch.Generator().manual_seed(0)
    n, tracks = 3000, 280
    tid = torch.randint(0, tracks, (n,), generator=g)
    order = torch.argsort(tid)
    same = tid[order][1:] == tid[order][:-1]
    src, dst = order[:-1][same], order[1:][same]            # chains inside each track
    drop = torch.rand(src.numel(), generator=g) < 0.1        # break some chains
    src, dst = src[~drop], dst[~drop]
    out = cluster_labels(src.cuda(), dst.cuda(), n, 3).cpu().numpy()
    _, lab = cc(coo_matrix((np.ones(src.numel()), (src.numpy(), dst.numpy())), shape=(n, n)), directed=False)
    present = np.zeros(n, bool)
    present[src.numpy()] = True
    present[dst.numpy()] = True
    counts = np.bincount(lab[present], minlength=lab.max() + 1)
    ref = np.where(present & (counts[lab] >= 3), lab, -1)
    assert _same_partition(out, ref)
    assert out.max() + 1 == len(np.unique(ref[ref >= 0]))    # consecutive ids


def test_gmm_fit_and_cut_vs_sklearn():
    from sklearn.mixture import GaussianMixture
    from hierarchicalgnn_amd.clustering import fit_gmm2_1d, solve_cut
    g = torch.Generator().manual_seed(1)
    v = torch.cat([0.6 * torch.randn(60000, generator=g) - 0.5, 0.9 * torch.randn(40000, generator=g) + 2.5])
    w, mu, var = fit_gmm2_1d(v.cuda())
    ref = GaussianMixture(n_components=2, random_state=0).fit(v.numpy().reshape(-1, 1))
    o_ref = np.argsort(ref.means_.ravel())
    o = torch.argsort(mu).cpu().numpy()
    assert np.allclose(mu.cpu().numpy()[o], ref.means_.ravel()[o_ref], atol=2e-2)
    assert np.allclose(var.cpu().numpy()[o], ref.covariances_.ravel()[o_ref], rtol=5e-2)
    assert np.allclose(w.cpu().numpy()[o], ref.weights_[o_ref], atol=1e-2)
    # the cut: r-times-likelier point of the reference's own function (HGNN_GMM.py:162-170)
    r = 5.0
    cut = solve_cut(w, mu, var, r)
    sig = lambda t: 1 / (1 + np.exp(-t))
    p = ref.predict_proba(np.array([[cut]]))[0]
    f = sig(r) * p[ref.means_.argmin()] - sig(-r) * p[ref.means_.argmax()]
    assert abs(f) < 2e-2
    assert ref.means_.min() < cut < ref.means_.max()


def test_gmm_edge_clustering_end_to_end():
    """embeddings of well-separated tracks: every track comes back as one cluster"""
    from hierarchicalgnn_amd.clustering import GMMEdgeClustering
    g = torch.Generator().manual_seed(2)
    tracks, hits = 150, 8
    centers = torch.nn.functional.normalize(torch.randn(tracks, 8, generator=g))
    tid = torch.arange(tracks).repeat_interleave(hits)
    emb = torch.nn.functional.normalize(centers[tid] + 0.02 * torch.randn(tracks * hits, 8, generator=g))
    n = tracks * hits
    i = torch.arange(n)
    true_e = torch.stack([i[:-1], i[1:]])[:, tid[:-1] == tid[1:]]
    fake_e = torch.stack([torch.randint(0, n, (600,), generator=g), torch.randint(0, n, (600,), generator=g)])
    fake_e = fake_e[:, tid[fake_e[0]] != tid[fake_e[1]]]
    graph = torch.cat([true_e, fake_e], 1)
    graph = torch.cat([graph, graph.flip(0)], 1)[:, torch.randperm(2 * graph.shape[1], generator=g)]
    m = GMMEdgeClustering(dict(min_cluster_size=3, cluster_granularity=0)).cuda().train()
    clusters = m(emb.cuda(), graph.cuda()).cpu()
    assert int(clusters.max()) + 1 == tracks
    assert _same_partition(clusters.numpy(), tid.numpy())
    assert torch.isfinite(m.score_cut).all()


def test_bc_model_forward_end_to_end_on_gpu():
    """BC-HGNN-GMM forward(x, edge_index) with the reference's weights on the track-like golden event:
    every stage on the GPU, output contract of HGNN_GMM.py:323-346"""
    from conftest import load_golden
    from hierarchicalgnn_amd.models import BC_MessagePassing
    z = load_golden("bc_hgnn_L32.npz")
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
    model = BC_MessagePassing(hp)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}, strict=True)
    model = model.cuda().eval()
    x = torch.from_numpy(z["x"]).cuda()
    graph = torch.from_numpy(z["edge_index"]).cuda()
    with torch.no_grad():
        bg, scores, emb = model(x, graph)
    assert bg.shape[0] == 2 and scores.shape == (bg.shape[1],) and emb.shape == (x.shape[0], hp["emb_dim"])
    assert float(scores.min()) > 0 and float(scores.max()) < 1
    assert int(bg[0].max()) < x.shape[0]
    # the event is 70 disconnected chains: the hierarchy finds cluster-like groups, not one blob
    assert int(bg[1].max()) + 1 >= 10
    assert torch.allclose(emb.norm(dim=1), torch.ones_like(emb[:, 0]), atol=1e-5)


def test_connected_components_long_chain_and_adversarial_labels():
    """a single 200k-vertex path with randomly permuted vertex ids (the worst case of label propagation:
    round-1's sweep loop stopped after 64 rounds) plus stars and isolated vertices: the union-find is exact"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components as cc
    from hierarchicalgnn_amd.clustering import connected_components
    g = torch.Generator().manual_seed(3)
    n_chain, n = 200_000, 230_000
    ids = torch.randperm(n_chain, generator=g)
    src, dst = ids[:-1], ids[1:]
    hub = torch.arange(n_chain, n_chain + 20_000)
    star_src = hub
    star_dst = (n_chain + (torch.arange(20_000) // 1000) * 1000)          # 20 stars of 1000
    src = torch.cat([src, star_src])
    dst = torch.cat([dst, star_dst])
    perm = torch.randperm(src.numel(), generator=g)
    src, dst = src[perm], dst[perm]
    labels = connected_components(src.cuda(), dst.cuda(), n).cpu().numpy()
    _, ref = cc(coo_matrix((np.ones(src.numel()), (src.numpy(), dst.numpy())), shape=(n, n)), directed=False)
    assert _same_partition(labels, ref)
    # the label of a component is its smallest vertex id; isolated vertices keep their own
    assert labels[ids.numpy()].min() == labels[ids.numpy()].max() == 0
    assert np.array_equal(labels[n_chain + 20_000:], np.arange(n_chain + 20_000, n))
    again = connected_components(src.cuda(), dst.cuda(), n).cpu().numpy()
    assert np.array_equal(labels, again)                                   # deterministic despite the races


def test_device_cut_equals_the_host_root_solve_and_score_cut_bookkeeping():
    from hierarchicalgnn_amd import _lib
    from hierarchicalgnn_amd.clustering import gmm2_state, solve_cut
    g = torch.Generator().manual_seed(4)
    v = torch.cat([0.5 * torch.randn(50000, generator=g) - 1.0, 0.7 * torch.randn(30000, generator=g) + 2.0]).cuda()
    st = gmm2_state(v)
    host = st.cpu()
    assert host[7] == 1.0 and 2 <= host[8] <= 100                          # converged on the device, in few passes
    for r, training in ((0.0, 1), (5.0, 1), (5.0, 0)):
        sc = torch.tensor([float("inf")], device="cuda")
        _lib.check(_lib.load().hgnn_gmm2_cut_f32(_lib.ptr(st), r, training, 0.95, _lib.ptr(sc),
                                                 _lib.current_stream(sc.device)))
        cut_dev = float(st.cpu()[13])
        cut_ref = solve_cut(st[0:2], st[2:4], st[4:6], r)
        assert abs(cut_dev - cut_ref) < 1e-9
        mid = 0.5 * float(host[2] + host[3])
        expect = 0.95 * mid + 0.05 * cut_ref if training else mid          # inf -> middle, then the EMA
        assert abs(float(sc) - expect) < 1e-5


def test_clustering_makes_exactly_one_host_read():
    """the whole decision (likelihoods, EM to convergence, cut, EMA, components, relabelling) under
    torch.cuda.set_sync_debug_mode("error"): any implicit synchronisation raises; the single deliberate
    read (the cluster count) is whitelisted and counted"""
    from hierarchicalgnn_amd import clustering
    from hierarchicalgnn_amd.clustering import GMMEdgeClustering
    g = torch.Generator().manual_seed(5)
    tracks, hits = 400, 9
    centers = torch.nn.functional.normalize(torch.randn(tracks, 8, generator=g))
    tid = torch.arange(tracks).repeat_interleave(hits)
    n = tracks * hits
    emb = torch.nn.functional.normalize(centers[tid] + 0.02 * torch.randn(n, 8, generator=g)).cuda()
    i = torch.arange(n)
    true_e = torch.stack([i[:-1], i[1:]])[:, tid[:-1] == tid[1:]]
    fake_e = torch.stack([torch.randint(0, n, (2000,), generator=g), torch.randint(0, n, (2000,), generator=g)])
    fake_e = fake_e[:, tid[fake_e[0]] != tid[fake_e[1]]]
    graph = torch.cat([true_e, fake_e], 1)
    graph = torch.cat([graph, graph.flip(0)], 1).cuda().contiguous()
    m = GMMEdgeClustering(dict(min_cluster_size=3, cluster_granularity=0)).cuda().train()
    m(emb, graph)                                                           # warm-up: code objects, index caches
    torch.cuda.synchronize()
    reads0 = clustering.stats["host_reads"]
    torch.cuda.set_sync_debug_mode("error")
    try:
        clusters = m(emb, graph)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert clustering.stats["host_reads"] - reads0 == 1
    assert _same_partition(clusters.cpu().numpy(), tid.numpy())
