#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Run in the build container only (it needs /root/reference, which never travels
to the GPU box):

    python tests/golden/make_golden.py

What it does
------------
The reference (clairesonglee/HierarchicalGNN @ /root/reference) is pure Python
but depends on packages that are not installed here (torch_scatter,
torch_geometric, pytorch_lightning, cugraph, cudf, cupy, frnn).  We inject
stand-in modules into ``sys.modules`` *before* importing the reference's own
``Modules/gnn_utils.py``, ``Modules/utils.py``, ``EdgeClassifier/Models/IN.py``
and ``BipartiteClassification/Models/HGNN_GMM.py`` and then run THE REFERENCE'S
code (its cells, its make_mlp, its forward()) on seeded CPU fp32 inputs.  The
inputs and outputs are stored as ``.npz`` (plain arrays, no pickle).

Stand-ins, and why they do not weaken the pin:
  * ``torch_scatter.scatter_add`` -> ``zeros(dim_size,F).scatter_add_(0, idx, src)``
    which is torch-scatter 2.0.9's documented CPU behaviour (broadcast the 1-D
    index, allocate zeros, Tensor.scatter_add_).  torch-scatter's source is not
    in this image, so this one equivalence rests on the call-site semantics.
    Every call is RECORDED, so K2-K5 fixtures come from the real call sites
    (gnn_utils.py:124,125,142,143 and HGNN_GMM.py:269).
  * pytorch_lightning / torch_geometric / wandb: containers only, no arithmetic.
  * frnn / cugraph / cudf / cupy: brute-force kNN and scipy connected
    components.  They only decide WHICH bipartite/super graph is built; the
    graphs and weights they lead to are captured and stored with the fixture,
    so the hot-path arithmetic that is pinned does not depend on them.

Nothing from the reference's text is stored: fixtures are data only.
"""
import os
import sys
import math
import types
import tempfile

sys.dont_write_bytecode = True
# never read or write bytecode caches inside the (read-only) reference tree
sys.pycache_prefix = tempfile.mkdtemp(prefix="golden_pyc_")

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("HGNN_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

torch.set_num_threads(1)  # sequential CPU scatter_add_: arrival-order sums
torch.use_deterministic_algorithms(True)

SCATTER_LOG = []  # every scatter_add call made by reference code


# --------------------------------------------------------------------------
# stand-in modules
# --------------------------------------------------------------------------
def _scatter_add(src, index, dim=0, dim_size=None, out=None):
    assert dim == 0 and index.dim() == 1
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    dim_size = int(dim_size)
    res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    res = res.scatter_add(0, idx, src)
    SCATTER_LOG.append(dict(src=src.detach().clone(), index=index.detach().clone(),
                            dim_size=dim_size, out=res.detach().clone()))
    return res


def _scatter_mean(src, index, dim=0, dim_size=None):
    dim_size = int(dim_size)
    s = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    s = s.scatter_add(0, idx, src)
    c = torch.zeros(dim_size, dtype=src.dtype).scatter_add(0, index, torch.ones_like(index, dtype=src.dtype))
    return s / c.clamp(min=1).view(-1, *([1] * (src.dim() - 1)))


def _install_stubs():
    ts = types.ModuleType("torch_scatter")
    ts.scatter_add = _scatter_add
    ts.scatter_mean = _scatter_mean
    ts.scatter_min = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError)
    ts.scatter_max = ts.scatter_min
    sys.modules["torch_scatter"] = ts

    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, hparams):
            self._hp = dict(hparams)

        @property
        def hparams(self):
            return self._hp

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl

    tg = types.ModuleType("torch_geometric")
    tgd = types.ModuleType("torch_geometric.data")

    class Data(dict):
        pass

    tgd.Data = Data
    tgd.DataLoader = object
    tg.data = tgd
    sys.modules["torch_geometric"] = tg
    sys.modules["torch_geometric.data"] = tgd

    cp = types.ModuleType("cupy")
    cp.asarray = lambda t: np.asarray(t.detach().cpu() if torch.is_tensor(t) else t)
    cp.vstack = np.vstack
    sys.modules["cupy"] = cp

    cudf = types.ModuleType("cudf")

    class Series:
        def __init__(self, t):
            self.v = np.asarray(t.detach().cpu() if torch.is_tensor(t) else t)

        def to_cupy(self):
            return self.v

    cudf.Series = Series
    cudf.DataFrame = dict
    sys.modules["cudf"] = cudf

    cg = types.ModuleType("cugraph")

    class Graph:
        def from_cudf_edgelist(self, df, source, destination):
            self.src = np.asarray(df[source])
            self.dst = np.asarray(df[destination])

    def connected_components(G):
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import connected_components as cc
        if len(G.src) == 0:
            raise ValueError("empty graph")
        verts = np.unique(np.concatenate([G.src, G.dst]))
        remap = {v: i for i, v in enumerate(verts)}
        s = np.array([remap[v] for v in G.src])
        d = np.array([remap[v] for v in G.dst])
        n = len(verts)
        _, labels = cc(coo_matrix((np.ones(len(s)), (s, d)), shape=(n, n)), directed=False)
        return {"labels": labels.astype(np.int64), "vertex": verts.astype(np.int64)}

    cg.Graph = Graph
    comp = types.ModuleType("cugraph.components")
    comp.connected_components = connected_components
    cg.components = comp
    st = types.ModuleType("cugraph.structure")
    sym = types.ModuleType("cugraph.structure.symmetrize")

    def symmetrize(src, dst):
        a = np.stack([np.concatenate([src.v, dst.v]), np.concatenate([dst.v, src.v])])
        a = np.unique(a, axis=1)
        return Series(a[0]), Series(a[1])

    sym.symmetrize = symmetrize
    st.symmetrize = sym
    cg.structure = st
    sys.modules["cugraph"] = cg
    sys.modules["cugraph.components"] = comp
    sys.modules["cugraph.structure"] = st
    sys.modules["cugraph.structure.symmetrize"] = sym

    fr = types.ModuleType("frnn")

    def frnn_grid_points(points1, points2, lengths1=None, lengths2=None, K=10, r=1.0, **kw):
        p1, p2 = points1[0], points2[0]
        d = torch.cdist(p1, p2)
        k = min(K, p2.shape[0])
        dist, idx = d.topk(k, dim=1, largest=False)
        rr = float(r) if not torch.is_tensor(r) else float(r.item())
        idx = torch.where(dist <= rr, idx, torch.full_like(idx, -1))
        if k < K:
            pad = torch.full((idx.shape[0], K - k), -1, dtype=idx.dtype)
            idx = torch.cat([idx, pad], 1)
            dist = torch.cat([dist, torch.zeros(dist.shape[0], K - k)], 1)
        return (dist ** 2).unsqueeze(0), idx.unsqueeze(0), None, None

    fr.frnn_grid_points = frnn_grid_points
    sys.modules["frnn"] = fr


def _import_reference():
    _install_stubs()
    sys.path.insert(0, os.path.join(REF, "Modules"))
    import gnn_utils  # noqa
    import utils  # noqa
    from EdgeClassifier.Models.IN import EC_InteractionGNN
    from BipartiteClassification.Models.HGNN_GMM import BC_HierarchicalGNN_GMM
    return gnn_utils, utils, EC_InteractionGNN, BC_HierarchicalGNN_GMM


def kaiming_init(model, gen):
    """Same scheme as the reference's training_utils.kaiming_init (:48-58):
    biases 0; '*0.weight' ~ N(0, 1/sqrt(fan_in)); other 2-D weights
    ~ N(0, sqrt(2)/sqrt(fan_in)); 1-D weights (LayerNorm/BatchNorm) untouched.
    (training_utils.py itself cannot be imported: it imports model packages that
    need cuml; the scheme is restated here and is only used to make weights.)"""
    for name, p in model.named_parameters():
        if name.endswith(".bias"):
            p.data.fill_(0)
        elif p.dim() < 2:
            continue
        elif name.endswith("0.weight"):
            p.data.copy_(torch.randn(p.shape, generator=gen) / math.sqrt(p.shape[1]))
        else:
            p.data.copy_(torch.randn(p.shape, generator=gen) * math.sqrt(2) / math.sqrt(p.shape[1]))
    # non-trivial LayerNorm affine and biases so parity covers them
    for name, p in model.named_parameters():
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn(p.shape, generator=gen))


def sd_np(module, prefix="sd."):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def grads_np(module, prefix="grad."):
    return {prefix + k: p.grad.detach().numpy().copy() for k, p in module.named_parameters()
            if p.grad is not None}


def synth_event(n_nodes, n_edges, gen):
    """small TrackML-shaped event: hits on layers, edges between adjacent layers."""
    layer = torch.randint(0, 10, (n_nodes,), generator=gen)
    r = (layer.float() + 1) / 10
    phi = (torch.rand(n_nodes, generator=gen) * 2 - 1)
    z = torch.randn(n_nodes, generator=gen).clamp(-3, 3) / 3
    x = torch.stack([r, phi, z], 1).float()
    src = torch.randint(0, n_nodes, (n_edges,), generator=gen)
    dst = torch.randint(0, n_nodes, (n_edges,), generator=gen)
    keep = src != dst
    return x, torch.stack([src[keep], dst[keep]])


def synth_tracks(n_tracks, hits_per_track, gen):
    """disconnected track-like chains so that connected components give many
    clusters whichever way the reference's GMM cut falls."""
    n = n_tracks * hits_per_track
    tid = torch.arange(n_tracks).repeat_interleave(hits_per_track)
    layer = torch.arange(hits_per_track).repeat(n_tracks)
    phi0 = torch.rand(n_tracks, generator=gen) * 2 - 1
    z0 = torch.randn(n_tracks, generator=gen).clamp(-3, 3) / 3
    x = torch.stack([(layer.float() + 1) / 10,
                     phi0[tid] + 0.01 * layer.float() + 0.002 * torch.randn(n, generator=gen),
                     z0[tid] * (layer.float() + 1) / 10], 1).float()
    i = torch.arange(n)
    e1 = torch.stack([i[layer < hits_per_track - 1], i[layer < hits_per_track - 1] + 1])
    e2 = torch.stack([i[layer < hits_per_track - 2], i[layer < hits_per_track - 2] + 2])
    graph = torch.cat([e1, e2], 1)
    graph = graph[:, torch.randperm(graph.shape[1], generator=gen)]
    return x, graph


# --------------------------------------------------------------------------
def gen_k1_cases():
    """K1 scatter_add(src, index, dim=0, dim_size) single-op cases."""
    g = torch.Generator().manual_seed(101)
    cases = {}

    def add(name, src, index, dim_size):
        out = _scatter_add(src, index, 0, dim_size)
        cases[name + ".src"] = src.numpy()
        cases[name + ".index"] = index.numpy()
        cases[name + ".dim_size"] = np.int64(dim_size)
        cases[name + ".out"] = out.numpy()

    add("random_L32", torch.randn(500, 32, generator=g), torch.randint(0, 64, (500,), generator=g), 64)
    add("random_L128", torch.randn(700, 128, generator=g), torch.randint(0, 90, (700,), generator=g), 90)
    add("random_L256", torch.randn(600, 256, generator=g), torch.randint(0, 80, (600,), generator=g), 80)
    add("random_L512", torch.randn(200, 512, generator=g), torch.randint(0, 30, (200,), generator=g), 30)
    # empty destinations + dim_size > max(index)+1
    add("sparse_dst", torch.randn(200, 64, generator=g), torch.randint(0, 10, (200,), generator=g) * 7, 100)
    # duplicate heavy: everything to 3 rows (forces list splitting)
    add("dup_heavy", torch.randn(1500, 128, generator=g), torch.randint(0, 3, (1500,), generator=g), 5)
    # odd feature widths (scalar fallback)
    add("odd_F3", torch.randn(100, 3, generator=g), torch.randint(0, 17, (100,), generator=g), 17)
    add("F8", torch.randn(400, 8, generator=g), torch.randint(0, 50, (400,), generator=g), 50)
    add("F248", torch.randn(150, 248, generator=g), torch.randint(0, 20, (150,), generator=g), 20)
    # empty input
    add("empty", torch.zeros(0, 32), torch.zeros(0, dtype=torch.long), 9)
    # sorted index
    idx = torch.sort(torch.randint(0, 33, (256,), generator=g)).values
    add("sorted", torch.randn(256, 128, generator=g), idx, 33)
    np.savez_compressed(os.path.join(OUT, "k1_scatter_add.npz"), **cases)
    print("k1_scatter_add.npz", len(cases) // 4, "cases")


def gen_ignn_cell(gnn_utils, latent, n_nodes, n_edges, seed):
    g = torch.Generator().manual_seed(seed)
    hp = dict(latent=latent, hidden=2 * latent, nb_edge_layer=2, nb_node_layer=3,
              layernorm=True, hidden_activation="GELU")
    cell = gnn_utils.InteractionGNNCell(hp)
    kaiming_init(cell, g)
    _, graph = synth_event(n_nodes, n_edges, g)
    graph = torch.cat([graph, graph.flip(0)], 1)
    nodes = torch.randn(n_nodes, latent, generator=g, requires_grad=True)
    edges = torch.randn(graph.shape[1], latent, generator=g, requires_grad=True)
    r1 = torch.randn(n_nodes, latent, generator=g)
    r2 = torch.randn(graph.shape[1], latent, generator=g)
    on, oe = cell(nodes, edges, graph)
    ((on * r1).sum() + (oe * r2).sum()).backward()
    d = dict(graph=graph.numpy(), nodes=nodes.detach().numpy(), edges=edges.detach().numpy(),
             r_nodes=r1.numpy(), r_edges=r2.numpy(),
             out_nodes=on.detach().numpy(), out_edges=oe.detach().numpy(),
             grad_nodes=nodes.grad.numpy(), grad_edges=edges.grad.numpy(),
             latent=np.int64(latent))
    d.update(sd_np(cell))
    d.update(grads_np(cell))
    fn = f"ignn_cell_L{latent}.npz"
    np.savez_compressed(os.path.join(OUT, fn), **d)
    print(fn)


def _bip(n_nodes, n_super, kb, ks, g):
    bg0 = torch.arange(n_nodes).repeat_interleave(kb)
    bg1 = torch.randint(0, n_super, (n_nodes * kb,), generator=g)
    keep = torch.rand(n_nodes * kb, generator=g) > 0.15  # ragged: some kNN slots are -1 in the reference
    bg = torch.stack([bg0[keep], bg1[keep]])
    bw = torch.exp(0.5 * torch.randn(bg.shape[1], 1, generator=g))
    bw = bw / bw.mean()
    s0 = torch.arange(n_super).repeat_interleave(ks)
    s1 = torch.randint(0, n_super, (n_super * ks,), generator=g)
    sg = torch.stack([torch.cat([s0, s1]), torch.cat([s1, s0])])
    sg = torch.unique(sg, dim=1)
    sw = torch.sigmoid(torch.randn(sg.shape[1], 1, generator=g))
    sw = sw / sw.mean()
    return bg, bw, sg, sw


def gen_hgnn_cell(gnn_utils, latent, n_nodes, n_edges, n_super, seed):
    g = torch.Generator().manual_seed(seed)
    hp = dict(latent=latent, hidden=2 * latent, nb_edge_layer=2, nb_node_layer=3,
              layernorm=True, hidden_activation="GELU")
    cell = gnn_utils.HierarchicalGNNCell(hp)
    kaiming_init(cell, g)
    _, graph = synth_event(n_nodes, n_edges, g)
    graph = torch.cat([graph, graph.flip(0)], 1)
    bg, bw, sg, sw = _bip(n_nodes, n_super, 5, 4, g)
    bw.requires_grad_(True)
    sw.requires_grad_(True)
    nodes = torch.randn(n_nodes, latent, generator=g, requires_grad=True)
    edges = torch.randn(graph.shape[1], latent, generator=g, requires_grad=True)
    sn = torch.randn(n_super, latent, generator=g, requires_grad=True)
    se = torch.randn(sg.shape[1], latent, generator=g, requires_grad=True)
    rs = [torch.randn(t.shape, generator=g) for t in (nodes, edges, sn, se)]
    outs = cell(nodes, edges, sn, se, graph, bg, bw, sg, sw)
    sum((o * r).sum() for o, r in zip(outs, rs)).backward()
    d = dict(graph=graph.numpy(), bipartite_graph=bg.numpy(), bipartite_edge_weights=bw.detach().numpy(),
             super_graph=sg.numpy(), super_edge_weights=sw.detach().numpy(),
             nodes=nodes.detach().numpy(), edges=edges.detach().numpy(),
             supernodes=sn.detach().numpy(), superedges=se.detach().numpy(),
             latent=np.int64(latent))
    for nm, o, r, i in zip(("nodes", "edges", "supernodes", "superedges"), outs, rs, (nodes, edges, sn, se)):
        d["out_" + nm] = o.detach().numpy()
        d["r_" + nm] = r.numpy()
        d["grad_" + nm] = i.grad.numpy()
    d["grad_bipartite_edge_weights"] = bw.grad.numpy()
    d["grad_super_edge_weights"] = sw.grad.numpy()
    d.update(sd_np(cell))
    d.update(grads_np(cell))
    fn = f"hgnn_cell_L{latent}.npz"
    np.savez_compressed(os.path.join(OUT, fn), **d)
    print(fn)


def gen_ec_in(EC, latent=32):
    """BASELINE config 1: flat EC-IN, latent=32, full forward()."""
    g = torch.Generator().manual_seed(303)
    hp = dict(spatial_channels=3, latent=latent, hidden=2 * latent, n_interaction_graph_iters=14,
              nb_node_layer=3, nb_edge_layer=2, output_layers=3, hidden_output_activation="GELU",
              hidden_activation="GELU", layernorm=True, share_weight=False)
    model = EC(hp)
    kaiming_init(model, g)
    x, graph = synth_event(2000, 12000, g)
    scores = model(x.clone(), graph)
    d = dict(x=x.numpy(), edge_index=graph.numpy(), scores=scores.detach().numpy(),
             n_params=np.int64(sum(p.numel() for p in model.parameters())))
    d.update({"hp." + k: np.array(v) for k, v in hp.items()})
    d.update(sd_np(model))
    np.savez_compressed(os.path.join(OUT, f"ec_in_L{latent}.npz"), **d)
    print(f"ec_in_L{latent}.npz params", int(d["n_params"]))


def gen_bc_hgnn(BC, latent=32):
    """BASELINE config 3 shape at small latent: BC-HGNN-GMM forward with every
    scatter_add call site captured (K1..K5) and the tensors that enter the
    HierarchicalGNNCell loop, so stand-in kNN/CC choices do not matter."""
    g = torch.Generator().manual_seed(404)
    torch.manual_seed(404)
    np.random.seed(404)
    hp = dict(spatial_channels=3, latent=latent, hidden=2 * latent, emb_dim=8,
              n_interaction_graph_iters=2, n_hierarchical_graph_iters=2,
              nb_node_layer=3, nb_edge_layer=2, output_layers=3, hidden_output_activation="Tanh",
              hidden_activation="GELU", layernorm=True, share_weight=False,
              bipartitegraph_sparsity=5, supergraph_sparsity=10, min_cluster_size=3,
              cluster_granularity=5)
    model = BC(hp)
    kaiming_init(model, g)
    model.eval()  # freeze BatchNorm1d(1) stats / knn_radius EMA: forward is then a pure function
    model.hgnn_block.super_graph_construction.knn_radius.fill_(2.0)
    model.hgnn_block.bipartite_graph_construction.knn_radius.fill_(2.0)
    x, graph = synth_tracks(70, 9, g)

    # capture the arguments of every HierarchicalGNNCell call
    cell_calls = []
    for cell in model.hgnn_block.hgnn_cells:
        cell.register_forward_hook(
            lambda m, inp, out: cell_calls.append(([t.detach().clone() for t in inp],
                                                    [t.detach().clone() for t in out])))
    SCATTER_LOG.clear()
    bg, bscores, emb = model(x.clone(), graph)
    d = dict(x=x.numpy(), edge_index=graph.numpy(), bipartite_graph=bg.numpy(),
             bipartite_scores=bscores.detach().numpy(), embeddings=emb.detach().numpy())
    d.update({"hp." + k: np.array(v) for k, v in hp.items()})
    d.update(sd_np(model))
    names = ["nodes", "edges", "supernodes", "superedges", "graph", "bipartite_graph",
             "bipartite_edge_weights", "super_graph", "super_edge_weights"]
    for i, (inp, out) in enumerate(cell_calls):
        for nm, t in zip(names, inp):
            d[f"cell{i}.in.{nm}"] = t.numpy()
        for nm, t in zip(names[:4], out):
            d[f"cell{i}.out.{nm}"] = t.numpy()
    for i, c in enumerate(SCATTER_LOG):
        d[f"scatter{i}.src"] = c["src"].numpy()
        d[f"scatter{i}.index"] = c["index"].numpy()
        d[f"scatter{i}.dim_size"] = np.int64(c["dim_size"])
        d[f"scatter{i}.out"] = c["out"].numpy()
    d["n_scatter"] = np.int64(len(SCATTER_LOG))
    d["n_cells"] = np.int64(len(cell_calls))
    np.savez_compressed(os.path.join(OUT, f"bc_hgnn_L{latent}.npz"), **d)
    print(f"bc_hgnn_L{latent}.npz scatter calls", len(SCATTER_LOG), "supernodes",
          cell_calls[0][0][2].shape[0], "bipartite edges", bg.shape[1])


def gen_pool(latent=64):
    """K5 initial super-node pooling expression, HGNN_GMM.py:269, in isolation
    (same torch expression the reference evaluates there)."""
    g = torch.Generator().manual_seed(505)
    n, s = 400, 23
    nodes = torch.randn(n, latent, generator=g)
    bg, bw, _, _ = _bip(n, s, 5, 4, g)
    out = _scatter_add((nn.functional.normalize(nodes, p=1)[bg[0]]) * bw, bg[1], dim=0, dim_size=s)
    np.savez_compressed(os.path.join(OUT, "k5_pool.npz"), nodes=nodes.numpy(), bipartite_graph=bg.numpy(),
                        bipartite_edge_weights=bw.numpy(), out=out.numpy())
    print("k5_pool.npz")


def gen_dataset(utils):
    """TrackMLDataset.__getitem__ (utils.py:37-110) run by the reference itself on a synthetic event;
    torch.load is pointed at an in-memory stand-in for the pickled PyG Data object."""
    g = torch.Generator().manual_seed(606)
    n, e = 400, 1500

    class Event(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    def make():
        pid = torch.randint(0, 60, (n,), generator=g)          # 0 = noise
        ev = Event(x=torch.randn(n, 3, generator=g), cell_data=torch.randn(n, 4, generator=g), pid=pid,
                   hid=torch.arange(n), pt=torch.rand(n, generator=g) * 3,
                   edge_index=torch.randint(0, n - 40, (2, e), generator=g),       # last 40 hits isolated
                   modulewise_true_edges=torch.randint(0, n, (2, 300), generator=g),
                   signal_true_edges=torch.randint(0, n, (2, 200), generator=g),
                   y=torch.randint(0, 2, (e,), generator=g).bool(), y_pid=torch.randint(0, 2, (e,), generator=g).bool(),
                   primary=torch.randint(0, 2, (n,), generator=g))
        return ev

    cases = [dict(noise=True, hard_ptcut=0, remove_isolated=True, primary=False, n_hits=5, edge_dropping_ratio=0.),
             dict(noise=False, hard_ptcut=1.0, remove_isolated=False, primary=True, n_hits=3, edge_dropping_ratio=0.),
             dict(noise=False, hard_ptcut=0, remove_isolated=True, primary=False, n_hits=5, edge_dropping_ratio=0.)]
    out = {}
    real_load = torch.load
    for ci, hp in enumerate(cases):
        ev = make()
        for k, v in ev.items():
            out[f"case{ci}.in.{k}"] = v.numpy().copy()
        for k, v in hp.items():
            out[f"case{ci}.hp.{k}"] = np.array(v)
        torch.load = lambda *a, **k: ev
        try:
            res = utils.TrackMLDataset(["/nonexistent/event0"], hp, stage="train")[0]
        finally:
            torch.load = real_load
        for k, v in res.items():
            if torch.is_tensor(v):
                out[f"case{ci}.out.{k}"] = v.numpy().copy()
    out["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "dataset_masking.npz"), **out)
    print("dataset_masking.npz", len(cases), "cases")


# --------------------------------------------------------------------------
# Model-level fixtures at the latents BASELINE.json names (128 / 256 / 512).  Weights are a pure
# function of (parameter name, seed) -- tests/golden/seeded.py -- so only inputs, outputs and
# per-parameter checksums are stored.
# --------------------------------------------------------------------------
sys.path.insert(0, OUT)
import seeded  # noqa: E402


def ref_config(relpath):
    """the reference's own YAML (data), resolved the way Modules/training_utils.py:13-20 does
    (training_utils itself cannot be imported: it pulls in cuml)"""
    import yaml
    with open(os.path.join(REF, "Modules", relpath)) as f:
        hp = yaml.safe_load(f)
    raw = dict(hp)
    if hp["hidden"] == "ratio":
        hp["hidden"] = hp["hidden_ratio"] * hp["latent"]
    hp.setdefault("cluster_granularity", 0)
    return raw, hp


def gen_ref_configs():
    """the hyper-parameter dictionaries of the two configs BASELINE.json names, as parsed data (JSON):
    what 'existing configs drop in' is tested against on the CPU"""
    import json
    out = {}
    for key, rel in (("EC-IN", "EdgeClassifier/Configs/IN.yaml"),
                     ("BC-HGNN-GMM", "BipartiteClassification/Configs/HGNN_GMM.yaml")):
        raw, hp = ref_config(rel)
        out[key] = {"raw": raw, "resolved_hidden": hp["hidden"]}
    # parameter counts of the reference's own classes built from those configs
    return out


def gen_ec_in_config(EC, configs):
    """BASELINE config 2: EC-IN exactly as EdgeClassifier/Configs/IN.yaml builds it (latent 128, hidden
    256, 14 cells, 4,441,089 parameters); forward + backward of the reference's own class."""
    _, hp = ref_config("EdgeClassifier/Configs/IN.yaml")
    model = EC(hp)
    seeded.fill_parameters(model, 128)
    g = torch.Generator().manual_seed(1281)
    x, graph = synth_event(1500, 9000, g)
    x = x.clone()
    scores = model(x, graph)              # the reference sets x.requires_grad = True itself (IN.py:120)
    r = seeded.randn(128, "r_scores", scores.shape[0])
    (scores * r).sum().backward()
    n_params = sum(p.numel() for p in model.parameters())
    configs["EC-IN"]["n_params"] = int(n_params)
    d = dict(x=x.detach().numpy(), edge_index=graph.numpy(), scores=scores.detach().numpy(),
             r_scores=r.numpy(), grad_x=x.grad.numpy(), n_params=np.int64(n_params),
             param_checksums=seeded.checksums(model.named_parameters()),
             grad_sketch=seeded.grad_sketch((n, p.grad) for n, p in model.named_parameters()),
             seed=np.int64(128))
    np.savez_compressed(os.path.join(OUT, "ec_in_L128.npz"), **d)
    print("ec_in_L128.npz params", n_params, "edges", graph.shape[1])


def gen_hgnn_cell_seeded(gnn_utils, latent=256, seed=256):
    """one HierarchicalGNNCell at BASELINE config 3's width: forward + every gradient"""
    g = torch.Generator().manual_seed(seed)
    hp = dict(latent=latent, hidden=2 * latent, nb_edge_layer=2, nb_node_layer=3,
              layernorm=True, hidden_activation="GELU")
    cell = gnn_utils.HierarchicalGNNCell(hp)
    seeded.fill_parameters(cell, seed)
    n_nodes, n_super = 120, 11
    _, graph = synth_event(n_nodes, 420, g)
    graph = torch.cat([graph, graph.flip(0)], 1)
    bg, bw, sg, sw = _bip(n_nodes, n_super, 5, 4, g)
    bw.requires_grad_(True)
    sw.requires_grad_(True)
    shapes = dict(nodes=(n_nodes, latent), edges=(graph.shape[1], latent), supernodes=(n_super, latent),
                  superedges=(sg.shape[1], latent))
    ins = {k: seeded.randn(seed, "in." + k, *v).requires_grad_(True) for k, v in shapes.items()}
    rs = {k: seeded.randn(seed, "r." + k, *v) for k, v in shapes.items()}
    outs = cell(ins["nodes"], ins["edges"], ins["supernodes"], ins["superedges"], graph, bg, bw, sg, sw)
    names = ("nodes", "edges", "supernodes", "superedges")
    sum((o * rs[k]).sum() for k, o in zip(names, outs)).backward()
    d = dict(graph=graph.numpy(), bipartite_graph=bg.numpy(), bipartite_edge_weights=bw.detach().numpy(),
             super_graph=sg.numpy(), super_edge_weights=sw.detach().numpy(), latent=np.int64(latent),
             seed=np.int64(seed), param_checksums=seeded.checksums(cell.named_parameters()),
             grad_sketch=seeded.grad_sketch((n, p.grad) for n, p in cell.named_parameters()),
             grad_bipartite_edge_weights=bw.grad.numpy(), grad_super_edge_weights=sw.grad.numpy())
    for k, o in zip(names, outs):
        d["out_" + k] = o.detach().numpy()
        d["grad_" + k] = ins[k].grad.numpy()
    # two full weight gradients (the widest GEMMs), the rest is pinned by the sketch
    for k in ("edge_network.0.weight", "supernode_network.3.weight"):
        d["grad." + k] = dict(cell.named_parameters())[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, f"hgnn_cell_L{latent}.npz"), **d)
    print(f"hgnn_cell_L{latent}.npz")


def gen_bc_hgnn_config(BC, latent, configs=None):
    """BASELINE config 3 (latent 256) / the fp32 reference of config 4 (latent 512): BC-HGNN-GMM as
    BipartiteClassification/Configs/HGNN_GMM.yaml builds it (6 + 6 cells; 25,299,957 parameters at
    latent 256), full forward of the reference's own class; the hierarchy its stand-in clustering /
    kNN led to is captured with the tensors entering the first HierarchicalGNNCell."""
    _, hp = ref_config("BipartiteClassification/Configs/HGNN_GMM.yaml")
    if latent != hp["latent"]:
        hp["latent"] = latent
        hp["hidden"] = hp["hidden_ratio"] * latent
    torch.manual_seed(latent)
    np.random.seed(latent)
    model = BC(hp)
    seeded.fill_parameters(model, latent)
    model.eval()
    model.hgnn_block.super_graph_construction.knn_radius.fill_(2.0)
    model.hgnn_block.bipartite_graph_construction.knn_radius.fill_(2.0)
    g = torch.Generator().manual_seed(latent + 1)
    x, graph = synth_tracks(60, 9, g)
    calls = []
    cells = model.hgnn_block.hgnn_cells
    cells[0].register_forward_hook(lambda m, inp, out: calls.append(("first", [t.detach().clone() for t in inp])))
    cells[len(cells) - 1].register_forward_hook(
        lambda m, inp, out: calls.append(("last", [t.detach().clone() for t in out])))
    with torch.no_grad():
        bg, bscores, emb = model(x.clone(), graph)
    first = [c for c in calls if c[0] == "first"][0][1]
    last = [c for c in calls if c[0] == "last"][0][1]
    names = ["nodes", "edges", "supernodes", "superedges", "graph", "bipartite_graph",
             "bipartite_edge_weights", "super_graph", "super_edge_weights"]
    n_params = sum(p.numel() for p in model.parameters())
    d = dict(x=x.numpy(), edge_index=graph.numpy(), bipartite_graph=bg.numpy(),
             bipartite_scores=bscores.numpy(), embeddings=emb.numpy(), n_params=np.int64(n_params),
             param_checksums=seeded.checksums(model.named_parameters()), seed=np.int64(latent),
             latent=np.int64(latent))
    for nm, t in zip(names, first):
        if nm == "edges":                 # [2E, latent]: every 4th row pins it
            d["cell0.in.edges_rows"] = np.arange(0, t.shape[0], 4)
            d["cell0.in.edges_sub"] = t[::4].numpy()
        elif nm != "graph":
            d[f"cell0.in.{nm}"] = t.numpy()
    d["last.out.nodes"] = last[0].numpy()
    d["last.out.supernodes"] = last[2].numpy()
    np.savez_compressed(os.path.join(OUT, f"bc_hgnn_L{latent}.npz"), **d)
    if configs is not None and latent == 256:
        configs["BC-HGNN-GMM"]["n_params"] = int(n_params)
    print(f"bc_hgnn_L{latent}.npz params", n_params, "supernodes", first[2].shape[0], "bipartite edges",
          bg.shape[1], "super edges", first[7].shape[1])


def gen_bc_hgnn_backward(BC, latent=256, c_emb=0.1):
    """BASELINE config 3 TRAINS (bipartite_classification_base.py:194-200 calls the forward of HGNN_GMM.py:323-346 in
    training mode and back-propagates through it): the reference's own BC_HierarchicalGNN_GMM built from HGNN_GMM.yaml,
    in train() mode, forward WITH autograd and backward of the surrogate loss

        loss = (bipartite_scores * r).sum() + c_emb * (emb * emb.roll(1, 0)).sum()

    (the reference's hinge / matching losses are host-side scipy code outside the hot path; a loss linear in the
    scores plus a bilinear form of the embeddings sends a generic gradient through both outputs).  The discrete
    hierarchy decision is captured so that a replay can inject it: cluster label per hit, bipartite and super graph
    topology.  Stored gradients: d loss / d x, the (sum, |.|-sum, probe) sketch of EVERY parameter gradient, the full
    gradient of two wide weights and d loss / d bipartite_edge_weights."""
    _, hp = ref_config("BipartiteClassification/Configs/HGNN_GMM.yaml")
    if latent != hp["latent"]:
        hp["latent"] = latent
        hp["hidden"] = hp["hidden_ratio"] * latent
    seed = 1000 + latent
    torch.manual_seed(seed)
    np.random.seed(seed)
    model = BC(hp)
    seeded.fill_parameters(model, seed)
    model.train()
    hb = model.hgnn_block
    hb.super_graph_construction.knn_radius.fill_(2.0)
    hb.bipartite_graph_construction.knn_radius.fill_(2.0)
    g = torch.Generator().manual_seed(seed + 1)
    x, graph = synth_tracks(60, 9, g)
    cap = {}
    real_clustering = hb.clustering

    def clustering(xx, emb, gr):
        cl = real_clustering(xx, emb, gr)
        cap["clusters"] = cl.detach().clone()
        return cl

    hb.clustering = clustering

    def bip_hook(m, inp, out):
        cap["bg"], cap["bw"] = out[0].detach().clone(), out[1]
        out[1].retain_grad()

    def sup_hook(m, inp, out):
        cap["sg"], cap["sw"] = out[0].detach().clone(), out[1]
        out[1].retain_grad()

    hb.bipartite_graph_construction.register_forward_hook(bip_hook)
    hb.super_graph_construction.register_forward_hook(sup_hook)
    x = x.clone()
    bg, scores, emb = model(x, graph)          # sets x.requires_grad itself (HGNN_GMM.py:326)
    emb.retain_grad()
    r = seeded.randn(seed, "r_scores", scores.shape[0])
    loss = (scores * r).sum() + c_emb * (emb * emb.roll(1, 0)).sum()
    loss.backward()
    assert torch.equal(bg, cap["bg"])
    n_params = sum(p.numel() for p in model.parameters())
    named = [(n, p) for n, p in model.named_parameters()]
    no_grad = [n for n, p in named if p.grad is None]
    params = dict(named)
    d = dict(x=x.detach().numpy(), edge_index=graph.numpy(), clusters=cap["clusters"].numpy(),
             n_clusters=np.int64(int(cap["clusters"].max()) + 1), bipartite_graph=cap["bg"].numpy(),
             super_graph=cap["sg"].numpy(), bipartite_edge_weights=cap["bw"].detach().numpy(),
             super_edge_weights=cap["sw"].detach().numpy(), bipartite_scores=scores.detach().numpy(),
             embeddings=emb.detach().numpy(), r_scores=r.numpy(), c_emb=np.float64(c_emb), loss=np.float64(float(loss)),
             grad_x=x.grad.numpy(), grad_embeddings=emb.grad.numpy(),
             grad_bipartite_edge_weights=cap["bw"].grad.numpy(), grad_super_edge_weights=cap["sw"].grad.numpy(),
             n_params=np.int64(n_params), seed=np.int64(seed), latent=np.int64(latent),
             param_checksums=seeded.checksums(model.named_parameters()),
             grad_sketch=seeded.grad_sketch((n, p.grad if p.grad is not None else torch.zeros_like(p))
                                            for n, p in named),
             params_without_grad=np.array(no_grad if no_grad else [""]))
    for k in ("ignn_block.ignn_cells.0.edge_network.0.weight", "hgnn_block.hgnn_cells.5.supernode_network.3.weight",
              "hgnn_block.bipartite_graph_construction.weight_normalization.weight", "bipartite_output_layer.0.weight"):
        d["grad." + k] = params[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, f"bc_hgnn_train_L{latent}.npz"), **d)
    print(f"bc_hgnn_train_L{latent}.npz params", n_params, "clusters", int(d["n_clusters"]), "bipartite edges",
          cap["bg"].shape[1], "super edges", cap["sg"].shape[1], "loss", float(loss), "params without grad", no_grad)


def gen_large(gnn_utils, EC, BC):
    import json
    configs = gen_ref_configs()
    gen_ec_in_config(EC, configs)
    gen_hgnn_cell_seeded(gnn_utils, 256)
    gen_bc_hgnn_config(BC, 256, configs)
    gen_bc_hgnn_config(BC, 512)
    gen_bc_hgnn_backward(BC, 256)
    with open(os.path.join(OUT, "ref_configs.json"), "w") as f:
        json.dump(configs, f, indent=1, sort_keys=True)
    print("ref_configs.json")


if __name__ == "__main__":
    gnn_utils, utils, EC, BC = _import_reference()
    if "--only-dataset" in sys.argv:
        gen_dataset(utils)
        sys.exit(0)
    if "--only-bc-backward" in sys.argv:
        gen_bc_hgnn_backward(BC, 256)
        sys.exit(0)
    if "--only-large" in sys.argv:
        gen_large(gnn_utils, EC, BC)
        sys.exit(0)
    gen_k1_cases()
    gen_pool()
    gen_ignn_cell(gnn_utils, 32, 150, 700, 201)
    gen_ignn_cell(gnn_utils, 128, 120, 500, 202)
    gen_hgnn_cell(gnn_utils, 32, 150, 600, 17, 211)
    gen_hgnn_cell(gnn_utils, 64, 100, 400, 11, 212)
    gen_ec_in(EC, 32)
    gen_bc_hgnn(BC, 32)
    gen_dataset(utils)
    gen_large(gnn_utils, EC, BC)
