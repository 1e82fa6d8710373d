"""GPU parity of the module-level boundary: InteractionGNNCell / HierarchicalGNNCell
loaded with the REFERENCE's state_dict (fixtures produced by running the
reference's own classes) reproduce its outputs and gradients."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _hp(latent, ckpt=True):
    return dict(latent=latent, hidden=2 * latent, nb_edge_layer=2, nb_node_layer=3, layernorm=True,
                hidden_activation="GELU", checkpointing=ckpt)


def _load(cell, z, prefix="sd."):
    sd = {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}
    missing, unexpected = cell.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return cell.cuda()


@pytest.mark.both_fp32_gemms(must_run=False)
@pytest.mark.parametrize("latent,ckpt", [(32, True), (32, False), (128, True)])
def test_interaction_cell(latent, ckpt):
    import hierarchicalgnn_amd as H
    z = load_golden(f"ignn_cell_L{latent}.npz")
    cell = _load(H.InteractionGNNCell(_hp(latent, ckpt)), z)
    nodes = torch.from_numpy(z["nodes"]).cuda().requires_grad_(True)
    edges = torch.from_numpy(z["edges"]).cuda().requires_grad_(True)
    graph = torch.from_numpy(z["graph"]).cuda()
    on, oe = cell(nodes, edges, graph)
    assert rel_err(on.detach().cpu().numpy(), z["out_nodes"]) <= TOL
    assert rel_err(oe.detach().cpu().numpy(), z["out_edges"]) <= TOL
    ((on * torch.from_numpy(z["r_nodes"]).cuda()).sum() + (oe * torch.from_numpy(z["r_edges"]).cuda()).sum()).backward()
    assert rel_err(nodes.grad.cpu().numpy(), z["grad_nodes"]) <= TOL
    assert rel_err(edges.grad.cpu().numpy(), z["grad_edges"]) <= TOL
    for k, p in cell.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), z["grad." + k]) <= TOL, k


@pytest.mark.parametrize("latent", [32, 64])
def test_hierarchical_cell(latent):
    import hierarchicalgnn_amd as H
    z = load_golden(f"hgnn_cell_L{latent}.npz")
    cell = _load(H.HierarchicalGNNCell(_hp(latent)), z)
    names = ("nodes", "edges", "supernodes", "superedges")
    t = {k: torch.from_numpy(z[k]).cuda().requires_grad_(True)
         for k in names + ("bipartite_edge_weights", "super_edge_weights")}
    outs = cell(t["nodes"], t["edges"], t["supernodes"], t["superedges"],
                torch.from_numpy(z["graph"]).cuda(), torch.from_numpy(z["bipartite_graph"]).cuda(),
                t["bipartite_edge_weights"], torch.from_numpy(z["super_graph"]).cuda(), t["super_edge_weights"])
    for nm, o in zip(names, outs):
        assert rel_err(o.detach().cpu().numpy(), z["out_" + nm]) <= TOL, nm
    sum((o * torch.from_numpy(z["r_" + nm]).cuda()).sum() for nm, o in zip(names, outs)).backward()
    for nm in t:
        assert rel_err(t[nm].grad.cpu().numpy(), z["grad_" + nm]) <= TOL, nm
    for k, p in cell.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), z["grad." + k]) <= TOL, k


def test_bc_hgnn_cell_loop_from_reference_forward():
    """the HierarchicalGNNCell inputs/outputs captured inside the reference's BC-HGNN-GMM forward"""
    import hierarchicalgnn_amd as H
    z = load_golden("bc_hgnn_L32.npz")
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
    names = ["nodes", "edges", "supernodes", "superedges", "graph", "bipartite_graph",
             "bipartite_edge_weights", "super_graph", "super_edge_weights"]
    for i in range(int(z["n_cells"])):
        cell = _load(H.HierarchicalGNNCell(hp), z, prefix=f"sd.hgnn_block.hgnn_cells.{i}.")
        args = [torch.from_numpy(z[f"cell{i}.in.{n}"]).cuda() for n in names]
        with torch.no_grad():
            outs = cell(*args)
        for nm, o in zip(names[:4], outs):
            assert rel_err(o.cpu().numpy(), z[f"cell{i}.out.{nm}"]) <= TOL, (i, nm)


def test_ec_in_full_forward_config1_shape():
    """BASELINE config 1/2 model: EC-IN forward(x, edge_index) with the reference's
    state_dict reproduces the reference's scores (latent=32, 14 cells, 286,977 params)."""
    from hierarchicalgnn_amd.models import EC_InteractionGNN
    z = load_golden("ec_in_L32.npz")
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
    model = EC_InteractionGNN(hp)
    assert sum(p.numel() for p in model.parameters()) == int(z["n_params"]) == 286977
    model = _load(model, z)
    x = torch.from_numpy(z["x"]).cuda()
    graph = torch.from_numpy(z["edge_index"]).cuda()
    with torch.no_grad():
        scores = model(x, graph)
    assert scores.shape == z["scores"].shape
    assert np.abs(scores.cpu().numpy() - z["scores"]).max() <= 1e-4
    # training-mode call: reentrant checkpointing + autograd through every HIP op
    scores = model(x, graph)
    scores.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_torch_scatter_shim_resolves_to_hip():
    import importlib
    import sys
    import conftest
    sys.path.insert(0, conftest.ROOT + "/torch_scatter_shim")
    try:
        ts = importlib.import_module("torch_scatter")
        src = torch.randn(100, 8).cuda()
        idx = torch.randint(0, 9, (100,)).cuda()
        ref = torch.zeros(9, 8).index_add_(0, idx.cpu(), src.cpu())
        assert rel_err(ts.scatter_add(src, idx, dim=0, dim_size=9).cpu().numpy(), ref.numpy()) <= TOL
        cnt = torch.bincount(idx.cpu(), minlength=9).clamp(min=1).float().unsqueeze(1)
        assert rel_err(ts.scatter_mean(src, idx, dim=0, dim_size=9).cpu().numpy(), (ref / cnt).numpy()) <= TOL
    finally:
        sys.path.pop(0)
        sys.modules.pop("torch_scatter", None)


def test_bc_hgnn_message_passing_against_reference_forward():
    """BASELINE config 3 arithmetic at small latent: IGNN block, K5 pooling + encoders (a8),
    HGNN cell loop and the bipartite head reproduce the tensors captured inside the
    reference's BC_HierarchicalGNN_GMM.forward (the hierarchy itself is taken from the capture)."""
    from hierarchicalgnn_amd.models import BC_MessagePassing
    z = load_golden("bc_hgnn_L32.npz")
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
    model = BC_MessagePassing(hp)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    missing, unexpected = model.load_state_dict(sd, strict=True)   # every reference key has a home
    assert not missing and not unexpected
    model = model.cuda().eval()
    x = torch.from_numpy(z["x"]).cuda()
    graph = torch.from_numpy(z["edge_index"]).cuda()
    t = lambda k: torch.from_numpy(z[k]).cuda()
    with torch.no_grad():
        directed, emb, nodes, edges, order = model.embed(x, graph)
        assert order is not None and bool((directed[1][1:] >= directed[1][:-1]).all())   # sorted layout
        assert rel_err(emb.cpu().numpy(), z["embeddings"]) <= TOL
        assert rel_err(nodes.cpu().numpy(), z["cell0.in.nodes"]) <= TOL
        assert rel_err(edges.cpu().numpy(), z["cell0.in.edges"][order.cpu().numpy()]) <= TOL
        d2, _, _, e2, o2 = model.embed(x, graph, restore_order=True)                     # interface order
        assert o2 is None and rel_err(e2.cpu().numpy(), z["cell0.in.edges"]) <= TOL
        means = t("cell0.in.supernodes")[:, :hp["emb_dim"]].contiguous()
        bg, bw = t("cell0.in.bipartite_graph"), t("cell0.in.bipartite_edge_weights")
        sg, sw = t("cell0.in.super_graph"), t("cell0.in.super_edge_weights")
        # a8: pooled + encoded supernodes / superedges as they enter the first cell
        blk = model.hgnn_block
        from hierarchicalgnn_amd import gather_scale_scatter, l1_row_scale
        pooled = gather_scale_scatter(nodes, bg[0], bg[1], means.shape[0], bw, row_scale=l1_row_scale(nodes))
        sn0 = torch.cat([means, blk._encode_supernodes(pooled)], dim=-1)
        assert rel_err(sn0.cpu().numpy(), z["cell0.in.supernodes"]) <= TOL
        se0 = blk._encode_superedges(sn0, sg)
        assert rel_err(se0.cpu().numpy(), z["cell0.in.superedges"]) <= TOL
        n_out, sn_out, e_out, se_out = blk(nodes, edges, directed, means, bg, bw, sg, sw)
        last = int(z["n_cells"]) - 1
        assert rel_err(n_out.cpu().numpy(), z[f"cell{last}.out.nodes"]) <= TOL
        assert rel_err(sn_out.cpu().numpy(), z[f"cell{last}.out.supernodes"]) <= TOL
        scores = model.score(n_out, sn_out, t("bipartite_graph"))
        assert np.abs(scores.cpu().numpy() - z["bipartite_scores"]).max() <= 1e-4


def test_hierarchy_from_clusters_matches_oracle_pieces():
    """centroids (K8) -> kNN graphs (K9) -> attention weights (K11), eval mode"""
    from hierarchicalgnn_amd.models import HierarchicalGNNBlock
    from oracle import hgnn_oracle as O
    g = torch.Generator().manual_seed(21)
    hp = dict(latent=32, hidden=64, emb_dim=8, nb_node_layer=3, nb_edge_layer=2, layernorm=True,
              hidden_activation="GELU", n_hierarchical_graph_iters=1, share_weight=False,
              supergraph_sparsity=4, bipartitegraph_sparsity=3)
    blk = HierarchicalGNNBlock(hp).cuda().eval()
    blk.super_graph_construction.knn_radius.fill_(1.5)
    blk.bipartite_graph_construction.knn_radius.fill_(1.5)
    n, c = 500, 37
    emb = torch.nn.functional.normalize(torch.randn(n, 8, generator=g))
    clusters = torch.randint(-1, c, (n,), generator=g)
    clusters[:c] = torch.arange(c)
    with torch.no_grad():
        means, bg, bw, sg, sw, _ = blk.hierarchy_from_clusters(emb.cuda(), clusters.cuda())
    m = clusters >= 0
    sums = O.scatter_add(emb[m], clusters[m], 0, c)
    cnt = torch.bincount(clusters[m], minlength=c).clamp(min=1).unsqueeze(1)
    means_ref = torch.nn.functional.normalize(sums / cnt)
    assert rel_err(means.cpu().numpy(), means_ref.numpy()) <= TOL
    idx, _ = O.knn_radius(emb, means_ref, 3, 1.5)
    assert bg.shape[1] == int((idx >= 0).sum())
    one, zero = torch.ones(1), torch.zeros(1)
    w_ref, _ = O.graph_edge_weights(emb, means_ref, bg.cpu(), one, zero, zero, one, "exp", True)
    assert rel_err(bw.cpu().numpy(), w_ref.numpy()) <= TOL
    w_ref, _ = O.graph_edge_weights(means_ref, means_ref, sg.cpu(), one, zero, zero, one, "sigmoid", True)
    assert rel_err(sw.cpu().numpy(), w_ref.numpy()) <= TOL
    assert torch.equal(sg.cpu(), torch.unique(torch.cat([sg.cpu(), sg.cpu().flip(0)], 1), dim=1))  # symmetric


def test_graphed_inference_replays_the_small_event_forward():
    """BASELINE config 1 shape (2k hits / 12k edges, latent 32): captured HIP graph == eager forward"""
    from hierarchicalgnn_amd.models import EC_InteractionGNN, GraphedInference
    z = load_golden("ec_in_L32.npz")
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
    model = _load(EC_InteractionGNN(hp), z).eval()
    x = torch.from_numpy(z["x"]).cuda()
    graph = torch.from_numpy(z["edge_index"]).cuda()
    g = GraphedInference(model, x, graph)
    s1 = g().clone()
    assert np.abs(s1.cpu().numpy() - z["scores"]).max() <= 1e-4
    x2 = x + 0.01 * torch.randn_like(x)
    s2 = g(x2).clone()
    with torch.no_grad():
        ref2 = model(x2, graph)
    assert torch.allclose(s2, ref2, rtol=1e-5, atol=1e-6)
    assert not torch.allclose(s1, s2)
