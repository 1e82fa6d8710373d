"""GPU parity of the fused gather->concat->MLP(+skip) fp32-MFMA kernel (K6+K7) through
``hgnn_mlp_forward_f32``: against the CPU oracle, the golden cell fixtures, and the
unfused GPU path at BASELINE widths."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _mk(in_w, L, layers, out_act, seed):
    from hierarchicalgnn_amd import make_mlp
    torch.manual_seed(seed)
    net = make_mlp(in_w, 2 * L, L, layers, layer_norm=True, output_activation=out_act, hidden_activation="GELU")
    for p in net.parameters():  # non-trivial LayerNorm affine / biases
        if p.dim() == 1:
            p.data.add_(0.2 * torch.randn_like(p))
    return net


@pytest.mark.parametrize("L,layers,nseg,M", [
    (32, 2, 3, 1000), (64, 2, 3, 777), (128, 2, 3, 515), (256, 2, 3, 300),
    (32, 3, 2, 640), (64, 3, 3, 129), (128, 3, 2, 200), (256, 3, 3, 150), (256, 3, 2, 64), (128, 2, 1, 1)])
def test_fused_mlp_vs_oracle(L, layers, nseg, M):
    from hierarchicalgnn_amd import fused
    from oracle import hgnn_oracle as O
    g = torch.Generator().manual_seed(L * 10 + layers)
    out_act = "Tanh" if layers == 2 else "GELU"
    net = _mk(nseg * L, L, layers, out_act, seed=L + layers)
    n_tab = 97
    table = torch.randn(n_tab, L, generator=g)
    idx0 = torch.randint(0, n_tab, (M,), generator=g)
    idx1 = torch.randint(0, n_tab, (M,), generator=g)
    direct = torch.randn(M, L, generator=g)
    segs_cpu = [(table, idx0), (table, idx1), (direct, None)][3 - nseg:]
    x = torch.cat([t if i is None else t[i] for t, i in segs_cpu], dim=1)
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    ref = O.mlp_apply(sd, "", x, layers, "GELU", out_act, True) + direct
    net = net.cuda()
    segs = [(t.cuda(), None if i is None else i.cuda()) for t, i in segs_cpu]
    with torch.no_grad():
        assert fused.supported(net, segs, segs[-1][0])
        n0 = fused.stats["fused_calls"]
        out = fused.fused_concat_mlp(net, segs, segs[-1][0])
        assert fused.stats["fused_calls"] == n0 + 1
    assert out.shape == ref.shape
    assert rel_err(out.cpu().numpy(), ref.numpy()) <= TOL


def test_fused_not_used_when_grad_is_recorded():
    from hierarchicalgnn_amd import fused
    net = _mk(3 * 32, 32, 2, "Tanh", 1).cuda()
    e = torch.randn(10, 32).cuda().requires_grad_(True)
    t = torch.randn(5, 32).cuda()
    i = torch.randint(0, 5, (10,)).cuda()
    assert not fused.supported(net, [(t, i), (t, i), (e, None)], e)
    with torch.no_grad():
        assert fused.supported(net, [(t, i), (t, i), (e, None)], e)


def test_unsupported_shapes_fall_to_library_path():
    from hierarchicalgnn_amd import fused, make_mlp
    with torch.no_grad():
        odd = make_mlp(24, 64, 32, 3, layer_norm=True).cuda()         # K=24: neither %16 nor <= 16
        assert not fused.supported(odd, [(torch.randn(9, 24).cuda(), None)], None)
        noln = make_mlp(32, 64, 32, 2, layer_norm=False).cuda()
        assert not fused.supported(noln, [(torch.randn(9, 32).cuda(), None)], None)
        wide = make_mlp(32, 64, 40, 3, layer_norm=True).cuda()        # out > hidden / 2: no tile layout for it
        assert not fused.supported(wide, [(torch.randn(9, 32).cuda(), None)], None)
        odd_out = make_mlp(32, 64, 22, 3, layer_norm=True).cuda()     # out not a multiple of 4
        assert not fused.supported(odd_out, [(torch.randn(9, 32).cuda(), None)], None)


@pytest.mark.parametrize("latent", [32, 128])
def test_interaction_cell_inference_uses_fused_and_matches_reference(latent):
    import hierarchicalgnn_amd as H
    from hierarchicalgnn_amd import fused
    z = load_golden(f"ignn_cell_L{latent}.npz")
    hp = dict(latent=latent, hidden=2 * latent, nb_edge_layer=2, nb_node_layer=3, layernorm=True,
              hidden_activation="GELU")
    cell = H.InteractionGNNCell(hp)
    cell.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")})
    cell = cell.cuda()
    n0 = fused.stats["fused_calls"]
    with torch.no_grad():
        on, oe = cell(torch.from_numpy(z["nodes"]).cuda(), torch.from_numpy(z["edges"]).cuda(),
                      torch.from_numpy(z["graph"]).cuda())
    assert fused.stats["fused_calls"] == n0 + 2          # node network + edge network
    assert rel_err(on.cpu().numpy(), z["out_nodes"]) <= TOL
    assert rel_err(oe.cpu().numpy(), z["out_edges"]) <= TOL


@pytest.mark.parametrize("latent", [32, 64])
def test_hierarchical_cell_inference_uses_fused_and_matches_reference(latent):
    import hierarchicalgnn_amd as H
    from hierarchicalgnn_amd import fused
    z = load_golden(f"hgnn_cell_L{latent}.npz")
    hp = dict(latent=latent, hidden=2 * latent, nb_edge_layer=2, nb_node_layer=3, layernorm=True,
              hidden_activation="GELU")
    cell = H.HierarchicalGNNCell(hp)
    cell.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")})
    cell = cell.cuda()
    names = ("nodes", "edges", "supernodes", "superedges")
    n0 = fused.stats["fused_calls"]
    with torch.no_grad():
        outs = cell(*[torch.from_numpy(z[k]).cuda() for k in names],
                    torch.from_numpy(z["graph"]).cuda(), torch.from_numpy(z["bipartite_graph"]).cuda(),
                    torch.from_numpy(z["bipartite_edge_weights"]).cuda(), torch.from_numpy(z["super_graph"]).cuda(),
                    torch.from_numpy(z["super_edge_weights"]).cuda())
    assert fused.stats["fused_calls"] == n0 + 4
    for nm, o in zip(names, outs):
        assert rel_err(o.cpu().numpy(), z["out_" + nm]) <= TOL, nm


def test_fused_vs_unfused_at_baseline_width():
    """L=256 (BASELINE headline width), 200k edges: fused MFMA kernel vs gather + library GEMMs"""
    import hierarchicalgnn_amd as H
    from hierarchicalgnn_amd import fused, mlp, synth
    torch.manual_seed(0)
    L = 256
    x, ei = synth.trackml_event(12_000, 100_000, seed=2)
    graph = synth.directed(ei).cuda()
    nodes = torch.randn(12_000, L, device="cuda")
    edges = torch.randn(graph.shape[1], L, device="cuda")
    net = _mk(3 * L, L, 2, "Tanh", 3).cuda()
    segs = [(nodes, graph[0]), (nodes, graph[1]), (edges, None)]
    with torch.no_grad():
        a = mlp.concat_mlp(net, segs, skip=edges)
        fused.set_enabled(False)
        try:
            b = mlp.concat_mlp(net, segs, skip=edges)
        finally:
            fused.set_enabled(True)
    assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= TOL


@pytest.mark.parametrize("L,kind", [(32, "node_enc"), (128, "edge_enc"), (256, "edge_enc"), (64, "node_enc")])
def test_fused_encoders_small_k(L, kind):
    """node / edge encoders (IN.py:26-46): K = 3 / 6 spatial coordinates, gathered 12-byte rows"""
    from hierarchicalgnn_amd import fused, make_mlp
    from oracle import hgnn_oracle as O
    g = torch.Generator().manual_seed(L)
    N, M = 300, 1111
    x = torch.rand(N, 3, generator=g) * 2 - 1
    i0 = torch.randint(0, N, (M,), generator=g)
    i1 = torch.randint(0, N, (M,), generator=g)
    torch.manual_seed(L + 1)
    if kind == "edge_enc":
        net = make_mlp(6, 2 * L, L, 2, layer_norm=True, output_activation="GELU", hidden_activation="GELU")
        segs_cpu, layers, xin = [(x, i0), (x, i1)], 2, torch.cat([x[i0], x[i1]], 1)
    else:
        net = make_mlp(3, 2 * L, L, 3, layer_norm=True, output_activation="GELU", hidden_activation="GELU")
        segs_cpu, layers, xin = [(x, None)], 3, x
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    ref = O.mlp_apply(sd, "", xin, layers, "GELU", "GELU", True)
    net = net.cuda()
    segs = [(t.cuda(), None if i is None else i.cuda()) for t, i in segs_cpu]
    with torch.no_grad():
        assert fused.supported(net, segs, None)
        out = fused.fused_concat_mlp(net, segs, None)
    assert rel_err(out.cpu().numpy(), ref.numpy()) <= TOL


@pytest.mark.parametrize("L,act", [(32, "GELU"), (128, "GELU"), (128, "Tanh"), (256, "Tanh"), (64, "ReLU")])
def test_fused_heads_width_one(L, act):
    """classifier heads 2L -> H -> H -> 1, plain last layer (IN.py:107-115; HGNN_GMM.py:313-321)"""
    from hierarchicalgnn_amd import fused, make_mlp
    from oracle import hgnn_oracle as O
    g = torch.Generator().manual_seed(L + 7)
    M = 777
    torch.manual_seed(L)
    net = make_mlp(2 * L, 2 * L, 1, 3, layer_norm=True, output_activation=None, hidden_activation=act)
    a = torch.randn(M, L, generator=g)
    tab = torch.randn(50, L, generator=g)
    idx = torch.randint(0, 50, (M,), generator=g)
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    ref = O.mlp_apply(sd, "", torch.cat([a, tab[idx]], 1), 3, act, None, True)
    net = net.cuda()
    segs = [(a.cuda(), None), (tab.cuda(), idx.cuda())]
    with torch.no_grad():
        assert fused.supported(net, segs, None)
        out = fused.fused_concat_mlp(net, segs, None)
    assert out.shape == (M, 1)
    assert rel_err(out.cpu().numpy(), ref.numpy()) <= TOL


@pytest.mark.parametrize("L,act,M", [(32, "GELU", 777), (128, "Tanh", 1500), (256, "Tanh", 400), (128, "GELU", 0)])
def test_fused_heads_width_one_under_autograd(L, act, M):
    """the score heads in a TRAINING step (IN.py:126-127, HGNN_GMM.py:342-344 under autograd): the two LayerNorm'ed
    hidden layers on the differentiable fused kernel (pre-LayerNorm dumps + hand-written backward), the plain last
    Linear trailing -- outputs and every gradient against autograd through the nn.Sequential itself"""
    from hierarchicalgnn_amd import fused, make_mlp, mlp
    g = torch.Generator().manual_seed(L + 11)
    torch.manual_seed(L + 1)
    net = make_mlp(2 * L, 2 * L, 1, 3, layer_norm=True, output_activation=None, hidden_activation=act).cuda()
    for p in net.parameters():
        if p.dim() == 1:
            p.data.add_(0.2 * torch.randn_like(p))
    a0 = torch.randn(M, L, generator=g).cuda()
    tab0 = torch.randn(60, L, generator=g).cuda()
    idx = torch.randint(0, 60, (M,), generator=g).cuda()
    r = torch.randn(M, 1, generator=g).cuda()

    def run(use_fused):
        for p in net.parameters():
            p.grad = None
        a, tab = a0.clone().requires_grad_(True), tab0.clone().requires_grad_(True)
        if use_fused:
            n0, h0 = fused.stats["fused_train_calls"], fused.stats.get("fused_head_train_calls", 0)
            out = mlp.concat_mlp(net, [(a, None), (tab, idx)])
            assert fused.stats["fused_train_calls"] == n0 + 1 and fused.stats["fused_head_train_calls"] == h0 + 1
        else:
            out = net(torch.cat([a, tab[idx]], dim=1))
        (out * r).sum().backward()
        return out.detach(), a.grad, tab.grad, [p.grad.clone() for p in net.parameters()]

    o, ga, gt, gp = run(True)
    o_ref, ga_ref, gt_ref, gp_ref = run(False)
    assert o.shape == (M, 1)
    if M == 0:
        assert all(float(x.abs().max()) == 0.0 for x in gp[:8])
        return
    assert rel_err(o.cpu().numpy(), o_ref.cpu().numpy()) <= TOL
    assert rel_err(ga.cpu().numpy(), ga_ref.cpu().numpy()) <= TOL
    assert rel_err(gt.cpu().numpy(), gt_ref.cpu().numpy()) <= TOL
    for (name, _), x, y in zip(net.named_parameters(), gp, gp_ref):
        assert rel_err(x.cpu().numpy(), y.cpu().numpy()) <= TOL, name


@pytest.mark.parametrize("L,layers", [(32, 2), (128, 2), (64, 3), (256, 2), (256, 3)])
def test_fused_train_backward_matches_autograd(L, layers):
    """differentiable fused MLP (kernel forward with pre-LN dumps + hand-written backward) against
    autograd through the unfused path: outputs and every gradient"""
    from hierarchicalgnn_amd import fused, mlp
    g = torch.Generator().manual_seed(L + layers)
    out_act = "Tanh" if layers == 2 else "GELU"
    net = _mk(3 * L, L, layers, out_act, seed=L).cuda()
    n_tab, M = 83, 500
    table0 = torch.randn(n_tab, L, generator=g).cuda()
    direct0 = torch.randn(M, L, generator=g).cuda()
    i0 = torch.randint(0, n_tab, (M,), generator=g).cuda()
    i1 = torch.randint(0, n_tab, (M,), generator=g).cuda()
    r = torch.randn(M, L, generator=g).cuda()
    results = {}
    for name, on in (("fused", True), ("library", False)):
        fused.set_enabled(True, train=on)
        net.zero_grad(set_to_none=True)
        table = table0.clone().requires_grad_(True)
        direct = direct0.clone().requires_grad_(True)
        n0 = fused.stats["fused_train_calls"]
        out = mlp.concat_mlp(net, [(table, i0), (table, i1), (direct, None)], skip=direct)
        assert (fused.stats["fused_train_calls"] == n0 + 1) == on
        (out * r).sum().backward()
        results[name] = [out.detach(), table.grad, direct.grad] + [p.grad.clone() for p in net.parameters()]
    fused.set_enabled(True)
    for a, b in zip(results["fused"], results["library"]):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= TOL


@pytest.mark.parametrize("W", [64, 128, 256, 512, 1024])
@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_ln_act_row_kernels_against_autograd(W, act):
    """hgnn_ln_act_forward_f32 / hgnn_ln_act_backward_f32 (one make_mlp layer's LayerNorm + activation and
    its backward incl. dgamma / dbeta / dbias column sums) against torch autograd; ragged row count"""
    from hierarchicalgnn_amd import fused
    g = torch.Generator().manual_seed(W + act)
    M = 1000 + act   # not a multiple of the 16-row step
    z = (torch.randn(M, W, generator=g) * 1.5 + 0.3).cuda()
    gamma = (1 + 0.2 * torch.randn(W, generator=g)).cuda()
    beta = (0.2 * torch.randn(W, generator=g)).cuda()
    go = torch.randn(M, W, generator=g).cuda()
    fn = {0: lambda t: t, 1: torch.nn.functional.gelu, 2: torch.tanh, 3: torch.relu}[act]
    zr = z.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = fn(torch.nn.functional.layer_norm(zr, [W], gr, br, 1e-5))
    ref.backward(go)
    out = fused._ln_act_forward(z, gamma, beta, act, 1e-5)
    assert rel_err(out.cpu().numpy(), ref.detach().cpu().numpy()) <= TOL
    dz, dg, db, dbias = fused._ln_act_backward(z, go, gamma, beta, act, 1e-5)
    assert rel_err(dz.cpu().numpy(), zr.grad.cpu().numpy()) <= TOL
    assert rel_err(dg.cpu().numpy(), gr.grad.cpu().numpy()) <= TOL
    assert rel_err(db.cpu().numpy(), br.grad.cpu().numpy()) <= TOL
    assert rel_err(dbias.cpu().numpy(), zr.grad.sum(0).cpu().numpy(), ) <= 10 * TOL   # sums to ~0: absolute scale
    # deterministic (per-workgroup partials, no atomics)
    dz2, dg2, db2, dbias2 = fused._ln_act_backward(z, go, gamma, beta, act, 1e-5)
    assert torch.equal(dz, dz2) and torch.equal(dg, dg2) and torch.equal(db, db2) and torch.equal(dbias, dbias2)


def test_ln_act_row_kernels_empty_and_errors():
    from hierarchicalgnn_amd import fused
    z = torch.zeros(0, 128).cuda()
    gamma, beta = torch.ones(128).cuda(), torch.zeros(128).cuda()
    assert fused._ln_act_forward(z, gamma, beta, 1, 1e-5).shape == (0, 128)
    dz, dg, db, dbias = fused._ln_act_backward(z, z, gamma, beta, 1, 1e-5)
    assert dz.shape == (0, 128) and float(dg.abs().sum()) == 0.0 and float(dbias.abs().sum()) == 0.0
    with pytest.raises(RuntimeError, match="width must be"):
        fused._ln_act_forward(torch.zeros(4, 96).cuda(), torch.ones(96).cuda(), torch.zeros(96).cuda(), 1, 1e-5)


@pytest.mark.parametrize("L,layers", [(32, 2), (64, 2), (128, 2), (256, 2), (128, 3)])
def test_preprojected_segments_equal_the_full_k_kernel(L, layers):
    """hgnn_mlp_desc.n_pre: projecting the gathered node segments through their block of the first Linear
    (N-row GEMMs) and gathering the projections inside the kernel == the kernel run on all 3L columns"""
    from hierarchicalgnn_amd import fused
    g = torch.Generator().manual_seed(7 * L + layers)
    net = _mk(3 * L, L, layers, "Tanh" if layers == 2 else "GELU", seed=L + 1).cuda()
    n_tab, M = 101, 1777
    table = torch.randn(n_tab, L, generator=g).cuda()
    direct = torch.randn(M, L, generator=g).cuda()
    i0 = torch.randint(0, n_tab, (M,), generator=g).cuda()
    i1 = torch.randint(0, n_tab, (M,), generator=g).cuda()
    segs = [(table, i0), (table, i1), (direct, None)]
    outs = {}
    try:
        with torch.no_grad():
            for on in (True, False):
                fused.set_preproject(on)
                d = fused._descriptor(net, segs, direct)[0]
                assert int(d.n_pre) == (2 if on else 0) and int(d.n_seg) == (1 if on else 3)
                outs[on] = fused.fused_concat_mlp(net, segs, direct)
    finally:
        fused.set_preproject(True)
    assert rel_err(outs[True].cpu().numpy(), outs[False].cpu().numpy()) <= 1e-5


def test_fused_paths_ignore_a_callers_autocast_region():
    """the fused kernels read fp32 buffers: host-side GEMMs around them (segment projections, weight
    gradients) must not follow an enclosing torch.autocast region into bf16"""
    from hierarchicalgnn_amd import mlp
    g = torch.Generator().manual_seed(11)
    L, M, n_tab = 64, 900, 50
    net = _mk(3 * L, L, 2, "Tanh", seed=3).cuda()
    table0 = torch.randn(n_tab, L, generator=g).cuda()
    direct0 = torch.randn(M, L, generator=g).cuda()
    i0 = torch.randint(0, n_tab, (M,), generator=g).cuda()
    i1 = torch.randint(0, n_tab, (M,), generator=g).cuda()
    res = {}
    for ac in (False, True):
        net.zero_grad(set_to_none=True)
        table = table0.clone().requires_grad_(True)
        direct = direct0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
            with torch.no_grad():
                inf = mlp.concat_mlp(net, [(table, i0), (table, i1), (direct, None)], skip=direct)
            out = mlp.concat_mlp(net, [(table, i0), (table, i1), (direct, None)], skip=direct)
        out.sum().backward()
        res[ac] = [inf, out.detach(), table.grad, direct.grad] + [p.grad.clone() for p in net.parameters()]
    for a, b in zip(res[True], res[False]):
        assert a.dtype == torch.float32 and torch.equal(a, b)


@pytest.mark.parametrize("L,emb", [(32, 8), (64, 8), (128, 8), (256, 8), (256, 12), (64, 4)])
def test_fused_supernode_encoder_partial_width(L, emb):
    """supernode encoder L -> H -> H -> L - emb_dim (HGNN_GMM.py:117,:270): LayerNorm + GELU over an output
    that is narrower than its last 16-feature tile row (248 at latent 256)"""
    from hierarchicalgnn_amd import fused, make_mlp
    from oracle import hgnn_oracle as O
    g = torch.Generator().manual_seed(L + emb)
    M = 333
    torch.manual_seed(L)
    net = make_mlp(L, 2 * L, L - emb, 3, layer_norm=True, output_activation="GELU", hidden_activation="GELU")
    for p in net.parameters():
        if p.dim() == 1:
            p.data.add_(0.2 * torch.randn_like(p))
    x = torch.randn(M, L, generator=g)
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    ref = O.mlp_apply(sd, "", x, 3, "GELU", "GELU", True)
    net = net.cuda()
    with torch.no_grad():
        assert fused.supported(net, [(x.cuda(), None)], None)
        n0 = fused.stats["fused_calls"]
        out = fused.fused_concat_mlp(net, [(x.cuda(), None)], None)
        assert fused.stats["fused_calls"] == n0 + 1
    assert out.shape == (M, L - emb)
    assert rel_err(out.cpu().numpy(), ref.numpy()) <= TOL


@pytest.mark.parametrize("L,emb,act", [(32, 8, "Tanh"), (128, 8, "Tanh"), (256, 8, "Tanh"), (256, 8, "GELU"), (64, 3, "Tanh"),
                                       (128, 32, "GELU")])
def test_fused_embedding_head(L, emb, act):
    """embedding head L -> H -> H -> emb_dim with a plain last layer (HGNN_GMM.py:74-82, :96)"""
    from hierarchicalgnn_amd import fused, make_mlp
    from oracle import hgnn_oracle as O
    g = torch.Generator().manual_seed(L + emb)
    M = 500
    torch.manual_seed(L + 1)
    net = make_mlp(L, 2 * L, emb, 3, layer_norm=True, output_activation=None, hidden_activation=act)
    x = torch.randn(M, L, generator=g)
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    ref = O.mlp_apply(sd, "", x, 3, act, None, True)
    net = net.cuda()
    with torch.no_grad():
        assert fused.supported(net, [(x.cuda(), None)], None)
        out = fused.fused_concat_mlp(net, [(x.cuda(), None)], None)
    assert out.shape == (M, emb)
    assert rel_err(out.cpu().numpy(), ref.numpy()) <= TOL


@pytest.mark.both_fp32_gemms
def test_bc_forward_latent256_has_no_library_mlp():
    """f2: every MLP of a BC-HGNN-GMM inference forward at the shipped config (latent 256) runs on the fused
    kernel -- encoders (node, edge, supernode, superedge), 6 + 6 cells, embedding head, bipartite head; none
    falls to the library path of mlp.concat_mlp"""
    import json
    import os
    import conftest
    from hierarchicalgnn_amd import fused, mlp, synth
    from hierarchicalgnn_amd.models import BC_MessagePassing
    with open(os.path.join(conftest.GOLDEN, "ref_configs.json")) as f:
        raw = json.load(f)["BC-HGNN-GMM"]["raw"]
    torch.manual_seed(0)
    model = BC_MessagePassing(raw).cuda().eval()
    x, ei = synth.trackml_event(3000, 18000, seed=5)
    x, ei = x.cuda(), ei.cuda()
    calls = {"n": 0}
    real = mlp.concat_mlp

    def counting(net, segments, skip=None, bf16_tail=False, out=None):
        calls["n"] += 1
        return real(net, segments, skip, bf16_tail, out)

    import hierarchicalgnn_amd.gnn_utils as gu
    import hierarchicalgnn_amd.models as mo
    gu.concat_mlp = mo.concat_mlp = counting
    try:
        n0 = fused.stats["fused_calls"]
        with torch.no_grad():
            directed, emb, nodes, edges, _ = model.embed(x, ei)
            bg, bw = synth.bipartite_assignment(3000, 40, 5, seed=2)
            sg, sw = synth.super_graph(40, 10, seed=2)
            means = torch.nn.functional.normalize(torch.randn(40, 8)).cuda()
            n_out, sn_out, _, _ = model.hgnn_block(nodes, edges, directed, means, bg.cuda(), bw.cuda(), sg.cuda(),
                                                    sw.cuda())
            s = model.score(n_out, sn_out, bg.cuda())
    finally:
        gu.concat_mlp = mo.concat_mlp = real
    assert bool(torch.isfinite(s).all())
    expected = 2 + 2 * 6 + 1 + 2 + 4 * 6 + 1        # encoders, IGNN cells, emb head, super encoders, HGNN cells, head
    assert calls["n"] == expected
    assert fused.stats["fused_calls"] - n0 == expected


@pytest.mark.parametrize("L,kind", [(64, "node_enc"), (128, "edge_enc"), (256, "edge_enc")])
def test_fused_train_small_k_encoders(L, kind):
    """the node / edge encoders (K = 3 / 6, IN.py:84-85) on the differentiable fused path: gradients w.r.t. the
    hit coordinates (the reference sets x.requires_grad, IN.py:82) and the weights against plain autograd"""
    from hierarchicalgnn_amd import fused, make_mlp, mlp
    g = torch.Generator().manual_seed(L + 11)
    N, M = 400, 2500
    x = (torch.rand(N, 3, generator=g) * 2 - 1).cuda()
    i0 = torch.randint(0, N, (M,), generator=g).cuda()
    i1 = torch.randint(0, N, (M,), generator=g).cuda()
    torch.manual_seed(L)
    if kind == "edge_enc":
        net = make_mlp(6, 2 * L, L, 2, layer_norm=True, output_activation="GELU", hidden_activation="GELU").cuda()
        rows = M
    else:
        net = make_mlp(3, 2 * L, L, 3, layer_norm=True, output_activation="GELU", hidden_activation="GELU").cuda()
        rows = N
    r = torch.randn(rows, L, generator=g).cuda()

    def run(use_fused):
        for p in net.parameters():
            p.grad = None
        xx = x.clone().requires_grad_(True)
        segs = [(xx, i0), (xx, i1)] if kind == "edge_enc" else [(xx, None)]
        if use_fused:
            n0 = fused.stats["fused_train_calls"]
            out = mlp.concat_mlp(net, segs)
            assert fused.stats["fused_train_calls"] == n0 + 1
        else:
            out = net(torch.cat([t if i is None else t[i] for t, i in segs], dim=1))
        (out * r).sum().backward()
        return out.detach(), xx.grad, [p.grad.clone() for p in net.parameters()]

    o_ref, gx_ref, gp_ref = run(False)
    o, gx, gp = run(True)
    assert rel_err(o.cpu().numpy(), o_ref.cpu().numpy()) <= TOL
    assert rel_err(gx.cpu().numpy(), gx_ref.cpu().numpy()) <= TOL
    for (name, _), a, b in zip(net.named_parameters(), gp, gp_ref):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= TOL, name


@pytest.mark.parametrize("layers,nseg,M", [(2, 3, 700), (3, 2, 333), (3, 3, 129)])
def test_fused_fp32_latent512_layer_chain(layers, nseg, M):
    """fp32 at latent 512 (hidden 1024 = 256 accumulators per lane: no single-launch kernel): one fused launch per
    [Linear, LayerNorm, act] layer, gathers / concat / pre-projection in the first, skip in the last"""
    from hierarchicalgnn_amd import fused
    from oracle import hgnn_oracle as O
    L = 512
    g = torch.Generator().manual_seed(layers * 10 + nseg)
    out_act = "Tanh" if layers == 2 else "GELU"
    net = _mk(nseg * L, L, layers, out_act, seed=layers)
    n_tab = 61
    table = torch.randn(n_tab, L, generator=g)
    idx0 = torch.randint(0, n_tab, (M,), generator=g)
    idx1 = torch.randint(0, n_tab, (M,), generator=g)
    direct = torch.randn(M, L, generator=g)
    segs_cpu = [(table, idx0), (table, idx1), (direct, None)][3 - nseg:]
    x = torch.cat([t if i is None else t[i] for t, i in segs_cpu], dim=1)
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    ref = O.mlp_apply(sd, "", x, layers, "GELU", out_act, True) + direct
    net = net.cuda()
    segs = [(t.cuda(), None if i is None else i.cuda()) for t, i in segs_cpu]
    with torch.no_grad():
        assert fused.supported(net, segs, segs[-1][0])
        n0 = fused.stats["fused_calls"]
        out = fused.fused_concat_mlp(net, segs, segs[-1][0])
        assert fused.stats["fused_calls"] == n0 + layers
    assert rel_err(out.cpu().numpy(), ref.numpy()) <= TOL


def test_fused_fp32_latent512_head_chain():
    """the width-1 classifier head at latent 512 (2L -> 1024 -> 1024 -> 1): the two LayerNorm layers as fused
    single-layer launches, the plain last Linear as a trailing matrix-vector product"""
    from hierarchicalgnn_amd import fused, make_mlp
    from oracle import hgnn_oracle as O
    g = torch.Generator().manual_seed(99)
    M, L = 600, 512
    torch.manual_seed(3)
    net = make_mlp(2 * L, 2 * L, 1, 3, layer_norm=True, output_activation=None, hidden_activation="GELU")
    a = torch.randn(M, L, generator=g)
    tab = torch.randn(40, L, generator=g)
    idx = torch.randint(0, 40, (M,), generator=g)
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    ref = O.mlp_apply(sd, "", torch.cat([a, tab[idx]], 1), 3, "GELU", None, True)
    net = net.cuda()
    segs = [(a.cuda(), None), (tab.cuda(), idx.cuda())]
    with torch.no_grad():
        assert fused.supported(net, segs, None)
        n0 = fused.stats["fused_calls"]
        out = fused.fused_concat_mlp(net, segs, None)
        assert fused.stats["fused_calls"] == n0 + 2
    assert out.shape == (M, 1)
    assert rel_err(out.cpu().numpy(), ref.numpy()) <= TOL
