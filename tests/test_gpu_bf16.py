"""bf16 feature rows (BASELINE config 4 dtype) through hgnn_*_bf16: fp32 accumulation, one rounding.
Oracle: the fp32 CPU restatement applied to the bf16-rounded inputs, rounded to bf16 once."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
BF16_TOL = 2.0 ** -7   # one bf16 ulp at the scale of the largest element (8 significant bits)


@pytest.mark.parametrize("F", [32, 64, 128, 256, 512, 40])
def test_k1_bf16_forward_backward(F):
    import hierarchicalgnn_amd as H
    from oracle import hgnn_oracle as O
    g = torch.Generator().manual_seed(F)
    M, N = 4000, 211
    src = torch.randn(M, F, generator=g).bfloat16()
    idx = torch.randint(0, N, (M,), generator=g)
    idx[:900] = 5                                    # list splitting
    ref = O.scatter_add(src.float(), idx, 0, N)      # fp32 accumulation of the bf16 values
    s = src.cuda().requires_grad_(True)
    out = H.scatter_add(s, idx.cuda(), dim=0, dim_size=N)
    assert out.dtype == torch.bfloat16
    o = out.detach().float().cpu()
    assert rel_err(o.numpy(), ref.numpy()) <= BF16_TOL
    # one rounding only: almost every element equals the correctly rounded fp32 sum
    assert float((o == ref.bfloat16().float()).float().mean()) > 0.98
    r = torch.randn(N, F, generator=g).bfloat16().cuda()
    (out * r).sum().backward()
    assert s.grad.dtype == torch.bfloat16
    assert torch.equal(s.grad, r[idx.cuda()])


def test_weighted_and_gathered_bf16():
    import hierarchicalgnn_amd as H
    from oracle import hgnn_oracle as O
    g = torch.Generator().manual_seed(3)
    N, S, B, F = 500, 31, 2600, 512
    X = torch.randn(N, F, generator=g).bfloat16()
    gi = torch.randint(0, N, (B,), generator=g)
    di = torch.randint(0, S, (B,), generator=g)
    w = torch.exp(0.3 * torch.randn(B, 1, generator=g))
    ref = O.scatter_add(w * X.float()[gi], di, 0, S)
    Xd = X.cuda().requires_grad_(True)
    wd = w.cuda().requires_grad_(True)
    out = H.gather_scale_scatter(Xd, gi.cuda(), di.cuda(), S, wd)
    assert out.dtype == torch.bfloat16
    assert rel_err(out.detach().float().cpu().numpy(), ref.numpy()) <= BF16_TOL
    r = torch.randn(S, F, generator=g).bfloat16()
    (out * r.cuda()).sum().backward()
    Xr = X.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    (O.scatter_add(wr * Xr[gi], di, 0, S) * r.float()).sum().backward()
    assert Xd.grad.dtype == torch.bfloat16 and wd.grad.dtype == torch.float32
    assert rel_err(Xd.grad.float().cpu().numpy(), Xr.grad.numpy()) <= 2 * BF16_TOL
    assert rel_err(wd.grad.cpu().numpy(), wr.grad.numpy()) <= 2 * BF16_TOL


def test_gather_rows_bf16_exact():
    import hierarchicalgnn_amd as H
    g = torch.Generator().manual_seed(4)
    t = torch.randn(300, 256, generator=g).bfloat16()
    idx = torch.randint(0, 300, (5000,), generator=g)
    out = H.gather_rows(t.cuda(), idx.cuda())
    assert torch.equal(out.cpu(), t[idx])


def test_bf16_unsupported_width_is_an_error():
    import hierarchicalgnn_amd as H
    with pytest.raises(RuntimeError, match="multiple of 8"):
        H.scatter_add(torch.zeros(10, 12).bfloat16().cuda(), torch.zeros(10, dtype=torch.long).cuda(), dim_size=2)


@pytest.mark.parametrize("L,layers,nseg,M,split", [
    (32, 2, 3, 700, False), (64, 2, 3, 513, False), (128, 2, 3, 300, False), (256, 2, 3, 200, False),
    (64, 3, 2, 400, False), (128, 3, 3, 129, False), (256, 3, 2, 90, False),
    # the feature-split kernel (wide layers; L = 512 is BASELINE config 4): ragged tails, 1..3 segments
    (128, 2, 3, 300, True), (256, 2, 3, 200, True), (512, 2, 3, 131, True), (128, 3, 3, 129, True),
    (256, 3, 2, 90, True), (512, 3, 2, 70, True), (256, 2, 1, 64, True), (256, 2, 3, 1, True),
    (256, 2, 3, 333, True), (256, 2, 2, 128, True),
    # more tiles than resident workgroups
    (256, 2, 3, 70001, True), (256, 3, 3, 40000, True),
    # pre-projected gathered segments in the bf16 split kernel (off by default: measured slower)
    (256, 2, 3, 2000, "preproject"), (128, 2, 3, 1500, "preproject")])
def test_fused_mlp_bf16_vs_oracle(L, layers, nseg, M, split):
    """bf16-MFMA fused MLP: against the fp32 oracle evaluated on the bf16-rounded inputs and weights"""
    from hierarchicalgnn_amd import _lib, fused, make_mlp
    fused._preproject_bf16 = split == "preproject"      # (forced on / off: the default decides by shape)
    split = bool(split)
    fused.set_bf16_split(split)
    from oracle import hgnn_oracle as O
    g = torch.Generator().manual_seed(L * 10 + layers)
    out_act = "Tanh" if layers == 2 else "GELU"
    torch.manual_seed(L + layers)
    net = make_mlp(nseg * L, 2 * L, L, layers, layer_norm=True, output_activation=out_act, hidden_activation="GELU")
    for p in net.parameters():
        if p.dim() == 1:
            p.data.add_(0.2 * torch.randn_like(p))
    n_tab = 97
    table = torch.randn(n_tab, L, generator=g).bfloat16()
    idx0 = torch.randint(0, n_tab, (M,), generator=g)
    idx1 = torch.randint(0, n_tab, (M,), generator=g)
    direct = torch.randn(M, L, generator=g).bfloat16()
    segs_cpu = [(table, idx0), (table, idx1), (direct, None)][3 - nseg:]
    x = torch.cat([t.float() if i is None else t.float()[i] for t, i in segs_cpu], dim=1)
    sd = {k: (v.detach().bfloat16().float() if k.endswith("weight") and v.dim() == 2 else v.detach())
          for k, v in net.state_dict().items()}
    ref = O.mlp_apply(sd, "", x, layers, "GELU", out_act, True) + direct.float()
    net = net.cuda()
    segs = [(t.cuda(), None if i is None else i.cuda()) for t, i in segs_cpu]
    try:
        with torch.no_grad():
            assert fused._wants_split(net, segs) == split
            assert fused.supported(net, segs, segs[-1][0])
            out = fused.fused_concat_mlp(net, segs, segs[-1][0])
    finally:
        fused.set_bf16_split(True)
        fused._preproject_bf16 = None
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape
    # hidden activations are rounded to bf16 between layers: a few bf16 ulps at the output scale
    assert rel_err(out.float().cpu().numpy(), ref.numpy()) <= 4 * BF16_TOL


@pytest.mark.parametrize("latent", [32, 128])
def test_interaction_cell_bf16_tracks_the_fp32_reference(latent):
    """whole InteractionGNNCell in bf16 (bf16 aggregation kernels + bf16-MFMA MLPs, fp32 master weights)
    against the reference's fp32 outputs: bf16-level agreement"""
    import numpy as np
    import hierarchicalgnn_amd as H
    from hierarchicalgnn_amd import fused
    from conftest import load_golden
    z = load_golden(f"ignn_cell_L{latent}.npz")
    hp = dict(latent=latent, hidden=2 * latent, nb_edge_layer=2, nb_node_layer=3, layernorm=True,
              hidden_activation="GELU")
    cell = H.InteractionGNNCell(hp)
    cell.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")})
    cell = cell.cuda()
    n0 = fused.stats["fused_calls"]
    with torch.no_grad():
        on, oe = cell(torch.from_numpy(z["nodes"]).cuda().bfloat16(), torch.from_numpy(z["edges"]).cuda().bfloat16(),
                      torch.from_numpy(z["graph"]).cuda())
    assert fused.stats["fused_calls"] == n0 + 2 and on.dtype == torch.bfloat16
    assert rel_err(on.float().cpu().numpy(), z["out_nodes"]) <= 6 * BF16_TOL
    assert rel_err(oe.float().cpu().numpy(), z["out_edges"]) <= 6 * BF16_TOL


def test_ec_in_forward_with_bf16_latents_tracks_the_reference_scores():
    """hparams["feature_dtype"] = "bf16" (BASELINE config 4 dtype at model level): fp32 encoders and head,
    bf16 latent rows through the 14 cells; scores stay close to the reference's fp32 scores"""
    import numpy as np
    from conftest import load_golden
    from hierarchicalgnn_amd.models import EC_InteractionGNN
    z = load_golden("ec_in_L32.npz")
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
    hp["feature_dtype"] = "bf16"
    model = EC_InteractionGNN(hp)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")})
    model = model.cuda().eval()
    x = torch.from_numpy(z["x"]).cuda()
    graph = torch.from_numpy(z["edge_index"]).cuda()
    with torch.no_grad():
        scores = model(x, graph)
    assert scores.dtype == torch.float32 and scores.shape == z["scores"].shape
    d = np.abs(scores.cpu().numpy() - z["scores"])
    assert d.mean() <= 0.01 and d.max() <= 0.1, (d.mean(), d.max())
    # training-mode call in bf16: autograd through the bf16 HIP ops + autocast library MLPs
    scores = model(x, graph)
    scores.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_bc_hgnn_block_with_bf16_latents_tracks_the_reference():
    """BC-HGNN-GMM message passing in bf16 on the hierarchy captured from the reference's forward"""
    import numpy as np
    from conftest import load_golden
    from hierarchicalgnn_amd.models import BC_MessagePassing
    z = load_golden("bc_hgnn_L32.npz")
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
    hp["feature_dtype"] = "bf16"
    model = BC_MessagePassing(hp)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}, strict=True)
    model = model.cuda().eval()
    x = torch.from_numpy(z["x"]).cuda()
    graph = torch.from_numpy(z["edge_index"]).cuda()
    t = lambda k: torch.from_numpy(z[k]).cuda()
    with torch.no_grad():
        directed, emb, nodes, edges, order = model.embed(x, graph)
        assert nodes.dtype == torch.bfloat16 and edges.dtype == torch.bfloat16 and emb.dtype == torch.float32
        assert rel_err(emb.cpu().numpy(), z["embeddings"]) <= 0.05
        means = t("cell0.in.supernodes")[:, :hp["emb_dim"]].contiguous()
        bg, bw = t("cell0.in.bipartite_graph"), t("cell0.in.bipartite_edge_weights")
        sg, sw = t("cell0.in.super_graph"), t("cell0.in.super_edge_weights")
        n_out, sn_out, _, _ = model.hgnn_block(nodes, edges, directed, means, bg, bw, sg, sw)
        assert n_out.dtype == torch.bfloat16 and sn_out.dtype == torch.bfloat16
        last = int(z["n_cells"]) - 1
        assert rel_err(n_out.float().cpu().numpy(), z[f"cell{last}.out.nodes"]) <= 0.1
        scores = model.score(n_out, sn_out, t("bipartite_graph"))
        d = np.abs(scores.cpu().numpy() - z["bipartite_scores"])
        assert d.mean() <= 0.02, (d.mean(), d.max())
        # and the whole forward (own hierarchy decision) runs
        bgr, s, e = model(x, graph)
        assert s.dtype == torch.float32 and torch.isfinite(s).all() and bgr.shape[1] == s.shape[0]


def test_latent512_bf16_mode_matches_its_fp32_model():
    """BASELINE config 4 widths (latent 512): encoders (fp32 first layer, bf16 tail), bf16 cells on the
    feature-split kernel, bf16 head -- against the same weights run in fp32"""
    from hierarchicalgnn_amd import fused, synth
    from hierarchicalgnn_amd.models import EC_InteractionGNN
    hp = dict(spatial_channels=3, latent=512, hidden=1024, n_interaction_graph_iters=2, nb_node_layer=3,
              nb_edge_layer=2, output_layers=3, hidden_output_activation="GELU", hidden_activation="GELU",
              layernorm=True, share_weight=False)
    torch.manual_seed(5)
    ref_model = EC_InteractionGNN(hp).cuda().eval()
    model = EC_InteractionGNN(dict(hp, feature_dtype="bf16")).cuda().eval()
    model.load_state_dict(ref_model.state_dict())
    x, ei = synth.trackml_event(1500, 9000, seed=3)
    x, ei = x.cuda(), ei.cuda()
    n0 = fused.stats["fused_calls"]
    with torch.no_grad():
        ref = ref_model(x, ei)
        n1 = fused.stats["fused_calls"]
        out = model(x, ei)
    assert fused.stats["fused_calls"] - n1 >= 4                   # 2 cells x (node + edge) fused in bf16
    assert n1 - n0 >= 2 * (2 + 3)                                   # fp32 at latent 512: one fused launch per layer (round 2)
    assert out.dtype == torch.float32 and out.shape == ref.shape
    d = (out - ref).abs()
    assert float(d.mean()) <= 0.01 and float(d.max()) <= 0.1, (float(d.mean()), float(d.max()))


@pytest.mark.parametrize("kind,L", [("head", 512), ("head", 256), ("mlp3", 512)])
def test_bf16_single_layer_chains(kind, L):
    """bf16 MLPs that do not fit one launch (heads with a plain last layer; 1024-wide layers at latent 512): one
    feature-split bf16-MFMA launch per [Linear, LayerNorm, act] layer (hgnn_mlp_forward_bf16_split, n_layers = 1)"""
    from hierarchicalgnn_amd import fused, make_mlp
    from oracle import hgnn_oracle as O
    g = torch.Generator().manual_seed(L + len(kind))
    M = 900
    torch.manual_seed(L)
    if kind == "head":
        net = make_mlp(2 * L, 2 * L, 1, 3, layer_norm=True, output_activation=None, hidden_activation="GELU")
        a = torch.randn(M, L, generator=g).bfloat16()
        tab = torch.randn(70, L, generator=g).bfloat16()
        idx = torch.randint(0, 70, (M,), generator=g)
        x = torch.cat([a.float(), tab.float()[idx]], 1)
        segs = [(a.cuda(), None), (tab.cuda(), idx.cuda())]
        ref = O.mlp_apply({k: v.detach() for k, v in net.state_dict().items()}, "", x, 3, "GELU", None, True)
        n_launch, skip = 2, None
    else:
        net = make_mlp(L, 2 * L, 2 * L, 3, layer_norm=True, output_activation="GELU", hidden_activation="GELU")
        a = torch.randn(M, L, generator=g).bfloat16()
        segs = [(a.cuda(), None)]
        ref = O.mlp_apply({k: v.detach() for k, v in net.state_dict().items()}, "", a.float(), 3, "GELU", "GELU", True)
        n_launch, skip = 3, None
    net = net.cuda()
    with torch.no_grad():
        assert fused.supported(net, segs, skip)
        n0 = fused.stats["fused_calls"]
        out = fused.fused_concat_mlp(net, segs, skip)
        assert fused.stats["fused_calls"] == n0 + n_launch
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape
    assert rel_err(out.float().cpu().numpy(), ref.numpy()) <= 3e-2          # three bf16 layers deep


@pytest.mark.parametrize("kind", ["node_enc", "edge_enc"])
def test_bf16_mode_encoders_at_latent512_hybrid_chain(kind):
    """config 4's encoders (IN.py:84-85 at latent 512, bf16 latent mode): first Linear as one fp32 fused layer (hit
    coordinates are not rounded), the wide tail on the bf16 feature-split kernel -- no library GEMM"""
    from hierarchicalgnn_amd import fused, make_mlp, mlp
    from oracle import hgnn_oracle as O
    L = 512
    g = torch.Generator().manual_seed(12)
    N, M = 300, 2000
    x = torch.rand(N, 3, generator=g) * 2 - 1
    i0 = torch.randint(0, N, (M,), generator=g)
    i1 = torch.randint(0, N, (M,), generator=g)
    torch.manual_seed(4)
    if kind == "edge_enc":
        net = make_mlp(6, 2 * L, L, 2, layer_norm=True, output_activation="GELU", hidden_activation="GELU")
        segs_cpu, layers, xin, launches = [(x, i0), (x, i1)], 2, torch.cat([x[i0], x[i1]], 1), 2
    else:
        net = make_mlp(3, 2 * L, L, 3, layer_norm=True, output_activation="GELU", hidden_activation="GELU")
        segs_cpu, layers, xin, launches = [(x, None)], 3, x, 2          # fp32 layer + one 2-layer bf16 launch
    ref = O.mlp_apply({k: v.detach() for k, v in net.state_dict().items()}, "", xin, layers, "GELU", "GELU", True)
    net = net.cuda()
    segs = [(t.cuda(), None if i is None else i.cuda()) for t, i in segs_cpu]
    with torch.no_grad():
        n0 = fused.stats["fused_calls"]
        out = mlp.concat_mlp(net, segs, bf16_tail=True)
        assert fused.stats["fused_calls"] - n0 == launches
    assert out.dtype == torch.bfloat16
    assert rel_err(out.float().cpu().numpy(), ref.numpy()) <= 2e-2
