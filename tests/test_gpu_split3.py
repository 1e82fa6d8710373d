"""Opt-in fast path of the fp32 MLPs (hgnn_mlp_forward_f32_split3): fp32 rows, GEMMs as split-bf16 products
(hi.hi + mid.hi + hi.mid on the bf16 matrix pipe, fp32 accumulation).  Bars: a single MLP within 2e-5 of an fp64
evaluation; BASELINE configs 2 and 3 at model level within north_star's 1e-4 of the REFERENCE's outputs (the same
fixtures and checks as the exact fp32 path, tests/test_gpu_configs.py)."""
import numpy as np
import pytest
import torch

from conftest import assert_parity, load_golden
import test_gpu_configs as C

pytestmark = pytest.mark.gpu


@pytest.fixture()
def split3():
    """no-grad forwards AND (opt-in) the training path -- differentiable forward, backward GEMMs -- on the split-bf16 kernels"""
    from hierarchicalgnn_amd import fused
    old, old_b = fused._fp32_split3, fused._fp32_split3_train
    fused.set_fp32_split3(True)
    fused.set_fp32_split3_training(True)
    yield fused
    fused.set_fp32_split3(old)
    fused.set_fp32_split3_training(old_b)


@pytest.mark.parametrize("L,layers,nseg,M", [(256, 2, 3, 1), (256, 2, 3, 333), (128, 2, 3, 1000), (256, 3, 2, 200),
                                             (128, 3, 3, 77), (256, 2, 1, 130), (128, 2, 2, 64),
                                             (256, 2, 3, 40000), (128, 2, 3, 50001)])   # the large ones pre-project
def test_split3_mlp_vs_fp64(split3, L, layers, nseg, M):
    from hierarchicalgnn_amd import make_mlp
    torch.manual_seed(L + 7 * layers + nseg)
    out_act = "Tanh" if layers == 2 else "GELU"
    net = make_mlp(nseg * L, 2 * L, L, layers, layer_norm=True, output_activation=out_act,
                   hidden_activation="GELU").cuda()
    for p in net.parameters():
        if p.dim() == 1:
            p.data.add_(0.2 * torch.randn_like(p))
    n_tab = max(8, M // 20)
    table = torch.randn(n_tab, L, device="cuda")
    i0 = torch.randint(0, n_tab, (M,), device="cuda")
    i1 = torch.randint(0, n_tab, (M,), device="cuda")
    direct = torch.randn(M, L, device="cuda")
    segs = [(table, i0), (table, i1), (direct, None)][3 - nseg:]
    with torch.no_grad():
        n0 = split3.stats.get("split3_calls", 0)
        assert split3.supported(net, segs, direct)
        out = split3.fused_concat_mlp(net, segs, direct)
        assert split3.stats.get("split3_calls", 0) == n0 + 1      # the split-bf16 kernel ran, not the fp32 one
        x = torch.cat([t.double() if i is None else t.double()[i] for t, i in segs], dim=1)
        ref = net.double()(x) + direct.double()
    assert out.dtype == torch.float32 and out.shape == ref.shape
    assert float((out.double() - ref).abs().max() / ref.abs().max()) <= 2e-5


@pytest.mark.parametrize("H,M", [(256, 1000), (512, 333)])
def test_split3_score_head_vs_fp64(split3, H, M):
    """K -> H -> H -> 1 with a plain last layer: hidden layers on the split-bf16 kernel, last Linear trailing"""
    from hierarchicalgnn_amd import make_mlp
    torch.manual_seed(H)
    L = H // 2
    net = make_mlp(2 * L, H, 1, 3, layer_norm=True, output_activation=None, hidden_activation="GELU").cuda()
    a, b = torch.randn(M, L, device="cuda"), torch.randn(M, L, device="cuda")
    with torch.no_grad():
        n0 = split3.stats.get("split3_calls", 0)
        out = split3.fused_concat_mlp(net, [(a, None), (b, None)], None)
        assert split3.stats.get("split3_calls", 0) == n0 + 1
        ref = net.double()(torch.cat([a, b], dim=1).double())
    assert out.shape == (M, 1)
    assert float((out.double() - ref).abs().max() / ref.abs().max()) <= 2e-5


def test_split3_config2_ec_in_latent128_within_the_parity_bar(split3):
    from hierarchicalgnn_amd.models import EC_InteractionGNN
    z = load_golden("ec_in_L128.npz")
    model = C._seeded(EC_InteractionGNN, C._cfg("EC-IN"), z)
    x = torch.from_numpy(z["x"]).cuda()
    graph = torch.from_numpy(z["edge_index"]).cuda()
    n0 = split3.stats.get("split3_calls", 0)
    with torch.inference_mode():
        scores = model(x, graph)
    assert split3.stats.get("split3_calls", 0) - n0 == 2 * 14 + 1  # every node / edge network of the 14 cells + the head
    assert np.abs(scores.cpu().numpy() - z["scores"]).max() <= C.TOL
    assert_parity(scores, z["scores"], C.TOL, "scores")


@pytest.mark.parametrize("M,K,N,cols", [(1, 256, 512, None), (333, 256, 512, None), (5000, 512, 256, (256, 512)),
                                        (70001, 512, 256, (0, 256)), (64, 128, 256, None)])
def test_split3_linear_data_gradient_vs_fp64(split3, M, K, N, cols):
    """dz . W[:, cols] (hgnn_linear_f32_split3): the M-row data-gradient GEMMs of the fp32 training backward"""
    g = torch.Generator(device="cuda").manual_seed(M + K)
    n_in = N if cols is None else 3 * N
    lin = torch.nn.Linear(n_in, K).cuda()                       # weight [K, n_in]
    dz = torch.randn(M, K, device="cuda", generator=g)
    n0 = split3.stats.get("split3_linear_calls", 0)
    out = split3._split3_linear(dz, lin.weight, cols, torch.nn.Sequential())
    assert out is not None and split3.stats.get("split3_linear_calls", 0) == n0 + 1
    W = lin.weight.detach().double()
    ref = dz.double() @ (W if cols is None else W[:, cols[0]:cols[1]])
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert float((out.double() - ref).abs().max() / ref.abs().max()) <= 2e-5


@pytest.mark.parametrize("M,Ho,Hi", [(1000, 64, 64), (4097, 512, 256), (33, 256, 512), (70000, 256, 128), (7, 8, 24), (0, 64, 64)])
def test_split3_weight_gradient_vs_fp64(M, Ho, Hi):
    """dz^T rows for fp32 operands (hgnn_wgrad_f32_split3): the M-row weight gradients of the fp32 training backward"""
    from hierarchicalgnn_amd.ops import wgrad_f32_split3
    g = torch.Generator(device="cuda").manual_seed(M + Ho)
    dz = torch.randn(M, Ho, device="cuda", generator=g)
    rows = torch.randn(M, Hi, device="cuda", generator=g)
    cs = torch.empty(Ho, device="cuda")
    out = wgrad_f32_split3(dz, rows, colsum=cs)
    ref = dz.double().t() @ rows.double()
    assert out.shape == (Ho, Hi) and out.dtype == torch.float32
    if M:
        assert float((out.double() - ref).abs().max() / ref.abs().max()) <= 2e-5
        assert float((cs.double() - dz.double().sum(0)).abs().max() / dz.double().sum(0).abs().max()) <= 2e-5
    else:
        assert float(out.abs().max()) == 0.0
    assert torch.equal(out, wgrad_f32_split3(dz, rows))                # deterministic


def test_split3_config2_training_step_gradients_within_the_bar(split3):
    """training mode (the reference's reentrant checkpointing): forward passes and pre-LayerNorm dumps on the split-bf16
    kernel, hand-written backward unchanged; input gradient and every weight-gradient sketch against the reference's"""
    from hierarchicalgnn_amd.models import EC_InteractionGNN
    z = load_golden("ec_in_L128.npz")
    model = C._seeded(EC_InteractionGNN, C._cfg("EC-IN"), z)
    x = torch.from_numpy(z["x"]).cuda().clone()
    graph = torch.from_numpy(z["edge_index"]).cuda()
    n0 = split3.stats.get("split3_calls", 0)
    scores = model(x, graph)
    assert x.requires_grad
    (scores * torch.from_numpy(z["r_scores"]).cuda()).sum().backward()
    assert split3.stats.get("split3_calls", 0) - n0 >= 2 * 2 * 14     # no-grad pass + recompute of every cell network
    assert split3.stats.get("split3_linear_calls", 0) > 0 and split3.stats.get("split3_wgrad_calls", 0) > 0
    assert np.abs(scores.detach().cpu().numpy() - z["scores"]).max() <= C.TOL
    assert_parity(x.grad, z["grad_x"], C.TOL, "d loss / d x")
    C._sketch_close(model, z)


def test_split3_is_the_process_default_and_hparams_pin_a_model():
    """default on (HGNN_FP32_SPLIT3 unset); hparams["fp32_gemm"] = "exact" / "split_bf16" pin one model either way"""
    import os
    from hierarchicalgnn_amd import fused
    from hierarchicalgnn_amd.models import EC_InteractionGNN
    if os.environ.get("HGNN_FP32_SPLIT3") is None:
        assert fused._fp32_split3 is True
    z = load_golden("ec_in_L128.npz")
    x = torch.from_numpy(z["x"]).cuda()
    graph = torch.from_numpy(z["edge_index"]).cuda()
    fast = C._seeded(EC_InteractionGNN, dict(C._cfg("EC-IN"), fp32_gemm="split_bf16"), z)
    exact = C._seeded(EC_InteractionGNN, dict(C._cfg("EC-IN"), fp32_gemm="exact"), z)
    with torch.inference_mode():
        n0 = fused.stats.get("split3_calls", 0)
        s_exact = exact(x, graph)
        assert fused.stats.get("split3_calls", 0) == n0
        s_fast = fast(x, graph)
        assert fused.stats.get("split3_calls", 0) == n0 + 2 * 14 + 1
    assert np.abs(s_fast.cpu().numpy() - z["scores"]).max() <= C.TOL
    assert float((s_fast - s_exact).abs().max()) > 0.0          # (they are different arithmetic)


def test_split3_config3_bc_hgnn_gmm_latent256_within_the_parity_bar(split3):
    from hierarchicalgnn_amd.models import BC_MessagePassing
    from hierarchicalgnn_amd.utils import process_hparams
    z = load_golden("bc_hgnn_L256.npz")
    raw = dict(C._cfg("BC-HGNN-GMM"), latent=256)
    model = C._seeded(BC_MessagePassing, raw, z).eval()
    n0 = split3.stats.get("split3_calls", 0)
    scores = C._bc_stages(model, z, process_hparams(raw), C.TOL, assert_parity)
    assert split3.stats.get("split3_calls", 0) > n0
    assert np.abs(scores.cpu().numpy() - z["bipartite_scores"]).max() <= C.TOL


def test_split3_forward_is_graph_capturable(split3):
    """the split-bf16 launches (weight streams cached after the warm-up, projections by library GEMMs) replay from a
    captured HIP graph: models.GraphedInference on a small latent-128 model, replay == eager"""
    from hierarchicalgnn_amd import synth
    from hierarchicalgnn_amd.models import EC_InteractionGNN, GraphedInference
    torch.manual_seed(0)
    hp = dict(spatial_channels=3, latent=128, hidden=256, n_interaction_graph_iters=3, nb_node_layer=3,
              nb_edge_layer=2, output_layers=3, hidden_output_activation="GELU", hidden_activation="GELU",
              layernorm=True, share_weight=False)
    model = EC_InteractionGNN(hp).cuda().eval()
    x, ei = synth.trackml_event(3000, 20000, seed=3)
    x, ei = x.cuda(), ei.cuda()
    with torch.no_grad():
        ref = model(x, ei)
        n0 = split3.stats.get("split3_calls", 0)
        g = GraphedInference(model, x, ei)
        assert split3.stats.get("split3_calls", 0) > n0
        assert torch.equal(g(x), ref) and torch.equal(g(x.clone()), ref)


def test_prepared_weight_cache_follows_weight_edits(split3):
    """The split-bf16 / bf16 kernels read PREPARED copies of the Linear weights (fused._WeightCache).  Every way a
    weight can change must reach the kernel: in-place ops on the parameter (version counter), ``.data`` edits
    followed by eval() / train() / to() / load_state_dict (FusedMLPSequential hooks), and ``.data`` edits followed
    by the explicit ``fused.clear_weight_cache()``; a deleted model must free its entries."""
    import gc
    from hierarchicalgnn_amd import make_mlp
    torch.manual_seed(5)
    L, M = 128, 500
    net = make_mlp(L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
    x = torch.randn(M, L, device="cuda")

    def run():
        with torch.no_grad():
            return split3.fused_concat_mlp(net, [(x, None)], None).clone()

    def ref():
        with torch.no_grad():
            return net(x)

    base = run()
    assert float((base - ref()).abs().max()) < 1e-4 and len(split3._wcache) >= 2
    with torch.no_grad():
        net[0].weight.normal_(0, 0.05)                       # in-place on the parameter: version counter
    a = run()
    assert float((a - ref()).abs().max()) < 1e-4 and float((a - base).abs().max()) > 1e-2
    net[0].weight.data.normal_(0, 0.05)                      # through .data: invisible to the version counter ...
    net.eval()                                               # ... dropped by the module hook
    b = run()
    assert float((b - ref()).abs().max()) < 1e-4 and float((b - a).abs().max()) > 1e-2
    net[3].weight.data.mul_(0.5)
    split3.clear_weight_cache(net)                           # ... or by the explicit call
    c = run()
    assert float((c - ref()).abs().max()) < 1e-4 and float((c - b).abs().max()) > 1e-3
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    sd["0.weight"] = sd["0.weight"] * 1.5
    torch.nn.ModuleDict({"n": net}).load_state_dict({"n." + k: v for k, v in sd.items()})   # through a parent
    d = run()
    assert float((d - ref()).abs().max()) < 1e-4 and float((d - c).abs().max()) > 1e-3
    n_before = len(split3._wcache)
    del net
    gc.collect()
    assert len(split3._wcache) <= n_before - 2               # weak references: the entries died with the model


# ----------------------------------------------------------------------------------------------- full-size A/B
AB_BOUND = 5e-5      # stated bound for |scores_split - scores_exact| at the sizes the numbers are quoted on


def _both_modes(fn):
    from hierarchicalgnn_amd import fused
    old = fused._fp32_split3
    out = {}
    try:
        for name, flag in (("exact", False), ("split", True)):
            fused.set_fp32_split3(flag)
            n0 = fused.stats.get("split3_calls", 0)
            with torch.inference_mode():
                out[name] = fn()
            ran = fused.stats.get("split3_calls", 0) - n0
            assert (ran > 0) == flag, (name, ran)
    finally:
        fused.set_fp32_split3(old)
    return out["exact"], out["split"]


def test_full_size_ab_ec_in_latent128_split_vs_exact():
    """config 2's model on the BASELINE event (N = 120k hits, E = 1M edges; the reference fixtures are 540-1,500
    hits): the shipped default (split-bf16 GEMMs) against the exact fp32 kernels, same weights, same event"""
    from hierarchicalgnn_amd import synth
    from hierarchicalgnn_amd.models import EC_InteractionGNN
    from golden import seeded
    model = EC_InteractionGNN(C._cfg("EC-IN"))
    seeded.fill_parameters(model, 7)
    model = model.cuda().eval()
    x, ei = synth.trackml_event(120_000, 1_000_000, seed=1234)
    x, ei = x.cuda(), ei.cuda()
    s_exact, s_split = _both_modes(lambda: model(x, ei).clone())
    d = (s_split - s_exact).abs()
    assert s_exact.shape == (1_000_000,) and float(s_exact.std()) > 1e-3
    assert float(d.max()) <= AB_BOUND, float(d.max())


def test_full_size_ab_bc_latent256_split_vs_exact():
    """config 3's model (latent 256, 6 + 6 cells) on the BASELINE event with a fixed synthetic hierarchy decision
    (phi-z cells; the kNN topology is taken from the exact run so that a tie cannot change the graphs between the two
    arms): embeddings, node / supernode latents and bipartite scores of the split-bf16 default against exact fp32"""
    from hierarchicalgnn_amd import synth
    from hierarchicalgnn_amd.models import BC_MessagePassing
    from golden import seeded
    model = BC_MessagePassing(C._cfg("BC-HGNN-GMM"))
    seeded.fill_parameters(model, 7)
    model = model.cuda().eval()
    model.hgnn_block.super_graph_construction.knn_radius.fill_(2.0)
    model.hgnn_block.bipartite_graph_construction.knn_radius.fill_(2.0)
    x, ei = synth.trackml_event(120_000, 1_000_000, seed=1234)
    x, ei = x.cuda(), ei.cuda()
    cl = ((x[:, 1] + 1) * 50).long().clamp(0, 99) * 100 + ((x[:, 2] + 1) * 50).long().clamp(0, 99)
    _, clusters = torch.unique(cl, return_inverse=True)
    n_cl = int(clusters.max()) + 1
    graphs = {}

    def forward():
        directed, emb, nodes, edges, _ = model.embed(x, ei)
        if not graphs:
            _, bg, _, sg, _, _ = model.hgnn_block.hierarchy_from_clusters(emb, clusters, n_cl)
            graphs["g"] = (bg, sg)
        means, bg, bw, sg, sw, _ = model.hgnn_block.hierarchy_from_clusters(emb, clusters, n_cl, graphs=graphs["g"])
        n_out, sn_out, _, _ = model.hgnn_block(nodes, edges, directed, means, bg, bw, sg, sw)
        return emb.clone(), n_out.clone(), sn_out.clone(), model.score(n_out, sn_out, bg).clone()

    exact, split = _both_modes(forward)
    for a, b, what in zip(split[:3], exact[:3], ("embeddings", "nodes", "supernodes")):
        assert_parity(a, b, 1e-4, what + " (split vs exact, full size)")
    d = (split[3] - exact[3]).abs()
    assert float(exact[3].std()) > 1e-3
    assert float(d.max()) <= AB_BOUND, float(d.max())


def test_checkpointed_training_forward_is_bitwise_deterministic_and_exact_by_default():
    """The split-bf16 evaluation is an INFERENCE default.  Inside a training step (the no-grad first pass of a reentrant
    checkpoint segment and its recompute under autograd) the default arithmetic is the exact fp32 kernels in BOTH
    passes: the cell's training-mode outputs equal, bit for bit, the outputs of the exact kernels, and no split-bf16
    launch happens -- while a plain no-grad call of the same cell does use split-bf16."""
    import hierarchicalgnn_amd as H
    from hierarchicalgnn_amd import fused, synth
    assert not fused._fp32_split3_train
    old = fused._fp32_split3
    fused.set_fp32_split3(True)
    try:
        torch.manual_seed(2)
        L = 128
        hp = dict(latent=L, hidden=2 * L, nb_edge_layer=2, nb_node_layer=3, layernorm=True, hidden_activation="GELU")
        cell = H.InteractionGNNCell(hp).cuda()
        x, ei = synth.trackml_event(3000, 20000, seed=4)
        graph = synth.directed(ei).cuda()
        nodes = torch.randn(3000, L, device="cuda")
        edges = torch.randn(graph.shape[1], L, device="cuda")
        n0 = fused.stats.get("split3_calls", 0)
        n_g, e_g = nodes.clone().requires_grad_(True), edges.clone().requires_grad_(True)
        on, oe = cell(n_g, e_g, graph)                       # training mode: two checkpointed segments
        (on.sum() + oe.sum()).backward()
        assert fused.stats.get("split3_calls", 0) == n0      # no split-bf16 launch inside the training step
        with torch.no_grad():
            sn, se = cell(nodes, edges, graph)               # inference: the split-bf16 default
            assert fused.stats.get("split3_calls", 0) > n0
            fused.set_fp32_split3(False)
            xn, xe = cell(nodes, edges, graph)               # inference on the exact kernels
        assert torch.equal(on.detach(), xn) and torch.equal(oe.detach(), xe)
        assert not torch.equal(se, xe) and float((se - xe).abs().max()) < 1e-4
    finally:
        fused.set_fp32_split3(old)


def test_dispatch_options_are_context_local():
    """``fused.options(...)`` overrides a dispatch switch for the current context (thread / task) only: the same MLP
    runs the exact fp32 kernel inside the block and the split-bf16 default outside it; another thread started inside
    the block still sees the process default"""
    import threading
    from hierarchicalgnn_amd import fused, make_mlp
    old = fused._fp32_split3
    fused.set_fp32_split3(True)
    try:
        torch.manual_seed(9)
        L, M = 128, 300
        net = make_mlp(L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
        x = torch.randn(M, L, device="cuda")
        with torch.no_grad():
            n0 = fused.stats.get("split3_calls", 0)
            a = fused.fused_concat_mlp(net, [(x, None)], None)
            assert fused.stats.get("split3_calls", 0) == n0 + 1
            seen = []
            with fused.options(fp32_split3=False):
                b = fused.fused_concat_mlp(net, [(x, None)], None)
                assert fused.stats.get("split3_calls", 0) == n0 + 1          # the exact kernel ran
                t = threading.Thread(target=lambda: seen.append(fused._opt("fp32_split3")))
                t.start()
                t.join()
                with pytest.raises(TypeError):
                    with fused.options(no_such_switch=1):
                        pass
            assert seen == [True]                                            # other contexts keep the default
            c = fused.fused_concat_mlp(net, [(x, None)], None)
            assert fused.stats.get("split3_calls", 0) == n0 + 2
        assert torch.equal(a, c) and not torch.equal(a, b) and float((a - b).abs().max()) < 1e-4
    finally:
        fused.set_fp32_split3(old)


@pytest.mark.parametrize("nseg,M,hidden_act,out_act,skip", [
    (3, 65536, "GELU", "Tanh", True),      # the edge update: two projected node segments + the edge rows
    (3, 66001, "GELU", "Tanh", True),      # ragged last tile (66001 = 515 tiles of 128 + 81 rows)
    (2, 70000, "GELU", "GELU", True),      # the node update's shape
    (1, 65900, "Tanh", "Tanh", True),      # one direct segment: no projected rows at all
    (3, 68000, "ReLU", "ReLU", True),      # activations without a compiled pair: the run-time switch
    (3, 67000, "GELU", "Tanh", False),     # no skip connection
])
def test_split3_rows128_kernel_variants_vs_fp64_and_vs_the_64_row_kernel(split3, nseg, M, hidden_act, out_act, skip):
    """K -> 512 -> 256 at M >= 65,536 runs the 128-row-tile kernel (r128::k_mlp_f32_split3_khalf); below, or with
    hgnn_set_option("mlp_split3_rows128", 0), the 64-row one.  Same arithmetic (bf16 split-3 products, fp32 accumulation and
    LayerNorm), different summation order in the row statistics: both within 2e-5 of fp64, and within 2e-5 (absolute) of
    each other."""
    from hierarchicalgnn_amd import _lib, make_mlp
    L = 256
    torch.manual_seed(nseg * 1000 + M)
    net = make_mlp(nseg * L, 2 * L, L, 2, layer_norm=True, output_activation=out_act, hidden_activation=hidden_act).cuda()
    for p in net.parameters():
        if p.dim() == 1:
            p.data.add_(0.2 * torch.randn_like(p))
    n_tab = M // 17
    table = torch.randn(n_tab, L, device="cuda")
    i0 = torch.randint(0, n_tab, (M,), device="cuda")
    i1 = torch.sort(torch.randint(0, n_tab, (M,), device="cuda")).values
    direct = torch.randn(M, L, device="cuda")
    segs = [(table, i0), (table, i1), (direct, None)][3 - nseg:]
    sk = direct if skip else None
    lib = _lib.load()
    with torch.no_grad():
        assert split3.supported(net, segs, sk)
        out128 = split3.fused_concat_mlp(net, segs, sk)
        try:
            _lib.check(lib.hgnn_set_option(b"mlp_split3_rows128", 0))
            out64 = split3.fused_concat_mlp(net, segs, sk)
        finally:
            _lib.check(lib.hgnn_set_option(b"mlp_split3_rows128", 1))
        x = torch.cat([t.double() if i is None else t.double()[i] for t, i in segs], dim=1)
        ref = net.double()(x) + (direct.double() if skip else 0)
        net.float()
    scale = float(ref.abs().max())
    # without the skip rows the output is the bare tanh (|out| <= 1, scale 1 instead of ~5): the same absolute error is a
    # larger fraction of it
    bar = 2e-5 if skip else 5e-5
    assert float((out128.double() - ref).abs().max()) / scale <= bar
    assert float((out64.double() - ref).abs().max()) / scale <= bar
    assert float((out128 - out64).abs().max()) <= 2e-5       # (absolute: a few fp32 roundings of the row statistics)
    assert not torch.equal(out128[-1], torch.zeros_like(out128[-1]))


def test_split3_rows128_training_forward_dumps_match_the_64_row_kernel(split3):
    """save_pre (the pre-LayerNorm dumps the backward recomputes from) out of the 128-row kernel: gradients of an edge-update
    MLP at M = 66,000 against fp64 autograd."""
    from hierarchicalgnn_amd import make_mlp
    L, M = 256, 66000
    torch.manual_seed(5)
    net = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
    n_tab = 2000
    table = torch.randn(n_tab, L, device="cuda", requires_grad=True)
    i0 = torch.randint(0, n_tab, (M,), device="cuda")
    i1 = torch.sort(torch.randint(0, n_tab, (M,), device="cuda")).values
    direct = torch.randn(M, L, device="cuda", requires_grad=True)
    gout = torch.randn(M, L, device="cuda")
    segs = [(table, i0), (table, i1), (direct, None)]
    assert split3.supported_train(net, segs, direct)
    out = split3.fused_concat_mlp_train(net, segs, direct)
    out.backward(gout)
    got = [table.grad.clone(), direct.grad.clone()] + [p.grad.clone() for p in net.parameters()]
    table.grad = direct.grad = None
    net.zero_grad()
    net64 = net.double()
    t64, d64 = table.detach().double().requires_grad_(), direct.detach().double().requires_grad_()
    ref = net64(torch.cat([t64[i0], t64[i1], d64], dim=1)) + d64
    ref.backward(gout.double())
    want = [t64.grad, d64.grad] + [p.grad for p in net64.parameters()]
    assert float((out.double() - ref).abs().max() / ref.abs().max()) <= 2e-5
    for g, w in zip(got, want):
        assert float((g.double() - w).abs().max() / w.abs().max()) <= 5e-5


@pytest.mark.parametrize("L,layers", [(128, 2), (128, 3)])
def test_split3_two_workgroups_per_cu_are_bitwise_the_one_workgroup_result(split3, L, layers):
    """The latent-128 kernels run TWO persistent workgroups per CU (74 KB of LDS each).  A 64-row / 4-wave variant of the
    latent-256 kernel built the same way produced wrong elements (lanes 48-63 of single registers in the activation phases)
    whenever two workgroups shared a CU -- in one of two otherwise equivalent builds, cause not found --
    and is therefore not shipped.  This guards the shipped two-workgroup kernels against the same fault: rows are
    independent, so the result with one workgroup per CU (hgnn_set_option("mlp_split3_one_wg", 1)) must be BITWISE the
    result with two."""
    from hierarchicalgnn_amd import _lib, make_mlp
    M = 300_007
    torch.manual_seed(L + layers)
    out_act = "Tanh" if layers == 2 else "GELU"
    net = make_mlp(3 * L, 2 * L, L, layers, layer_norm=True, output_activation=out_act, hidden_activation="GELU").cuda()
    n_tab = M // 17
    table = torch.randn(n_tab, L, device="cuda")
    i0 = torch.randint(0, n_tab, (M,), device="cuda")
    i1 = torch.sort(torch.randint(0, n_tab, (M,), device="cuda")).values
    direct = torch.randn(M, L, device="cuda")
    segs = [(table, i0), (table, i1), (direct, None)]
    lib = _lib.load()
    with torch.no_grad():
        two = [split3.fused_concat_mlp(net, segs, direct) for _ in range(3)]
        try:
            _lib.check(lib.hgnn_set_option(b"mlp_split3_one_wg", 1))
            one = split3.fused_concat_mlp(net, segs, direct)
        finally:
            _lib.check(lib.hgnn_set_option(b"mlp_split3_one_wg", 0))
    for t in two:
        assert torch.equal(t, one)


@pytest.mark.parametrize("dtype,L,layers,M,split", [
    ("f32", 128, 2, 200_003, True),     # latent-128 edge update, two workgroups per CU
    ("f32", 128, 3, 120_000, True),     # latent-128 node update (three layers)
    ("f32", 256, 2, 200_003, True),     # latent-256 edge update on 128-row tiles
    ("f32", 256, 2, 40_000, True),      # ... on 64-row tiles (below two tiles per CU)
    ("f32", 256, 2, 100_000, False),    # exact fp32 MFMA kernel
    ("bf16", 256, 2, 200_003, False),   # bf16 rows, feature-split kernel
    ("bf16", 512, 2, 100_000, False),
])
def test_mlp_kernels_are_bitwise_repeatable(dtype, L, layers, M, split):
    """No atomics, no cross-workgroup reduction, fixed summation orders: every fused MLP kernel must return the SAME bits on
    every run.  (The experimental two-workgroup tile of DESIGN.md section 3 (8) fails exactly this; a data race or a
    scheduling-dependent hazard in a shipped kernel would show here as run-to-run differences.)"""
    from hierarchicalgnn_amd import fused, make_mlp
    old = fused._fp32_split3
    fused.set_fp32_split3(split)
    try:
        torch.manual_seed(L + layers + M)
        out_act = "Tanh" if layers == 2 else "GELU"
        net = make_mlp(3 * L, 2 * L, L, layers, layer_norm=True, output_activation=out_act, hidden_activation="GELU").cuda()
        n_tab = M // 17
        table = torch.randn(n_tab, L, device="cuda")
        direct = torch.randn(M, L, device="cuda")
        if dtype == "bf16":
            table, direct = table.bfloat16(), direct.bfloat16()
        i0 = torch.randint(0, n_tab, (M,), device="cuda")
        i1 = torch.sort(torch.randint(0, n_tab, (M,), device="cuda")).values
        segs = [(table, i0), (table, i1), (direct, None)]
        with torch.no_grad():
            assert fused.supported(net, segs, direct)
            first = fused.fused_concat_mlp(net, segs, direct)
            for _ in range(4):
                assert torch.equal(fused.fused_concat_mlp(net, segs, direct), first)
        assert bool(torch.isfinite(first.float()).all())
    finally:
        fused.set_fp32_split3(old)
