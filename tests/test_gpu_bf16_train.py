"""bf16 training path (BASELINE config 4 dtype): hand-written bf16-MFMA weight gradient, bf16 LayerNorm /
activation row kernels, and the differentiable bf16 fused MLP against fp32 references."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,Ho,Hi", [(1000, 64, 64), (4097, 512, 256), (33, 256, 512), (20000, 128, 384),
                                      (7, 8, 24), (65536, 1024, 512), (0, 64, 64)])
def test_wgrad_bf16_vs_fp32_matmul(M, Ho, Hi):
    """dW = dz^T rows: bf16 operands are exact in fp32, products are exact, only the fp32 summation order
    differs from the reference -> relative error at the 1e-5 level"""
    from hierarchicalgnn_amd.ops import wgrad_bf16
    g = torch.Generator(device="cuda").manual_seed(M + Ho)
    dz = torch.randn(M, Ho, device="cuda", generator=g).bfloat16()
    rows = torch.randn(M, Hi, device="cuda", generator=g).bfloat16()
    out = wgrad_bf16(dz, rows)
    ref = dz.float().t() @ rows.float()
    assert out.shape == (Ho, Hi) and out.dtype == torch.float32
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) <= 2e-5 if M else float(out.abs().max()) == 0.0
    assert torch.equal(out, wgrad_bf16(dz, rows))                      # deterministic


def test_wgrad_bf16_column_slices():
    """first-layer use: the input rows are a column slice of a concat-free segment and the result goes into a
    column slice of dW"""
    from hierarchicalgnn_amd.ops import wgrad_bf16
    g = torch.Generator(device="cuda").manual_seed(3)
    M = 5000
    dz = torch.randn(M, 256, device="cuda", generator=g).bfloat16()
    wide = torch.randn(M, 384, device="cuda", generator=g).bfloat16()
    dW = torch.zeros(256, 384, device="cuda")
    wgrad_bf16(dz, wide[:, 128:256], out=dW[:, 128:256])
    ref = dz.float().t() @ wide[:, 128:256].float()
    assert rel_err(dW[:, 128:256].cpu().numpy(), ref.cpu().numpy()) <= 2e-5
    assert float(dW[:, :128].abs().max()) == 0.0 and float(dW[:, 256:].abs().max()) == 0.0


@pytest.mark.parametrize("W,act", [(128, 1), (256, 2), (512, 1), (64, 3), (1024, 1), (1024, 2)])
def test_ln_act_bf16_row_kernels(W, act):
    """bf16 rows, fp32 arithmetic: against fp32 autograd on the same (bf16-exact) inputs; the only differences
    are the final roundings to bf16 (2^-9 relative)"""
    from hierarchicalgnn_amd import fused
    g = torch.Generator(device="cuda").manual_seed(W + act)
    M = 1003
    z = torch.randn(M, W, device="cuda", generator=g).bfloat16()
    da = torch.randn(M, W, device="cuda", generator=g).bfloat16()
    gamma = (1 + 0.2 * torch.randn(W, device="cuda", generator=g))
    beta = 0.2 * torch.randn(W, device="cuda", generator=g)
    acts = {1: torch.nn.functional.gelu, 2: torch.tanh, 3: torch.relu}
    zf = z.float().requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = acts[act](torch.nn.functional.layer_norm(zf, [W], gm, bt, 1e-5))
    ref.backward(da.float())
    out = fused._ln_act_forward(z, gamma, beta, act, 1e-5)
    assert out.dtype == torch.bfloat16
    assert rel_err(out.float().cpu().numpy(), ref.detach().cpu().numpy()) <= 4e-3
    dz, dg, db, dbias = fused._ln_act_backward(z, da, gamma, beta, act, 1e-5)
    assert dz.dtype == torch.bfloat16
    assert rel_err(dz.float().cpu().numpy(), zf.grad.cpu().numpy()) <= 4e-3
    assert rel_err(dg.cpu().numpy(), gm.grad.cpu().numpy()) <= 1e-4        # column sums are fp32 throughout
    assert rel_err(db.cpu().numpy(), bt.grad.cpu().numpy()) <= 1e-4
    assert rel_err(dbias.cpu().numpy(), zf.grad.sum(0).cpu().numpy()) <= 2e-3   # sums the ROUNDED dz rows


def _net(in_w, L, layers, out_act, seed):
    from hierarchicalgnn_amd import make_mlp
    torch.manual_seed(seed)
    net = make_mlp(in_w, 2 * L, L, layers, layer_norm=True, output_activation=out_act, hidden_activation="GELU")
    for p in net.parameters():
        if p.dim() == 1:
            p.data.add_(0.2 * torch.randn_like(p))
    return net


# stated bf16 bounds of the differentiable bf16 MLP against fp32 autograd on the same weights / inputs: every
# stored row (inputs, z_l, dz_l, gathered sums) carries one bf16 rounding (unit round-off 2^-9 = 2e-3), the
# GEMMs accumulate in fp32; two or three layers deep this stays within 2e-2 normwise for outputs and data
# gradients and 3e-2 for weight gradients.
BF16_OUT, BF16_GRAD = 2e-2, 3e-2


@pytest.mark.parametrize("L,layers", [(128, 2), (256, 2), (128, 3), (256, 3), (512, 2), (512, 3)])
def test_fused_train_bf16_matches_fp32_autograd(L, layers):
    from hierarchicalgnn_amd import _lib, fused, mlp
    g = torch.Generator().manual_seed(L + layers)
    out_act = "Tanh" if layers == 2 else "GELU"
    net = _net(3 * L, L, layers, out_act, L).cuda()
    n_tab, M = 211, 3000
    table = torch.randn(n_tab, L, generator=g).cuda().bfloat16()
    idx0 = torch.randint(0, n_tab, (M,), generator=g).cuda()
    idx1 = torch.randint(0, n_tab, (M,), generator=g).cuda()
    direct = torch.randn(M, L, generator=g).cuda().bfloat16()
    r = torch.randn(M, L, generator=g).cuda()

    def run(bf16):
        for p in net.parameters():
            p.grad = None
        t = (table if bf16 else table.float()).detach().clone().requires_grad_(True)
        d = (direct if bf16 else direct.float()).detach().clone().requires_grad_(True)
        if bf16:
            n0 = fused.stats["fused_train_calls"]
            out = mlp.concat_mlp(net, [(t, idx0), (t, idx1), (d, None)], skip=d)
            assert fused.stats["fused_train_calls"] == n0 + 1 and out.dtype == torch.bfloat16
        else:
            x = torch.cat([t[idx0], t[idx1], d], dim=1)
            out = net(x) + d
        (out.float() * r).sum().backward()
        return out.detach().float(), t.grad.float(), d.grad.float(), [p.grad.clone() for p in net.parameters()]

    o_ref, gt_ref, gd_ref, gp_ref = run(False)
    o, gt, gd, gp = run(True)
    assert rel_err(o.cpu().numpy(), o_ref.cpu().numpy()) <= BF16_OUT
    assert rel_err(gt.cpu().numpy(), gt_ref.cpu().numpy()) <= BF16_GRAD
    assert rel_err(gd.cpu().numpy(), gd_ref.cpu().numpy()) <= BF16_GRAD
    for (name, _), a, b in zip(net.named_parameters(), gp, gp_ref):
        assert a.dtype == torch.float32                                   # fp32 master weights get fp32 gradients
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= BF16_GRAD, name


@pytest.mark.parametrize("bf16", [True, False])
def test_fused_train_on_zero_rows(bf16):
    """M == 0 (a shard without edges, an empty super graph): forward returns [0, L], backward returns zero parameter
    and table gradients without launching anything (the bf16 backward used to reject its own NULL dumps)"""
    from hierarchicalgnn_amd import fused, mlp
    L = 128
    net = _net(3 * L, L, 2, "Tanh", L).cuda()
    dt = torch.bfloat16 if bf16 else torch.float32
    table = torch.randn(50, L, device="cuda").to(dt).requires_grad_(True)
    idx = torch.zeros(0, dtype=torch.long, device="cuda")
    direct = torch.zeros(0, L, device="cuda", dtype=dt, requires_grad=True)
    n0 = fused.stats["fused_train_calls"]
    out = mlp.concat_mlp(net, [(table, idx), (table, idx), (direct, None)], skip=direct)
    assert fused.stats["fused_train_calls"] == n0 + 1 and tuple(out.shape) == (0, L)
    out.float().sum().backward()
    assert table.grad is not None and float(table.grad.abs().max()) == 0.0 and tuple(direct.grad.shape) == (0, L)
    for p in net.parameters():
        assert p.grad is not None and float(p.grad.abs().max()) == 0.0


@pytest.mark.parametrize("ckpt", [True, False])
def test_interaction_cell_bf16_training_against_the_fp32_golden_gradients(ckpt):
    """config 4's dtype through a whole cell at latent 128: the REFERENCE's fp32 outputs and gradients
    (ignn_cell_L128.npz, produced by the reference's own InteractionGNNCell) within the stated bf16 bounds"""
    import hierarchicalgnn_amd as H
    from hierarchicalgnn_amd import fused
    z = load_golden("ignn_cell_L128.npz")
    hp = dict(latent=128, hidden=256, nb_edge_layer=2, nb_node_layer=3, layernorm=True, hidden_activation="GELU",
              checkpointing=ckpt)
    cell = H.InteractionGNNCell(hp)
    cell.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}, strict=True)
    cell = cell.cuda()
    nodes = torch.from_numpy(z["nodes"]).cuda().bfloat16().requires_grad_(True)
    edges = torch.from_numpy(z["edges"]).cuda().bfloat16().requires_grad_(True)
    graph = torch.from_numpy(z["graph"]).cuda()
    n0 = fused.stats["fused_train_calls"]
    on, oe = cell(nodes, edges, graph)
    assert on.dtype == torch.bfloat16 and oe.dtype == torch.bfloat16
    ((on.float() * torch.from_numpy(z["r_nodes"]).cuda()).sum() + (oe.float() * torch.from_numpy(z["r_edges"]).cuda()).sum()).backward()
    assert fused.stats["fused_train_calls"] - n0 == 2                     # node + edge network on the bf16 train path
    assert rel_err(on.detach().float().cpu().numpy(), z["out_nodes"]) <= BF16_OUT
    assert rel_err(oe.detach().float().cpu().numpy(), z["out_edges"]) <= BF16_OUT
    assert rel_err(nodes.grad.float().cpu().numpy(), z["grad_nodes"]) <= BF16_GRAD
    assert rel_err(edges.grad.float().cpu().numpy(), z["grad_edges"]) <= BF16_GRAD
    for k, p in cell.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), z["grad." + k]) <= BF16_GRAD, k


@pytest.mark.parametrize("M,K,N,act", [(1000, 256, 512, 1), (777, 512, 256, 1), (130, 128, 256, 2), (64, 256, 128, 3),
                                        (5000, 512, 512, 1), (1, 128, 128, 1)])
def test_fused_backward_layer_ln_form(M, K, N, act):
    """hgnn_mlp_backward_layer_bf16, LayerNorm form: dz' = dLN(act'(LN(z')) * (dz W)), a' = act(LN(z')), dgamma,
    dbeta -- against fp32 autograd on the same bf16-exact inputs"""
    from hierarchicalgnn_amd import fused
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    dz = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(K, N, device="cuda", generator=g) / K ** 0.5)
    z = (1.5 * torch.randn(M, N, device="cuda", generator=g) + 0.2).bfloat16()
    gamma = 1 + 0.2 * torch.randn(N, device="cuda", generator=g)
    beta = 0.2 * torch.randn(N, device="cuda", generator=g)
    acts = {1: torch.nn.functional.gelu, 2: torch.tanh, 3: torch.relu}
    zf = z.float().requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a_ref = acts[act](torch.nn.functional.layer_norm(zf, [N], gm, bt, 1e-5))
    da = dz.float() @ W.bfloat16().float()
    a_ref.backward(da)
    dzp, a_prev, dg, db = fused._bwd_layer(dz, W, z, gamma, beta, act, 1e-5, want_a=True)
    assert dzp.dtype == torch.bfloat16 and dzp.shape == (M, N)
    assert rel_err(a_prev.float().cpu().numpy(), a_ref.detach().cpu().numpy()) <= 4e-3
    assert rel_err(dzp.float().cpu().numpy(), zf.grad.cpu().numpy()) <= 5e-3
    assert rel_err(dg.cpu().numpy(), gm.grad.cpu().numpy()) <= 1e-4          # fp32 throughout
    assert rel_err(db.cpu().numpy(), bt.grad.cpu().numpy()) <= 1e-4
    again = fused._bwd_layer(dz, W, z, gamma, beta, act, 1e-5, want_a=True)
    assert torch.equal(again[0], dzp) and torch.equal(again[2], dg)           # deterministic


@pytest.mark.parametrize("M,K,N,with_skip", [(1000, 512, 256, True), (333, 256, 128, False), (4097, 1024, 512, True)])
def test_fused_backward_layer_input_form(M, K, N, with_skip):
    """input form: dx = dz W (+ skip) -- the gradient of a direct first-layer segment with the skip connection's
    gradient folded into the epilogue"""
    from hierarchicalgnn_amd import fused
    g = torch.Generator(device="cuda").manual_seed(M + K)
    dz = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    Wfull = torch.randn(K, 3 * N, device="cuda", generator=g) / K ** 0.5
    W = Wfull[:, N:2 * N]                                                  # a column slice, as for a segment
    skip = torch.randn(M, N, device="cuda", generator=g).bfloat16() if with_skip else None
    out = fused._bwd_layer(dz, W, None, None, None, 0, 1e-5, skip=skip)[0]
    ref = dz.float() @ W.bfloat16().float()
    if with_skip:
        ref = ref + skip.float()
    assert out.dtype == torch.bfloat16
    assert rel_err(out.float().cpu().numpy(), ref.cpu().numpy()) <= 4e-3


def test_wgrad_bf16_column_sums():
    from hierarchicalgnn_amd.ops import wgrad_bf16
    g = torch.Generator(device="cuda").manual_seed(8)
    for M, Ho, Hi in ((5000, 512, 256), (700, 256, 512), (129, 128, 128)):
        dz = torch.randn(M, Ho, device="cuda", generator=g).bfloat16()
        rows = torch.randn(M, Hi, device="cuda", generator=g).bfloat16()
        cs = torch.empty(Ho, device="cuda")
        out = wgrad_bf16(dz, rows, colsum=cs)
        assert rel_err(out.cpu().numpy(), (dz.float().t() @ rows.float()).cpu().numpy()) <= 2e-5
        assert rel_err(cs.cpu().numpy(), dz.float().sum(0).cpu().numpy()) <= 2e-5
