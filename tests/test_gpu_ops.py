"""GPU parity tests of the operator-level boundary (K1..K6) through the C ABI.

Bar: <= 1e-4 rel (fp32) against the CPU oracle / golden fixtures, as
BASELINE.json states; the kernels are expected to land ~1e-6.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4  # BASELINE.json north_star: within 1e-4 rel fp32


@pytest.fixture(scope="module")
def H():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    import hierarchicalgnn_amd as H
    from hierarchicalgnn_amd import _lib
    _lib.load()  # fail loudly if the HIP library is not built
    return H


@pytest.fixture(scope="module")
def O():
    from oracle import hgnn_oracle
    return hgnn_oracle


def dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    return t.cuda()


# ------------------------------------------------------------------ K1
def _k1_cases():
    z = load_golden("k1_scatter_add.npz")
    return z, sorted({k.split(".")[0] for k in z.files})


@pytest.mark.parametrize("chunk", [0, 5])
def test_k1_golden(H, chunk):
    z, cases = _k1_cases()
    for c in cases:
        src, index, n = dev(z[c + ".src"]), dev(z[c + ".index"]), int(z[c + ".dim_size"])
        plan = H.GraphPlan(index, n, chunk=chunk)
        out = H.scatter_add(src, index, dim=0, dim_size=n, plan=plan)
        assert out.shape == z[c + ".out"].shape, c
        assert rel_err(out.cpu().numpy(), z[c + ".out"]) <= TOL, (c, chunk)


def test_k1_default_dim_size_and_cache(H):
    z, _ = _k1_cases()
    src, index = dev(z["random_L32.src"]), dev(z["random_L32.index"])
    H.clear_plan_cache()
    a = H.scatter_add(src, index)
    b = H.scatter_add(src, index, dim=0, dim_size=int(index.max()) + 1)
    assert torch.equal(a, b)
    st = H.plan_cache_stats()
    assert st["hits"] >= 1
    # an in-place edit of the index must invalidate the cached plan
    index2 = index.clone()
    c = H.scatter_add(src, index2, dim_size=64)
    index2[:10] = 0
    d = H.scatter_add(src, index2, dim_size=64)
    ref = torch.zeros(64, 32).index_add_(0, index2.cpu(), src.cpu())
    assert rel_err(d.cpu().numpy(), ref.numpy()) <= TOL
    assert not torch.equal(c, d)


@pytest.mark.parametrize("F", [1, 3, 4, 8, 24, 32, 64, 100, 128, 248, 256, 260, 512, 1024, 1100])
def test_k1_random_widths_vs_oracle(H, O, F):
    g = torch.Generator().manual_seed(F)
    M, N = 3000, 157
    src = torch.randn(M, F, generator=g)
    index = torch.randint(0, N, (M,), generator=g)
    index[:700] = 3  # one heavy destination: list splitting at the default chunk (32)
    ref = O.scatter_add(src, index, dim=0, dim_size=N)
    out = H.scatter_add(src.cuda(), index.cuda(), dim=0, dim_size=N)
    assert rel_err(out.cpu().numpy(), ref.numpy()) <= TOL


def test_k1_empty_and_degenerate(H):
    out = H.scatter_add(torch.zeros(0, 16).cuda(), torch.zeros(0, dtype=torch.long).cuda(), dim=0, dim_size=7)
    assert out.shape == (7, 16) and float(out.abs().sum()) == 0.0
    out = H.scatter_add(torch.ones(5, 16).cuda(), torch.zeros(5, dtype=torch.long).cuda(), dim=0, dim_size=1)
    assert torch.allclose(out.cpu(), torch.full((1, 16), 5.0))
    out = H.scatter_add(torch.zeros(0, 16).cuda(), torch.zeros(0, dtype=torch.long).cuda(), dim=0, dim_size=0)
    assert out.shape == (0, 16)


def test_k1_out_of_range_index_raises(H):
    src = torch.ones(6, 8).cuda()
    with pytest.raises(RuntimeError, match="out of range"):
        H.scatter_add(src, torch.tensor([0, 1, 2, 9, 1, 0]).cuda(), dim=0, dim_size=4)
    with pytest.raises(RuntimeError, match="out of range"):
        H.scatter_add(src, torch.tensor([0, 1, 2, -1, 1, 0]).cuda(), dim=0, dim_size=4)


def test_k1_3d_src(H, O):
    g = torch.Generator().manual_seed(9)
    src = torch.randn(50, 4, 8, generator=g)
    index = torch.randint(0, 11, (50,), generator=g)
    out = H.scatter_add(src.cuda(), index.cuda(), dim=0, dim_size=11)
    assert out.shape == (11, 4, 8)
    assert rel_err(out.cpu().numpy(), O.scatter_add(src, index, 0, 11).numpy()) <= TOL


def test_plan_internals(H):
    g = torch.Generator().manual_seed(5)
    M, N = 5000, 64
    index = torch.randint(0, N, (M,), generator=g)
    index[:900] = 7
    plan = H.GraphPlan(index.cuda(), N, chunk=100)
    perm = plan.perm[:M].cpu().long()
    # stable sort by destination
    ref = torch.sort(index, stable=True).indices
    assert torch.equal(perm, ref)
    rowptr = plan.rowptr[:N + 1].cpu().long()
    deg = torch.bincount(index, minlength=N)
    assert torch.equal(rowptr[1:] - rowptr[:-1], deg)
    assert torch.equal(plan.dst32[:M].cpu().long(), index)
    c = plan.counts_host()
    n_split = int((deg > 100).sum())
    assert c["split"] == n_split and c["err"] == 0 and c["valid"] == M
    nch = torch.where(deg > 100, (deg + 99) // 100, torch.ones_like(deg))
    assert c["work"] == int(nch.sum())
    assert c["partial"] == int(nch[deg > 100].sum())
    # work items tile every list exactly once, in order
    wb = plan.wi_begin[:c["work"]].cpu()
    we = plan.wi_end[:c["work"]].cpu()
    assert int((we - wb).sum()) == M and int((we - wb).max()) <= 100


def test_k1_deterministic(H):
    g = torch.Generator().manual_seed(6)
    src = torch.randn(20000, 256, generator=g).cuda()
    index = torch.randint(0, 900, (20000,), generator=g).cuda()
    a = H.scatter_add(src, index, dim_size=900)
    b = H.scatter_add(src, index, dim_size=900)
    assert torch.equal(a, b)


def test_k1_backward_vs_oracle(H, O):
    g = torch.Generator().manual_seed(7)
    for F in (32, 128, 256, 6):
        src = torch.randn(800, F, generator=g)
        index = torch.randint(0, 60, (800,), generator=g)
        r = torch.randn(60, F, generator=g)
        s_ref = src.clone().requires_grad_(True)
        (O.scatter_add(s_ref, index, 0, 60) * r).sum().backward()
        s = src.cuda().requires_grad_(True)
        (H.scatter_add(s, index.cuda(), 0, 60) * r.cuda()).sum().backward()
        assert rel_err(s.grad.cpu().numpy(), s_ref.grad.numpy()) <= TOL


# ------------------------------------------------------------------ K4 weighted scatter
def test_k4_weighted_scatter_fwd_bwd(H, O):
    g = torch.Generator().manual_seed(8)
    Q, S, F = 900, 37, 64
    se = torch.randn(Q, F, generator=g)
    sw = torch.rand(Q, 1, generator=g) + 0.1
    idx = torch.randint(0, S, (Q,), generator=g)
    r = torch.randn(S, F, generator=g)
    a, w = se.clone().requires_grad_(True), sw.clone().requires_grad_(True)
    ref = O.scatter_add(a * w, idx, 0, S)          # gnn_utils.py:143
    (ref * r).sum().backward()
    a2, w2 = se.cuda().requires_grad_(True), sw.cuda().requires_grad_(True)
    out = H.scatter_add(a2, idx.cuda(), dim=0, dim_size=S, weight=w2)
    (out * r.cuda()).sum().backward()
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= TOL
    assert rel_err(a2.grad.cpu().numpy(), a.grad.numpy()) <= TOL
    assert w2.grad.shape == w.grad.shape
    assert rel_err(w2.grad.cpu().numpy(), w.grad.numpy()) <= TOL


# ------------------------------------------------------------------ K2 / K3 / K5
@pytest.mark.parametrize("F", [32, 256, 10])
def test_k2_k3_gather_scale_scatter_fwd_bwd(H, O, F):
    g = torch.Generator().manual_seed(10 + F)
    N, S, B = 400, 23, 1700
    X = torch.randn(N, F, generator=g)
    gi = torch.randint(0, N, (B,), generator=g)
    di = torch.randint(0, S, (B,), generator=g)
    di[:600] = 2  # supernode fan-in skew
    w = torch.exp(0.3 * torch.randn(B, 1, generator=g))
    r = torch.randn(S, F, generator=g)
    Xr, wr = X.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = O.scatter_add(wr * Xr[gi], di, 0, S)     # gnn_utils.py:142
    (ref * r).sum().backward()
    Xd, wd = X.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    out = H.gather_scale_scatter(Xd, gi.cuda(), di.cuda(), S, wd)
    (out * r.cuda()).sum().backward()
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= TOL
    assert rel_err(Xd.grad.cpu().numpy(), Xr.grad.numpy()) <= TOL
    assert rel_err(wd.grad.cpu().numpy(), wr.grad.numpy()) <= TOL


def test_k5_pool_golden_and_grads(H, O):
    z = load_golden("k5_pool.npz")
    nodes = torch.from_numpy(z["nodes"])
    bg = torch.from_numpy(z["bipartite_graph"])
    bw = torch.from_numpy(z["bipartite_edge_weights"])
    S = z["out"].shape[0]
    nd = nodes.cuda().requires_grad_(True)
    wd = bw.cuda().requires_grad_(True)
    out = H.gather_scale_scatter(nd, bg[0].cuda(), bg[1].cuda(), S, wd, row_scale=H.l1_row_scale(nd))
    assert rel_err(out.detach().cpu().numpy(), z["out"]) <= TOL
    g = torch.Generator().manual_seed(3)
    r = torch.randn(S, nodes.shape[1], generator=g)
    (out * r.cuda()).sum().backward()
    nr, wr = nodes.clone().requires_grad_(True), bw.clone().requires_grad_(True)
    (O.supernode_pool(nr, bg, wr, S) * r).sum().backward()
    assert rel_err(nd.grad.cpu().numpy(), nr.grad.numpy()) <= TOL
    assert rel_err(wd.grad.cpu().numpy(), wr.grad.numpy()) <= TOL


def test_bc_hgnn_call_sites(H):
    """every scatter_add the reference's BC-HGNN-GMM forward issued (K1..K5 call sites)"""
    z = load_golden("bc_hgnn_L32.npz")
    for i in range(int(z["n_scatter"])):
        out = H.scatter_add(dev(z[f"scatter{i}.src"]), dev(z[f"scatter{i}.index"]), dim=0,
                            dim_size=int(z[f"scatter{i}.dim_size"]))
        assert rel_err(out.cpu().numpy(), z[f"scatter{i}.out"]) <= TOL, i


# ------------------------------------------------------------------ K6
@pytest.mark.parametrize("F", [3, 32, 256, 512])
def test_k6_gather_rows_fwd_bwd(H, F):
    g = torch.Generator().manual_seed(20 + F)
    N, M = 300, 2100
    t = torch.randn(N, F, generator=g)
    idx = torch.randint(0, N, (M,), generator=g)
    r = torch.randn(M, F, generator=g)
    tr = t.clone().requires_grad_(True)
    (tr[idx] * r).sum().backward()
    td = t.cuda().requires_grad_(True)
    out = H.gather_rows(td, idx.cuda())
    assert torch.equal(out.detach().cpu(), t[idx])  # a gather is exact
    (out * r.cuda()).sum().backward()
    assert rel_err(td.grad.cpu().numpy(), tr.grad.numpy()) <= TOL


# ------------------------------------------------------------------ BASELINE-size properties
@pytest.fixture(scope="module")
def big(H):
    from hierarchicalgnn_amd import synth
    x, ei = synth.trackml_event(120_000, 1_000_000, seed=1234)
    graph = synth.directed(ei).cuda()
    g = torch.Generator(device="cuda").manual_seed(1235)
    src = torch.randn(graph.shape[1], 256, device="cuda", generator=g)
    return graph, src


def test_full_size_properties(H, big):
    """N=120k, M=2M, L=256 (BASELINE headline shape): size-independent properties"""
    graph, src = big
    N, M = 120_000, graph.shape[1]
    idx = graph[1]
    out = H.scatter_add(src, idx, dim=0, dim_size=N)
    # (1) conservation: column sums are preserved (fp64 accumulate of both sides)
    assert rel_err(out.double().sum(0).cpu().numpy(), src.double().sum(0).cpu().numpy()) <= 1e-6
    # (2) linearity
    a, b = 0.75, -1.5
    y = torch.roll(src, 1, 0)
    lhs = H.scatter_add(a * src + b * y, idx, dim=0, dim_size=N)
    rhs = a * out + b * H.scatter_add(y, idx, dim=0, dim_size=N)
    assert rel_err(lhs.cpu().numpy(), rhs.cpu().numpy()) <= TOL
    # (3) permutation invariance of the edge order
    p = torch.randperm(M, device="cuda")
    out_p = H.scatter_add(src[p], idx[p].contiguous(), dim=0, dim_size=N)
    assert rel_err(out_p.cpu().numpy(), out.cpu().numpy()) <= TOL
    # (4) oracle on a random sample of destination rows (CPU, arrival order)
    gsel = torch.Generator().manual_seed(1)
    rows = torch.randint(0, N, (300,), generator=gsel)
    idx_c = idx.cpu()
    for d in rows.tolist():
        e = torch.nonzero(idx_c == d).squeeze(1)
        ref = src[e.cuda()].cpu().sum(0) if e.numel() else torch.zeros(256)
        assert rel_err(out[d].cpu().numpy(), ref.numpy()) <= TOL
    # (5) idempotent / deterministic
    assert torch.equal(out, H.scatter_add(src, idx, dim=0, dim_size=N))
    # (6) isolated hits stay exactly zero
    deg = torch.bincount(idx, minlength=N)
    assert float(out[deg == 0].abs().sum()) == 0.0


def test_full_size_backward_is_a_gather(H, big):
    graph, src = big
    N = 120_000
    s = src.clone().requires_grad_(True)
    r = torch.randn(N, 256, device="cuda")
    (H.scatter_add(s, graph[1], dim=0, dim_size=N) * r).sum().backward()
    assert torch.equal(s.grad, r[graph[1]])


def test_full_size_supernode_pooling_skew(H, O):
    """K3 at the BASELINE HGNN shape: B=600k -> S=10k with heavy fan-in skew (max ~12k rows)"""
    from hierarchicalgnn_amd import synth
    bg, w = synth.bipartite_assignment(120_000, 10_000, 5)
    g = torch.Generator().manual_seed(2)
    X = torch.randn(120_000, 64, generator=g)
    ref = O.scatter_add(w * X[bg[0]], bg[1], 0, 10_000)
    out = H.gather_scale_scatter(X.cuda(), bg[0].cuda(), bg[1].cuda(), 10_000, w.cuda())
    assert rel_err(out.cpu().numpy(), ref.numpy()) <= TOL


def test_c_abi_is_graph_capturable(H):
    """no allocation / host sync inside the launch functions: K1 + its backward kernel replay
    from a captured HIP graph with new data"""
    g = torch.Generator().manual_seed(12)
    M, N, F = 20000, 700, 256
    idx = torch.randint(0, N, (M,), generator=g).cuda()
    plan = H.get_plan(idx, N)
    src = torch.randn(M, F, generator=g).cuda()
    gout = torch.randn(N, F, generator=g).cuda()
    from hierarchicalgnn_amd.ops import _seg_reduce, _spread_rows
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        _seg_reduce(plan, src, None, None)
        _spread_rows(plan, gout)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = _seg_reduce(plan, src, None, None)
        back = _spread_rows(plan, gout)
    src.copy_(torch.randn(M, F, generator=g))
    gout.copy_(torch.randn(N, F, generator=g))
    graph.replay()
    torch.cuda.synchronize()
    ref = torch.zeros(N, F, device="cuda").index_add_(0, idx, src)
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) <= TOL
    assert torch.equal(back, gout[idx])


def test_sorted_index_takes_the_streaming_path(H, O):
    g = torch.Generator().manual_seed(13)
    M, N, F = 30000, 500, 256
    idx = torch.sort(torch.randint(0, N, (M,), generator=g)).values
    idx[idx == 17] = 18                      # an empty destination inside the range
    src = torch.randn(M, F, generator=g)
    plan = H.get_plan(idx.cuda(), N)
    assert plan.sorted and plan.c.src_row is None
    out = H.scatter_add(src.cuda(), idx.cuda(), dim=0, dim_size=N, plan=plan)
    assert rel_err(out.cpu().numpy(), O.scatter_add(src, idx, 0, N).numpy()) <= TOL
    # backward through the sorted plan
    s = src.cuda().requires_grad_(True)
    r = torch.randn(N, F, generator=g).cuda()
    (H.scatter_add(s, idx.cuda(), dim=0, dim_size=N) * r).sum().backward()
    assert torch.equal(s.grad, r[idx.cuda()])
    # an unsorted index does not
    idx2 = idx.clone()
    idx2[0], idx2[-1] = idx2[-1].item(), idx2[0].item()
    assert not H.get_plan(idx2.cuda(), N).sorted


def test_randomized_shapes_against_oracle(H, O):
    """40 random (M, N, F, skew, weighting) combinations, forward and backward, incl. tiny and ragged ones"""
    g = torch.Generator().manual_seed(2024)
    for trial in range(40):
        M = int(torch.randint(0, 3000, (1,), generator=g))
        N = int(torch.randint(1, 400, (1,), generator=g))
        F = [1, 2, 4, 5, 8, 12, 16, 32, 36, 64, 96, 128, 200, 256, 384, 512][int(torch.randint(0, 16, (1,), generator=g))]
        src = torch.randn(M, F, generator=g)
        idx = torch.randint(0, N, (M,), generator=g)
        if M > 10 and trial % 3 == 0:
            idx[: M // 2] = int(torch.randint(0, N, (1,), generator=g))       # heavy destination
        weighted = trial % 2 == 1
        w = torch.rand(M, 1, generator=g) + 0.1 if weighted else None
        r = torch.randn(N, F, generator=g)
        s_ref = src.clone().requires_grad_(True)
        w_ref = w.clone().requires_grad_(True) if weighted else None
        ref = O.scatter_add(s_ref * w_ref if weighted else s_ref, idx, 0, N)
        (ref * r).sum().backward()
        s = src.cuda().requires_grad_(True)
        wd = w.cuda().requires_grad_(True) if weighted else None
        out = H.scatter_add(s, idx.cuda(), dim=0, dim_size=N, weight=wd)
        (out * r.cuda()).sum().backward()
        tag = (trial, M, N, F, weighted)
        assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= TOL, tag
        if M:
            assert rel_err(s.grad.cpu().numpy(), s_ref.grad.numpy()) <= TOL, tag
            if weighted:
                assert rel_err(wd.grad.cpu().numpy(), w_ref.grad.numpy()) <= TOL, tag


def test_inference_tensor_index_refilled_in_place_is_never_served_from_a_cache(H):
    """an index created under torch.inference_mode() has no version counter but CAN be overwritten in place there (a
    static edge_index buffer refilled per event): plans / int32 copies / derived tensors must follow the contents"""
    from hierarchicalgnn_amd import plan as P
    from hierarchicalgnn_amd.gnn_utils import InteractionGNNCell
    torch.manual_seed(3)
    N, M, L = 300, 2000, 32
    cell = InteractionGNNCell(dict(latent=L, hidden=2 * L, nb_edge_layer=2, nb_node_layer=3, layernorm=True,
                                   hidden_activation="GELU")).cuda().eval()
    with torch.inference_mode():
        g1 = torch.randint(0, N, (2, M), device="cuda")
        g2 = torch.randint(0, N, (2, M), device="cuda")
        buf = g1.clone()
        assert buf.is_inference()
        src = torch.randn(M, L, device="cuda")
        nodes = torch.randn(N, L, device="cuda")
        a1 = H.scatter_add(src, buf[1], dim=0, dim_size=N)
        n1, e1 = cell(nodes, src, buf)
        buf.copy_(g2)                                        # same storage, same shape, new event
        a2 = H.scatter_add(src, buf[1], dim=0, dim_size=N)
        n2, e2 = cell(nodes, src, buf)
        r1 = torch.zeros(N, L, device="cuda").index_add_(0, g1[1], src)
        r2 = torch.zeros(N, L, device="cuda").index_add_(0, g2[1], src)
        assert float((a1 - r1).abs().max()) < 1e-4 and float((a2 - r2).abs().max()) < 1e-4
        m1, f1 = cell(nodes, src, g1.clone())
        m2, f2 = cell(nodes, src, g2.clone())
        assert torch.equal(n1, m1) and torch.equal(e1, f1) and torch.equal(n2, m2) and torch.equal(e2, f2)
        assert not torch.equal(e1, e2)
    assert P.stable_index(g1) is not g1 and not P.stable_index(g1).is_inference()
    k = torch.randint(0, N, (M,), device="cuda")
    assert P.stable_index(k) is k
