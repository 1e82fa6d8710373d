#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: HIP ops vs plain torch on the same device, many shapes.
scatter_add (K1, weighted, gathered), the fused MLP (fp32: inference, training gradients; bf16), the
LayerNorm/activation row kernels.  Usage: fuzz_parity.py [n_cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hierarchicalgnn_amd as H
from hierarchicalgnn_amd import fused, make_mlp, mlp

fused.set_fp32_split3(False)   # the exact fp32 kernels are the default subject; the split-bf16 path has its own block
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = torch.Generator().manual_seed(seed)


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(float(b.double().abs().max()), 1e-30))


worst = {}


def note(name, err, tol, ctx):
    worst[name] = max(worst.get(name, 0.0), err)
    assert err <= tol, (name, err, tol, ctx)


for case in range(n_cases):
    # ---------------- scatter_add family
    N, M, F = ri(1, 3000), ri(0, 20000), [4, 8, 12, 32, 64, 100, 128, 256, 512][ri(0, 8)]
    skew = ri(0, 2)
    idx = torch.randint(0, N, (M,), generator=g)
    if skew == 1 and M:
        idx = (idx % max(1, N // 50))            # duplicate-heavy
    if skew == 2 and M:
        idx = torch.sort(idx).values             # destination-sorted
    src = torch.randn(M, F, generator=g).cuda()
    idx_c = idx.cuda()
    out = H.scatter_add(src, idx_c, dim=0, dim_size=N)
    ref = torch.zeros(N, F, device="cuda").index_add_(0, idx_c, src)
    note("scatter_add", rel(out, ref) if M else float(out.abs().sum()), 1e-5, (N, M, F, skew))
    w = torch.rand(M, 1, generator=g).cuda()
    out = H.scatter_add(src, idx_c, dim=0, dim_size=N, weight=w)
    ref = torch.zeros(N, F, device="cuda").index_add_(0, idx_c, src * w)
    note("scatter_add_weighted", rel(out, ref) if M else 0.0, 1e-5, (N, M, F, skew))
    R = ri(1, 500)
    table = torch.randn(R, F, generator=g).cuda()
    gi = torch.randint(0, R, (M,), generator=g).cuda()
    out = H.gather_scale_scatter(table, gi, idx_c, N, w)
    ref = torch.zeros(N, F, device="cuda").index_add_(0, idx_c, table[gi] * w)
    note("gather_scale_scatter", rel(out, ref) if M else 0.0, 1e-5, (N, M, F, R))

    # ---------------- fixed-radius kNN (exact: compare index sets and distances) and per-edge dots
    from hierarchicalgnn_amd.ops import edge_dot, knn_radius
    D, K = [2, 3, 8, 12, 16][ri(0, 4)], [1, 3, 5, 10, 16][ri(0, 4)]
    nq, npnt = ri(1, 700), ri(1, 900)
    q = torch.randn(nq, D, generator=g).cuda()
    pts = torch.randn(npnt, D, generator=g).cuda()
    radius = float(torch.rand(1, generator=g)) * 3 + 0.2
    idx_k, d2_k = knn_radius(q, pts, K, radius, return_dist2=True)
    dd = ((q[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    dd = torch.where(dd < radius * radius, dd, torch.full_like(dd, float("inf")))
    kk = min(K, npnt)
    ref_d, _ = torch.topk(dd, kk, dim=1, largest=False)
    got_d = torch.where(idx_k[:, :kk] >= 0, d2_k[:, :kk], torch.full_like(d2_k[:, :kk], float("inf")))
    fin = torch.isfinite(ref_d)
    assert bool((torch.isfinite(got_d) == fin).all()), ("knn count", nq, npnt, D, K, radius)
    note("knn_dist2", float((got_d[fin] - ref_d[fin]).abs().max()) if bool(fin.any()) else 0.0, 1e-4, (nq, npnt, D, K))
    assert bool((idx_k[:, kk:] == -1).all())
    Bq = ri(0, 3000)
    ai = torch.randint(0, nq, (Bq,), generator=g).cuda()
    bi = torch.randint(0, npnt, (Bq,), generator=g).cuda()
    note("edge_dot", rel(edge_dot(q, ai, pts, bi), (q[ai] * pts[bi]).sum(-1)) if Bq else 0.0, 1e-5, (Bq, D))

    # ---------------- fused MLP, fp32 inference + training gradients
    L = [32, 64, 128, 256][ri(0, 3)]
    layers = ri(2, 3)
    Mm = ri(1, 3000)
    n_tab = ri(1, 400)
    net = make_mlp(3 * L, 2 * L, L, layers, layer_norm=True, output_activation=["Tanh", "GELU"][ri(0, 1)],
                   hidden_activation="GELU").cuda()
    tab = torch.randn(n_tab, L, generator=g).cuda()
    direct = torch.randn(Mm, L, generator=g).cuda()
    i0 = torch.randint(0, n_tab, (Mm,), generator=g).cuda()
    i1 = torch.randint(0, n_tab, (Mm,), generator=g).cuda()
    segs = [(tab, i0), (tab, i1), (direct, None)]
    with torch.no_grad():
        out = mlp.concat_mlp(net, segs, skip=direct)
        ref = net(torch.cat([tab[i0], tab[i1], direct], dim=1)) + direct
    note("fused_mlp_f32", rel(out, ref), 1e-4, (L, layers, Mm, n_tab))
    if L >= 128:
        # opt-in split-bf16 evaluation of the same fp32 MLP (hgnn_mlp_forward_f32_split3): against fp64
        fused.set_fp32_split3(True)
        try:
            n_s3 = fused.stats.get("split3_calls", 0)
            with torch.no_grad():
                o3 = mlp.concat_mlp(net, segs, skip=direct)
                ref64 = net.double()(torch.cat([tab[i0], tab[i1], direct], dim=1).double()) + direct.double()
                net.float()
            assert fused.stats.get("split3_calls", 0) == n_s3 + 1
        finally:
            fused.set_fp32_split3(False)
        note("fused_mlp_f32_split3_vs_fp64", rel(o3, ref64), 3e-5, (L, layers, Mm, n_tab))
    t1, d1 = tab.clone().requires_grad_(True), direct.clone().requires_grad_(True)
    r = torch.randn(Mm, L, generator=g).cuda()
    n0 = fused.stats["fused_train_calls"]
    (mlp.concat_mlp(net, [(t1, i0), (t1, i1), (d1, None)], skip=d1) * r).sum().backward()
    assert fused.stats["fused_train_calls"] == n0 + 1
    got = [t1.grad, d1.grad] + [p.grad.clone() for p in net.parameters()]
    net.zero_grad(set_to_none=True)
    t2, d2 = tab.clone().requires_grad_(True), direct.clone().requires_grad_(True)
    ((net(torch.cat([t2[i0], t2[i1], d2], dim=1)) + d2) * r).sum().backward()
    want = [t2.grad, d2.grad] + [p.grad.clone() for p in net.parameters()]
    for k, (a, b) in enumerate(zip(got, want)):
        note("fused_mlp_f32_grads", rel(a, b), 2e-4, (L, layers, Mm, n_tab, k))

    # ---------------- the 128-row tile of the split-bf16 fp32 MLP (K -> 512 -> 256 from 65,536 rows), every 8th case
    if case % 8 == 0:
        Lb, nseg_b = 256, ri(1, 3)
        Mb = ri(65536, 90000)
        nt = ri(50, 5000)
        netb = make_mlp(nseg_b * Lb, 2 * Lb, Lb, 2, layer_norm=True, output_activation=["Tanh", "GELU", "ReLU"][ri(0, 2)],
                        hidden_activation=["GELU", "Tanh"][ri(0, 1)]).cuda()
        tb = torch.randn(nt, Lb, generator=g).cuda()
        db = torch.randn(Mb, Lb, generator=g).cuda()
        j0 = torch.randint(0, nt, (Mb,), generator=g).cuda()
        j1 = torch.randint(0, nt, (Mb,), generator=g).cuda()
        if ri(0, 1):
            j1 = torch.sort(j1).values
        segs_b = [(tb, j0), (tb, j1), (db, None)][3 - nseg_b:]
        skip_b = db if ri(0, 3) else None
        fused.set_fp32_split3(True)
        try:
            n_s3 = fused.stats.get("split3_calls", 0)
            with torch.no_grad():
                ob = mlp.concat_mlp(netb, segs_b, skip=skip_b)
                xb = torch.cat([t.double() if i is None else t.double()[i] for t, i in segs_b], dim=1)
                rb = netb.double()(xb) + (db.double() if skip_b is not None else 0)
                netb.float()
            assert fused.stats.get("split3_calls", 0) == n_s3 + 1
        finally:
            fused.set_fp32_split3(False)
        err_b = float((ob.double() - rb).abs().max()) / max(float(rb.abs().max()), 1.0)
        note("fused_mlp_f32_split3_rows128_vs_fp64", err_b, 5e-5, (nseg_b, Mb, nt, skip_b is not None))

    # ---------------- fused MLP, bf16
    if L >= 64:
        segs16 = [(tab.bfloat16(), i0), (tab.bfloat16(), i1), (direct.bfloat16(), None)]
        with torch.no_grad():
            out = mlp.concat_mlp(net, segs16, skip=segs16[2][0]).float()
        note("fused_mlp_bf16_vs_fp32", rel(out, ref), 3e-2, (L, layers, Mm, n_tab))
        if L >= 128:
            # second session of round 2: pre-projected node segments (whole-row LDS-DMA gather) forced on / off
            for name, pre in (("preproject_on", True), ("preproject_off", False)):
                fused._preproject_bf16 = pre
                try:
                    with torch.no_grad():
                        o2 = mlp.concat_mlp(net, segs16, skip=segs16[2][0]).float()
                finally:
                    fused._preproject_bf16 = None
                note("fused_mlp_bf16_" + name + "_vs_fp32", rel(o2, ref), 3e-2, (L, layers, Mm, n_tab))

    # ---------------- round 2: bf16 training kernels (weight gradient, fused backward layer, training gradients)
    from hierarchicalgnn_amd.ops import wgrad_bf16
    Mw, Ho, Hi = ri(0, 9000), 8 * ri(1, 80), 8 * ri(1, 80)
    dzw = torch.randn(Mw, Ho, generator=g).cuda().bfloat16()
    rows = torch.randn(Mw, Hi, generator=g).cuda().bfloat16()
    cs = torch.empty(Ho, device="cuda")
    outw = wgrad_bf16(dzw, rows, colsum=cs)
    refw = dzw.float().t() @ rows.float()
    note("wgrad_bf16", rel(outw, refw) if Mw else float(outw.abs().sum()), 3e-5, (Mw, Ho, Hi))
    note("wgrad_bf16_colsum", rel(cs, dzw.float().sum(0)) if Mw else float(cs.abs().sum()), 3e-5, (Mw, Ho))
    Kb, Nb, Mb = 128 * ri(1, 6), [128, 256, 512][ri(0, 2)], ri(1, 4000)
    actb = ri(1, 3)
    dzb = torch.randn(Mb, Kb, generator=g).cuda().bfloat16()
    Wb = (torch.randn(Kb, Nb, generator=g) / Kb ** 0.5).cuda()
    zb = (1.5 * torch.randn(Mb, Nb, generator=g) + 0.2).cuda().bfloat16()
    gmb = (1 + 0.2 * torch.randn(Nb, generator=g)).cuda()
    btb = (0.2 * torch.randn(Nb, generator=g)).cuda()
    zf = zb.float().requires_grad_(True)
    gmr, btr = gmb.clone().requires_grad_(True), btb.clone().requires_grad_(True)
    fn = {1: torch.nn.functional.gelu, 2: torch.tanh, 3: torch.relu}[actb]
    fn(torch.nn.functional.layer_norm(zf, [Nb], gmr, btr, 1e-5)).backward(dzb.float() @ Wb.bfloat16().float())
    dzp, _, dgb, dbb = fused._bwd_layer(dzb, Wb, zb, gmb, btb, actb, 1e-5, want_a=True)
    note("bwd_layer_bf16_dz", rel(dzp.float(), zf.grad), 6e-3, (Mb, Kb, Nb, actb))
    note("bwd_layer_bf16_dgamma", rel(dgb, gmr.grad), 2e-4, (Mb, Kb, Nb, actb))
    note("bwd_layer_bf16_dbeta", rel(dbb, btr.grad), 2e-4, (Mb, Kb, Nb, actb))
    skb = torch.randn(Mb, Nb, generator=g).cuda().bfloat16()
    dxb = fused._bwd_layer(dzb, Wb, None, None, None, 0, 1e-5, skip=skb)[0]
    note("bwd_layer_bf16_input_form", rel(dxb.float(), dzb.float() @ Wb.bfloat16().float() + skb.float()), 6e-3, (Mb, Kb, Nb))
    if L >= 128:
        t3 = tab.bfloat16().clone().requires_grad_(True)
        d3 = direct.bfloat16().clone().requires_grad_(True)
        net.zero_grad(set_to_none=True)
        n0 = fused.stats["fused_train_calls"]
        (mlp.concat_mlp(net, [(t3, i0), (t3, i1), (d3, None)], skip=d3).float() * r).sum().backward()
        assert fused.stats["fused_train_calls"] == n0 + 1
        got16 = [t3.grad.float(), d3.grad.float()] + [p.grad.clone() for p in net.parameters()]
        for k, (a, b) in enumerate(zip(got16, want)):
            note("fused_mlp_bf16_train_grads_vs_fp32", rel(a, b), 5e-2, (L, layers, Mm, n_tab, k))

    # ---------------- round 2: connected components (union-find) against a sequential union-find on the host
    from hierarchicalgnn_amd.clustering import connected_components
    nv, ne = ri(1, 4000), ri(0, 6000)
    cs_, cd_ = torch.randint(0, nv, (ne,), generator=g), torch.randint(0, nv, (ne,), generator=g)
    lab = connected_components(cs_.cuda(), cd_.cuda(), nv).cpu().tolist()
    par = list(range(nv))

    def find(v):
        while par[v] != v:
            par[v] = par[par[v]]
            v = par[v]
        return v

    for u, v in zip(cs_.tolist(), cd_.tolist()):
        ru, rv = find(u), find(v)
        if ru != rv:
            par[max(ru, rv)] = min(ru, rv)
    assert lab == [find(v) for v in range(nv)], ("connected_components", nv, ne)

print("cases", n_cases, "seed", seed, "worst relative errors:", {k: f"{v:.2e}" for k, v in worst.items()})
