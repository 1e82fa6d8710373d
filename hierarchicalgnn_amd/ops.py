"""Operator-level drop-ins backed by libhgnn_hip.so.

``scatter_add(src, index, dim=0, dim_size=N)`` keeps torch_scatter's calling
convention exactly as the reference uses it (Modules/gnn_utils.py:50,124,125,
142,143; BipartiteClassification/Models/HGNN_GMM.py:269).  The fused forms
(``gather_scale_scatter``) compute ``scatter_add(w * X[g], d, dim_size)``
without materialising the ``[B, L]`` product the reference builds first.

All functions are ``torch.autograd.Function``s with hand-written HIP backward
kernels; they hold no per-call state and are safe under reentrant
``torch.utils.checkpoint`` recompute (gnn_utils.py:14-15).  fp32, HIP device
tensors only: anything else raises (there is no CPU fallback).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from .plan import GraphPlan, get_plan


def _require_hip(t: torch.Tensor, name: str, allow_bf16: bool = False):
    if not t.is_cuda:
        raise RuntimeError(f"hierarchicalgnn_amd: `{name}` must be a HIP device tensor "
                           "(no CPU fallback; build + run on an MI355X)")
    if t.dtype != torch.float32 and not (allow_bf16 and t.dtype == torch.bfloat16):
        raise RuntimeError(f"hierarchicalgnn_amd: `{name}` must be float32"
                           + (" or bfloat16" if allow_bf16 else "") + f", got {t.dtype}")


def _f32(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """weights / row scales are always handed to the kernels as fp32"""
    return None if t is None else (t if t.dtype == torch.float32 else t.float())


def _seg_reduce(plan: GraphPlan, src2d: torch.Tensor, weight: Optional[torch.Tensor],
                row_scale: Optional[torch.Tensor]) -> torch.Tensor:
    F = int(src2d.shape[1])
    out = torch.empty((plan.N, F), dtype=src2d.dtype, device=src2d.device)
    if plan.N == 0 or F == 0:
        return out
    lib = _lib.load()
    weight, row_scale = _f32(weight), _f32(row_scale)
    bf16 = src2d.dtype == torch.bfloat16
    fn = lib.hgnn_segment_reduce_bf16 if bf16 else lib.hgnn_segment_reduce_f32
    with torch.cuda.device(src2d.device):
        _lib.check(fn(ctypes.byref(plan.c), _lib.ptr(src2d), F, _lib.ptr(weight), _lib.ptr(row_scale),
                      _lib.ptr(out), _lib.ptr(plan.partial(F)), _lib.current_stream(src2d.device)),
                   "hgnn_segment_reduce_bf16" if bf16 else "hgnn_segment_reduce_f32")
    return out


def _gather_rows(table: torch.Tensor, idx32: torch.Tensor, M: int, weight: Optional[torch.Tensor] = None,
                 row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    F = int(table.shape[1])
    out = torch.empty((M, F), dtype=table.dtype, device=table.device)
    if M == 0 or F == 0:
        return out
    lib = _lib.load()
    weight, row_scale = _f32(weight), _f32(row_scale)
    with torch.cuda.device(table.device):
        if table.dtype == torch.bfloat16:
            if row_scale is not None:
                raise RuntimeError("gather_rows: row_scale is not implemented for bfloat16 tables")
            _lib.check(lib.hgnn_gather_rows_bf16(
                _lib.ptr(table), int(table.shape[0]), F, _lib.ptr(idx32), M, _lib.ptr(weight),
                _lib.ptr(out), _lib.current_stream(table.device)), "hgnn_gather_rows_bf16")
        else:
            _lib.check(lib.hgnn_gather_rows_f32(
                _lib.ptr(table), int(table.shape[0]), F, _lib.ptr(idx32), M, _lib.ptr(weight),
                _lib.ptr(row_scale), _lib.ptr(out), _lib.current_stream(table.device)),
                "hgnn_gather_rows_f32")
    return out


def _spread_rows(plan: GraphPlan, table: torch.Tensor, weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[e] = w[e] * table[index[e]] walked in destination order (each table row read once)"""
    F = int(table.shape[1])
    alloc = torch.empty if plan.validated else torch.zeros
    out = alloc((plan.M, F), dtype=table.dtype, device=table.device)
    if plan.M == 0 or F == 0:
        return out
    lib = _lib.load()
    weight = _f32(weight)
    bf16 = table.dtype == torch.bfloat16
    fn = lib.hgnn_spread_rows_bf16 if bf16 else lib.hgnn_spread_rows_f32
    with torch.cuda.device(table.device):
        _lib.check(fn(ctypes.byref(plan.c), _lib.ptr(table), F, _lib.ptr(weight), _lib.ptr(out),
                      _lib.current_stream(table.device)),
                   "hgnn_spread_rows_bf16" if bf16 else "hgnn_spread_rows_f32")
    return out


def _edge_dot(A: torch.Tensor, ai: Optional[torch.Tensor], B: torch.Tensor, bi: Optional[torch.Tensor],
              M: int) -> torch.Tensor:
    F = int(A.shape[1])
    A, B = _f32(A), _f32(B)   # bf16 rows are widened once here (B <= 600k rows); the dot is fp32
    out = torch.empty((M,), dtype=torch.float32, device=A.device)
    if M == 0:
        return out
    lib = _lib.load()
    with torch.cuda.device(A.device):
        _lib.check(lib.hgnn_edge_dot_f32(
            _lib.ptr(A), _lib.ptr(ai), int(A.shape[0]), _lib.ptr(B), _lib.ptr(bi), int(B.shape[0]),
            F, M, _lib.ptr(out), _lib.current_stream(A.device)), "hgnn_edge_dot_f32")
    return out


# --------------------------------------------------------------------------- K1 / K4
class _ScatterAdd(torch.autograd.Function):
    """out[d] = sum_{e: index[e]=d} (w[e] *) src[e]"""

    @staticmethod
    def forward(ctx, src, weight, plan: GraphPlan):
        src_c = src.contiguous()
        w_c = weight.contiguous().view(-1) if weight is not None else None
        ctx.plan = plan
        ctx.has_w = weight is not None
        ctx.w_shape = weight.shape if weight is not None else None
        ctx.w_dtype = weight.dtype if weight is not None else None
        if ctx.has_w:
            ctx.save_for_backward(src_c, w_c)
        return _seg_reduce(plan, src_c, w_c, None)

    @staticmethod
    def backward(ctx, grad_out):
        plan = ctx.plan
        g = grad_out.contiguous()
        grad_src = grad_w = None
        if ctx.has_w:
            src_c, w_c = ctx.saved_tensors
            if ctx.needs_input_grad[0]:
                grad_src = _spread_rows(plan, g, weight=w_c)
            if ctx.needs_input_grad[1]:
                grad_w = _edge_dot(src_c, None, g, plan.dst32, plan.M).view(ctx.w_shape).to(ctx.w_dtype)
        elif ctx.needs_input_grad[0]:
            grad_src = _spread_rows(plan, g)
        return grad_src, grad_w, None


def scatter_add(src: torch.Tensor, index: torch.Tensor, dim: int = 0, dim_size: Optional[int] = None,
                out=None, plan: Optional[GraphPlan] = None, weight: Optional[torch.Tensor] = None,
                validate: bool = True) -> torch.Tensor:
    """torch_scatter.scatter_add for the call shape the reference uses.

    src   Tensor[M, ...] float32; index LongTensor[M] (broadcast along features); dim must be 0.
    Returns a freshly allocated Tensor[dim_size, ...] with zero rows for absent
    destinations.  Differentiable w.r.t. ``src`` (and ``weight``).
    ``weight`` (Tensor[M] or [M,1]) is an extension: fused ``scatter_add(src*weight, ...)``
    as at Modules/gnn_utils.py:143.  ``validate=False`` (an index this package produced itself, known to
    be in range) skips the one host read a new plan makes to raise on out-of-range entries.
    """
    if dim != 0 and dim != -src.dim():
        raise RuntimeError("hierarchicalgnn_amd.scatter_add: only dim=0 is implemented (the reference's use)")
    if out is not None:
        raise RuntimeError("hierarchicalgnn_amd.scatter_add: `out=` is not supported")
    _require_hip(src, "src", allow_bf16=True)
    if index.dim() != 1 or index.shape[0] != src.shape[0]:
        raise RuntimeError("hierarchicalgnn_amd.scatter_add: index must be 1-D with one entry per row of src")
    if index.device != src.device:
        raise RuntimeError("hierarchicalgnn_amd.scatter_add: index and src are on different devices")
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() else 0
    dim_size = int(dim_size)
    if plan is None:
        plan = get_plan(index, dim_size, validate=validate)
    elif plan.M != index.numel() or plan.N != dim_size or plan.c.has_gather:
        raise RuntimeError("hierarchicalgnn_amd.scatter_add: plan does not match index/dim_size")
    trailing = tuple(src.shape[1:])
    width = 1
    for t in trailing:
        width *= int(t)
    src2d = src.reshape(src.shape[0], width)
    if weight is not None:
        _require_hip(weight, "weight", allow_bf16=True)
        if weight.numel() != src.shape[0]:
            raise RuntimeError("hierarchicalgnn_amd.scatter_add: weight must have one entry per row")
    res = _ScatterAdd.apply(src2d, weight, plan)
    return res.reshape((dim_size,) + trailing)


# --------------------------------------------------------------------------- K2 / K3 / K5
class _GatherScaleScatter(torch.autograd.Function):
    """out[d] = sum_{b: dst[b]=d} w[b] * rs[g[b]] * X[g[b]]"""

    @staticmethod
    def forward(ctx, X, weight, row_scale, plan_fwd: GraphPlan, plan_bwd: GraphPlan):
        X_c = X.contiguous()
        w_c = weight.contiguous().view(-1)
        rs_c = row_scale.contiguous().view(-1) if row_scale is not None else None
        ctx.plan_fwd, ctx.plan_bwd = plan_fwd, plan_bwd
        ctx.w_shape, ctx.w_dtype = weight.shape, weight.dtype
        ctx.rs_shape = row_scale.shape if row_scale is not None else None
        ctx.rs_dtype = row_scale.dtype if row_scale is not None else None
        ctx.save_for_backward(X_c, w_c, rs_c)
        return _seg_reduce(plan_fwd, X_c, w_c, rs_c)

    @staticmethod
    def backward(ctx, grad_out):
        X_c, w_c, rs_c = ctx.saved_tensors
        pf, pb = ctx.plan_fwd, ctx.plan_bwd
        g = grad_out.contiguous()
        grad_X = grad_w = grad_rs = None
        if ctx.needs_input_grad[0] or (rs_c is not None and ctx.needs_input_grad[2]):
            # T[s] = sum_{b: g[b]=s} w[b] * grad_out[dst[b]]   (transposed plan)
            T = _seg_reduce(pb, g, w_c, None)
            if rs_c is not None:
                if ctx.needs_input_grad[2]:
                    grad_rs = (T.float() * X_c.float()).sum(dim=1).view(ctx.rs_shape).to(ctx.rs_dtype)
                if ctx.needs_input_grad[0]:
                    grad_X = (T * rs_c.unsqueeze(1)).to(T.dtype)
            else:
                grad_X = T
        if ctx.needs_input_grad[1]:
            # dw[b] = rs[g[b]] * <grad_out[dst[b]], X[g[b]]>
            d = _edge_dot(g, pf.dst32, X_c, pb.dst32, pf.M)
            if rs_c is not None:
                d = d * rs_c[pb.dst32.long()[:pf.M]] if pf.M else d
            grad_w = d.view(ctx.w_shape).to(ctx.w_dtype)
        return grad_X, grad_w, grad_rs, None, None


def gather_scale_scatter(X: torch.Tensor, gather_index: torch.Tensor, dst_index: torch.Tensor, dim_size: int,
                         weight: torch.Tensor, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused ``scatter_add(weight * (row_scale[:,None] * X)[gather_index], dst_index, dim=0, dim_size)``.

    K2: Modules/gnn_utils.py:124  (X=supernodes, gather=bipartite_graph[1], dst=bipartite_graph[0])
    K3: Modules/gnn_utils.py:142  (X=nodes,      gather=bipartite_graph[0], dst=bipartite_graph[1])
    K5: BipartiteClassification/Models/HGNN_GMM.py:269 (K3 with row_scale = 1/||nodes||_1)
    The [B, L] product is never written to HBM.  Differentiable w.r.t. X, weight, row_scale.
    """
    _require_hip(X, "X", allow_bf16=True)
    _require_hip(weight, "weight", allow_bf16=True)
    if X.dim() != 2:
        raise RuntimeError("gather_scale_scatter: X must be [rows, features]")
    if weight.numel() != gather_index.numel():
        raise RuntimeError("gather_scale_scatter: weight must have one entry per (gather, dst) pair")
    n_src = int(X.shape[0])
    plan_fwd = get_plan(dst_index, int(dim_size), gather_index, n_src)
    plan_bwd = get_plan(gather_index, n_src, dst_index, int(dim_size))
    if row_scale is not None:
        _require_hip(row_scale, "row_scale", allow_bf16=True)
        if row_scale.numel() != n_src:
            raise RuntimeError("gather_scale_scatter: row_scale must have one entry per row of X")
    return _GatherScaleScatter.apply(X, weight, row_scale, plan_fwd, plan_bwd)


# --------------------------------------------------------------------------- K6
class _GatherRows(torch.autograd.Function):
    """out[e] = table[index[e]]; backward = segmented reduce over the CSR by `index`"""

    @staticmethod
    def forward(ctx, table, plan: GraphPlan):
        ctx.plan = plan
        return _spread_rows(plan, table.contiguous())

    @staticmethod
    def backward(ctx, grad_out):
        if not ctx.needs_input_grad[0]:
            return None, None
        return _seg_reduce(ctx.plan, grad_out.contiguous(), None, None), None


def gather_rows(table: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """``table[index]`` for a 2-D table and a 1-D int64 index (Modules/gnn_utils.py:61,134,152).
    Forward is a whole-row HIP gather; backward is the atomics-free segmented reduce."""
    _require_hip(table, "table", allow_bf16=True)
    if table.dim() != 2 or index.dim() != 1:
        raise RuntimeError("gather_rows: table must be 2-D and index 1-D")
    plan = get_plan(index, int(table.shape[0]))
    return _GatherRows.apply(table, plan)


def l1_row_scale(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """1 / max(||x_i||_1, eps): the per-row factor of F.normalize(x, p=1) (HGNN_GMM.py:269)"""
    return 1.0 / x.abs().sum(dim=1).clamp_min(eps)


# --------------------------------------------------------------------------- K11 per-edge dot products
class _EdgeDot(torch.autograd.Function):
    """out[e] = <A[ai[e]], B[bi[e]]>  (torch.einsum('ij,ij->i', A[ai], B[bi]), gnn_utils.py:208)"""

    @staticmethod
    def forward(ctx, A, B, ai, bi):
        from .plan import get_index32
        A_c, B_c = A.contiguous(), B.contiguous()
        ctx.save_for_backward(A_c, B_c)
        ctx.idx = (ai, bi)
        return _edge_dot(A_c, get_index32(ai, A_c.shape[0]), B_c, get_index32(bi, B_c.shape[0]), int(ai.numel()))

    @staticmethod
    def backward(ctx, g):
        A_c, B_c = ctx.saved_tensors
        ai, bi = ctx.idx
        g_c = g.contiguous()
        grad_A = grad_B = None
        if ctx.needs_input_grad[0]:   # dA[a] = sum_{e: ai[e]=a} g[e] * B[bi[e]]
            grad_A = _seg_reduce(get_plan(ai, int(A_c.shape[0]), bi, int(B_c.shape[0])), B_c, g_c, None)
        if ctx.needs_input_grad[1]:
            grad_B = _seg_reduce(get_plan(bi, int(B_c.shape[0]), ai, int(A_c.shape[0])), A_c, g_c, None)
        return grad_A, grad_B, None, None


def edge_dot(A: torch.Tensor, ai: torch.Tensor, B: torch.Tensor, bi: torch.Tensor) -> torch.Tensor:
    """per-edge dot product of two gathered rows without materialising the gathers; differentiable"""
    _require_hip(A, "A")
    _require_hip(B, "B")
    if A.dim() != 2 or B.dim() != 2 or A.shape[1] != B.shape[1] or ai.shape != bi.shape or ai.dim() != 1:
        raise RuntimeError("edge_dot: A[*,F], B[*,F] and 1-D index tensors of equal length expected")
    return _EdgeDot.apply(A, B, ai, bi)


def knn_radius(query: torch.Tensor, points: torch.Tensor, k: int, radius, return_dist2: bool = False):
    """<=k nearest `points` of every `query` row with squared distance < radius^2, ascending,
    -1 padded: the idxs of frnn.frnn_grid_points (Modules/utils.py:232) for one batch.
    ``radius``: a float, or a 1-element float32 device tensor (the module's ``knn_radius`` buffer) that the
    kernel reads itself -- no host read of it."""
    _require_hip(query, "query")
    _require_hip(points, "points")
    if query.dim() != 2 or points.dim() != 2 or query.shape[1] != points.shape[1]:
        raise RuntimeError("knn_radius: query[N,D] and points[S,D] expected")
    q, p = query.detach().contiguous(), points.detach().contiguous()
    nq, D = int(q.shape[0]), int(q.shape[1])
    r_dev, r_val = None, 0.0
    if torch.is_tensor(radius):
        if radius.numel() != 1 or radius.device != q.device:
            raise RuntimeError("knn_radius: a tensor radius must have one element on the queries' device")
        r_dev = radius.detach().reshape(1).float().contiguous()
    else:
        r_val = float(radius)
    idx = torch.empty((nq, int(k)), dtype=torch.int64, device=q.device)
    d2 = torch.empty((nq, int(k)), dtype=torch.float32, device=q.device) if return_dist2 else None
    lib = _lib.load()
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.hgnn_knn_workspace_bytes(nq, int(p.shape[0]), int(k), ctypes.byref(nbytes)),
               "hgnn_knn_workspace_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=q.device) if nbytes.value else None
    with torch.cuda.device(q.device):
        _lib.check(lib.hgnn_knn_radius_ws_f32(_lib.ptr(q), nq, _lib.ptr(p), int(p.shape[0]), D, int(k),
                                              ctypes.c_float(r_val), _lib.ptr(r_dev), _lib.ptr(idx), _lib.ptr(d2),
                                              _lib.ptr(ws), nbytes.value, _lib.current_stream(q.device)),
                   "hgnn_knn_radius_ws_f32")
    return (idx, d2) if return_dist2 else idx


def wgrad_bf16(dz: torch.Tensor, rows: torch.Tensor, out: Optional[torch.Tensor] = None,
               colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``dz^T @ rows`` for bf16 ``dz[M, Ho]`` and ``rows[M, Hi]`` (row-major, possibly column slices of wider
    matrices) in fp32: the weight gradient of a Linear layer, by the hand-written split-K bf16-MFMA kernel
    (``hgnn_wgrad_bf16``; deterministic).  ``out`` (fp32 [Ho, Hi], possibly a column slice) is written in place;
    ``colsum`` (fp32 [Ho], optional) receives ``dz.sum(0)`` -- the Linear's bias gradient -- from the same pass."""
    _require_hip(dz, "dz", allow_bf16=True)
    _require_hip(rows, "rows", allow_bf16=True)
    if dz.dtype != torch.bfloat16 or rows.dtype != torch.bfloat16 or dz.dim() != 2 or rows.dim() != 2 \
            or dz.shape[0] != rows.shape[0]:
        raise RuntimeError("wgrad_bf16: bf16 dz[M, Ho] and rows[M, Hi] expected")
    if dz.stride(1) != 1:
        dz = dz.contiguous()
    if rows.stride(1) != 1:
        rows = rows.contiguous()
    M, Ho, Hi = int(dz.shape[0]), int(dz.shape[1]), int(rows.shape[1])
    if out is None:
        out = torch.empty((Ho, Hi), dtype=torch.float32, device=dz.device)
    elif out.dtype != torch.float32 or tuple(out.shape) != (Ho, Hi) or out.stride(1) != 1:
        raise RuntimeError("wgrad_bf16: out must be fp32 [Ho, Hi] with unit column stride")
    lib = _lib.load()
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.hgnn_wgrad_workspace_bytes(M, Ho, Hi, ctypes.byref(nbytes)), "hgnn_wgrad_workspace_bytes")
    ws = torch.empty(max(nbytes.value, 16), dtype=torch.uint8, device=dz.device)
    lda = int(dz.stride(0)) if M > 1 else Ho
    ldb = int(rows.stride(0)) if M > 1 else Hi
    with torch.cuda.device(dz.device):
        if colsum is not None and (colsum.dtype != torch.float32 or colsum.numel() != Ho or not colsum.is_contiguous()):
            raise RuntimeError("wgrad_bf16: colsum must be a contiguous fp32 [Ho] tensor")
        _lib.check(lib.hgnn_wgrad_bf16(_lib.ptr(dz), lda, _lib.ptr(rows), ldb, M, Ho, Hi,
                                       ctypes.c_void_p(out.data_ptr()), int(out.stride(0)),
                                       ctypes.c_void_p(colsum.data_ptr()) if colsum is not None else None,
                                       _lib.ptr(ws), ws.numel(), _lib.current_stream(dz.device)), "hgnn_wgrad_bf16")
    return out


def wgrad_f32_split3(dz: torch.Tensor, rows: torch.Tensor, out: Optional[torch.Tensor] = None,
                     colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``dz^T @ rows`` for FP32 ``dz[M, Ho]`` and ``rows[M, Hi]`` with the products evaluated as split-bf16
    (``hgnn_wgrad_f32_split3``: hi.hi + mid.hi + hi.mid on the bf16 matrix pipe, fp32 accumulation, deterministic):
    the M-row weight gradients of the fp32 training backward.  Same conventions as :func:`wgrad_bf16`."""
    _require_hip(dz, "dz")
    _require_hip(rows, "rows")
    if dz.dtype != torch.float32 or rows.dtype != torch.float32 or dz.dim() != 2 or rows.dim() != 2 \
            or dz.shape[0] != rows.shape[0]:
        raise RuntimeError("wgrad_f32_split3: fp32 dz[M, Ho] and rows[M, Hi] expected")
    if dz.stride(1) != 1 or (dz.shape[0] > 1 and dz.stride(0) % 4) or dz.data_ptr() % 16:
        dz = dz.contiguous()
    if rows.stride(1) != 1 or (rows.shape[0] > 1 and rows.stride(0) % 4) or rows.data_ptr() % 16:
        rows = rows.contiguous()
    M, Ho, Hi = int(dz.shape[0]), int(dz.shape[1]), int(rows.shape[1])
    if out is None:
        out = torch.empty((Ho, Hi), dtype=torch.float32, device=dz.device)
    elif out.dtype != torch.float32 or tuple(out.shape) != (Ho, Hi) or out.stride(1) != 1:
        raise RuntimeError("wgrad_f32_split3: out must be fp32 [Ho, Hi] with unit column stride")
    lib = _lib.load()
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.hgnn_wgrad_workspace_bytes(M, Ho, Hi, ctypes.byref(nbytes)), "hgnn_wgrad_workspace_bytes")
    ws = torch.empty(max(nbytes.value, 16), dtype=torch.uint8, device=dz.device)
    lda = int(dz.stride(0)) if M > 1 else Ho
    ldb = int(rows.stride(0)) if M > 1 else Hi
    with torch.cuda.device(dz.device):
        if colsum is not None and (colsum.dtype != torch.float32 or colsum.numel() != Ho or not colsum.is_contiguous()):
            raise RuntimeError("wgrad_f32_split3: colsum must be a contiguous fp32 [Ho] tensor")
        _lib.check(lib.hgnn_wgrad_f32_split3(_lib.ptr(dz), lda, _lib.ptr(rows), ldb, M, Ho, Hi,
                                             ctypes.c_void_p(out.data_ptr()), int(out.stride(0)),
                                             ctypes.c_void_p(colsum.data_ptr()) if colsum is not None else None,
                                             _lib.ptr(ws), ws.numel(), _lib.current_stream(dz.device)),
                   "hgnn_wgrad_f32_split3")
    return out
