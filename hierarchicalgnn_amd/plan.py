"""GraphPlan: the per-event, destination-sorted aggregation plan.

The topology of an event is constant across every message-passing iteration of
one forward (reference: EdgeClassifier/Models/IN.py:87-88 loops 14 cells over
the same ``graph``; BipartiteClassification/Models/HGNN_GMM.py:93-94,:275-284
loop 6 + 6), so the int64 -> int32 conversion, the stable sort by destination,
the CSR and the degree-skew work list are built ONCE (``hgnn_plan_build`` in
libhgnn_hip.so) and reused by every ``scatter_add`` / gather call.

Plans are found again through a small cache keyed on the identity of the index
tensor(s).  The cache keeps a strong reference to the tensors it is keyed on
(so their storage cannot be recycled under a live key) and checks the tensors'
in-place version counters, so it never returns a plan for different contents.
Autograd / ``torch.utils.checkpoint`` recompute re-enters with the same index
tensors and therefore hits the cache; no per-call state is kept.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict
from typing import Optional

import torch

from . import _lib

_CACHE: "OrderedDict[tuple, GraphPlan]" = OrderedDict()
_CACHE_SIZE = 16
_STATS = {"built": 0, "hits": 0}


def _i32(n, device):
    return torch.empty(max(int(n), 1), dtype=torch.int32, device=device)


class GraphPlan:
    """Device-side plan for ``out[d] = sum_{e: dst[e]=d} w[e] * table[gather[e]]``.

    dst_index    LongTensor[M]  destination row of every edge (e.g. ``graph[1]``)
    dim_size     int            number of destination rows N
    gather_index LongTensor[M]  optional source-table row of every edge
                                (e.g. ``bipartite_graph[0]``); ``None`` means the
                                table row of edge e is e itself (plain scatter_add)
    n_src        int            rows of the source table (required with gather_index)
    """

    def __init__(self, dst_index: torch.Tensor, dim_size: int, gather_index: Optional[torch.Tensor] = None,
                 n_src: Optional[int] = None, chunk: int = 0, validate: bool = True):
        if not dst_index.is_cuda:
            raise RuntimeError("GraphPlan needs a HIP device index tensor: hierarchicalgnn_amd has no CPU path")
        if dst_index.dim() != 1 or dst_index.dtype != torch.int64:
            raise RuntimeError("GraphPlan: index must be a 1-D int64 tensor (PyG edge_index row)")
        if gather_index is not None:
            if gather_index.shape != dst_index.shape or gather_index.dtype != torch.int64 \
                    or gather_index.device != dst_index.device:
                raise RuntimeError("GraphPlan: gather_index must match index in shape, dtype and device")
            if n_src is None:
                raise RuntimeError("GraphPlan: n_src is required with gather_index")
        lib = _lib.load()
        dev = dst_index.device
        M = int(dst_index.numel())
        N = int(dim_size)
        R = int(n_src) if gather_index is not None else M
        self.device = dev
        self.M, self.N, self.R = M, N, R
        c = _lib.HgnnPlan()
        _lib.check(lib.hgnn_plan_dims(M, N, R, int(chunk), ctypes.byref(c)), "hgnn_plan_dims")
        self.chunk = int(c.chunk)
        self.max_partial = int(c.max_partial)
        with torch.cuda.device(dev):
            dst_c = dst_index.contiguous()
            gat_c = gather_index.contiguous() if gather_index is not None else None
            self.perm = _i32(M, dev)
            self.src_row = _i32(M, dev)
            self.dst32 = _i32(M, dev)
            self.rowptr = _i32(N + 1, dev)
            self.wi_begin = _i32(c.max_work, dev)
            self.wi_end = _i32(c.max_work, dev)
            self.wi_target = _i32(c.max_work, dev)
            self.wi_dst = _i32(c.max_work, dev)
            self.split_dst = _i32(c.max_split, dev)
            self.split_pbegin = _i32(c.max_split + 1, dev)
            self.counts = torch.zeros(8, dtype=torch.int32, device=dev)
            for name in ("perm", "src_row", "dst32", "rowptr", "wi_begin", "wi_end", "wi_target", "wi_dst",
                         "split_dst", "split_pbegin", "counts"):
                setattr(c, name, getattr(self, name).data_ptr())
            nbytes = ctypes.c_size_t(0)
            _lib.check(lib.hgnn_plan_workspace_bytes(M, N, ctypes.byref(nbytes)), "hgnn_plan_workspace_bytes")
            ws = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device=dev)
            _lib.check(lib.hgnn_plan_build(_lib.ptr(dst_c), _lib.ptr(gat_c), ctypes.byref(c), _lib.ptr(ws),
                                           ws.numel(), _lib.current_stream(dev)), "hgnn_plan_build")
            # `ws`, `dst_c`, `gat_c` are only used by kernels already enqueued on this
            # stream; the caching allocator orders their reuse after them.
        self.c = c
        self.validated = bool(validate)
        self.sorted = False
        self._partial = {}
        _STATS["built"] += 1
        if validate:
            # one host sync per plan (= per event), never per aggregation call
            host = self.counts.cpu()
            # index already sorted by destination (a model-level layout choice, models.py): the
            # reduce can stream rows begin..end contiguously instead of gathering them by id
            if gather_index is None and int(host[_lib.CNT_UNSORTED]) == 0 and M > 0:
                self.sorted = True
                self.c.src_row = None
            if int(host[_lib.CNT_ERR]) != 0:
                raise RuntimeError(
                    f"GraphPlan: index out of range for dim_size={N}" +
                    (f" / source rows={R}" if gather_index is not None else ""))

    # scratch for the partial sums of split destinations, one buffer per feature width
    def partial(self, F: int) -> torch.Tensor:
        buf = self._partial.get(F)
        if buf is None:
            buf = torch.empty(max(self.max_partial, 1) * F, dtype=torch.float32, device=self.device)
            self._partial[F] = buf
        return buf

    def counts_host(self):
        c = self.counts.cpu().tolist()
        return dict(work=c[_lib.CNT_WORK], split=c[_lib.CNT_SPLIT], partial=c[_lib.CNT_PARTIAL],
                    err=c[_lib.CNT_ERR], valid=c[_lib.CNT_VALID])


_UNCACHEABLE = object()


def _key(t: Optional[torch.Tensor]):
    if t is None:
        return None
    # inference tensors (created under torch.inference_mode()) track no version counter, yet CAN be edited in place
    # inside inference mode (a reused static ``edge_index`` buffer refilled with ``copy_``): nothing tells two events
    # apart, so nothing derived from such a tensor is ever cached (``stable_index`` gives callers that loop over many
    # aggregations of one event a normal clone to key on)
    if t.is_inference():
        return _UNCACHEABLE
    return (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), t.device.index)


def stable_index(t: torch.Tensor) -> torch.Tensor:
    """the tensor itself, or -- for an inference tensor -- a NORMAL clone of its current contents.  The cells and
    models call this once per forward on the caller's graph: every plan / int32 copy / derived tensor of that forward
    is then keyed on the clone (cache hits across the 14 / 6 + 6 aggregations of the event), and a later in-place
    refill of the caller's buffer yields a new clone, hence new plans."""
    if torch.is_tensor(t) and t.is_inference():
        with torch.inference_mode(False):
            return t.clone()
    return t


def get_plan(dst_index: torch.Tensor, dim_size: int, gather_index: Optional[torch.Tensor] = None,
             n_src: Optional[int] = None, validate: bool = True) -> GraphPlan:
    """cached GraphPlan for (dst_index, dim_size[, gather_index, n_src])"""
    key = (_key(dst_index), int(dim_size), _key(gather_index), None if n_src is None else int(n_src))
    if key[0] is _UNCACHEABLE or key[2] is _UNCACHEABLE:
        with torch.inference_mode(False):
            return GraphPlan(stable_index(dst_index), dim_size,
                             None if gather_index is None else stable_index(gather_index), n_src, validate=validate)
    hit = _CACHE.get(key)
    if hit is not None:
        _CACHE.move_to_end(key)
        _STATS["hits"] += 1
        return hit[0]
    # cached objects outlive the call: build them as NORMAL tensors even under torch.inference_mode()
    # (an inference tensor cached here could later not be saved for backward by a training-mode call)
    with torch.inference_mode(False):
        plan = GraphPlan(dst_index, dim_size, gather_index, n_src, validate=validate)
    # the strong references below pin the storages the key's data_ptr()s refer to
    _CACHE[key] = (plan, dst_index, gather_index)
    while len(_CACHE) > _CACHE_SIZE:
        _CACHE.popitem(last=False)
    return plan


def clear_plan_cache():
    _CACHE.clear()
    _IDX_CACHE.clear()
    _MEMO.clear()


def plan_cache_stats():
    return dict(_STATS, size=len(_CACHE))


# --------------------------------------------------------------------------- int32 gather indices
_IDX_CACHE: "OrderedDict[tuple, tuple]" = OrderedDict()


def get_index32(index: torch.Tensor, limit: int) -> torch.Tensor:
    """cached, range-checked int32 copy of an int64 gather index (``graph[0]`` ...) for the
    fused MLP kernel; same identity/version keying as ``get_plan``."""
    if not index.is_cuda or index.dtype != torch.int64 or index.dim() != 1:
        raise RuntimeError("get_index32: index must be a 1-D int64 HIP tensor")
    key = (_key(index), int(limit))
    cacheable = key[0] is not _UNCACHEABLE
    hit = _IDX_CACHE.get(key) if cacheable else None
    if hit is not None:
        _IDX_CACHE.move_to_end(key)
        return hit[0]
    # reuse a destination plan's dst32 if one exists for this index (same contents)
    for (k_dst, k_n, k_g, _), (plan, _, _) in (_CACHE.items() if cacheable else ()):
        if k_dst == key[0] and k_g is None and k_n == int(limit):
            out = plan.dst32[:index.numel()] if index.numel() else plan.dst32[:0]
            _IDX_CACHE[key] = (out, index)
            return out
    lib = _lib.load()
    M = int(index.numel())
    with torch.inference_mode(False):      # cached: must be a normal tensor (see get_plan)
        out = torch.empty(max(M, 1), dtype=torch.int32, device=index.device)
        err = torch.zeros(1, dtype=torch.int32, device=index.device)
    with torch.cuda.device(index.device):
        idx_c = index.contiguous()
        _lib.check(lib.hgnn_index_to_i32(_lib.ptr(idx_c), M, int(limit), _lib.ptr(out), _lib.ptr(err),
                                         _lib.current_stream(index.device)), "hgnn_index_to_i32")
    if int(err.item()) != 0:
        raise RuntimeError(f"gather index out of range for a table of {limit} rows")
    out = out[:M]
    if not cacheable:
        return out
    _IDX_CACHE[key] = (out, index)
    while len(_IDX_CACHE) > _CACHE_SIZE:
        _IDX_CACHE.popitem(last=False)
    return out


# --------------------------------------------------------------------------- derived index tensors
_MEMO: "OrderedDict[tuple, tuple]" = OrderedDict()


def memo(index: torch.Tensor, tag: str, make):
    """``make()`` once per (index tensor identity/version, tag): per-event index tensors derived from
    the caller's graph (the directed graph, its destination-sorted copy, the inverse permutation).
    Handing back the SAME derived tensors on every call is what lets their plans hit the cache above
    -- and what makes a whole forward capturable into a HIP graph (no sort, no plan build on replay)."""
    key = (_key(index), tag)
    if key[0] is _UNCACHEABLE:
        with torch.inference_mode(False):
            return make()
    hit = _MEMO.get(key)
    if hit is not None:
        _MEMO.move_to_end(key)
        return hit[0]
    with torch.inference_mode(False):      # cached: normal tensors (see get_plan)
        out = make()
    _MEMO[key] = (out, index)
    while len(_MEMO) > 4 * _CACHE_SIZE:
        _MEMO.popitem(last=False)
    return out
