"""Fused gather -> concat -> [Linear -> LayerNorm -> act] x {2,3} -> (+skip) on MFMA:
``hgnn_mlp_forward_f32`` (csrc/mlp_fused.hip, fp32 rows) and ``hgnn_mlp_forward_bf16``
(csrc/mlp_fused_bf16.hip, bf16 rows).  Covers the cell networks, the encoders (small-K mode)
and the width-1 classifier heads.

``supported`` decides per call; when it says no, ``concat_mlp`` evaluates the same
Sequential with HIP row gathers + library GEMMs (still on the GPU).  The fused kernel is
used whenever autograd is not recording: inference, and the first (no-grad) pass of every
reentrant ``torch.utils.checkpoint`` segment -- which is how the reference runs all of its
updates (Modules/gnn_utils.py:14-15).  When autograd records, the differentiable variant runs
(``_FusedMLPTrain``: the same kernel dumping the pre-LayerNorm outputs, backward = HIP
LayerNorm/activation row kernels + library GEMMs + segmented-reduce scatter); shapes it does not
cover take the library path.
"""
from __future__ import annotations

import contextlib
import contextvars
import ctypes
import os
import weakref
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import _lib
from .plan import get_index32

_ACT = {nn.GELU: 1, nn.Tanh: 2, nn.ReLU: 3}
_enabled = True
# The differentiable variant (kernel forward + hand-written backward).  With ATen elementwise kernels in
# the backward it was 2 % slower than autograd through the library path; with the one-pass HIP
# LayerNorm/activation kernels (csrc/ln_act.hip) and the first-layer gradients factored through a
# segment-reduced dz it is faster (L=256, M=2M cell step: 58 vs 122 ms checkpointed, 45 vs 92 ms
# without checkpointing) and is the default when autograd records.
_train_enabled = True
stats = {"fused_calls": 0, "fused_train_calls": 0}

# Dispatch switches.  The module-level ``_name`` variables below are the PROCESS DEFAULTS (what the ``set_*`` functions
# and the HGNN_* environment variables change).  ``options(...)`` overrides any of them for the current context only --
# a ``contextvars`` context, i.e. per thread / per asyncio task -- so two models served from two threads can run
# different arithmetic (``with fused.options(fp32_split3=False): model(x, g)``) without touching shared state.
_ctx_options = contextvars.ContextVar("hgnn_fused_options", default=None)
_ctx_training_forward = contextvars.ContextVar("hgnn_fused_training_forward", default=0)
_OPTION_NAMES = ("enabled", "train_enabled", "preproject", "preproject_bf16", "fp32_split3", "fp32_split3_train",
                 "bf16_split", "train_bf16_enabled", "wgrad_hip", "bwd_fused")


def _opt(name: str):
    o = _ctx_options.get()
    if o is not None and name in o:
        return o[name]
    return globals()["_" + name]


@contextlib.contextmanager
def options(**overrides):
    """Context-local dispatch overrides: ``enabled``, ``train_enabled``, ``preproject``, ``preproject_bf16``,
    ``fp32_split3``, ``fp32_split3_train``, ``bf16_split``, ``train_bf16_enabled``, ``wgrad_hip``, ``bwd_fused``.
    Nested contexts stack; nothing outside the ``with`` block (or in another thread) sees the change.
    NOTE for training: autograd runs the backward (and the recompute of reentrant checkpoints) on its own worker
    thread, which does NOT inherit this context -- switches that must hold during ``backward()`` (``fp32_split3_train``,
    ``train_enabled``, ``train_bf16_enabled``, ``wgrad_hip``, ``bwd_fused``) are to be set as process defaults
    (``set_*`` functions / HGNN_* environment variables); ``options`` is for inference-side choices."""
    bad = set(overrides) - set(_OPTION_NAMES)
    if bad:
        raise TypeError(f"fused.options: unknown option(s) {sorted(bad)}; known: {_OPTION_NAMES}")
    merged = dict(_ctx_options.get() or {})
    merged.update(overrides)
    token = _ctx_options.set(merged)
    try:
        yield
    finally:
        _ctx_options.reset(token)


def set_enabled(flag: bool, train: bool = None):
    """switch the fused kernels off/on (A/B measurements, debugging); `train` separately switches the
    differentiable variant (default: follows `flag`)"""
    global _enabled, _train_enabled
    _enabled = bool(flag)
    _train_enabled = bool(train) if train is not None else bool(flag)


def _parse(net: nn.Sequential):
    """[(Linear, LayerNorm or None, act_code)] or None if the Sequential is not
    (Linear -> LayerNorm -> act) repeated, optionally ending in one plain Linear (a head)"""
    mods = list(net)
    layers = []
    i = 0
    while i < len(mods):
        lin = mods[i]
        if not isinstance(lin, nn.Linear) or lin.bias is None:
            return None
        if i + 1 == len(mods):                       # plain last layer (output_activation=None)
            layers.append((lin, None, 0))
            break
        if i + 2 >= len(mods):
            return None
        ln, act = mods[i + 1], mods[i + 2]
        if not isinstance(ln, nn.LayerNorm) or type(act) not in _ACT:
            return None
        if isinstance(act, nn.GELU) and getattr(act, "approximate", "none") != "none":
            return None
        if not ln.elementwise_affine or ln.bias is None:
            return None
        layers.append((lin, ln, _ACT[type(act)]))
        i += 3
    return layers or None


_preproject = True
# bf16 split kernel: implemented and parity-tested, but measured SLOWER (L=256: 3.7 vs 2.7 ms; L=512: 9.6 vs
# 8.6 ms): the accumulator-layout gather of bf16 P rows is 32-byte pieces issued at the start of every
# 64-row tile, which costs more than the 2/3 of the first layer's MFMAs it saves at 16x the matrix rate.
_preproject_bf16 = None   # bf16 feature-split kernel: None = by shape (first-layer width >= 512, i.e. latent >= 256:
                          # kernel 2.33-2.42 vs 2.70-2.80 ms at latent 256, 7.1 vs 8.6 ms at latent 512 incl. the
                          # projection GEMMs; latent 128 is 7 % slower projected), True / False = forced (A/B, tests)


def set_preproject(flag: bool) -> None:
    """A/B switch: project gathered segments through their block of the first Linear before the kernel"""
    global _preproject
    _preproject = bool(flag)


def _projected_segments(segments, M):
    """which gathered segments to pre-project (hgnn_mlp_desc.n_pre): table[idx] enters the first Linear
    linearly, W_s table[idx[e]] = (table W_s^T)[idx[e]], so a table with few rows (nodes: N = M / 16.7) is
    projected once by an N-row GEMM and only gathered inside the kernel -- the edge network's first layer
    keeps K = L of its 3L input columns.  At least one segment must stay in the kernel's K loop."""
    if not _opt("preproject") or len(segments) < 2 or any(int(t.shape[1]) % 16 for t, _ in segments):
        return []
    cand = [i for i, (t, idx) in enumerate(segments) if idx is not None and 4 * int(t.shape[0]) <= M]
    cand = cand[:2]
    if len(cand) == len(segments):
        cand = cand[:-1]
    return cand


def _pad_rows(t: torch.Tensor, rows: int) -> torch.Tensor:
    """zero-pad the leading dimension to `rows` (padded parameters of a partial last layer)"""
    out = torch.zeros((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    out[:t.shape[0]] = t
    return out


def _descriptor(net, segments, skip, dry=False, split_proj=False):
    """(descriptor, keep-alive list, M, n_out) for hgnn_mlp_forward_f32, or None.  ``dry``: only decide
    supportability (no projection GEMMs are run; the descriptor must not be launched).  ``split_proj``: the caller will
    run the split-bf16 kernel on this descriptor: the N-row projection GEMMs of the gathered segments use the same
    arithmetic (hgnn_linear_f32_split3, four products) instead of the library's fp32 GEMM."""
    layers = _parse(net)
    if layers is None or len(layers) not in (1, 2, 3) or not (1 <= len(segments) <= 3):
        return None
    if any(ln is None for _, ln, _ in layers[:-1]):
        return None
    d = _lib.HgnnMlpDesc()
    keep = []
    M = None
    for table, index in segments:
        if table.dim() != 2 or not table.is_cuda or table.dtype != torch.float32:
            return None
        rows = int(index.numel()) if index is not None else int(table.shape[0])
        if M is None:
            M = rows
        elif M != rows:
            return None
    K_full = sum(int(t.shape[1]) for t, _ in segments)
    lin0 = layers[0][0]
    if lin0.in_features != K_full:
        return None
    proj = _projected_segments(segments, M) if layers[0][1] is not None else []
    split_P = {}
    if split_proj and proj and not dry:
        # the pre-projections in the consuming kernel's own arithmetic; both segments of one table (nodes[graph[0]],
        # nodes[graph[1]]) in one launch
        cols, c0 = {}, 0
        for i, (t, _) in enumerate(segments):
            cols[i] = (c0, c0 + int(t.shape[1]))
            c0 += int(t.shape[1])
        t0 = segments[proj[0]][0]
        same = len(proj) == 2 and segments[proj[1]][0].data_ptr() == t0.data_ptr() \
            and segments[proj[1]][0].shape == t0.shape and t0.is_contiguous() and segments[proj[1]][0].is_contiguous()
        groups = [proj] if same else [[i] for i in proj]
        for grp in groups:
            res = _split3_project(segments[grp[0]][0].detach(), lin0.weight, [cols[i] for i in grp])
            if res is not None:
                split_P.update(dict(zip(grp, res)))
    col = 0
    kept_cols = []
    n_kept = n_pre = 0
    for i, (table, index) in enumerate(segments):
        t = table if table.is_contiguous() else table.contiguous()
        keep.append(t)
        w = int(t.shape[1])
        i32 = None
        if index is not None:
            i32 = get_index32(index, int(t.shape[0]))
            keep.append(i32)
        if i in proj:
            if dry:
                P = t
            else:
                P = split_P.get(i)
                if P is None:
                    with torch.autocast("cuda", enabled=False):   # the kernel reads P as fp32
                        P = torch.matmul(t.detach(), lin0.weight.detach()[:, col:col + w].t())   # [rows, H]
                keep.append(P)
            d.pre_table[n_pre] = P.data_ptr()
            d.pre_index[n_pre] = i32.data_ptr() if i32.numel() else None
            n_pre += 1
        else:
            d.seg_table[n_kept] = t.data_ptr()
            d.seg_width[n_kept] = w
            d.seg_index[n_kept] = (i32.data_ptr() if i32.numel() else None) if i32 is not None else None
            kept_cols.append((col, col + w))
            n_kept += 1
        col += w
    d.n_seg, d.n_pre = n_kept, n_pre
    n = len(layers)
    d.n_layers = n
    K = sum(c1 - c0 for c0, c1 in kept_cols)
    d.width[0] = K
    eps = None
    for l, (lin, ln, act) in enumerate(layers):
        if l > 0 and lin.in_features != d.width[l]:
            return None
        params = [lin.weight, lin.bias] + ([ln.weight, ln.bias] if ln is not None else [])
        for p in params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                return None
        W, b = lin.weight, lin.bias
        if l == 0 and n_pre:
            # the kernel's K loop only sees the columns of the segments that stayed
            W = W.detach() if dry else torch.cat([W.detach()[:, c0:c1] for c0, c1 in kept_cols], dim=1).contiguous()
            keep.append(W)
        if l == 0 and any(int(t.shape[1]) % 16 for t, _ in segments):
            if K > 16:
                return None
            # small-K mode (encoders, K = 3 / 6): one zero-padded 16-column chunk
            W = torch.nn.functional.pad(W.detach(), (0, 16 - K)).contiguous()
            keep.append(W)
            d.w0_cols = 16
        lnw_t, lnb_t = (ln.weight, ln.bias) if ln is not None else (None, None)
        o_l = int(lin.out_features)
        if l == n - 1 and ln is None:
            if o_l > 32:
                return None
            # head (width-1 classifiers, emb_dim-wide embedding head): the plain last layer is stored
            # zero-padded as 32 rows
            W, b = _pad_rows(W.detach(), 32), _pad_rows(b.detach(), 32)
            keep += [W, b]
            d.w_last_rows = 32
        elif l == n - 1 and n == 3 and o_l not in (32, 64, 128, 256) and layers[0][0].out_features % 2 == 0:
            # narrower than its tile row (supernode encoder, L - emb_dim outputs): Linear / LayerNorm
            # parameters zero-padded to P = hidden / 2 rows; the kernel normalises over the real width
            P = layers[0][0].out_features // 2
            if not (P - 16 < o_l < P) or o_l % 4:
                return None
            W, b = _pad_rows(W.detach(), P), _pad_rows(b.detach(), P)
            lnw_t, lnb_t = _pad_rows(ln.weight.detach(), P), _pad_rows(ln.bias.detach(), P)
            keep += [W, b, lnw_t, lnb_t]
            d.w_last_rows = P
        d.W[l], d.b[l] = W.data_ptr(), b.data_ptr()
        if ln is not None:
            d.ln_w[l], d.ln_b[l] = lnw_t.data_ptr(), lnb_t.data_ptr()
            if eps is None:
                eps = ln.eps
            elif eps != ln.eps:
                return None
        else:
            d.ln_w[l] = d.ln_b[l] = None
        d.width[l + 1] = lin.out_features
        d.act[l] = act
    d.ln_eps = float(eps)
    n_out = int(d.width[n])
    if skip is not None:
        if tuple(skip.shape) != (M, n_out) or not skip.is_cuda or skip.dtype != torch.float32:
            return None
        sk = skip if skip.is_contiguous() else skip.contiguous()
        keep.append(sk)
        d.skip = sk.data_ptr()
    else:
        d.skip = None
    d.M = M
    return d, keep, M, n_out


def _kslot_perm(k: int, device) -> torch.Tensor:
    """column order of W_{l>=1} for the bf16 kernel: slot 32c+8g+j <- feature 32c+(j<4 ? 4g+j : 16+4g+j-4)"""
    c = torch.arange(k // 32, device=device).view(-1, 1, 1) * 32
    g = torch.arange(4, device=device).view(1, -1, 1)
    j = torch.arange(8, device=device).view(1, 1, -1)
    feat = c + torch.where(j < 4, 4 * g + j, 16 + 4 * g + j - 4)
    return feat.reshape(-1)


def _fragment_order(W: torch.Tensor) -> torch.Tensor:
    """bf16 W[F, K] in MFMA A-fragment order (include/hgnn_hip.h, hgnn_mlp_forward_bf16_split):
    [k-chunk c][16-feature tile T][lane = 16*(k-group) + row][8 values]"""
    F, K = W.shape
    return W.view(F // 16, 16, K // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous()


class _WeightCache:
    """prepared (re-laid-out / bf16 / split) copies of Linear weights, one per (weight object, layout).

    Validity = same Parameter OBJECT (held by weak reference: a deleted model frees its entries), same storage
    address and same in-place version counter.  ``optimizer.step()``, ``load_state_dict`` and every other in-place op
    on the parameter bump that counter.  Writes through ``param.data`` (``p.data.normal_()``, hand-written
    ``p.data -= lr * g``, EMA / SWA swaps) do NOT -- PyTorch gives ``.data`` a version counter of its own -- so:
      * the Sequentials built by ``utils.make_mlp`` drop their entries in ``train()`` / ``eval()`` / ``to()`` /
        ``load_state_dict`` (``FusedMLPSequential``), and
      * code that edits ``.data`` between two forwards without any of these calls ``fused.clear_weight_cache()``.
    """

    def __init__(self):
        self._d = {}

    def get(self, weight, tag):
        try:
            ver = weight._version
        except RuntimeError:              # inference tensors carry no version counter: never cached
            return None
        e = self._d.get((id(weight), tag))
        if e is not None and e[0]() is weight and e[1] == ver and e[2] == weight.data_ptr():
            return e[3]
        return None

    def put(self, weight, tag, value):
        try:
            ver = weight._version
        except RuntimeError:
            return
        key = (id(weight), tag)
        try:
            ref = weakref.ref(weight, lambda _r, k=key, d=self._d: d.pop(k, None))
        except TypeError:
            return
        self._d[key] = (ref, ver, weight.data_ptr(), value)

    def drop(self, weights):
        ids = {id(w) for w in weights}
        for k in [k for k in self._d if k[0] in ids]:
            self._d.pop(k, None)

    def clear(self):
        self._d.clear()

    def __len__(self):
        return len(self._d)


_wcache = _WeightCache()


def clear_weight_cache(module: Optional[nn.Module] = None) -> None:
    """forget the prepared weight copies (of ``module``'s parameters, or all).  Needed only after editing weights
    through ``param.data`` -- every in-place op on the parameter itself is detected (see ``_WeightCache``)."""
    if module is None:
        _wcache.clear()
    else:
        _wcache.drop(list(module.parameters()))


def _prepared_weight(weight, order, kept_cols):
    """bf16 copy of a Linear weight (optionally only the column blocks of the segments that stay in the kernel's K
    loop) in the kernel's fragment order, cached per weight object and version: an inference forward re-lays out
    nothing, a training step once per optimizer update instead of once per call (forward + checkpoint recompute)"""
    tag = (order.__name__, kept_cols)
    hit = _wcache.get(weight, tag)
    if hit is not None:
        return hit
    W = weight.detach()
    if kept_cols is not None:
        W = torch.cat([W[:, c0:c1] for c0, c1 in kept_cols], dim=1)
    W = order(W.to(torch.bfloat16).contiguous())
    _wcache.put(weight, tag, W)     # one prepared copy per weight and layout: a new version replaces the old one
    return W


def _split_forward(d, out, dev):
    """launch the wide-layer bf16 kernel the descriptor's weights were laid out for"""
    _lib.check(_lib.load().hgnn_mlp_forward_bf16_split(ctypes.byref(d), _lib.ptr(out), _lib.current_stream(dev)),
               "hgnn_mlp_forward_bf16_split")


_fp32_split3 = os.environ.get("HGNN_FP32_SPLIT3", "1") != "0"


_fp32_split3_train = os.environ.get("HGNN_FP32_SPLIT3_TRAIN", "0") == "1"


def set_fp32_split3_training(flag: bool) -> None:
    """OPT-IN (HGNN_FP32_SPLIT3_TRAIN=1): the split-bf16 arithmetic also under autograd -- the differentiable forward
    with its pre-LayerNorm dumps (hgnn_mlp_forward_f32_split3), the M-row data gradients dz.W (hgnn_linear_f32_split3,
    four products) and weight gradients dz^T a (hgnn_wgrad_f32_split3) of the backward.  EC-IN fp32 training step at
    latent 256: 883 -> 568 ms, latent 128 285 -> 235 ms, config 3 744 -> 486 ms.  Every gradient bar of the GPU suite
    holds with it on (269 passed), normwise with >= 6x margin, but the ELEMENT-WISE 1e-4 bar only by 3-7 % on two
    fixtures (9.3e-5 on one weight gradient of the latent-256 HGNN cell, 7.8e-5 on config 2's input gradient): by
    default gradients are therefore computed on the exact fp32 path and only no-grad forwards (inference, and the
    no-grad pass of a reentrant checkpoint) use split-bf16."""
    global _fp32_split3_train
    _fp32_split3_train = bool(flag)


def set_fp32_split3(flag: bool) -> None:
    """The forwards of the fp32 MLPs at latent 128 / 256 (node / edge / supernode / superedge networks, the hidden layers
    of the score heads; inference and the forward passes of a training step) evaluate their GEMMs as split-bf16
    products on the bf16 matrix pipe (hgnn_mlp_forward_f32_split3: hi.hi + mid.hi + hi.mid, exact products, fp32
    accumulation; rows, LayerNorm, activations, skip in fp32): <= 1.6e-5 against the reference at every checked stage
    of BASELINE configs 2 and 3, inside north_star's 1e-4, and the whole GPU parity suite holds with it on.
    DEFAULT ON for no-grad forwards (set_fp32_split3_training switches it on under autograd too); ``set_fp32_split3(False)``, ``HGNN_FP32_SPLIT3=0`` or ``hparams["fp32_gemm"] = "exact"`` select the
    exact fp32 matrix instruction (fmaf-chain arithmetic) instead."""
    global _fp32_split3
    _fp32_split3 = bool(flag)


class training_forward:
    """context: the enclosed no-grad evaluation belongs to a TRAINING step (the first pass of a reentrant
    ``torch.utils.checkpoint`` segment, gnn_utils._maybe_checkpoint).  The split-bf16 evaluation of the fp32 MLPs is an
    INFERENCE default; inside this context a no-grad forward uses exactly what the recompute under autograd will use
    (the exact fp32 kernels, or split-bf16 everywhere with set_fp32_split3_training(True)): forward values and the
    point the gradients are taken at agree bitwise, as with the reference's own recompute.  Context-local
    (``contextvars``): a training step in one thread does not change what an inference call in another thread runs."""

    def __enter__(self):
        self._token = _ctx_training_forward.set(_ctx_training_forward.get() + 1)

    def __exit__(self, *exc):
        _ctx_training_forward.reset(self._token)
        return False


def _split3_on(net) -> bool:
    if _ctx_training_forward.get() > 0 and not _opt("fp32_split3_train"):
        return False
    o = _ctx_options.get()
    if o is not None and "fp32_split3" in o:        # an explicit context override outranks the per-module mark
        return bool(o["fp32_split3"])
    flag = getattr(net, "_hgnn_split3", None)      # per-module override (hparams["fp32_gemm"])
    return _fp32_split3 if flag is None else bool(flag)


def _split3_weight(weight, kept_cols, panels: bool, transpose: bool = False):
    """bf16 split stream of an fp32 Linear weight (per 32-wide k-chunk of the kept columns: W_hi, then W_mid) in
    A-fragment order, cached per weight object and version (``_WeightCache``)"""
    tag = ("split3", kept_cols, panels, transpose)
    hit = _wcache.get(weight, tag)
    if hit is not None:
        return hit
    W = weight.detach().float()
    if kept_cols is not None:
        W = torch.cat([W[:, c0:c1] for c0, c1 in kept_cols], dim=1)
    if transpose:
        W = W.t().contiguous()
    hi = W.to(torch.bfloat16)
    mid = (W - hi.float()).to(torch.bfloat16)
    F, K = W.shape
    # per 32-wide k-chunk: the chunk's W_hi columns, then its W_mid columns (virtual chunks 2c, 2c + 1)
    Wv = torch.stack([hi.view(F, K // 32, 32), mid.view(F, K // 32, 32)], dim=2).reshape(F, 2 * K)
    Wv = _fragment_order(Wv.contiguous())
    _wcache.put(weight, tag, Wv)
    return Wv


def _split3_linear(x: torch.Tensor, weight, cols, net) -> Optional[torch.Tensor]:
    """x [M, K] . W[:, cols]  for a Linear weight W [K, in] (``cols`` = (c0, c1) or None): the data gradient dz . W of
    the fp32 training backward as ONE split-bf16 GEMM (hgnn_linear_f32_split3), or None when the split-bf16 mode is
    off / the shape has no instantiation (the caller then uses the library's fp32 GEMM)"""
    K = int(x.shape[1])
    N = int(weight.shape[1]) if cols is None else cols[1] - cols[0]
    if not (_opt("fp32_split3_train") and _split3_on(net)) or x.dtype != torch.float32 or not x.is_cuda or K % 128 \
            or N not in (256, 512) \
            or int(x.shape[0]) == 0:
        return None
    # the kernel wants the weight of the Linear that maps K -> N, i.e. W[:, cols]^T  [N, K]
    Wv = _split3_weight(weight, None if cols is None else (tuple(cols),), False, transpose=True)
    xc = x if x.is_contiguous() else x.contiguous()
    out = torch.empty((int(x.shape[0]), N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().hgnn_linear_f32_split3(_lib.ptr(xc), int(x.shape[0]), K, _lib.ptr(Wv), N, None,
                                                      _lib.ptr(out), _lib.current_stream(x.device)),
                   "hgnn_linear_f32_split3")
    stats["split3_linear_calls"] = stats.get("split3_linear_calls", 0) + 1
    return out


def _split3_project(table: torch.Tensor, weight, cols_list):
    """[table [R, w] . W[:, cols]^T -> [R, H] for cols in cols_list] (one or two column blocks of the first Linear over
    the SAME table): the pre-projections of the gathered segments (``_projected_segments``) in the split-bf16 arithmetic
    of the kernel that consumes them (three products), ONE launch (hgnn_project_f32_split3); None when the shape has no
    instantiation"""
    R, K = int(table.shape[0]), int(table.shape[1])
    N = int(weight.shape[0])
    if R == 0 or K % 128 or N not in (256, 512) or table.dtype != torch.float32 or not table.is_cuda \
            or not 1 <= len(cols_list) <= 2:
        return None
    # the kernel wants the weight of the Linear that maps K -> N: W[:, cols]  [N, K]
    Wv = [_split3_weight(weight, (tuple(c),), False) for c in cols_list]
    tc = table if table.is_contiguous() else table.contiguous()
    outs = [torch.empty((R, N), dtype=torch.float32, device=table.device) for _ in cols_list]
    with torch.cuda.device(table.device):
        _lib.check(_lib.load().hgnn_project_f32_split3(
            _lib.ptr(tc), R, K, _lib.ptr(Wv[0]), _lib.ptr(Wv[1]) if len(Wv) > 1 else None, N, _lib.ptr(outs[0]),
            _lib.ptr(outs[1]) if len(outs) > 1 else None, _lib.current_stream(table.device)), "hgnn_project_f32_split3")
    stats["split3_project_calls"] = stats.get("split3_project_calls", 0) + 1
    return outs


def _split3_applies(net, segments, training: bool = False) -> bool:
    """the opt-in split-bf16 path is switched on for this network and hgnn_mlp_forward_f32_split3 has its shape
    (``training``: the call comes from the differentiable forward, which dumps the pre-LayerNorm rows)"""
    if not _split3_on(net) or (training and not _opt("fp32_split3_train")):
        return False
    if not training and torch.is_grad_enabled() and any(p.requires_grad for p in net.parameters()):
        return False
    layers = _parse(net)
    if layers is None or len(layers) not in (2, 3) or any(ln is None for _, ln, _ in layers):
        return False
    if any(t.dtype != torch.float32 or int(t.shape[1]) % 128 for t, _ in segments):
        return False
    o = layers[-1][0].out_features
    head_body = len(layers) == 2 and layers[0][0].out_features == o and o in (256, 512)
    return head_body or (o in (128, 256) and all(lin.out_features == 2 * o for lin, _, _ in layers[:-1]))


def _try_split3(net, segments, d, keep, training: bool = False):
    """re-point a ready fp32 descriptor at split-3 weight streams if the opt-in path supports its shape"""
    if not _split3_applies(net, segments, training) or int(d.w0_cols) != 0 or int(d.w_last_rows) != 0:
        return False
    layers = _parse(net)
    kept = None
    if int(d.n_pre):
        # the columns of the segments that stayed in the kernel's K loop, in order
        col, kept, proj = 0, [], _projected_segments(segments, int(d.M))
        for i, (t, _) in enumerate(segments):
            w = int(t.shape[1])
            if i not in proj:
                kept.append((col, col + w))
            col += w
        kept = tuple(kept)
    for l, (lin, _, _) in enumerate(layers):
        W = _split3_weight(lin.weight, kept if l == 0 else None, l == 0)
        keep.append(W)
        d.W[l] = W.data_ptr()
    if not bool(_lib.load().hgnn_mlp_supported_f32_split3(ctypes.byref(d))):
        raise RuntimeError("fused_concat_mlp: split-3 descriptor rejected")
    return True


_bf16_split = True


def set_bf16_split(flag: bool) -> None:
    """use the feature-split bf16 kernel for the wide layers (L >= 128) when it supports the shape"""
    global _bf16_split
    _bf16_split = bool(flag)


def _wants_split(net, segments) -> bool:
    layers = _parse(net)
    if not _opt("bf16_split") or layers is None or len(layers) not in (1, 2, 3):
        return False
    if any(int(t.shape[1]) % 128 for t, _ in segments):
        return False
    widths = [lin.out_features for lin, _, _ in layers]
    o = widths[-1]
    if len(layers) == 1:
        return layers[0][1] is not None and o in (256, 512, 1024)  # single layers (chains)
    return o in (128, 256, 512) and all(w == 2 * o for w in widths[:-1])


def _descriptor_bf16(net, segments, skip, split=False, dry=False):
    """descriptor for hgnn_mlp_forward_bf16 (bf16 rows, bf16 slot-ordered weights, fp32 bias / LayerNorm)
    or, with ``split``, for hgnn_mlp_forward_bf16_split (weights in A-fragment order; gathered segments
    of small tables pre-projected as in the fp32 path, P_s rounded once to bf16)"""
    layers = _parse(net)
    if layers is None or len(layers) not in (1, 2, 3) or not (1 <= len(segments) <= 3):
        return None
    if any(ln is None for _, ln, _ in layers):
        return None
    d = _lib.HgnnMlpDesc()
    keep = []
    M = None
    for table, index in segments:
        if table.dim() != 2 or not table.is_cuda or table.dtype != torch.bfloat16:
            return None
        rows = int(index.numel()) if index is not None else int(table.shape[0])
        if M is None:
            M = rows
        elif M != rows:
            return None
    lin0 = layers[0][0]
    if lin0.in_features != sum(int(t.shape[1]) for t, _ in segments) or not lin0.weight.is_cuda:
        return None
    want_pre = _opt("preproject_bf16")
    if want_pre is None:
        widths = [lin.out_features for lin, _, _ in layers]
        want_pre = len(layers) > 1 and lin0.out_features >= 512
    proj = _projected_segments(segments, M) if (split and want_pre) else []
    col = n_kept = n_pre = 0
    kept_cols = []
    for i, (table, index) in enumerate(segments):
        t = table if table.is_contiguous() else table.contiguous()
        keep.append(t)
        w = int(t.shape[1])
        i32 = None
        if index is not None:
            i32 = get_index32(index, int(t.shape[0]))
            keep.append(i32)
        if i in proj:
            if dry:
                P = t
            else:
                with torch.autocast("cuda", enabled=False):
                    # bf16 operands, fp32 accumulation, one rounding: the arithmetic the kernel's own MFMAs would do
                    P = torch.matmul(t, lin0.weight.detach()[:, col:col + w].to(torch.bfloat16).t())
                keep.append(P)
            d.pre_table[n_pre] = P.data_ptr()
            d.pre_index[n_pre] = i32.data_ptr() if i32.numel() else None
            n_pre += 1
        else:
            d.seg_table[n_kept] = t.data_ptr()
            d.seg_width[n_kept] = w
            d.seg_index[n_kept] = (i32.data_ptr() if i32.numel() else None) if i32 is not None else None
            kept_cols.append((col, col + w))
            n_kept += 1
        col += w
    d.n_seg, d.n_pre = n_kept, n_pre
    n = len(layers)
    d.n_layers = n
    d.width[0] = sum(c1 - c0 for c0, c1 in kept_cols)
    eps = None
    order = _fragment_order
    for l, (lin, ln, act) in enumerate(layers):
        if (l > 0 and lin.in_features != d.width[l]) or not lin.weight.is_cuda:
            return None
        W = lin.weight.detach()
        if split:
            k_in = sum(c1 - c0 for c0, c1 in kept_cols) if (l == 0 and n_pre) else W.shape[1]
            if k_in % 32 or lin.out_features % 64:
                return None
            if not dry:
                W = _prepared_weight(lin.weight, order, tuple(kept_cols) if (l == 0 and n_pre) else None)
        else:
            if l > 0:
                if lin.in_features % 32:
                    return None
                W = W[:, _kslot_perm(lin.in_features, W.device)]
            W = W.to(torch.bfloat16).contiguous()
        small = [p.detach().float().contiguous() for p in (lin.bias, ln.weight, ln.bias)]
        keep += [W] + small
        d.W[l], d.b[l], d.ln_w[l], d.ln_b[l] = W.data_ptr(), small[0].data_ptr(), small[1].data_ptr(), small[2].data_ptr()
        d.width[l + 1] = lin.out_features
        d.act[l] = act
        if eps is None:
            eps = ln.eps
        elif eps != ln.eps:
            return None
    d.ln_eps = float(eps)
    n_out = int(d.width[n])
    if skip is not None:
        if tuple(skip.shape) != (M, n_out) or not skip.is_cuda or skip.dtype != torch.bfloat16:
            return None
        sk = skip if skip.is_contiguous() else skip.contiguous()
        keep.append(sk)
        d.skip = sk.data_ptr()
    else:
        d.skip = None
    d.M = M
    return d, keep, M, n_out


def _layer_chain(net):
    """fp32 MLPs too wide for one launch (latent 512: a 1024-wide hidden layer is 256 accumulators per lane): the
    [Linear, LayerNorm, act] triples as single-layer Sequentials sharing the parameters (a plain last Linear -- the
    width-1 heads -- stays a trailing ``nn.Linear``: an M x 1024 x 1 product), or None"""
    layers = _parse(net)
    if layers is None or len(layers) < 2 or any(ln is None for _, ln, _ in layers[:-1]):
        return None
    fused_layers = layers if layers[-1][1] is not None else layers[:-1]
    if any(lin.out_features not in (512, 1024) or lin.in_features % 16 for lin, _, _ in fused_layers[1:]) \
            or fused_layers[0][0].out_features not in (512, 1024):
        return None
    mods = list(net)
    chain = [nn.Sequential(*mods[3 * i:3 * i + 3]) for i in range(len(fused_layers))]
    if len(fused_layers) < len(layers):
        chain.append(layers[-1][0])                      # the plain nn.Linear itself
    return chain


def _chain_supported(net, segments, skip) -> bool:
    chain = _layer_chain(net)
    if chain is None:
        return False
    bf16 = _is_bf16(segments)
    try:
        if bf16:
            if not _wants_split(chain[0], segments):
                return False
            first = _descriptor_bf16(chain[0], segments, None, split=True, dry=True)
            ok = first is not None and bool(_lib.load().hgnn_mlp_supported_bf16_split(ctypes.byref(first[0])))
        else:
            first = _descriptor(chain[0], segments, None, dry=True)
            ok = first is not None and bool(_lib.load().hgnn_mlp_supported(ctypes.byref(first[0])))
    except RuntimeError:
        return False
    if not ok:
        return False
    last = _parse(net)[-1][0]
    if isinstance(chain[-1], nn.Linear) and skip is not None:
        return False                                      # heads have no skip connection
    dt = torch.bfloat16 if bf16 else torch.float32
    return skip is None or (skip.is_cuda and skip.dtype == dt and tuple(skip.shape) == (first[2], last.out_features))


def _is_bf16(segments) -> bool:
    return all(t.dtype == torch.bfloat16 for t, _ in segments)


def supported(net, segments, skip, allow_chain: bool = True) -> bool:  # noqa: C901
    """``allow_chain``: accept fp32 MLPs that run as one launch per layer (latent 512); a caller that has a cheaper
    alternative for them (bf16 tail of the encoders in bf16 mode) passes False"""
    if not _opt("enabled"):
        return False
    if _is_bf16(segments):
        if torch.is_grad_enabled():
            tensors = [t for t, _ in segments] + ([skip] if skip is not None else []) + list(net.parameters())
            if any(t.requires_grad for t in tensors):
                return False
        split = _wants_split(net, segments)
        try:
            desc = _descriptor_bf16(net, segments, skip, split, dry=True)
        except RuntimeError:
            return False
        lib = _lib.load()
        if desc is not None:
            if split and bool(lib.hgnn_mlp_supported_bf16_split(ctypes.byref(desc[0]))):
                return True
            if not split and bool(lib.hgnn_mlp_supported_bf16(ctypes.byref(desc[0]))):
                return True
        return allow_chain and _chain_supported(net, segments, skip)
    if torch.is_grad_enabled():
        tensors = [t for t, _ in segments] + ([skip] if skip is not None else []) + list(net.parameters())
        if any(t.requires_grad for t in tensors):
            return False
    try:
        desc = _descriptor(net, segments, skip, dry=True)
    except RuntimeError:
        return False
    if desc is not None and bool(_lib.load().hgnn_mlp_supported(ctypes.byref(desc[0]))):
        return True
    return allow_chain and _chain_supported(net, segments, skip)


def _split3_head(net, segments, skip) -> bool:
    """score heads (K -> H -> H -> w, plain last layer; IN.py:107-115, HGNN_GMM.py:313-321) under the opt-in
    split-bf16 mode: the two LayerNorm'ed hidden layers run on hgnn_mlp_forward_f32_split3, the plain last Linear is
    a trailing matrix-vector product over the hidden rows"""
    if not _split3_on(net) or skip is not None:
        return False
    layers = _parse(net)
    if layers is None or len(layers) != 3 or layers[2][1] is not None or layers[0][1] is None or layers[1][1] is None:
        return False
    h = layers[0][0].out_features
    if h not in (256, 512) or layers[1][0].out_features != h or len(list(net)) != 7:
        return False
    return all(t.dtype == torch.float32 and int(t.shape[1]) % 128 == 0 for t, _ in segments)


def _out_buffer(out, M, n_out, dtype, dev):
    """caller-supplied output rows (a contiguous [M, n_out] row block, e.g. a slice of a larger edge table) or a
    fresh tensor"""
    if out is None:
        return torch.empty((M, n_out), dtype=dtype, device=dev)
    if tuple(out.shape) != (M, n_out) or out.dtype != dtype or out.device != dev or not out.is_contiguous():
        raise RuntimeError(f"fused_concat_mlp: out must be a contiguous {dtype} [{M}, {n_out}] tensor on {dev}")
    return out


def fused_concat_mlp(net, segments, skip: Optional[torch.Tensor], out: Optional[torch.Tensor] = None):
    """``out``: write the result rows into this tensor (no-grad callers that assemble one edge table from several
    calls -- the interior / boundary split of a sharded edge update -- avoid a 2 GB concatenation); only the
    single-launch paths take it"""
    bf16 = _is_bf16(segments)
    if not bf16 and _split3_head(net, segments, skip):
        mods = list(net)
        body = nn.Sequential(*mods[:6])
        body._hgnn_split3 = True
        hid = fused_concat_mlp(body, segments, None)
        res = torch.nn.functional.linear(hid, mods[6].weight, mods[6].bias)
        return res if out is None else out.copy_(res)
    if len(_parse(net) or []) > 1:
        if bf16:
            sp = _wants_split(net, segments)
            whole = _descriptor_bf16(net, segments, skip, sp, dry=True)
            whole_ok = whole is not None and bool((_lib.load().hgnn_mlp_supported_bf16_split if sp else
                                                   _lib.load().hgnn_mlp_supported_bf16)(ctypes.byref(whole[0])))
        else:
            whole = _descriptor(net, segments, skip, dry=True)
            whole_ok = whole is not None and (bool(_lib.load().hgnn_mlp_supported(ctypes.byref(whole[0])))
                                              or _split3_applies(net, segments))
        if not whole_ok and _chain_supported(net, segments, skip):
            # one launch per layer; the hidden rows make one trip through HBM (fp32 at latent 512)
            chain = _layer_chain(net)
            segs, out_c = segments, None
            for i, sub in enumerate(chain):
                if isinstance(sub, nn.Linear):
                    out_c = torch.nn.functional.linear(out_c, sub.weight.to(out_c.dtype), sub.bias.to(out_c.dtype))
                else:
                    out_c = fused_concat_mlp(sub, segs, skip if i == len(chain) - 1 else None)
                    segs = [(out_c, None)]
            return out_c if out is None else out.copy_(out_c)
    split = bf16 and _wants_split(net, segments)
    desc = _descriptor_bf16(net, segments, skip, split) if bf16 else \
        _descriptor(net, segments, skip, split_proj=_split3_applies(net, segments))
    if desc is None:
        raise RuntimeError("fused_concat_mlp: unsupported arguments (call supported() first)")
    d, keep, M, n_out = desc
    dev = segments[0][0].device
    out = _out_buffer(out, M, n_out, torch.bfloat16 if bf16 else torch.float32, dev)
    if M == 0:
        return out
    lib = _lib.load()
    with torch.cuda.device(dev):
        if split:
            _split_forward(d, out, dev)
        elif bf16:
            _lib.check(lib.hgnn_mlp_forward_bf16(ctypes.byref(d), _lib.ptr(out), _lib.current_stream(dev)),
                       "hgnn_mlp_forward_bf16")
        elif _try_split3(net, segments, d, keep):
            _lib.check(lib.hgnn_mlp_forward_f32_split3(ctypes.byref(d), _lib.ptr(out), _lib.current_stream(dev)),
                       "hgnn_mlp_forward_f32_split3")
            stats["split3_calls"] = stats.get("split3_calls", 0) + 1
        else:
            _lib.check(lib.hgnn_mlp_forward_f32(ctypes.byref(d), _lib.ptr(out), _lib.current_stream(dev)),
                       "hgnn_mlp_forward_f32")
    del keep
    stats["fused_calls"] += 1
    return out


# --------------------------------------------------------------------------- training variant
def _ln_act_forward(z, gamma, beta, act, eps):
    """act(LayerNorm(z)) in one HIP pass (csrc/ln_act.hip)"""
    out = torch.empty_like(z)
    M, W = int(z.shape[0]), int(z.shape[1])
    if M:
        lib = _lib.load()
        fn = lib.hgnn_ln_act_forward_bf16 if z.dtype == torch.bfloat16 else lib.hgnn_ln_act_forward_f32
        with torch.cuda.device(z.device):
            _lib.check(fn(
                _lib.ptr(z), M, W, _lib.ptr(gamma.detach().float().contiguous()),
                _lib.ptr(beta.detach().float().contiguous()),
                int(act), float(eps), _lib.ptr(out), _lib.current_stream(z.device)), "hgnn_ln_act_forward")
    return out


def _ln_act_backward(z, grad_out, gamma, beta, act, eps):
    """(grad_z, grad_gamma, grad_beta, grad_bias) of a = act(LayerNorm(z)), z = x W^T + bias, in one HIP pass"""
    M, W = int(z.shape[0]), int(z.shape[1])
    dz = torch.empty_like(z)
    partials = torch.empty((_lib.LN_ACT_BLOCKS, 3, W), dtype=torch.float32, device=z.device)
    go = grad_out.contiguous().to(z.dtype)
    lib = _lib.load()
    fn = lib.hgnn_ln_act_backward_bf16 if z.dtype == torch.bfloat16 else lib.hgnn_ln_act_backward_f32
    with torch.cuda.device(z.device):
        _lib.check(fn(
            _lib.ptr(z), _lib.ptr(go), M, W, _lib.ptr(gamma.detach().float().contiguous()),
            _lib.ptr(beta.detach().float().contiguous()), int(act), float(eps), _lib.ptr(dz), _lib.ptr(partials),
            _lib.current_stream(z.device)), "hgnn_ln_act_backward")
    sums = partials.sum(dim=0)
    return dz, sums[0], sums[1], sums[2]


def _atb(A: torch.Tensor, B: torch.Tensor, net=None) -> torch.Tensor:
    """A^T B for tall A [n, p], B [n, q] with a small [p, q] result: a weight gradient.  With the split-bf16 mode on
    (``net`` given) and fp32 operands of at least 4096 rows: the hand-written split-K kernel on the bf16 matrix pipe
    (ops.wgrad_f32_split3: ~1.7 ms where the library paths below need 3.6-4 ms at n = 2M).  Otherwise: the library picks
    a small output tile without split-K for these (p*q/1024 workgroups walking all n rows: 1.2 ms for the
    8 GFLOP of n = 120k, ~110 TFLOP/s at n = 2M); a batched product over row blocks plus one sum fills the
    chip (fixed summation order: deterministic).  EC-IN training step 368 -> 306 ms."""
    n = int(A.shape[0])
    if net is not None and _opt("fp32_split3_train") and _split3_on(net) and n >= 4096 and A.dtype == torch.float32 and B.dtype == torch.float32 \
            and A.is_cuda and int(A.shape[1]) % 8 == 0 and int(B.shape[1]) % 8 == 0:
        from .ops import wgrad_f32_split3
        stats["split3_wgrad_calls"] = stats.get("split3_wgrad_calls", 0) + 1
        return wgrad_f32_split3(A, B)
    chunks = max(16, min(256, n // 8192))   # 4096 / 8192 rows per block measured equal, 32768 6 % slower
    c = n // chunks
    if c < 256:
        return A.t() @ B
    main = c * chunks
    out = torch.bmm(A[:main].reshape(chunks, c, A.shape[1]).transpose(1, 2),
                    B[:main].reshape(chunks, c, B.shape[1])).sum(dim=0)
    if main < n:
        out += A[main:].t() @ B[main:]
    return out


def _zero_grads(ctx, tables, params, grad_out, n_seg):
    """backward of an MLP that was called on M == 0 rows (a shard without edges, an empty super graph): every
    parameter and table gradient is zero; no kernel is launched"""
    gt = [torch.zeros_like(t) if ctx.needs_input_grad[3 + i] else None for i, t in enumerate(tables)]
    gs = [grad_out] if ctx.has_skip else []
    return (None, None, None, *gt, *gs, *[torch.zeros_like(p) for p in params])


class _FusedMLPTrain(torch.autograd.Function):
    """Differentiable fused MLP.  Forward = the same MFMA kernel, additionally dumping each layer's
    pre-LayerNorm output z_l (``save_pre``).  Backward is written out by hand: LayerNorm / activation
    are re-derived from z_l with elementwise ATen kernels, data and weight gradients are library
    GEMMs, and the gradients of gathered segments go back through the atomics-free segmented
    reduce.  Compared with autograd through the unfused path this skips the second forward's GEMMs,
    the [M,3L] concat and the gathered copies; with 288 GB of HBM the reference's checkpointing can
    also be switched off (hparams["checkpointing"]=False) and the dumps kept instead."""

    @staticmethod
    def forward(ctx, net, indices, has_skip, *tensors):
        n_seg = len(indices)
        tables = list(tensors[:n_seg])
        skip = tensors[n_seg] if has_skip else None
        params = tensors[n_seg + (1 if has_skip else 0):]
        segments = [(t, i) for t, i in zip(tables, indices)]
        desc = _descriptor(net, segments, skip)
        if desc is None:
            raise RuntimeError("fused_concat_mlp_train: unsupported arguments (call supported_train() first)")
        d, keep, M, n_out = desc
        n = int(d.n_layers)
        head = int(d.w_last_rows) == 32 and _parse(net)[-1][1] is None
        if head:
            n -= 1             # a score head: the LayerNorm'ed hidden layers are differentiated here, the plain last
                               # Linear is applied by the caller on the returned hidden rows (``fused_head_train``)
        dev = tables[0].device
        zs = [torch.empty((M, int(d.width[l + 1])), dtype=torch.float32, device=dev) for l in range(n)]
        for l in range(n):
            d.save_pre[l] = zs[l].data_ptr() if M else None
        out = torch.empty((M, n_out), dtype=torch.float32, device=dev)
        if M:
            with torch.cuda.device(dev):
                if not head and _try_split3(net, segments, d, keep, training=True):
                    # opt-in: the forward (and its dumps) on the split-bf16 kernel; the backward below is unchanged
                    _lib.check(_lib.load().hgnn_mlp_forward_f32_split3(ctypes.byref(d), _lib.ptr(out),
                                                                       _lib.current_stream(dev)),
                               "hgnn_mlp_forward_f32_split3")
                    stats["split3_calls"] = stats.get("split3_calls", 0) + 1
                else:
                    _lib.check(_lib.load().hgnn_mlp_forward_f32(ctypes.byref(d), _lib.ptr(out),
                                                                _lib.current_stream(dev)), "hgnn_mlp_forward_f32")
        del keep
        stats["fused_train_calls"] += 1
        ctx.indices, ctx.has_skip, ctx.n = indices, has_skip, n
        ctx.net = net
        ctx.acts = [int(d.act[l]) for l in range(n)]
        ctx.eps = float(d.ln_eps)
        ctx.save_for_backward(*tables, *params, *zs)
        if head:
            # the hidden rows feeding the plain last Linear: one LayerNorm / activation row pass over the last dump
            lays = _parse(net)
            return _ln_act_forward(zs[n - 1], lays[n - 1][1].weight, lays[n - 1][1].bias, ctx.acts[n - 1], ctx.eps)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        with torch.autocast("cuda", enabled=False):   # fp32 GEMMs whatever the caller's autocast state
            return _FusedMLPTrain._backward(ctx, grad_out.float())

    @staticmethod
    def _backward(ctx, grad_out):
        from .ops import _seg_reduce
        from .plan import get_plan
        n, indices = ctx.n, ctx.indices
        n_seg = len(indices)
        saved = ctx.saved_tensors
        tables = saved[:n_seg]
        params = saved[n_seg:n_seg + 4 * n]
        zs = saved[n_seg + 4 * n:]
        if int(grad_out.shape[0]) == 0:
            return _zero_grads(ctx, tables, params, grad_out, n_seg)
        W = [params[4 * l] for l in range(n)]
        lw = [lin.weight for lin, _, _ in _parse(ctx.net)][:n]   # the Parameter objects themselves (weight-prep cache keys)
        lnw = [params[4 * l + 2] for l in range(n)]
        lnb = [params[4 * l + 3] for l in range(n)]
        aten = torch.ops.aten
        g = grad_out.contiguous()
        hip_rows = all(int(z.shape[1]) in (64, 128, 256, 512, 1024) for z in zs)
        ys, means, rstds, outs = [], [], [], []
        if hip_rows:
            # hidden activations a_l = act(LN(z_l)) for the weight gradients: one HIP pass per layer
            outs = [_ln_act_forward(zs[l], lnw[l], lnb[l], ctx.acts[l], ctx.eps) if l < n - 1 else None
                    for l in range(n)]
        else:
            # re-derive y_l = LN(z_l) (+ statistics) and a_l = act(y_l) with ATen elementwise kernels
            for l in range(n):
                y, mean, rstd = torch.native_layer_norm(zs[l], [zs[l].shape[1]], lnw[l], lnb[l], ctx.eps)
                ys.append(y)
                means.append(mean)
                rstds.append(rstd)
                code = ctx.acts[l]
                if l < n - 1 or code == 2:
                    outs.append(torch.tanh(y) if code == 2 else (torch.nn.functional.gelu(y) if code == 1
                                else (torch.relu(y) if code == 3 else y)))
                else:
                    outs.append(None)
        grads_params = [None] * (4 * n)
        grads_tables = [None] * n_seg
        da = g
        for l in range(n - 1, -1, -1):
            code = ctx.acts[l]
            if hip_rows:
                # act' -> LayerNorm backward -> dgamma / dbeta / dbias in ONE pass over (z_l, da)
                dz, dlw, dlb, dbias = _ln_act_backward(zs[l], da, lnw[l], lnb[l], code, ctx.eps)
                grads_params[4 * l + 1] = dbias
            else:
                if code == 1:
                    dy = aten.gelu_backward(da, ys[l], approximate="none")
                elif code == 2:
                    dy = aten.tanh_backward(da, outs[l])
                elif code == 3:
                    dy = da * (ys[l] > 0)
                else:
                    dy = da
                dz, dlw, dlb = aten.native_layer_norm_backward(dy, zs[l], [zs[l].shape[1]], means[l], rstds[l],
                                                               lnw[l], lnb[l], [True, True, True])
                ys[l] = None
                grads_params[4 * l + 1] = dz.sum(dim=0)
            grads_params[4 * l + 2] = dlw
            grads_params[4 * l + 3] = dlb
            if l > 0:
                grads_params[4 * l] = _atb(dz, outs[l - 1], ctx.net)
                da = _split3_linear(dz, lw[l], None, ctx.net)
                if da is None:
                    da = dz @ W[l]
                outs[l - 1] = None
            else:
                # First layer.  A gathered segment x_s = table[idx] enters linearly, so both of its
                # gradients factor through S = segment_reduce(dz, idx)  ([table rows, H], one HBM pass):
                #   dW[:, s] = dz^T table[idx] = S^T table        (K = table rows instead of M)
                #   d table  = scatter(dz W_s, idx) = S W_s
                # i.e. for nodes[graph[k]] the two M-row GEMMs (and the gathered copy / the [M,K]
                # concat they would read) become two N-row GEMMs: 16x fewer FLOPs at M/N = 16.7.
                # Only a direct segment (the edge rows themselves) keeps M-row GEMMs.
                dW_cols = []
                col = 0
                for s_i in range(n_seg):
                    idx = indices[s_i]
                    tab = tables[s_i].contiguous()
                    w_s = int(tab.shape[1])
                    W_s = W[0][:, col:col + w_s]
                    if idx is not None:
                        S = _seg_reduce(get_plan(idx, int(tab.shape[0])), dz, None, None)
                        dW_cols.append(_atb(S, tab, ctx.net))
                        if ctx.needs_input_grad[3 + s_i]:
                            grads_tables[s_i] = S @ W_s
                        del S
                    else:
                        dW_cols.append(_atb(dz, tab, ctx.net))
                        if ctx.needs_input_grad[3 + s_i]:
                            gt = _split3_linear(dz, lw[0], (col, col + w_s), ctx.net)
                            grads_tables[s_i] = gt if gt is not None else dz @ W_s
                    col += w_s
                grads_params[0] = dW_cols[0] if n_seg == 1 else torch.cat(dW_cols, dim=1)
        grad_skip = [g] if ctx.has_skip else []
        return (None, None, None, *grads_tables, *grad_skip, *grads_params)


def supported_train(net, segments, skip) -> bool:
    """differentiable fused path: cell networks (LayerNorm on every layer, 16-aligned segments); bf16 rows:
    the feature-split kernel's shapes at latent 128 / 256 (``_FusedMLPTrainBf16``)"""
    if not _opt("train_enabled") or not torch.is_grad_enabled():
        return False
    if _is_bf16(segments):
        if not _opt("train_bf16_enabled") or not _wants_split(net, segments):
            return False
        layers = _parse(net)
        if any(lin.out_features not in (64, 128, 256, 512, 1024) for lin, _, _ in layers):
            return False                              # widths of the bf16 LayerNorm / activation row kernels
        try:
            desc = _descriptor_bf16(net, segments, skip, split=True, dry=True)
        except RuntimeError:
            return False
        return desc is not None and bool(_lib.load().hgnn_mlp_supported_bf16_split(ctypes.byref(desc[0])))
    try:
        desc = _descriptor(net, segments, skip, dry=True)
    except RuntimeError:
        return False
    if desc is None:
        return False
    d = desc[0]
    # narrow encoders (zero-padded LayerNorm'ed last layer) have no dumps; the small-K encoders (w0_cols = 16: the
    # kernel-side zero padding of W[0]) do -- their backward uses the unpadded parameters; score heads (plain last
    # layer stored as 32 rows) dump their two hidden layers (``fused_head_train``)
    if int(d.w_last_rows) != 0 and not _is_head(net):
        return False
    if _is_head(net) and (skip is not None or any(int(d.width[l + 1]) not in (64, 128, 256, 512) for l in range(2))):
        return False                                   # widths of the LayerNorm / activation row kernels
    return bool(_lib.load().hgnn_mlp_supported(ctypes.byref(d)))


def _is_head(net) -> bool:
    layers = _parse(net)
    return layers is not None and len(layers) == 3 and layers[2][1] is None and layers[0][1] is not None \
        and layers[1][1] is not None and layers[2][0].out_features <= 32


def fused_concat_mlp_train(net, segments, skip: Optional[torch.Tensor]):
    layers = _parse(net)
    if not _is_bf16(segments) and _is_head(net):
        # score heads (IN.py:107-115,126-127; HGNN_GMM.py:313-321,342-344) under autograd: the two LayerNorm'ed hidden
        # layers on the differentiable fused kernel (dumps + hand-written backward), the plain last Linear -- an
        # [M, H] x [H, w<=32] product -- applied to the returned hidden rows
        body = layers[:2]
        params = []
        for lin, ln, _ in body:
            params += [lin.weight, lin.bias, ln.weight, ln.bias]
        hidden = _FusedMLPTrain.apply(net, tuple(i for _, i in segments), False, *[t for t, _ in segments], *params)
        last = layers[2][0]
        stats["fused_head_train_calls"] = stats.get("fused_head_train_calls", 0) + 1
        return torch.nn.functional.linear(hidden, last.weight, last.bias)
    params = []
    for lin, ln, _ in layers:
        params += [lin.weight, lin.bias, ln.weight, ln.bias]
    tables = [t for t, _ in segments]
    indices = tuple(i for _, i in segments)
    extra = [skip] if skip is not None else []
    fn = _FusedMLPTrainBf16 if _is_bf16(segments) else _FusedMLPTrain
    return fn.apply(net, indices, skip is not None, *tables, *extra, *params)


# --------------------------------------------------------------------------- bf16 training variant
_train_bf16_enabled = True
_wgrad_hip = True          # A/B: hand-written split-K bf16-MFMA weight gradient vs the library's TN GEMM


def set_train_bf16(flag: bool, wgrad_hip: bool = True) -> None:
    global _train_bf16_enabled, _wgrad_hip
    _train_bf16_enabled, _wgrad_hip = bool(flag), bool(wgrad_hip)


def _wgrad(dz: torch.Tensor, rows: torch.Tensor, colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 dz^T rows of bf16 operands (+ optionally dz's column sums = the bias gradient, from the same pass)"""
    if _opt("wgrad_hip"):
        from .ops import wgrad_bf16
        return wgrad_bf16(dz, rows, colsum=colsum)
    if colsum is not None:
        colsum.copy_(dz.float().sum(dim=0))
    return (dz.t() @ rows).float()


_bwd_fused = True          # A/B: hand-written fused data gradient (hgnn_mlp_backward_layer_bf16) vs library GEMM + row passes


def set_bwd_fused(flag: bool) -> None:
    global _bwd_fused
    _bwd_fused = bool(flag)


def _bwd_layer_supported(K: int, N: int) -> bool:
    return bool(_lib.load().hgnn_mlp_backward_layer_supported_bf16(int(K), int(N)))


def _bwd_layer(dz, W, z_prev, gamma, beta, act, eps, skip=None, want_a=False):
    """``hgnn_mlp_backward_layer_bf16``.  W: the Linear's fp32 master weight [K, N] (or a column slice of it).
    LayerNorm form (z_prev given): returns (dz_prev, a_prev or None, dgamma, dbeta); input form: (dx,)."""
    M, K = int(dz.shape[0]), int(dz.shape[1])
    N = int(W.shape[1])
    dev = dz.device
    wt = _fragment_order(W.detach().t().to(torch.bfloat16).contiguous())      # W^T [N, K] in A-fragment order
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ln = z_prev is not None
    a_prev = torch.empty((M, N), dtype=torch.bfloat16, device=dev) if (ln and want_a) else None
    partials = torch.empty((_lib.MLP_BWD_BLOCKS, 2, N), dtype=torch.float32, device=dev) if ln else None
    gm = gamma.detach().float().contiguous() if ln else None
    bt = beta.detach().float().contiguous() if ln else None
    sk = skip.contiguous() if skip is not None else None
    with torch.cuda.device(dev):
        _lib.check(_lib.load().hgnn_mlp_backward_layer_bf16(
            _lib.ptr(dz.contiguous()), M, K, N, _lib.ptr(wt), _lib.ptr(z_prev), _lib.ptr(gm), _lib.ptr(bt), int(act),
            float(eps), _lib.ptr(sk), _lib.ptr(out), _lib.ptr(a_prev), _lib.ptr(partials),
            _lib.current_stream(dev)), "hgnn_mlp_backward_layer_bf16")
    if not ln:
        return (out,)
    sums = partials.sum(dim=0)
    return out, a_prev, sums[0], sums[1]


def _seg_reduce_wide(plan, dz):
    """segmented reduce of bf16 rows wider than the bf16 row kernel's 512 columns: column blocks"""
    from .ops import _seg_reduce
    F = int(dz.shape[1])
    if F <= 512:
        return _seg_reduce(plan, dz, None, None)
    return torch.cat([_seg_reduce(plan, dz[:, c:c + 512].contiguous(), None, None) for c in range(0, F, 512)], dim=1)


class _FusedMLPTrainBf16(torch.autograd.Function):
    """Differentiable fused MLP on bf16 rows (BASELINE config 4 dtype; fp32 master weights).

    forward : the feature-split bf16-MFMA kernel, additionally dumping every layer's pre-LayerNorm rows
              z_l in bf16 (``hgnn_mlp_desc.save_pre``);
    backward: per layer ONE HIP row pass for activation' -> LayerNorm backward -> dgamma / dbeta / dbias
              (``hgnn_ln_act_backward_bf16``), the hidden activations recomputed from z_l in one more
              (``hgnn_ln_act_forward_bf16``), the WEIGHT gradients by the hand-written split-K bf16-MFMA
              kernel ``hgnn_wgrad_bf16`` (fp32 accumulation, deterministic: the library's TN GEMM reaches
              140-370 TFLOP/s on this tall reduction, 3.3-4.3x slower), the data gradients by bf16
              library GEMMs; gathered segments factor through the atomics-free segmented reduce exactly
              as in the fp32 variant (N-row products instead of M-row ones)."""

    @staticmethod
    def forward(ctx, net, indices, has_skip, *tensors):
        n_seg = len(indices)
        tables = list(tensors[:n_seg])
        skip = tensors[n_seg] if has_skip else None
        params = tensors[n_seg + (1 if has_skip else 0):]
        segments = [(t, i) for t, i in zip(tables, indices)]
        desc = _descriptor_bf16(net, segments, skip, split=True)
        if desc is None:
            raise RuntimeError("fused_concat_mlp_train (bf16): unsupported arguments (call supported_train() first)")
        d, keep, M, n_out = desc
        n = int(d.n_layers)
        dev = tables[0].device
        zs = [torch.empty((M, int(d.width[l + 1])), dtype=torch.bfloat16, device=dev) for l in range(n)]
        for l in range(n):
            d.save_pre[l] = zs[l].data_ptr() if M else None
        out = torch.empty((M, n_out), dtype=torch.bfloat16, device=dev)
        if M:
            with torch.cuda.device(dev):
                _split_forward(d, out, dev)
        del keep
        stats["fused_train_calls"] += 1
        ctx.indices, ctx.has_skip, ctx.n = indices, has_skip, n
        ctx.acts = [int(d.act[l]) for l in range(n)]
        ctx.eps = float(d.ln_eps)
        # skip is the same tensor as a direct (un-gathered) segment: its gradient is folded into that segment's
        ctx.skip_seg = -1
        if has_skip:
            for s_i in range(n_seg):
                t = tables[s_i]
                if indices[s_i] is None and ctx.needs_input_grad[3 + s_i] and ctx.needs_input_grad[3 + n_seg] \
                        and t.data_ptr() == skip.data_ptr() and t.shape == skip.shape and t.stride() == skip.stride():
                    ctx.skip_seg = s_i
        ctx.save_for_backward(*tables, *params, *zs)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        with torch.autocast("cuda", enabled=False):
            return _FusedMLPTrainBf16._backward(ctx, grad_out)

    @staticmethod
    def _backward(ctx, grad_out):
        from .plan import get_plan
        n, indices = ctx.n, ctx.indices
        n_seg = len(indices)
        saved = ctx.saved_tensors
        tables = saved[:n_seg]
        params = saved[n_seg:n_seg + 4 * n]
        zs = saved[n_seg + 4 * n:]
        if int(grad_out.shape[0]) == 0:
            return _zero_grads(ctx, tables, params, grad_out, n_seg)
        W = [params[4 * l] for l in range(n)]
        lnw = [params[4 * l + 2] for l in range(n)]
        lnb = [params[4 * l + 3] for l in range(n)]
        bf = torch.bfloat16
        g = grad_out.contiguous().to(bf)
        grads_params = [None] * (4 * n)
        grads_tables = [None] * n_seg
        pdt = [params[4 * l].dtype for l in range(n)]
        # last layer: its LayerNorm / activation backward has no GEMM in front of it (one HIP row pass)
        dz, dlw, dlb, dbias = _ln_act_backward(zs[n - 1], g, lnw[n - 1], lnb[n - 1], ctx.acts[n - 1], ctx.eps)
        grads_params[4 * (n - 1) + 1] = dbias.to(pdt[n - 1])
        grads_params[4 * (n - 1) + 2] = dlw.to(pdt[n - 1])
        grads_params[4 * (n - 1) + 3] = dlb.to(pdt[n - 1])
        for l in range(n - 1, 0, -1):
            K, N = int(W[l].shape[0]), int(W[l].shape[1])
            need_bias = grads_params[4 * l + 1] is None
            colsum = torch.empty(K, dtype=torch.float32, device=dz.device) if need_bias and _opt("wgrad_hip") else None
            if _opt("bwd_fused") and _bwd_layer_supported(K, N):
                # hand-written data gradient fused with the LayerNorm / activation backward of the layer below
                dz_prev, a_prev, dlw, dlb = _bwd_layer(dz, W[l], zs[l - 1], lnw[l - 1], lnb[l - 1], ctx.acts[l - 1],
                                                       ctx.eps, want_a=True)
                grads_params[4 * l] = _wgrad(dz, a_prev, colsum).to(pdt[l])
                del a_prev
                dbias_prev = None                                         # comes out of layer l-1's weight gradient
            else:
                a_prev = _ln_act_forward(zs[l - 1], lnw[l - 1], lnb[l - 1], ctx.acts[l - 1], ctx.eps)
                grads_params[4 * l] = _wgrad(dz, a_prev, colsum).to(pdt[l])
                del a_prev
                da = dz @ W[l].detach().to(bf)                            # data gradient: bf16 library GEMM
                dz_prev, dlw, dlb, dbias_prev = _ln_act_backward(zs[l - 1], da, lnw[l - 1], lnb[l - 1],
                                                                 ctx.acts[l - 1], ctx.eps)
                del da
            if need_bias:
                grads_params[4 * l + 1] = (colsum if colsum is not None else dz.float().sum(dim=0)).to(pdt[l])
            grads_params[4 * (l - 1) + 2] = dlw.to(pdt[l - 1])
            grads_params[4 * (l - 1) + 3] = dlb.to(pdt[l - 1])
            if dbias_prev is not None:
                grads_params[4 * (l - 1) + 1] = dbias_prev.to(pdt[l - 1])
            dz = dz_prev
        # first layer: gathered segments factor through S = segment_reduce(dz, idx) (see _FusedMLPTrain)
        need_bias = grads_params[1] is None
        W0 = W[0].detach().to(bf)
        dW = torch.empty(tuple(W[0].shape), dtype=torch.float32, device=dz.device)
        H0 = int(W[0].shape[0])
        col = 0
        for s_i in range(n_seg):
            idx = indices[s_i]
            tab = tables[s_i].contiguous()
            w_s = int(tab.shape[1])
            W_s = W0[:, col:col + w_s]
            if idx is not None:
                S = _seg_reduce_wide(get_plan(idx, int(tab.shape[0])), dz)
                dW[:, col:col + w_s] = _wgrad(S, tab)
                if ctx.needs_input_grad[3 + s_i]:
                    grads_tables[s_i] = S @ W_s
                del S
            else:
                colsum = None
                if need_bias and _opt("wgrad_hip"):
                    colsum = torch.empty(H0, dtype=torch.float32, device=dz.device)
                dW[:, col:col + w_s] = _wgrad(dz, tab, colsum)
                if colsum is not None:
                    grads_params[1] = colsum.to(pdt[0])
                    need_bias = False
                if ctx.needs_input_grad[3 + s_i]:
                    folded = ctx.skip_seg == s_i          # edges + MLP(..., edges): both gradients in one epilogue
                    if _opt("bwd_fused") and _bwd_layer_supported(H0, w_s):
                        grads_tables[s_i] = _bwd_layer(dz, W[0][:, col:col + w_s], None, None, None, 0, ctx.eps,
                                                       skip=g if folded else None)[0]
                    elif folded:
                        grads_tables[s_i] = torch.addmm(g, dz, W_s)
                    else:
                        grads_tables[s_i] = dz @ W_s
            col += w_s
        if need_bias:
            grads_params[1] = dz.float().sum(dim=0).to(pdt[0])
        grads_params[0] = dW.to(pdt[0])
        grad_skip = [None if ctx.skip_seg >= 0 else g] if ctx.has_skip else []
        return (None, None, None, *grads_tables, *grad_skip, *grads_params)
