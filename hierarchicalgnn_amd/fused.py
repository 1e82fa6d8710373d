"""Fused gather -> concat -> [Linear -> LayerNorm -> act] x {2,3} -> (+skip) on fp32 MFMA
(``hgnn_mlp_forward_f32``, csrc/mlp_fused.hip).

``supported`` decides per call; when it says no, ``concat_mlp`` evaluates the same
Sequential with HIP row gathers + library GEMMs (still on the GPU).  The fused kernel is
forward-only, so it is used whenever autograd is not recording: inference, and the first
(no-grad) pass of every reentrant ``torch.utils.checkpoint`` segment -- which is how the
reference runs all of its updates (Modules/gnn_utils.py:14-15).  The recompute pass inside
backward needs saved activations and takes the differentiable path.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import _lib
from .plan import get_index32

_ACT = {nn.GELU: 1, nn.Tanh: 2, nn.ReLU: 3}
_enabled = True
stats = {"fused_calls": 0}


def set_enabled(flag: bool):
    """switch the fused kernel off/on (A/B measurements, debugging)"""
    global _enabled
    _enabled = bool(flag)


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


def _descriptor(net, segments, skip):
    layers = _parse(net)
    if layers is None or len(layers) not in (2, 3) or not (1 <= len(segments) <= 3):
        return None
    if any(ln is None for _, ln, _ in layers[:-1]):
        return None
    d = _lib.HgnnMlpDesc()
    keep = []
    d.n_seg = len(segments)
    M = None
    for i, (table, index) in enumerate(segments):
        if table.dim() != 2 or not table.is_cuda or table.dtype != torch.float32:
            return None
        t = table if table.is_contiguous() else table.contiguous()
        keep.append(t)
        rows = int(index.numel()) if index is not None else int(t.shape[0])
        if M is None:
            M = rows
        elif M != rows:
            return None
        d.seg_table[i] = t.data_ptr()
        d.seg_width[i] = int(t.shape[1])
        if index is not None:
            i32 = get_index32(index, int(t.shape[0]))
            keep.append(i32)
            d.seg_index[i] = i32.data_ptr() if i32.numel() else None
        else:
            d.seg_index[i] = None
    n = len(layers)
    d.n_layers = n
    K = sum(int(t.shape[1]) for t, _ in segments)
    d.width[0] = K
    eps = None
    for l, (lin, ln, act) in enumerate(layers):
        if lin.in_features != d.width[l]:
            return None
        params = [lin.weight, lin.bias] + ([ln.weight, ln.bias] if ln is not None else [])
        for p in params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                return None
        W, b = lin.weight, lin.bias
        if l == 0 and any(int(t.shape[1]) % 16 for t, _ in segments):
            if K > 16:
                return None
            # small-K mode (encoders, K = 3 / 6): one zero-padded 16-column chunk
            W = torch.nn.functional.pad(W.detach(), (0, 16 - K)).contiguous()
            keep.append(W)
            d.w0_cols = 16
        if l == n - 1 and ln is None:
            if lin.out_features != 1:
                return None
            # width-1 head: the plain last layer is stored zero-padded as 32 rows
            Wp = torch.zeros((32, lin.in_features), dtype=torch.float32, device=W.device)
            Wp[0] = W.detach()[0]
            bp = torch.zeros(32, dtype=torch.float32, device=W.device)
            bp[0] = b.detach()[0]
            keep += [Wp, bp]
            W, b = Wp, bp
            d.w_last_rows = 32
        d.W[l], d.b[l] = W.data_ptr(), b.data_ptr()
        if ln is not None:
            d.ln_w[l], d.ln_b[l] = ln.weight.data_ptr(), ln.bias.data_ptr()
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


def supported(net, segments, skip) -> bool:
    if not _enabled:
        return False
    if torch.is_grad_enabled():
        tensors = [t for t, _ in segments] + ([skip] if skip is not None else []) + list(net.parameters())
        if any(t.requires_grad for t in tensors):
            return False
    try:
        desc = _descriptor(net, segments, skip)
    except RuntimeError:
        return False
    if desc is None:
        return False
    return bool(_lib.load().hgnn_mlp_supported(ctypes.byref(desc[0])))


def fused_concat_mlp(net, segments, skip: Optional[torch.Tensor]):
    desc = _descriptor(net, segments, skip)
    if desc is None:
        raise RuntimeError("fused_concat_mlp: unsupported arguments (call supported() first)")
    d, keep, M, n_out = desc
    dev = segments[0][0].device
    out = torch.empty((M, n_out), dtype=torch.float32, device=dev)
    if M == 0:
        return out
    with torch.cuda.device(dev):
        _lib.check(_lib.load().hgnn_mlp_forward_f32(ctypes.byref(d), _lib.ptr(out), _lib.current_stream(dev)),
                   "hgnn_mlp_forward_f32")
    del keep
    stats["fused_calls"] += 1
    return out
