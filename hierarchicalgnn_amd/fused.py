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
    """[(Linear, LayerNorm, act_code)] or None if the Sequential is not Linear->LN->act repeated"""
    mods = list(net)
    if len(mods) % 3 != 0 or not mods:
        return None
    layers = []
    for i in range(0, len(mods), 3):
        lin, ln, act = mods[i], mods[i + 1], mods[i + 2]
        if not isinstance(lin, nn.Linear) or not isinstance(ln, nn.LayerNorm) or type(act) not in _ACT:
            return None
        if isinstance(act, nn.GELU) and getattr(act, "approximate", "none") != "none":
            return None
        if lin.bias is None or not ln.elementwise_affine or ln.bias is None:
            return None
        layers.append((lin, ln, _ACT[type(act)]))
    return layers


def _descriptor(net, segments, skip):
    layers = _parse(net)
    if layers is None or len(layers) not in (2, 3) or not (1 <= len(segments) <= 3):
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
    d.n_layers = len(layers)
    d.width[0] = sum(int(t.shape[1]) for t, _ in segments)
    eps = None
    for l, (lin, ln, act) in enumerate(layers):
        if lin.in_features != d.width[l]:
            return None
        for p in (lin.weight, lin.bias, ln.weight, ln.bias):
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                return None
        d.W[l], d.b[l] = lin.weight.data_ptr(), lin.bias.data_ptr()
        d.ln_w[l], d.ln_b[l] = ln.weight.data_ptr(), ln.bias.data_ptr()
        d.width[l + 1] = lin.out_features
        d.act[l] = act
        if eps is None:
            eps = ln.eps
        elif eps != ln.eps:
            return None
    d.ln_eps = float(eps)
    if skip is not None:
        if tuple(skip.shape) != (M, d.width[len(layers)]) or not skip.is_cuda or skip.dtype != torch.float32:
            return None
        sk = skip if skip.is_contiguous() else skip.contiguous()
        keep.append(sk)
        d.skip = sk.data_ptr()
    else:
        d.skip = None
    d.M = M
    return d, keep, M, int(d.width[len(layers)])


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
