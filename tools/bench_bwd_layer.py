#!/usr/bin/env python3
"""Fused backward layer (hgnn_mlp_backward_layer_bf16) vs the unfused pieces it replaces (bf16 library GEMM + HIP
LayerNorm/activation row passes), M = 2M rows.  Usage: bench_bwd_layer.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import _lib, fused


def t(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


M = 2_000_000
out = {}
lib = _lib.load()
for K, N in ((256, 512), (128, 256), (512, 512)):
    dz = torch.randn(M, K, device="cuda").bfloat16()
    W = torch.randn(K, N, device="cuda") / K ** 0.5
    z = torch.randn(M, N, device="cuda").bfloat16()
    gm, bt = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    Wb = W.bfloat16()

    def unfused():
        da = dz @ Wb
        dzp = fused._ln_act_backward(z, da, gm, bt, 1, 1e-5)
        a = fused._ln_act_forward(z, gm, bt, 1, 1e-5)
        return dzp, a

    r = {"unfused_ms": t(unfused)}
    r["fused_ms"] = t(lambda: fused._bwd_layer(dz, W, z, gm, bt, 1, 1e-5, want_a=True))
    out[f"ln_form_K{K}_N{N}"] = r
    del dz, z
for K, N in ((512, 256), (256, 128)):
    dz = torch.randn(M, K, device="cuda").bfloat16()
    W = torch.randn(K, N, device="cuda") / K ** 0.5
    skip = torch.randn(M, N, device="cuda").bfloat16()
    Wb = W.bfloat16()
    out[f"input_form_K{K}_N{N}"] = {"unfused_addmm_ms": t(lambda: torch.addmm(skip, dz, Wb)),
                                    "fused_ms": t(lambda: fused._bwd_layer(dz, W, None, None, None, 0, 1e-5, skip=skip))}
print(json.dumps(out, indent=1))
