#!/usr/bin/env python3
"""Hand-written bf16-MFMA weight gradient (hgnn_wgrad_bf16) vs the library's bf16 TN GEMM, M = 2M rows."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd.ops import wgrad_bf16


def t(fn, reps=10):
    for _ in range(3):
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
for Ho, Hi in ((256, 512), (512, 256), (512, 768), (256, 128), (1024, 512), (512, 1024)):
    dz = torch.randn(M, Ho, device="cuda", dtype=torch.bfloat16)
    a = torch.randn(M, Hi, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * Ho * Hi
    by = 2.0 * M * (Ho + Hi)
    ms = t(lambda: wgrad_bf16(dz, a))
    lib = t(lambda: dz.t() @ a)
    err = float(((wgrad_bf16(dz, a) - (dz.t() @ a).float()).abs().max() / (dz.t() @ a).float().abs().max()))
    out[f"{Ho}x{Hi}"] = {"hip_ms": ms, "hip_TFLOPs": fl / ms / 1e9, "hip_row_GBps_read_once": by / ms / 1e6,
                         "library_ms": lib, "library_TFLOPs": fl / lib / 1e9, "speedup": lib / ms,
                         "max_diff_vs_library_bf16_result": err}
    del dz, a
print(json.dumps(out, indent=1))
