#!/usr/bin/env python3
"""Library (hipBLASLt / rocBLAS through torch.matmul) bf16 GEMM rates at the shapes of the MLP backward:
dgrad  da = dz[M,Ho] @ W[Ho,Hi]          (NN, M = 2M rows)
wgrad  dW = dz[M,Ho]^T @ a[M,Hi]         (TN, reduction over the 2M rows)
the bar a hand-written bf16-MFMA backward has to clear."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


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
for Ho, Hi in ((256, 512), (512, 768), (512, 256), (512, 1024), (1024, 1536), (1024, 512)):
    dz = torch.randn(M, Ho, device="cuda", dtype=torch.bfloat16)
    W = torch.randn(Ho, Hi, device="cuda", dtype=torch.bfloat16)
    a = torch.randn(M, Hi, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * Ho * Hi
    ms = t(lambda: dz @ W)
    out[f"dgrad_{Ho}x{Hi}"] = {"ms": ms, "TFLOPs": fl / ms / 1e9}
    ms = t(lambda: dz.t() @ a)
    out[f"wgrad_{Ho}x{Hi}"] = {"ms": ms, "TFLOPs": fl / ms / 1e9}
    del dz, W, a
print(json.dumps(out, indent=1))
