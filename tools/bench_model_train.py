#!/usr/bin/env python3
"""One training step (forward + BCE loss + backward) of the flat EC-IN model (BASELINE config 2: latent 128,
14 cells) on the synthetic TrackML-shaped event, the way the reference trains it (reentrant checkpointing),
HIP path vs library path.  Usage: bench_model_train.py [L] [nockpt] [bf16]
(bf16: latent rows in bf16 -- BASELINE config 4's dtype; fp32 master weights; the differentiable bf16 fused MLP with
the hand-written bf16-MFMA weight gradient; "library" then means bf16 autocast GEMMs)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import fused, synth
from hierarchicalgnn_amd.models import EC_InteractionGNN

L = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ckpt = "nockpt" not in sys.argv[2:]
bf16 = "bf16" in sys.argv[2:]
torch.manual_seed(1236)
hp = dict(spatial_channels=3, latent=L, hidden=2 * L, n_interaction_graph_iters=14, nb_node_layer=3,
          nb_edge_layer=2, output_layers=3, hidden_output_activation="GELU", hidden_activation="GELU",
          layernorm=True, share_weight=False, checkpointing=ckpt)
if bf16:
    hp["feature_dtype"] = "bf16"
model = EC_InteractionGNN(hp).cuda().train()
x, ei = synth.trackml_event()
x, ei = x.cuda(), ei.cuda()
target = (torch.rand(ei.shape[1], device="cuda") < 0.3).float()
res = {"model": "EC-IN", "latent": L, "feature_dtype": "bf16" if bf16 else "fp32", "cells": 14, "checkpointing": ckpt, "N": x.shape[0], "E": ei.shape[1]}


def step():
    scores = model(x, ei)
    loss = torch.nn.functional.binary_cross_entropy(scores, target)
    loss.backward()
    model.zero_grad(set_to_none=True)
    return float(loss.detach())


for name, on in (("hip", True), ("library", False))[:1 if os.environ.get("TRAIN_HIP_ONLY") else 2]:
    fused.set_enabled(on)
    step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        loss = step()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    res[f"train_step_{name}_ms"] = ts[1]
    res[f"loss_{name}"] = loss
    res[f"peak_mem_GB_{name}"] = torch.cuda.max_memory_allocated() / 2**30
fused.set_enabled(True)
if bf16 and os.environ.get("TRAIN_WGRAD_AB"):
    fused.set_train_bf16(True, wgrad_hip=False)          # A/B: the library's TN GEMM for the weight gradients
    step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        step()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    res["train_step_hip_with_library_wgrad_ms"] = ts[1]
    fused.set_train_bf16(True, wgrad_hip=True)
res["events_per_s_hip"] = 1e3 / res["train_step_hip_ms"]
print(json.dumps(res, indent=1))
