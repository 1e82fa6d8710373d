#!/usr/bin/env python3
"""Edge update (gather -> concat -> MLP -> +skip) at the BASELINE shape in bf16: fused bf16-MFMA kernel
vs the library path (HIP gathers + hipBLASLt bf16 GEMMs + ATen LN/GELU) vs the fp32 fused kernel.
Usage: bench_mlp_bf16.py [L]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierarchicalgnn_amd import fused, make_mlp, mlp, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
x, ei = synth.trackml_event()
graph = synth.directed(ei).cuda()
N, M = 120_000, graph.shape[1]
net = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda()
nodes32 = torch.randn(N, L, device="cuda")
edges32 = torch.randn(M, L, device="cuda")
nodes16, edges16 = nodes32.bfloat16(), edges32.bfloat16()
net16 = make_mlp(3 * L, 2 * L, L, 2, layer_norm=True, output_activation="Tanh", hidden_activation="GELU").cuda().bfloat16()
net16.load_state_dict({k: v.bfloat16() for k, v in net.state_dict().items()})
flop = 2 * (3 * L * 2 * L + 2 * L * L) * M


def timeit(fn, reps=5):
    fn()
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


res = {"L": L, "N": N, "M": M, "flop": flop}
with torch.no_grad():
    seg32 = [(nodes32, graph[0]), (nodes32, graph[1]), (edges32, None)]
    seg16 = [(nodes16, graph[0]), (nodes16, graph[1]), (edges16, None)]
    if L <= 256:
        t = timeit(lambda: mlp.concat_mlp(net, seg32, skip=edges32))
        res["fp32_fused_ms"], res["fp32_fused_tflops"] = t, flop / t / 1e9
    from hierarchicalgnn_amd import _lib
    lib = _lib.load()
    if L <= 256:
        fused.set_bf16_split(False)
        t = timeit(lambda: mlp.concat_mlp(net, seg16, skip=edges16))
        res["bf16_fused_wave_owns_all_features_ms"] = t
        fused.set_bf16_split(True)
    a = mlp.concat_mlp(net, seg16, skip=edges16)
    t = timeit(lambda: mlp.concat_mlp(net, seg16, skip=edges16))
    res["bf16_fused_ms"], res["bf16_fused_tflops"] = t, flop / t / 1e9
    res["bf16_fused_io_GBps"] = (2 * L * M * 5 + 8 * M) / t / 1e6   # 3 gathered rows + skip + out, bf16
    fused.set_enabled(False)
    b = mlp.concat_mlp(net16, seg16, skip=edges16)
    t = timeit(lambda: mlp.concat_mlp(net16, seg16, skip=edges16))
    res["bf16_library_ms"], res["bf16_library_tflops"] = t, flop / t / 1e9
    fused.set_enabled(True)
    ref = mlp.concat_mlp(net, seg32, skip=edges32)
    res["bf16_fused_rel_err_vs_fp32"] = float((a.float() - ref).abs().max() / ref.abs().max())
    res["bf16_library_rel_err_vs_fp32"] = float((b.float() - ref).abs().max() / ref.abs().max())
print(json.dumps(res, indent=1))
