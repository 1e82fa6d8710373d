"""Fused gather -> concat -> Linear -> LayerNorm -> act ... (+skip) on fp32 MFMA.

Placeholder switchboard until the MFMA kernel lands: ``supported`` returns False,
so ``concat_mlp`` takes the HIP-gather + library-GEMM path.
"""
from __future__ import annotations


def supported(net, segments, skip) -> bool:
    return False


def fused_concat_mlp(net, segments, skip):
    raise RuntimeError("fused MLP kernel is not built")
