"""``concat -> make_mlp Sequential -> (+skip)`` evaluation for the cells.

Every MLP on the hot path has the shape  out = MLP(cat[seg_0, seg_1, seg_2]) + skip
where each segment is either a tensor or a row gather ``table[index]``
(Modules/gnn_utils.py:52-53, :61-62, :124-126, :134, :142-144, :152).

``concat_mlp`` is the single entry point the cells use.  Gathers run in the HIP
row-gather kernel (backward: atomics-free segmented reduce).  The dense layers
are evaluated by the fused fp32-MFMA kernel when the shape is supported
(``fused.py``: forward-only when autograd is off, a differentiable variant with a
hand-written backward when it records), otherwise by the library GEMM path (rocBLAS
through ATen) -- all on the GPU; none is a CPU fallback.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .ops import gather_rows

Segment = Tuple[torch.Tensor, Optional[torch.Tensor]]  # (table, index or None)


def concat_mlp(net: nn.Sequential, segments: Sequence[Segment], skip: Optional[torch.Tensor] = None,
               bf16_tail: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``bf16_tail`` (bf16 latent mode, encoders): when an fp32-input MLP has no fused kernel, keep its
    first Linear in fp32 (hit coordinates must not be rounded to 8 bits) and run the rest -- the wide
    GEMMs -- in bf16; the result is bf16."""
    from . import fused
    if out is not None:
        # caller-supplied output rows (no-grad only): straight into the fused kernel when it takes the shape,
        # otherwise evaluated as usual and copied
        if torch.is_grad_enabled() and any(t.requires_grad for t, _ in segments):
            raise RuntimeError("concat_mlp(out=...) is a no-grad path")
        if not bf16_tail and fused.supported(net, segments, skip):
            return fused.fused_concat_mlp(net, segments, skip, out=out)
        return out.copy_(concat_mlp(net, segments, skip, bf16_tail))
    if bf16_tail and skip is None and len(net) > 3 and not torch.is_grad_enabled() and fused._opt("enabled"):
        # bf16 latent mode, encoders: hybrid chain -- the first Linear (hit coordinates: must not be rounded to 8
        # bits) as ONE fp32 fused layer, the wide tail on the bf16 feature-split kernel (edge encoder at latent 256,
        # 2M rows: 3.0 instead of 5.6 ms for the all-fp32 fused kernel)
        first, rest = nn.Sequential(*list(net)[:3]), nn.Sequential(*list(net)[3:])
        if all(t.dtype == torch.float32 for t, _ in segments) and fused.supported(first, segments, None):
            probe = [(torch.empty((0, net[0].out_features), dtype=torch.bfloat16, device=segments[0][0].device), None)]
            if fused.supported(rest, probe, None):
                y = fused.fused_concat_mlp(first, segments, None)
                return fused.fused_concat_mlp(rest, [(y.to(torch.bfloat16), None)], None)
    if fused.supported(net, segments, skip, allow_chain=not bf16_tail):
        return fused.fused_concat_mlp(net, segments, skip)
    if fused.supported_train(net, segments, skip):
        return fused.fused_concat_mlp_train(net, segments, skip)
    parts: List[torch.Tensor] = []
    for table, index in segments:
        parts.append(table if index is None else gather_rows(table, index))
    x = parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)
    if bf16_tail and x.dtype == torch.float32 and x.is_cuda and isinstance(net[0], nn.Linear) and len(net) > 1:
        y = net[0](x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net[1:](y)
        y = y.to(torch.bfloat16)
    elif x.dtype == torch.bfloat16 and next(net.parameters()).dtype == torch.float32:
        # bf16 feature rows with fp32 master weights (hparams["feature_dtype"] = "bf16"): library
        # GEMMs in bf16, LayerNorm statistics in fp32 -- what autocast does
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
        y = y.to(torch.bfloat16)
    else:
        y = net(x)
    return y if skip is None else y + skip
