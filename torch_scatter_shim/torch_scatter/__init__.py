"""``torch_scatter`` import shim: put ``<repo>/torch_scatter_shim`` on PYTHONPATH and the
reference's ``from torch_scatter import scatter_add`` (Modules/gnn_utils.py:5,
BipartiteClassification/Models/HGNN_GMM.py:5) resolves to the MI355X HIP kernel.

Only the call shape on the hot path is accelerated (``scatter_add`` with a 1-D index,
``dim=0``).  ``scatter_mean`` (HGNN_GMM.py:251, centroids of 8-wide embeddings) is
expressed through the same kernel.  ``scatter_min/max`` are evaluation-only in the
reference (tracking_utils.py) and are not provided.
"""
import torch

from hierarchicalgnn_amd.ops import scatter_add  # noqa: F401


def scatter_sum(src, index, dim=0, out=None, dim_size=None):
    return scatter_add(src, index, dim=dim, dim_size=dim_size, out=out)


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() else 0
    total = scatter_add(src, index, dim=dim, dim_size=dim_size, out=out)
    ones = torch.ones(src.shape[0], 1, dtype=src.dtype, device=src.device)
    count = scatter_add(ones, index, dim=0, dim_size=dim_size).clamp_(min=1)
    return total / count.view(-1, *([1] * (src.dim() - 1)))


def _unsupported(*a, **k):
    raise NotImplementedError("torch_scatter shim (hierarchicalgnn_amd): only scatter_add/scatter_sum/"
                              "scatter_mean are provided; scatter_min/max are evaluation-side in the reference")


scatter_min = scatter_max = _unsupported
