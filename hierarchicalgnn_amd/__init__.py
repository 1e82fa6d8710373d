"""hierarchicalgnn_amd -- MI355X-native message-passing engine for the Exa.TrkX
hierarchical GNN (reference: clairesonglee/HierarchicalGNN).

Drop-in surface (same names / signatures as the reference):
    scatter_add(src, index, dim=0, dim_size=N)          <- torch_scatter.scatter_add
    InteractionGNNCell(hparams), HierarchicalGNNCell(hparams)   <- Modules/gnn_utils.py
    make_mlp(...)                                        <- Modules/utils.py

Everything on the hot path runs in hand-written HIP kernels loaded from
libhgnn_hip.so through the C ABI of include/hgnn_hip.h; there is no CPU or
eager fallback.
"""
from .ops import scatter_add, gather_scale_scatter, gather_rows, l1_row_scale  # noqa: F401
from .plan import GraphPlan, get_plan, clear_plan_cache, plan_cache_stats  # noqa: F401
from .utils import make_mlp  # noqa: F401
from .gnn_utils import InteractionGNNCell, HierarchicalGNNCell  # noqa: F401

__version__ = "0.1.0"
