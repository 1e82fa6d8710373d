"""Drop-in message-passing cells: same class names, constructor (`hparams` dict),
sub-module names (hence ``state_dict`` keys) and ``forward`` signatures as the
reference's Modules/gnn_utils.py, evaluated by the MI355X HIP kernels.

    InteractionGNNCell      <- Modules/gnn_utils.py:17-71
    HierarchicalGNNCell     <- Modules/gnn_utils.py:73-169

Differences that are deliberate (and invisible through the interface):
  * the scatter_add aggregations are an atomics-free destination-sorted
    segmented reduce over a per-event GraphPlan (bitwise reproducible);
  * ``w * X[g]`` products ([B, L], 614 MB at L=256) are never materialised:
    K2/K3 run as one fused gather-scale-reduce kernel;
  * gathers feeding the edge MLP run as whole-row HIP gathers whose backward is
    the same segmented reduce (no atomics in backward either).
Like the reference, each update runs under ``torch.utils.checkpoint`` when
gradients are being recorded (gnn_utils.py:14-15); pass
``hparams["checkpointing"] = False`` to keep activations instead.
"""
from __future__ import annotations

import functools

import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from .mlp import concat_mlp
from .ops import gather_scale_scatter, scatter_add
from .plan import stable_index
from .utils import make_mlp


def _maybe_checkpoint(enabled, fn, *args):
    if enabled and torch.is_grad_enabled() and any(torch.is_tensor(a) and a.requires_grad for a in args):
        # the no-grad first pass of a reentrant checkpoint is PART OF A TRAINING STEP: it must use the arithmetic its
        # backward-time recompute will use (bitwise-deterministic recompute), so the inference-only
        # split-bf16 evaluation of the fp32 MLPs is held off while it runs (fused.training_forward)
        from . import fused
        with fused.training_forward():
            return checkpoint(fn, *args, use_reentrant=True)
    return fn(*args)


class InteractionGNNCell(nn.Module):
    edge_update_takes_out = True     # edge_update(..., out=rows): partition.py's zero-copy interior / boundary split

    def __init__(self, hparams):
        super().__init__()
        L, H = hparams["latent"], hparams["hidden"]
        act = hparams["hidden_activation"]
        # edge network: 3L -> H -> L, Tanh output          (gnn_utils.py:22-30)
        self.edge_network = make_mlp(3 * L, H, L, hparams["nb_edge_layer"], layer_norm=hparams["layernorm"],
                                     output_activation="Tanh", hidden_activation=act)
        # node network: 2L -> H -> H -> L                   (gnn_utils.py:33-41)
        self.node_network = make_mlp(2 * L, H, L, hparams["nb_node_layer"], layer_norm=hparams["layernorm"],
                                     output_activation=act, hidden_activation=act)
        self.hparams = hparams
        self._ckpt = bool(hparams.get("checkpointing", True))

    # gnn_utils.py:46-54
    def _node_update(self, nodes, edges, graph):
        edge_messages = scatter_add(edges, graph[1], dim=0, dim_size=nodes.shape[0])
        return concat_mlp(self.node_network, [(nodes, None), (edge_messages, None)], skip=nodes)

    # gnn_utils.py:57-64
    def _edge_update(self, nodes, edges, graph, out=None):
        return concat_mlp(self.edge_network, [(nodes, graph[0]), (nodes, graph[1]), (edges, None)], skip=edges,
                          out=out)

    def node_update(self, nodes, edges, graph):
        return _maybe_checkpoint(self._ckpt, self._node_update, nodes, edges, graph)

    def edge_update(self, nodes, edges, graph, out=None):
        """``out`` (no-grad): write the updated rows into a caller-supplied row block (partition.py assembles a
        shard's edge table from an interior and a boundary call)"""
        if out is not None:
            return self._edge_update(nodes, edges, graph, out=out)
        return _maybe_checkpoint(self._ckpt, self._edge_update, nodes, edges, graph)

    # gnn_utils.py:66-71 -- the edge update sees the UPDATED nodes
    def forward(self, nodes, edges, graph):
        graph = stable_index(graph)          # inference-tensor graphs: one normal clone per call keys the caches
        nodes = self.node_update(nodes, edges, graph)
        edges = self.edge_update(nodes, edges, graph)
        return nodes, edges


class HierarchicalGNNCell(nn.Module):
    edge_update_takes_out = True

    def __init__(self, hparams):
        super().__init__()
        L, H = hparams["latent"], hparams["hidden"]
        act = hparams["hidden_activation"]
        ln = hparams["layernorm"]
        ne, nn_ = hparams["nb_edge_layer"], hparams["nb_node_layer"]
        # gnn_utils.py:77-115
        self.edge_network = make_mlp(3 * L, H, L, ne, layer_norm=ln, output_activation="Tanh", hidden_activation=act)
        self.node_network = make_mlp(3 * L, H, L, nn_, layer_norm=ln, output_activation=act, hidden_activation=act)
        self.supernode_network = make_mlp(3 * L, H, L, nn_, layer_norm=ln, output_activation=act,
                                          hidden_activation=act)
        self.superedge_network = make_mlp(3 * L, H, L, ne, layer_norm=ln, output_activation="Tanh",
                                          hidden_activation=act)
        self.hparams = hparams
        self._ckpt = bool(hparams.get("checkpointing", True))
        # multi-GPU hook (partition.py): combines the per-shard node->supernode sums (K3) of a
        # node-partitioned event; None on a single GPU
        self.node_message_reduce = None

    # gnn_utils.py:120-127  (K2 + K1)
    def _node_update(self, nodes, edges, supernodes, graph, bipartite_graph, bipartite_edge_weights):
        supernode_messages = gather_scale_scatter(supernodes, bipartite_graph[1], bipartite_graph[0],
                                                  nodes.shape[0], bipartite_edge_weights)
        edge_messages = scatter_add(edges, graph[1], dim=0, dim_size=nodes.shape[0])
        return concat_mlp(self.node_network, [(nodes, None), (edge_messages, None), (supernode_messages, None)],
                          skip=nodes)

    # gnn_utils.py:130-135
    def _edge_update(self, nodes, edges, graph, out=None):
        return concat_mlp(self.edge_network, [(nodes, graph[0]), (nodes, graph[1]), (edges, None)], skip=edges,
                          out=out)

    # gnn_utils.py:138-145  (K3 + K4)
    def _supernode_update(self, nodes, supernodes, superedges, bipartite_graph, bipartite_edge_weights,
                          super_graph, super_edge_weights, node_message_reduce=None):
        node_messages = gather_scale_scatter(nodes, bipartite_graph[0], bipartite_graph[1],
                                             supernodes.shape[0], bipartite_edge_weights)
        reduce = node_message_reduce if node_message_reduce is not None else self.node_message_reduce
        if reduce is not None:
            node_messages = reduce(node_messages)
        attention_messages = scatter_add(superedges, super_graph[1], dim=0, dim_size=supernodes.shape[0],
                                         weight=super_edge_weights)
        return concat_mlp(self.supernode_network,
                          [(supernodes, None), (attention_messages, None), (node_messages, None)], skip=supernodes)

    # gnn_utils.py:148-153
    def _superedge_update(self, supernodes, superedges, super_graph, super_edge_weights):
        return concat_mlp(self.superedge_network,
                          [(supernodes, super_graph[0]), (supernodes, super_graph[1]), (superedges, None)],
                          skip=superedges)

    def node_update(self, *a):
        return _maybe_checkpoint(self._ckpt, self._node_update, *a)

    def edge_update(self, *a, out=None):
        if out is not None:                      # no-grad, caller-supplied output rows (partition.py)
            return self._edge_update(*a, out=out)
        return _maybe_checkpoint(self._ckpt, self._edge_update, *a)

    def supernode_update(self, *a, node_message_reduce=None):
        """``node_message_reduce`` (multi-GPU, partition.py): combines the per-shard node->supernode sums.
        It is bound into the checkpointed function itself, so the backward-time recompute of a reentrant
        checkpoint issues the same collective as the forward did."""
        fn = self._supernode_update
        if node_message_reduce is not None:
            fn = functools.partial(self._supernode_update, node_message_reduce=node_message_reduce)
        return _maybe_checkpoint(self._ckpt, fn, *a)

    def superedge_update(self, *a):
        return _maybe_checkpoint(self._ckpt, self._superedge_update, *a)

    # gnn_utils.py:155-169 : supernode -> node -> superedge -> edge
    def forward(self, nodes, edges, supernodes, superedges, graph, bipartite_graph, bipartite_edge_weights,
                super_graph, super_edge_weights):
        graph, bipartite_graph, super_graph = (stable_index(g) for g in (graph, bipartite_graph, super_graph))
        supernodes = self.supernode_update(nodes, supernodes, superedges, bipartite_graph,
                                           bipartite_edge_weights, super_graph, super_edge_weights)
        nodes = self.node_update(nodes, edges, supernodes, graph, bipartite_graph, bipartite_edge_weights)
        superedges = self.superedge_update(supernodes, superedges, super_graph, super_edge_weights)
        edges = self.edge_update(nodes, edges, graph)
        return nodes, edges, supernodes, superedges
