"""Model-level mirrors for the flat EC-IN model (BASELINE configs 1-2), so that the
``forward(x, edge_index)`` contract can be exercised end to end without the
reference's Lightning / PyG base classes (which are not needed for inference or
for a plain training loop).

    InteractionGNNBlock  <- EdgeClassifier/Models/IN.py:15-95
    EC_InteractionGNN    <- EdgeClassifier/Models/IN.py:97-128

Sub-module names (``ignn_block.node_encoder``, ``ignn_block.edge_encoder``,
``ignn_block.ignn_cells.{i}``, ``edge_classifier``) and therefore the
``state_dict`` keys are the reference's, so its checkpoints load with
``load_state_dict(ckpt["state_dict"])``.  In the reference these classes derive
from a LightningModule; to use the HIP cells inside the reference's own
training script, import-substitute ``gnn_utils`` instead.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .gnn_utils import InteractionGNNCell, _maybe_checkpoint
from .mlp import concat_mlp
from .utils import make_mlp


class InteractionGNNBlock(nn.Module):
    def __init__(self, hparams, iterations, emb=True):
        super().__init__()
        act, ln = hparams["hidden_activation"], hparams["layernorm"]
        self.node_encoder = make_mlp(hparams["spatial_channels"], hparams["hidden"], hparams["latent"],
                                     hparams["nb_node_layer"], output_activation=act, hidden_activation=act,
                                     layer_norm=ln)
        self.edge_encoder = make_mlp(2 * hparams["spatial_channels"], hparams["hidden"], hparams["latent"],
                                     hparams["nb_edge_layer"], layer_norm=ln, output_activation=act,
                                     hidden_activation=act)
        if hparams["share_weight"]:                      # IN.py:51-56: one cell object reused
            cell = InteractionGNNCell(hparams)
            cells = [cell for _ in range(iterations)]
        else:
            cells = [InteractionGNNCell(hparams) for _ in range(iterations)]
        self.ignn_cells = nn.ModuleList(cells)
        if emb:
            self.output_layer = make_mlp(hparams["latent"], hparams["hidden"], hparams["emb_dim"],
                                         hparams["output_layers"], layer_norm=ln, output_activation=None,
                                         hidden_activation=hparams["hidden_output_activation"])
        self.emb = emb
        self.hparams = hparams
        self._ckpt = bool(hparams.get("checkpointing", True))

    def _encode_nodes(self, x):
        return concat_mlp(self.node_encoder, [(x, None)])

    def _encode_edges(self, x, graph):
        return concat_mlp(self.edge_encoder, [(x, graph[0]), (x, graph[1])])

    def forward(self, x, graph):
        if torch.is_grad_enabled() and x.is_leaf and not x.requires_grad:
            x.requires_grad = True                       # IN.py:82 (reentrant checkpoint needs a grad input)
        nodes = _maybe_checkpoint(self._ckpt, self._encode_nodes, x)              # IN.py:84
        edges = _maybe_checkpoint(self._ckpt, self._encode_edges, x, graph)       # IN.py:85
        for cell in self.ignn_cells:                                              # IN.py:87-88
            nodes, edges = cell(nodes, edges, graph)
        if self.emb:
            emb = nn.functional.normalize(self.output_layer(nodes))
            return emb, nodes, edges
        return nodes, edges


class EC_InteractionGNN(nn.Module):
    """flat interaction-network edge classifier: forward(x[N,3], graph[2,E]) -> scores[E]"""

    def __init__(self, hparams):
        super().__init__()
        self.hparams = dict(hparams)
        self.ignn_block = InteractionGNNBlock(hparams, hparams["n_interaction_graph_iters"], emb=False)
        self.edge_classifier = make_mlp(2 * hparams["latent"], hparams["hidden"], 1, hparams["output_layers"],
                                        layer_norm=hparams["layernorm"], output_activation=None,
                                        hidden_activation=hparams["hidden_output_activation"])

    def forward(self, x, graph):
        directed_graph = torch.cat([graph, graph.flip(0)], dim=1)                 # IN.py:122
        nodes, edges = self.ignn_block(x, directed_graph)
        e = graph.shape[1]
        # IN.py:126 -- relies on the ORIGINAL edge order: edges[:E] pairs with edges[E:]
        scores = concat_mlp(self.edge_classifier, [(edges[:e], None), (edges[e:], None)]).squeeze(-1)
        return torch.sigmoid(scores)
