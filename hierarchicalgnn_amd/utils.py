"""Host-side mirror of the one helper of the reference's Modules/utils.py that is
on the hot path: ``make_mlp`` (utils.py:169-196).

It builds the same ``nn.Sequential`` (same sub-module indices, hence the same
``state_dict`` keys: ``{0,3,6}.weight`` Linear, ``{1,4,7}.weight`` LayerNorm
when ``layer_norm=True``), so reference checkpoints load unchanged.  The
modules only HOLD the parameters; on an MI355X the cells evaluate them with the
fused HIP/MFMA kernel (``fused_mlp``), not with these layers' own forward.
"""
from __future__ import annotations

import torch.nn as nn


class FusedMLPSequential(nn.Sequential):
    """``nn.Sequential`` (same sub-module indices, same ``state_dict`` keys) that forgets the fused kernels' prepared
    copies of its weights whenever the module is switched / moved / reloaded: ``train()``, ``eval()``, ``to()`` /
    ``cuda()`` / ``float()`` (``_apply``) and ``load_state_dict``.  In-place ops on the parameters themselves are
    caught by their version counters; these hooks cover edits through ``param.data`` that are followed by one of
    the calls above (``fused._WeightCache``)."""

    def _drop_prepared(self):
        from . import fused
        fused.clear_weight_cache(self)

    def train(self, mode: bool = True):
        self._drop_prepared()
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._drop_prepared()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._drop_prepared()
        return super().load_state_dict(*a, **k)

    def _load_from_state_dict(self, *a, **k):       # reached when a PARENT module's load_state_dict recurses
        self._drop_prepared()
        return super()._load_from_state_dict(*a, **k)


def make_mlp(input_size, hidden_size, output_size, hidden_layers, hidden_activation="GELU",
             output_activation="GELU", layer_norm=False):
    """[Linear -> (LayerNorm) -> act] x (hidden_layers-1) -> Linear -> (LayerNorm -> act)"""
    hidden_act = getattr(nn, hidden_activation)
    out_act = getattr(nn, output_activation) if output_activation is not None else None
    sizes = [input_size] + [hidden_size] * (hidden_layers - 1) + [output_size]
    layers = []
    for i in range(hidden_layers - 1):
        layers.append(nn.Linear(sizes[i], sizes[i + 1]))
        if layer_norm:
            layers.append(nn.LayerNorm(sizes[i + 1]))
        layers.append(hidden_act())
    layers.append(nn.Linear(sizes[-2], sizes[-1]))
    if out_act is not None:
        if layer_norm:
            layers.append(nn.LayerNorm(sizes[-1]))
        layers.append(out_act())
    return FusedMLPSequential(*layers)


def process_hparams(hparams):
    """The reference's config contract (Modules/training_utils.py:13-20): its shipped YAMLs carry
    ``hidden: ratio`` + ``hidden_ratio`` (EdgeClassifier/Configs/IN.yaml:35-36,
    BipartiteClassification/Configs/HGNN_GMM.yaml), resolved to ``hidden_ratio * latent`` before the
    model is built, and ``cluster_granularity`` defaults to 0.  Returns a resolved COPY; the model
    mirrors call this in their constructors, so a raw YAML dict drops in."""
    hp = dict(hparams)
    if hp.get("hidden") == "ratio":
        if "hidden_ratio" not in hp:
            raise KeyError("hparams['hidden'] == 'ratio' needs hparams['hidden_ratio'] (training_utils.py:14-15)")
        hp["hidden"] = int(hp["hidden_ratio"] * hp["latent"])
    elif isinstance(hp.get("hidden"), str):
        raise ValueError(f"hparams['hidden'] must be an int or 'ratio', got {hp['hidden']!r}")
    hp.setdefault("cluster_granularity", 0)
    return hp
