"""On-disk events and the per-event masking of ``TrackMLDataset.__getitem__`` (reference
Modules/utils.py:28-113).  Host-side index bookkeeping, no kernels.

The reference stores each event as a torch-pickled PyG ``Data`` object.  Unpickling that needs
``torch_geometric`` and executes code from the file, so this package reads a plain, pickle-free
container instead: one ``.npz`` per event holding the same named arrays (``x, cell_data, pid, hid,
pt, edge_index, modulewise_true_edges, signal_true_edges, y, y_pid[, primary]``).  ``save_event`` writes it;
converting an existing TrackML directory is a one-off ``save_event(path, data.to_dict())`` in an
environment that has PyG.

``prepare_event`` applies exactly the reference's filtering: noise / hard pT cut / isolated-hit
removal masks, pT of noise hits set to 0, ``nhits`` per particle, ``signal_mask``, optional random
edge dropping, re-indexing of every edge list through the inverse mask, ``inverse_mask`` kept for
un-masking at evaluation time.
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset

EDGE_LISTS = ("modulewise_true_edges", "signal_true_edges", "edge_index")
NODE_FIELDS = ("x", "cell_data", "pid", "hid", "pt", "signal_mask")


def save_event(path: str, event: Dict[str, torch.Tensor]) -> None:
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                  for k, v in event.items() if k != "dir"})


def load_event(path: str) -> Dict[str, torch.Tensor]:
    with np.load(path, allow_pickle=False) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def prepare_event(event: Dict[str, torch.Tensor], hparams, generator: torch.Generator = None) -> Dict[str, torch.Tensor]:
    """utils.py:57-108 on a dict of CPU tensors; returns a new dict (inputs are not modified)"""
    ev = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in event.items()}
    pid, pt = ev["pid"], ev["pt"]
    if hparams["noise"]:
        mask = pid == pid                                  # only NaN particle ids are dropped
    else:
        mask = pid != 0
    if hparams["hard_ptcut"] > 0:
        mask = mask & (pt > hparams["hard_ptcut"])
    if hparams["remove_isolated"]:
        node_mask = torch.zeros(pid.shape, dtype=torch.bool)
        node_mask[ev["edge_index"].unique()] = True
        mask = mask & node_mask
    pt[pid == 0] = 0
    inverse_mask = torch.zeros(len(pid), dtype=torch.long)
    inverse_mask[mask] = torch.arange(int(mask.sum()))
    ev["inverse_mask"] = torch.arange(len(mask))[mask]
    _, inverse, counts = pid.unique(return_inverse=True, return_counts=True)
    ev["nhits"] = counts[inverse]
    if hparams["primary"]:
        ev["signal_mask"] = (ev["nhits"] >= hparams["n_hits"]) & (ev["primary"] == 1)
    else:
        ev["signal_mask"] = ev["nhits"] >= hparams["n_hits"]
    if hparams.get("edge_dropping_ratio", 0) != 0:
        keep = torch.rand(ev["edge_index"].shape[1], generator=generator) >= hparams["edge_dropping_ratio"]
        ev["edge_index"] = ev["edge_index"][:, keep]
        ev["y"], ev["y_pid"] = ev["y"][keep], ev["y_pid"][keep]
    graph_mask = mask[ev["edge_index"]].all(0)
    for k in ("y", "y_pid"):
        ev[k] = ev[k][graph_mask]
    for k in EDGE_LISTS:
        e = ev[k]
        ev[k] = inverse_mask[e[:, mask[e].all(0)]]
    for k in NODE_FIELDS:
        ev[k] = ev[k][mask]
    if hparams["primary"]:
        ev["primary"] = ev["primary"][mask]
    return ev


class TrackMLDataset(Dataset):
    """same constructor and indexing contract as the reference class, over ``.npz`` events"""

    def __init__(self, dirs: Sequence[str], hparams, stage: str = "train", device: str = "cpu"):
        super().__init__()
        self.dirs, self.num = list(dirs), len(dirs)
        self.device, self.stage, self.hparams = device, stage, hparams

    def __getitem__(self, key):
        ev = prepare_event(load_event(self.dirs[key]), self.hparams)
        ev["dir"] = self.dirs[key]
        return ev

    def __len__(self):
        return self.num
