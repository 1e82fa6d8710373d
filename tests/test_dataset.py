"""Event container + per-event masking against the reference's own
TrackMLDataset.__getitem__ (fixture produced by tests/golden/make_golden.py).  CPU only."""
import numpy as np
import torch

from conftest import load_golden
from hierarchicalgnn_amd.dataset import TrackMLDataset, load_event, prepare_event, save_event


def _case(z, ci):
    ev = {k[len(f"case{ci}.in."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"case{ci}.in.")}
    hp = {k[len(f"case{ci}.hp."):]: z[k].item() for k in z.files if k.startswith(f"case{ci}.hp.")}
    out = {k[len(f"case{ci}.out."):]: z[k] for k in z.files if k.startswith(f"case{ci}.out.")}
    return ev, hp, out


def test_prepare_event_matches_reference():
    z = load_golden("dataset_masking.npz")
    for ci in range(int(z["n_cases"])):
        ev, hp, ref = _case(z, ci)
        before = {k: v.clone() for k, v in ev.items()}
        got = prepare_event(ev, hp)
        assert all(torch.equal(ev[k], before[k]) for k in ev)            # inputs untouched
        for k, v in ref.items():
            assert k in got, (ci, k)
            assert got[k].shape == v.shape, (ci, k)
            assert np.array_equal(got[k].numpy(), v), (ci, k)


def test_event_container_roundtrip_and_dataset(tmp_path):
    z = load_golden("dataset_masking.npz")
    ev, hp, ref = _case(z, 0)
    path = str(tmp_path / "event0.npz")
    save_event(path, ev)
    back = load_event(path)
    assert set(back) == set(ev) and all(torch.equal(back[k], ev[k]) for k in ev)
    ds = TrackMLDataset([path], hp, stage="train")
    item = ds[0]
    assert len(ds) == 1 and item["dir"] == path
    assert np.array_equal(item["edge_index"].numpy(), ref["edge_index"])
    # masked edge lists only reference surviving hits
    assert int(item["edge_index"].max()) < item["x"].shape[0]
