"""Weights and inputs that are a pure function of (name, shape, seed).

The model-level fixtures at the latents BASELINE.json names (EC-IN latent 128: 4.4 M parameters,
BC-HGNN-GMM latent 256: 25.3 M, latent 512: 101 M) would be 18 / 101 / 404 MB if the reference's
``state_dict`` were stored with them.  Instead ``tests/golden/make_golden.py`` fills the REFERENCE
model with the weights defined here, runs it, and stores only inputs, outputs and one checksum row
per parameter; a test fills the mirror model with the same function (same parameter names = the
weight ABI, same CPU generator stream) and first checks the checksums, so a generator mismatch is
reported as such and never as a parity failure.

The init scheme is the reference's ``kaiming_init`` (Modules/training_utils.py:48-58: first-layer
weights ~ N(0, 1/fan_in), other 2-D weights ~ N(0, 2/fan_in)), with non-trivial biases and
LayerNorm/BatchNorm affines so that parity covers them.
"""
import math
import zlib

import numpy as np
import torch


def _gen(seed: int, name: str) -> torch.Generator:
    return torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))


def fill_parameters(module: torch.nn.Module, seed: int) -> None:
    """in place, on CPU fp32 parameters"""
    with torch.no_grad():
        for name, p in module.named_parameters():
            g = _gen(seed, name)
            if name.endswith(".bias"):
                v = 0.1 * torch.randn(p.shape, generator=g)
            elif p.dim() < 2:
                v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            elif name.endswith("0.weight"):
                v = torch.randn(p.shape, generator=g) / math.sqrt(p.shape[1])
            else:
                v = torch.randn(p.shape, generator=g) * (math.sqrt(2) / math.sqrt(p.shape[1]))
            p.copy_(v)


def checksums(named_tensors) -> np.ndarray:
    """[n, 2] float64: (sum, sum of |.|) per tensor, in name order"""
    rows = []
    for _, t in sorted(named_tensors, key=lambda kv: kv[0]):
        t = t.detach().double().cpu()
        rows.append([float(t.sum()), float(t.abs().sum())])
    return np.asarray(rows, dtype=np.float64).reshape(-1, 2)


def check_parameters(module: torch.nn.Module, expected: np.ndarray) -> None:
    got = checksums(module.named_parameters())
    assert got.shape == expected.shape, f"parameter count differs: {got.shape} vs fixture {expected.shape}"
    if not np.allclose(got, expected, rtol=1e-9, atol=1e-9):
        bad = int(np.argmax(np.abs(got - expected).sum(axis=1)))
        name = sorted(n for n, _ in module.named_parameters())[bad]
        raise AssertionError(f"seeded weights differ from the fixture's at {name}: the CPU generator stream of "
                             f"this torch build is not the one the fixture was made with (not a parity failure)")


def randn(seed: int, name: str, *shape) -> torch.Tensor:
    return torch.randn(*shape, generator=_gen(seed, name))


def grad_sketch(named_grads) -> np.ndarray:
    """[n, 3] float64 per gradient tensor, in name order: (sum, sum of |.|, <grad, probe>) with a fixed
    pseudo-random probe of the gradient's shape -- a compact pin of every weight gradient"""
    rows = []
    for name, gten in sorted(named_grads, key=lambda kv: kv[0]):
        gten = gten.detach().double().cpu()
        probe = torch.randn(gten.shape, generator=_gen(977, name), dtype=torch.float64)
        rows.append([float(gten.sum()), float(gten.abs().sum()), float((gten * probe).sum())])
    return np.asarray(rows, dtype=np.float64).reshape(-1, 3)
