import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no libhgnn_hip.so (build artefacts are git-ignored): build it once, in-tree,
    when hipcc is available (cross-compiles without a GPU).  Tests never fall back to anything else."""
    import shutil
    from hierarchicalgnn_amd import build as b
    if not b.up_to_date() and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        b.build(verbose=False)


@pytest.fixture(autouse=True)
def _exact_fp32_matrix_arithmetic(request):
    """The process default evaluates the fp32 MLPs' GEMMs as split-bf16 products (fused.set_fp32_split3, <= 1.6e-5 vs
    the reference); the GPU parity tests pin the EXACT fp32 kernels unless they are the split-bf16 tests themselves
    (tests/test_gpu_split3.py), so that both kernels stay covered.  The whole suite also passes with the default left
    on (HGNN_KEEP_DEFAULT_FP32_GEMM=1 python -m pytest -m gpu; profiles/r02_gpu_suite_split3_on.txt)."""
    if "gpu" not in request.keywords or request.module.__name__ == "test_gpu_split3" \
            or os.environ.get("HGNN_KEEP_DEFAULT_FP32_GEMM") == "1":
        yield
        return
    from hierarchicalgnn_amd import fused
    old = fused._fp32_split3
    fused.set_fp32_split3(False)
    yield
    fused.set_fp32_split3(old)
    assert not fused._fp32_split3_train or os.environ.get("HGNN_FP32_SPLIT3_TRAIN") == "1"   # opt-in stays opt-in


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max(|b|_inf, tiny): the 'rel' of the 1e-4 fp32 parity bar"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def elem_err(a, b):
    """element-wise counterpart of rel_err: max over elements of |a-b| / (|b| + rms(b)).  rel_err divides
    every difference by the LARGEST reference magnitude (normwise); this one divides by the element's own
    magnitude plus the tensor's typical magnitude, so small elements are held to the bar as well."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    rms = max(float(np.sqrt(np.mean(b * b))), 1e-30)
    return float((np.abs(a - b) / (np.abs(b) + rms)).max())


def assert_parity(a, b, tol=1e-4, what=""):
    """the fp32 parity bar, normwise AND element-wise"""
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    b = b.detach().cpu().numpy() if hasattr(b, "detach") else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    r, e = rel_err(a, b), elem_err(a, b)
    assert r <= tol and e <= tol, f"{what}: normwise {r:.3g}, element-wise {e:.3g} > {tol:g}"
