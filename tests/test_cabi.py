"""CPU-only checks of the drop-in boundary: libhgnn_hip.so loads, exports every
symbol include/hgnn_hip.h declares, and its host-side entry points behave."""
import ctypes
import os
import re

import pytest

import conftest
from hierarchicalgnn_amd import _lib

HEADER = os.path.join(conftest.ROOT, "include", "hgnn_hip.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hgnn_[a-z0-9_]+)\s*\(", txt)))


def test_library_is_built_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m hierarchicalgnn_amd.build` (or __graft_entry__.build())"
    lib = _lib.load()
    assert lib.hgnn_abi_version() == _lib.ABI_VERSION


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 10
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hgnn_hip.h but not exported"
    assert set(_lib.declared_symbols()) == set(names), "python binding and header disagree"


def test_header_abi_version_matches_binding():
    m = re.search(r"#define\s+HGNN_ABI_VERSION\s+(\d+)", open(HEADER).read())
    assert int(m.group(1)) == _lib.ABI_VERSION


def test_plan_dims_host_logic():
    lib = _lib.load()
    p = _lib.HgnnPlan()
    _lib.check(lib.hgnn_plan_dims(2_000_000, 120_000, 2_000_000, 0, ctypes.byref(p)))
    assert p.chunk == 2_000_000 // 4096 // 4 == 122
    assert p.max_work >= 120_000 + 2_000_000 // p.chunk
    assert p.max_partial >= 2 * (2_000_000 // p.chunk)
    _lib.check(lib.hgnn_plan_dims(100, 10, 100, 0, ctypes.byref(p)))
    assert p.chunk == 32  # clamped
    _lib.check(lib.hgnn_plan_dims(100, 10, 100, 7, ctypes.byref(p)))
    assert p.chunk == 7
    rc = lib.hgnn_plan_dims(-1, 10, 10, 0, ctypes.byref(p))
    assert rc != 0 and b"negative" in lib.hgnn_last_error()
    rc = lib.hgnn_plan_dims(1 << 33, 10, 10, 0, ctypes.byref(p))
    assert rc != 0


def test_struct_mirrors_match_the_compiled_layout():
    lib = _lib.load()
    assert ctypes.sizeof(_lib.HgnnPlan) == lib.hgnn_sizeof_plan() == 24 + 8 + 24 + 88
    assert ctypes.sizeof(_lib.HgnnMlpDesc) == lib.hgnn_sizeof_mlp_desc()


def test_mlp_supported_is_a_host_side_shape_check():
    lib = _lib.load()
    d = _lib.HgnnMlpDesc()
    assert lib.hgnn_mlp_supported(ctypes.byref(d)) == 0
    d.n_seg, d.n_layers = 3, 2
    for i in range(3):
        d.seg_width[i] = 256
    d.width[0], d.width[1], d.width[2] = 768, 512, 256
    for l in range(2):
        d.W[l] = d.b[l] = d.ln_w[l] = d.ln_b[l] = 64  # non-NULL
    assert lib.hgnn_mlp_supported(ctypes.byref(d)) == 1
    d.width[1] = 500                                    # H != 2L
    assert lib.hgnn_mlp_supported(ctypes.byref(d)) == 0
    d.width[1] = 512
    d.seg_width[0] = 250                                # segment not a multiple of 16
    assert lib.hgnn_mlp_supported(ctypes.byref(d)) == 0


def test_unknown_option_is_an_error():
    lib = _lib.load()
    assert lib.hgnn_set_option(b"nt_loads", 1) == 0
    assert lib.hgnn_set_option(b"no_such_option", 1) != 0
    # the experimental two-workgroup tile of the split3 MLP is refused without HGNN_EXPERIMENTAL (include/hgnn_hip.h)
    if "HGNN_EXPERIMENTAL" not in os.environ:
        assert lib.hgnn_set_option(b"mlp_split3_rows128", 2) != 0 and b"experimental" in lib.hgnn_last_error()
    assert lib.hgnn_set_option(b"mlp_split3_rows128", 3) != 0
    assert lib.hgnn_set_option(b"mlp_split3_rows128", 1) == 0
    # round-2 A/B variants that measured slower were removed with their options (include/hgnn_hip.h)
    for gone in (b"seg_grouped", b"mlp_rows128", b"mlp_f32_waves", b"mlp_split_shape", b"seg_xcd"):
        assert lib.hgnn_set_option(gone, 1) != 0


def test_ops_refuse_cpu_tensors_loudly():
    import torch
    import hierarchicalgnn_amd as H
    with pytest.raises(RuntimeError, match="HIP device"):
        H.scatter_add(torch.zeros(4, 8), torch.zeros(4, dtype=torch.long), dim=0, dim_size=2)
    with pytest.raises(RuntimeError, match="HIP device"):
        H.gather_rows(torch.zeros(4, 8), torch.zeros(4, dtype=torch.long))
