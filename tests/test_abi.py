"""CPU: the C-ABI library loads and exports exactly what include/hvla.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "hvla.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hvla_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from hypervla import _native
    assert _declared() == sorted(_native.EXPORTS)


def test_library_exports_every_declared_symbol():
    from hypervla import _native
    if not os.path.exists(_native.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_native.lib_path())
    for name in _declared():
        assert hasattr(lib, name), name
    _native.load_library()


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hypervla import _native
    from hypervla.config import FULL
    from hypervla.model import HyperVLA
    with pytest.raises(_native.NativeError):
        _native.Context(FULL, 0, 4)                      # hvla_create -> HVLA_E_DEVICE
    with pytest.raises(RuntimeError):
        HyperVLA.from_synthetic(FULL)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hyper-vla_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "hvla_ref_" not in txt, f
