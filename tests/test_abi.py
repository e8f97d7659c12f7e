"""CPU: the C-ABI library builds, loads, and exports every symbol include/sir_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

from sir_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sir_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sir_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_native.LIB_PATH):
        _native.build()
    handle = ctypes.CDLL(_native.LIB_PATH)
    names = _declared()
    assert len(names) >= 8
    for n in names:
        assert hasattr(handle, n), f"{n} declared in sir_hip.h but not exported"
    assert sorted(_native.SIGNATURES) == names, "python binding table out of sync with the header"


def test_abi_version_and_error_string():
    lib = _native.lib()
    assert lib.sir_abi_version() == 1
    assert isinstance(lib.sir_last_error(), bytes)


def test_no_cpu_fallback():
    """Product ops refuse to run without a HIP device instead of falling back."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sir_amd import featurizer
    with pytest.raises(_native.SirError):
        featurizer.get_featurizer()
