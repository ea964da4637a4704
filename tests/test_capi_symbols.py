"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly the
symbols include/arrow_amd.h declares (no compute calls are made here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "arrow_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(arx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from arrow_amd import _lib

    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from arrow_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/arrow_amd.h but not exported"
    loaded = _lib.load()
    assert loaded.arx_abi_version() == 3


def test_missing_library_fails_loudly(tmp_path):
    from arrow_amd import _lib

    with pytest.raises(_lib.ArrowDeviceError, match="no CPU fallback"):
        _lib.load(str(tmp_path / "libarrow_amd.so"))


def test_no_gpu_fails_loudly_instead_of_falling_back():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import numpy as np

    import arrow_amd as amd

    amd.array.set_default_device(None)
    with pytest.raises(amd.ArrowDeviceError, match="no CPU fallback"):
        amd.Array.from_numpy(np.arange(4))


def test_product_does_not_import_the_oracle_or_emulator():
    """The oracle and the emulator are test infrastructure: nothing under arrow_amd/ may use them."""
    pkg = os.path.join(ROOT, "arrow_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", src, flags=re.M), f
                assert "hip_emu" not in src and "arx_oracle" not in src, f
