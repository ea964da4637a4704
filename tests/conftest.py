import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "emu: runs the kernel sources under the CPU SIMT emulator")


# The GPU tier runs the §8(a)-(e) rows first: the full-size BASELINE configs, then the kernel-level parity grids, then the
# rest; the plugin scripts (the widest and youngest file) last.  A late break in a §8(f) script can then not hide the
# rows the bench times (VERDICT r4 "Next round" 2b).  The CPU tier keeps pytest's order.
_GPU_ORDER = ("test_gpu_full_size.py", "test_gpu_parity.py", "test_c_abi_contract.py", "test_gpu_group_keys.py", "test_parquet.py",
              "test_sharded_rccl_plugin.py")


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        if item.get_closest_marker("gpu") is None:
            return len(_GPU_ORDER)
        if name == "test_gpu_arrow_plugin.py":
            return len(_GPU_ORDER) + 1
        return _GPU_ORDER.index(name) if name in _GPU_ORDER else len(_GPU_ORDER)

    items.sort(key=rank)        # (stable: the order inside a file is untouched)


@pytest.fixture
def gpu_ctx():
    """Real device: libarrow_amd.so + cuda:0.  Fails loudly if either is missing."""
    import torch

    import arrow_amd
    from arrow_amd import _lib, array

    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    _lib._lib = None
    lib = _lib.get_lib()
    assert lib.arx_device_count() >= 1
    array.set_default_device(None)
    yield arrow_amd
    torch.cuda.synchronize()
    # the full-size tests leave ~190 GB in torch's caching allocator; nothing outside torch (the plugin's pool, the
    # subprocess-based tests) can reuse those blocks, so every test hands its memory back
    import gc

    gc.collect()
    torch.cuda.empty_cache()


@pytest.fixture
def emu_ctx():
    """The same Python host layer, but the shared library is the kernel sources compiled for the
    host against tests/emu/hip_emu.h and buffers are host tensors.  Test-only plumbing."""
    import arrow_amd
    from arrow_amd import _lib, array

    from tests.emu.build_emu import build

    saved = _lib._lib
    _lib._lib = _lib.load(build())
    array.set_default_device("cpu")
    try:
        yield arrow_amd
    finally:
        _lib._lib = saved
        array.set_default_device(None)
