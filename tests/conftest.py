import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "emu: runs the kernel sources under the CPU SIMT emulator")


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
