"""The Arrow registration shim's DEVICE-RESIDENT paths in the GPU-less CPU tier.

The same C++ sources (arrow_amd/csrc/arrow_plugin.cc + plugin/*.inc) are built against the emulated
kernel library and a host-memory stand-in for the HIP runtime (tests/emu/plugin_hip): Arrow sees kROCM
buffers (non-CPU, `Buffer::data()` is null), the shim reads their addresses, the kernel sources run
under the SIMT emulator.  The scripts are the very ones the GPU tier runs (the rows of tests/plugin_scripts.py::CASES,
the one table both tiers are parametrized over), scaled down: kROCM memory manager + C Device Data round trips, device-aware filter / take / casts /
comparisons / arithmetic / Kleene logic / sorts / utf8, the aggregate_rocm Acero node and whole Acero
plans over device-resident tables.  Test infrastructure only — the product build is untouched."""
import os
import subprocess
import sys

import pytest

from . import plugin_scripts as S

pytestmark = pytest.mark.emu


@pytest.mark.parametrize("script,marker,scale", [pytest.param(c[1], c[2], c[3], id=c[0]) for c in S.CASES])
def test_plugin_emulated(script, marker, scale):
    pytest.importorskip("pyarrow")
    env = dict(os.environ, ARROW_AMD_PLUGIN_EMULATED="1", ARROW_AMD_TEST_SCALE=str(scale), ARROW_AMD_TEST_LIGHT="1")
    r = subprocess.run([sys.executable, "-c", f"ROOT = {S.ROOT!r}\n" + script], capture_output=True, text=True,
                       timeout=1500, cwd=S.ROOT, env=env)
    assert r.returncode == 0 and marker in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
