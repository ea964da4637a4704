"""Root conftest: the CPU tier (`python -m pytest tests/ -m "not gpu"`) runs on four worker processes.

The tier is ~1150 tests of which the emulated ones (kernel sources under the SIMT emulator, the plugin scripts in
subprocesses) take 16 - 20 minutes on one core; they are independent of each other (the emulator libraries are built once
under a file lock, tests/emu/build_emu.py; rendezvous ports are picked per test), so pytest-xdist cuts the tier to ~6
minutes.  Only that one invocation is touched: a `-m gpu` run (one GPU: serial), an explicit -n, a -k selection,
--collect-only or ARROW_AMD_TEST_SERIAL=1 keep pytest's single process.  The xdist workers run this hook too: they are
recognised three ways (config.workerinput, PYTEST_XDIST_WORKER, the marker this hook leaves in the environment) and never
ask for workers of their own."""
import os

import pytest

_MARK = "ARROW_AMD_CPU_TIER_WORKERS"


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER") or os.environ.get(_MARK):
        return None
    opt = config.option
    if os.environ.get("ARROW_AMD_TEST_SERIAL") == "1" or not config.pluginmanager.hasplugin("xdist"):
        return None
    if getattr(opt, "markexpr", "") != "not gpu" or getattr(opt, "keyword", "") or getattr(opt, "numprocesses", None):
        return None
    if getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    os.environ[_MARK] = "4"          # inherited by the workers (and by the tests' own subprocesses): no second level
    opt.numprocesses = 4
    if getattr(opt, "dist", "no") == "no":
        opt.dist = "load"
    return None
