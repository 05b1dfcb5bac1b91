# -*- coding: utf-8 -*-
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """`pytest -m "not gpu"` on a host without a GPU, no -n given: the emulator tests are single-threaded and take ~25 minutes
    one after the other, ~7 spread over 8 workers -- so the CPU suite asks pytest-xdist for workers itself (it runs before
    xdist's own hook of the same name, which turns `numprocesses` into workers).  Never for GPU runs: those tests own the device."""
    if hasattr(config, "workerinput") or os.environ.get("WN_TEST_SERIAL"):
        return None
    opt = config.option
    if getattr(opt, "numprocesses", "absent") is not None:   # xdist not installed, or -n given
        return None
    if "not gpu" not in (getattr(opt, "markexpr", "") or "") or getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    try:
        import torch
        if torch.cuda.is_available():
            return None
    except Exception:  # noqa: BLE001
        return None
    opt.numprocesses = max(1, min(8, os.cpu_count() or 1))
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "emu: runs the HIP kernel sources compiled for the host (index-math check)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible and -m gpu was not requested."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
