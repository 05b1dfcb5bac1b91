# -*- coding: utf-8 -*-
"""Load the host-compiled kernel emulator (TEST INFRASTRUCTURE, see tests/emu/hip_emu.h)."""
import functools

from pytorchwavenetvocoder_amd import _lib


@functools.lru_cache(maxsize=1)
def emu_library():
    from tests.emu import build_emu
    return _lib.load_library(build_emu.build(), _test_emulator=True)
