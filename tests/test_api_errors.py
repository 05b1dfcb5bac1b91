# -*- coding: utf-8 -*-
"""Error behaviour of the host mirror and the C ABI: bad arguments must fail loudly (ValueError from
the Python mirror, WnError carrying wn_last_error() from the library), never compute on garbage and
never fall back to a CPU path.  Runs on the kernel emulator (test infrastructure), no GPU needed."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

from pytorchwavenetvocoder_amd import _lib
from pytorchwavenetvocoder_amd.nets import WaveNet
from tests.emu_util import emu_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(**kw):
    cfg = dict(n_quantize=16, n_aux=4, n_resch=8, n_skipch=8, dilation_depth=2, dilation_repeat=1, kernel_size=2,
               upsampling_factor=4)
    cfg.update(kw)
    return WaveNet(_library=emu_library(), **cfg)


def test_forward_argument_checks():
    m = _model()
    x = torch.zeros(1, 16, dtype=torch.long)
    h = torch.zeros(1, 4, 4)
    m(x, h)  # sanity: the valid call works
    with pytest.raises(ValueError):
        m.engine.forward(x.float(), h)                      # tokens must be int64
    with pytest.raises(ValueError):
        m.engine.forward(torch.zeros(1, 18, dtype=torch.long), torch.zeros(1, 4, 4))   # T % U != 0
    with pytest.raises(ValueError):
        m.engine.forward(x, torch.zeros(1, 4, 5))           # aux frames != T / U
    with pytest.raises(ValueError):
        m.engine.forward(x, torch.zeros(1, 3, 4))           # aux channels != n_aux
    with pytest.raises(ValueError):
        m.engine.forward(x[0], h)                           # (T,) instead of (B, T)


def test_backward_needs_forward_and_matching_shape():
    m = _model()
    with pytest.raises(_lib.WnError):
        m.engine.backward(torch.zeros(1, 16, 16))
    x = torch.zeros(1, 16, dtype=torch.long)
    h = torch.zeros(1, 4, 4)
    m.engine.forward(x, h)
    with pytest.raises((ValueError, _lib.WnError)):
        m.engine.backward(torch.zeros(1, 16, 12))           # dlogits of another T


def test_loss_window_checked_by_the_library():
    m = _model()
    x = torch.zeros(1, 16, dtype=torch.long)
    h = torch.zeros(1, 4, 4)
    logits = m.engine.forward(x, h)
    t = torch.zeros(1, 16, dtype=torch.long)
    for bad in (-1, 16, 100):
        with pytest.raises(_lib.WnError) as e:
            m.engine.loss(logits, t, t_start=bad)
        assert "t_start" in str(e.value)                    # the library's message travels with the exception
    with pytest.raises(ValueError):
        m.engine.mol_loss(logits, torch.zeros(1, 16))       # softmax model: out_channels is not 3 * n_mix


def test_c_abi_rejects_bad_configs_and_null_pointers():
    lib = emu_library()
    cfg = _lib.WnConfig(16, 4, 8, 8, 2, 1, 2, 4, 16)
    assert lib.wn_param_count(ctypes.byref(cfg)) > 0
    for field, val in (("n_quantize", 0), ("n_resch", 0), ("kernel_size", 0), ("dilation_depth", 0), ("n_skipch", -3)):
        bad = _lib.WnConfig(16, 4, 8, 8, 2, 1, 2, 4, 16)
        setattr(bad, field, val)
        assert lib.wn_param_count(ctypes.byref(bad)) == -1, field
        assert lib.wn_workspace_bytes(ctypes.byref(bad), 1, 16) == 0, field
        assert lib.wn_last_error()                          # a message is left for the caller
    ws = torch.zeros(lib.wn_workspace_bytes(ctypes.byref(cfg), 1, 16) // 4 + 1)
    rc = lib.wn_forward(ctypes.byref(cfg), 1, 16, None, None, None, None, ws.data_ptr(), ws.numel() * 4, 0, None)
    assert rc != 0 and b"NULL" in lib.wn_last_error()
    p = torch.zeros(lib.wn_param_count(ctypes.byref(cfg)))
    x = torch.zeros(1, 16, dtype=torch.long)
    h = torch.zeros(1, 4, 4)
    out = torch.zeros(1, 16, 16)
    rc = lib.wn_forward(ctypes.byref(cfg), 1, 16, p.data_ptr(), x.data_ptr(), h.data_ptr(), out.data_ptr(), ws.data_ptr(),
                        16, 0, None)                        # workspace far too small
    assert rc != 0 and b"workspace" in lib.wn_last_error()


def test_decode_argument_checks():
    m = _model()
    x = torch.zeros(2, 3, dtype=torch.long)
    h = torch.zeros(2, 4, 4)
    with pytest.raises(ValueError):
        m.engine.decode(x, h, [4])                          # one length per utterance
    with pytest.raises(ValueError):
        m.engine.decode(x, h, [4, 4], mode="nucleus")       # unknown token choice
    with pytest.raises(ValueError):
        m.engine.decode(x, h, [4, 400])                     # aux features do not cover the request
    with pytest.raises(ValueError):
        m.engine.decode(x, h, [4, 4], mode="mol")           # softmax model


def test_decode_residency_query_and_mode_bits_through_the_c_abi(monkeypatch):
    """ABI v8: wn_decode_layered_residency reports the grid of the persistent launch and what the device keeps resident, bad
    arguments return < 0, and the layout of the decode state depends on the per-call mode bits (not on process-wide state)."""
    lib = emu_library()
    m = WaveNet(_library=lib, n_quantize=256, n_aux=80, n_resch=512, n_skipch=256, dilation_depth=10, dilation_repeat=3,
                kernel_size=2, upsampling_factor=80)
    cfg = ctypes.byref(m.engine.cfg)
    wg, cap = ctypes.c_int(-1), ctypes.c_int(-1)
    assert lib.wn_decode_layered_residency(None, 2, 0, ctypes.byref(wg), ctypes.byref(cap)) < 0
    assert lib.wn_decode_layered_residency(cfg, 0, 0, ctypes.byref(wg), ctypes.byref(cap)) < 0
    assert lib.wn_decode_layered_residency(cfg, 2, 0, None, None) == 1                       # NULL outputs are allowed
    assert lib.wn_decode_layered_residency(cfg, 2, 0, ctypes.byref(wg), ctypes.byref(cap)) == 1 and wg.value == 64
    assert lib.wn_decode_layered_residency(cfg, 2, _lib.DECODE_GRANULES, ctypes.byref(wg), ctypes.byref(cap)) == 1 and wg.value == 128
    # flags / granules / launches: three different state layouts for the same model and batch
    n_flags = lib.wn_decode_layered_state_floats(cfg, 17, 0)
    n_gran = lib.wn_decode_layered_state_floats(cfg, 17, _lib.DECODE_GRANULES)
    assert n_gran > n_flags > 0                                                              # (private queue copies per unit)
    assert lib.wn_decode_layered_error_offset(cfg, 17, 0) != lib.wn_decode_layered_error_offset(cfg, 17, _lib.DECODE_GRANULES)
    monkeypatch.setenv("WN_COOP_CAPACITY", "10")                                             # a device that keeps 10 workgroups
    assert lib.wn_decode_layered_residency(cfg, 2, 0, ctypes.byref(wg), ctypes.byref(cap)) == 0 and (wg.value, cap.value) == (64, 10)
    assert lib.wn_decode_layered_error_offset(cfg, 2, 0) < 0 and 0 < lib.wn_decode_layered_state_floats(cfg, 17, 0) < n_flags


def test_missing_library_fails_loudly():
    """No HIP library -> import of the product path raises; nothing silently takes over."""
    code = ("import os; os.environ['WN_LIB_PATH'] = '/nonexistent/libwavenet_hip.so'\n"
            "from pytorchwavenetvocoder_amd.nets import WaveNet\n"
            "WaveNet(16, 4, 8, 8, 2, 1, 2, 4)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0
    assert "libwavenet_hip" in r.stderr


def test_fp16_pair_mode_takes_the_measured_maximum_only_for_exactly_the_tensor_the_loss_call_returned():
    """engine._dw_mode_flags (host logic of WN_FLAG_DW_F16PAIR, ABI v9): the scale of the fp16 pair split comes from max |dlogits|.
    WN_FLAG_DW_F16_AMAX_WS (the maximum the loss call measured, left in the workspace) is passed only for the tensor OBJECT a loss
    call returned, in the state it was returned in, on the same workspace; any other tensor -- another object at the same address
    (the caching allocator hands addresses out again), a view, a tensor written in place -- keeps the flag ALONE (the library then
    scans the tensor it is given) unless the caller gives its own bound (| EXP_VALID); without the flag nothing is touched."""
    import torch
    from tests.emu_util import emu_library
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine
    eng = WaveNetEngine(32, 4, 16, 16, 2, 1, 2, 4, device="cpu", library=emu_library())
    F = _lib.FLAG_AUX_FUSED | _lib.FLAG_DW_F16PAIR
    WS, VALID = _lib.FLAG_DW_F16_AMAX_WS, _lib.FLAG_DW_F16_EXP_VALID
    dl = torch.zeros(1, 32, 8)
    assert eng._dw_mode_flags(F, dl) == F                                   # nothing noted yet: the library scans
    eng._note_bound(dl, 1.0, 1000)
    assert eng._dw_mode_flags(F, dl) == F | WS
    assert eng._dw_mode_flags(F | (63 << _lib.DW_F16_EXP_SHIFT) | VALID, dl) == F | WS   # stale exponent bits in the flag word are dropped
    assert eng._dw_mode_flags(F, dl.view(1, 32, 8)) == F                    # same memory, another object
    assert eng._dw_mode_flags(F, dl.clone()) == F
    assert eng._dw_mode_flags(F, dl.clone(), dlogits_bound=0.3) == F | (1 << _lib.DW_F16_EXP_SHIFT) | VALID
    assert eng._dw_mode_flags(F, dl, dlogits_bound=1.0) == F | VALID        # e = 0 is a statement now, not a default
    assert eng._dw_mode_flags(_lib.FLAG_AUX_FUSED, dl) == _lib.FLAG_AUX_FUSED
    dl.add_(1.0)                                                            # written in place: the measured maximum is void
    assert eng._dw_mode_flags(F, dl) == F
    eng._note_bound(dl, 0.5, 1000)
    assert eng._dw_mode_flags(F, dl) == F | WS
    eng._ws_key = (1, 64)                                                   # another workspace since the loss call: its word is not this tensor's
    assert eng._dw_mode_flags(F, dl) == F
    eng._ws_key = None
    del dl
    other = torch.zeros(1, 32, 8)                                           # the noted tensor is gone; whatever takes its place does not inherit
    assert eng._dw_mode_flags(F, other) == F
    eng._note_bound(other, 0.0, 1000)                                       # grad_scale 0: nothing measured worth using
    assert eng._dw_mode_flags(F, other) == F
    with pytest.raises(ValueError):
        eng._dw_mode_flags(F, other, dlogits_bound=0.0)
