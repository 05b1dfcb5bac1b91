# -*- coding: utf-8 -*-
"""WN_FLAG_DW_F16PAIR on the GPU (through the C ABI): the weight-gradient contractions with the fp16 pair split (two fp16
pieces per operand, three products on v_mfma_f32_32x32x16_f16; csrc/wn_gemm6.hip k_gemm6_dw<.., F16>) against EVERY gate of the
six-bf16-product mode -- the golden gradients (1e-4 of a tensor's maximum) and the golden weights after the Adam steps
(1e-2 lr: the gate the three-bf16-product mode WN_FLAG_DW_3PRODUCT misses) of reference train.py:527-540 --, the overflow
fall-back (a gradient outside fp16's range makes the conditional six-product launches do the work: the six-product mode's result
bit for bit) and where the scale comes from (ABI v9): max |dlogits| MEASURED by the loss call, or by a scan of any other tensor."""
import pytest
import torch

from tests import parity_common as PC
from tests.golden_util import GoldenCase

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _lib():
    from pytorchwavenetvocoder_amd import _lib as L
    lib = L.load_library()
    assert not lib.is_emulator
    return lib


def _flags():
    """the fp16 pair split of the WEIGHT GRADIENTS alone (every other contraction with six bf16 products): what this file pins"""
    from pytorchwavenetvocoder_amd import _lib as L
    from pytorchwavenetvocoder_amd.engine import SIX_PRODUCT_FLAGS
    return SIX_PRODUCT_FLAGS | L.FLAG_DW_F16PAIR


@pytest.mark.parametrize("name", ["tiny_k2_up", "tiny_k3_noup", "r64_k2_up", "r64_k3_up"])
def test_golden_gradients_and_after_adam_state_with_the_fp16_pair_split(name, monkeypatch):
    monkeypatch.setenv("WN_ENGINE_FLAGS", str(_flags()))
    g = GoldenCase(name)
    eng = PC.check_golden_case(g, _lib(), DEV)
    assert eng.flags == _flags()
    if name != "r64_k3_up":   # (the module-level golden cases of tests/test_gpu_parity.py)
        model, _ = PC.check_module_training(g, _lib(), DEV)
        x, h, t = g.x.to(DEV), g.h.to(DEV), g.t.to(DEV)
        log = PC.launch_log(_lib(), lambda: model.loss_and_backward(x, h, t))
        assert log.get("dw_redo_if_overflow", 0) >= 1, log   # the mode was on: every fp16 launch is followed by its conditional redo


def test_overflow_falls_back_to_the_six_products_bit_for_bit_and_unknown_gradients_are_scanned():
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib as L
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    cfg_t = (256, 80, 64, 256, 10, 1, 2, 80)
    cfg = O.OracleConfig(*cfg_t)
    B, T = 2, cfg.receptive_field + 1200
    T = (T + 79) // 80 * 80
    params = O.random_params(cfg, 31, scale=0.05)
    x, h, t = O.synthetic_batch(cfg, B, T, 32)
    eng = WaveNetEngine(*cfg_t, device=DEV, library=_lib())
    load_state_into_flat(eng, params)
    six = _flags() & ~L.FLAG_DW_F16PAIR
    eng.flags = six
    loss, dl = eng.forward_loss(x.to(DEV), h.to(DEV), t.to(DEV))
    g6 = eng.backward(dl).clone()
    eng.flags = _flags()
    log = PC.launch_log(_lib(), lambda: eng.backward(dl))
    g16 = eng.grads().clone()
    assert log.get("dw_redo_if_overflow", 0) >= 5, log
    scale = float(g6.abs().max())
    err = float((g16 - g6).abs().max()) / scale
    adam = float(((g16 - g6).abs() / (g6.abs() + 1e-8)).max())
    print("fp16 pair vs six bf16 products, 10-layer 64/256 model, B=2, T=%d: max |diff| / max %.3g, Adam-update metric %.3g" % (T, err, adam))
    assert err <= 1e-6 and not torch.equal(g16, g6)
    # promise far too small: the scaled gradient leaves fp16's range, every block raises the word, the redo launches do the work
    gov = eng.backward(dl, dlogits_bound=2.0 ** -40).clone()
    # ... bit for bit -- except the skip_1x1 / res_1x1 gradients where the fused skip + res launch applies (k_dw_skipres: one layer
    # bucket, skip channels a multiple of 256): its redo launches run the six products under the fused launch's split-K plan, the
    # six-product mode under the plans of the two separate launches: the same arithmetic in another summation order
    same = torch.ones_like(g6, dtype=torch.bool)
    for layer in range(eng.n_layers if hasattr(eng, "n_layers") else cfg.dilation_depth * cfg.dilation_repeat):
        for kind in (L.P_SKIP_W, L.P_SKIP_B, L.P_RES_W, L.P_RES_B):
            off, n = eng.param_slice(kind, layer)
            same[off:off + n] = False
    assert torch.equal(gov[same], g6[same])
    assert float((gov - g6).abs().max()) / scale <= 2e-7
    # and the next call with the true bound is the fp16 result again (the word is cleared per call)
    assert torch.equal(eng.backward(dl).clone(), g16)
    # a gradient the engine did not make (autograd's grad_output, a modified tensor): scanned for its maximum -- the same number the
    # loss epilogue measured, so the same bits
    log = PC.launch_log(_lib(), lambda: eng.backward(dl.clone()))
    assert log.get("dw_absmax_scan") == 1, log
    assert torch.equal(eng.grads(), g16)
    dl.mul_(1.0)
    assert torch.equal(eng.backward(dl).clone(), g16)
    # the caller's own word is taken as given (same power of two here)
    assert torch.equal(eng.backward(dl, dlogits_bound=1.0 / (B * (T - cfg.receptive_field))).clone(), g16)
    # underflow: a gradient 2^-30 smaller keeps the mode's accuracy with the measured scale; under a promise that is
    # merely safe (|g| <= 1: what F16PAIR without an exponent meant in ABI v8) every scaled element is below fp16's subnormals
    small = dl * 2.0 ** -30
    g_scan = eng.backward(small).clone() * 2.0 ** 30
    assert float((g_scan - g6).abs().max()) / scale <= 1e-6
    g_loose = eng.backward(small, dlogits_bound=1.0).clone() * 2.0 ** 30
    assert float((g_loose - g6).abs().max()) / scale > 1e-2
    # a C caller that sets WN_FLAG_DW_F16PAIR and nothing else gets the scan, not e = 0
    import ctypes
    from pytorchwavenetvocoder_amd.engine import _ptr, _stream_handle
    ws = eng.workspace(B, T)
    xd, hd = eng._last_inputs
    rc = eng.lib.wn_backward_window(ctypes.byref(eng.cfg), B, T, _ptr(eng.flat_params), _ptr(xd), _ptr(hd), _ptr(small),
                                    eng.receptive_field, _ptr(eng.grads()), _ptr(ws), ws.numel() * 4, None, 0, 0,
                                    L.FLAG_AUX_FUSED | L.FLAG_DW_F16PAIR, _stream_handle(eng.device))
    eng.lib.check(rc, "wn_backward_window")
    assert float((eng.grads() * 2.0 ** 30 - g6).abs().max()) / scale <= 1e-6
    # the engine's DEFAULT adds WN_FLAG_MM_F16PAIR (the k_gemm6 contractions of forward and backward by the same split, each with
    # its conditional redo): against six products everywhere within 5e-6 of the largest gradient; with a promise 2^23 too small
    # every fp16 launch of the backward pass -- data gradients included -- raises the word and is redone: finite, same accuracy
    from pytorchwavenetvocoder_amd.engine import DEFAULT_FLAGS
    assert DEFAULT_FLAGS & L.FLAG_MM_F16PAIR
    eng.flags = DEFAULT_FLAGS
    loss, dl2 = eng.forward_loss(x.to(DEV), h.to(DEV), t.to(DEV))
    log = PC.launch_log(_lib(), lambda: eng.backward(dl2))
    gd = eng.grads().clone()
    assert log.get("mm_redo_if_overflow", 0) >= 3, log
    assert float((gd - g6).abs().max()) / scale <= 5e-6
    gd_ov = eng.backward(dl2, dlogits_bound=2.0 ** -40).clone()
    assert bool(torch.isfinite(gd_ov).all()) and float((gd_ov - g6).abs().max()) / scale <= 5e-6
