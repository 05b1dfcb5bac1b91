# -*- coding: utf-8 -*-
"""GPU parity of the two aux-gradient modes of wn_backward: WN_FLAG_AUX_FUSED (partial sums inside the gate kernel; the
engine's default since round 2, so every default-mode test exercises it) and the separate wn_aux_bwd launch (flags
without it) -- each against the oracle and the golden case, and against each other at round-off."""
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


def test_aux_gradient_modes_agree_and_match_the_oracle():
    """WN_FLAG_AUX_FUSED (the gate kernel leaves the partial sums of the aux-path gradients, dP is not re-read) and the
    separate-launch mode: config-2 model on an oracle-sized window against the oracle, against each other at round-off;
    run to run bitwise."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib as L
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    cfg_t = (256, 80, 64, 256, 10, 3, 2, 80)
    from pytorchwavenetvocoder_amd.engine import DEFAULT_FLAGS
    assert DEFAULT_FLAGS & L.FLAG_AUX_FUSED
    # ... each with the one-launch-per-layer backward chain (default) and with the former gate' + dX launch pair
    for flags in (L.FLAG_AUX_FUSED, 0, L.FLAG_AUX_FUSED | L.FLAG_NO_CHAIN, L.FLAG_NO_CHAIN):
        e, gerr = PC.run_oracle_vs_engine(cfg_t, 1, 3120, 21, _lib(), DEV, scale=0.05, flags=flags)
        print("aux mode %d: logits err %.3g, worst grad rel err %.3g" % (flags, e, gerr))
        PC.check_golden_case(GoldenCase("r64_k2_up"), _lib(), DEV, flags=flags)
    cfg = O.OracleConfig(*cfg_t)
    params, x, h, t, margin, sd = PC.pick_instance(cfg, 2, 3120 + 80, 41, 0.05)
    x, h, t = x.to(DEV), h.to(DEV), t.to(DEV)
    res = []
    for flags in (L.FLAG_NO_CHAIN, L.FLAG_AUX_FUSED, L.FLAG_AUX_FUSED):
        eng = WaveNetEngine(*cfg_t, device=DEV, library=_lib())
        eng.flags = flags
        load_state_into_flat(eng, params)
        logits = eng.forward(x, h)
        loss, dl = eng.loss(logits, t)
        res.append(eng.backward(dl, layers_per_bucket=10).clone())
    assert torch.equal(res[1], res[2])
    assert float((res[1] - res[0]).abs().max()) <= 1e-5 * float(res[0].abs().max())


def test_chain_kernel_ragged_shapes():
    """The one-launch-per-layer backward chain on shapes the big cases do not have: a half-full last tile (T % 32 == 16),
    three sequences, kernel_size 1, a single layer (head + tail only), no upsampling layer (aux partials fall back)."""
    from pytorchwavenetvocoder_amd import _lib as L
    A = L.FLAG_AUX_FUSED
    for cfg_t, B, T, seed, flags in [((64, 6, 64, 32, 2, 2, 2, 16), 3, 80, 42, A), ((64, 6, 64, 32, 3, 1, 1, 16), 2, 48, 41, A),
                                     ((64, 6, 64, 32, 1, 1, 2, 16), 1, 32, 44, A), ((64, 6, 64, 64, 3, 2, 2, 0), 1, 70, 43, A),
                                     ((64, 6, 64, 32, 2, 2, 2, 16), 3, 80, 42, 0), ((256, 8, 64, 128, 5, 2, 2, 16), 2, 304, 45, A)]:
        e, g = PC.run_oracle_vs_engine(cfg_t, B, T, seed, _lib(), DEV, flags=flags, scale=0.2 if cfg_t[3] < 128 else 0.1)
        print("chain ragged", cfg_t, B, T, "logits %.3g grads %.3g" % (e, g))


def test_loss_window_backward_vs_full_backward():
    """wn_backward_window (what loss_and_backward calls: post-net / skip part over [rf rounded down to 128, T) only)
    against the full-range wn_backward on the same dlogits: config-2 model (rf = 3070 -> t0 = 2944), three sequences of 6400
    (ragged last 128-column tile of the window), every launch mode; dSkip exactly zero in front of the window; T barely past
    the receptive field (a window of one tile); and the module-level training half-step uses it."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib as L
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    from pytorchwavenetvocoder_amd.nets import WaveNet
    cfg_t = (256, 80, 64, 256, 10, 3, 2, 80)
    cfg = O.OracleConfig(*cfg_t)
    rf = cfg.receptive_field
    for B, T in ((3, 6400), (2, 3120)):
        # both calls take the same ReLU sub-gradients (the saved activations), so no kink-free instance is needed
        params = O.random_params(cfg, 43, scale=0.05)
        x, h, t = (v.to(DEV) for v in O.synthetic_batch(cfg, B, T, 44))
        for flags in (L.FLAG_AUX_FUSED, L.FLAG_AUX_FUSED | L.FLAG_NO_CHAIN, L.FLAG_NO_FUSED, L.FLAG_EXACT_MFMA):
            eng = WaveNetEngine(*cfg_t, device=DEV, library=_lib())
            eng.flags = flags
            load_state_into_flat(eng, params)
            logits = eng.forward(x, h)
            loss, dl = eng.loss(logits, t)
            full = eng.backward(dl).clone()
            win = eng.backward(dl, t_first=rf).clone()
            dSk = eng.saved(L.WS_DSKIP)
            assert float(dSk[:, :, :rf].abs().max()) == 0.0 and float(dSk[:, :, rf:].abs().max()) > 0.0
            err = float((win - full).abs().max()) / float(full.abs().max())
            print("loss window B=%d T=%d flags=%d: |window - full| / max = %.3g" % (B, T, flags, err))
            assert err <= 2e-6
            assert torch.equal(win, eng.backward(dl, t_first=rf))   # run to run bitwise
    m = WaveNet(*cfg_t).to(DEV)
    load_state_into_flat(m.engine, params)
    m.loss_and_backward(x, h, t)
    g_mod = m.engine.grads().clone()
    logits = m.engine.forward(x, h)
    loss, dl = m.engine.loss(logits, t)
    g_sep = m.engine.backward(dl, t_first=rf)   # (the module takes its dlogits from the loss epilogue of conv_post_2: round-off apart)
    assert float((g_mod - g_sep).abs().max()) <= 5e-6 * float(g_sep.abs().max())


def test_cross_entropy_epilogue_vs_separate_loss_kernel():
    """wn_forward_loss on the GPU: the cross-entropy as the epilogue of the conv_post_2 contraction (logits never written)
    against wn_forward + wn_softmax_ce_loss -- config-2 model on three ragged sequences (last 128-column block partly
    outside T), several loss windows and gradient scales, no gradient buffer; then one full training half-step through
    the module against the same step assembled from the separate entry points; bitwise run to run."""
    import ctypes
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    from pytorchwavenetvocoder_amd.nets import WaveNet
    cfg_t = (256, 80, 64, 256, 10, 3, 2, 80)
    cfg = O.OracleConfig(*cfg_t)
    rf = cfg.receptive_field
    B, T = 3, 3120 + 80 * 13       # 4160 = 32.5 blocks of 128 columns
    params = O.random_params(cfg, 47, scale=0.05)
    x, h, t = (v.to(DEV) for v in O.synthetic_batch(cfg, B, T, 48))
    eng = WaveNetEngine(*cfg_t, device=DEV, library=_lib())
    load_state_into_flat(eng, params)
    assert eng.lib.wn_forward_loss_fused(ctypes.byref(eng.cfg), B, T, eng.flags) == 1
    logits = eng.forward(x, h)
    for t_start, gs in ((None, 1.0), (rf + 77, 0.125), (0, 1.0)):
        l0, d0 = eng.loss(logits, t, t_start=t_start, grad_scale=gs)
        l1, d1 = eng.forward_loss(x, h, t, t_start=t_start, grad_scale=gs)
        l2, d2 = eng.forward_loss(x, h, t, t_start=t_start, grad_scale=gs)
        e_l = abs(float(l1) - float(l0)) / max(1.0, abs(float(l0)))
        e_d = float((d1 - d0).abs().max()) / float(d0.abs().max())
        print("CE epilogue t_start=%s: loss rel err %.3g, dlogits err / max %.3g" % (t_start, e_l, e_d))
        assert e_l <= 2e-6 and e_d <= 1e-5
        assert torch.equal(l1, l2) and torch.equal(d1, d2)
        ts = rf if t_start is None else t_start
        assert ts == 0 or float(d1[:, :, :ts].abs().max()) == 0.0
    l3, none = eng.forward_loss(x, h, t, want_grad=False)
    assert none is None and abs(float(l3) - float(eng.loss(logits, t)[0])) <= 2e-6 * abs(float(l3))
    # the module's training half-step = forward_loss + windowed backward
    m = WaveNet(*cfg_t).to(DEV)
    load_state_into_flat(m.engine, params)
    lm = m.loss_and_backward(x, h, t)
    g_mod = m.engine.grads().clone()
    logits = m.engine.forward(x, h)
    ls, dl = m.engine.loss(logits, t)
    g_sep = m.engine.backward(dl, t_first=rf)
    assert abs(float(lm) - float(ls)) <= 2e-6 * abs(float(ls))
    assert float((g_mod - g_sep).abs().max()) <= 5e-6 * float(g_sep.abs().max())
