# -*- coding: utf-8 -*-
"""Shared parity checks: HIP path (real GPU library or host emulator build) vs the oracle / golden
fixtures.  Gates (north_star "within 1e-4 fp32"):
    logits max-abs <= 1e-4 ; loss <= 1e-5 abs ; grads <= 1e-4 of each tensor's max-abs ;
    weights after Adam <= 1e-2 * lr (SURVEY's 1e-6 abs at the reference default lr = 1e-4; Adam's
    normalised update m/(sqrt(v)+eps) of an element whose gradient is ~0 is sign-like, i.e.
    ill-conditioned in the gradient's last bits, so the gate has to scale with the step size).
"""
import ctypes

import torch

from oracle import wavenet_oracle as O
from pytorchwavenetvocoder_amd import _lib
from pytorchwavenetvocoder_amd.engine import WaveNetEngine, flat_to_state, load_state_into_flat
from pytorchwavenetvocoder_amd.nets import WaveNet
from pytorchwavenetvocoder_amd.optim import FusedAdam
from tests.golden_util import rel_to_max

TOL_LOGITS = 1e-4
TOL_LOSS = 1e-5
TOL_GRAD = 1e-4
TOL_ADAM_REL_LR = 1e-2  # |w - w_ref| <= 1e-2 * lr  (= SURVEY's 1e-6 at the reference's lr=1e-4)


def check_golden_case(g, lib, device, flags=0, layers_per_bucket=0):
    """Engine-level (C-ABI) forward / loss / backward against a golden case."""
    eng = WaveNetEngine(*g.cfg.as_tuple(), device=device, library=lib)
    eng.flags = flags
    assert eng.receptive_field == g.rf
    load_state_into_flat(eng, g.params)
    x, h, t = g.x.to(device), g.h.to(device), g.t.to(device)
    logits = eng.forward(x, h)
    assert tuple(logits.shape) == (g.B, g.cfg.n_quantize, g.T)
    err = float((logits.transpose(1, 2).cpu() - g.logits).abs().max())
    assert err <= TOL_LOGITS, "logits max-abs err %g" % err
    loss, dl = eng.loss(logits, t)
    assert abs(float(loss.cpu()) - g.loss) <= TOL_LOSS
    grads = flat_to_state(eng, eng.backward(dl, layers_per_bucket=layers_per_bucket).cpu(), O.param_shapes(g.cfg))
    for k, ref in g.grads.items():
        if ref is None:
            assert float(grads[k].abs().max()) == 0.0, k
        else:
            e = rel_to_max(grads[k], ref)
            assert e <= TOL_GRAD, "%s: grad rel err %g" % (k, e)
    return eng


def check_module_training(g, lib, device):
    """nn.Module-level: autograd path, fused path and FusedAdam against the golden after-state."""
    model = WaveNet(*g.cfg.as_tuple(), _library=lib)
    assert list(model.state_dict().keys()) == list(O.param_shapes(g.cfg).keys())
    model.load_state_dict(g.params)
    model.to(device)
    x, h, t = g.x.to(device), g.h.to(device), g.t.to(device)
    out = model(x, h)
    assert tuple(out.shape) == (g.B, g.T, g.cfg.n_quantize)
    loss = torch.nn.CrossEntropyLoss()(out[:, g.rf:].contiguous().view(-1, g.cfg.n_quantize),
                                       t[:, g.rf:].contiguous().view(-1))
    loss.backward()
    assert abs(float(loss) - g.loss) <= TOL_LOSS
    for k, p in model.named_parameters():
        ref = g.grads[k]
        if ref is None:
            assert p.grad is None, k
        else:
            assert rel_to_max(p.grad.cpu(), ref) <= TOL_GRAD, k
    model.zero_grad()
    opt = FusedAdam(model, lr=g.adam_lr, weight_decay=g.wd)
    for s in range(g.adam_steps):
        l = model.loss_and_backward(x, h, t)
        opt.step()
        assert abs(float(l.cpu()) - float(g.z["loss_step%d" % s])) <= TOL_LOSS
    for k, v in model.state_dict().items():
        e = float((v.cpu() - g.after[k]).abs().max())
        assert e <= TOL_ADAM_REL_LR * g.adam_lr, "%s: |dw| err %g (lr %g)" % (k, e, g.adam_lr)
    return model, opt


def check_reference_training_loop(g, lib, device):
    """The reference's own loop (train.py:527-540): model(x, h) -> CrossEntropyLoss on [:, rf:] -> backward ->
    torch.optim.Adam(model.parameters()).step(), i.e. autograd + a stock optimizer on the flat-buffer views,
    must land on the reference's weights after the golden number of steps."""
    model = WaveNet(*g.cfg.as_tuple(), _library=lib)
    model.load_state_dict(g.params)
    model.to(device)
    x, h, t = g.x.to(device), g.h.to(device), g.t.to(device)
    opt = torch.optim.Adam(model.parameters(), lr=g.adam_lr, weight_decay=g.wd)
    crit = torch.nn.CrossEntropyLoss()
    Q = g.cfg.n_quantize
    for s in range(g.adam_steps):
        out = model(x, h)
        loss = crit(out[:, g.rf:].contiguous().view(-1, Q), t[:, g.rf:].contiguous().view(-1))
        opt.zero_grad()
        loss.backward()
        opt.step()
        assert abs(float(loss) - float(g.z["loss_step%d" % s])) <= TOL_LOSS
    for k, v in model.state_dict().items():
        e = float((v.cpu() - g.after[k]).abs().max())
        assert e <= TOL_ADAM_REL_LR * g.adam_lr, "%s: |dw| err %g (lr %g)" % (k, e, g.adam_lr)


KINK_MARGIN = 1e-5  # required min |pre-ReLU| on loss positions (10x the observed fp32 forward error)


def pick_instance(cfg, B, T, seed, scale, tries=40):
    """Seeded synthetic (params, x, h, t) whose ReLU-kink margin is >= KINK_MARGIN (see
    oracle.relu_kink_margin: gradient parity is only defined away from the ReLU kinks)."""
    for i in range(tries):
        sd = seed + 1009 * i
        params = O.random_params(cfg, sd, scale=scale)
        x, h, t = O.synthetic_batch(cfg, B, T, sd + 1)
        margin = O.relu_kink_margin(cfg, params, x, h)
        if margin >= KINK_MARGIN:
            return params, x, h, t, margin, sd
    raise RuntimeError("no instance with ReLU margin >= %g in %d tries" % (KINK_MARGIN, tries))


def run_oracle_vs_engine(cfg_tuple, B, T, seed, lib, device, flags=0, scale=0.1, check_grads=True):
    """Live oracle vs HIP path on seeded synthetic inputs (sizes the oracle finishes in seconds)."""
    cfg = O.OracleConfig(*cfg_tuple)
    params, x, h, t, margin, sd = pick_instance(cfg, B, T, seed, scale)
    loss_ref, logits_ref, grads_ref = O.train_step(cfg, params, None, x, h, t)
    eng = WaveNetEngine(*cfg_tuple, device=device, library=lib)
    eng.flags = flags
    load_state_into_flat(eng, params)
    logits = eng.forward(x.to(device), h.to(device))
    err = float((logits.transpose(1, 2).cpu() - logits_ref).abs().max())
    assert err <= TOL_LOGITS, "logits max-abs err %g" % err
    loss, dl = eng.loss(logits, t.to(device))
    assert abs(float(loss.cpu()) - float(loss_ref)) <= TOL_LOSS
    worst = 0.0
    if check_grads:
        grads = flat_to_state(eng, eng.backward(dl).cpu(), O.param_shapes(cfg))
        for k, ref in grads_ref.items():
            if ref is None:
                assert float(grads[k].abs().max()) == 0.0, k
            else:
                e = rel_to_max(grads[k], ref)
                worst = max(worst, e)
                assert e <= TOL_GRAD, "%s: grad rel err %g (seed %d, ReLU margin %.3g)" % (k, e, sd, margin)
    return err, worst


def gemm_reference(M, N, K, A, B):
    return A @ B
