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


def check_golden_case(g, lib, device, flags=None, layers_per_bucket=0):
    """Engine-level (C-ABI) forward / loss / backward against a golden case."""
    eng = WaveNetEngine(*g.cfg.as_tuple(), device=device, library=lib)
    if flags is not None:   # None: the engine's default launch mode (engine.DEFAULT_FLAGS)
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
    grads = flat_to_state(eng, eng.backward(dl, layers_per_bucket=layers_per_bucket, t_first=eng.receptive_field).cpu(), O.param_shapes(g.cfg))
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


def run_oracle_vs_engine(cfg_tuple, B, T, seed, lib, device, flags=None, scale=0.1, check_grads=True):
    """Live oracle vs HIP path on seeded synthetic inputs (sizes the oracle finishes in seconds)."""
    cfg = O.OracleConfig(*cfg_tuple)
    params, x, h, t, margin, sd = pick_instance(cfg, B, T, seed, scale)
    loss_ref, logits_ref, grads_ref = O.train_step(cfg, params, None, x, h, t)
    eng = WaveNetEngine(*cfg_tuple, device=device, library=lib)
    if flags is not None:
        eng.flags = flags
    load_state_into_flat(eng, params)
    logits = eng.forward(x.to(device), h.to(device))
    err = float((logits.transpose(1, 2).cpu() - logits_ref).abs().max())
    assert err <= TOL_LOGITS, "logits max-abs err %g" % err
    loss, dl = eng.loss(logits, t.to(device))
    assert abs(float(loss.cpu()) - float(loss_ref)) <= TOL_LOSS
    worst = 0.0
    if check_grads:
        grads = flat_to_state(eng, eng.backward(dl, t_first=eng.receptive_field).cpu(), O.param_shapes(cfg))   # the training step's call (loss window)
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


def run_fullsize_vs_oracle(cfg_tuple, B, T, seed, lib, device, flag_sets, scale=0.05, threads=32):
    """Live oracle vs HIP path at sizes where ReLU kinks are CERTAIN to occur (BASELINE config 2: 8 x 19970 x 512 ReLU
    inputs -- hundreds within 1e-5 of zero): forward / loss parity on random trained-scale weights as everywhere else,
    every layer's saved input x_l against the oracle's, and gradient parity of EVERY tensor for the same sub-gradient
    choice at the kinks -- the oracle back-propagates through its two ReLUs with the HIP path's own (output > 0) masks
    (``oracle.forward(relu_masks=...)``), and every element where that choice differs from the oracle's own sign must be
    within 1e-5 of the kink in the oracle.  ``flag_sets``: wn_backward modes to check against the one oracle run (they
    must not change the forward).  Returns a dict of the observed errors."""
    import os
    cfg = O.OracleConfig(*cfg_tuple)
    params = O.random_params(cfg, seed, scale=scale)
    x, h, t = O.synthetic_batch(cfg, B, T, seed + 1)
    eng = WaveNetEngine(*cfg_tuple, device=device, library=lib)
    load_state_into_flat(eng, params)
    eng.flags = flag_sets[0]
    xd, hd, td = x.to(device), h.to(device), t.to(device)
    logits = eng.forward(xd, hd)
    loss_sep, dl_sep = eng.loss(logits, td)
    # the training step's own call: forward + loss in one, the cross-entropy as the epilogue of conv_post_2 (wn_forward_loss);
    # its loss and dlogits are what is compared with the oracle below, and against the separate entry points here
    loss, dl = eng.forward_loss(xd, hd, td)
    assert abs(float(loss.cpu()) - float(loss_sep.cpu())) <= 2e-6 * max(1.0, abs(float(loss_sep.cpu())))
    assert float((dl - dl_sep).abs().max()) <= 1e-5 * float(dl_sep.abs().max())
    del dl_sep
    m_skip = (eng.saved(_lib.WS_RELU_SKIP) > 0).float().cpu()
    m_post = (eng.saved(_lib.WS_RELU_POST1) > 0).float().cpu()
    try:
        navail = len(os.sched_getaffinity(0))
    except AttributeError:
        navail = os.cpu_count() or 1
    old_threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(threads, navail)))   # oneDNN convolutions of this size slow down beyond ~32 threads
    try:
        loss_ref, logits_ref, grads_ref, inter = O.train_step(cfg, params, None, x, h, t, relu_masks=(m_skip, m_post),
                                                             return_intermediates=True)
    finally:
        torch.set_num_threads(old_threads)
    out = {}
    out["logits"] = float((logits.transpose(1, 2).cpu() - logits_ref).abs().max())
    assert out["logits"] <= TOL_LOGITS, "logits max-abs err %g" % out["logits"]
    out["loss"] = abs(float(loss.cpu()) - float(loss_ref))
    assert out["loss"] <= TOL_LOSS, "loss err %g" % out["loss"]
    # sub-gradient choices that differ from the oracle's own sign: only ever at the kink
    rf = cfg.receptive_field
    out["kink_flips"] = 0
    for pre, m in ((inter["skip_sum"], m_skip), (inter["post1_pre"], m_post)):
        differ = ((pre > 0).float() != m)
        n = int(differ.sum())
        out["kink_flips"] += n
        if n:
            worst = float(pre[differ].abs().max())
            assert worst <= 1e-5, "ReLU mask differs from the oracle's at an element %g away from the kink" % worst
    out["near_kink_1e-5"] = int(((inter["skip_sum"][:, :, rf:].abs() < 1e-5).sum() + (inter["post1_pre"][:, :, rf:].abs() < 1e-5).sum()))
    # every layer's input x_l (x_0 = front conv output, x_l = residual output of layer l-1; wavenet.py:534-536)
    X = eng.saved(_lib.WS_X)
    worst_x = 0.0
    for l in range(len(cfg.dilations)):
        ref = inter["x0"] if l == 0 else inter["layer_out"][l - 1]
        e = float((X[l].cpu() - ref).abs().max())
        worst_x = max(worst_x, e)
        assert e <= TOL_LOGITS, "layer %d input max-abs err %g" % (l, e)
    out["layer_inputs"] = worst_x
    out["grads"] = {}
    out["forward_modes"] = {}
    fwd_flags = flag_sets[0]
    for flags in flag_sets:
        eng.flags = flags
        if (flags ^ fwd_flags) & (_lib.FLAG_MM_F16PAIR | _lib.FLAG_FUSED_F16PAIR | _lib.FLAG_CHAIN_F16PAIR):
            # a flag set that changes the FORWARD arithmetic (WN_FLAG_MM_F16PAIR / _FUSED_F16PAIR: contractions by the fp16 pair split):
            # its own forward against the same oracle run -- logits, loss -- and its backward under the SAME sub-gradient choice as
            # the oracle's masks: elements whose ReLU sign differs from the first forward's are genuine ties (asserted: within 1e-5
            # of the kink) and take the first forward's choice
            lg2 = eng.forward(xd, hd)
            e_lg = float((lg2.transpose(1, 2).cpu() - logits_ref).abs().max())
            assert e_lg <= TOL_LOGITS, "logits max-abs err %g (flags %d)" % (e_lg, flags)
            del lg2
            loss2, dl = eng.forward_loss(xd, hd, td)
            assert abs(float(loss2.cpu()) - float(loss_ref)) <= TOL_LOSS
            X2 = eng.saved(_lib.WS_X)
            worst_x2 = 0.0
            for l in range(len(cfg.dilations)):
                ref_x = inter["x0"] if l == 0 else inter["layer_out"][l - 1]
                e_x = float((X2[l].cpu() - ref_x).abs().max())
                worst_x2 = max(worst_x2, e_x)
                assert e_x <= TOL_LOGITS, "layer %d input max-abs err %g (flags %d)" % (l, e_x, flags)
            flips = 0
            for kind, m in ((_lib.WS_RELU_SKIP, m_skip), (_lib.WS_RELU_POST1, m_post)):
                sv = eng.saved(kind)[:, :, rf:]
                md = m[:, :, rf:].to(sv.device) > 0
                differ = (sv > 0) != md
                n = int(differ.sum())
                if n:
                    assert float(sv[differ].abs().max()) <= 1e-5
                    sv[differ] = torch.where(md[differ], torch.full_like(sv[differ], 1e-30), torch.zeros_like(sv[differ]))
                flips += n
            out["forward_modes"][flags] = {"logits": e_lg, "loss": abs(float(loss2.cpu()) - float(loss_ref)), "layer_inputs": worst_x2,
                                           "kink_ties_vs_first_forward": flips}
            fwd_flags = flags
        grads = flat_to_state(eng, eng.backward(dl, t_first=eng.receptive_field).cpu(), O.param_shapes(cfg))   # the training step's call (loss window)
        worst, worst_k = 0.0, None
        for k, ref in grads_ref.items():
            if ref is None:
                assert float(grads[k].abs().max()) == 0.0, k
            else:
                e = rel_to_max(grads[k], ref)
                if e > worst:
                    worst, worst_k = e, k
                assert e <= TOL_GRAD, "%s: grad rel err %g (flags %d)" % (k, e, flags)
        out["grads"][flags] = (worst, worst_k)
    return out


def run_mol_vs_restatement(cfg_tuple, n_mix, B, T, seed, lib, device, scale=0.05, threads=32, tol_grad=TOL_GRAD):
    """Mixture-of-logistics head (NOT in the reference: parity unpinned by it) against this repo's own restatement of the
    published formula (oracle.mol_nll).  With 65536 classes the published formula takes a bin's mass as the DIFFERENCE of two
    sigmoids one part in 1e5 apart, so its fp32 evaluation carries up to per cent of rounding error in the gradient; the kernel
    computes the same mass without the subtraction (wn_elem.hip: mol_component), which makes the fp64 evaluation of the
    restatement the checker and the fp32 one a reported figure:
      * network output and loss against the oracle's (fp32 network, fp32 restatement);
      * d(loss)/d(output) against the restatement in fp64 ON THE KERNEL'S OWN network output: 1e-4 of the maximum;
        against the fp32 restatement: no further than that evaluation is from fp64 itself (+1e-4);
      * every parameter gradient against the oracle's fp32 autograd of the network, fed the fp64 head gradient at ITS output,
        with the HIP path's own ReLU sub-gradient choice (run_fullsize_vs_oracle's method): ``tol_grad`` of a tensor's maximum."""
    import os
    import numpy as np
    cfg = O.OracleConfig(*cfg_tuple, out_channels=3 * n_mix)
    rf = cfg.receptive_field
    params = O.random_params(cfg, seed, scale=scale)
    x, h, _ = O.synthetic_batch(cfg, B, T, seed + 1)
    y = torch.from_numpy(np.random.RandomState(seed + 2).uniform(-1, 1, (B, T)).astype(np.float32))
    eng = WaveNetEngine(*cfg_tuple, device=device, library=lib, out_channels=3 * n_mix)
    load_state_into_flat(eng, params)
    out = eng.forward(x.to(device), h.to(device))
    loss, dout = eng.mol_loss(out, y.to(device))
    grads = flat_to_state(eng, eng.backward(dout, t_first=rf).cpu(), O.param_shapes(cfg))
    m_skip = (eng.saved(_lib.WS_RELU_SKIP) > 0).float().cpu()
    m_post = (eng.saved(_lib.WS_RELU_POST1) > 0).float().cpu()

    def head(o, dt):
        oi = o.detach().clone().to(dt).requires_grad_(True)
        l = O.mol_nll(oi, y.to(dt), start=rf)
        l.backward()
        return float(l.detach()), oi.grad.float()

    try:
        navail = len(os.sched_getaffinity(0))
    except AttributeError:
        navail = os.cpu_count() or 1
    old_threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(threads, navail)))
    try:
        leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        out_ref = O.forward(cfg, leaves, x, h, relu_masks=(m_skip, m_post))      # (B, T, 3 n_mix)
        loss_ref, g_ref = head(out_ref, torch.float64)
        out_ref.backward(gradient=g_ref)
        ok = out.transpose(1, 2).cpu().contiguous()
        _, g64 = head(ok, torch.float64)
        loss32, g32 = head(ok, torch.float32)
    finally:
        torch.set_num_threads(old_threads)
    r = {}
    dk = dout.transpose(1, 2).cpu()
    den = float(g64.abs().max())
    r["out"] = float((ok - out_ref.detach()).abs().max())
    r["loss_rel"] = abs(float(loss.cpu()) - loss_ref) / abs(loss_ref)
    r["dout_vs_fp64"] = float((dk - g64).abs().max()) / den
    r["dout_vs_fp32"] = float((dk - g32).abs().max()) / den
    r["fp32_restatement_vs_fp64"] = float((g32 - g64).abs().max()) / den
    r["dout_vs_oracle_network"] = float((dk - g_ref).abs().max()) / float(g_ref.abs().max())
    r["grad"], r["grad_key"] = 0.0, None
    for k, v in leaves.items():
        if v.grad is None or float(v.grad.abs().max()) == 0.0:
            assert float(grads[k].abs().max()) == 0.0, k
            continue
        e = rel_to_max(grads[k], v.grad)
        if e > r["grad"]:
            r["grad"], r["grad_key"] = e, k
    assert r["out"] <= TOL_LOGITS and r["loss_rel"] <= 1e-4 and r["dout_vs_fp64"] <= 1e-4, r
    assert r["dout_vs_fp32"] <= r["fp32_restatement_vs_fp64"] + 1e-4 and r["grad"] <= tol_grad, r
    return r


def launch_log(lib, fn):
    """Run ``fn()`` with the library's per-launch log on; returns {tag: launches}.  (On the GPU the log carries HIP-event
    times as well -- bench.py's kernel table; the host emulator keeps the counts.)"""
    import json
    lib.wn_prof_enable(1)
    try:
        fn()
    finally:
        lib.wn_prof_enable(0)
    if not getattr(lib, "is_emulator", False) and torch.cuda.is_available():
        torch.cuda.synchronize()   # the report reads HIP events: the launches must have completed
    need = lib.wn_prof_report(None, 0)
    buf = ctypes.create_string_buffer(max(need, 16))
    lib.wn_prof_report(buf, len(buf))
    return {k: v["count"] for k, v in json.loads(buf.value.decode() or "{}").items()}


def launch_sequence(lib, fn):
    """Run ``fn()`` with the per-launch log on; returns the tags in ISSUE ORDER, including the ``bucket_event`` marks
    wn_backward leaves where it records a gradient-bucket event (wn_prof_sequence)."""
    lib.wn_prof_enable(1)
    try:
        fn()
    finally:
        lib.wn_prof_enable(0)
    need = lib.wn_prof_sequence(None, 0)
    buf = ctypes.create_string_buffer(max(need, 16))
    lib.wn_prof_sequence(buf, len(buf))
    s = buf.value.decode()
    return s.split(",") if s else []
