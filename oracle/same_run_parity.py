# -*- coding: utf-8 -*-
"""TEST INFRASTRUCTURE.  The parity gate of north_star / BASELINE.md section 2 on ONE instance, in ONE job:

    "outputs match the reference PyTorch CPU forward on identical inputs within 1e-4 fp32, with the reference CPU path
     timed in the same run"

``reference_step``  runs ONE training step (train.py:527-540) of the reference's own module (``oracle/_ref``; the restatement
                    in ``wavenet_oracle.py`` where the copy is absent -- bit-identical) on the tensors it is given, times it, and
                    keeps what a comparison needs: logits, loss, every gradient tensor, the weights after the Adam step, and the
                    outputs of the two ReLUs of ``_postprocess`` (wavenet.py:519,521; taken with forward PRE-hooks on
                    ``conv_post_1`` / ``conv_post_2`` -- the module itself is not touched).
``gpu_step_vs_reference``  runs the same step on the HIP path from the same ``state_dict`` on the same ``x, h, t`` and compares.

ReLU kinks.  At the benchmark's size (8 x 19970 x 512 ReLU inputs) some pre-activations lie within round-off of zero, and two
correct fp32 evaluations pick different sub-gradients there; one differing choice moves whole gradient rows by O(1 / positions)
(DESIGN.md section 4).  Here THE REFERENCE'S choice is the common one: between the HIP forward and the HIP backward the saved
ReLU outputs of the (few) elements whose sign differs from the reference's are overwritten with the reference's values (0, or
its tiny positive value).  Every such element must be within ``KINK`` of zero ON BOTH SIDES (HIP value and reference value), i.e.
it is a genuine tie, and their number is reported (``kink_flips``).  Nothing else of the HIP step is touched; the reference's
step is the unmodified module.

Used by ``bench.py`` (cpu_baseline leg: the same reference step is the warm-up of the CPU timing) and ``tests/``; never by the
product.
"""
import time
from collections import OrderedDict

import torch

from . import ref_step as RS
from . import wavenet_oracle as O

SIGN_LIKE_GRAD = 1e-7  # 10 x Adam's eps: below it the first Adam update lr g / (|g| + eps) is sign-like (sensitivity ~1 / eps)
KINK = 1e-5           # a differing sub-gradient choice must be this close to the kink in BOTH evaluations
GATES = {"logits_maxabs": 1e-4, "loss_abs": 1e-5, "worst_grad_rel": 1e-4, "after_adam_maxabs_over_lr": 1e-2}


def reference_step(cfg_t, state, x, h, t, lr=1e-4, weight_decay=0.0, threads=None, keep_forward=True):
    """``threads``: torch CPU threads for this step (restored afterwards).  ``keep_forward`` False drops logits / ReLU outputs
    (a step that only serves ``reference_self_noise``).  One step of the reference's module from ``state`` on (x, h, t) (CPU tensors).  Returns a dict with ``kind``
    ("reference" | "port"), ``seconds`` (wall time of the step), ``loss``, ``logits`` (B, T, Q), ``grads`` {key: tensor | None},
    ``after`` {key: tensor}, ``relu_skip`` / ``relu_post1`` (B, S, T) = outputs of the two ReLUs, and ``trainer`` (reference
    kind: the ``ReferenceTrainer``, positioned after this step, so that a caller can go on timing steps)."""
    cfg = O.OracleConfig(*cfg_t)
    out = {}
    old_threads = torch.get_num_threads()
    if threads:
        torch.set_num_threads(int(threads))
    out["threads"] = torch.get_num_threads()
    try:
        _reference_step(cfg_t, cfg, state, x, h, t, lr, weight_decay, out)
    finally:
        torch.set_num_threads(old_threads)
    if not keep_forward:
        for k in ("logits", "relu_skip", "relu_post1"):
            out[k] = None
    return out


def _reference_step(cfg_t, cfg, state, x, h, t, lr, weight_decay, out):
    if RS.available():
        tr = RS.ReferenceTrainer(cfg_t, state=OrderedDict((k, v.clone()) for k, v in state.items()), lr=lr,
                                 weight_decay=weight_decay)
        cap = {}
        hooks = [tr.model.conv_post_1.register_forward_pre_hook(lambda m, a: cap.__setitem__("relu_skip", a[0].detach())),
                 tr.model.conv_post_2.register_forward_pre_hook(lambda m, a: cap.__setitem__("relu_post1", a[0].detach()))]
        t0 = time.time()
        loss, logits = tr.step(x, h, t)
        out["seconds"] = time.time() - t0
        for hk in hooks:
            hk.remove()
        out.update(kind="reference", loss=float(loss), logits=logits.detach(), trainer=tr,
                   grads=OrderedDict((k, None if p.grad is None else p.grad.detach().clone()) for k, p in tr.model.named_parameters()),
                   after=OrderedDict((k, v.detach().clone()) for k, v in tr.model.state_dict().items()),
                   relu_skip=cap["relu_skip"], relu_post1=cap["relu_post1"])
    else:
        params = OrderedDict((k, v.clone()) for k, v in state.items())
        opt = O.OracleAdam(lr=lr, weight_decay=weight_decay)
        t0 = time.time()
        loss, logits, grads, inter = O.train_step(cfg, params, opt, x, h, t, return_intermediates=True)
        out["seconds"] = time.time() - t0
        out.update(kind="port", loss=float(loss), logits=logits, trainer=None, grads=grads, after=params,
                   relu_skip=inter["skip_sum"].clamp_min(0), relu_post1=inter["post1_pre"].clamp_min(0))


def reference_self_noise(ref_a, ref_b, lr=1e-4):
    """The SAME reference step evaluated twice with different CPU thread counts (oneDNN / autograd partition their sums by
    thread): the reference's own reproducibility in the quantities the gates are stated in.  What it says about the after-Adam
    gate at the benchmark's size: an element whose gradient is below ~10 eps (eps = 1e-8 of Adam) takes a sign-like update
    lr g / (|g| + eps) whose sensitivity to the gradient is 1 / eps, so two fp32 evaluations of one model differ there by per cent
    of lr -- measured here, not argued."""
    wg, wk, wa, wak, n_over, over_g = 0.0, None, 0.0, None, 0, 0.0
    for k in ref_a["after"]:
        d = (ref_a["after"][k] - ref_b["after"][k]).abs()
        e = float(d.max()) / lr
        over = d > GATES["after_adam_maxabs_over_lr"] * lr
        n_over += int(over.sum())
        g = ref_b["grads"].get(k)
        if g is not None:
            if bool(over.any()):
                over_g = max(over_g, float(g[over].abs().max()))
            r = _rel_to_max(ref_a["grads"][k], g)
            if r > wg:
                wg, wk = r, k
        if e > wa:
            wa, wak = e, k
    return {"threads": [ref_a.get("threads"), ref_b.get("threads")], "loss_abs": abs(ref_a["loss"] - ref_b["loss"]),
            "worst_grad_rel": wg, "worst_grad_key": wk, "after_adam_maxabs_over_lr": wa, "after_adam_worst_key": wak,
            "after_adam_elements_over_gate": n_over, "after_adam_over_gate_max_abs_reference_grad": over_g,
            "note": "the reference's own step, same state and tensors, two thread counts: its reproducibility in the gates' units"}


def _rel_to_max(a, b):
    den = float(b.abs().max())
    return float((a - b).abs().max()) / (den if den > 0 else 1.0)


def gpu_step_vs_reference(model, make_optimizer, ref, x, h, t, init_state, flags, lr=1e-4, layers_per_bucket=0, free_ref_logits=False):
    """The same training step on the HIP path: ``model`` (the product's WaveNet on the GPU) is reset to ``init_state``,
    ``model(x, h)`` gives the logits, then the training half-step the benchmark times (``engine.forward_loss`` +
    ``engine.backward`` over the loss window, ``layers_per_bucket`` as timed) and one step of ``make_optimizer(model, lr)``.
    ``flags``: engine launch / arithmetic mode of this comparison.  Returns the ``parity`` dict of bench.py's JSON line."""
    from pytorchwavenetvocoder_amd import _lib
    eng = model.engine
    dev = eng.device
    old_flags = eng.flags
    rf = model.receptive_field
    try:
        model.load_state_dict(init_state)
        eng.flags = int(flags)
        xd, hd, td = x.to(dev), h.to(dev), t.to(dev)
        with torch.no_grad():
            logits = model(xd, hd)                                         # (B, T, Q) view, wn_forward
        res = {"mode_flags": int(flags)}
        # compared in pieces: the (B, T, Q) logits are 189 MB at the benchmark's size
        worst = 0.0
        for b in range(logits.size(0)):
            worst = max(worst, float((logits[b].cpu() - ref["logits"][b]).abs().max()))
        res["logits_maxabs"] = worst
        del logits
        loss, dl = eng.forward_loss(xd, hd, td)                           # the timed step's own forward: CE as the epilogue of conv_post_2
        model._fwd_serial += 1
        res["loss_abs"] = abs(float(loss.cpu()) - ref["loss"])
        res["loss"] = float(loss.cpu())
        res["loss_reference"] = ref["loss"]
        # the reference's sub-gradient choice at the ReLU kinks (see the module docstring)
        flips, dist = 0, 0.0
        for kind, key in ((_lib.WS_RELU_SKIP, "relu_skip"), (_lib.WS_RELU_POST1, "relu_post1")):
            sv = eng.saved(kind)[:, :, rf:]
            rv = ref[key][:, :, rf:].to(dev)
            differ = (sv > 0) != (rv > 0)
            n = int(differ.sum())
            if n:
                dist = max(dist, float(sv[differ].abs().max()), float(rv[differ].abs().max()))
                sv[differ] = rv[differ]
            flips += n
            del rv, differ
        res["kink_flips"] = flips
        res["kink_flip_max_distance"] = dist
        flat = eng.backward(dl, layers_per_bucket=layers_per_bucket, t_first=rf)
        worst, worst_k = 0.0, None
        for (k, p), (off, n, shape, dead) in zip(model.named_parameters(), model._param_slices):
            g_ref = ref["grads"][k]
            if g_ref is None:
                assert dead, k
                p.grad = None
                continue
            p.grad = flat[off:off + n].view(shape)
            e = _rel_to_max(p.grad.cpu(), g_ref)
            if e > worst:
                worst, worst_k = e, k
        res["worst_grad_rel"], res["worst_grad_key"] = worst, worst_k
        opt = make_optimizer(model, lr)
        opt.step()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        # Weights after the Adam step.  The first Adam update is lr g / (|g| + eps): for an element whose gradient is below
        # ~eps = 1e-8 it is sign-like, i.e. its sensitivity to the gradient is 1 / eps, and a difference at fp32 round-off of the
        # tensor's maximum moves it by per cent of lr.  The gate is applied to every element all the same; what is reported
        # beside it says where the worst element sits: its own reference gradient, and the largest reference gradient among the
        # elements over the gate (0.0 when there is none).
        worst, worst_k, n_over, n_all, over_g, worst_g = 0.0, None, 0, 0, 0.0, None
        for k, v in model.state_dict().items():
            d = (v.cpu() - ref["after"][k]).abs()
            e = float(d.max())
            over = d > GATES["after_adam_maxabs_over_lr"] * lr
            n_over += int(over.sum())
            n_all += d.numel()
            g_ref = ref["grads"].get(k)
            if g_ref is not None and bool(over.any()):
                over_g = max(over_g, float(g_ref[over].abs().max()))
            if e > worst:
                worst, worst_k = e, k
                worst_g = None if g_ref is None else (float(g_ref.flatten()[int(d.argmax())]), float(g_ref.abs().max()))
        res["after_adam_maxabs_over_lr"] = worst / lr
        res["after_adam_worst_key"] = worst_k
        res["after_adam_worst_element_reference_grad"] = None if worst_g is None else worst_g[0]
        res["after_adam_worst_tensor_reference_grad_max"] = None if worst_g is None else worst_g[1]
        res["after_adam_elements_over_gate"] = n_over
        res["after_adam_over_gate_max_abs_reference_grad"] = over_g
        res["after_adam_elements"] = n_all
        res["gates"] = dict(GATES, kink_flip_max_distance=KINK)
        met = {"logits": res["logits_maxabs"] <= GATES["logits_maxabs"], "loss": res["loss_abs"] <= GATES["loss_abs"],
               "grads": res["worst_grad_rel"] <= GATES["worst_grad_rel"],
               "after_adam": res["after_adam_maxabs_over_lr"] <= GATES["after_adam_maxabs_over_lr"], "kinks": dist <= KINK}
        res["gates_met"] = {k: bool(v) for k, v in met.items()}
        res["pass"] = bool(all(met.values()))            # all four gates as stated (+ every kink tie a genuine tie)
        # the after-Adam gate restricted to the elements where it is a statement about the GRADIENT: reference gradient >= 10 eps
        res["after_adam_well_conditioned_pass"] = bool(over_g < SIGN_LIKE_GRAD)
        return res
    finally:
        eng.flags = old_flags
