# -*- coding: utf-8 -*-
"""Fused Adam over the model's flat parameter buffer (one HIP launch per step).

Drop-in for ``torch.optim.Adam(model.parameters(), lr=..., weight_decay=...)`` as used by the
reference at train.py:457-460: same update rule (L2-in-gradient weight decay, bias correction),
parameters without a gradient are skipped (the dead last ``res_1x1``), and ``state_dict()`` /
``load_state_dict()`` use torch.optim.Adam's format (``{"state": {idx: {"step", "exp_avg",
"exp_avg_sq"}}, "param_groups": [...]}``) so checkpoints written by either optimizer resume with
the other (train.py:315-332,503-513).
"""
import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not hasattr(model, "engine"):
            raise TypeError("FusedAdam takes the WaveNet model (it updates the model's flat buffer)")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                        maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        self.model = model
        super(FusedAdam, self).__init__(list(model.parameters()), defaults)
        self._step = 0
        self._exp_avg = None
        self._exp_avg_sq = None

    def _buffers(self):
        eng = self.model.engine
        if self._exp_avg is None or self._exp_avg.device != eng.flat_params.device:
            old = (self._exp_avg, self._exp_avg_sq)
            self._exp_avg = torch.zeros_like(eng.flat_params)
            self._exp_avg_sq = torch.zeros_like(eng.flat_params)
            if old[0] is not None:
                self._exp_avg.copy_(old[0])
                self._exp_avg_sq.copy_(old[1])
        return self._exp_avg, self._exp_avg_sq

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        eng = self.model.engine
        m, v = self._buffers()
        flat_g = eng.grads()
        # gradients produced by the autograd path live in separate tensors: gather them
        lo = flat_g.data_ptr()
        hi = lo + flat_g.numel() * 4
        skipped = []   # live parameters without a gradient: torch.optim.Adam leaves them (and their moments) untouched
        for p, (off, n, shape, dead) in zip(self.model.parameters(), self.model._param_slices):
            if p.grad is None:
                if not dead:
                    flat_g[off:off + n].zero_()
                    sl = slice(off, off + n)
                    skipped.append((sl, eng.flat_params[sl].clone(), m[sl].clone(), v[sl].clone()))
                continue
            if not (lo <= p.grad.data_ptr() < hi):
                flat_g[off:off + n].copy_(p.grad.reshape(-1))
        group = self.param_groups[0]
        self._step += 1
        eng.adam_step(m, v, self._step, group["lr"], group["betas"], group["eps"], group["weight_decay"])
        for sl, p0, m0, v0 in skipped:   # the one launch covers the whole flat buffer: put the skipped slices back
            eng.flat_params[sl].copy_(p0)
            m[sl].copy_(m0)
            v[sl].copy_(v0)
        return loss

    # ---- torch.optim.Adam compatible (de)serialisation --------------------------------------
    def state_dict(self):
        m, v = self._buffers()
        state = {}
        if self._step > 0:
            for idx, (off, n, shape, dead) in enumerate(self.model._param_slices):
                if dead:
                    continue
                state[idx] = {"step": torch.tensor(float(self._step)),
                              "exp_avg": m[off:off + n].view(shape).clone(),
                              "exp_avg_sq": v[off:off + n].view(shape).clone()}
        groups = []
        for g in self.param_groups:
            gg = {k: val for k, val in g.items() if k != "params"}
            gg["params"] = list(range(len(self.model._param_slices)))
            groups.append(gg)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        m, v = self._buffers()
        m.zero_()
        v.zero_()
        step = 0
        for idx, st in sd["state"].items():
            off, n, shape, dead = self.model._param_slices[int(idx)]
            m[off:off + n].copy_(st["exp_avg"].reshape(-1))
            v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            step = max(step, int(float(st["step"])))
        self._step = step
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in sg:
                    g[k] = tuple(sg[k]) if k == "betas" else sg[k]
