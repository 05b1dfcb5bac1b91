# -*- coding: utf-8 -*-
"""Helpers to load the committed golden fixtures (outputs of the reference itself)."""
import os
from collections import OrderedDict

import numpy as np
import torch

from oracle import wavenet_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["tiny_k2_up", "tiny_k3_noup", "tiny_init", "r64_k2_up", "r64_k3_up"]


class GoldenCase(object):
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.name = name
        self.z = z
        self.cfg = O.OracleConfig(*[int(v) for v in z["cfg"]])
        self.B, self.T, self.seed = int(z["B"]), int(z["T"]), int(z["seed"])
        self.mode = str(z["mode"])
        self.wd = float(z["wd"])
        self.rf = int(z["rf"])
        self.adam_lr = float(z["adam_lr"])
        self.adam_steps = int(z["adam_steps"])
        self.x, self.h, self.t = O.synthetic_batch(self.cfg, self.B, self.T, self.seed + 1000)
        if self.mode == "init":
            self.params = OrderedDict((k, torch.from_numpy(z["param/" + k].copy()))
                                      for k in O.param_shapes(self.cfg))
        else:
            self.params = O.random_params(self.cfg, self.seed)
        self.logits = torch.from_numpy(z["logits"])
        self.loss = float(z["loss"])
        self.grads = OrderedDict()
        for k in O.param_shapes(self.cfg):
            if "grad/" + k in z.files:
                self.grads[k] = torch.from_numpy(z["grad/" + k])
            else:
                assert "gradnone/" + k in z.files, k
                self.grads[k] = None
        self.after = OrderedDict((k, torch.from_numpy(z["after/" + k])) for k in O.param_shapes(self.cfg))

    def clone_params(self):
        return OrderedDict((k, v.clone()) for k, v in self.params.items())


def rel_to_max(a, b):
    """max|a-b| / max|b|  (SURVEY.md 8d gradient gate)."""
    denom = float(b.abs().max())
    if denom == 0.0:
        return float((a - b).abs().max())
    return float((a - b).abs().max()) / denom
