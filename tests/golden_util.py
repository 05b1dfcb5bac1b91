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


# ---- window-slicer fixtures (tests/golden/slicer.npz, written by tests/golden/make_slicer_golden.py) ----
SLICER_U, SLICER_DIM, SLICER_RF, SLICER_Q = 10, 5, 61, 256
# name: (batch_length, batch_size, use_upsampling_layer, use_speaker_code, stats dtype, batches recorded)
SLICER_CASES = {
    "utt_up": (None, 1, True, False, "float32", 5),
    "utt_noup": (None, 1, False, False, "float32", 5),
    "win_up": (200, 2, True, False, "float32", 6),
    "win_noup": (200, 2, False, False, "float32", 6),
    "win_up_spk": (170, 3, True, True, "float32", 4),
    "win_up_f64stats": (200, 2, True, False, "float64", 3),
}


def slicer_corpus():
    """Synthetic 3-utterance corpus of the window-slicer fixtures: int16 waveforms (so that a wav file
    and an in-memory float32 copy are bit-equal after /32768), float32 features, a speaker code per
    utterance and non-trivial statistics.  Utterance 0 has more samples than frames * U, utterance 1
    fewer (both branches of validate_length, reference train.py:35-64), utterance 2 matches."""
    rs = np.random.RandomState(77)
    utts = []
    for i, (frames, extra) in enumerate([(64, 13), (71, -27), (58, 0)]):
        wav = (rs.uniform(-0.9, 0.9, size=frames * SLICER_U + extra) * 32767).astype(np.int16)
        feat = rs.standard_normal((frames, SLICER_DIM)).astype(np.float32)
        code = rs.standard_normal((2,)).astype(np.float32)
        utts.append((wav, feat, code))
    mean = rs.standard_normal(SLICER_DIM) * 0.3
    scale = rs.uniform(0.5, 2.0, SLICER_DIM)
    return utts, mean, scale
