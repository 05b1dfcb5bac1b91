#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Does the arithmetic of the weight-gradient contractions change a TRAINING RUN?  (WN_FLAG_DW_F16PAIR.)

Trains the benchmark's model (30 layers, 64/256 channels, B = 8 windows of batch_len 20000 -> T = 23040) from one seed for N
Adam steps on a fresh synthetic minibatch per step (the reference's loop, train.py:527-540), once per arithmetic:
  six   six bf16 products everywhere                     (engine.flags & ~FLAG_DW_F16PAIR)
  f16   two fp16 pieces / three products for the weight gradients, the engine's default
  bf3   two bf16 pieces / three products (WN_FLAG_DW_3PRODUCT, opt-in)
and prints the loss curves side by side, the largest loss difference to `six`, and the distance of the final weights from
`six` in units of lr x steps (what N sign-like Adam updates could move a weight at most).  Two runs of `six` itself are
bit-identical (fixed-order reductions), so every difference printed is the arithmetic's.

    python tools/dw_mode_training_run.py [--steps 200] [--lr 1e-4]      (GPU, ~15 s)
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pytorchwavenetvocoder_amd import _lib  # noqa: E402
from pytorchwavenetvocoder_amd.engine import DEFAULT_FLAGS  # noqa: E402
from pytorchwavenetvocoder_amd.nets import WaveNet, initialize  # noqa: E402
from pytorchwavenetvocoder_amd.optim import FusedAdam  # noqa: E402


def run(flags, steps, lr, B, T, dev):
    torch.manual_seed(1)
    model = WaveNet(256, 80, 64, 256, 10, 3, 2, 80)
    model.apply(initialize)
    model.to(dev)
    model.engine.flags = flags
    opt = FusedAdam(model, lr=lr)
    g = torch.Generator().manual_seed(11)
    losses = []
    for _ in range(steps):
        xx = torch.randint(0, 256, (B, T + 1), generator=g)
        x, t = xx[:, :-1].contiguous().to(dev), xx[:, 1:].contiguous().to(dev)
        h = torch.randn(B, 80, T // 80, generator=g).to(dev)
        losses.append(model.loss_and_backward(x, h, t))
        opt.step()
    losses = torch.stack([l.reshape(()) for l in losses]).cpu()
    flat = model.engine.flat_params.detach().clone().cpu()
    return losses, flat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--T", type=int, default=23040)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    six = DEFAULT_FLAGS & ~_lib.NARROW_FLAGS
    # round 6: "default" = fp16 pairs for the weight gradients, the k_gemm6 contractions and the fused forward block (DEFAULT_FLAGS)
    modes = [("six", six), ("six_again", six), ("f16", six | _lib.FLAG_DW_F16PAIR), ("bf3", six | _lib.FLAG_DW_3PRODUCT),
             ("default", DEFAULT_FLAGS)]
    res = {n: run(f, a.steps, a.lr, a.batch, a.T, dev) for n, f in modes}
    l6, w6 = res["six"]
    out = {"steps": a.steps, "lr": a.lr, "B": a.batch, "T": a.T, "loss_first": float(l6[0]), "loss_last": float(l6[-1])}
    for n, _ in modes[1:]:
        l, w = res[n]
        out[n] = {"max_abs_loss_diff": float((l - l6).abs().max()), "loss_last": float(l[-1]),
                  "max_weight_diff_over_lr_steps": float((w - w6).abs().max()) / (a.lr * a.steps),
                  "rms_weight_diff_over_lr_steps": float((w - w6).pow(2).mean().sqrt()) / (a.lr * a.steps),
                  "bit_identical": bool(torch.equal(l, l6) and torch.equal(w, w6))}
    print(json.dumps(out, indent=1))
    every = max(1, a.steps // 10)
    print("step   " + "  ".join("%-12s" % n for n, _ in modes))
    for s in list(range(0, a.steps, every)) + [a.steps - 1]:
        print("%5d  " % s + "  ".join("%-12.7f" % float(res[n][0][s]) for n, _ in modes))


if __name__ == "__main__":
    main()
