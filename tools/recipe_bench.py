#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""training step of the recipe-size model (n_resch = 512, n_skipch = 256: egs/*/run.sh defaults) at the
benchmark's window (batch_length 20000 -> T = 23040), per-kernel HIP-event table included.

    python tools/recipe_bench.py [--batch 4] [--steps 5] [--aux 80]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorchwavenetvocoder_amd.nets import WaveNet, initialize  # noqa: E402
from pytorchwavenetvocoder_amd.optim import FusedAdam  # noqa: E402


def measure(resch=512, kernel_size=2, upsampling=80, T=23040, batch=4, steps=5, aux=80, device="cuda:0", with_kernels=True,
            n_mixture=0, layers_per_bucket=0):
    """Median-free quick timing of one training step (forward + CE + backward + Adam) of a 30-layer model of the given
    geometry on synthetic data; returns a dict (ms_per_step, samples_per_sec, approx_train_tflops, per-kernel table)."""
    dev = torch.device(device)
    torch.manual_seed(1)
    R, S, A, U, L, K = resch, 256, aux, upsampling, 30, kernel_size
    model = WaveNet(256, A, R, S, 10, 3, K, U, n_mixture=n_mixture)
    model.apply(initialize)
    model.to(dev)
    B = batch
    g = torch.Generator().manual_seed(7)
    xx = torch.randint(0, 256, (B, T + 1), generator=g)
    x, t = xx[:, :-1].contiguous().to(dev), xx[:, 1:].contiguous().to(dev)
    h = torch.randn(B, A, T // U, generator=g).to(dev)
    y = (torch.rand(B, T, generator=g) * 2 - 1).to(dev)   # mixture head: the waveform value of the next sample
    opt = FusedAdam(model, lr=1e-4)

    def step():
        loss = (model.mol_loss_and_backward(x, h, y, layers_per_bucket=layers_per_bucket) if n_mixture > 0
                else model.loss_and_backward(x, h, t, layers_per_bucket=layers_per_bucket))
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flop_fwd = 2.0 * B * T * (L * (2 * R * R * K + 2 * A * R / U + R * S + R * R) + S * S + S * 256)
    out = {"model": "%d/%d, A=%d, K=%d, U=%d, 30 layers%s" % (R, S, A, K, U, ", %d-component mixture-of-logistics head" % n_mixture if n_mixture else ""),
           "B": B, "T": T, "rf": model.receptive_field,
           "ms_per_step": dt * 1e3, "samples_per_sec": B * (T - model.receptive_field) / dt,
           "approx_train_tflops": 3 * flop_fwd / dt / 1e12, "loss": float(loss)}
    if with_kernels:
        lib = model.engine.lib
        lib.wn_prof_enable(1)
        step()
        torch.cuda.synchronize()
        lib.wn_prof_enable(0)
        need = lib.wn_prof_report(None, 0)
        buf = ctypes.create_string_buffer(max(need, 16))
        lib.wn_prof_report(buf, len(buf))
        prof = json.loads(buf.value.decode() or "{}")
        out["kernels"] = {k: {"launches": v["count"], "ms": v["ms"],
                              "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["flops"] > 0 and v["ms"] > 0 else None,
                              "GBps": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["bytes"] > 0 and v["ms"] > 0 else None}
                          for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
    del model, opt
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--aux", type=int, default=80)
    ap.add_argument("--resch", type=int, default=512)
    ap.add_argument("--kernel-size", type=int, default=2)
    ap.add_argument("--upsampling", type=int, default=80)
    ap.add_argument("--T", type=int, default=23040, help="model inputs per window (BASELINE configs[3]: 26112)")
    ap.add_argument("--lpb", type=int, default=0, help="layers per gradient bucket / weight-gradient launch group (0: all layers)")
    ap.add_argument("--n-mixture", type=int, default=0, help="mixture-of-logistics head with this many components (0: softmax)")
    args = ap.parse_args()
    print(json.dumps(measure(args.resch, args.kernel_size, args.upsampling, args.T, args.batch, args.steps, args.aux,
                             n_mixture=args.n_mixture, layers_per_bucket=args.lpb)))


if __name__ == "__main__":
    main()
