#!/usr/bin/env python
"""BASELINE config 5: samples/sec of the autoregressive decode kernel on the cfg-2 model
(30 layers, 64 residual / 256 skip channels, 80-dim aux, U=80), argmax mode, seed token 128.

    python tools/decode_bench.py [--batch 1 8 64] [--samples 4000]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorchwavenetvocoder_amd.nets import WaveNet, initialize  # noqa: E402


def measure(model, B, n, device, layered=None):
    """Returns (generated samples/s per utterance stream, total, seconds) with the context walk
    (receptive-field prefill, wavenet.py:338-349) timed separately."""
    eng = model.engine
    x = torch.full((B, 1), 128, dtype=torch.int64, device=device)
    nf = (n + 1 + 79) // 80
    h = torch.randn(B, 80, nf, device=device)
    eng.decode(x, h, [8] * B, layered=layered)  # warm-up (packs weights, loads the kernel)
    torch.cuda.synchronize(device)
    t0 = time.time()
    eng.decode(x, h, [1] * B, layered=layered)   # context only: rf steps
    torch.cuda.synchronize(device)
    t_ctx = time.time() - t0
    t0 = time.time()
    eng.decode(x, h, [n] * B, layered=layered)
    torch.cuda.synchronize(device)
    t_all = time.time() - t0
    gen = max(t_all - t_ctx, 1e-9)
    return {"batch": B, "n_samples": n, "context_steps": eng.receptive_field, "context_s": t_ctx, "total_s": t_all,
            "us_per_step": gen / (n - 1) * 1e6, "samples_per_sec_per_utt": (n - 1) / gen,
            "samples_per_sec": B * (n - 1) / gen, "samples_per_sec_incl_context": B * n / t_all}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 8, 64, 256])
    ap.add_argument("--samples", type=int, default=4000)
    args = ap.parse_args()
    device = torch.device("cuda:0")
    torch.manual_seed(1)
    model = WaveNet(256, 80, 64, 256, 10, 3, 2, 80)
    model.apply(initialize)
    model.to(device)
    for B in args.batch:
        print(json.dumps(measure(model, B, args.samples, device)), flush=True)


if __name__ == "__main__":
    main()
