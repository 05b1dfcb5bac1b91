#!/usr/bin/env python
"""Decode speed of the recipe-size model (n_resch = 512, n_skipch = 256: egs/arctic/sd/run.sh:46-52; the one-workgroup kernel
does not cover it) through the any-size path: the persistent launch of csrc/wn_dlp.hip (B <= 64) and, beside it, the
layer-wise launches it replaces (``layered="launches"``); tokens of the two compared.

    python tools/recipe_decode_probe.py [--kernel-size 2] [--steps 400]            (GPU)
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


def measure(model, B, n, dev, layered):
    x = torch.full((B, 1), 128, dtype=torch.int64, device=dev)
    h = torch.randn(B, 80, (n + 80) // 80 + 1, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    model.engine.decode(x, h, [2] * B, layered=layered)          # warm-up (allocations, first launch)
    torch.cuda.synchronize()
    t0 = time.time(); model.engine.decode(x, h, [2] * B, layered=layered); torch.cuda.synchronize(); t_ctx = time.time() - t0
    t0 = time.time(); toks = model.engine.decode(x, h, [n] * B, layered=layered); torch.cuda.synchronize(); t_all = time.time() - t0
    gen = max(t_all - t_ctx, 1e-9)
    return {"us_per_step": gen / (n - 2) * 1e6, "samples_per_sec": B * (n - 2) / gen, "context_s": t_ctx}, toks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel-size", type=int, default=2)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--batches", default="1,4,16,32,64,256")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    m = WaveNet(256, 80, 512, 256, 10, 3, a.kernel_size, 80)
    m.apply(initialize)
    m.to(dev)
    assert not m.engine.decode_supported()
    for B in [int(v) for v in a.batches.split(",")]:
        row = {"model": "512/256 recipe size, K=%d" % a.kernel_size, "batch": B}
        modes = (("persistent", True), ("launches", "launches")) if B <= 64 else (("launches", "launches"),)
        toks = {}
        for name, lay in modes:
            row[name], toks[name] = measure(m, B, a.steps, dev, lay)
        if len(toks) == 2:
            row["tokens_equal"] = all(bool((p == q).all()) for p, q in zip(toks["persistent"], toks["launches"]))
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
