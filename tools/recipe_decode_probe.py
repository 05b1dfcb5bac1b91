#!/usr/bin/env python
"""Decode speed of the recipe-size model (n_resch = 512, n_skipch = 256: egs/arctic/sd/run.sh:46-52; the one-workgroup kernel
does not cover it) through the any-size path: the persistent launch of csrc/wn_dlp.hip / wn_dlpf.hip (in groups of 48 utterances beyond that) and, beside it, the
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
    """us per generated step = (time of n steps - time of n / 4 steps) / (3 n / 4): the context pass, the weight packing and the
    persistent kernels' set-up (private queue copies: 200 MB per workgroup at 16 utterances) are in both runs."""
    x = torch.full((B, 1), 128, dtype=torch.int64, device=dev)
    h = torch.randn(B, 80, (n + 80) // 80 + 1, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    n1 = max(2, n // 4)

    def run(k):
        torch.cuda.synchronize()
        t0 = time.time()
        toks = model.engine.decode(x, h, [k] * B, layered=layered)
        torch.cuda.synchronize()
        return time.time() - t0, toks
    run(n1)                      # warm-up (allocations, first launch)
    t1 = min(run(n1)[0] for _ in range(2))
    t2, toks = run(n)
    t2 = min(t2, run(n)[0])
    per = max(t2 - t1, 1e-9) / (n - n1)
    return {"us_per_step": per * 1e6, "samples_per_sec": B / per, "fixed_s": t1 - per * n1}, toks


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
        modes = (("persistent", True), ("launches", "launches"))   # (more than 48 utterances: persistent launches in groups of 48)
        toks = {}
        for name, lay in modes:
            row[name], toks[name] = measure(m, B, a.steps, dev, lay)
        if len(toks) == 2:
            row["tokens_equal"] = all(bool((p == q).all()) for p, q in zip(toks["persistent"], toks["launches"]))
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
