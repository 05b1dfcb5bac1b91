#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""A/B of the stream-overlap launch modes on the BASELINE configs[1] training step (one process, one model):

    serial        WN_FLAG_NO_OVERLAP           everything on the caller's stream
    bwd           0                            weight gradients on the side stream beside the gate'/dX chain
    bwd+fwd       WN_FLAG_FWD_OVERLAP          + skip-sum in three chunks beside the residual stack
    fwd only      measured as forward-only time of the same modes

Prints ms/step per mode (two interleaved rounds) and writes the flags of the fastest mode to
gpurun_out/best_flags.txt.  Checks that the gradients of `serial` and `bwd` are bit-identical."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (configuration and geometry of the benchmark)


def main():
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.nets import WaveNet, initialize
    from pytorchwavenetvocoder_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = WaveNet(**bench.CFG2)
    model.apply(initialize)
    model.to(dev)
    rf = model.receptive_field
    bl, frames, T = bench.geometry(rf, bench.BATCH_LENGTH, bench.CFG2["upsampling_factor"])
    B = bench.BATCH_PER_GPU
    gen = torch.Generator().manual_seed(1234)
    xx = torch.randint(0, 256, (B, T + 1), generator=gen)
    x, t = xx[:, :-1].contiguous().to(dev), xx[:, 1:].contiguous().to(dev)
    h = torch.randn(B, 80, frames, generator=gen).to(dev)
    opt = FusedAdam(model, lr=1e-4)
    modes = [("serial", _lib.FLAG_NO_OVERLAP), ("bwd", 0), ("bwd+fwd", _lib.FLAG_FWD_OVERLAP)]
    eng = model.engine

    # bit-identity of the gradients (no optimizer step in between)
    grads = {}
    for name, fl in modes[:2]:
        eng.flags = fl
        model.loss_and_backward(x, h, t)
        torch.cuda.synchronize()
        grads[name] = eng.grads().clone()
    same = bool(torch.equal(grads["serial"], grads["bwd"]))
    print("gradients serial == side-stream, bitwise:", same, flush=True)

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def step():
        model.loss_and_backward(x, h, t)
        opt.step()

    def fwd():
        eng.forward(x, h)

    res = {name: {"step_ms": [], "fwd_ms": []} for name, _ in modes}
    for rnd in range(3):
        for name, fl in modes:
            eng.flags = fl
            timed(step, 3)
            res[name]["step_ms"].append(timed(step, 15))
            timed(fwd, 2)
            res[name]["fwd_ms"].append(timed(fwd, 10))
    best = None
    for name, fl in modes:
        r = res[name]
        r["flags"] = fl
        r["best_step_ms"] = min(r["step_ms"])
        r["best_fwd_ms"] = min(r["fwd_ms"])
        print("%-8s flags %d  step ms %s  forward ms %s" % (
            name, fl, " ".join("%.3f" % v for v in r["step_ms"]), " ".join("%.3f" % v for v in r["fwd_ms"])), flush=True)
        if best is None or r["best_step_ms"] < res[best]["best_step_ms"]:
            best = name
    print("fastest:", best, "flags", res[best]["flags"], flush=True)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "best_flags.txt"), "w") as fh:
        fh.write("%d\n" % res[best]["flags"])
    print(json.dumps({"bitwise_equal": same, "modes": res, "fastest": best}), flush=True)
    if not same:
        sys.exit(1)


if __name__ == "__main__":
    main()
