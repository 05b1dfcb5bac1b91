#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""A/B of the stream-overlap launch modes on the BASELINE configs[1] training step (one process, one model):

    serial        0                                          everything on the caller's stream, one weight-gradient group per bucket
    serial/5      WN_FLAG_DW_FLUSH(5)                        same, groups of 5 layers
    bwd/n         WN_FLAG_BWD_OVERLAP | WN_FLAG_DW_FLUSH(n)  weight gradients on the side stream beside the gate'/dX chain
    bwd/n+fwd     ... | WN_FLAG_FWD_OVERLAP                  + skip-sum in three chunks beside the residual stack
    fwd           WN_FLAG_FWD_OVERLAP                        only the forward mode

Prints ms/step and forward ms per mode (three interleaved rounds), writes gpurun_out/overlap_probe_<prio>.json and
the flags of the fastest mode.  WN_SIDE_PRIORITY=normal|low (default low) is the priority of the library's side
stream; run once per value.  Checks that gradients of serial/5 and bwd/5 are bit-identical."""
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
    F = _lib.flag_dw_flush
    BW, FW = _lib.FLAG_BWD_OVERLAP, _lib.FLAG_FWD_OVERLAP
    modes = [("serial/5", F(5)), ("bwd/5", BW | F(5)), ("serial", 0), ("bwd/3", BW | F(3)), ("bwd/10", BW | F(10)),
             ("bwd/30", BW | F(30)), ("bwd/5+fwd", BW | FW | F(5)), ("bwd/10+fwd", BW | FW | F(10)), ("fwd", FW)]
    prio = os.environ.get("WN_SIDE_PRIORITY", "low")
    eng = model.engine

    # bit-identity of the gradients (no optimizer step in between)
    grads = {}
    for name, fl in modes[:2]:
        eng.flags = fl
        model.loss_and_backward(x, h, t)
        torch.cuda.synchronize()
        grads[name] = eng.grads().clone()
    same = bool(torch.equal(grads["serial/5"], grads["bwd/5"]))
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
        print("%-10s flags %4d  step ms %s  forward ms %s" % (
            name, fl, " ".join("%.3f" % v for v in r["step_ms"]), " ".join("%.3f" % v for v in r["fwd_ms"])), flush=True)
        if best is None or r["best_step_ms"] < res[best]["best_step_ms"]:
            best = name
    print("side-stream priority %s: fastest %s (flags %d, %.3f ms/step)" % (prio, best, res[best]["flags"],
                                                                             res[best]["best_step_ms"]), flush=True)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "overlap_probe_%s.json" % prio), "w") as fh:
        json.dump({"priority": prio, "bitwise_equal": same, "modes": res, "fastest": best,
                   "fastest_flags": res[best]["flags"], "fastest_ms": res[best]["best_step_ms"]}, fh)
    if not same:
        sys.exit(1)


if __name__ == "__main__":
    main()
