#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Co-scheduling A/B: the persistent chain kernels on WN_CHAIN_BLOCKS CUs, the matrix-bound
contractions (post-net / skip weight gradients, chunked skip-sum) on a CU-masked side stream (WN_SIDE_CUS).  Both
knobs are read once per process, so run one process per setting (tools/cosched_probe.sh).  Steps run on a non-NULL
torch stream: a CU-masked stream is a blocking stream and would serialise against the NULL stream."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


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
    bl, frames, T = bench.geometry(rf, bench.BATCH_LENGTH, 80)
    B = bench.BATCH_PER_GPU
    gen = torch.Generator().manual_seed(1234)
    xx = torch.randint(0, 256, (B, T + 1), generator=gen)
    x, t = xx[:, :-1].contiguous().to(dev), xx[:, 1:].contiguous().to(dev)
    h = torch.randn(B, 80, frames, generator=gen).to(dev)
    opt = FusedAdam(model, lr=1e-4)
    eng = model.engine
    BW, HD, FW = _lib.FLAG_BWD_OVERLAP, _lib.FLAG_BWD_OVERLAP_HEAD, _lib.FLAG_FWD_OVERLAP
    modes = [("serial", 0), ("head", BW | HD), ("head+fwd", BW | HD | FW), ("fwd", FW)]
    tag = "blocks=%s side_cus=%s" % (os.environ.get("WN_CHAIN_BLOCKS", "256"), os.environ.get("WN_SIDE_CUS", "-"))
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        grads = {}
        for name, fl in modes[:2]:
            eng.flags = fl
            model.loss_and_backward(x, h, t)
            torch.cuda.synchronize()
            grads[name] = eng.grads().clone()
        same = bool(torch.equal(grads["serial"], grads["head"]))

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

        res = {name: [] for name, _ in modes}
        for rnd in range(3):
            for name, fl in modes:
                eng.flags = fl
                timed(step, 3)
                res[name].append(timed(step, 12))
    print("%-28s bitwise %s | " % (tag, same) + " | ".join("%s %.3f" % (n, min(v)) for n, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
