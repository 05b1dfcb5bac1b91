#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Diagnostic: run forward/backward repeatedly (fused and layered kernels) on the GPU and report
which gradient elements differ between repetitions / between the two kernel paths."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import wavenet_oracle as O  # noqa: E402
from pytorchwavenetvocoder_amd import _lib  # noqa: E402
from pytorchwavenetvocoder_amd.engine import WaveNetEngine, key_to_kind, load_state_into_flat, state_keys  # noqa: E402


def main():
    cfg_t = tuple(int(v) for v in sys.argv[1].split(","))
    B, T, seed, scale, reps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])
    dev = "cuda:0"
    cfg = O.OracleConfig(*cfg_t)
    params = O.random_params(cfg, seed, scale=scale)
    x, h, t = O.synthetic_batch(cfg, B, T, seed + 1)
    x, h, t = x.to(dev), h.to(dev), t.to(dev)
    res = {}
    for flags in (1, 0):
        eng = WaveNetEngine(*cfg_t, device=dev, library=_lib.load_library())
        eng.flags = flags
        load_state_into_flat(eng, params)
        outs = []
        for r in range(reps):
            logits = eng.forward(x, h)
            loss, dl = eng.loss(logits, t)
            g = eng.backward(dl).clone()
            torch.cuda.synchronize()
            outs.append((logits.clone(), g))
        res[flags] = (eng, outs)
    eng, ref_outs = res[1]
    gref = ref_outs[0][1]
    spans = []
    for k in state_keys(eng.cfg):
        off, n = eng.param_slice(*key_to_kind(k))
        spans.append((off, off + n, k))

    def describe(name, g):
        d = (g - gref).abs()
        bad = (d > 1e-5 * gref.abs().max()).nonzero().flatten()
        print("%s: max|diff| %.3g, n_bad %d of %d" % (name, float(d.max()), bad.numel(), g.numel()))
        if bad.numel():
            per = {}
            for i in bad.tolist():
                for lo, hi, k in spans:
                    if lo <= i < hi:
                        per.setdefault(k, []).append(i - lo)
                        break
            for k, idxs in list(per.items())[:10]:
                print("    %-28s n=%d first idx %s" % (k, len(idxs), idxs[:12]))

    for flags in (1, 0):
        for r, (lg, g) in enumerate(res[flags][1]):
            describe("flags=%d rep %d (logits diff vs layered rep0 %.3g)" % (
                flags, r, float((lg - ref_outs[0][0]).abs().max())), g)


if __name__ == "__main__":
    main()
