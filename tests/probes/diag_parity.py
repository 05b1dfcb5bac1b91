#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Diagnostic: per-tensor gradient errors of the HIP path vs the fp32 oracle and an fp64 oracle run
(the fp64 run gives the fp32 noise floor of the reference arithmetic itself).

    python tools/diag_parity.py Q,A,R,S,dd,dr,K,U B T seed scale [flags]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import wavenet_oracle as O  # noqa: E402  (checker only)
from pytorchwavenetvocoder_amd import _lib  # noqa: E402
from pytorchwavenetvocoder_amd.engine import WaveNetEngine, flat_to_state, load_state_into_flat  # noqa: E402
from tests.golden_util import rel_to_max  # noqa: E402


def main():
    cfg_t = tuple(int(v) for v in sys.argv[1].split(","))
    B, T, seed, scale = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
    flags = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    dev = "cuda:0"
    cfg = O.OracleConfig(*cfg_t)
    params = O.random_params(cfg, seed, scale=scale)
    x, h, t = O.synthetic_batch(cfg, B, T, seed + 1)
    t0 = time.time()
    loss_ref, logits_ref, grads_ref = O.train_step(cfg, params, None, x, h, t)
    p64 = {k: v.double() for k, v in params.items()}
    loss64, logits64, grads64 = O.train_step(cfg, p64, None, x, h.double(), t)
    print("oracle f32+f64 %.1fs  threads=%d" % (time.time() - t0, torch.get_num_threads()))
    eng = WaveNetEngine(*cfg_t, device=dev, library=_lib.load_library())
    eng.flags = flags
    load_state_into_flat(eng, params)
    logits = eng.forward(x.to(dev), h.to(dev))
    lg = logits.transpose(1, 2).cpu()
    print("logits: mine-vs-f32 %.3g | f32-vs-f64 %.3g | mine-vs-f64 %.3g" % (
        float((lg - logits_ref).abs().max()), float((logits_ref.double() - logits64).abs().max()),
        float((lg.double() - logits64).abs().max())))
    loss, dl = eng.loss(logits, t.to(dev))
    print("loss: mine %.7f f32 %.7f f64 %.7f" % (float(loss.cpu()), float(loss_ref), float(loss64)))
    gd = flat_to_state(eng, eng.backward(dl).cpu(), O.param_shapes(cfg))
    rows = []
    for k, ref in grads_ref.items():
        if ref is None:
            continue
        rows.append((rel_to_max(gd[k], ref), rel_to_max(ref.double(), grads64[k]),
                     rel_to_max(gd[k].double(), grads64[k]), float(grads64[k].abs().max()), k))
    rows.sort(reverse=True)
    for r in rows[:12]:
        print("  mine-vs-f32 %.3g | f32-vs-f64 %.3g | mine-vs-f64 %.3g | max|g| %.3g  %s" % r)
    print("  ... best:", "%.3g %s" % (rows[-1][0], rows[-1][4]))


if __name__ == "__main__":
    main()
