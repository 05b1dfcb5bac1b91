#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Signed per-layer drift of the any-size (n_resch % 128 == 0) backward pass: the chain_pair_diff.py method on the WIDE path.

VERDICT r03: the recipe-size model's worst gradient (dil_tanh.0.conv.bias) is 6.3e-5 of its maximum at T = 23040 and
5.9e-6 at T = 3200 -- the signature of a systematic (signed) per-layer bias of dX that a bias-type gradient, a sum over every
position, amplifies.  This runs ONE forward (split arithmetic) and then the backward pass in
  E: exact f32-MFMA arithmetic (WN_FLAG_EXACT_MFMA: k-ordered fp32 fma chains) = the reference on the same saved tensors,
  S: the split contractions of the library under test (default flags)
and prints per layer, for dP (sigmoid rows / tanh rows) and dX: max |S - E| / max |E| and the MEAN signed difference
relative to the mean magnitude (non-zero mean = bias), then the worst flat-gradient tensors of S against E.

    python tools/wide_drift_probe.py [--resch 512] [--B 2] [--T 23040] [--layers ...]        (GPU)
Run it with WN_LIB_PATH=<variant .so> to measure a variant build (tools/build_variant.sh).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wavenet_oracle as O  # noqa: E402
from pytorchwavenetvocoder_amd import _lib as L  # noqa: E402
from pytorchwavenetvocoder_amd.engine import DEFAULT_FLAGS, WaveNetEngine, flat_to_state, load_state_into_flat  # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--resch", type=int, default=512)
    ap.add_argument("--skipch", type=int, default=256)
    ap.add_argument("--kernel-size", type=int, default=2)
    ap.add_argument("--upsampling", type=int, default=80)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--T", type=int, default=23040)
    ap.add_argument("--scale", type=float, default=0.02)
    ap.add_argument("--seed", type=int, default=112)
    a = ap.parse_args()
    cfg_t = (256, 80, a.resch, a.skipch, 10, 3, a.kernel_size, a.upsampling)
    cfg = O.OracleConfig(*cfg_t)
    params = O.random_params(cfg, a.seed, scale=a.scale)
    x, h, t = O.synthetic_batch(cfg, a.B, a.T, a.seed + 1)
    xd, hd, td = x.to(DEV), h.to(DEV), t.to(DEV)
    eng = WaveNetEngine(*cfg_t, device=DEV, library=L.load_library())
    load_state_into_flat(eng, params)
    eng.flags = DEFAULT_FLAGS
    logits = eng.forward(xd, hd)   # ONE forward: both backward passes read the same saved tensors
    loss, dl = eng.loss(logits, td)
    del logits
    got = {}
    for name, fl in (("E", L.FLAG_EXACT_MFMA), ("S", DEFAULT_FLAGS)):
        eng.flags = fl
        eng._fwd_flags = fl   # (the any-size path saves the same tensors in both arithmetic modes)
        g = eng.backward(dl, t_first=eng.receptive_field).clone()
        # keep E's big tensors on the host (the workspace is reused by the next pass)
        dp, dx = eng.saved(L.WS_DP), eng.saved(L.WS_DX)
        got[name] = (dp.cpu() if name == "E" else dp, dx.cpu() if name == "E" else dx, g)
    nl = len(cfg.dilations)
    R = a.resch

    def stat(s, e):
        e = e.to(s.device)
        d = (s - e).double()
        return float(d.abs().max() / e.abs().max()), float(d.mean() / e.abs().double().mean())

    print("lib: %s   model R=%d S=%d K=%d U=%d   B=%d T=%d" % (os.environ.get("WN_LIB_PATH", "(in-tree)"), R, a.skipch,
                                                              a.kernel_size, a.upsampling, a.B, a.T))
    print("layer |  dP sigmoid rows: max, bias |  dP tanh rows: max, bias |  dX: max, bias")
    for l in range(nl - 1, -1, -1):
        row = []
        for sel in ("sig", "tanh", "dx"):
            if sel == "dx":
                s, e = got["S"][1][l], got["E"][1][l]
            else:
                sl = slice(0, R) if sel == "sig" else slice(R, 2 * R)
                s, e = got["S"][0][l][:, sl], got["E"][0][l][:, sl]
            row += list(stat(s, e))
        print("%5d | %s" % (l, "  ".join("%9.2e" % v for v in row)))
    ge, gs = got["E"][2].cpu(), got["S"][2].cpu()
    print("flat gradient S vs E: max %.3e of max" % float((gs - ge).abs().max() / ge.abs().max()))
    shapes = O.param_shapes(cfg)
    se, ss = flat_to_state(eng, ge, shapes), flat_to_state(eng, gs, shapes)
    worst = sorted(((float((ss[k] - se[k]).abs().max() / max(float(se[k].abs().max()), 1e-30)), k) for k in se), reverse=True)
    print("worst tensors (S vs E, relative to the tensor's max): " + ", ".join("%s %.2e" % (k, v) for v, k in worst[:8]))
    by_kind = {}
    for v, k in worst:
        kind = ".".join(p for p in k.split(".") if not p.isdigit())
        by_kind[kind] = max(by_kind.get(kind, 0.0), v)
    print("worst per tensor kind: " + ", ".join("%s %.1e" % (k, v) for k, v in sorted(by_kind.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
