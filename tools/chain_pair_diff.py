#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Where do the backward chain (k_chain64s) and the gate' + dX launch pair start to differ, and which of them is the
biased one?  Runs one backward pass of the config-2 model in three modes on the same batch --
  E: exact f32 MFMA arithmetic (WN_FLAG_EXACT_MFMA: k-ordered fp32 fma chains; launch pair structure) = the reference,
  P: split arithmetic, launch pair (WN_FLAG_NO_CHAIN),
  C: split arithmetic, chain (default)
and prints per layer, for dP (sigmoid rows / tanh rows) and dX: max |X - E| / max |E| and the MEAN signed difference
relative to the mean magnitude (a non-zero mean = a systematic bias, which is what a bias gradient -- a sum over 184 320
positions -- amplifies).      python tools/chain_pair_diff.py [B T]      (GPU)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wavenet_oracle as O  # noqa: E402
from pytorchwavenetvocoder_amd import _lib as L  # noqa: E402
from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat  # noqa: E402

DEV = "cuda:0"


def main():
    B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 9600)
    cfg_t = (256, 80, 64, 256, 10, 3, 2, 80)
    cfg = O.OracleConfig(*cfg_t)
    params = O.random_params(cfg, 101, scale=0.05)
    x, h, t = O.synthetic_batch(cfg, B, T, 102)
    xd, hd, td = x.to(DEV), h.to(DEV), t.to(DEV)
    A = L.FLAG_AUX_FUSED
    modes = [("E", L.FLAG_EXACT_MFMA), ("P", A | L.FLAG_NO_CHAIN), ("C", A)]
    got = {}
    eng = WaveNetEngine(*cfg_t, device=DEV, library=L.load_library())
    load_state_into_flat(eng, params)
    eng.flags = A
    logits = eng.forward(xd, hd)          # ONE forward (split arithmetic): all three backward passes read the same saved tensors
    loss, dl = eng.loss(logits, td)
    for name, fl in modes:
        eng.flags = fl
        g = eng.backward(dl, t_first=eng.receptive_field).clone()
        got[name] = (eng.saved(L.WS_DP).clone(), eng.saved(L.WS_DX).clone(), g)
    Ls = len(cfg.dilations)

    def stat(a, e):
        d = (a - e).double()
        return float(d.abs().max() / e.abs().max()), float(d.mean() / e.abs().double().mean())

    print("layer |  dP sigmoid rows: P max, P bias, C max, C bias |  dP tanh rows: P max, P bias, C max, C bias |  dX: P max, P bias, C max, C bias")
    for l in range(Ls - 1, -1, -1):
        row = []
        for sel in ("sig", "tanh", "dx"):
            for m in ("P", "C"):
                if sel == "dx":
                    a, e = got[m][1][l], got["E"][1][l]
                else:
                    sl = slice(0, 64) if sel == "sig" else slice(64, 128)
                    a, e = got[m][0][l][:, sl], got["E"][0][l][:, sl]
                row += list(stat(a, e))
        print("%5d | %s" % (l, "  ".join("%9.2e" % v for v in row)))
    ge = got["E"][2]
    for m in ("P", "C"):
        print("%s flat gradient vs E: max %.3e of max" % (m, float((got[m][2] - ge).abs().max() / ge.abs().max())))


if __name__ == "__main__":
    main()
