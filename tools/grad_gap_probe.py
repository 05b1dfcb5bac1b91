#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Why is the worst gradient tensor 4x further from the fp32 oracle in chain mode than in launch-pair mode?
("What's weak": dil_tanh.3.conv.bias 2.6e-5 vs 6.3e-6 at full size.)

Runs the config-2 model on one mid-size batch against BOTH an fp32 and an fp64 evaluation of the oracle (same ReLU
sub-gradient choice: the HIP path's masks) and prints, per launch mode, the five worst tensors against each -- the fp64 column
says which evaluation is actually closer to the exact gradient, the fp32 column is what the parity tests measure.

    python tools/grad_gap_probe.py [B T]      (GPU)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wavenet_oracle as O  # noqa: E402
from pytorchwavenetvocoder_amd import _lib as L  # noqa: E402
from pytorchwavenetvocoder_amd.engine import DEFAULT_FLAGS, WaveNetEngine, flat_to_state, load_state_into_flat  # noqa: E402

DEV = "cuda:0"


def main():
    # grad_gap_probe.py [B T [kernel_size upsampling [seed]]]   (configs[3] geometry: 2 6656 3 256 6 = the reduced-size test)
    a = [int(v) for v in sys.argv[1:]]
    B, T = (a[0], a[1]) if len(a) > 1 else (2, 9600)
    K, U = (a[2], a[3]) if len(a) > 3 else (2, 80)
    seed = a[4] if len(a) > 4 else 101
    cfg_t = (256, 80, 64, 256, 10, 3, K, U)
    cfg = O.OracleConfig(*cfg_t)
    params = O.random_params(cfg, seed, scale=0.05)
    x, h, t = O.synthetic_batch(cfg, B, T, seed + 1)
    print("model K=%d U=%d rf=%d, B=%d T=%d (%d loss positions per sequence), seed %d" % (K, U, cfg.receptive_field, B, T,
                                                                                          T - cfg.receptive_field, seed))
    eng = WaveNetEngine(*cfg_t, device=DEV, library=L.load_library())
    load_state_into_flat(eng, params)
    eng.flags = DEFAULT_FLAGS
    loss, dl = eng.forward_loss(x.to(DEV), h.to(DEV), t.to(DEV))
    m_skip = (eng.saved(L.WS_RELU_SKIP) > 0).float().cpu()
    m_post = (eng.saved(L.WS_RELU_POST1) > 0).float().cpu()
    torch.set_num_threads(32)
    _, _, g32 = O.train_step(cfg, params, None, x, h, t, relu_masks=(m_skip, m_post))
    p64 = {k: v.double() for k, v in params.items()}
    _, _, g64 = O.train_step(cfg, p64, None, x, h.double(), t, relu_masks=(m_skip.double(), m_post.double()))

    def rel(a, b):
        return float((a.double() - b.double()).abs().max()) / max(float(b.abs().max()), 1e-300)

    print("fp32 oracle vs fp64 oracle: worst %s" % sorted(((rel(g32[k], g64[k]), k) for k in g64 if g64[k] is not None), reverse=True)[:3])
    modes = [("chain (default)", DEFAULT_FLAGS), ("launch pair", DEFAULT_FLAGS | L.FLAG_NO_CHAIN)]
    for name, fl in modes:
        eng.flags = fl
        g = flat_to_state(eng, eng.backward(dl, t_first=eng.receptive_field).cpu(), O.param_shapes(cfg))
        e32 = sorted(((rel(g[k], g32[k]), k) for k in g64 if g64[k] is not None), reverse=True)
        e64 = sorted(((rel(g[k], g64[k]), k) for k in g64 if g64[k] is not None), reverse=True)
        print("%-22s vs fp32 oracle: %s" % (name, ["%.2e %s" % v for v in e32[:5]]))
        print("%-22s vs fp64 oracle: %s" % (name, ["%.2e %s" % v for v in e64[:5]]))
        by_kind = {}
        for v, k in e64:
            kind = ".".join(p for p in k.split(".") if not p.isdigit())
            by_kind[kind] = max(by_kind.get(kind, 0.0), v)
        print("   worst per tensor kind (fp64): " + ", ".join("%s %.1e" % (k, v) for k, v in sorted(by_kind.items(), key=lambda kv: -kv[1])[:8]))


if __name__ == "__main__":
    main()
