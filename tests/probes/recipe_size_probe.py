#!/usr/bin/env python
"""the recipe-size model (n_resch=512, n_skipch=256, egs/*/run.sh defaults) through the
any-size GEMM kernels: parity vs the oracle on a short window and step time at batch_length 20000."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import wavenet_oracle as O  # noqa: E402  (checker only)
from pytorchwavenetvocoder_amd.nets import WaveNet  # noqa: E402
from pytorchwavenetvocoder_amd.optim import FusedAdam  # noqa: E402

dev = torch.device("cuda:0")
cfg_t = (256, 80, 512, 256, 10, 3, 2, 80)
cfg = O.OracleConfig(*cfg_t)
params = O.random_params(cfg, 5, scale=0.02)
model = WaveNet(*cfg_t)
model.load_state_dict(params)
model.to(dev)
# parity on a short window (the oracle needs seconds at this size)
T = 3200
x, h, t = O.synthetic_batch(cfg, 1, T, 6)
ref = O.forward(cfg, params, x, h)
out = model(x.to(dev), h.to(dev))
print("recipe-size logits max-abs err vs oracle: %.3e (T=%d)" % (float((out.detach().cpu() - ref).abs().max()), T))
# step time
for B in (2, 4):
    T = 23040
    x, h, t = O.synthetic_batch(cfg, B, T, 7)
    x, h, t = x.to(dev), h.to(dev), t.to(dev)
    opt = FusedAdam(model, lr=1e-4)
    for _ in range(2):
        model.loss_and_backward(x, h, t)
        opt.step()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 5
    for _ in range(n):
        model.loss_and_backward(x, h, t)
        opt.step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    flop = 3 * 2 * B * T * (30 * (2 * 512 * 512 * 2 + 512 * 256 + 512 * 512) + 256 * 256 * 2)
    print(json.dumps({"model": "512/256 recipe size", "B": B, "T": T, "ms_per_step": dt * 1e3,
                      "samples_per_sec": B * (T - 3070) / dt, "approx_tflops": flop / dt / 1e12}))
