#!/usr/bin/env python
"""BASELINE configs[3] geometry with the softmax head the reference actually has (LJSpeech recipe shape:
80-dim mel aux, kernel_size 3, upsampling_factor 256, rf 6139; batch_length 20000 -> 19973, T = 26112):
parity vs the oracle on a short window and step time.  (The mixture-of-logistics head of configs[3] does
not exist in the reference; the second part times the same geometry with this repo's MoL head, 10 components.)"""
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
cfg_t = (256, 80, 64, 256, 10, 3, 3, 256)
cfg = O.OracleConfig(*cfg_t)
assert cfg.receptive_field == 6139
params = O.random_params(cfg, 5, scale=0.05)
model = WaveNet(*cfg_t)
model.load_state_dict(params)
model.to(dev)
T = 6400
x, h, t = O.synthetic_batch(cfg, 1, T, 6)
ref = O.forward(cfg, params, x, h)
out = model(x.to(dev), h.to(dev))
print("configs[3] geometry (softmax head) logits max-abs err vs oracle: %.3e (T=%d)" % (float((out.detach().cpu() - ref).abs().max()), T))
B, T = 8, 26112
x, h, t = O.synthetic_batch(cfg, B, T, 7)
x, h, t = x.to(dev), h.to(dev), t.to(dev)
opt = FusedAdam(model, lr=1e-4)
for _ in range(3):
    model.loss_and_backward(x, h, t)
    opt.step()
torch.cuda.synchronize()
t0 = time.time()
n = 10
for _ in range(n):
    model.loss_and_backward(x, h, t)
    opt.step()
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(json.dumps({"config": "K=3 U=256 rf=6139, B=8, T=26112 (19973 loss positions)", "ms_per_step": dt * 1e3,
                  "samples_per_sec": B * (T - 6139) / dt}))

# ---- the same geometry with the mixture-of-logistics head (10 components, 16-bit classes) ----
import numpy as np  # noqa: E402
del model, opt
torch.manual_seed(2)
mol = WaveNet(*cfg_t, n_mixture=10)
from pytorchwavenetvocoder_amd.nets import initialize  # noqa: E402
mol.apply(initialize)
mol.to(dev)
y = torch.from_numpy(np.random.RandomState(3).uniform(-1, 1, (B, T)).astype(np.float32)).to(dev)
opt = FusedAdam(mol, lr=1e-4)
for _ in range(3):
    loss = mol.mol_loss_and_backward(x, h, y)
    opt.step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(n):
    loss = mol.mol_loss_and_backward(x, h, y)
    opt.step()
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(json.dumps({"config": "configs[3]: K=3 U=256 rf=6139, B=8, T=26112, mixture-of-logistics head (10 components)",
                  "ms_per_step": dt * 1e3, "samples_per_sec": B * (T - 6139) / dt, "loss": float(loss)}))
