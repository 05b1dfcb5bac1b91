#!/usr/bin/env python
"""Does capturing the whole training step in a HIP graph shrink the inter-kernel gaps?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorchwavenetvocoder_amd.nets import WaveNet, initialize
from pytorchwavenetvocoder_amd.optim import FusedAdam

dev = torch.device("cuda:0")
torch.manual_seed(1)
m = WaveNet(256, 80, 64, 256, 10, 3, 2, 80); m.apply(initialize); m.to(dev)
B, T = 8, 23040
x = torch.randint(0, 256, (B, T), device=dev); t = torch.randint(0, 256, (B, T), device=dev)
h = torch.randn(B, 80, T // 80, device=dev)
opt = FusedAdam(m, lr=1e-4)

def step():
    m.loss_and_backward(x, h, t)
    opt.step()

for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10): step()
torch.cuda.synchronize()
print("eager  ms/step %.3f" % ((time.time() - t0) / 10 * 1e3))
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print("graph  ms/step %.3f" % ((time.time() - t0) / 10 * 1e3))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
