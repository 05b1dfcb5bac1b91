#!/usr/bin/env python
"""How much of the training step is launch gaps?  Times forward + loss + backward of the config-2 model launched eagerly
(110 kernel launches through the C ABI) and replayed from one captured HIP graph.   gpurun -- python tools/graph_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytorchwavenetvocoder_amd.nets import WaveNet, initialize  # noqa: E402

dev = "cuda:0"
torch.manual_seed(1)
m = WaveNet(256, 80, 64, 256, 10, 3, 2, 80); m.apply(initialize); m.to(dev)
B, T = 8, 20000
x = torch.randint(0, 256, (B, T), device=dev); h = torch.randn(B, 80, T // 80, device=dev)
t = torch.randint(0, 256, (B, T), device=dev)


def step():
    return m.loss_and_backward(x, h, t)


for _ in range(5):
    step()
torch.cuda.synchronize()


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = sorted(timed(step) for _ in range(5))[2]
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
graphed = sorted(timed(g.replay) for _ in range(5))[2]
ref = float(step())
g.replay()
print("forward+loss+backward: eager %.3f ms, one HIP graph %.3f ms (loss eager %.6f, graph %.6f)" % (eager, graphed, ref, float(loss)))
