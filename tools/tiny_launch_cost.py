#!/usr/bin/env python
"""Fixed cost of one launch of the fused kernels: training steps on a minibatch of a few tiles (B = 1, T = 3200), to be run
under rocprofv3 --kernel-trace --stats.   The per-launch duration at (almost) no work is prologue + dispatch ramp + drain."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytorchwavenetvocoder_amd.nets import WaveNet, initialize  # noqa: E402

dev = "cuda:0"
torch.manual_seed(1)
m = WaveNet(256, 80, 64, 256, 10, 3, 2, 80); m.apply(initialize); m.to(dev)
B, T = int(os.environ.get("TINY_B", 1)), int(os.environ.get("TINY_T", 3200))
x = torch.randint(0, 256, (B, T), device=dev); h = torch.randn(B, 80, T // 80, device=dev)
t = torch.randint(0, 256, (B, T), device=dev)
for _ in range(10):
    m.loss_and_backward(x, h, t)
torch.cuda.synchronize()
