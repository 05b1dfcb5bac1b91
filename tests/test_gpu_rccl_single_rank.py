# -*- coding: utf-8 -*-
"""RCCL on the hardware one box has: a process group of ONE rank with the ``nccl`` backend (= RCCL on ROCm).  The
``GradientReducer`` is told to exchange even when alone, so the step runs exactly the N > 1 code path -- bucket events
recorded by wn_backward, ``dist.all_reduce`` of every contiguous gradient range on the side stream (RCCL kernels on this
GPU, beside the backward kernels), join before Adam -- and a one-rank all-reduce is the identity: the gradients must be
BIT-identical to the plain step.  (The reference's counterpart is nn.DataParallel, train.py:449-454.)  What this cannot
show is a transfer over xGMI: that needs the multi-GPU node only the driver has."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CODE = r"""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = "%(port)d"
from pytorchwavenetvocoder_amd.distributed import GradientReducer, rccl_footprint_defaults
from pytorchwavenetvocoder_amd.nets import WaveNet, initialize
from oracle import wavenet_oracle as O
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
rccl_footprint_defaults()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg_t = (256, 80, 64, 256, 10, 3, 2, 80)
cfg = O.OracleConfig(*cfg_t)
torch.manual_seed(3)
model = WaveNet(*cfg_t)
model.apply(initialize)
model.to(dev)
x, h, t = O.synthetic_batch(cfg, 2, 3200 + 80 * 3, 9)
x, h, t = x.to(dev), h.to(dev), t.to(dev)
plain = GradientReducer(model)
l0 = plain.loss_and_backward(x, h, t)
g0 = model.engine.grads().clone()
for lpb in (None, 10):
    red = GradientReducer(model, layers_per_bucket=lpb, exchange_when_alone=True)
    assert red.exchange_alone and red.cuda
    model.engine.grads().zero_()
    l1 = red.loss_and_backward(x, h, t)
    torch.cuda.synchronize()
    g1 = model.engine.grads()
    assert torch.equal(l0, l1), (float(l0), float(l1))
    if lpb is None:
        assert torch.equal(g0, g1), float((g0 - g1).abs().max())
    else:   # other launch groups: other split-K plans, round-off only
        assert float((g0 - g1).abs().max()) <= 2e-5 * float(g0.abs().max())
dist.destroy_process_group()
print("RCCL single-rank exchange ok")
"""


def test_rccl_backend_single_rank_exchange():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CODE % {"root": root, "port": port}], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0 and "RCCL single-rank exchange ok" in out, out[-3000:]
