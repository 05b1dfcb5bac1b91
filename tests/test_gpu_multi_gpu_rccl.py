# -*- coding: utf-8 -*-
"""The data-parallel path ON REAL MULTI-GPU HARDWARE: two RCCL ranks, one per GPU (the reference's counterpart is
nn.DataParallel, train.py:449-454).  Runs when the box shows at least two devices and SKIPS otherwise -- the pool's GPU boxes
have one; the first multi-GPU box this suite meets validates the path by itself:

  * ``bench.py --gpus 2`` under ``torch.distributed.run`` (exactly the driver's launch line) prints one JSON line whose
    ``comm`` block reports 2 ranks from the ``nccl`` backend, three gradient buckets and the exposed exchange time per step;
  * three training steps of two ranks on two halves of a minibatch leave BIT-identical parameters on both ranks, equal (to
    round-off of the summation order) to one process stepping through the whole minibatch."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _need_two_gpus():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (this box shows %d): the scaling path is validated by the 2-rank gloo tests, the two-processes-"
                    "on-one-GPU test and the single-rank RCCL test instead" % torch.cuda.device_count())


def test_bench_two_ranks_over_rccl():
    _need_two_gpus()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--repeats", "1", "--profile-steps", "0", "--no-decode", "--no-extras", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16
    c = d["comm"]
    assert c["ranks_reported_by_backend"] == 2 and c["backend"] == "nccl" and c["buckets"] == 3
    assert c["exposed_ms_per_step"]["steps"] == 5 and c["exposed_ms_per_step"]["mean"] >= 0.0
    print("2 x MI355X over RCCL: %.3f ms per step, exposed exchange %.3f ms per step" % (d["ms_per_step"], c["exposed_ms_per_step"]["mean"]))


CODE = r"""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from pytorchwavenetvocoder_amd.distributed import GradientReducer, rccl_footprint_defaults
from pytorchwavenetvocoder_amd.nets import WaveNet, initialize
from pytorchwavenetvocoder_amd.optim import FusedAdam
from oracle import wavenet_oracle as O
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
rccl_footprint_defaults()
dist.init_process_group("nccl", device_id=dev)
cfg_t = (256, 80, 64, 256, 10, 3, 2, 80)
cfg = O.OracleConfig(*cfg_t)
torch.manual_seed(3)
model = WaveNet(*cfg_t); model.apply(initialize); model.to(dev)
dist.broadcast(model.engine.flat_params, src=0)
x, h, t = O.synthetic_batch(cfg, 4, 3200 + 80 * 3, 9)
lo, hi = rank * 2, rank * 2 + 2
opt = FusedAdam(model, lr=1e-4)
red = GradientReducer(model)
for _ in range(3):
    red.loss_and_backward(x[lo:hi].to(dev), h[lo:hi].to(dev), t[lo:hi].to(dev))
    opt.step()
torch.cuda.synchronize()
mine = model.engine.flat_params.clone()
other = mine.clone()
dist.broadcast(other, src=0)
assert torch.equal(mine, other), "rank %%d diverged from rank 0" %% rank
if rank == 0:   # one process, the whole minibatch
    torch.manual_seed(3)
    ref = WaveNet(*cfg_t); ref.apply(initialize); ref.to(dev)
    ropt = FusedAdam(ref, lr=1e-4)
    for _ in range(3):
        ref.loss_and_backward(x.to(dev), h.to(dev), t.to(dev))
        ropt.step()
    torch.cuda.synchronize()
    d = float((ref.engine.flat_params - mine).abs().max())
    assert d <= 3 * 1e-4 * 0.05, d      # three Adam steps of lr 1e-4: sign-like updates may differ on a few near-zero gradients
    print("two RCCL ranks == one process: max |dw| %%.3g after 3 steps" %% d)
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rccl_ranks_keep_identical_parameters_and_match_one_process(tmp_path):
    _need_two_gpus()
    script = tmp_path / "two_ranks.py"
    script.write_text(CODE % {"root": ROOT})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "two RCCL ranks == one process" in r.stdout
