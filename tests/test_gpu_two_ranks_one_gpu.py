# -*- coding: utf-8 -*-
"""The N > 1 step on the hardware one box has: two ranks (gloo rendezvous and all-reduce, both processes on cuda:0) run
the CUDA branch of ``GradientReducer`` -- bucket events recorded by wn_backward, all-reduce of each contiguous gradient
range on a side stream while the backward kernels of the next layers run, join before Adam (reference being replaced:
nn.DataParallel, train.py:449-454).  The all-reduced gradient buffer must be BIT-identical to what one process gets
from the same two half-batches in the same bucket structure (g_rank0 + g_rank1: a two-term sum has one order), and
equal to the full-batch gradient to round-off."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = (256, 80, 64, 256, 10, 3, 2, 80)
B, T, SEED, LPB = 4, 3200, 51, 10


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import wavenet_oracle as O
        from pytorchwavenetvocoder_amd.distributed import GradientReducer
        from pytorchwavenetvocoder_amd.nets import WaveNet
        from pytorchwavenetvocoder_amd.optim import FusedAdam
        dev = torch.device("cuda", 0)
        cfg = O.OracleConfig(*CFG)
        params = O.random_params(cfg, SEED, scale=0.05)
        x, h, t = O.synthetic_batch(cfg, B, T, SEED + 1)
        per = B // world
        sl = slice(rank * per, (rank + 1) * per)
        model = WaveNet(*CFG)
        model.load_state_dict(params)
        model.to(dev)
        red = GradientReducer(model, layers_per_bucket=LPB)
        assert red.cuda and red.world == world and len(red.events) == len(red.ranges) == 1 + 3 + 1
        opt = FusedAdam(model, lr=1e-3)
        loss = red.loss_and_backward(x[sl].contiguous().to(dev), h[sl].contiguous().to(dev), t[sl].contiguous().to(dev))
        torch.cuda.synchronize()
        g = model.engine.grads().clone()
        opt.step()
        torch.cuda.synchronize()
        torch.save({"grads": g.cpu(), "loss": float(loss), "params": model.engine.flat_params.cpu()},
                   os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_through_the_cuda_reducer(tmp_path):
    import torch.multiprocessing as mp
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd.nets import WaveNet
    from pytorchwavenetvocoder_amd.optim import FusedAdam
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(str(tmp_path / "rank0.pt"))
    r1 = torch.load(str(tmp_path / "rank1.pt"))
    assert torch.equal(r0["grads"], r1["grads"]) and torch.equal(r0["params"], r1["params"])   # identical replicas
    dev = torch.device("cuda", 0)
    cfg = O.OracleConfig(*CFG)
    params = O.random_params(cfg, SEED, scale=0.05)
    x, h, t = O.synthetic_batch(cfg, B, T, SEED + 1)
    model = WaveNet(*CFG)
    model.load_state_dict(params)
    model.to(dev)
    halves, losses = [], []
    for r in range(2):
        sl = slice(r * (B // 2), (r + 1) * (B // 2))
        loss = model.loss_and_backward(x[sl].contiguous().to(dev), h[sl].contiguous().to(dev), t[sl].contiguous().to(dev),
                                       grad_scale=0.5, layers_per_bucket=LPB)
        halves.append(model.engine.grads().clone())
        losses.append(float(loss))
    assert losses == [r0["loss"], r1["loss"]]
    assert torch.equal((halves[0] + halves[1]).cpu(), r0["grads"]), "all-reduced gradients differ from g_rank0 + g_rank1"
    # one process on the whole minibatch (what the reference's single loss over the gathered logits gives)
    model.loss_and_backward(x.to(dev), h.to(dev), t.to(dev), layers_per_bucket=LPB)
    full = model.engine.grads().cpu()
    assert float((full - r0["grads"]).abs().max()) <= 2e-5 * float(full.abs().max())
    opt = FusedAdam(model, lr=1e-3)
    opt.step()
    torch.cuda.synchronize()
    assert float((model.engine.flat_params.cpu() - r0["params"]).abs().max()) <= 1e-2 * 1e-3


def test_train_cli_with_eight_ranks_on_one_gpu(tmp_path):
    """``train.py --n_gpus 8`` (reference flag train.py:388-389; there: nn.DataParallel over 8 devices, :449-454) as EIGHT
    processes: the argument path, the rendezvous, the slicer's window sharding with UNEVEN shards (batch_size 11 over 8 ranks:
    three ranks own two windows, five own one), the weighted loss, the bucketed exchange, identical Adam steps and rank 0's
    checkpoints.  The pool's boxes have one GPU, so the ranks share it and rendezvous over gloo (WN_TRAIN_BACKEND=gloo: the
    all-reduce of CUDA tensors stages through the host); everything above the transport is what an 8-GPU node runs."""
    import subprocess
    import sys
    import numpy as np
    from tests.test_train_cli import make_corpus
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    wavs, feats, stats = make_corpus(str(tmp_path), n=12)
    (tmp_path / "wav.scp").write_text("\n".join(wavs) + "\n")
    (tmp_path / "feats.scp").write_text("\n".join(feats) + "\n")
    expdir = str(tmp_path / "exp")
    env = dict(os.environ, WN_TRAIN_BACKEND="gloo", PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "wavenet_vocoder.bin.train", "--waveforms", str(tmp_path / "wav.scp"), "--feats", str(tmp_path / "feats.scp"),
           "--stats", stats, "--expdir", expdir, "--feature_type", "melspc", "--n_aux", "7", "--n_resch", "64", "--n_skipch", "32",
           "--dilation_depth", "3", "--dilation_repeat", "1", "--upsampling_factor", "80", "--batch_length", "400", "--batch_size", "11",
           "--iters", "6", "--intervals", "3", "--checkpoint_interval", "3", "--n_gpus", "8", "--verbose", "1"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "(iter:6) average loss" in r.stderr and "final checkpoint created." in r.stderr
    ck3 = torch.load(os.path.join(expdir, "checkpoint-3.pkl"), weights_only=False)
    assert ck3["iterations"] == 3 and np.isfinite(float(next(iter(ck3["model"].values())).abs().sum()))
    assert os.path.exists(os.path.join(expdir, "checkpoint-final.pkl"))
