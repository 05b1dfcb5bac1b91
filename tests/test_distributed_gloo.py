# -*- coding: utf-8 -*-
"""N>1 path on CPU: two processes (gloo, world_size 2), each running its shard of the minibatch
through the kernel emulator build, bucketed gradient all-reduce (GradientReducer) and FusedAdam.
The result must equal ONE process running the whole minibatch (global-batch mean loss, which is
what the reference's nn.DataParallel + CrossEntropyLoss(mean) computes, train.py:449-454,534-539)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wavenet_oracle as O

CFG = (32, 6, 8, 12, 3, 2, 2, 4)
B, T, SEED = 4, 48, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, btot):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytorchwavenetvocoder_amd.distributed import GradientReducer
        from pytorchwavenetvocoder_amd.nets import WaveNet
        from pytorchwavenetvocoder_amd.optim import FusedAdam
        from tests.emu_util import emu_library
        cfg = O.OracleConfig(*CFG)
        params = O.random_params(cfg, SEED)
        from pytorchwavenetvocoder_amd.bin.train import _shard_range
        x, h, t = O.synthetic_batch(cfg, btot, T, SEED + 1)
        lo, hi = _shard_range(btot, (rank, world))   # uneven when btot % world != 0 (train.py weights by B_local / B)
        sl = slice(lo, hi)
        model = WaveNet(*CFG, _library=emu_library())
        model.load_state_dict(params)
        opt = FusedAdam(model, lr=1e-3)
        red = GradientReducer(model, layers_per_bucket=2)
        assert red.world == world and len(red.ranges) == 1 + 3 + 1
        losses = []
        for _ in range(2):
            loss = red.loss_and_backward(x[sl].contiguous(), h[sl].contiguous(), t[sl].contiguous(),
                                         grad_scale=(hi - lo) / float(btot))
            opt.step()
            losses.append(float(loss))
        rep = red.comm_report()   # what bench.py's `comm` block prints for N > 1 (the exposed-exchange events are CUDA-only)
        assert rep["world"] == world and rep["backend"] == "gloo" and rep["buckets"] == len(red.ranges)
        assert rep["bucket_bytes"] == [4 * (b - a) for a, b in red.ranges] and sum(rep["bucket_bytes"]) == 4 * model.engine.n_params
        assert rep["exchange"].startswith("one in-place all-reduce per bucket") and "exposed_ms_per_step" not in rep
        if rank == 0:
            torch.save({"params": model.engine.flat_params.clone(), "losses": losses}, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("btot", [B, 3])
def test_two_rank_data_parallel_equals_single_process(tmp_path, btot):
    from pytorchwavenetvocoder_amd.nets import WaveNet
    from pytorchwavenetvocoder_amd.optim import FusedAdam
    from tests.emu_util import emu_library
    emu_library()  # build once in the parent
    out_path = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out_path, btot), nprocs=2, join=True)
    got = torch.load(out_path)

    cfg = O.OracleConfig(*CFG)
    params = O.random_params(cfg, SEED)
    x, h, t = O.synthetic_batch(cfg, btot, T, SEED + 1)
    model = WaveNet(*CFG, _library=emu_library())
    model.load_state_dict(params)
    opt = FusedAdam(model, lr=1e-3)
    for _ in range(2):
        model.loss_and_backward(x, h, t)
        opt.step()
    ref = model.engine.flat_params
    assert float((got["params"] - ref).abs().max()) <= 1e-2 * 1e-3  # 1e-2 * lr, see parity_common
    # and the single-process run itself matches the oracle's training step
    oparams = {k: v.clone() for k, v in params.items()}
    oopt = O.OracleAdam(lr=1e-3)
    for _ in range(2):
        O.train_step(cfg, oparams, oopt, x, h, t)
    for k, v in model.state_dict().items():
        assert float((v - oparams[k]).abs().max()) <= 1e-5, k


def test_rccl_footprint_default_respects_the_environment(monkeypatch):
    """bench.py / train.py cap RCCL at 16 channels before init_process_group("nccl") (the persistent chain kernels
    leave 16 CUs); an explicit setting in the environment wins."""
    from pytorchwavenetvocoder_amd.distributed import rccl_footprint_defaults
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    rccl_footprint_defaults()
    assert os.environ["NCCL_MAX_NCHANNELS"] == "16"
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "4")
    rccl_footprint_defaults()
    assert os.environ["NCCL_MAX_NCHANNELS"] == "4"


def test_bench_reports_the_stream_mode():
    import bench
    from pytorchwavenetvocoder_amd import _lib
    assert bench.stream_mode(0) == "serial (one stream)"
    assert "side stream" in bench.stream_mode(_lib.FLAG_BWD_OVERLAP | _lib.flag_dw_flush(10))
    assert "every 10 walked layers" in bench.stream_mode(_lib.FLAG_BWD_OVERLAP | _lib.flag_dw_flush(10))
    assert "skip-sum" in bench.stream_mode(_lib.FLAG_FWD_OVERLAP)
