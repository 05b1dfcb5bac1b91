# -*- coding: utf-8 -*-
"""Host side of the training CLI: flags, the four batching modes of train_generator (the shape
contracts of the reference's test/test_generator.py:53-135), checkpoint helpers.  CPU only -- the
HIP step itself is covered by the gpu-marked test at the bottom."""
import os

import numpy as np
import pytest
import torch

from pytorchwavenetvocoder_amd.bin import train as T
from pytorchwavenetvocoder_amd.nets import encode_mu_law
from pytorchwavenetvocoder_amd.utils import check_hdf5, read_hdf5, shape_hdf5, write_hdf5

FS, U, DIM = 16000, 80, 7


def make_corpus(root, n=3, seed=0):
    from scipy.io import wavfile
    rs = np.random.RandomState(seed)
    wavs, feats = [], []
    os.makedirs(os.path.join(root, "wav"))
    os.makedirs(os.path.join(root, "h5"))
    for i in range(n):
        frames = 60 + 7 * i
        x = (rs.uniform(-0.5, 0.5, size=frames * U + 13) * 32767).astype(np.int16)
        w = os.path.join(root, "wav", "utt%d.wav" % i)
        wavfile.write(w, FS, x)
        f = os.path.join(root, "h5", "utt%d.h5" % i)
        write_hdf5(f, "/melspc", rs.standard_normal((frames, DIM)).astype(np.float32))
        wavs.append(w)
        feats.append(f)
    stats = os.path.join(root, "stats.h5")
    write_hdf5(stats, "/melspc/mean", np.zeros(DIM, dtype=np.float32))
    write_hdf5(stats, "/melspc/scale", np.ones(DIM, dtype=np.float32))
    return wavs, feats, stats


def test_hdf5_helpers_roundtrip(tmp_path):
    f = str(tmp_path / "a.h5")
    write_hdf5(f, "/world", np.arange(12.0).reshape(3, 4))
    write_hdf5(f, "/melspc/mean", np.ones(5))
    assert check_hdf5(f, "/world") and check_hdf5(f, "/melspc/mean") and not check_hdf5(f, "/nope")
    assert shape_hdf5(f, "/world") == (3, 4)
    np.testing.assert_array_equal(read_hdf5(f, "/world"), np.arange(12.0).reshape(3, 4))


def test_flags_match_reference_cli():
    p = T.get_parser()
    a = p.parse_args(["--waveforms", "w", "--feats", "f", "--stats", "s", "--expdir", "e", "--resume", "",
                      "--use_upsampling_layer", "false", "--n_gpus", "2", "--feature_type", "melspc"])
    assert a.resume == "" and a.use_upsampling_layer == 0 and a.n_gpus == 2
    # defaults of reference train.py:339-393
    assert (a.n_quantize, a.n_aux, a.n_resch, a.n_skipch, a.dilation_depth, a.dilation_repeat, a.kernel_size) == \
        (256, 28, 512, 256, 10, 1, 2)
    assert (a.upsampling_factor, a.lr, a.weight_decay, a.batch_length, a.batch_size, a.iters) == \
        (80, 1e-4, 0.0, 20000, 1, 200000)
    assert (a.checkpoint_interval, a.intervals, a.seed, a.verbose) == (10000, 100, 1, 1)


@pytest.mark.parametrize("batch_length,upsample", [(None, True), (None, False), (1000, True), (1000, False)])
def test_generator_modes(tmp_path, batch_length, upsample):
    wavs, feats, stats = make_corpus(str(tmp_path))
    rf, bs = 300, 2
    gen = T.train_generator(wavs, feats, receptive_field=rf, batch_length=batch_length, batch_size=bs,
                            feature_type="melspc", wav_transform=lambda x: encode_mu_law(x, 256),
                            feat_transform=lambda h: h, shuffle=False, upsampling_factor=U,
                            use_upsampling_layer=upsample, device=None)
    for _ in range(4):
        (x, h), t = gen.next()
        assert x.dtype == torch.int64 and t.dtype == torch.int64 and h.dtype == torch.float32
        assert x.size(1) == t.size(1) and h.size(1) == DIM
        assert x.size(0) == (bs if batch_length is not None else 1)
        if upsample:
            assert h.size(2) * U == x.size(1)          # reference test_generator.py:80-81
        else:
            assert h.size(2) == x.size(1)
        assert torch.equal(x[:, 1:], t[:, :-1])        # t is x advanced by one sample
        if batch_length is not None and upsample:
            bl = batch_length - (rf + batch_length) % U
            assert x.size(1) == (rf + bl) // U * U     # reference train.py:106-110,204-205
        assert int(x.min()) >= 0 and int(x.max()) < 256
    gen.close()


def test_sharded_generator_is_a_partition_of_the_unsharded_one(tmp_path):
    """shard=(rank, world): every rank yields its DataParallel chunk of the same minibatches."""
    wavs, feats, _ = make_corpus(str(tmp_path), n=4)
    kw = dict(receptive_field=15, batch_length=400, batch_size=5, feature_type="melspc", shuffle=False,
              wav_transform=lambda x: encode_mu_law(x, 256), upsampling_factor=U, use_upsampling_layer=True, device=None)
    full = T.train_generator(wavs, feats, **kw)
    parts = [T.train_generator(wavs, feats, shard=(r, 2), **kw) for r in range(2)]
    try:
        for _ in range(3):
            (fx, fh), ft = full.next()
            got = [p.next() for p in parts]
            assert [g[0][0].size(0) for g in got] == [3, 2]
            assert torch.equal(torch.cat([g[0][0] for g in got]), fx)
            assert torch.equal(torch.cat([g[0][1] for g in got]), fh)
            assert torch.equal(torch.cat([g[1] for g in got]), ft)
    finally:
        full.close()
        for p in parts:
            p.close()


def test_shard_ranges_are_never_empty_and_partition_the_minibatch():
    """batch_size not a multiple of the rank count (12 windows on 8 GPUs): every rank owns at least one window (an empty
    rank would never reach the gradient all-reduce and the step would deadlock), sizes differ by at most one, and the
    ranges tile [0, batch_size) in rank order.  Fewer windows than ranks is an error, not a hang."""
    for bs, world in [(12, 8), (5, 2), (8, 8), (64, 8), (9, 4)]:
        rs = [T._shard_range(bs, (r, world)) for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == bs
        assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
        sizes = [hi - lo for lo, hi in rs]
        assert min(sizes) >= 1 and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        T._shard_range(3, (0, 4))
    assert T._shard_range(7, None) == (0, 7)


def test_generator_yields_the_waveform_targets_of_the_mixture_head(tmp_path):
    wavs, feats, _ = make_corpus(str(tmp_path), n=2)
    gen = T.train_generator(wavs, feats, receptive_field=15, batch_length=400, batch_size=2, feature_type="melspc",
                            shuffle=False, wav_transform=lambda x: encode_mu_law(x, 256), upsampling_factor=U,
                            use_upsampling_layer=True, device=None, with_wave=True)
    try:
        (bx, bh), bt, by = gen.next()
        assert by.dtype == torch.float32 and tuple(by.shape) == tuple(bt.shape)
        assert float(by.abs().max()) <= 1.0
        # the mu-law token of the waveform target is the token target
        assert torch.equal(torch.from_numpy(encode_mu_law(by.numpy().astype(np.float64), 256)).long(), bt)
    finally:
        gen.close()


def test_validate_length():
    x, y = T.validate_length(np.zeros(1000), np.zeros((12, 3)), 80)
    assert len(x) == len(y) * 80
    x, y = T.validate_length(np.zeros(900), np.zeros((12, 3)), 80)
    assert len(x) == len(y) * 80 and len(y) == 11
    x, y = T.validate_length(np.zeros(10), np.zeros((12, 3)))
    assert len(x) == len(y) == 10


@pytest.mark.gpu
def test_train_cli_runs_checkpoints_and_resumes(tmp_path):
    """End to end on the GPU: 6 iterations, checkpoint at 4, resume to 6 -> identical final weights."""
    wavs, feats, stats = make_corpus(str(tmp_path), n=4)
    scp_w, scp_f = str(tmp_path / "wav.scp"), str(tmp_path / "feats.scp")
    open(scp_w, "w").write("\n".join(wavs) + "\n")
    open(scp_f, "w").write("\n".join(feats) + "\n")
    common = ["--waveforms", scp_w, "--feats", scp_f, "--stats", stats, "--feature_type", "melspc",
              "--n_aux", str(DIM), "--n_resch", "64", "--n_skipch", "32", "--dilation_depth", "4",
              "--dilation_repeat", "2", "--upsampling_factor", str(U), "--batch_length", "800", "--batch_size", "2",
              "--intervals", "2", "--checkpoint_interval", "4", "--lr", "1e-3", "--verbose", "0"]
    e1, e2 = str(tmp_path / "exp1"), str(tmp_path / "exp2")
    T.main(common + ["--expdir", e1, "--iters", "6", "--resume", ""])
    assert os.path.exists(e1 + "/model.conf") and os.path.exists(e1 + "/checkpoint-4.pkl")
    final1 = torch.load(e1 + "/checkpoint-final.pkl", weights_only=False)["model"]
    ck = torch.load(e1 + "/checkpoint-4.pkl", weights_only=False)
    assert set(ck.keys()) == {"model", "optimizer", "iterations"} and ck["iterations"] == 4
    assert list(final1.keys())[:2] == ["causal.conv.weight", "causal.conv.bias"]
    conf = torch.load(e1 + "/model.conf", weights_only=False)
    assert conf.n_resch == 64
    # the loss must go down on this tiny corpus and all weights stay finite
    assert all(torch.isfinite(v).all() for v in final1.values())


# ---- the window slicer against the REFERENCE'S OWN generator (tests/golden/slicer.npz) ----
def _write_slicer_corpus(root):
    from scipy.io import wavfile
    from tests import golden_util as GU
    utts, mean, scale = GU.slicer_corpus()
    wavs, feats = [], []
    for i, (wav, feat, code) in enumerate(utts):
        w, f = os.path.join(root, "utt%d.wav" % i), os.path.join(root, "utt%d.h5" % i)
        wavfile.write(w, FS, wav)
        write_hdf5(f, "/melspc", feat)
        write_hdf5(f, "/speaker_code", code)
        wavs.append(w)
        feats.append(f)
    return wavs, feats, mean, scale


def _slicer_kwargs(name, mean, scale):
    from tests import golden_util as GU
    bl, bs, up, spk, sdt, nb = GU.SLICER_CASES[name]
    mean, scale = mean.astype(sdt), scale.astype(sdt)
    if spk:
        mean = np.concatenate([mean, np.zeros(2, sdt)])
        scale = np.concatenate([scale, np.ones(2, sdt)])
    kw = dict(receptive_field=GU.SLICER_RF, batch_length=bl, batch_size=bs, feature_type="melspc",
              wav_transform=lambda x: encode_mu_law(x, GU.SLICER_Q), feat_transform=T.make_feat_transform(mean, scale),
              shuffle=False, upsampling_factor=GU.SLICER_U, use_upsampling_layer=up, use_speaker_code=spk, device=None)
    return kw, nb


@pytest.mark.parametrize("workers,pre", [(0, False), (3, False), (0, True), (3, True)])
@pytest.mark.parametrize("name", ["utt_up", "utt_noup", "win_up", "win_noup", "win_up_spk", "win_up_f64stats"])
def test_window_slicer_is_bit_equal_to_the_reference_generator(tmp_path, name, workers, pre):
    """x, h, t of every minibatch are BIT-equal to what the reference's train_generator (train.py:67-312, with
    validate_length :35-64, mu-law and the StandardScaler transform :463-470) yields on the same corpus, in all four
    batching modes, with the speaker code, across the end of an epoch (the carried-over buffer) -- fixture written by
    tests/golden/make_slicer_golden.py from the reference's own code.  ``workers``: everything in the producer thread (0, the
    reference's structure) and with the order-preserving worker pool (round 6: file reads, per-window transforms and
    minibatch assembly as pool jobs consumed in submission order).  ``pre``: the two per-element transforms applied once per
    utterance at read time (what the CLI does) instead of once per window."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "slicer.npz"))
    wavs, feats, mean, scale = _write_slicer_corpus(str(tmp_path))
    kw, nb = _slicer_kwargs(name, mean, scale)
    gen = T.train_generator(wavs, feats, workers=workers, transforms_elementwise=pre, **kw)
    try:
        for i in range(nb):
            (x, h), t = gen.next()
            assert x.dtype == torch.int64 and t.dtype == torch.int64 and h.dtype == torch.float32
            np.testing.assert_array_equal(x.numpy(), z["%s/%d/x" % (name, i)].astype(np.int64))
            np.testing.assert_array_equal(t.numpy(), z["%s/%d/t" % (name, i)].astype(np.int64))
            gh = z["%s/%d/h" % (name, i)]
            assert h.numpy().shape == gh.shape and (h.numpy().view(np.uint32) == gh.view(np.uint32)).all()
    finally:
        gen.close()


def test_sharded_window_slicer_against_the_reference_generator(tmp_path):
    """Two ranks' shards of the windowed mode, concatenated in rank order, are the reference's minibatches."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "slicer.npz"))
    wavs, feats, mean, scale = _write_slicer_corpus(str(tmp_path))
    kw, nb = _slicer_kwargs("win_up_spk", mean, scale)
    parts = [T.train_generator(wavs, feats, shard=(r, 2), **kw) for r in range(2)]
    try:
        for i in range(nb):
            got = [p.next() for p in parts]
            np.testing.assert_array_equal(torch.cat([g[0][0] for g in got]).numpy(), z["win_up_spk/%d/x" % i].astype(np.int64))
            np.testing.assert_array_equal(torch.cat([g[0][1] for g in got]).numpy(), z["win_up_spk/%d/h" % i])
    finally:
        for p in parts:
            p.close()


def test_validate_length_against_the_reference():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "slicer.npz"))
    for nx, ny, U, lx, ly in z["validate_length"]:
        x, y = T.validate_length(np.arange(nx), np.zeros((ny, 3)), None if U < 0 else int(U))
        assert (len(x), len(y)) == (lx, ly)


def test_utterance_mode_with_more_ranks_than_batch_size_does_not_hang(tmp_path):
    """batch_length None (utterance batches, effective batch size 1, the CLI default batch_size) on 2 ranks: utterances are
    dealt round-robin; the window-shard check (batch_size >= world) does not apply (round-2 advisor finding)."""
    wavs, feats, _ = make_corpus(str(tmp_path), n=4)
    kw = dict(receptive_field=15, batch_length=None, batch_size=1, feature_type="melspc", shuffle=False,
              wav_transform=lambda x: encode_mu_law(x, 256), upsampling_factor=U, use_upsampling_layer=True, device=None)
    full = T.train_generator(wavs, feats, **kw)
    parts = [T.train_generator(wavs, feats, shard=(r, 2), **kw) for r in range(2)]
    try:
        for i in range(4):
            (fx, _fh), _ft = full.next()
            (px, _ph), _pt = parts[i % 2].next()
            assert torch.equal(fx, px)
    finally:
        full.close()
        for p in parts:
            p.close()


def test_a_dying_producer_raises_in_the_consumer_instead_of_hanging(tmp_path):
    """An exception inside the generator thread reaches the consumer's next() (it used to leave it blocked forever)."""
    wavs, feats, _ = make_corpus(str(tmp_path), n=2)
    gen = T.train_generator(wavs, feats, receptive_field=15, batch_length=400, batch_size=1, feature_type="melspc",
                            shuffle=False, upsampling_factor=U, device=None, shard=(0, 2))   # batch_size < world
    with pytest.raises(ValueError):
        gen.next()
