#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Training CLI, drop-in for the reference's ``wavenet_vocoder/bin/train.py``.

Same command line (every flag of train.py:339-393, ``strtobool`` booleans, ``--resume ""``), same
outputs (``expdir/model.conf``, ``checkpoint-%d.pkl`` = {"model","optimizer","iterations"},
``checkpoint-final.pkl`` = {"model"}; train.py:315-332,429,564-568) and the same log lines
("(iter:%d) average loss = %.6f (%.3f sec / batch)", train.py:547), so ``egs/*/run.sh`` stage 4
runs unchanged.  What differs is how a step is executed:

  * forward / cross-entropy / backward run in the HIP kernels (``WaveNet.loss_and_backward``) and
    Adam is one fused launch over the flat parameter buffer (``FusedAdam``);
  * ``--n_gpus N`` starts ONE PROCESS PER GPU (torch.distributed over RCCL) instead of
    ``nn.DataParallel`` (train.py:449-454): every rank takes its 1/N slice of each minibatch along
    dim 0 -- exactly the chunks DataParallel scatters -- computes its own loss, and the flat
    gradient buffer is all-reduced in buckets on a side stream while backward is still running;
  * the wav / feature / stats readers fall back to scipy / .npz when soundfile / h5py are absent.
"""
import argparse
import logging
import os
import sys
import time

import numpy as np
import torch

from pytorchwavenetvocoder_amd.nets import WaveNet, encode_mu_law, initialize
from pytorchwavenetvocoder_amd.utils import background, extend_time, find_files, make_feat_transform, read_hdf5, read_txt


def strtobool(v):
    """distutils.util.strtobool (the reference's flag parser, train.py:13,366-369)."""
    v = str(v).lower()
    if v in ("y", "yes", "t", "true", "on", "1"):
        return 1
    if v in ("n", "no", "f", "false", "off", "0"):
        return 0
    raise ValueError("invalid truth value %r" % (v,))


def read_wav(path):
    """float32 waveform in [-1, 1] and sampling rate (soundfile when available, else scipy)."""
    try:
        import soundfile as sf
        return sf.read(path, dtype=np.float32)
    except ImportError:
        from scipy.io import wavfile
        fs, x = wavfile.read(path)
        if x.dtype == np.int16:
            x = x.astype(np.float32) / 32768.0
        elif x.dtype == np.int32:
            x = x.astype(np.float32) / 2147483648.0
        else:
            x = x.astype(np.float32)
        return x, fs


def validate_length(x, y, upsampling_factor=None):
    """Trim waveform ``x`` and features ``y`` to consistent lengths.  A deliberate RESTATEMENT of reference train.py:35-64
    (13 lines of required index behaviour; pinned against the reference by tests/golden/slicer.npz)."""
    if upsampling_factor is None:
        n = min(x.shape[0], y.shape[0])
        x, y = x[:n], y[:n]
        assert len(x) == len(y)
    else:
        if x.shape[0] > y.shape[0] * upsampling_factor:
            x = x[:y.shape[0] * upsampling_factor]
        if x.shape[0] < y.shape[0] * upsampling_factor:
            mod_y = y.shape[0] * upsampling_factor - x.shape[0]
            mod_y_frame = mod_y // upsampling_factor + 1
            y = y[:-mod_y_frame]
            x = x[:y.shape[0] * upsampling_factor]
        assert len(x) == len(y) * upsampling_factor
    return x, y


def _to_batch(xs, hs, ts, device, ys=None):
    tensors = [torch.stack(xs), torch.stack(hs), torch.stack(ts)] + ([torch.stack(ys)] if ys else [])
    if device is not None:
        if torch.device(device).type == "cuda":  # pinned staging: the copies overlap the step that is running
            tensors = [v.pin_memory() for v in tensors]
        tensors = [v.to(device, non_blocking=True) for v in tensors]
    if ys:
        return (tensors[0], tensors[1]), tensors[2], tensors[3]
    return (tensors[0], tensors[1]), tensors[2]


def _shard_range(batch_size, shard):
    """Window indices [lo, hi) of a minibatch owned by rank ``shard[0]`` of ``shard[1]`` -- the chunks
    ``nn.DataParallel`` scatters (reference train.py:449-454)."""
    if shard is None:
        return 0, batch_size
    rank, world = shard
    if batch_size < world:
        raise ValueError("batch_size (%d) must be >= the number of ranks (%d): every rank needs a window of each "
                         "minibatch, or it would never reach the gradient all-reduce" % (batch_size, world))
    # as even as possible, never empty: the first batch_size % world ranks own one window more
    per, extra = divmod(batch_size, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def slicer_workers():
    """Worker threads of the window slicer: ``WN_SLICER_WORKERS`` (0 = everything in the producer thread, as the reference does),
    default min(4, CPUs - 1).  numpy / zlib / file reads / torch.stack release the GIL, so threads do run in parallel here."""
    v = os.environ.get("WN_SLICER_WORKERS")
    if v is not None:
        return max(0, int(v))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(0, min(4, n - 1))


class _Done(object):
    """A finished 'future' (the inline path of the slicer: same code, no pool)."""

    def __init__(self, value):
        self.value = value

    def result(self):
        return self.value


@background(max_prefetch=16)
def train_generator(wav_list, feat_list, receptive_field,
                    batch_length=None,
                    batch_size=1,
                    feature_type="world",
                    wav_transform=None,
                    feat_transform=None,
                    shuffle=True,
                    upsampling_factor=80,
                    use_upsampling_layer=True,
                    use_speaker_code=False,
                    device="auto",
                    shard=None,
                    with_wave=False,
                    workers=None,
                    transforms_elementwise=False):
    """Minibatch generator with the reference's four batching modes (train.py:67-312).

    Yields ``((batch_x, batch_h), batch_t)``: x/t int64 (B, T) with t the next sample of x, h float
    (B, D, T) -- or (B, D, T / upsampling_factor) with the upsampling layer.  Windows hold
    ``receptive_field + batch_length`` samples and advance by ``batch_length`` (the first
    receptive_field outputs of every window carry no loss, train.py:535).

    ``shard=(rank, world)``: yield only this rank's windows of every minibatch (same minibatch
    composition as the unsharded generator; the mu-law / scaling work of the other ranks' windows is
    skipped instead of being done ``world`` times).
    ``with_wave=True`` (mixture-of-logistics head): additionally yields ``batch_y`` float (B, T), the waveform
    value of the next sample at every position (the un-quantised counterpart of ``batch_t``).

    ``workers`` (default ``slicer_workers()``): the reference's generator is ONE numpy thread behind a queue that is in effect one
    deep (train.py:67, utils.py:216) -- at 8.7 ms per minibatch on an MI355X that thread (file reads 3 ms, mu-law 1.5 ms, scaler,
    stacking, pinning: 6.5 ms per minibatch of 8 windows) is as slow as the step.  Here the SEQUENTIAL part -- the order of the
    utterances, the running buffers, where each window is cut: a few slices per window -- stays in the producer thread, and the
    work hangs off it as jobs of a thread pool whose results are consumed in submission order: (1) read + validate utterance
    i + 1 ... i + 8 while utterance i is being cut, (2) mu-law / scaler / tensor conversion per window, (3) stack + pinned
    staging + H2D per minibatch.  Same functions on the same slices in the same order: the minibatches are bit-identical to the
    single-thread generator's (tests/test_train_cli.py: the reference generator's golden minibatches, with and without workers).

    ``transforms_elementwise=True`` (what the CLI passes: its mu-law and scaler are per-sample / per-frame maps): the two
    transforms are applied ONCE per utterance when it is read instead of once per window on the buffer slices -- windows overlap
    by the receptive field (15 % of the samples at the benchmark's geometry), and the work moves into the coarse per-utterance
    read jobs.  Same values bit for bit (a per-element map commutes with slicing; same dtypes: golden test)."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    if device == "auto":
        device = torch.device("cuda") if torch.cuda.is_available() else None
    if shuffle:
        n_files = len(wav_list)
        idx = np.random.permutation(n_files)
        wav_list = [wav_list[i] for i in idx]
        feat_list = [feat_list[i] for i in idx]
    if batch_length is not None and use_upsampling_layer:
        batch_mod = (receptive_field + batch_length) % upsampling_factor
        logging.warning("batch length is decreased due to upsampling (%d -> %d)" % (
            batch_length, batch_length - batch_mod))
        batch_length -= batch_mod
    if batch_length is None and batch_size > 1:
        logging.warning("in utterance batch mode, batchsize will be 1.")
    pre = bool(transforms_elementwise)
    n_workers = slicer_workers() if workers is None else max(0, int(workers))
    pool = ThreadPoolExecutor(max_workers=n_workers, thread_name_prefix="wn_slicer") if n_workers > 0 else None

    def submit(fn, *a):
        return pool.submit(fn, *a) if pool is not None else _Done(fn(*a))

    def load(wavfile, featfile):
        """one utterance from its two files, trimmed to consistent lengths (train.py:202-212)"""
        x, _fs = read_wav(wavfile)
        h = read_hdf5(featfile, "/" + feature_type)
        if not use_upsampling_layer:
            h = extend_time(h, upsampling_factor)
        if use_speaker_code:
            sc = read_hdf5(featfile, "/speaker_code")
            h = np.concatenate([h, np.tile(sc, [h.shape[0], 1])], axis=1)
        x, h = validate_length(x, h, upsampling_factor if use_upsampling_layer else None)
        if not pre:
            return x, h, None, None
        # the buffers' dtypes (float32 waveform; features promoted with float32) are what the per-window transforms would see
        xf = np.asarray(x, dtype=np.float32)
        hf = np.asarray(h, dtype=np.result_type(np.float32, np.asarray(h).dtype))
        return xf, hf, (wav_transform(xf) if wav_transform is not None else xf), (feat_transform(hf) if feat_transform is not None else hf)

    def prep(x_, h_, xt_=None, ht_=None):
        """one window: (tokens, features, un-quantised waveform); xt_ / ht_: the slices of the already transformed utterances"""
        raw = torch.from_numpy(np.asarray(x_, dtype=np.float32))
        if xt_ is not None:
            x_, h_ = xt_, ht_
        else:
            if wav_transform is not None:
                x_ = wav_transform(x_)
            if feat_transform is not None:
                h_ = feat_transform(h_)
        return torch.from_numpy(np.asarray(x_)).long(), torch.from_numpy(np.asarray(h_)).float(), raw

    def assemble(window_jobs):
        """one minibatch from its windows' jobs (submitted before this one: the pool's queue is FIFO, so they are running or done)"""
        xs, hs, ts, ys = [], [], [], []
        for job in window_jobs:
            x_, h_, raw = job.result()
            hs.append(h_.transpose(0, 1) if use_upsampling_layer else h_[:-1].transpose(0, 1))
            xs.append(x_[:-1])
            ts.append(x_[1:])
            ys.append(raw[1:])
        return _to_batch(xs, hs, ts, device, ys if with_wave else None)

    def utterances():
        """(x, h) of the utterances in list order, epoch after epoch, ``None`` between two epochs (the reference's generator starts
        every walk of the list with an empty minibatch, train.py:196-200); the list is reshuffled where the reference reshuffles
        it (when it has been walked: the same sequence of np.random calls); up to ``ahead`` utterances are being read in front of
        the consumer, across the epoch boundary too"""
        nonlocal wav_list, feat_list
        ahead = 2 * n_workers if pool is not None else 0
        pending = deque()
        while True:
            for pair in zip(wav_list, feat_list):
                pending.append(submit(load, *pair))
                while len(pending) > ahead:
                    yield pending.popleft().result()
            pending.append(_Done(None))
            if shuffle:
                idx = np.random.permutation(n_files)
                wav_list = [wav_list[i] for i in idx]
                feat_list = [feat_list[i] for i in idx]

    # window sharding only exists in the windowed modes; utterance batches (batch_length None, effective batch size 1)
    # are dealt round-robin by utterance below, so a batch_size below the world size is fine there
    my_lo, my_hi = _shard_range(batch_size, shard) if batch_length is not None else (0, batch_size)
    ready = deque()                     # minibatch jobs in the order they will be yielded
    max_ready = 2 * n_workers + 1 if pool is not None else 1
    x_buffer = h_buffer = xt_buffer = ht_buffer = None
    window_jobs = []
    n_in_batch = 0
    n_seen = 0                          # utterances of the current epoch (utterance mode: round-robin over ranks)
    try:
        for utt in utterances():
            if utt is None:             # a new walk of the list: the reference drops a partly filled minibatch here (the buffers stay)
                window_jobs = []
                n_in_batch = 0
                n_seen = 0
                continue
            x, h, xt, ht = utt
            if batch_length is not None:
                # windowed minibatches over a running buffer of concatenated utterances
                if x_buffer is None:
                    x_buffer = np.empty((0), dtype=np.float32)
                    h_buffer = np.empty((0, h.shape[1]), dtype=np.float32)
                    if pre:
                        xt_buffer = np.empty((0), dtype=np.asarray(xt).dtype)
                        ht_buffer = np.empty((0, h.shape[1]), dtype=np.asarray(ht).dtype)
                x_buffer = np.concatenate([x_buffer, x], axis=0)
                h_buffer = np.concatenate([h_buffer, h], axis=0)
                if pre:
                    xt_buffer = np.concatenate([xt_buffer, xt], axis=0)
                    ht_buffer = np.concatenate([ht_buffer, ht], axis=0)
                if use_upsampling_layer:
                    h_bs = (receptive_field + batch_length) // upsampling_factor   # frames per window
                    x_bs = h_bs * upsampling_factor + 1                            # samples per window
                    h_ss = batch_length // upsampling_factor                       # window stride
                    x_ss = h_ss * upsampling_factor
                    more = lambda: len(h_buffer) > h_bs                            # noqa: E731
                else:
                    x_bs = h_bs = receptive_field + batch_length
                    x_ss = h_ss = batch_length
                    more = lambda: len(x_buffer) > x_bs                            # noqa: E731
                while more():
                    if my_lo <= n_in_batch < my_hi:   # windows of other ranks are only skipped over
                        if pre:
                            window_jobs.append(submit(prep, x_buffer[:x_bs], h_buffer[:h_bs], xt_buffer[:x_bs], ht_buffer[:h_bs]))
                        else:
                            window_jobs.append(submit(prep, x_buffer[:x_bs], h_buffer[:h_bs]))
                    n_in_batch += 1
                    h_buffer = h_buffer[h_ss:]
                    x_buffer = x_buffer[x_ss:]
                    if pre:
                        ht_buffer = ht_buffer[h_ss:]
                        xt_buffer = xt_buffer[x_ss:]
                    if n_in_batch == batch_size:
                        if window_jobs:
                            ready.append(submit(assemble, window_jobs))
                        window_jobs = []
                        n_in_batch = 0
                        while len(ready) >= max_ready:
                            yield ready.popleft().result()
            else:
                # one utterance per batch; with several ranks utterance i goes to rank i mod world
                n_seen += 1
                if shard is not None and (n_seen - 1) % shard[1] != shard[0]:
                    continue
                if use_upsampling_layer:
                    h = h[:-1]
                    x = x[:-upsampling_factor + 1]
                    if pre:
                        ht = ht[:-1]
                        xt = xt[:-upsampling_factor + 1]
                ready.append(submit(assemble, [submit(prep, x, h, xt, ht) if pre else submit(prep, x, h)]))
                while len(ready) >= max_ready:
                    yield ready.popleft().result()
    finally:
        if pool is not None:
            pool.shutdown(wait=False)


def save_checkpoint(checkpoint_dir, model, optimizer, iterations):
    """``checkpoint-%d.pkl`` = {"model", "optimizer", "iterations"} (reference train.py:315-332)."""
    checkpoint = {"model": model.state_dict(), "optimizer": optimizer.state_dict(), "iterations": iterations}
    if not os.path.exists(checkpoint_dir):
        os.makedirs(checkpoint_dir)
    torch.save(checkpoint, checkpoint_dir + "/checkpoint-%d.pkl" % iterations)
    logging.info("%d-iter checkpoint created." % iterations)


# (flag, default, type, help) -- the reference's command line, train.py:339-393
_FLAGS = [
    # paths
    ("waveforms", argparse.SUPPRESS, str, "wav directory or .scp list"),
    ("feats", argparse.SUPPRESS, str, "aux-feature directory or .scp list"),
    ("stats", argparse.SUPPRESS, str, "statistics file (mean / scale per feature type)"),
    ("expdir", argparse.SUPPRESS, str, "output directory for model.conf and checkpoints"),
    # network
    ("n_quantize", 256, int, "mu-law levels"),
    ("n_aux", 28, int, "aux feature dimension"),
    ("n_resch", 512, int, "residual channels"),
    ("n_skipch", 256, int, "skip channels"),
    ("dilation_depth", 10, int, "dilations 1..2^(depth-1) per cycle"),
    ("dilation_repeat", 1, int, "number of dilation cycles"),
    ("kernel_size", 2, int, "taps of the dilated causal convolutions"),
    ("upsampling_factor", 80, int, "samples per aux frame"),
    ("use_upsampling_layer", True, strtobool, "learned upsampling layer (else features are repeated on the host)"),
    ("use_speaker_code", False, strtobool, "append /speaker_code to the features"),
    # optimisation
    ("lr", 1e-4, float, "Adam learning rate"),
    ("weight_decay", 0.0, float, "L2 coefficient"),
    ("batch_length", 20000, int, "loss-bearing samples per window"),
    ("batch_size", 1, int, "windows per minibatch (global, split across GPUs)"),
    ("iters", 200000, int, "training iterations"),
    # misc
    ("checkpoint_interval", 10000, int, "iterations between checkpoints"),
    ("intervals", 100, int, "iterations between log lines"),
    ("seed", 1, int, "random seed"),
    ("n_gpus", 1, int, "GPUs (one process each)"),
    ("verbose", 1, int, "0 warnings, 1 info, >1 debug"),
]


def get_parser():
    parser = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    for name, default, typ, text in _FLAGS:
        if default is argparse.SUPPRESS:
            parser.add_argument("--" + name, required=True, type=typ, help=text)
        else:
            parser.add_argument("--" + name, default=default, type=typ, help=text)
    parser.add_argument("--feature_type", default="world", choices=["world", "melspc"], type=str)
    # extension (not a reference flag): mixture-of-logistics output head with N components, 0 = softmax head
    parser.add_argument("--n_mixture", default=0, type=int, help="mixture-of-logistics components (0: softmax head)")
    parser.add_argument("--log_scale_min", default=-7.0, type=float,
                        help="mixture head: clamp of the log-scales (saved in model.conf, so decode.py samples with it)")
    parser.add_argument("--resume", default=None, nargs="?", type=str, help="checkpoint to continue from")
    return parser


def _fmt_eta(seconds):
    seconds = int(seconds)
    days, rem = divmod(seconds, 86400)
    hours, rem = divmod(rem, 3600)
    minutes, secs = divmod(rem, 60)
    return "%02d:%02d:%02d:%02d" % (days, hours, minutes, secs)


def _worker(rank, world, args, port):
    """One training process (one GPU)."""
    import torch.distributed as dist
    from pytorchwavenetvocoder_amd.distributed import GradientReducer
    from pytorchwavenetvocoder_amd.optim import FusedAdam

    level = logging.INFO if args.verbose == 1 else logging.DEBUG if args.verbose > 1 else logging.WARNING
    logging.basicConfig(level=level, format='%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s',
                        datefmt='%m/%d/%Y %I:%M:%S')
    if args.verbose < 1:
        logging.warning("logging is disabled.")
    if not torch.cuda.is_available():
        logging.error("gpu is not available. please check the setting.")  # reference train.py:524-525
        sys.exit(1)
    # WN_TRAIN_BACKEND=gloo lets the N-rank control flow (sharded slicer, bucketed exchange, identical Adam) be exercised on a box
    # with fewer GPUs than ranks -- the ranks then share devices; real runs use nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("WN_TRAIN_BACKEND", "nccl")
    dev_index = rank if backend == "nccl" else rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        if backend == "nccl":
            from pytorchwavenetvocoder_amd.distributed import rccl_footprint_defaults
            rccl_footprint_defaults()
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    is_main = rank == 0
    if is_main:
        for key, value in vars(args).items():
            logging.info("%s = %s" % (key, str(value)))
        if not os.path.exists(args.expdir):
            os.makedirs(args.expdir)

    # fix seed (identical on every rank: same shuffles, same initial weights)
    os.environ['PYTHONHASHSEED'] = str(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if is_main:
        torch.save(args, args.expdir + "/model.conf")

    upsampling_factor = args.upsampling_factor if args.use_upsampling_layer else 0
    model = WaveNet(n_quantize=args.n_quantize, n_aux=args.n_aux, n_resch=args.n_resch, n_skipch=args.n_skipch,
                    dilation_depth=args.dilation_depth, dilation_repeat=args.dilation_repeat,
                    kernel_size=args.kernel_size, upsampling_factor=upsampling_factor, n_mixture=args.n_mixture,
                    log_scale_min=args.log_scale_min)
    if is_main:
        logging.info(model)
    model.apply(initialize)
    model.train()
    if world > args.batch_size:
        logging.warning("batch size is less than number of gpus.")

    # transforms (reference: StandardScaler + mu-law, train.py:464-470)
    mean = read_hdf5(args.stats, "/" + args.feature_type + "/mean")
    scale = read_hdf5(args.stats, "/" + args.feature_type + "/scale")
    wav_transform = lambda x: encode_mu_law(x, args.n_quantize)   # noqa: E731
    feat_transform = make_feat_transform(mean, scale)

    if os.path.isdir(args.waveforms):
        filenames = sorted(find_files(args.waveforms, "*.wav", use_dir_name=False))
        wav_list = [args.waveforms + "/" + filename for filename in filenames]
        feat_list = [args.feats + "/" + filename.replace(".wav", ".h5") for filename in filenames]
    elif os.path.isfile(args.waveforms):
        wav_list = read_txt(args.waveforms)
        feat_list = read_txt(args.feats)
    else:
        logging.error("--waveforms should be directory or list.")
        sys.exit(1)
    assert len(wav_list) == len(feat_list)
    logging.info("number of training data = %d." % len(wav_list))
    generator = train_generator(
        wav_list, feat_list,
        receptive_field=model.receptive_field,
        batch_length=args.batch_length,
        batch_size=args.batch_size,
        feature_type=args.feature_type,
        wav_transform=wav_transform,
        feat_transform=feat_transform,
        shuffle=True,
        upsampling_factor=args.upsampling_factor,
        use_upsampling_layer=args.use_upsampling_layer,
        shard=(rank, world) if world > 1 else None,
        with_wave=args.n_mixture > 0,
        use_speaker_code=args.use_speaker_code,
        device=device,
        transforms_elementwise=True)   # mu-law and the standard scaler are per-sample / per-frame maps

    optimizer = FusedAdam(model, lr=args.lr, weight_decay=args.weight_decay)
    if args.resume is not None and len(args.resume) != 0:
        checkpoint = torch.load(args.resume, map_location=lambda storage, loc: storage, weights_only=False)
        iterations = checkpoint["iterations"]
        model.load_state_dict(checkpoint["model"])
        optimizer.load_state_dict(checkpoint["optimizer"])
        logging.info("restored from %d-iter checkpoint." % iterations)
    else:
        iterations = 0
    model.to(device)
    reducer = GradientReducer(model)
    if world > 1:
        dist.broadcast(model.engine.flat_params, src=0)

    loss_acc = torch.zeros(1, device=device)
    total = 0.0
    for i in range(iterations, args.iters):
        start = time.time()
        item = generator.next()
        (batch_x, batch_h), batch_t = item[0], item[1]
        # A rank's loss is the mean over ITS windows: weighting it (and its gradients) by B_local / batch_size makes the
        # sum over ranks the mean over the whole minibatch -- what the reference's single CrossEntropyLoss over the
        # gathered logits computes (train.py:534-536) -- also when batch_size is not a multiple of the rank count.
        share = batch_x.size(0) / float(args.batch_size) if (world > 1 and args.batch_length is not None) else 1.0 / world
        batch_loss = reducer.loss_and_backward(batch_x, batch_h, batch_t, y=item[2] if args.n_mixture > 0 else None,
                                               grad_scale=share)
        optimizer.step()
        loss_acc += batch_loss.detach() * (share * world)
        if args.verbose > 1:
            logging.debug("batch loss = %.3f (%.3f sec / batch)" % (batch_loss.item(), time.time() - start))
        total += time.time() - start

        if (i + 1) % args.intervals == 0:
            torch.cuda.synchronize(device)
            if world > 1:
                dist.all_reduce(loss_acc)
                loss_acc /= world
            if is_main:
                logging.info("(iter:%d) average loss = %.6f (%.3f sec / batch)" % (
                    i + 1, loss_acc.item() / args.intervals, total / args.intervals))
                logging.info("estimated required time = " + _fmt_eta((args.iters - (i + 1)) * (total / args.intervals)))
            loss_acc.zero_()
            total = 0.0
        if (i + 1) % args.checkpoint_interval == 0 and is_main:
            save_checkpoint(args.expdir, model, optimizer, i + 1)

    generator.close()   # stop the producer thread before the interpreter tears the runtime down under it
    if is_main:
        torch.save({"model": model.state_dict()}, args.expdir + "/checkpoint-final.pkl")
        logging.info("final checkpoint created.")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    """RUN TRAINING."""
    args = get_parser().parse_args(argv)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    n_ranks = world_env if world_env > 1 else args.n_gpus
    if n_ranks > 1 and args.batch_length is not None and args.batch_size < n_ranks:
        # every rank must own at least one window of each minibatch (the reference scatters the same way)
        logging.error("--batch_size (%d) must be >= the number of GPUs (%d)." % (args.batch_size, n_ranks))
        sys.exit(1)
    if world_env > 1:  # launched by torch.distributed.run: one process per GPU already
        _worker(int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))), world_env, args,
                int(os.environ.get("MASTER_PORT", "29500")))
    elif args.n_gpus > 1:
        import socket
        import torch.multiprocessing as mp
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_worker, args=(args.n_gpus, args, port), nprocs=args.n_gpus, join=True)
    else:
        _worker(0, 1, args, 0)


if __name__ == "__main__":
    main()
