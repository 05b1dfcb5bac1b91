#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""End-to-end rate of ``wavenet_vocoder/bin/train.py`` on the BENCHMARK's model and minibatch geometry:
real files -> window slicer (reference train.py:67-248) -> pinned H2D -> training step, against bench.py's device-resident step.

    python tools/train_e2e_bench.py --producer-only            # CPU: ms per minibatch of the slicer alone (no GPU needed)
    python tools/train_e2e_bench.py --iters 300                # GPU: train.py end to end, (sec / batch) of its own log lines

A synthetic corpus is written to a temporary directory first: N utterances of 16 kHz audio + 80-dim "melspc" features at an
80-sample shift + the statistics file, in the layout the recipes' stages 0-3 leave (wav list + feature list)."""
import argparse
import io
import json
import logging
import os
import re
import sys
import tempfile
import time

import numpy as np
from scipy.io import wavfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FS, SHIFT, DIM = 16000, 80, 80


def make_corpus(root, n_utt, frames, seed=0):
    from pytorchwavenetvocoder_amd.utils import write_hdf5
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "wav"), exist_ok=True)
    os.makedirs(os.path.join(root, "hdf5"), exist_ok=True)
    wavs, feats = [], []
    for i in range(n_utt):
        n = frames + int(rs.randint(-50, 50))
        t = np.arange(n * SHIFT + int(rs.randint(0, 60)))
        f0 = rs.uniform(100.0, 240.0)
        x = 0.4 * np.sin(2 * np.pi * f0 * t / FS) + 0.1 * np.sin(2 * np.pi * 2.7 * f0 * t / FS) + 0.02 * rs.standard_normal(len(t))
        w = os.path.join(root, "wav", "utt%04d.wav" % i)
        f = os.path.join(root, "hdf5", "utt%04d.h5" % i)
        wavfile.write(w, FS, (x * 32767).astype(np.int16))
        write_hdf5(f, "/melspc", rs.standard_normal((n, DIM)).astype(np.float32))
        wavs.append(w)
        feats.append(f)
    open(os.path.join(root, "wav.scp"), "w").write("\n".join(wavs) + "\n")
    open(os.path.join(root, "feats.scp"), "w").write("\n".join(feats) + "\n")
    stats = os.path.join(root, "stats.h5")
    write_hdf5(stats, "/melspc/mean", np.zeros(DIM, dtype=np.float32))
    write_hdf5(stats, "/melspc/scale", np.ones(DIM, dtype=np.float32))
    return os.path.join(root, "wav.scp"), os.path.join(root, "feats.scp"), stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=48)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--batch_length", type=int, default=20000)
    ap.add_argument("--producer-only", action="store_true")
    ap.add_argument("--batches", type=int, default=60, help="minibatches timed with --producer-only")
    ap.add_argument("--workers", type=int, default=None, help="WN_SLICER_WORKERS for the run")
    ap.add_argument("--per-window-transforms", action="store_true",
                    help="--producer-only: transforms once per window (the reference's structure) instead of once per utterance (the CLI)")
    args = ap.parse_args()
    if args.workers is not None:
        os.environ["WN_SLICER_WORKERS"] = str(args.workers)
    tmp = tempfile.mkdtemp(prefix="wn_e2e_")
    wav_scp, feat_scp, stats = make_corpus(tmp, args.utterances, args.frames)
    from pytorchwavenetvocoder_amd.bin import train as T
    out = {"utterances": args.utterances, "frames_per_utterance": args.frames, "batch_size": args.batch_size,
           "batch_length": args.batch_length, "slicer_workers": os.environ.get("WN_SLICER_WORKERS"),
           "transforms": "per window" if args.per_window_transforms else "per utterance"}
    if args.producer_only:
        from pytorchwavenetvocoder_amd.nets import encode_mu_law
        from pytorchwavenetvocoder_amd.utils import make_feat_transform, read_hdf5, read_txt
        ft = make_feat_transform(read_hdf5(stats, "/melspc/mean"), read_hdf5(stats, "/melspc/scale"))
        gen = T.train_generator(read_txt(wav_scp), read_txt(feat_scp), receptive_field=3070, batch_length=args.batch_length,
                                batch_size=args.batch_size, feature_type="melspc", wav_transform=lambda x: encode_mu_law(x, 256),
                                feat_transform=ft, shuffle=True, upsampling_factor=80, use_upsampling_layer=True, device=None,
                                transforms_elementwise=not args.per_window_transforms)
        gen.next()
        time.sleep(1.0)          # let the prefetch queue fill: the steady-state rate is what is measured
        for _ in range(20):
            gen.next()
        t0 = time.time()
        for _ in range(args.batches):
            gen.next()
        out["producer_ms_per_minibatch"] = (time.time() - t0) / args.batches * 1e3
        print(json.dumps(out))
        return
    stream = io.StringIO()
    stamps = []

    class Stamping(logging.StreamHandler):
        def emit(self, record):
            if "average loss" in record.getMessage():
                stamps.append(time.time())     # train.py synchronises the device right before this line: true time per interval
            logging.StreamHandler.emit(self, record)
    handler = Stamping(stream)
    handler.setLevel(logging.INFO)
    logging.getLogger().addHandler(handler)
    logging.getLogger().setLevel(logging.INFO)
    expdir = os.path.join(tmp, "exp")
    t0 = time.time()
    T.main(["--waveforms", wav_scp, "--feats", feat_scp, "--stats", stats, "--expdir", expdir, "--feature_type", "melspc",
            "--n_aux", "80", "--n_resch", "64", "--n_skipch", "256", "--dilation_depth", "10", "--dilation_repeat", "3",
            "--kernel_size", "2", "--upsampling_factor", "80", "--batch_size", str(args.batch_size),
            "--batch_length", str(args.batch_length), "--iters", str(args.iters), "--intervals", "50",
            "--checkpoint_interval", "1000000", "--verbose", "1"])
    out["wall_s"] = time.time() - t0
    secs = [float(m.group(1)) for m in re.finditer(r"\(([0-9.]+) sec / batch\)", stream.getvalue())]
    out["sec_per_batch_by_interval"] = secs
    steady = secs[1:] if len(secs) > 1 else secs
    out["host_ms_per_iteration_as_logged"] = 1e3 * sum(steady) / max(len(steady), 1)   # train.py's own figure: host time, launches are asynchronous
    gaps = [b - a for a, b in zip(stamps[:-1], stamps[1:])]
    out["end_to_end_ms_per_iteration"] = (1e3 * sum(gaps) / (50.0 * len(gaps))) if gaps else None   # wall time between log lines (device synchronised there)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
