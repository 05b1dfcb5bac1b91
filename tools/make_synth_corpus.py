#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Synthetic corpus in the layout the recipe's stages 0-3 leave behind (egs/arctic/sd/run.sh): one training utterance and
two evaluation utterances of the "slt" speaker, 16 kHz wav + 28-dim "world" features at a 5 ms shift (80 samples per
frame), statistics file.  Files are written with the package's own helpers (hdf5 when h5py is there, the .npz container
otherwise -- the training / decoding CLIs read both)."""
import os
import sys

import numpy as np
from scipy.io import wavfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorchwavenetvocoder_amd.utils import write_hdf5  # noqa: E402

FS, SHIFT, DIM = 16000, 80, 28


def utterance(rs, frames):
    t = np.arange(frames * SHIFT + 40)
    f0 = rs.uniform(120.0, 220.0)
    x = 0.4 * np.sin(2 * np.pi * f0 * t / FS) + 0.1 * np.sin(2 * np.pi * 3.1 * f0 * t / FS) + 0.01 * rs.standard_normal(len(t))
    feats = rs.standard_normal((frames, DIM)).astype(np.float32)
    feats[:, 1] = f0          # a continuous-f0-like column, the rest noise
    return (x * 32767).astype(np.int16), feats


def main(recipe):
    rs = np.random.RandomState(0)
    for sub in ("data/tr_slt", "data/ev_slt", "wav_hpf/tr_slt", "hdf5/tr_slt", "hdf5/ev_slt"):
        os.makedirs(os.path.join(recipe, sub), exist_ok=True)
    wavs, feats = [], []
    for name, frames in (("arctic_a0001", 900),):
        x, h = utterance(rs, frames)
        w, f = "wav_hpf/tr_slt/%s.wav" % name, "hdf5/tr_slt/%s.h5" % name
        wavfile.write(os.path.join(recipe, w), FS, x)
        write_hdf5(os.path.join(recipe, f), "/world", h)
        wavs.append(w)
        feats.append(f)
    open(os.path.join(recipe, "data/tr_slt/wav_hpf.scp"), "w").write("\n".join(wavs) + "\n")
    open(os.path.join(recipe, "data/tr_slt/feats.scp"), "w").write("\n".join(feats) + "\n")
    stats = os.path.join(recipe, "data/tr_slt/stats.h5")
    write_hdf5(stats, "/world/mean", np.zeros(DIM, dtype=np.float32))
    write_hdf5(stats, "/world/scale", np.ones(DIM, dtype=np.float32))
    ev = []
    for name, frames in (("arctic_b0001", 12), ("arctic_b0002", 9)):
        _, h = utterance(rs, frames)
        f = "hdf5/ev_slt/%s.h5" % name
        write_hdf5(os.path.join(recipe, f), "/world", h)
        ev.append(f)
    open(os.path.join(recipe, "data/ev_slt/feats.scp"), "w").write("\n".join(ev) + "\n")


if __name__ == "__main__":
    main(sys.argv[1])
