#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Golden fixture of the WINDOW SLICER: runs the reference's own ``train_generator``
(/root/reference/wavenet_vocoder/bin/train.py:67-312, with ``validate_length`` :35-64 and the
StandardScaler / mu-law transforms of :463-470) on a small synthetic corpus and stores what it yields.

    python tests/golden/make_slicer_golden.py        # writes tests/golden/slicer.npz

The reference's train.py imports soundfile, h5py (through wavenet_vocoder.utils) and torchvision, none
of which is installed here; they are only used to READ files and to compose two lambdas, so this script
puts minimal stand-ins into ``sys.modules`` that serve the corpus from memory.  Every array stored here
is produced by the reference's generator code itself (index arithmetic, validate_length, transforms);
the stand-ins do no arithmetic.  Build container only -- tests read the .npz.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")   # first: ``wavenet_vocoder`` must be the reference, not the alias package

MEM = {}   # file name -> {"wav": float32 array} or {"/path": array}

sf = types.ModuleType("soundfile")
sf.read = lambda name, dtype=np.float32: (MEM[name]["wav"].astype(dtype), 16000)
sys.modules["soundfile"] = sf


class _H5File(object):
    def __init__(self, name, mode="r"):
        self.d = MEM[name]

    def __contains__(self, k):
        return k in self.d

    def __getitem__(self, k):
        return _H5Data(self.d[k])

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _H5Data(object):
    def __init__(self, a):
        self.a = a
        self.shape = a.shape

    def __getitem__(self, k):
        return self.a.copy()

    @property
    def value(self):
        return self.a.copy()


h5 = types.ModuleType("h5py")
h5.File = _H5File
sys.modules["h5py"] = h5

tv = types.ModuleType("torchvision")
tvt = types.ModuleType("torchvision.transforms")


class Compose(object):   # torchvision.transforms.Compose: apply the callables in order
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


tvt.Compose = Compose
tv.transforms = tvt
sys.modules["torchvision"] = tv
sys.modules["torchvision.transforms"] = tvt

import os.path as _osp  # noqa: E402

_exists = _osp.exists
os.path.exists = lambda p: p in MEM or _exists(p)   # read_hdf5 checks the file first (utils.py:50-52)

from sklearn.preprocessing import StandardScaler  # noqa: E402
from wavenet_vocoder.bin import train as REF  # noqa: E402   (REFERENCE)
from wavenet_vocoder.nets import encode_mu_law  # noqa: E402  (REFERENCE)

import golden_util as GU  # noqa: E402  (corpus definition only)

assert REF.__file__.startswith("/root/reference"), REF.__file__


def main():
    utts, mean, scale = GU.slicer_corpus()
    wavs, feats = [], []
    for i, (wav, feat, code) in enumerate(utts):
        w, f = "mem/utt%d.wav" % i, "mem/utt%d.h5" % i
        MEM[w] = {"wav": wav.astype(np.float32) / 32768.0}
        MEM[f] = {"/melspc": feat, "/speaker_code": code}
        wavs.append(w)
        feats.append(f)
    out = {}
    for name, (bl, bs, up, spk, sdt, nb) in GU.SLICER_CASES.items():
        scaler = StandardScaler()                                   # train.py:463-465
        scaler.mean_ = mean.astype(sdt)
        scaler.scale_ = scale.astype(sdt)
        if spk:   # statistics cover the feature dimensions only; the recipes scale before appending nothing -- the
            # reference applies feat_transform to [features | speaker code], so the statistics carry the code columns too
            scaler.mean_ = np.concatenate([scaler.mean_, np.zeros(2, sdt)])
            scaler.scale_ = np.concatenate([scaler.scale_, np.ones(2, sdt)])
        wav_transform = Compose([lambda x: encode_mu_law(x, GU.SLICER_Q)])      # train.py:466-467
        feat_transform = Compose([lambda x, scaler=scaler: scaler.transform(x)])               # train.py:468-469
        gen = REF.train_generator(wavs, feats, receptive_field=GU.SLICER_RF, batch_length=bl, batch_size=bs,
                                  feature_type="melspc", wav_transform=wav_transform, feat_transform=feat_transform,
                                  shuffle=False, upsampling_factor=GU.SLICER_U, use_upsampling_layer=up,
                                  use_speaker_code=spk)
        for i in range(nb):
            (x, h), t = gen.next()
            out["%s/%d/x" % (name, i)] = x.numpy().astype(np.int16)
            out["%s/%d/h" % (name, i)] = h.numpy()
            out["%s/%d/t" % (name, i)] = t.numpy().astype(np.int16)
            assert h.dtype.is_floating_point and str(h.dtype) == "torch.float32"
        print("%s: %d batches, x %s h %s" % (name, nb, tuple(x.shape), tuple(h.shape)))
    # validate_length by itself (train.py:35-64): lengths only
    vl = []
    for nx, ny, U in [(1000, 12, 80), (900, 12, 80), (960, 12, 80), (10, 12, None), (15, 12, None), (961, 12, 80), (80, 2, 80)]:
        x, y = REF.validate_length(np.arange(nx), np.zeros((ny, 3)), U)
        vl.append((nx, ny, -1 if U is None else U, len(x), len(y)))
    out["validate_length"] = np.array(vl, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "slicer.npz"), **out)
    print("wrote slicer.npz (%d arrays)" % len(out))
    sys.stdout.flush()
    os._exit(0)   # the reference's producer threads never end


if __name__ == "__main__":
    main()
