#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Waveform generation CLI with the command line, inputs and outputs of the reference's
``wavenet_vocoder/bin/decode.py`` (flags :181-203; ``expdir/model.conf`` + ``stats`` lookup
:224-231; one ``<feat id>.wav`` (PCM 16) per feature file in ``--outdir`` :316-327), driving the
HIP decode kernel through ``WaveNet.fast_generate`` / ``batch_fast_generate``.

What differs from the reference:
  * ``--n_gpus N`` starts one process per GPU, each with its share of the file list (the reference
    does the same with ``mp.Process``, decode.py:262,330-338);
  * every utterance of a batch is its own workgroup on the GPU, so batching costs no per-sample speed:
    ``--batch_size`` only bounds how many utterances are resident at once;
  * soundfile / sklearn / torchvision are not needed (scipy writes the wav, the scaler is two numpy ops).
"""
import argparse
import logging
import math
import os
import sys

import numpy as np
import torch

from pytorchwavenetvocoder_amd.nets import WaveNet, decode_mu_law, encode_mu_law
from pytorchwavenetvocoder_amd.utils import extend_time, find_files, make_feat_transform, read_hdf5, read_txt, shape_hdf5

# (flag, default, type, help) -- reference decode.py:181-203
_FLAGS = [
    ("feats", argparse.SUPPRESS, str, "list or directory of aux feat files"),
    ("checkpoint", argparse.SUPPRESS, str, "model file"),
    ("outdir", argparse.SUPPRESS, str, "directory to save generated samples"),
    ("stats", None, str, "hdf5 file including statistics"),
    ("config", None, str, "configure file"),
    ("fs", 16000, int, "sampling rate"),
    ("batch_size", 32, int, "number of batch size in decoding"),
    ("n_gpus", 1, int, "number of gpus"),
    ("intervals", 1000, int, "log interval"),
    ("seed", 1, int, "seed number"),
    ("verbose", 1, int, "log level"),
]


def get_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for name, default, typ, text in _FLAGS:
        if default is argparse.SUPPRESS:
            parser.add_argument("--" + name, required=True, type=typ, help=text)
        else:
            parser.add_argument("--" + name, default=default, type=typ, help=text)
    return parser


def pad_list(batch_list, pad_value=0.0):
    """(T_i, C) arrays -> (B, T_max, C), padded with ``pad_value`` (reference decode.py:31-49)."""
    tmax = max(b.shape[0] for b in batch_list)
    out = np.full((len(batch_list), tmax, batch_list[0].shape[-1]), pad_value, dtype=np.float64)
    for i, b in enumerate(batch_list):
        out[i, :b.shape[0]] = b
    return out


def _load_utterance(featfile, feature_type, wav_transform, feat_transform, upsampling_factor, use_upsampling_layer,
                    use_speaker_code):
    """Seed token + conditioning features of one file (reference decode.py:81-95 / 133-147)."""
    x = np.zeros((1))
    h = read_hdf5(featfile, "/" + feature_type)
    if not use_upsampling_layer:
        h = extend_time(h, upsampling_factor)
    if use_speaker_code:
        sc = read_hdf5(featfile, "/speaker_code")
        h = np.concatenate([h, np.tile(sc, [h.shape[0], 1])], axis=1)
    if wav_transform is not None:
        x = wav_transform(x)
    if feat_transform is not None:
        h = feat_transform(h)
    n_samples = (h.shape[0] if not use_upsampling_layer else h.shape[0] * upsampling_factor) - 1
    return x, h, n_samples, os.path.basename(featfile).replace(".h5", "")


def decode_generator(feat_list, batch_size=32, feature_type="world", wav_transform=None, feat_transform=None,
                     upsampling_factor=80, use_upsampling_layer=True, use_speaker_code=False, device=None):
    """Yields ``feat_id, (x, h, n_samples)`` for ``batch_size == 1`` and
    ``feat_ids, (batch_x, batch_h, n_samples_list)`` otherwise, batches sorted by length
    (reference decode.py:52-175)."""
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    args = (feature_type, wav_transform, feat_transform, upsampling_factor, use_upsampling_layer, use_speaker_code)
    if batch_size == 1:
        for featfile in feat_list:
            x, h, n_samples, feat_id = _load_utterance(featfile, *args)
            x = torch.from_numpy(np.asarray(x)).long().unsqueeze(0).to(device)               # 1 x 1
            h = torch.from_numpy(np.asarray(h)).float().transpose(0, 1).unsqueeze(0).to(device)  # 1 x C x T
            yield feat_id, (x, h, n_samples)
        return
    lengths = [shape_hdf5(f, "/" + feature_type)[0] for f in feat_list]
    feat_list = [feat_list[i] for i in np.argsort(lengths, kind="stable")]
    n_batch = math.ceil(len(feat_list) / batch_size)
    for chunk in np.array_split(np.array(feat_list, dtype=object), n_batch):
        xs, hs, n_list, ids = [], [], [], []
        for featfile in chunk.tolist():
            x, h, n_samples, feat_id = _load_utterance(featfile, *args)
            xs.append(x)
            hs.append(h)
            n_list.append(n_samples)
            ids.append(feat_id)
        batch_x = torch.from_numpy(np.stack(xs, axis=0)).long().to(device)
        batch_h = torch.from_numpy(pad_list(hs)).float().transpose(1, 2).contiguous().to(device)
        yield ids, (batch_x, batch_h, n_list)


def write_wav(path, wav, fs):
    """16-bit PCM like ``sf.write(path, wav, fs, "PCM_16")`` (reference decode.py:318,326)."""
    try:
        import soundfile as sf
        sf.write(path, wav, fs, "PCM_16")
    except ImportError:
        from scipy.io import wavfile
        wavfile.write(path, fs, np.clip(np.round(wav * 32768.0), -32768, 32767).astype(np.int16))


def build_model(config):
    upsampling_factor = config.upsampling_factor if config.use_upsampling_layer else 0
    return WaveNet(n_quantize=config.n_quantize, n_aux=config.n_aux, n_resch=config.n_resch,
                   n_skipch=config.n_skipch, dilation_depth=config.dilation_depth,
                   dilation_repeat=config.dilation_repeat, kernel_size=config.kernel_size,
                   upsampling_factor=upsampling_factor, n_mixture=getattr(config, "n_mixture", 0),
                   log_scale_min=getattr(config, "log_scale_min", -7.0))


def _worker(gpu, feat_list, args, config, mean, scale):
    with torch.no_grad():
        _decode_files(gpu, feat_list, args, config, mean, scale)


def _decode_files(gpu, feat_list, args, config, mean, scale):
    torch.cuda.set_device(gpu)
    device = torch.device("cuda", gpu)
    model = build_model(config)
    model.load_state_dict(torch.load(args.checkpoint, map_location="cpu", weights_only=False)["model"])
    model.eval()
    model.to(device)
    generator = decode_generator(
        feat_list, batch_size=args.batch_size, feature_type=config.feature_type,
        wav_transform=lambda x: encode_mu_law(x, config.n_quantize),
        # the same float32-in-place scaling as training and as the reference's scaler.transform
        feat_transform=make_feat_transform(mean, scale),
        upsampling_factor=config.upsampling_factor, use_upsampling_layer=config.use_upsampling_layer,
        use_speaker_code=config.use_speaker_code, device=device)
    if args.batch_size > 1:
        for feat_ids, (batch_x, batch_h, n_samples_list) in generator:
            logging.info("decoding start")
            samples_list = model.batch_fast_generate(batch_x, batch_h, n_samples_list, args.intervals)
            for feat_id, samples in zip(feat_ids, samples_list):
                write_wav(args.outdir + "/" + feat_id + ".wav", decode_mu_law(samples, config.n_quantize), args.fs)
                logging.info("wrote %s.wav in %s." % (feat_id, args.outdir))
    else:
        for feat_id, (x, h, n_samples) in generator:
            logging.info("decoding %s (length = %d)" % (feat_id, n_samples))
            samples = model.fast_generate(x, h, n_samples, args.intervals)
            write_wav(args.outdir + "/" + feat_id + ".wav", decode_mu_law(samples, config.n_quantize), args.fs)
            logging.info("wrote %s.wav in %s." % (feat_id, args.outdir))


def main(argv=None):
    args = get_parser().parse_args(argv)
    level = logging.INFO if args.verbose > 0 else logging.WARNING
    logging.basicConfig(level=level, format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s",
                        datefmt="%m/%d/%Y %I:%M:%S")
    if args.verbose <= 0:
        logging.warning("logging is disabled.")
    for key, value in vars(args).items():
        logging.info("%s = %s" % (key, str(value)))
    if args.stats is None:
        args.stats = os.path.dirname(args.checkpoint) + "/stats.h5"
    if args.config is None:
        args.config = os.path.dirname(args.checkpoint) + "/model.conf"
    if not os.path.exists(args.stats):
        raise FileNotFoundError("statistics file is missing (%s)." % (args.stats))
    if not os.path.exists(args.config):
        raise FileNotFoundError("config file is missing (%s)." % (args.config))
    if not os.path.exists(args.outdir):
        os.makedirs(args.outdir)
    os.environ["PYTHONHASHSEED"] = str(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    config = torch.load(args.config, weights_only=False)
    if os.path.isdir(args.feats):
        feat_list = sorted(find_files(args.feats, "*.h5"))
    elif os.path.isfile(args.feats):
        feat_list = read_txt(args.feats)
    else:
        logging.error("--feats should be directory or list.")
        sys.exit(1)
    if not torch.cuda.is_available():
        logging.error("decode.py needs an MI355X (there is no CPU fallback).")
        sys.exit(1)
    mean = read_hdf5(args.stats, "/" + config.feature_type + "/mean")
    scale = read_hdf5(args.stats, "/" + config.feature_type + "/scale")
    shares = [s.tolist() for s in np.array_split(np.array(feat_list, dtype=object), args.n_gpus)]
    if args.n_gpus == 1:
        _worker(0, shares[0], args, config, mean, scale)
        return
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(gpu, share, args, config, mean, scale)) for gpu, share in enumerate(shares)]
    for p in procs:
        p.start()
    for p in procs:
        p.join()
    if any(p.exitcode != 0 for p in procs):
        sys.exit(1)


if __name__ == "__main__":
    main()
