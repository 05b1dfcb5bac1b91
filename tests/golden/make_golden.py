#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Generate golden fixtures by running the REFERENCE ITSELF (run in the build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Imports ``wavenet_vocoder.nets`` from /root/reference (wavenet.py needs only numpy+torch),
drives it with the 13-line training-step restatement of ``wavenet_vocoder/bin/train.py:527-540``
(the reference train.py itself exits without CUDA, train.py:516-525), and stores inputs seeds +
outputs.  /root/reference does not exist on the GPU box, so nothing else may import it; tests
only read the .npz files written here.

Inputs are generated from numpy RandomState streams (machine independent) by the helpers in
oracle/wavenet_oracle.py (``synthetic_batch``, ``random_params``) -- these produce INPUTS only;
every OUTPUT stored here comes from the reference's own code.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from wavenet_vocoder.nets import WaveNet, initialize, encode_mu_law, decode_mu_law  # noqa: E402  (REFERENCE)
from wavenet_vocoder.nets import UpSampling  # noqa: E402  (REFERENCE)

from oracle import wavenet_oracle as O  # noqa: E402  (input generators only)

CASES = {
    # name: (cfg tuple (Q,A,R,S,dd,dr,K,U), B, T, seed, param mode, adam wd)
    "tiny_k2_up": ((256, 5, 4, 4, 3, 2, 2, 10), 2, 60, 11, "random", 0.0),
    "tiny_k3_noup": ((256, 5, 8, 12, 3, 1, 3, 0), 1, 50, 12, "random", 0.01),
    "tiny_init": ((256, 6, 8, 8, 4, 1, 2, 5), 2, 40, 13, "init", 0.0),
    "r64_k2_up": ((64, 20, 64, 32, 4, 1, 2, 16), 1, 96, 14, "random", 0.0),
    "r64_k3_up": ((64, 8, 64, 32, 3, 1, 3, 8), 2, 64, 15, "random", 0.0),
}
ADAM_LR = 1e-3
ADAM_STEPS = 2


def run_case(name, spec):
    cfg_t, B, T, seed, mode, wd = spec
    cfg = O.OracleConfig(*cfg_t)
    torch.manual_seed(seed)
    model = WaveNet(*cfg_t)
    if mode == "init":
        model.apply(initialize)
        params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    else:
        params = O.random_params(cfg, seed)
        model.load_state_dict(params)
    model.train()
    x, h, t = O.synthetic_batch(cfg, B, T, seed + 1000)
    # loss start: the reference uses the receptive field; tiny cases have T > rf by construction
    rf = model.receptive_field
    assert rf == cfg.receptive_field and T > rf, (rf, T)

    optimizer = torch.optim.Adam(model.parameters(), lr=ADAM_LR, weight_decay=wd)   # train.py:457-460
    criterion = torch.nn.CrossEntropyLoss()                                         # train.py:461
    out = {}
    for step in range(ADAM_STEPS):
        batch_output = model(x, h)                                                   # train.py:533
        batch_loss = criterion(
            batch_output[:, rf:].contiguous().view(-1, cfg.n_quantize),
            t[:, rf:].contiguous().view(-1))                                         # train.py:534-536
        optimizer.zero_grad()
        batch_loss.backward()
        if step == 0:
            out["logits"] = batch_output.detach().numpy().copy()
            out["loss"] = np.array(batch_loss.item(), dtype=np.float64)
            for k, p in model.named_parameters():
                if p.grad is None:
                    out["gradnone/" + k] = np.array(1)
                else:
                    out["grad/" + k] = p.grad.detach().numpy().copy()
        optimizer.step()                                                             # train.py:539
        out["loss_step%d" % step] = np.array(batch_loss.item(), dtype=np.float64)
    for k, v in model.state_dict().items():
        out["after/" + k] = v.detach().numpy().copy()
    if mode == "init":
        for k, v in params.items():
            out["param/" + k] = v.numpy().copy()
    out["cfg"] = np.array(cfg_t, dtype=np.int64)
    out["B"] = np.array(B)
    out["T"] = np.array(T)
    out["seed"] = np.array(seed)
    out["mode"] = np.array(mode)
    out["wd"] = np.array(wd)
    out["rf"] = np.array(rf)
    out["adam_lr"] = np.array(ADAM_LR)
    out["adam_steps"] = np.array(ADAM_STEPS)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote %s: loss=%.6f rf=%d" % (name, out["loss"], rf))


def mulaw_vectors():
    xs = np.concatenate([np.linspace(-1, 1, 513), np.array([0.0, 1e-4, -1e-4, 0.5, -0.5])])
    enc = encode_mu_law(xs, 256)
    dec = decode_mu_law(np.arange(256), 256)
    enc16 = encode_mu_law(xs, 16)
    np.savez_compressed(os.path.join(HERE, "mulaw.npz"), x=xs, enc256=enc, dec256=dec, enc16=enc16)
    print("wrote mulaw: enc(0)=%d" % encode_mu_law(np.array([0.0]))[0])


def upsampling_vectors():
    torch.manual_seed(3)
    up = UpSampling(10)
    with torch.no_grad():
        up.conv.weight.copy_(torch.from_numpy(np.random.RandomState(3).standard_normal((1, 1, 1, 10))))
        up.conv.bias.fill_(0.25)
    h = torch.from_numpy(np.random.RandomState(4).standard_normal((2, 3, 7))).float()
    y = up(h)
    np.savez_compressed(os.path.join(HERE, "upsampling.npz"), h=h.numpy(),
                        w=up.conv.weight.detach().numpy(), b=up.conv.bias.detach().numpy(),
                        y=y.detach().numpy())
    print("wrote upsampling", tuple(y.shape))


def model_facts():
    """receptive_field / parameter count of the BASELINE config-2 model straight from the reference."""
    m = WaveNet(256, 80, 64, 256, 10, 3, 2, 80)
    n = sum(p.numel() for p in m.parameters())
    keys = list(m.state_dict().keys())
    np.savez_compressed(os.path.join(HERE, "cfg2_facts.npz"), rf=np.array(m.receptive_field),
                        n_params=np.array(n), keys=np.array(keys),
                        shapes=np.array([str(tuple(v.shape)) for v in m.state_dict().values()]))
    print("cfg2: rf=%d n_params=%d n_keys=%d" % (m.receptive_field, n, len(keys)))


# ---- generation (BASELINE config 5): reference generate / fast_generate / batch_fast_generate ----
DECODE_CASES = {
    # name: (cfg tuple, context length T0, n_samples list (batch), seed, param scale)
    "decode_tiny_k2_up": ((32, 5, 8, 12, 3, 2, 2, 4), 1, [24, 17], 21, 0.5),
    "decode_tiny_k3_noup": ((32, 5, 8, 12, 3, 1, 3, 0), 7, [20, 20, 9], 22, 0.5),
    "decode_r64_k2_up": ((256, 12, 64, 96, 4, 2, 2, 8), 1, [40, 33], 23, 0.2),
    "decode_r64_longctx": ((64, 6, 64, 32, 3, 1, 2, 0), 30, [25], 24, 0.3),
}


def decode_inputs(cfg, T0, n_list, seed):
    """Machine-independent inputs of a decode case (numpy RandomState)."""
    rs = np.random.RandomState(seed + 2000)
    B = len(n_list)
    x = torch.from_numpy(rs.randint(0, cfg.n_quantize, (B, T0))).long()
    U = cfg.upsampling_factor
    tot = max(n_list) + T0
    nf = (tot + U - 1) // U if U > 0 else tot
    h = torch.from_numpy(rs.standard_normal((B, cfg.n_aux, nf)).astype(np.float32))
    return x, h


def run_decode_case(name, spec):
    cfg_t, T0, n_list, seed, scale = spec
    cfg = O.OracleConfig(*cfg_t)
    params = O.random_params(cfg, seed, scale=scale)
    model = WaveNet(*cfg_t)
    model.load_state_dict(params)
    model.eval()
    x, h = decode_inputs(cfg, T0, n_list, seed)
    out = {"cfg": np.array(cfg_t), "T0": np.array(T0), "n_list": np.array(n_list), "seed": np.array(seed),
           "scale": np.array(scale)}
    margins = []
    with torch.no_grad():
        for b, n in enumerate(n_list):
            hb = h[b:b + 1]
            fast = model.fast_generate(x[b:b + 1], hb, n, mode="argmax")          # REFERENCE
            naive = model.generate(x[b:b + 1], hb, n, mode="argmax")              # REFERENCE
            out["fast/%d" % b] = np.asarray(fast)
            out["naive/%d" % b] = np.asarray(naive)
            # top-2 logit margin along the generated path (oracle logits; tokens are checked equal)
            tok, lg = O.fast_generate(cfg, params, x[b:b + 1], hb, n, return_logits=True)
            assert (tok == np.asarray(fast)).all(), name
            top2 = lg.topk(2, dim=1).values
            margins.append(float((top2[:, 0] - top2[:, 1]).min()))
            out["logits/%d" % b] = lg.numpy()
        batch = model.batch_fast_generate(x, h, list(n_list), mode="argmax")       # REFERENCE
        for i, a in enumerate(batch):
            out["batch/%d" % i] = np.asarray(a)
    out["min_margin"] = np.array(min(margins))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote %s: min top-2 margin %.3e, distinct tokens %d" % (
        name, min(margins), len(set(np.concatenate([out["fast/%d" % b] for b in range(len(n_list))]).tolist()))))


if __name__ == "__main__":
    for name, spec in CASES.items():
        run_case(name, spec)
    for name, spec in DECODE_CASES.items():
        run_decode_case(name, spec)
    mulaw_vectors()
    upsampling_vectors()
    model_facts()
