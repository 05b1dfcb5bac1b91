# -*- coding: utf-8 -*-
"""Checks of the mixture-of-logistics output head (BASELINE configs[3]).  The reference has no such head:
the checker is this repo's own restatement of the published formulas in oracle/ ("parity unpinned by the
reference"), evaluated in fp32 like the kernel and, for the conditioning of the formula itself, in fp64."""
import numpy as np
import torch

from oracle import wavenet_oracle as O
from pytorchwavenetvocoder_amd.engine import WaveNetEngine
from pytorchwavenetvocoder_amd.nets import WaveNet, encode_mu_law
from tests import parity_common as PC

NM = 4
CFG = (32, 4, 8, 12, 3, 2, 2, 4)


def check_mol_loss_op(lib, device):
    eng = WaveNetEngine(*CFG, device=device, library=lib, out_channels=3 * 5)
    rs = np.random.RandomState(0)
    B, T, nm = 2, 300, 5
    out = torch.from_numpy(rs.standard_normal((B, 3 * nm, T)).astype(np.float32))
    out[:, 2 * nm:] = out[:, 2 * nm:] * 2 - 4      # log-scales around -4, some below the -7 clamp
    out[0, 2 * nm:, 20:30] = -30.0                 # clamped: zero gradient to the raw log-scale
    y = torch.from_numpy(rs.uniform(-1, 1, (B, T)).astype(np.float32))
    y[0, 7], y[1, 9], y[0, 11] = -1.0, 1.0, 0.9995  # both edge branches
    out[1, nm:2 * nm, 40] = y[1, 40] + 0.5          # far-away narrow components -> "pdf at the bin centre" branch
    out[1, 2 * nm:, 40] = -6.9

    def orc(dt):
        ri = out.transpose(1, 2).clone().to(dt).requires_grad_(True)
        loss = O.mol_nll(ri, y.to(dt), start=5)
        loss.backward()
        return float(loss.detach()), ri.grad.float()

    l32, g32 = orc(torch.float32)
    l64, g64 = orc(torch.float64)
    loss, dout = eng.mol_loss(out.to(device), y.to(device), t_start=5)
    dk = dout.transpose(1, 2).cpu()
    den = float(g64.abs().max())
    assert abs(float(loss.cpu()) - l64) <= 1e-5 * abs(l64) and abs(l32 - l64) <= 1e-4 * abs(l64)
    # the kernel takes a bin's mass without subtracting two sigmoids one bin apart (wn_elem.hip: mol_component), so it sits on
    # the fp64 evaluation of the published formula; the fp32 evaluation of the same formula is the one that is ~1e-3 off
    e32 = float((g32 - g64).abs().max()) / den
    assert float((dk - g64).abs().max()) / den <= 2e-5
    assert float((dk - g32).abs().max()) / den <= e32 + 2e-5
    assert float(dk[0, 20:30, 2 * nm:].abs().max()) == 0.0      # clamped log-scales get no gradient
    assert float(dk[:, :5].abs().max()) == 0.0                  # positions before t_start


def _kink_free_instance(cfg, B, T):
    for seed in range(9, 60):
        params = O.random_params(cfg, seed, scale=0.3)
        x, h, t = O.synthetic_batch(cfg, B, T, seed + 1)
        if O.relu_kink_margin(cfg, params, x, h) >= PC.KINK_MARGIN:
            return params, x, h, seed
    raise RuntimeError("no kink-free instance")


def check_mol_training_step(lib, device):
    cfg = O.OracleConfig(*CFG, out_channels=3 * NM)
    B, T = 2, 48
    params, x, h, seed = _kink_free_instance(cfg, B, T)
    y = torch.from_numpy(np.random.RandomState(seed).uniform(-1, 1, (B, T)).astype(np.float32))
    model = WaveNet(*CFG, n_mixture=NM, _library=lib)
    assert list(model.state_dict().keys()) == list(O.param_shapes(cfg).keys())
    model.load_state_dict(params)
    model.to(device)
    def oracle(dt):
        preq = {k: v.clone().to(dt).requires_grad_(True) for k, v in params.items()}
        ref = O.mol_nll(O.forward(cfg, preq, x, h.to(dt)), y.to(dt), start=cfg.receptive_field)
        ref.backward()
        return float(ref.detach()), {k: (v.grad.float() if v.grad is not None else None) for k, v in preq.items()}

    l32, g32 = oracle(torch.float32)
    l64, g64 = oracle(torch.float64)
    loss = model.mol_loss_and_backward(x.to(device), h.to(device), y.to(device))
    assert abs(float(loss.cpu()) - l64) <= 1e-4 * abs(l64)
    # With 65536 classes the published formula (differences of sigmoids one bin apart) carries ~1e-3 relative
    # noise in fp32: the kernel (which avoids the subtraction) must be closer to the fp64 evaluation than that.
    for k, p in model.named_parameters():
        if p.grad is None:
            assert g64[k] is None or float(g64[k].abs().max()) == 0.0, k
            continue
        ek, eo = PC.rel_to_max(p.grad.cpu(), g64[k]), PC.rel_to_max(g32[k], g64[k])
        assert ek <= 1e-4 + eo, (k, ek, eo)


def check_mol_generation(lib, device):
    """Tokens / values drawn by the decode path == the oracle sampler fed the same uniforms on the oracle's own
    network outputs (teacher forced on the generated tokens)."""
    cfg = O.OracleConfig(*CFG, out_channels=3 * NM)
    params = O.random_params(cfg, 9, scale=0.3)
    model = WaveNet(*CFG, n_mixture=NM, _library=lib)
    model.load_state_dict(params)
    model.to(device)
    xs = torch.tensor([[3, 17], [30, 1]]).long()
    hs = torch.from_numpy(np.random.RandomState(12).standard_normal((2, 4, 8)).astype(np.float32))
    n = 14
    assert model.engine.decode_supported()   # the persistent kernel covers this size; both paths are checked
    for layered in (False, True):
        _check_mol_generation_path(model, cfg, params, xs, hs, n, device, layered)
    # module API: fast_generate draws from the mixture, values in range
    out = model.fast_generate(xs[:1].to(device), hs[:1].to(device), 10, mode="sampling")
    assert out.shape == (10,) and out.min() >= 0 and out.max() < cfg.n_quantize


def _check_mol_generation_path(model, cfg, params, xs, hs, n, device, layered):
    toks, outs = model.engine.decode(xs.to(device), hs.to(device), [n, n - 5], mode="mol", return_logits=True, layered=layered)
    u = model.engine.last_uniforms.cpu()
    wave = [w.cpu() for w in model.engine.last_wave]
    rf = cfg.receptive_field
    n_pad = rf - xs.size(1)
    cfg0 = O.OracleConfig(*CFG[:7], 0, out_channels=3 * NM)
    hup = O.upsampling(hs, params["upsampling.conv.weight"], params["upsampling.conv.bias"])
    hup = torch.nn.functional.pad(hup, (n_pad, 0), "replicate")
    for b, nb in enumerate([n, n - 5]):
        full = torch.cat([torch.full((1, n_pad), cfg.n_quantize // 2), xs[b:b + 1], toks[b].cpu()[None]], 1)
        ref_out = O.forward(cfg0, params, full, hup[b:b + 1, :, :full.size(1)])[0]
        assert float((outs[b].cpu() - ref_out[rf - 1:rf - 1 + nb]).abs().max()) <= 1e-4
        for i in range(nb):
            pos = rf - 1 + i
            xo = O.mol_sample(ref_out[pos], u[b, pos + 1])
            assert abs(xo - float(wave[b][i])) <= 1e-4
            if abs(abs(xo) - 1.0) > 1e-3:   # away from the clip, the mu-law bin of the value
                tok = int(encode_mu_law(np.array([float(wave[b][i])]), cfg.n_quantize)[0])
                assert abs(tok - int(toks[b][i])) <= 1     # fp32 vs fp64 mu-law at a bin edge
