# -*- coding: utf-8 -*-
"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz).

Tolerances are the gates of SURVEY.md 8d: logits max-abs <= 1e-4, loss <= 1e-5,
gradients <= 1e-4 relative to each tensor's max-abs, weights after Adam <= 1e-6 abs
(oracle and reference run the same torch CPU ops, so the observed differences are ~0).
"""
import numpy as np
import pytest
import torch

from oracle import wavenet_oracle as O
from tests.golden_util import CASES, GOLDEN_DIR, GoldenCase, rel_to_max
import os


@pytest.mark.parametrize("name", CASES)
def test_forward_loss_grads_adam(name):
    g = GoldenCase(name)
    assert g.cfg.receptive_field == g.rf
    params = g.clone_params()
    opt = O.OracleAdam(lr=g.adam_lr, weight_decay=g.wd)
    for step in range(g.adam_steps):
        loss, logits, grads = O.train_step(g.cfg, params, opt, g.x, g.h, g.t)
        if step == 0:
            assert tuple(logits.shape) == (g.B, g.T, g.cfg.n_quantize)
            assert float((logits - g.logits).abs().max()) <= 1e-4
            assert abs(float(loss) - g.loss) <= 1e-5
            for k, ref in g.grads.items():
                if ref is None:
                    assert grads[k] is None, k
                else:
                    assert rel_to_max(grads[k], ref) <= 1e-4, k
        assert abs(float(loss) - float(g.z["loss_step%d" % step])) <= 1e-5
    for k, ref in g.after.items():
        assert float((params[k] - ref).abs().max()) <= 1e-6, k


def test_dead_last_res_1x1():
    g = GoldenCase("tiny_k2_up")
    L = len(g.cfg.dilations)
    assert g.grads["res_1x1.%d.weight" % (L - 1)] is None
    assert g.grads["res_1x1.%d.bias" % (L - 1)] is None
    assert g.grads["res_1x1.%d.weight" % (L - 2)] is not None


def test_mulaw_golden():
    z = np.load(os.path.join(GOLDEN_DIR, "mulaw.npz"))
    np.testing.assert_array_equal(O.encode_mu_law(z["x"], 256), z["enc256"])
    np.testing.assert_array_equal(O.encode_mu_law(z["x"], 16), z["enc16"])
    np.testing.assert_allclose(O.decode_mu_law(np.arange(256), 256), z["dec256"], rtol=0, atol=0)
    assert O.encode_mu_law(np.array([0.0]))[0] == 128


def test_upsampling_golden():
    z = np.load(os.path.join(GOLDEN_DIR, "upsampling.npz"))
    y = O.upsampling(torch.from_numpy(z["h"]), torch.from_numpy(z["w"]), torch.from_numpy(z["b"]))
    assert float((y - torch.from_numpy(z["y"])).abs().max()) <= 1e-6
    # closed form used by the HIP path: out[b,c,f*U+j] = h[b,c,f]*w[j] + bias
    h, w, b = z["h"], z["w"].reshape(-1), float(z["b"][0])
    closed = (h[:, :, :, None] * w[None, None, None, :] + b).reshape(h.shape[0], h.shape[1], -1)
    np.testing.assert_allclose(closed, z["y"], atol=1e-6)


def test_cfg2_facts():
    z = np.load(os.path.join(GOLDEN_DIR, "cfg2_facts.npz"))
    cfg = O.OracleConfig(256, 80, 64, 256, 10, 3, 2, 80)
    assert cfg.receptive_field == int(z["rf"]) == 3070
    shapes = O.param_shapes(cfg)
    assert list(shapes.keys()) == [str(k) for k in z["keys"]]
    assert [str(tuple(s)) for s in shapes.values()] == [str(s) for s in z["shapes"]]
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(z["n_params"]) == 1594897
    geo = O.batch_geometry(cfg.receptive_field, 20000, 80)
    assert geo == {"batch_length": 19970, "frames": 288, "T": 23040, "loss_positions": 19970}


# ---- generation (BASELINE config 5): oracle restatement vs the reference's own outputs ----------
from tests.decode_common import DECODE_CASES, DecodeCase  # noqa: E402


@pytest.mark.parametrize("name", DECODE_CASES)
def test_oracle_generation_vs_reference(name):
    g = DecodeCase(name)
    for b, n in enumerate(g.n_list):
        xb, hb = g.x[b:b + 1], g.h[b:b + 1]
        fast, lg = O.fast_generate(g.cfg, g.params, xb, hb, n, return_logits=True)
        assert (fast == g.fast[b]).all()
        assert float((lg - g.logits[b]).abs().max()) <= 1e-5
        if b == 0 or not name.startswith("decode_r64"):   # the naive generator runs a full forward per sample: one utterance of the wide cases
            assert (O.generate(g.cfg, g.params, xb, hb, n) == g.naive[b]).all()
    batch = O.batch_fast_generate(g.cfg, g.params, g.x, g.h, g.n_list)
    for a, r in zip(batch, g.batch):
        assert (a == r).all()


def test_restatement_equals_the_reference_module_copy():
    """oracle/_ref/wavenet.py (the reference's own model file, copied at build time by oracle/build_ref.py; absent from the
    repository's history) under the reference's training loop == the restatement, bit for bit, incl. the Adam step."""
    import pytest
    from oracle import ref_step as RS
    if not RS.available():
        pytest.skip("oracle/_ref/wavenet.py not built (needs /root/reference at build time)")
    import torch
    from oracle import wavenet_oracle as O
    old_threads = torch.get_num_threads()
    torch.set_num_threads(1)   # oneDNN's thread partition changes the summation order: one thread makes both runs the same program
    try:
        _check_restatement_vs_reference(RS, O, torch)
    finally:
        torch.set_num_threads(old_threads)


def _check_restatement_vs_reference(RS, O, torch):
    for cfg_t, B, T in [((256, 5, 4, 4, 3, 2, 2, 10), 2, 60), ((64, 8, 64, 32, 3, 1, 3, 8), 2, 64), ((32, 6, 8, 8, 4, 1, 2, 0), 1, 40)]:
        cfg = O.OracleConfig(*cfg_t)
        p = O.random_params(cfg, 5)
        x, h, t = O.synthetic_batch(cfg, B, T, 6)
        tr = RS.ReferenceTrainer(cfg_t, state=p, lr=1e-3, weight_decay=0.01)
        pp = {k: v.clone() for k, v in p.items()}
        opt = O.OracleAdam(lr=1e-3, weight_decay=0.01)
        for _ in range(2):
            l_ref, out_ref = tr.step(x, h, t)
            l, lg, _g = O.train_step(cfg, pp, opt, x, h, t)
            # same torch ops in the same order: bit-identical here; the tolerance only covers oneDNN picking another kernel
            assert abs(l_ref - float(l)) <= 1e-6 and float((out_ref.detach() - lg).abs().max()) <= 2e-6
        for k, v in tr.model.state_dict().items():
            assert float((v - pp[k]).abs().max()) <= 1e-2 * 1e-3, k   # 1e-2 * lr: the Adam gate of SURVEY 8d
