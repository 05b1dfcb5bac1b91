# -*- coding: utf-8 -*-
"""Pins of the any-size persistent decode launches (csrc/wn_dlp.hip, wn_dlpf.hip, wn_dlpm.hip; reference
wavenet.py:355-385, 397-511, 538-549) at the recipes' own size on the GPU:

* the kernel_size 3 class of wn_dlpf.hip (``k_dlpf<64,16>``: egs/ljspeech/sd-melspc/run.sh:29, n_resch 512) against the
  queue algorithm (oracle) and the layer-wise launches;
* LONG horizons: more generated steps than the longest dilation ring holds (kernel_size 2: > 1024, kernel_size 3: > 2048), every
  per-step logit row against the TRAINING forward of the same library on the same tokens (two very different sets of kernels;
  the training forward itself is pinned to the oracle at full size, tests/test_gpu_fullsize.py) -- one utterance (wn_dlp.hip),
  2 / 17 / 48 / 64 (wn_dlpf.hip, 1 - 4 column blocks), 70 (two groups), and the context walked by the persistent launch itself;
* 300 steps of every batch class against the layer-wise launches (the in-suite form of tools/decode_equivalence_soak.py);
* the residency check in front of the launch (occupancy x CUs) and the path it chooses on a device that is too small.
"""
import numpy as np
import pytest
import torch

from oracle import wavenet_oracle as O
from tests import parity_common as PC

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
RECIPE_K2 = (256, 80, 512, 256, 10, 3, 2, 80)     # egs/arctic/sd/run.sh:46-52
RECIPE_K3 = (256, 80, 512, 256, 10, 3, 3, 256)    # egs/ljspeech/sd-melspc/run.sh:29 (kernel_size 3, hop 256)
_MODELS = {}


def _model(cfg_t, seed=41, scale=0.02):
    from pytorchwavenetvocoder_amd.nets import WaveNet
    key = (cfg_t, seed, scale)
    if key not in _MODELS:
        _MODELS.clear()   # one recipe-size model (and its workspaces) alive at a time
        cfg = O.OracleConfig(*cfg_t)
        params = O.random_params(cfg, seed, scale=scale)
        model = WaveNet(*cfg_t)
        model.load_state_dict(params)
        model.to(DEV)
        _MODELS[key] = (cfg, params, model)
    return _MODELS[key]


def _safe(ref_rows):
    top2 = ref_rows.topk(2, dim=1).values
    return (top2[:, 0] - top2[:, 1]) > 1e-3


def test_flag_hand_off_kernel_size_3_class_at_the_recipes_own_size():
    """``k_dlpf<64,16>`` on the hardware: 2 utterances (one column block) and 18 ragged ones (two blocks, 128 workgroups): every
    utterance against the layer-wise launches, utterances 0 / 1 resp. 0 / 17 against the oracle's fast_generate; the launch log
    shows the persistent kernel and no layer-wise launch."""
    cfg, params, model = _model(RECIPE_K3)
    assert not model.engine.decode_supported()
    for B, n, vs_oracle in ((2, 9, (0, 1)), (18, 8, (0, 17))):
        rs = np.random.RandomState(50 + B)
        x = torch.from_numpy(rs.randint(0, 256, (B, 4))).long()
        h = torch.from_numpy(rs.standard_normal((B, 80, 2)).astype(np.float32))
        ns = [n - (b % 3) for b in range(B)]
        assert model.engine.decode_residency(B)[0]
        out = {}
        log = PC.launch_log(model.engine.lib, lambda: out.update(p=model.engine.decode(x.to(DEV), h.to(DEV), ns, return_logits=True, chunk=5)))
        assert log.get("dlpf_steps", 0) >= 2 and "dl_dilated" not in log and "dlpm_steps" not in log and "dlp_steps" not in log, log
        tp, lp = out["p"]
        tl, ll = model.engine.decode(x.to(DEV), h.to(DEV), ns, return_logits=True, layered="launches")
        worst = 0.0
        for i in range(B):
            worst = max(worst, float((lp[i] - ll[i]).abs().max()))
            assert float((lp[i] - ll[i]).abs().max()) <= 1e-4, (B, i)
            safe = _safe(ll[i]).cpu().numpy()
            assert (tp[i].cpu().numpy()[safe] == tl[i].cpu().numpy()[safe]).all(), (B, i)
        for i in vs_oracle:
            rt, rl = O.fast_generate(cfg, params, x[i:i + 1], h[i:i + 1], ns[i], return_logits=True)
            e = float((lp[i].cpu() - rl).abs().max())
            assert e <= 1e-4, (B, i, e)
            safe = _safe(rl).numpy()
            assert (tp[i].cpu().numpy()[safe] == np.asarray(rt)[safe]).all(), (B, i)
            print("k_dlpf<64,16> B=%d utterance %d vs oracle: logits err %.3g (max |logit| %.3g); vs launches worst %.3g"
                  % (B, i, e, float(rl.abs().max()), worst))


def _teacher_forced(cfg_t, B, T0, n, prefill, chunk=4096):
    """decode B utterances from a context of T0 > receptive-field tokens; every generated step's logits against the training
    forward on [context | generated tokens]; returns (worst logit error, max |logit|, launch log)."""
    cfg, params, model = _model(cfg_t)
    eng = model.engine
    U = cfg.upsampling_factor
    assert T0 > eng.receptive_field and (T0 + n) % U == 0
    rs = np.random.RandomState(60 + B)
    x = torch.from_numpy(rs.randint(0, cfg.n_quantize, (B, T0))).long().to(DEV)
    h = torch.from_numpy(rs.standard_normal((B, cfg.n_aux, (T0 + n) // U)).astype(np.float32)).to(DEV)
    ns = [n - 16 * (b % 3) for b in range(B)]
    out = {}
    log = PC.launch_log(eng.lib, lambda: out.update(p=eng.decode(x, h, ns, return_logits=True, prefill=prefill, chunk=chunk)))
    toks, lg = out["p"]
    worst, scale, n_safe, n_tot = 0.0, 0.0, 0, 0
    for b0 in range(0, B, 8):   # the training forward in groups of 8 sequences (bounded workspace)
        b1 = min(b0 + 8, B)
        full = torch.zeros((b1 - b0, T0 + n), dtype=torch.int64, device=DEV)
        full[:, :T0] = x[b0:b1]
        for b in range(b0, b1):
            full[b - b0, T0:T0 + ns[b]] = toks[b]
        logits = eng.forward(full, h[b0:b1].contiguous())           # (b, Q, T0 + n)
        for b in range(b0, b1):
            ref = logits[b - b0, :, T0 - 1:T0 - 1 + ns[b]].transpose(0, 1)   # generated sample i was chosen from position T0-1+i
            e = float((lg[b] - ref).abs().max())
            worst, scale = max(worst, e), max(scale, float(ref.abs().max()))
            assert e <= 1e-4, (cfg_t, B, b, prefill, e)
            safe = _safe(ref)
            assert bool((ref.argmax(1)[safe] == toks[b][safe]).all()), (cfg_t, B, b, prefill)
            n_safe += int(safe.sum())
            n_tot += ns[b]
    assert n_safe > n_tot // 2
    return worst, scale, log


@pytest.mark.parametrize("B", [1, 2, 17, 48, 64, 70])
def test_long_horizon_teacher_forced_kernel_size_2(B):
    """n_resch 512, kernel_size 2 (rings of up to 512 positions): 1120 generated steps from a 3200-token context -- every ring
    wraps twice inside ONE decode call and across its chunk boundaries (chunk 500: three launches)."""
    worst, scale, log = _teacher_forced(RECIPE_K2, B, 3200, 1120, "parallel", chunk=500)
    kern = "dlp_steps" if B == 1 else "dlpf_steps"
    assert log.get(kern, 0) >= 3 and "dl_dilated" not in log, log   # (64 utterances: four column blocks = 256 workgroups in ONE launch)
    print("LONG HORIZON K=2 B=%d: 1120 steps, per-step logits vs training forward worst %.3g (max |logit| %.3g)" % (B, worst, scale))


@pytest.mark.parametrize("B", [2, 17])
def test_long_horizon_teacher_forced_kernel_size_3(B):
    """n_resch 512, kernel_size 3 (``k_dlpf<64,16>``; rings of up to 1024 positions, receptive field 6139): 2304 generated
    steps from a 6400-token context."""
    worst, scale, log = _teacher_forced(RECIPE_K3, B, 6400, 2304, "parallel", chunk=1000)
    assert log.get("dlpf_steps", 0) >= 3 and "dl_dilated" not in log, log
    print("LONG HORIZON K=3 B=%d: 2304 steps, per-step logits vs training forward worst %.3g (max |logit| %.3g)" % (B, worst, scale))


def test_context_walked_by_the_persistent_launch_itself():
    """prefill="walk": the persistent launch steps through the 3200 context positions teacher forced (its own queue writes
    build every ring from zero) before it generates -- 3 utterances through wn_dlpf.hip, one through wn_dlp.hip."""
    for B in (3, 1):
        worst, scale, log = _teacher_forced(RECIPE_K2, B, 3200, 160, "walk", chunk=1500)
        assert log.get("dlp_steps" if B == 1 else "dlpf_steps", 0) >= 3 and "dl_dilated" not in log, log
        print("WALKED CONTEXT K=2 B=%d: logits vs training forward worst %.3g (max |logit| %.3g)" % (B, worst, scale))


def test_300_steps_of_every_batch_class_against_the_layer_wise_launches():
    """The in-suite form of tools/decode_equivalence_soak.py: 1 / 3 / 19 / 48 utterances, 300 generated steps from a short
    context (left padding, replicated first aux column): every logit row within 1e-4 of the layer-wise launches, tokens
    equal wherever the launches' argmax is not a near-tie; the granule hand-off (wn_dlpm.hip) on 19 utterances as well."""
    cfg, params, model = _model(RECIPE_K2)
    eng = model.engine
    for B, lay in ((1, None), (3, None), (19, None), (19, "granules"), (48, None)):
        rs = np.random.RandomState(70 + B)
        x = torch.from_numpy(rs.randint(0, 256, (B, 4))).long().to(DEV)
        h = torch.from_numpy(rs.standard_normal((B, 80, 5)).astype(np.float32)).to(DEV)
        ns = [300 - 7 * (b % 4) for b in range(B)]
        tp, lp = eng.decode(x, h, ns, return_logits=True, layered=lay)
        tl, ll = eng.decode(x, h, ns, return_logits=True, layered="launches")
        worst, differ = 0.0, 0
        for b in range(B):
            e = float((lp[b] - ll[b]).abs().max())
            worst = max(worst, e)
            assert e <= 1e-4, (B, lay, b, e)
            safe = _safe(ll[b])
            assert bool((tp[b][safe] == tl[b][safe]).all()), (B, lay, b)
            differ += int((tp[b] != tl[b]).sum())
        print("SOAK B=%d %s: 300 steps, logits vs launches worst %.3g, %d tokens differ (near-ties)" % (B, lay or "flags", worst, differ))


def test_residency_is_asked_of_the_device_before_the_launch(monkeypatch):
    """wn_decode_layered_residency: the grid the persistent launch needs and what the device keeps resident (occupancy x CUs);
    on an MI355X (256 CUs, one such workgroup per CU) the largest launch (192 workgroups) fits.  With the capacity forced down
    (test knob WN_COOP_CAPACITY: a partitioned / smaller part) the library chooses layer-wise launches itself -- same tokens --
    and the engine's groups shrink to what fits."""
    cfg, params, model = _model(RECIPE_K2)
    eng = model.engine
    ok, wg, cap = eng.decode_residency(48)
    assert ok and wg == 192 and cap >= 192, (ok, wg, cap)
    ok1, wg1, cap1 = eng.decode_residency(1)
    assert ok1 and wg1 == 128 and cap1 >= 128
    print("residency: 48 utterances need %d workgroups, the device keeps %d; one utterance %d of %d" % (wg, cap, wg1, cap1))
    rs = np.random.RandomState(7)
    B = 17
    x = torch.from_numpy(rs.randint(0, 256, (B, 4))).long().to(DEV)
    h = torch.from_numpy(rs.standard_normal((B, 80, 2)).astype(np.float32)).to(DEV)
    ns = [6] * B
    tp, lp = eng.decode(x, h, ns, return_logits=True)
    monkeypatch.setenv("WN_COOP_CAPACITY", "100")
    assert eng.decode_residency(17) == (False, 128, 100) and eng.decode_residency(16) == (True, 64, 100)
    assert eng._persistent_groups(30, None, "argmax") == [(0, 16), (16, 30)]
    monkeypatch.setenv("WN_COOP_CAPACITY", "50")   # not even one column block's 64 workgroups
    assert eng.decode_residency(16) == (False, 64, 50) and eng._persistent_groups(17, None, "argmax") is None
    out = {}
    log = PC.launch_log(eng.lib, lambda: out.update(c=eng.decode(x, h, ns, return_logits=True)))
    assert "dlpf_steps" not in log and log.get("dl_dilated", 0) > 0, log
    monkeypatch.delenv("WN_COOP_CAPACITY")
    for b in range(B):
        assert float((out["c"][1][b] - lp[b]).abs().max()) <= 1e-4
        safe = _safe(lp[b])
        assert bool((out["c"][0][b][safe] == tp[b][safe]).all())


def _hog_helper():
    import ctypes
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available to build the stand-in kernel")
    out_dir = os.path.join(root, "tests", "emu", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "gpu_hog.so")
    src = os.path.join(root, "tests", "gpu_helpers", "hog.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def test_a_launch_that_cannot_be_resident_times_out_and_the_decode_is_redone_by_layer_wise_launches():
    """What the occupancy query cannot see: another kernel holds compute units when the persistent launch starts.  A stand-in
    (tests/gpu_helpers/hog.hip: one workgroup per CU with 150 KB of LDS, on a side stream) keeps all but 32 CUs busy; the
    16-utterance launch needs 64 workgroups at once, so half of them poll for flags nobody can publish until their bounded
    polls run out: the error word is set, the workgroups that start afterwards return at once, ``WaveNetEngine.decode`` sees the
    word after the chunk, warns and decodes again by layer-wise launches (which need no co-residency) -- same tokens as the
    undisturbed persistent launch.  Everything is bounded: the stand-in stops on a word the test sets (or after 40 s)."""
    import time
    import warnings
    hog = _hog_helper()
    cfg, params, model = _model(RECIPE_K2)
    eng = model.engine
    rs = np.random.RandomState(9)
    B = 16
    x = torch.from_numpy(rs.randint(0, 256, (B, 4))).long().to(DEV)
    h = torch.from_numpy(rs.standard_normal((B, 80, 2)).astype(np.float32)).to(DEV)
    ns = [5] * B
    ok, wg, cap = eng.decode_residency(B)
    assert ok and wg == 64
    ref_t, ref_l = eng.decode(x, h, ns, return_logits=True)          # the undisturbed persistent launch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if cus < 128 or cap < 64:
        pytest.skip("needs a device that keeps the 64-workgroup launch resident with CUs to spare")
    stop = torch.zeros(1, dtype=torch.int32, device=DEV)
    started = torch.zeros(1, dtype=torch.int64, device=DEV)
    side = torch.cuda.Stream(device=DEV)
    torch.cuda.synchronize()
    n_hog = cus - 32
    rc = hog.hog_launch(n_hog, 150 * 1024, stop.data_ptr(), 40 * 100000000, started.data_ptr(), side.cuda_stream)
    assert rc == 0
    t0 = time.time()
    while int(started.item()) < n_hog and time.time() - t0 < 5.0:   # (the copy engine reads the counter: no CU needed)
        time.sleep(0.01)
    assert int(started.item()) == n_hog, "the stand-in did not get its %d compute units" % n_hog
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            t0 = time.time()
            toks, lg = eng.decode(x, h, ns, return_logits=True)
            dt = time.time() - t0
    finally:
        stop.fill_(1)
        torch.cuda.synchronize()
    assert any("layer-wise launches" in str(m.message) for m in w), [str(m.message) for m in w]
    for b in range(B):
        assert float((lg[b] - ref_l[b]).abs().max()) <= 1e-4
        safe = _safe(ref_l[b])
        assert bool((toks[b][safe] == ref_t[b][safe]).all())
    print("TIME-OUT PATH: %d of %d CUs held by another kernel; the 64-workgroup launch gave up and the decode was redone by "
          "layer-wise launches in %.1f s, tokens equal" % (n_hog, cus, dt))
