# -*- coding: utf-8 -*-
"""GPU parity tests proper: the gfx950 library called through the C-ABI on a real MI355X against
(a) the committed golden vectors produced by the reference itself, (b) the live CPU oracle on
seeded inputs at sizes it finishes in seconds, and (c) size-independent properties at
BASELINE.json's full size (config 2: B=8, T=23040)."""
import ctypes

import pytest
import torch

from tests import parity_common as PC
from tests.golden_util import CASES, GoldenCase

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _lib():
    from pytorchwavenetvocoder_amd import _lib as L
    lib = L.load_library()
    assert not lib.is_emulator
    return lib


@pytest.mark.parametrize("name", ["tiny_k3_noup", "tiny_init", "r64_k2_up"])
def test_reference_training_loop_with_stock_adam(name):
    """autograd + torch.optim.Adam on the flat-buffer parameter views == the reference's weights."""
    PC.check_reference_training_loop(GoldenCase(name), _lib(), DEV)


@pytest.mark.parametrize("name", CASES)
def test_engine_vs_golden(name):
    PC.check_golden_case(GoldenCase(name), _lib(), DEV)


@pytest.mark.parametrize("name", ["r64_k2_up", "r64_k3_up"])
def test_layered_kernels_vs_golden_r64(name):
    from pytorchwavenetvocoder_amd import _lib as L
    PC.check_golden_case(GoldenCase(name), _lib(), DEV, flags=L.FLAG_NO_FUSED)


@pytest.mark.parametrize("name", ["tiny_k2_up", "tiny_k3_noup", "r64_k2_up"])
def test_module_training_vs_golden(name):
    PC.check_module_training(GoldenCase(name), _lib(), DEV)


@pytest.mark.parametrize("name,lpb", [("tiny_k2_up", 4), ("r64_k2_up", 3), ("r64_k3_up", 1)])
def test_bucketed_backward(name, lpb):
    PC.check_golden_case(GoldenCase(name), _lib(), DEV, layers_per_bucket=lpb)


def test_ragged_and_odd_shapes_vs_oracle():
    PC.run_oracle_vs_engine((37, 7, 12, 20, 2, 2, 2, 0), 3, 77, 5, _lib(), DEV)
    PC.run_oracle_vs_engine((37, 7, 12, 20, 3, 1, 3, 7), 2, 91, 6, _lib(), DEV)
    PC.run_oracle_vs_engine((48, 9, 96, 160, 2, 1, 2, 4), 1, 72, 7, _lib(), DEV)


def test_vector_staging_paths():
    from pytorchwavenetvocoder_amd import _lib as L
    PC.run_oracle_vs_engine((64, 8, 64, 64, 3, 1, 2, 8), 1, 256, 9, _lib(), DEV)
    PC.run_oracle_vs_engine((64, 8, 64, 64, 3, 1, 2, 8), 1, 256, 9, _lib(), DEV, flags=L.FLAG_NO_FUSED)


def test_reference_test_shapes():
    """The model shapes of the reference's own test/test_wavenet.py:31-71 (shape assertions)."""
    from pytorchwavenetvocoder_amd.nets import WaveNet, initialize
    for args, hlen in [((256, 28, 32, 128, 10, 1, 2), 100), ((256, 28, 32, 128, 10, 1, 2, 10), 10),
                       ((256, 28, 32, 128, 10, 1, 3, 10), 10)]:
        net = WaveNet(*args)
        net.apply(initialize)
        net.eval()
        net.to(DEV)
        x = torch.randint(0, 256, (1, 100), device=DEV)
        h = torch.rand(1, 28, hlen, device=DEV)
        y = net(x, h)[0]
        assert y.size(0) == 100 and y.size(1) == 256


def test_cfg2_model_midsize_vs_oracle():
    """The BASELINE config-2 MODEL (30 layers, 64/256 ch, A=80, U=80) on a window the CPU oracle
    finishes in seconds (T=3200/3280 > rf=3070), trained-scale weights, instances with a verified
    ReLU-kink margin (parity_common.pick_instance)."""
    cfg_t = (256, 80, 64, 256, 10, 3, 2, 80)
    e1, g1 = PC.run_oracle_vs_engine(cfg_t, 1, 3120, 21, _lib(), DEV, scale=0.05)
    e2, g2 = PC.run_oracle_vs_engine(cfg_t, 1, 3120, 22, _lib(), DEV, scale=0.02)
    print("cfg2-model logits err %.3g / %.3g, worst grad rel err %.3g / %.3g" % (e1, e2, g1, g2))


def test_cfg2_fused_equals_layered_midsize():
    """Fused R=64 kernels and the layered any-size kernels are two implementations of the same
    math; they must agree to fp32 round-off."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib as L
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    cfg_t = (256, 80, 64, 256, 10, 3, 2, 80)
    cfg = O.OracleConfig(*cfg_t)
    params, x, h, t, margin, sd = PC.pick_instance(cfg, 1, 3120, 31, 0.05)
    outs = []
    for flags in (0, L.FLAG_NO_FUSED):
        eng = WaveNetEngine(*cfg_t, device=DEV, library=_lib())
        eng.flags = flags
        load_state_into_flat(eng, params)
        logits = eng.forward(x.to(DEV), h.to(DEV))
        loss, dl = eng.loss(logits, t.to(DEV))
        g = eng.backward(dl).clone()
        outs.append((logits.clone(), float(loss.cpu()), g))
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= 2e-5
    assert abs(outs[0][1] - outs[1][1]) <= 1e-6
    assert float((outs[0][2] - outs[1][2]).abs().max()) <= 1e-5 * float(outs[1][2].abs().max())


def test_stream_overlap_modes_midsize():
    """Opt-in overlap modes.  With WN_FLAG_BWD_OVERLAP wn_backward's weight gradients run on the library's side stream
    beside the gate'/dX chain: same kernels and, for the same launch-group size (WN_FLAG_DW_FLUSH), the same reduction
    order -> bit-identical gradients for every bucket size, run to run (the fork/join must leave nothing pending).
    Different group sizes re-associate the split-K sums, and WN_FLAG_FWD_OVERLAP (skip-sum in chunks beside the
    residual stack) re-associates the skip sum: equal to round-off, and still within the oracle gates."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib as L
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    cfg_t = (256, 80, 64, 256, 10, 3, 2, 80)
    cfg = O.OracleConfig(*cfg_t)
    params, x, h, t, margin, sd = PC.pick_instance(cfg, 2, 3120, 41, 0.05)
    x, h, t = x.to(DEV), h.to(DEV), t.to(DEV)

    def run(flags, lpb):
        eng = WaveNetEngine(*cfg_t, device=DEV, library=_lib())
        eng.flags = flags
        load_state_into_flat(eng, params)
        out = []
        for rep in range(2):
            logits = eng.forward(x, h)
            loss, dl = eng.loss(logits, t)
            g = eng.backward(dl, layers_per_bucket=lpb).clone()
            torch.cuda.synchronize()
            out.append((logits.clone(), float(loss.cpu()), g))
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][2], out[1][2]), (flags, lpb)
        return out[0]

    F5 = L.flag_dw_flush(5)
    for lpb in (0, 10, 7):
        serial = run(F5, lpb)
        over = run(L.FLAG_BWD_OVERLAP | F5, lpb)
        assert torch.equal(serial[0], over[0]) and serial[1] == over[1]
        assert torch.equal(serial[2], over[2]), "side-stream weight gradients differ from the serial ones (lpb %d)" % lpb
    base = run(0, 0)  # serial, one launch group per bucket
    for lpb in (0, 10):  # head-only overlap keeps the bucket-size groups of the serial sequence: bit-identical to it
        head = run(L.FLAG_BWD_OVERLAP | L.FLAG_BWD_OVERLAP_HEAD, lpb)
        ref = base if lpb == 0 else run(0, lpb)
        assert torch.equal(head[0], ref[0]) and torch.equal(head[2], ref[2]), "head-only overlap, lpb %d" % lpb
    for flags in (L.FLAG_BWD_OVERLAP, L.FLAG_BWD_OVERLAP | L.flag_dw_flush(3), L.FLAG_FWD_OVERLAP,
                  L.FLAG_FWD_OVERLAP | L.FLAG_BWD_OVERLAP | L.flag_dw_flush(30)):
        r = run(flags, 0)
        assert float((r[0] - base[0]).abs().max()) <= 2e-5, flags
        assert abs(r[1] - base[1]) <= 1e-6, flags
        assert float((r[2] - base[2]).abs().max()) <= 1e-5 * float(base[2].abs().max()), flags
    e, gerr = PC.run_oracle_vs_engine(cfg_t, 1, 3120, 21, _lib(), DEV, scale=0.05, flags=L.FLAG_FWD_OVERLAP)
    print("fwd-overlap logits err %.3g, worst grad rel err %.3g" % (e, gerr))


def test_cfg2_full_size_properties():
    """BASELINE config 2 at FULL size (B=8, T=23040): properties that need no oracle run.
      * finite loss close to ln(256) for random-init weights and random targets;
      * batch independence: sequence b computed alone (B=1) gives the same logits;
      * causality: changing x[t0:] / h frames from t0 on leaves logits before t0 bit-identical;
      * zero loss-gradient region: dlogits is zero for t < rf;
      * directional derivative of the loss along the gradient matches |g|^2.
    """
    import math
    from pytorchwavenetvocoder_amd.nets import WaveNet, initialize
    torch.manual_seed(1)
    cfg = dict(n_quantize=256, n_aux=80, n_resch=64, n_skipch=256, dilation_depth=10, dilation_repeat=3,
               kernel_size=2, upsampling_factor=80)
    model = WaveNet(**cfg)
    model.apply(initialize)
    model.to(DEV)
    rf = model.receptive_field
    assert rf == 3070
    B, T, U = 8, 23040, 80
    g = torch.Generator().manual_seed(5)
    xx = torch.randint(0, 256, (B, T + 1), generator=g)
    x, t = xx[:, :-1].contiguous().to(DEV), xx[:, 1:].contiguous().to(DEV)
    h = torch.randn(B, 80, T // U, generator=g).to(DEV)
    eng = model.engine
    logits = eng.forward(x, h).clone()
    assert torch.isfinite(logits).all()
    loss, dl = eng.loss(logits, t)
    assert abs(float(loss.cpu()) - math.log(256)) < 0.5
    assert float(dl[:, :, :rf].abs().max()) == 0.0
    flat = eng.backward(dl).clone()
    assert torch.isfinite(flat).all()
    lo, hi = eng.dead_range
    assert float(flat[lo:hi].abs().max()) == 0.0
    # run-to-run reproducibility: every reduction in the path has a fixed order (no float atomics
    # on global memory), so the same inputs give the same bits
    logits_b = eng.forward(x, h).clone()
    assert torch.equal(logits_b, logits)
    loss_b, dl_b = eng.loss(logits_b, t)
    assert torch.equal(loss_b, loss)
    flat_b = eng.backward(dl_b)
    assert torch.equal(flat_b, flat)
    # batch independence
    one = eng.forward(x[3:4].contiguous(), h[3:4].contiguous())
    assert float((one[0] - logits[3]).abs().max()) <= 2e-5
    # causality (t0 on a frame boundary so the aux features before t0 are unchanged)
    t0 = 80 * 150
    x2, h2 = x.clone(), h.clone()
    x2[:, t0:] = (x2[:, t0:] + 17) % 256
    h2[:, :, t0 // U:] += 1.0
    l2 = eng.forward(x2, h2)
    assert torch.equal(l2[:, :, :t0], logits[:, :, :t0])
    assert not torch.equal(l2[:, :, t0:], logits[:, :, t0:])
    # directional derivative along the gradient:  L(p - eps g) - L(p) ~= -eps |g|^2
    p0 = eng.flat_params.clone()
    gnorm2 = float((flat.double() ** 2).sum())
    eps = 0.05 / max(gnorm2 ** 0.5, 1e-12)
    base = float(loss.cpu())
    eng.flat_params.copy_(p0 - eps * flat)
    lm, _ = eng.loss(eng.forward(x, h), t, want_grad=False)
    eng.flat_params.copy_(p0 + eps * flat)
    lp, _ = eng.loss(eng.forward(x, h), t, want_grad=False)
    eng.flat_params.copy_(p0)
    fd = (float(lp.cpu()) - float(lm.cpu())) / (2 * eps)
    assert abs(fd - gnorm2) <= 0.05 * gnorm2 + 1e-6, (fd, gnorm2, base)


# ---- autoregressive decode (BASELINE config 5, wn_decode.hip) -----------------------------------
import numpy as np  # noqa: E402

from oracle import wavenet_oracle as O  # noqa: E402
from tests.decode_common import DECODE_CASES, check_decode_case  # noqa: E402


@pytest.mark.parametrize("name", DECODE_CASES)
def test_decode_vs_reference_generation(name):
    """Decode kernel == the reference's fast_generate / batch_fast_generate golden outputs."""
    check_decode_case(name, _lib(), DEV)


def test_decode_cfg2_teacher_forced_equals_training_forward():
    """Full-size model (30 layers, 64/256 channels, U=80): the logits the decode kernel computes
    while it walks a context longer than the receptive field must equal the training forward's
    logits at the same positions (same HIP library, two very different kernels), and the generated
    continuation must be the argmax path of the training forward."""
    from pytorchwavenetvocoder_amd.nets import WaveNet
    cfg = O.OracleConfig(256, 80, 64, 256, 10, 3, 2, 80)
    params = O.random_params(cfg, 77, scale=0.1)
    model = WaveNet(*cfg.as_tuple())
    model.load_state_dict(params)
    model.to(DEV)
    B, T0, n = 2, 3360, 80          # T0 > rf = 3070, multiple of U
    rs = np.random.RandomState(78)
    x = torch.from_numpy(rs.randint(0, 256, (B, T0))).long().to(DEV)
    h = torch.from_numpy(rs.standard_normal((B, 80, (T0 + n) // 80)).astype(np.float32)).to(DEV)
    # "walk": the decode kernel alone builds the queues (3360 teacher-forced steps); "parallel": the residual
    # stack of the training forward builds them (what generation does by default)
    for prefill in ("walk", "parallel"):
        toks, lg = model.engine.decode(x, h, [n, n - 17], mode="argmax", return_logits=True, chunk=1000, prefill=prefill)
        full = torch.cat([x, torch.stack([toks[0], torch.cat([toks[1], toks[1].new_zeros(17)])])], dim=1)
        logits = model.engine.forward(full, h)          # (B, Q, T0+n)
        for b, nb in enumerate([n, n - 17]):
            # generated sample i was chosen from the logits of position T0-1+i
            ref = logits[b, :, T0 - 1:T0 - 1 + nb].transpose(0, 1)
            assert float((lg[b] - ref).abs().max()) <= 1e-4, prefill
            top2 = ref.topk(2, dim=1).values
            safe = (top2[:, 0] - top2[:, 1]) > 1e-3
            assert bool((ref.argmax(1)[safe] == toks[b][safe]).all()), prefill
            assert int(safe.sum()) > nb // 2
    # a short context (one token, left padding of rf-1 positions with the first upsampled aux column replicated)
    x1 = x[:, :1].contiguous()
    tw, lw = model.engine.decode(x1, h, [40, 40], mode="argmax", return_logits=True, prefill="walk")
    tp, lp = model.engine.decode(x1, h, [40, 40], mode="argmax", return_logits=True, prefill="parallel")
    for b in range(B):
        assert float((lw[b] - lp[b]).abs().max()) <= 1e-4
        top2 = lw[b].topk(2, dim=1).values
        safe = (top2[:, 0] - top2[:, 1]) > 1e-3
        assert bool((tw[b][safe] == tp[b][safe]).all())


def test_decode_wide_model_layered_path_long_k():
    """n_resch = 256: the layer-wise contractions have K = 512 and run the 16-wave variant of the skinny
    matrix kernel; tokens / logits must still be the queue algorithm's (oracle)."""
    from pytorchwavenetvocoder_amd.nets import WaveNet
    cfg = O.OracleConfig(64, 6, 256, 288, 3, 2, 2, 4)
    params = O.random_params(cfg, 11, scale=0.05)
    model = WaveNet(*cfg.as_tuple())
    model.load_state_dict(params)
    model.to(DEV)
    assert not model.engine.decode_supported()
    x = torch.tensor([[5, 9, 1, 30], [7, 7, 2, 0], [1, 2, 3, 4]]).long()
    h = torch.from_numpy(np.random.RandomState(12).standard_normal((3, 6, 10)).astype(np.float32))
    ref, ref_lg = O.batch_fast_generate(cfg, params, x, h, [24, 24, 24], return_logits=True)
    for prefill in ("parallel", "walk"):
        toks, lg = model.engine.decode(x.to(DEV), h.to(DEV), [24, 24, 24], return_logits=True, prefill=prefill)
        for b in range(3):
            assert float((lg[b].cpu() - ref_lg[b]).abs().max()) <= 1e-4, (prefill, b)
            top2 = ref_lg[b].topk(2, dim=1).values
            safe = ((top2[:, 0] - top2[:, 1]) > 1e-3).numpy()
            assert (toks[b].cpu().numpy()[safe] == ref[b][safe]).all(), (prefill, b)


def test_decode_any_size_persistent_launch_vs_oracle_and_launches():
    """csrc/wn_dlp.hip / wn_dlpm.hip: the any-size decode as ONE launch of workgroups that hand their vectors to each other as
    tagged granules (wavenet.py:355-385, 538-549, 518-523 with the res 1x1 folded into the next layer's newest tap).  A
    128-channel model (kernel_size 3; 7 ragged utterances) and the recipes' own size (n_resch 512 / n_skipch 256, 2 utterances):
    by default both through wn_dlpf.hip (plain vectors + flags), with ``layered="granules"`` through wn_dlpm.hip / the fp32 VALU
    kernel wn_dlp.hip: logits within
    1e-4 of the queue algorithm (oracle), tokens equal wherever the oracle's argmax is not a near-tie, the same against the
    layer-wise launches it replaces, both ways of building the context queues, and the sampling mode runs."""
    from pytorchwavenetvocoder_amd.nets import WaveNet
    for cfg_t, B, n, scale, seed in (((64, 4, 128, 128, 3, 2, 3, 4), 7, 26, 0.1, 21), ((256, 80, 512, 256, 10, 3, 2, 80), 2, 12, 0.02, 22)):
        cfg = O.OracleConfig(*cfg_t)
        params = O.random_params(cfg, seed, scale=scale)
        model = WaveNet(*cfg_t)
        model.load_state_dict(params)
        model.to(DEV)
        assert not model.engine.decode_supported()
        import ctypes
        assert model.engine.lib.wn_decode_layered_error_offset(ctypes.byref(model.engine.cfg), B, 0) >= 0   # the persistent path applies
        rs = np.random.RandomState(seed)
        x = torch.from_numpy(rs.randint(0, cfg.n_quantize, (B, 4))).long()
        U = max(cfg.upsampling_factor, 1)
        h = torch.from_numpy(rs.standard_normal((B, cfg.n_aux, (4 + n) // U + 2)).astype(np.float32))
        ns = [n - (b % 5) for b in range(B)]
        ref, ref_lg = O.batch_fast_generate(cfg, params, x, h, ns, return_logits=True)
        order = sorted(range(B), key=lambda i: (ns[i], i))   # oracle order: shortest first
        tl, ll = model.engine.decode(x.to(DEV), h.to(DEV), ns, return_logits=True, layered="launches")
        for prefill in ("parallel", "walk"):
            tp, lp = model.engine.decode(x.to(DEV), h.to(DEV), ns, return_logits=True, prefill=prefill, chunk=7)
            for k, i in enumerate(order):
                assert float((lp[i].cpu() - ref_lg[k]).abs().max()) <= 1e-4, (cfg_t, prefill, i)
                assert float((lp[i] - ll[i]).abs().max()) <= 1e-4, (cfg_t, prefill, i)
                top2 = ref_lg[k].topk(2, dim=1).values
                safe = ((top2[:, 0] - top2[:, 1]) > 1e-3).numpy()
                assert (tp[i].cpu().numpy()[safe] == ref[k][safe]).all(), (cfg_t, prefill, i)
                assert (tp[i].cpu().numpy()[safe] == tl[i].cpu().numpy()[safe]).all(), (cfg_t, prefill, i)
        ts = model.engine.decode(x.to(DEV), h.to(DEV), ns, mode="sampling")
        assert all(int(t.min()) >= 0 and int(t.max()) < cfg.n_quantize and len(t) == k for t, k in zip(ts, ns))
        # the granule hand-off: the fp32 VALU kernel wn_dlp.hip up to 4 utterances, wn_dlpm.hip from 5 on
        tg, lg = model.engine.decode(x.to(DEV), h.to(DEV), ns, return_logits=True, layered="granules")
        for i in range(B):
            assert float((lg[i] - ll[i]).abs().max()) <= 1e-4, (cfg_t, "granules", i)
            top2 = ll[i].topk(2, dim=1).values
            safe = ((top2[:, 0] - top2[:, 1]) > 1e-3).cpu().numpy()
            assert (tg[i].cpu().numpy()[safe] == tl[i].cpu().numpy()[safe]).all(), (cfg_t, "granules", i)


def test_decode_any_size_wide_batches_on_the_matrix_cores():
    """csrc/wn_dlpf.hip / wn_dlpm.hip (5 .. 48 utterances: 16 x 16 tiles of v_mfma_f32_16x16x4_f32, one set of n_resch / 8
    workgroups per block of 16 utterances; hand-off by plain vectors + flags / by granules) at the recipes' own size (n_resch 512
    / n_skipch 256): 18 ragged utterances = 2 blocks = 128 workgroups.
    Every utterance against the layer-wise launches it replaces (logits 1e-4, tokens equal away from near-ties), the first
    one, the last one of block 0 and the last one of block 1 against the queue algorithm (oracle) as well; the launch log shows
    the persistent kernel and no layer-wise launch; the sampling mode draws the same tokens as the launches."""
    from pytorchwavenetvocoder_amd.nets import WaveNet
    from tests import parity_common as PC
    cfg_t, B, n, seed = (256, 80, 512, 256, 10, 3, 2, 80), 18, 10, 23
    cfg = O.OracleConfig(*cfg_t)
    params = O.random_params(cfg, seed, scale=0.02)
    model = WaveNet(*cfg_t)
    model.load_state_dict(params)
    model.to(DEV)
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.randint(0, cfg.n_quantize, (B, 4))).long()
    h = torch.from_numpy(rs.standard_normal((B, cfg.n_aux, (4 + n) // 80 + 2)).astype(np.float32))
    ns = [n - (b % 4) for b in range(B)]
    out = {}
    log = PC.launch_log(model.engine.lib, lambda: out.update(p=model.engine.decode(x.to(DEV), h.to(DEV), ns, return_logits=True)))
    assert log.get("dlpf_steps", 0) >= 1 and "dl_dilated" not in log and "dlp_steps" not in log and "dlpm_steps" not in log, log
    tp, lp = out["p"]
    # the same through the granule hand-off (wn_dlpm.hip: the kernel of the classes wn_dlpf.hip does not cover)
    log = PC.launch_log(model.engine.lib, lambda: out.update(g=model.engine.decode(x.to(DEV), h.to(DEV), ns, return_logits=True, layered="granules")))
    assert log.get("dlpm_steps", 0) >= 1 and "dlpf_steps" not in log and "dl_dilated" not in log, log
    for i in range(B):
        assert float((out["g"][1][i] - lp[i]).abs().max()) <= 1e-5 and torch.equal(out["g"][0][i], tp[i]), i
    tl, ll = model.engine.decode(x.to(DEV), h.to(DEV), ns, return_logits=True, layered="launches")
    for i in range(B):
        assert float((lp[i] - ll[i]).abs().max()) <= 1e-4, i
        top2 = ll[i].topk(2, dim=1).values
        safe = ((top2[:, 0] - top2[:, 1]) > 1e-3).cpu().numpy()
        assert (tp[i].cpu().numpy()[safe] == tl[i].cpu().numpy()[safe]).all(), i
    for i in (0, 15, 17):
        rt, rl = O.fast_generate(cfg, params, x[i:i + 1], h[i:i + 1], ns[i], return_logits=True)
        assert float((lp[i].cpu() - rl).abs().max()) <= 1e-4, i
        top2 = rl.topk(2, dim=1).values
        safe = ((top2[:, 0] - top2[:, 1]) > 1e-3).numpy()
        assert (tp[i].cpu().numpy()[safe] == np.asarray(rt)[safe]).all(), i
    torch.manual_seed(5)
    sp = model.engine.decode(x.to(DEV), h.to(DEV), ns, mode="sampling")
    torch.manual_seed(5)
    sl = model.engine.decode(x.to(DEV), h.to(DEV), ns, mode="sampling", layered="launches")
    assert all(torch.equal(a, b) for a, b in zip(sp, sl))


def test_decode_any_size_model_uses_the_layered_path():
    """n_resch = 128 is outside the compiled classes of the persistent decode kernel: fast_generate /
    batch_fast_generate run the layer-wise path and must reproduce the queue algorithm (oracle)."""
    from pytorchwavenetvocoder_amd.nets import WaveNet
    cfg = O.OracleConfig(32, 4, 128, 160, 3, 2, 2, 4)
    params = O.random_params(cfg, 3, scale=0.1)
    model = WaveNet(*cfg.as_tuple())
    model.load_state_dict(params)
    model.to(DEV)
    assert not model.engine.decode_supported()
    x = torch.tensor([[5, 9, 1, 30], [7, 7, 2, 0]]).long()
    h = torch.from_numpy(np.random.RandomState(4).standard_normal((2, 4, 12)).astype(np.float32))
    ref, ref_lg = O.batch_fast_generate(cfg, params, x, h, [30, 22], return_logits=True)
    toks, lg = model.engine.decode(x.to(DEV), h.to(DEV), [30, 22], return_logits=True)
    for b, i in enumerate([1, 0]):   # oracle order: shortest first
        assert float((lg[i].cpu() - ref_lg[b]).abs().max()) <= 1e-4
        top2 = ref_lg[b].topk(2, dim=1).values
        safe = ((top2[:, 0] - top2[:, 1]) > 1e-3).numpy()
        assert (toks[i].cpu().numpy()[safe] == ref[b][safe]).all()
    out = model.batch_fast_generate(x.to(DEV), h.to(DEV), [30, 22], mode="argmax")
    assert [len(o) for o in out] == [22, 30]


def test_training_learns_a_structured_signal_and_split_tracks_exact():
    """End to end: 150 fused-Adam steps on mu-law sine waves conditioned on their frequency must drive the
    loss far below ln(256), and the default (split-bf16) arithmetic must follow the exact-f32-MFMA
    trajectory."""
    from pytorchwavenetvocoder_amd import _lib as L
    from pytorchwavenetvocoder_amd.nets import WaveNet, encode_mu_law, initialize
    from pytorchwavenetvocoder_amd.optim import FusedAdam
    rs = np.random.RandomState(0)
    B, T, U = 4, 2048, 16
    freqs = rs.uniform(0.01, 0.05, size=B)
    tt = np.arange(T + 1)
    wav = np.stack([0.8 * np.sin(2 * np.pi * f * tt) for f in freqs])
    q = torch.from_numpy(encode_mu_law(wav, 256)).long()
    x, t = q[:, :-1].contiguous().to(DEV), q[:, 1:].contiguous().to(DEV)
    h = torch.from_numpy(np.repeat(freqs[:, None, None] * 20.0, T // U, axis=2).astype(np.float32)).repeat(1, 4, 1).to(DEV)
    curves = []
    for flags in (0, L.FLAG_EXACT_MFMA):
        torch.manual_seed(3)
        model = WaveNet(256, 4, 64, 64, 6, 2, 2, U)
        model.apply(initialize)
        model.to(DEV)
        model.engine.flags = flags
        opt = FusedAdam(model, lr=2e-3)
        losses = []
        for _ in range(150):
            losses.append(model.loss_and_backward(x, h, t))
            opt.step()
        curves.append(torch.stack(losses).squeeze().cpu())
    split, exact = curves
    assert float(split[0]) > 5.0 and float(split[-1]) < 2.5, (float(split[0]), float(split[-1]))
    assert float((split[:3] - exact[:3]).abs().max()) < 1e-4       # identical start ...
    assert float((split[:20] - exact[:20]).abs().max()) < 1e-2     # ... and the same trajectory (Adam amplifies round-off)
    assert abs(float(split[-1]) - float(exact[-1])) < 0.15
