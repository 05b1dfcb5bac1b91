# -*- coding: utf-8 -*-
"""CPU-only: the HIP kernel SOURCES compiled for the host (tests/emu) against the reference's
golden vectors and the oracle.  Validates index arithmetic / LDS layouts / MFMA lane maps of the
exact code that hipcc builds for gfx950; the same checks run on the real GPU in test_gpu_parity.py."""
import pytest
import torch

from tests import parity_common as PC
from tests.emu_util import emu_library
from tests.golden_util import CASES, GoldenCase
from tests.decode_common import DECODE_CASES, check_decode_case

pytestmark = pytest.mark.emu


@pytest.mark.parametrize("name", CASES)
def test_engine_vs_golden(name):
    PC.check_golden_case(GoldenCase(name), emu_library(), "cpu")


@pytest.mark.parametrize("name", ["tiny_k2_up", "tiny_k3_noup"])
def test_module_training_vs_golden(name):
    PC.check_module_training(GoldenCase(name), emu_library(), "cpu")


@pytest.mark.parametrize("name", ["tiny_k3_noup", "tiny_init"])
def test_reference_training_loop_with_stock_adam(name):
    PC.check_reference_training_loop(GoldenCase(name), emu_library(), "cpu")


@pytest.mark.parametrize("name", ["r64_k2_up", "r64_k3_up"])
def test_layered_path_vs_golden_r64(name):
    from pytorchwavenetvocoder_amd import _lib
    PC.check_golden_case(GoldenCase(name), emu_library(), "cpu", flags=_lib.FLAG_NO_FUSED)


@pytest.mark.parametrize("name,lpb", [("tiny_k2_up", 4), ("tiny_k2_up", 1), ("r64_k2_up", 3), ("tiny_k3_noup", 2)])
def test_bucketed_backward(name, lpb):
    """Gradient buckets (layer-batched weight-gradient launches) of any size give the same result."""
    PC.check_golden_case(GoldenCase(name), emu_library(), "cpu", layers_per_bucket=lpb)


@pytest.mark.parametrize("name", ["r64_k2_up", "r64_k3_up"])
def test_overlap_launch_sequences(name):
    """The overlap modes of wn_forward / wn_backward change the launch sequence (skip-sum in chunks of layers that
    accumulate in place; weight gradients issued per bucket from a second context).  The emulator has one in-order
    stream, so this checks the sequences' arithmetic; the stream fork/join itself is covered on the GPU."""
    from pytorchwavenetvocoder_amd import _lib
    PC.check_golden_case(GoldenCase(name), emu_library(), "cpu", flags=_lib.FLAG_FWD_OVERLAP | _lib.FLAG_BWD_OVERLAP)
    if name == "r64_k3_up":   # the K = 3 case repeats the launch sequences of the K = 2 one: the first mode is enough
        return
    PC.check_golden_case(GoldenCase(name), emu_library(), "cpu", flags=_lib.FLAG_FWD_OVERLAP | _lib.FLAG_EXACT_MFMA,
                         layers_per_bucket=1)
    PC.check_golden_case(GoldenCase(name), emu_library(), "cpu", flags=_lib.FLAG_BWD_OVERLAP)
    # launch groups of 1 / 2 walked layers inside buckets of 2 / the whole stack
    PC.check_golden_case(GoldenCase(name), emu_library(), "cpu", flags=_lib.FLAG_BWD_OVERLAP | _lib.flag_dw_flush(1), layers_per_bucket=2)
    PC.check_golden_case(GoldenCase(name), emu_library(), "cpu", flags=_lib.flag_dw_flush(2))
    # only the post-net / skip weight gradients from the second context
    PC.check_golden_case(GoldenCase(name), emu_library(), "cpu", flags=_lib.FLAG_BWD_OVERLAP | _lib.FLAG_BWD_OVERLAP_HEAD,
                         layers_per_bucket=1)


def test_aux_gradient_partials_in_the_gate_kernel():
    """WN_FLAG_AUX_FUSED: the gate kernel reduces w[j]*dP over 16-sample groups (DPP rows on the GPU) and dP*G over
    the channel rows, wn_aux_finish sums groups per frame / frames per phase; the gradients of upsampling.conv and
    aux_1x1_* must meet the same gates as the wn_aux_bwd path.  U = 16, 32, 80 (1, 2, 5 groups per frame), a last tile
    that is half full (T % 32 == 16), several sequences, one launch group per layer."""
    from pytorchwavenetvocoder_amd import _lib
    F = _lib.FLAG_AUX_FUSED
    PC.check_golden_case(GoldenCase("r64_k2_up"), emu_library(), "cpu", flags=F)
    PC.check_golden_case(GoldenCase("r64_k2_up"), emu_library(), "cpu", flags=F | _lib.FLAG_BWD_OVERLAP, layers_per_bucket=1)
    PC.run_oracle_vs_engine((64, 6, 64, 32, 3, 1, 2, 80), 1, 160, 31, emu_library(), "cpu", flags=F, scale=0.2)
    PC.run_oracle_vs_engine((64, 6, 64, 32, 2, 2, 2, 16), 2, 48, 32, emu_library(), "cpu", flags=F, scale=0.2)
    PC.run_oracle_vs_engine((64, 6, 64, 32, 2, 1, 2, 32), 2, 128, 33, emu_library(), "cpu", flags=F, scale=0.2)
    # U % 16 != 0: the flag falls back to wn_aux_bwd
    PC.check_golden_case(GoldenCase("r64_k3_up"), emu_library(), "cpu", flags=F)


def test_bench_self_check_of_the_aux_fused_mode():
    """bench.check_aux_fused compares the gradients of the two aux-gradient modes (separate wn_aux_bwd launch vs the
    partial sums inside the gate kernel, the default) on a batch; here on a small fused-kernel model under the emulator."""
    import bench
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd.nets import WaveNet
    cfg_t = (64, 6, 64, 32, 2, 2, 2, 16)
    cfg = O.OracleConfig(*cfg_t)
    model = WaveNet(*cfg_t, _library=emu_library())
    model.load_state_dict(O.random_params(cfg, 11, scale=0.2))
    x, h, t = O.synthetic_batch(cfg, 2, 64, 12)
    flags0 = model.engine.flags
    ok, worst = bench.check_aux_fused(model, x, h, t)
    assert ok and 0.0 < worst <= 1e-5, worst   # > 0: the fused mode really ran (its sums re-associate)
    assert model.engine.flags == flags0


def test_ragged_T_and_odd_channels():
    # T not a multiple of any tile, channel counts not multiples of 32, B=3
    PC.run_oracle_vs_engine((37, 7, 12, 20, 2, 2, 2, 0), 3, 77, 5, emu_library(), "cpu")
    PC.run_oracle_vs_engine((37, 7, 12, 20, 3, 1, 3, 7), 2, 91, 6, emu_library(), "cpu")


def test_vector_staging_paths():
    """T large enough for interior 128-wide tiles: exercises the 16-byte staging path of the GEMM (and its
    per-tile fallback to scalar loads for taps whose shift is not a multiple of 4)."""
    from pytorchwavenetvocoder_amd import _lib
    PC.run_oracle_vs_engine((64, 8, 64, 64, 3, 1, 2, 8), 1, 256, 9, emu_library(), "cpu")
    PC.run_oracle_vs_engine((64, 8, 64, 64, 3, 1, 2, 8), 1, 256, 9, emu_library(), "cpu", flags=_lib.FLAG_NO_FUSED)


def test_split_bf16_contractions_vs_exact_mfma():
    """Skip-sum / post-net contractions with >= 128 output rows run on the bf16 matrix cores with a 3-way
    operand split (wn_gemm6.hip); they must meet the same gates as the exact-f32 MFMA path
    (WN_FLAG_EXACT_MFMA), including a ragged T, K not a multiple of 16 and M not a multiple of 256."""
    from pytorchwavenetvocoder_amd import _lib
    for flags in (0, _lib.FLAG_EXACT_MFMA):
        PC.run_oracle_vs_engine((200, 6, 16, 136, 2, 2, 2, 0), 1, 45, 21, emu_library(), "cpu", flags=flags, scale=0.2)
    PC.run_oracle_vs_engine((256, 5, 64, 256, 2, 1, 2, 4), 1, 36, 22, emu_library(), "cpu", scale=0.2)
    # any-size layered path (n_resch = 128): dilated taps with shifts, residual and accumulate epilogues
    PC.run_oracle_vs_engine((40, 5, 128, 144, 3, 1, 3, 0), 1, 40, 23, emu_library(), "cpu", scale=0.1)


def test_wide_channels_multi_tile():
    # R > 64 exercises the 128-wide tiles and multi-tile M
    PC.run_oracle_vs_engine((48, 9, 96, 160, 2, 1, 2, 4), 1, 72, 7, emu_library(), "cpu")


@pytest.mark.parametrize("name", DECODE_CASES)
def test_decode_vs_reference_generation(name):
    """Decode kernel (wn_decode.hip) == the reference's fast_generate / batch_fast_generate outputs."""
    # the layer-wise path is exercised on the small cases here and on all cases on the GPU
    check_decode_case(name, emu_library(), "cpu", layered_too=name.startswith("decode_tiny") or name.endswith("longctx"))


def test_cpu_tensors_rejected_by_product_binding():
    """The product binding must refuse CPU tensors (no CPU fallback)."""
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.nets import WaveNet
    lib = _lib.load_library()  # the real gfx950 library (cross-compiled; loads without a GPU)
    assert not lib.is_emulator
    model = WaveNet(16, 4, 4, 4, 2, 1, 2, 0)
    x = torch.zeros(1, 8, dtype=torch.long)
    h = torch.zeros(1, 4, 8)
    with pytest.raises(_lib.WnError):
        model(x, h)


def test_decode_context_lengths_and_ragged_requests():
    """Contexts shorter than / equal to / longer than the receptive field (left padding, exact fit, the tail
    passed to the one-pass context), zero-length and ragged requests, K = 3 and no upsampling layer, on both
    decode paths: walking the context == one pass over it, and both == the queue algorithm (oracle)."""
    import numpy as np
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd.nets import WaveNet
    for cfg_t in [(16, 4, 8, 12, 3, 2, 2, 4), (16, 4, 8, 12, 2, 2, 3, 0)]:
        cfg = O.OracleConfig(*cfg_t)
        params = O.random_params(cfg, 5, scale=0.3)
        m = WaveNet(*cfg_t, _library=emu_library())
        m.load_state_dict(params)
        rf = m.receptive_field
        for T0 in ((1, rf, rf + 1, 3 * rf) if cfg_t[7] > 0 else (1, rf + 1)):   # the second model repeats two of the lengths
            rs = np.random.RandomState(T0)
            x = torch.from_numpy(rs.randint(0, 16, (3, T0))).long()
            U = cfg_t[7]
            tot = T0 + 9
            h = torch.from_numpy(rs.standard_normal((3, 4, (tot + U - 1) // U if U > 0 else tot)).astype(np.float32))
            ns = [9, 0, 1]
            for lay in (False, True):
                a, la = m.engine.decode(x, h, ns, return_logits=True, layered=lay, prefill="walk")
                b, lb = m.engine.decode(x, h, ns, return_logits=True, layered=lay, prefill="parallel")
                for i, n in enumerate(ns):
                    assert a[i].numel() == n and b[i].numel() == n
                    if n:
                        assert float((la[i] - lb[i]).abs().max()) < 1e-4, (cfg_t, T0, lay, i)
                        assert (a[i] == b[i]).all(), (cfg_t, T0, lay, i)
            ref, rl = O.batch_fast_generate(cfg, params, x, h, [9, 9, 9], return_logits=True)
            c, lc = m.engine.decode(x, h, [9, 9, 9], return_logits=True)
            for i in range(3):
                assert float((lc[i] - rl[i]).abs().max()) < 1e-4, (cfg_t, T0, i)
                assert (c[i].numpy() == ref[i]).all(), (cfg_t, T0, i)
            if U == 0:
                # no upsampling layer: the aux columns are projected per chunk of steps; on the any-size path every chunk
                # after the first passes params = NULL to wn_decode_layered_prepare (weights stay packed in the state)
                for lay in (False, True):
                    e, le = m.engine.decode(x, h, [9, 9, 9], return_logits=True, layered=lay, chunk=4)
                    for i in range(3):
                        assert float((le[i] - rl[i]).abs().max()) < 1e-4, (cfg_t, T0, lay, i)
                        assert (e[i].numpy() == ref[i]).all(), (cfg_t, T0, lay, i)


def test_persistent_tile_loop_several_tiles_per_wave():
    """The fused kernels are persistent grids whose waves walk several tiles with cross-tile operand prefetch; on the
    emulator's small cases every wave gets at most one tile.  WN_CHAIN_BLOCKS (read once per process -> subprocess) caps
    the grid at 8 workgroups, so the first wave of every workgroup walks 2 of the 72 tiles: default kernels, the tap-interleaved dX order and
    the aux-fused gate kernel against the oracle.  (This case found an aliasing bug of the emulator itself: the 16-byte
    bf16-MFMA payload of wave w overlapped the shuffle slots of wave w + 1, visible only when one wave is in its matrix
    phase while its neighbour shuffles.)"""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from tests import parity_common as PC\n"
        "from tests.emu_util import emu_library\n"
        "from pytorchwavenetvocoder_amd import _lib\n"
        "for fl in (0, _lib.FLAG_AUX_FUSED):\n"
        "    e, g = PC.run_oracle_vs_engine((32, 4, 64, 32, 2, 1, 2, 16), 1, 2304, 51, emu_library(), 'cpu', flags=fl, scale=0.2)\n"
        "    print('flags', fl, 'logits', e, 'grads', g)\n"
        # dilations up to 64 (history taps 1 / 2 tiles back), 100 tiles on 64 waves : several rounds per wave, a partial last round
        "e, g = PC.run_oracle_vs_engine((32, 4, 64, 32, 7, 1, 2, 16), 1, 3200, 52, emu_library(), 'cpu', flags=_lib.FLAG_AUX_FUSED, scale=0.2)\n"
        "print('flags chains', 'logits', e, 'grads', g)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, WN_CHAIN_BLOCKS="8")
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-2000:]
    assert out.count("flags") == 3, out[-2000:]


def test_fused_adam_skips_live_parameters_without_a_gradient_like_torch_adam():
    """torch.optim.Adam leaves a parameter whose .grad is None untouched (value AND moments); FusedAdam's one launch
    covers the whole flat buffer, so it has to put such slices back.  Two steps, the second with two live tensors'
    gradients removed, against torch.optim.Adam on a plain copy of the tensors."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd.nets import WaveNet
    from pytorchwavenetvocoder_amd.optim import FusedAdam
    cfg_t = (32, 6, 8, 12, 3, 2, 2, 4)
    cfg = O.OracleConfig(*cfg_t)
    params = O.random_params(cfg, 3)
    x, h, t = O.synthetic_batch(cfg, 2, 48, 4)
    model = WaveNet(*cfg_t, _library=emu_library())
    model.load_state_dict(params)
    opt = FusedAdam(model, lr=1e-2, weight_decay=1e-3)
    ref = {k: v.clone().requires_grad_(True) for k, v in model.state_dict().items()}
    ropt = torch.optim.Adam(list(ref.values()), lr=1e-2, weight_decay=1e-3)
    drop = {"conv_post_1.weight", "dil_tanh.1.conv.bias"}
    for step in range(2):
        model.loss_and_backward(x, h, t)
        for k, p in model.named_parameters():
            ref[k].grad = None if p.grad is None else p.grad.clone()
            if step == 1 and k in drop:
                p.grad = None
                ref[k].grad = None
        opt.step()
        ropt.step()
    for k, v in model.state_dict().items():
        assert float((v - ref[k].detach()).abs().max()) <= 1e-6, k


def test_op_level_entry_points():
    """wn_op_front / wn_op_causal_conv (a subset of the GPU cases of tests/test_gpu_ops.py) on the host-compiled kernels."""
    from tests import ops_common as OC
    for case in OC.FRONT_CASES[1:3] + [OC.FRONT_CASES[5], OC.FRONT_CASES[8]]:   # the last two: the LDS-table gather (B * T >= 16384), whole table / three row groups
        OC.check_op_front(emu_library(), "cpu", *case)
    for case in (OC.CONV_CASES[2], OC.CONV_CASES[3], (64, 64, 2, 512, 1, 150)):
        OC.check_op_causal_conv(emu_library(), "cpu", *case)
    for case in [(12, 20, 3, 7, 3, 91), (64, 64, 2, 16, 2, 130), (8, 4, 2, 512, 1, 70)]:   # backward (ABI v9): taps transposed, fixed-order dW / db
        OC.check_op_causal_conv_backward(emu_library(), "cpu", *case)


def test_saved_workspace_regions_and_given_relu_subgradient():
    """wn_workspace_region (the views the full-size GPU parity test reads) on a small fused-kernel model: layer inputs,
    gate halves and their product against the oracle's intermediates; and the oracle's gradients under the HIP path's
    own ReLU masks equal its ordinary gradients when no element sits on a kink (the masks then ARE the signs)."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    cfg_t = (64, 8, 64, 64, 3, 2, 2, 16)
    cfg = O.OracleConfig(*cfg_t)
    params, x, h, t, margin, sd = PC.pick_instance(cfg, 2, 96, 13, 0.1)
    eng = WaveNetEngine(*cfg_t, device="cpu", library=emu_library())
    load_state_into_flat(eng, params)
    logits = eng.forward(x, h)
    loss, dl = eng.loss(logits, t)
    m1 = (eng.saved(_lib.WS_RELU_SKIP) > 0).float()
    m2 = (eng.saved(_lib.WS_RELU_POST1) > 0).float()
    l0, lg0, g0 = O.train_step(cfg, params, None, x, h, t)
    l1, lg1, g1, inter = O.train_step(cfg, params, None, x, h, t, relu_masks=(m1, m2), return_intermediates=True)
    assert torch.equal((inter["skip_sum"] > 0).float(), m1) and torch.equal((inter["post1_pre"] > 0).float(), m2)
    assert torch.equal(lg0, lg1) and torch.equal(l0, l1)
    for k in g0:
        assert (g0[k] is None and g1[k] is None) or torch.equal(g0[k], g1[k]), k
    X, S, Z = (eng.saved(k) for k in (_lib.WS_X, _lib.WS_SIGMOID, _lib.WS_Z))
    L = len(cfg.dilations)
    assert tuple(X.shape) == (L, 2, 64, 96)
    for l in range(L):
        ref = inter["x0"] if l == 0 else inter["layer_out"][l - 1]
        assert float((X[l] - ref).abs().max()) <= 1e-5, l
    # the fused kernels save the sigmoid half and z = sigmoid * tanh (the tanh half is rebuilt as z / s in backward) ...
    assert float(S.min()) > 0.0 and float(S.max()) <= 1.0 and float((Z / S).abs().max()) <= 1.0 + 1e-6
    # ... the any-size kernels save all three
    eng2 = WaveNetEngine(*cfg_t, device="cpu", library=emu_library())
    eng2.flags = _lib.FLAG_NO_FUSED
    load_state_into_flat(eng2, params)
    eng2.forward(x, h)
    S2, G2, Z2 = (eng2.saved(k) for k in (_lib.WS_SIGMOID, _lib.WS_TANH, _lib.WS_Z))
    assert float((S2 * G2 - Z2).abs().max()) <= 1e-6 and float(G2.abs().max()) <= 1.0
    assert float((S2 - S).abs().max()) <= 1e-5 and float((Z2 - Z).abs().max()) <= 1e-5
    eng.backward(dl)
    dP, dX, dSk = eng.saved(_lib.WS_DP), eng.saved(_lib.WS_DX), eng.saved(_lib.WS_DSKIP)
    assert tuple(dP.shape) == (L, 2, 128, 96) and tuple(dX.shape) == (L, 2, 64, 96) and tuple(dSk.shape) == (2, 64, 96)
    assert float(dSk[:, :, :cfg.receptive_field].abs().max()) == 0.0   # no loss before the receptive field
    # and a flipped mask element changes the oracle's gradients (the override is live)
    m1b = m1.clone()
    m1b[0, 0, cfg.receptive_field] = 1.0 - m1b[0, 0, cfg.receptive_field]
    _, _, g2 = O.train_step(cfg, params, None, x, h, t, relu_masks=(m1b, m2))
    assert any(g2[k] is not None and not torch.equal(g2[k], g0[k]) for k in g0)


def test_backward_chain_kernel_modes():
    """The one-launch-per-layer backward chain (k_chain64s: dX_l and gate'_{l-1} fused, skip part of dZ pre-contracted;
    the default) against the golden case and the oracle with and without the aux partial sums, kernel_size 1 and 2, a
    half-full last tile and several sequences; the former gate' + dX launch pair (WN_FLAG_NO_CHAIN) stays covered, and
    the two structures agree to round-off."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.engine import DEFAULT_FLAGS, WaveNetEngine, load_state_into_flat
    A, NC = _lib.FLAG_AUX_FUSED, _lib.FLAG_NO_CHAIN
    # chain + aux partials (+ the fp16 pair split of the weight gradients and of the k_gemm6 contractions) is what every default-flag test runs
    assert DEFAULT_FLAGS == A | _lib.FLAG_DW_F16PAIR | _lib.FLAG_MM_F16PAIR | _lib.FLAG_FUSED_F16PAIR
    for flags in (0, A, NC, A | NC):
        PC.check_golden_case(GoldenCase("r64_k2_up"), emu_library(), "cpu", flags=flags)
    PC.run_oracle_vs_engine((64, 6, 64, 32, 3, 1, 1, 16), 2, 48, 41, emu_library(), "cpu", flags=A, scale=0.2)   # K = 1
    PC.run_oracle_vs_engine((64, 6, 64, 32, 2, 2, 2, 16), 3, 80, 42, emu_library(), "cpu", flags=A, scale=0.2)   # T % 32 == 16
    PC.run_oracle_vs_engine((64, 6, 64, 64, 3, 2, 2, 0), 1, 70, 43, emu_library(), "cpu", flags=A, scale=0.2)    # no upsampling, ragged T
    PC.run_oracle_vs_engine((64, 6, 64, 32, 1, 1, 2, 16), 1, 32, 44, emu_library(), "cpu", flags=A, scale=0.2)   # a single layer
    # kernel_size 3 (round 3: split forward block and chain kernel with the res-1x1 fragments read from the global weight
    # image, the three taps filling the LDS): dilations up to 64 = history taps up to 4 tiles back, with and without aux partials
    PC.run_oracle_vs_engine((64, 6, 64, 32, 7, 1, 3, 16), 1, 288, 46, emu_library(), "cpu", flags=A, scale=0.15)
    PC.run_oracle_vs_engine((64, 6, 64, 32, 2, 2, 3, 0), 2, 48, 47, emu_library(), "cpu", flags=0, scale=0.2)
    eng3 = WaveNetEngine(64, 6, 64, 32, 2, 2, 3, 16, device="cpu", library=emu_library())
    load_state_into_flat(eng3, O.random_params(O.OracleConfig(64, 6, 64, 32, 2, 2, 3, 16), 3, scale=0.1))
    x3, h3, t3 = O.synthetic_batch(O.OracleConfig(64, 6, 64, 32, 2, 2, 3, 16), 1, 64, 4)

    def step3():
        loss3, dl3 = eng3.forward_loss(x3, h3, t3)
        eng3.backward(dl3, t_first=eng3.receptive_field)
    log3 = PC.launch_log(emu_library(), step3)
    assert log3.get("fused_bwd_chain") == 3 and log3.get("fused_resblock_fwd") == 4 and log3.get("fused_bwd_dx") == 1, log3
    cfg_t = (64, 6, 64, 32, 3, 2, 2, 16)
    cfg = O.OracleConfig(*cfg_t)
    params, x, h, t, margin, sd = PC.pick_instance(cfg, 2, 96, 45, 0.2)
    res = []
    for flags in (A, A | NC, A):
        eng = WaveNetEngine(*cfg_t, device="cpu", library=emu_library())
        eng.flags = flags
        load_state_into_flat(eng, params)
        logits = eng.forward(x, h)
        loss, dl = eng.loss(logits, t)
        res.append(eng.backward(dl, layers_per_bucket=2).clone())
    assert torch.equal(res[0], res[2])
    assert float((res[0] - res[1]).abs().max()) <= 1e-5 * float(res[1].abs().max())


def test_split_contraction_interior_and_edge_blocks():
    """k_gemm6's epilogue has a fast path for blocks whose whole 256 x 128 tile lies inside the output (scalar row
    offsets, no range selects) and the general path for ragged edges: 256 skip channels and 6 layers give the post-net
    launches (bias, relu, mask epilogues) and the all-layer skip-gradient contraction (M = 320) one interior and one edge
    block each along both axes at T = 160."""
    from pytorchwavenetvocoder_amd import _lib
    PC.run_oracle_vs_engine((64, 6, 64, 256, 3, 2, 2, 16), 1, 160, 46, emu_library(), "cpu", flags=_lib.FLAG_AUX_FUSED, scale=0.1)


def test_gate_epilogues_of_the_wide_model_path():
    """n_resch % 128 == 0 (the recipes' 512): the any-size path runs the gate as the epilogue of the dilated contraction
    (sigmoid / tanh rows paired by the weight packing) and gate' as the epilogue of the dZ contraction.  R = 128 and 256
    (one and two 256-row blocks of the forward contraction), ragged T (edge blocks), last layer without a residual input,
    against the oracle."""
    PC.run_oracle_vs_engine((64, 6, 128, 128, 2, 2, 2, 16), 1, 144, 51, emu_library(), "cpu", scale=0.1)
    PC.run_oracle_vs_engine((32, 4, 256, 128, 1, 1, 2, 8), 1, 72, 52, emu_library(), "cpu", scale=0.1)    # two 256-row blocks, one layer
    PC.run_oracle_vs_engine((32, 4, 128, 64, 1, 1, 3, 0), 1, 70, 53, emu_library(), "cpu", scale=0.1)   # one layer, K = 3, no upsampling
    # the launches of a step really are the epilogue variants: no separate gate kernels, no plain dilated contraction
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    cfg_t = (64, 6, 128, 128, 2, 2, 2, 16)
    cfg = O.OracleConfig(*cfg_t)
    eng = WaveNetEngine(*cfg_t, device="cpu", library=emu_library())
    load_state_into_flat(eng, O.random_params(cfg, 1, scale=0.1))
    x, h, t = O.synthetic_batch(cfg, 1, 48, 2)

    def step():
        logits = eng.forward(x, h)
        loss, dl = eng.loss(logits, t)
        eng.backward(dl)
    log = PC.launch_log(emu_library(), step)
    assert log.get("fwd_dilated_gate") == 4 and log.get("bwd_dz_res_gate") == 3 and log.get("bwd_dz_skip_gate") == 1, log
    assert "gate_fwd" not in log and "gate_bwd" not in log and "fwd_dilated_layered" not in log, log
    # round 4: the per-layer weight sets of the wide path are split ONCE per step by the batched launch of pack_weights
    # (one gemm6_pack launch in the whole step: round 3 issued one per contraction, 149 per step at the recipe size)
    assert log.get("gemm6_pack") == 1, log


def test_launch_sequence_of_the_default_training_step():
    """What a default-mode step of a fused-kernel model launches (the host build's launch log): one forward block per
    layer, ONE all-layer skip-gradient contraction, the gate' head, L - 1 chain launches, the dX tail, the weight-image
    pack; and the former structure under WN_FLAG_NO_CHAIN."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    cfg_t = (64, 6, 64, 32, 3, 2, 2, 16)
    cfg = O.OracleConfig(*cfg_t)
    x, h, t = O.synthetic_batch(cfg, 1, 64, 2)
    logs = {}
    for flags in (_lib.FLAG_AUX_FUSED, _lib.FLAG_AUX_FUSED | _lib.FLAG_NO_CHAIN):
        eng = WaveNetEngine(*cfg_t, device="cpu", library=emu_library())
        eng.flags = flags
        load_state_into_flat(eng, O.random_params(cfg, 1, scale=0.1))

        def step():
            logits = eng.forward(x, h)
            loss, dl = eng.loss(logits, t)
            eng.backward(dl)
        logs[flags] = PC.launch_log(emu_library(), step)
    a, b = logs[_lib.FLAG_AUX_FUSED], logs[_lib.FLAG_AUX_FUSED | _lib.FLAG_NO_CHAIN]
    assert a["fused_resblock_fwd"] == 6 and a["fused_bwd_chain"] == 5 and a["fused_bwd_gate"] == 1 and a["fused_bwd_dx"] == 1, a
    assert a["bwd_dz_skip_all"] == 1 and a["fused_pack_images"] == 1 and "aux_bwd" not in a and a["aux_finish"] >= 1, a
    assert b["fused_bwd_gate"] == 6 and b["fused_bwd_dx"] == 6 and "fused_bwd_chain" not in b and "bwd_dz_skip_all" not in b, b


def test_gradient_bucket_events_follow_the_launches_that_fill_the_bucket():
    """N > 1 readiness (row e): wn_backward records bucket i's event AFTER the last launch that writes into bucket i and BEFORE
    the chain launches of the layers that follow, so the all-reduce of a finished bucket runs under the rest of the backward
    pass (distributed.GradientReducer waits on exactly these events).  Checked on the issue-order log for the three-bucket
    structure and for mid-chain splits (groups of layers)."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    cfg_t = (64, 6, 64, 32, 3, 2, 2, 16)   # 6 layers
    cfg = O.OracleConfig(*cfg_t)
    x, h, t = O.synthetic_batch(cfg, 1, 64, 2)
    eng = WaveNetEngine(*cfg_t, device="cpu", library=emu_library())
    load_state_into_flat(eng, O.random_params(cfg, 1, scale=0.1))
    chain_tags = ("fused_bwd_chain", "fused_bwd_gate", "fused_bwd_dx")
    for lpb, nb in ((0, 3), (3, 4), (2, 5)):
        loss, dl = eng.forward_loss(x, h, t)
        seq = PC.launch_sequence(emu_library(), lambda: eng.backward(dl, events=list(range(1, nb + 1)), layers_per_bucket=lpb,
                                                                      t_first=eng.receptive_field))
        ev = [i for i, s in enumerate(seq) if s == "bucket_event"]
        assert len(ev) == nb and ev[-1] == len(seq) - 1, (lpb, seq)
        chain_pos = [i for i, s in enumerate(seq) if s in chain_tags]
        # bucket 0 = post-net + skip: final before the residual chain starts, after dw_post2 / dw_post1 / dw_skip
        assert max(i for i, s in enumerate(seq) if s in ("dw_post2", "dw_post1", "dw_skip")) < ev[0] < chain_pos[0], (lpb, seq)
        # layer buckets: the k-th one is recorded after k * lpb layers' chain launches (head + chain + tail = L + 1 launches
        # for L layers: the tail belongs to layer 0) and before the next layer's launch
        per = lpb if lpb else 6
        for k in range(1, nb - 1):
            done = min(k * per, 6)
            n_launch_before = sum(1 for i in chain_pos if i < ev[k])
            assert n_launch_before == done + 1, (lpb, k, seq)
        # the front-conv / upsampling bucket is last
        assert any(s.startswith("dw_front") or s == "front_dw" for s in seq[ev[-2]:ev[-1]]), (lpb, seq[ev[-2]:])


def test_loss_window_backward_equals_the_full_backward():
    """wn_backward_window (ABI v5): with the loss on [:, rf:] (train.py:534-536) the post-net / skip part of the backward pass
    runs over [t0, T) only, t0 = rf rounded down to a 128-column tile.  rf = 128 here, so t0 = 128 > 0: same gradients as
    the full-range backward at round-off (a different split-K plan), against the oracle, dSkip exactly zero in front of
    the window, and the launch log shows its zero-fill -- for the chain mode and the exact-f32-MFMA kernels (the launch pair
    and the any-size path take the same windowed contractions: GPU test)."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    cfg_t = (32, 4, 64, 32, 7, 1, 2, 16)
    cfg = O.OracleConfig(*cfg_t)
    assert cfg.receptive_field == 128
    B, T = 1, 272
    params, x, h, t, margin, sd = PC.pick_instance(cfg, B, T, 61, 0.1)
    _, _, grads_ref = O.train_step(cfg, params, None, x, h, t)
    # (the launch pair WN_FLAG_NO_CHAIN takes the same windowed contractions as the chain mode: GPU test only)
    for flags in (_lib.FLAG_AUX_FUSED, _lib.FLAG_EXACT_MFMA):   # (any-size path: GPU test)
        eng = WaveNetEngine(*cfg_t, device="cpu", library=emu_library())
        eng.flags = flags
        load_state_into_flat(eng, params)
        logits = eng.forward(x, h)
        loss, dl = eng.loss(logits, t)
        assert float(dl[:, :, :128].abs().max()) == 0.0
        full = eng.backward(dl).clone()
        log = PC.launch_log(emu_library(), lambda: eng.backward(dl, t_first=cfg.receptive_field))
        win = eng.grads().clone()
        assert log.get("fill_cols") == 1, log   # dSkip; the chain kernel takes dZs as zero in front of the window without a fill
        dSk = eng.saved(_lib.WS_DSKIP)
        assert float(dSk[:, :, :128].abs().max()) == 0.0 and float(dSk[:, :, 128:].abs().max()) > 0.0
        scale = float(full.abs().max())
        assert float((win - full).abs().max()) <= 2e-6 * scale, (flags, float((win - full).abs().max()), scale)
        grads = PC.flat_to_state(eng, win, O.param_shapes(cfg))
        for k, ref in grads_ref.items():
            if ref is not None:
                assert PC.rel_to_max(grads[k], ref) <= PC.TOL_GRAD, (flags, k)
    # t_first below one tile, or WN_LOSS_WINDOW=0 semantics (t_first = 0): the plain backward, bit for bit
    eng.flags = _lib.FLAG_AUX_FUSED
    with pytest.raises(_lib.WnError):   # the forward in the workspace is the exact-MFMA family's
        eng.backward(dl)
    eng.forward(x, h)
    a = eng.backward(dl).clone()
    b = eng.backward(dl, t_first=127).clone()
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


def test_cross_entropy_as_the_epilogue_of_conv_post_2():
    """wn_forward_loss (ABI v5): with a softmax head of 128..256 classes the loss is the epilogue of the conv_post_2
    contraction (k_gemm6: max / sum of exponentials across lane halves and the two wave rows of a block, logits never
    written).  Same loss and dlogits as wn_forward + wn_softmax_ce_loss and as the oracle; ragged last 128-column tile;
    targets of every class range; 200 classes (the second wave row holds part of them) and 256; fallback
    (64 classes: wn_forward_loss_fused == 0) through the logits scratch; the launch log shows no softmax_ce launch."""
    import ctypes
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    for cfg_t, B, T, fused in (((256, 6, 64, 128, 2, 1, 2, 16), 2, 144, 1), ((200, 4, 64, 128, 2, 1, 2, 0), 1, 150, 1),
                                ((64, 4, 64, 128, 2, 1, 2, 8), 1, 72, 0)):
        cfg = O.OracleConfig(*cfg_t)
        params = O.random_params(cfg, 71, scale=0.3)
        x, h, t = O.synthetic_batch(cfg, B, T, 72)
        eng = WaveNetEngine(*cfg_t, device="cpu", library=emu_library())
        load_state_into_flat(eng, params)
        assert eng.lib.wn_forward_loss_fused(ctypes.byref(eng.cfg), B, T, eng.flags) == fused
        logits = eng.forward(x, h)
        loss0, dl0 = eng.loss(logits, t, grad_scale=0.5)
        log = PC.launch_log(emu_library(), lambda: eng.forward_loss(x, h, t, grad_scale=0.5))
        loss1, dl1 = eng.forward_loss(x, h, t, grad_scale=0.5)
        assert ("fwd_post2_ce" in log) == bool(fused) and ("softmax_ce" in log) == (not fused), log
        rf = cfg.receptive_field
        assert float(dl1[:, :, :rf].abs().max()) == 0.0
        assert abs(float(loss1) - float(loss0)) <= 2e-6 * max(1.0, abs(float(loss0)))
        # p = e / sum vs exp(v - lse): a few ulp of a probability near 1, seen through p - 1
        assert float((dl1 - dl0).abs().max()) <= 1e-5 * float(dl0.abs().max()), float((dl1 - dl0).abs().max())
        loss_ref, logits_ref, _ = O.train_step(cfg, params, None, x, h, t)
        assert abs(float(loss1) - float(loss_ref)) <= PC.TOL_LOSS
        # another loss window and no gradient buffer
        loss2, none = eng.forward_loss(x, h, t, t_start=rf + 5, want_grad=False)
        loss3, _ = eng.loss(logits, t, t_start=rf + 5, want_grad=False)
        assert none is None and abs(float(loss2) - float(loss3)) <= 2e-6 * max(1.0, abs(float(loss3)))
        # the backward pass after forward_loss sees the same saved activations as after forward
        g1 = eng.backward(dl1, t_first=rf).clone()
        eng.forward(x, h)
        g0 = eng.backward(dl0, t_first=rf).clone()
        assert float((g1 - g0).abs().max()) <= 5e-6 * float(g0.abs().max())


def test_windowed_forward_loss_workspace_and_the_parameter_guard():
    """wn_forward_loss runs the skip sum / post-net over the loss window only (ABI v7 contract): without
    WN_FLAG_WS_FINITE it zero-fills relu(skip) / relu(post1) in front of the window, so a backward pass with an EARLIER
    window start (wn_backward: t_first = 0) multiplies dlogits == 0 with zeros, not with whatever the workspace held
    (here: NaN everywhere) -- same gradients as the windowed backward, which is the engine's default after forward_loss.
    And the guards of the Python engine: parameters modified between forward and backward raise (like torch.autograd),
    repack=True rebuilds the packed weight sets instead (WN_FLAG_REPACK)."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    cfg_t = (256, 4, 64, 128, 7, 1, 2, 16)
    cfg = O.OracleConfig(*cfg_t)
    rf = cfg.receptive_field
    assert rf == 128
    B, T = 1, 272
    params = O.random_params(cfg, 91, scale=0.2)
    x, h, t = O.synthetic_batch(cfg, B, T, 92)
    eng = WaveNetEngine(*cfg_t, device="cpu", library=emu_library())
    load_state_into_flat(eng, params)
    eng.workspace(B, T).fill_(float("nan"))
    eng.ws_finite = False
    out = {}
    log = PC.launch_log(emu_library(), lambda: out.update(r=eng.forward_loss(x, h, t)))
    loss, dl = out["r"]
    assert log.get("fwd_post2_ce") == 1 and log.get("fill_cols") == 3, log   # dlogits, relu(skip), relu(post1)
    assert float(eng.saved(_lib.WS_RELU_SKIP)[:, :, :rf].abs().max()) == 0.0
    assert float(eng.saved(_lib.WS_RELU_POST1)[:, :, :rf].abs().max()) == 0.0
    g_win = eng.backward(dl).clone()               # default window: the forward's
    g_full = eng.backward(dl, t_first=0).clone()   # what wn_backward does
    assert bool(torch.isfinite(g_win).all()) and bool(torch.isfinite(g_full).all())
    assert float((g_win - g_full).abs().max()) <= 2e-6 * float(g_full.abs().max())
    loss_ref, _, grads_ref = O.train_step(cfg, params, None, x, h, t)
    assert abs(float(loss) - float(loss_ref)) <= PC.TOL_LOSS
    grads = PC.flat_to_state(eng, g_win, O.param_shapes(cfg))
    for k, ref in grads_ref.items():
        if ref is not None:
            assert PC.rel_to_max(grads[k], ref) <= PC.TOL_GRAD, k
    # the caller vouches for a workspace that is NOT finite: documents what the flag means (no fill launches)
    eng.ws_finite = True
    log = PC.launch_log(emu_library(), lambda: out.update(r=eng.forward_loss(x, h, t)))
    assert log.get("fill_cols") == 1, log
    loss, dl = out["r"]   # (the gradient tensor of the LAST loss call is the one backward() knows the bound of: fp16 pair split)
    # parameter guard
    eng.flat_params[:4] += 0.0    # an in-place write, whatever its value
    with pytest.raises(_lib.WnError):
        eng.backward(dl)
    g_re = eng.backward(dl, repack=True).clone()
    assert torch.equal(g_re, g_win)
    eng.forward_loss(x, h, t)
    eng.adam_step(torch.zeros_like(eng.flat_params), torch.zeros_like(eng.flat_params), 1, 0.0)   # in-library update
    with pytest.raises(_lib.WnError):
        eng.backward(dl)


def test_any_size_decode_as_one_persistent_launch():
    """csrc/wn_dlp.hip on the emulator's cooperative launch (every workgroup alive at once; they hand their vectors to each
    other as tagged granules): a 32-channel model = 2 workgroups, kernel_size 2 and 3, one utterance, three ragged ones and
    4 (a full column block) through ``layered="granules"`` (by default batches of 2 and more go to wn_dlpf.hip where it covers
    the model, next test; with the granule hand-off to wn_dlpm.hip from 5 on) -- logits within 1e-4 of the queue algorithm (oracle; the res 1x1 is folded into the next layer's
    newest tap, so the rounding differs from the layer-wise launches), tokens equal to the oracle's and to the launches', the
    launch log shows ONE dlp_steps launch per chunk and no layer-wise launch, and inverse-CDF sampling on the same draws picks
    the same tokens as the launches."""
    import numpy as np
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd.nets import WaveNet
    for K, B in ((2, 1), (3, 3), (2, 4)):
        cfg_t = (32, 4, 32, 32, 3, 2, K, 4)
        cfg = O.OracleConfig(*cfg_t)
        params = O.random_params(cfg, 9 + K, scale=0.3)
        model = WaveNet(*cfg_t, _library=emu_library())
        model.load_state_dict(params)   # (a size the one-workgroup kernel covers as well: layered=True selects the any-size path)
        rs = np.random.RandomState(12 + B)
        xs = torch.from_numpy(rs.randint(0, 32, (B, 5))).long()
        hs = torch.from_numpy(rs.standard_normal((B, 4, 8)).astype(np.float32))
        ns = [9 - (b % 3) for b in range(B)]
        out = {}
        log = PC.launch_log(emu_library(), lambda: out.update(p=model.engine.decode(xs, hs, ns, chunk=4, return_logits=True, layered="granules")))
        assert log.get("dlp_steps", 0) >= 2 and "dl_dilated" not in log and "dl_res" not in log, log
        tp, lp = out["p"]
        tl, ll = model.engine.decode(xs, hs, ns, chunk=4, return_logits=True, layered="launches")
        for b in range(B):
            rt, rl = O.fast_generate(cfg, params, xs[b:b + 1], hs[b:b + 1], ns[b], return_logits=True)
            assert float((lp[b] - rl).abs().max()) <= 1e-4, (K, B, b)
            top2 = rl.topk(2, dim=1).values
            safe = ((top2[:, 0] - top2[:, 1]) > 1e-3).numpy()
            assert (tp[b].numpy()[safe] == np.asarray(rt)[safe]).all() and (tp[b].numpy()[safe] == tl[b].numpy()[safe]).all(), (K, B, b)
        if B == 3:
            torch.manual_seed(5)
            sp = model.engine.decode(xs, hs, ns, mode="sampling", layered="granules")
            torch.manual_seed(5)
            sl = model.engine.decode(xs, hs, ns, mode="sampling", layered="launches")
            assert all(torch.equal(a, b) for a, b in zip(sp, sl))


def test_any_size_decode_of_wide_batches_on_the_matrix_cores():
    """csrc/wn_dlpf.hip and wn_dlpm.hip (batches of 5 .. 48 utterances: 16-row sets x 16-column blocks on v_mfma_f32_16x16x4_f32,
    8 channels per workgroup, one set of workgroups per column block; hand-off by flags + plain vectors / by granules) on the
    emulator's cooperative launch: a 32-channel model = 4 workgroups
    per block, kernel_size 2 with 12 ragged utterances (one ragged column block) and kernel_size 3 with 19 (a full block and a
    block of 3: 8 workgroups) -- logits within 1e-4 of the queue
    algorithm (oracle), tokens equal to the oracle's and to the layer-wise launches', ONE dlpm_steps launch per chunk and no
    layer-wise launch in the log, inverse-CDF sampling on the same draws equal to the launches'."""
    import numpy as np
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd.nets import WaveNet
    for K, B in ((2, 12), (3, 19)):
        cfg_t = (32, 4, 32, 32, 3, 2, K, 4)
        cfg = O.OracleConfig(*cfg_t)
        params = O.random_params(cfg, 9 + K, scale=0.3)
        model = WaveNet(*cfg_t, _library=emu_library())
        model.load_state_dict(params)
        rs = np.random.RandomState(12 + B)
        xs = torch.from_numpy(rs.randint(0, 32, (B, 5))).long()
        hs = torch.from_numpy(rs.standard_normal((B, 4, 8)).astype(np.float32))
        ns = [6 - (b % 3) for b in range(B)]
        out = {}
        log = PC.launch_log(emu_library(), lambda: out.update(p=model.engine.decode(xs, hs, ns, chunk=4, return_logits=True, layered=True)))
        # (both plans are of the class wn_dlpf.hip covers: plain vectors + one flag per unit, inputs by global -> LDS transfers,
        # shared queue rings)
        assert log.get("dlpf_steps", 0) >= 2 and "dlp_steps" not in log and "dl_dilated" not in log and "dl_res" not in log, log
        # the same through the granule kernel (wn_dlpm.hip: the recipes' kernel_size 3 class, and the A/B partner)
        log2 = PC.launch_log(emu_library(), lambda: out.update(g=model.engine.decode(xs, hs, ns, chunk=4, return_logits=True, layered="granules")))
        assert log2.get("dlpm_steps", 0) >= 2 and "dlpf_steps" not in log2 and "dl_dilated" not in log2, log2
        for b in range(B):
            assert float((out["g"][1][b] - out["p"][1][b]).abs().max()) <= 1e-5 and torch.equal(out["g"][0][b], out["p"][0][b]), (K, B, b)
        tp, lp = out["p"]
        tl, ll = model.engine.decode(xs, hs, ns, chunk=4, return_logits=True, layered="launches")
        for b in range(B):
            rt, rl = O.fast_generate(cfg, params, xs[b:b + 1], hs[b:b + 1], ns[b], return_logits=True)
            assert float((lp[b] - rl).abs().max()) <= 1e-4, (K, B, b)
            top2 = rl.topk(2, dim=1).values
            safe = ((top2[:, 0] - top2[:, 1]) > 1e-3).numpy()
            assert (tp[b].numpy()[safe] == np.asarray(rt)[safe]).all() and (tp[b].numpy()[safe] == tl[b].numpy()[safe]).all(), (K, B, b)
        if K == 2:
            torch.manual_seed(5)
            sp = model.engine.decode(xs, hs, ns, mode="sampling", layered=True)
            torch.manual_seed(5)
            sl = model.engine.decode(xs, hs, ns, mode="sampling", layered="launches")
            assert all(torch.equal(a, b) for a, b in zip(sp, sl))


def test_any_size_decode_of_more_utterances_than_one_persistent_launch_takes():
    """70 utterances (one persistent launch of the flag hand-off kernel takes 64 = four column blocks): the engine sends the batch
    through the persistent launch in groups (64 + 6) -- tokens equal to the layer-wise launches' on the whole batch, in the sampling mode too (the draws are made for
    the whole batch before it is cut)."""
    import numpy as np
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd.nets import WaveNet
    cfg_t = (32, 4, 32, 32, 3, 2, 2, 4)
    cfg = O.OracleConfig(*cfg_t)
    model = WaveNet(*cfg_t, _library=emu_library())
    model.load_state_dict(O.random_params(cfg, 11, scale=0.3))
    B = 70
    rs = np.random.RandomState(3)
    xs = torch.from_numpy(rs.randint(0, 32, (B, 5))).long()
    hs = torch.from_numpy(rs.standard_normal((B, 4, 8)).astype(np.float32))
    ns = [3 - (b % 2) for b in range(B)]
    assert model.engine._persistent_groups(B, True, "argmax") == [(0, 64), (64, 70)]
    out = {}
    log = PC.launch_log(emu_library(), lambda: out.update(p=model.engine.decode(xs, hs, ns, layered=True)))
    assert log.get("dlpf_steps", 0) == 2 and "dl_dilated" not in log, log
    tl = model.engine.decode(xs, hs, ns, layered="launches")
    assert all(torch.equal(a, b) for a, b in zip(out["p"], tl))
    torch.manual_seed(5)
    sp = model.engine.decode(xs, hs, ns, mode="sampling", layered=True)
    torch.manual_seed(5)
    sl = model.engine.decode(xs, hs, ns, mode="sampling", layered="launches")
    assert all(torch.equal(a, b) for a, b in zip(sp, sl))


def test_any_size_decode_flag_hand_off_kernel_size_3_class():
    """The kernel_size 3 class of csrc/wn_dlpf.hip (64 tile steps per wave: (K + 1) n_resch > 1536 -- the ljspeech recipes'
    n_resch 512 / kernel_size 3; no room in LDS for the x / skip set's own copy of z, it reads the gate set's order): 416 channels,
    2 layers, 2 utterances on the emulator's cooperative launch (52 workgroups) -- logits within 1e-5 of the queue algorithm
    (oracle), tokens equal to the oracle's away from near-ties.  (Against the layer-wise launches at the recipes' own size: GPU
    suite and tools/decode_equivalence_soak.py.)"""
    import numpy as np
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd.nets import WaveNet
    cfg_t = (32, 4, 416, 32, 2, 1, 3, 4)
    cfg = O.OracleConfig(*cfg_t)
    params = O.random_params(cfg, 13, scale=0.05)
    model = WaveNet(*cfg_t, _library=emu_library())
    model.load_state_dict(params)
    rs = np.random.RandomState(3)
    xs = torch.from_numpy(rs.randint(0, 32, (2, 4))).long()
    hs = torch.from_numpy(rs.standard_normal((2, 4, 4)).astype(np.float32))
    ns = [3, 2]
    out = {}
    log = PC.launch_log(emu_library(), lambda: out.update(p=model.engine.decode(xs, hs, ns, return_logits=True, layered=True)))
    assert log.get("dlpf_steps", 0) == 1 and "dl_dilated" not in log and "dlpm_steps" not in log, log
    for b in range(2):
        rt, rl = O.fast_generate(cfg, params, xs[b:b + 1], hs[b:b + 1], ns[b], return_logits=True)
        assert float((out["p"][1][b] - rl).abs().max()) <= 1e-5, b
        top2 = rl.topk(2, dim=1).values
        safe = ((top2[:, 0] - top2[:, 1]) > 1e-3).numpy()
        assert (out["p"][0][b].numpy()[safe] == np.asarray(rt)[safe]).all(), b


def test_front_conv_weight_gradient_on_the_matrix_cores():
    """k_front_dw_mfma (R = 32 or a multiple of 64, K * Q <= 1024): the front conv's weight gradient as a contraction over time with a
    one-hot B operand built from the token indices, instead of LDS float atomics.  256 classes x 2 taps (all 16 column
    tiles), a chunk that ends inside a 32-step iteration (T % 32 != 0), two sequences, R = 32 (one row tile) and
    kernel_size 1 / 3 tables, and 256 classes x 3 taps (24 column tiles: the first 8 waves take two; the configs[3] front
    conv) -- every gradient against the oracle (the front conv's is causal.weight / causal.bias)."""
    for cfg_t, B, T, seed in (((256, 4, 64, 32, 2, 1, 2, 8), 2, 72, 81), ((128, 4, 32, 32, 2, 1, 3, 0), 1, 45, 82),
                              ((256, 4, 64, 32, 2, 1, 1, 8), 1, 40, 83), ((256, 4, 64, 32, 2, 1, 3, 8), 1, 48, 84),
                              ((256, 4, 128, 32, 2, 1, 2, 8), 2, 40, 85)):   # 128 channels: two groups of 64 rows (blockIdx.z)
        log = {}

        def run():
            e, g = PC.run_oracle_vs_engine(cfg_t, B, T, seed, emu_library(), "cpu", scale=0.2)
            log["err"] = (e, g)
        counts = PC.launch_log(emu_library(), run)
        assert counts.get("dw_front_scatter") == 1 and "dw_front_onehot" not in counts, counts


def test_transpose_op_on_the_emulator():
    """wn_op_transpose_last2 ((B, R, C) -> (B, C, R), the autograd bridge's layout change): bit-exact, ragged tiles."""
    lib = emu_library()
    for B, R, C in ((2, 37, 70), (1, 64, 32)):
        x = torch.randn(B, R, C)
        y = torch.empty(B, C, R)
        lib.check(lib.wn_op_transpose_last2(x.data_ptr(), y.data_ptr(), B, R, C, None), "wn_op_transpose_last2")
        assert torch.equal(y, x.transpose(1, 2).contiguous())
    assert lib.wn_op_transpose_last2(x.data_ptr(), x.data_ptr(), 1, 4, 4, None) != 0   # in place: refused


def test_persistent_decode_dispatch_rules():
    """Which path a batch takes at the recipes' size (n_resch 512; no launch: plan queries only): one persistent launch up to 64
    utterances (flag hand-off; 48 with the granule hand-off), two groups up to 128, the layer-wise launches beyond and for explicit requests; the kernel_size 3 class likewise;
    1024 channels are outside the compiled classes (launches at any batch)."""
    import ctypes
    from pytorchwavenetvocoder_amd import _lib as L
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine
    lib = emu_library()
    for K in (2, 3):
        eng = WaveNetEngine(256, 80, 512, 256, 10, 3, K, 80, device="cpu", library=lib)
        cfg = ctypes.byref(eng.cfg)
        assert not eng.decode_supported()
        for B in (1, 2, 5, 16, 17, 48, 49, 64):
            assert lib.wn_decode_layered_error_offset(cfg, B, 0) >= 0, (K, B)      # one persistent launch
            assert eng._persistent_groups(B, None, "argmax") is None
        assert lib.wn_decode_layered_error_offset(cfg, 65, 0) < 0
        assert lib.wn_decode_layered_error_offset(cfg, 49, L.DECODE_GRANULES) < 0  # the granule kernels take 48
        assert eng._persistent_groups(65, None, "argmax") == [(0, 64), (64, 65)]
        assert eng._persistent_groups(128, True, "sampling") == [(0, 64), (64, 128)]
        assert eng._persistent_groups(129, None, "argmax") is None                 # three groups: the launches are as fast
        assert eng._persistent_groups(64, "launches", "argmax") is None
        assert eng._persistent_groups(64, None, "mol") is None
    # 1024 channels: (K + 1) n_resch exceeds every compiled class -- layer-wise launches at any batch
    big = WaveNetEngine(256, 80, 1024, 256, 10, 3, 2, 80, device="cpu", library=lib)
    assert lib.wn_decode_layered_error_offset(ctypes.byref(big.cfg), 1, 0) < 0 and big._persistent_groups(64, None, "argmax") is None
    small = WaveNetEngine(256, 80, 64, 256, 10, 3, 2, 80, device="cpu", library=lib)
    assert small.decode_supported() and small._persistent_groups(64, None, "argmax") is None   # the one-workgroup kernel takes it


def test_persistent_decode_residency_check_and_fall_backs(monkeypatch):
    """Every workgroup of the persistent decode launch waits for the others, so all of them must be resident at once.  The
    library asks the device (occupancy x CUs; here the test knob WN_COOP_CAPACITY stands in for a small / partitioned part)
    where it chooses the path: a grid that does not fit decodes by layer-wise launches, same tokens.  And the other way a
    launch can fail -- resident in principle, but another kernel held the CUs, the bounded polls time out, the error word in
    the state is set --: the next launch returns at once, the engine raises WnDecodeTimeout inside, warns and decodes again by
    layer-wise launches with the same draws."""
    import ctypes
    import warnings
    import numpy as np
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib as L
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine
    from pytorchwavenetvocoder_amd.nets import WaveNet
    lib = emu_library()
    big = WaveNetEngine(256, 80, 512, 256, 10, 3, 2, 80, device="cpu", library=lib)
    assert big.decode_residency(2) == (True, 64, 0x7fffffff) and big.decode_residency(48)[:2] == (True, 192)
    assert big.decode_residency(1)[:2] == (True, 128) and big.decode_residency(64)[:2] == (True, 256) and big.decode_residency(65)[:2] == (False, 0)
    monkeypatch.setenv("WN_COOP_CAPACITY", "100")
    assert big.decode_residency(2) == (True, 64, 100) and big.decode_residency(17) == (False, 128, 100)
    assert big.decode_residency(1) == (False, 128, 100)
    assert lib.wn_decode_layered_error_offset(ctypes.byref(big.cfg), 17, 0) < 0
    assert big._persistent_groups(40, None, "argmax") is None                    # three groups: the launches
    assert big._persistent_groups(30, None, "argmax") == [(0, 16), (16, 30)]     # groups of what DOES fit
    monkeypatch.delenv("WN_COOP_CAPACITY")

    cfg_t = (32, 4, 32, 32, 3, 2, 2, 4)
    cfg = O.OracleConfig(*cfg_t)
    model = WaveNet(*cfg_t, _library=lib)
    model.load_state_dict(O.random_params(cfg, 11, scale=0.3))
    rs = np.random.RandomState(3)
    B = 3
    xs = torch.from_numpy(rs.randint(0, 32, (B, 5))).long()
    hs = torch.from_numpy(rs.standard_normal((B, 4, 8)).astype(np.float32))
    ns = [6, 5, 4]
    out = {}
    log = PC.launch_log(lib, lambda: out.update(p=model.engine.decode(xs, hs, ns, chunk=3, return_logits=True, layered=True)))
    assert log.get("dlpf_steps", 0) >= 2 and "dl_dilated" not in log, log
    # (1) a device that keeps fewer workgroups resident than the launch needs: layer-wise launches, chosen by the library
    monkeypatch.setenv("WN_COOP_CAPACITY", "3")
    log = PC.launch_log(lib, lambda: out.update(c=model.engine.decode(xs, hs, ns, chunk=3, return_logits=True, layered=True)))
    assert "dlpf_steps" not in log and "dlp_steps" not in log and log.get("dl_dilated", 0) > 0, log
    monkeypatch.delenv("WN_COOP_CAPACITY")
    for b in range(B):
        assert torch.equal(out["c"][0][b], out["p"][0][b]) and float((out["c"][1][b] - out["p"][1][b]).abs().max()) <= 1e-5
    # (2) a time-out: the error word is set before the first persistent launch of the decode
    real = lib.lib.wn_decode_layered_steps
    calls = []

    def steps(cfgp, Bn, params, G, F, n_pad, samples, Ttot, tf, te, p0, p1, state, nst, uni, lo, mode, wave, lsm, st):
        if not (mode & L.DECODE_BY_LAUNCHES):
            eoff = lib.wn_decode_layered_error_offset(cfgp, Bn, mode & L.DECODE_GRANULES)
            assert eoff >= 0
            ctypes.c_int.from_address(state.value + 4 * eoff).value = 1
            calls.append("persistent")
        else:
            calls.append("launches")
        return real(cfgp, Bn, params, G, F, n_pad, samples, Ttot, tf, te, p0, p1, state, nst, uni, lo, mode, wave, lsm, st)
    lib.wn_decode_layered_steps = steps
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            tf_, lf = model.engine.decode(xs, hs, ns, chunk=3, return_logits=True, layered=True)
            assert any("layer-wise launches" in str(x.message) for x in w)
        assert calls[0] == "persistent" and calls.count("persistent") == 1 and calls.count("launches") >= 2, calls
        for b in range(B):
            assert torch.equal(tf_[b], out["p"][0][b]) and float((lf[b] - out["p"][1][b]).abs().max()) <= 1e-5
        torch.manual_seed(5)
        del calls[:]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sf = model.engine.decode(xs, hs, ns, mode="sampling", layered=True)
        assert calls[0] == "persistent"
    finally:
        del lib.wn_decode_layered_steps
    torch.manual_seed(5)
    sp = model.engine.decode(xs, hs, ns, mode="sampling", layered=True)
    assert all(torch.equal(a, b) for a, b in zip(sf, sp))


def test_weight_gradients_by_the_fp16_pair_split_and_their_overflow_redo():
    """WN_FLAG_DW_F16PAIR (csrc/wn_gemm6.hip k_gemm6_dw<.., F16>): two fp16 pieces per operand, three products, the gradient
    operand scaled by 2^(e + 8) where 2^-e bounds the MEASURED max |dlogits| (ABI v9: by the loss call, or by a scan of the tensor
    given to backward).  Against the six-bf16-product mode: within 1e-6 of the largest gradient (the three-bf16-product mode:
    ~2e-6 ... 1e-5); every fp16 launch is followed by ONE conditional six-product launch; a PROMISED bound far too small drives the
    scaled gradients out of fp16's range -> the overflow word -> the redo launches do the work: the default's result bit for
    bit; a gradient tensor the loss call did not make is scanned (same maximum, same bits); a gradient 2^-30 smaller than the loss
    call's keeps its accuracy (measured scale) where a merely SAFE promise (|g| <= 1) underflows fp16 entirely.
    (The golden gates, incl. the weights after Adam, are GPU tests: tests/test_gpu_dw_f16pair.py.)"""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    assert [(_lib.dw_f16_exp(b) >> _lib.DW_F16_EXP_SHIFT) & 63 for b in (3.0, 1.0, 0.5, 0.3, 2.0 ** -17, 6e-6, 1e-30)] == [0, 0, 1, 1, 17, 17, 63]
    assert all(_lib.dw_f16_exp(b) & _lib.FLAG_DW_F16_EXP_VALID for b in (3.0, 1e-30))
    cfg_t = (32, 4, 64, 32, 7, 1, 2, 16)
    cfg = O.OracleConfig(*cfg_t)
    B, T = 1, 272
    params, x, h, t, margin, sd = PC.pick_instance(cfg, B, T, 61, 0.1)
    _, _, grads_ref = O.train_step(cfg, params, None, x, h, t)
    eng = WaveNetEngine(*cfg_t, device="cpu", library=emu_library())
    load_state_into_flat(eng, params)
    res, logs = {}, {}
    for name, flags in (("six", _lib.FLAG_AUX_FUSED), ("f16", _lib.FLAG_AUX_FUSED | _lib.FLAG_DW_F16PAIR)):
        eng.flags = flags
        loss, dl = eng.forward_loss(x, h, t)
        logs[name] = PC.launch_log(emu_library(), lambda: eng.backward(dl))
        res[name] = eng.grads().clone()
        grads = PC.flat_to_state(eng, res[name], O.param_shapes(cfg))
        for k, ref in grads_ref.items():
            if ref is not None:
                assert PC.rel_to_max(grads[k], ref) <= PC.TOL_GRAD, (name, k)
    n_dw = sum(v for k, v in logs["six"].items() if k.startswith("dw_") and k != "dw_front_scatter")
    assert "dw_redo_if_overflow" not in logs["six"] and logs["f16"].get("dw_redo_if_overflow") == n_dw, logs
    scale = float(res["six"].abs().max())
    err = float((res["f16"] - res["six"]).abs().max())
    assert 0.0 < err <= 1e-6 * scale, (err, scale)
    # overflow: the promise is 2^20 too small for this gradient
    assert torch.equal(eng.backward(dl, dlogits_bound=2.0 ** -40).clone(), res["six"])
    # not the tensor the loss call returned: the library scans it -- the same maximum the loss epilogue measured, the same bits
    log = PC.launch_log(emu_library(), lambda: eng.backward(dl.clone()))
    assert log.get("dw_absmax_scan") == 1 and "dw_absmax_scan" not in logs["f16"], log
    assert torch.equal(eng.grads(), res["f16"])
    # underflow: the same gradient 2^-30 smaller.  Measured scale: the accuracy of the mode; a bound that is merely
    # safe (|g| <= 1, i.e. what a forgotten exponent meant in ABI v8) puts every scaled element below fp16's smallest subnormal
    small = dl * 2.0 ** -30
    g_scan = eng.backward(small).clone() * 2.0 ** 30
    assert float((g_scan - res["six"]).abs().max()) <= 1e-6 * scale
    g_loose = eng.backward(small, dlogits_bound=1.0).clone() * 2.0 ** 30
    assert float((g_loose - res["six"]).abs().max()) > 1e-2 * scale
    # an all-zero gradient has no maximum: the redo is forced, the result is exact zeros
    z = eng.backward(torch.zeros_like(dl)).clone()
    assert float(z.abs().max()) == 0.0
    # (the word is cleared per call, in-place modified gradients, the caller's own bound: tests/test_gpu_dw_f16pair.py)


def test_skip_gradient_on_the_256_row_tile():
    """n_skipch = 256 with L * n_resch >= 512: the skip weight gradient (256 x 512, k = every position) takes the 256 x 128 tile
    k_gemm6_dw<4,2> (wn_gemm6_dw_tall, since round 5 from 256 output rows on: z of every layer is read once), in the default
    arithmetic (fp16 pair split) and with six bf16 products, ragged k-chunks and the interior fast pass -- every gradient
    against the oracle."""
    from pytorchwavenetvocoder_amd import _lib
    cfg_t = (32, 4, 64, 256, 4, 2, 2, 16)
    for flags in (None, _lib.FLAG_AUX_FUSED):
        e, g = PC.run_oracle_vs_engine(cfg_t, 1, 400, 71, emu_library(), "cpu", flags=flags, scale=0.1)
        assert g <= 2e-5, (flags, g)


def test_skip_and_res_weight_gradients_in_one_launch():
    """k_dw_skipres8 (csrc/wn_gemm6.hip, WnDwSkipRes): with the fp16 pair split and ONE layer bucket the skip_1x1 and res_1x1 weight
    gradients of every layer come from one launch that reads z once (reference: wavenet.py:534-536 backward).  Five layers (the
    last column tile is half empty, the last layer's res_1x1 is dead), 512 skip channels (two row blocks: the second one has no
    dX product), k-chunks of 10 and 9 steps (pipelined pass of 9 + a single step): every gradient against the oracle; the two
    separate launches (taken when the layers are flushed in several buckets) agree with it to fp32 summation order; a promise far
    too small raises the overflow word and the redo launches (six products, the fused launch's split-K plan) do the work."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    cfg_t = (32, 4, 64, 512, 5, 1, 2, 16)
    cfg = O.OracleConfig(*cfg_t)
    B, T = 1, 304   # (one sequence: 1024 ReLU inputs per position, and a kink-free instance must still exist)
    params, x, h, t, margin, sd = PC.pick_instance(cfg, B, T, 91, 0.1, tries=400)
    _, _, grads_ref = O.train_step(cfg, params, None, x, h, t)
    eng = WaveNetEngine(*cfg_t, device="cpu", library=emu_library())
    load_state_into_flat(eng, params)
    eng.flags = _lib.FLAG_AUX_FUSED | _lib.FLAG_DW_F16PAIR
    loss, dl = eng.forward_loss(x, h, t)
    res, logs = {}, {}
    for name, lpb in (("fused", 0), ("buckets", 2)):
        logs[name] = PC.launch_log(emu_library(), lambda: eng.backward(dl, layers_per_bucket=lpb))
        res[name] = eng.grads().clone()
        grads = PC.flat_to_state(eng, res[name], O.param_shapes(cfg))
        for k, ref in grads_ref.items():
            if ref is not None:
                assert PC.rel_to_max(grads[k], ref) <= PC.TOL_GRAD, (name, k)
    assert logs["fused"].get("dw_skip_res") == 1 and "dw_skip" not in logs["fused"] and "dw_res" not in logs["fused"], logs["fused"]
    assert "dw_skip_res" not in logs["buckets"] and logs["buckets"].get("dw_skip") == 1 and logs["buckets"].get("dw_res") == 3, logs["buckets"]
    scale = float(res["buckets"].abs().max())
    assert float((res["fused"] - res["buckets"]).abs().max()) <= 1e-6 * scale
    # the dead res_1x1 of the last layer stays exactly zero
    lo, hi = eng.dead_range
    assert float(res["fused"][lo:hi].abs().max()) == 0.0
    # bucket events (distributed.GradientReducer): with one layer bucket the skip_1x1 tensors are part of THAT bucket
    # (wn_bucket_range), so the head bucket (post-net) is still final before the chain starts, and the layer bucket's event follows
    # the fused launch and its reductions
    seq = PC.launch_sequence(emu_library(), lambda: eng.backward(dl, events=[1, 2, 3], layers_per_bucket=0))
    ev = [i for i, s_ in enumerate(seq) if s_ == "bucket_event"]
    assert len(ev) == 3 and ev[-1] == len(seq) - 1, seq
    fused_at = seq.index("dw_skip_res")
    chain_at = [i for i, s_ in enumerate(seq) if s_ in ("fused_bwd_chain", "fused_bwd_gate", "fused_bwd_dx")]
    assert max(seq.index("dw_post2"), seq.index("dw_post1")) < ev[0] < chain_at[0], seq
    assert chain_at[-1] < fused_at < ev[1] and "reduce_partials" in seq[fused_at:ev[1]], seq
    ranges = eng.bucket_ranges(0)
    skip_lo, _ = eng.param_slice(_lib.P_SKIP_W, 0)
    assert ranges[0] == (0, skip_lo) and ranges[1][0] == skip_lo and ranges[-1][1] == eng.n_params, ranges
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])), ranges
    many = eng.bucket_ranges(2)   # several layer buckets: the skip_1x1 tensors stay in the head bucket
    assert many[0][0] == 0 and many[0][1] > skip_lo and all(a[1] == b[0] for a, b in zip(many, many[1:])) and many[-1][1] == eng.n_params, many
    # overflow -> the conditional six-product launches behind the fused one: the six-product mode's arithmetic
    eng.flags = _lib.FLAG_AUX_FUSED
    six = eng.backward(dl).clone()
    eng.flags = _lib.FLAG_AUX_FUSED | _lib.FLAG_DW_F16PAIR
    gov = eng.backward(dl, dlogits_bound=2.0 ** -40).clone()
    assert bool(torch.isfinite(gov).all()) and float((gov - six).abs().max()) <= 2e-7 * scale
    assert 0.0 < float((res["fused"] - six).abs().max()) <= 1e-6 * scale


def test_launch_knobs_of_the_split_contractions_agree_with_each_other(tmp_path):
    """The A/B knobs of csrc/wn_gemm6.hip are read once per process (-> subprocesses).  WN_G6_NARROW=0 takes the all-layer skip
    gradient (M = 64 L > 256 rows, K = n_skipch) back from k_gemm6n's 128-row blocks to k_gemm6's 256-row blocks: the same bits (same
    k order, product order and tile signs).  WN_DW_SKIPRES=0 takes the skip_1x1 / res_1x1 weight gradients back from the fused launch
    to the two separate ones: the same values to fp32 summation order.  Six layers: M = 384 = three 128-row blocks, 1.5 256-row ones."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from tests.emu_util import emu_library\n"
        "from oracle import wavenet_oracle as O\n"
        "from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat, DEFAULT_FLAGS\n"
        "cfg_t = (32, 4, 64, 256, 3, 2, 2, 16)\n"
        "cfg = O.OracleConfig(*cfg_t)\n"
        "params = O.random_params(cfg, 7, scale=0.1); x, h, t = O.synthetic_batch(cfg, 1, 304, 8)\n"
        "eng = WaveNetEngine(*cfg_t, device='cpu', library=emu_library())\n"
        "load_state_into_flat(eng, params)\n"
        "eng.flags = DEFAULT_FLAGS\n"
        "loss, dl = eng.forward_loss(x, h, t)\n"
        "eng.backward(dl)\n"
        "torch.save(eng.grads().clone(), sys.argv[1])\n" % root)
    out = {}
    for name, env in (("default", {}), ("wide_rows", {"WN_G6_NARROW": "0"}), ("separate", {"WN_DW_SKIPRES": "0"})):
        path = str(tmp_path / (name + ".pt"))
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        out[name] = torch.load(path)
    assert bool(torch.isfinite(out["default"]).all()) and float(out["default"].abs().max()) > 0.0
    assert torch.equal(out["default"], out["wide_rows"])
    scale = float(out["separate"].abs().max())
    diff = float((out["default"] - out["separate"]).abs().max())
    assert 0.0 < diff <= 1e-6 * scale, (diff, scale)


def test_same_run_parity_helper_on_the_emulator():
    """oracle/same_run_parity.py (bench.py's `parity` block, tests/test_gpu_fullsize.py's benchmark-instance gate) on a tiny
    initialize()d model under the emulator: the reference module's own step (oracle/_ref; the restatement where the copy is
    absent) against the HIP step from the same state_dict on the same tensors, incl. the weights after one Adam step."""
    from oracle import same_run_parity as SRP
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd.engine import DEFAULT_FLAGS
    from pytorchwavenetvocoder_amd.nets import WaveNet, initialize
    from pytorchwavenetvocoder_amd.optim import FusedAdam
    cfg_t = (32, 6, 64, 32, 3, 1, 2, 16)
    torch.manual_seed(1)
    model = WaveNet(*cfg_t, _library=emu_library())
    model.apply(initialize)
    init_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x, h, t = O.synthetic_batch(O.OracleConfig(*cfg_t), 2, 64, 5)
    ref = SRP.reference_step(cfg_t, init_state, x, h, t, lr=1e-4)
    res = SRP.gpu_step_vs_reference(model, lambda m, lr: FusedAdam(m, lr=lr), ref, x, h, t, init_state, DEFAULT_FLAGS, lr=1e-4)
    G = SRP.GATES
    assert res["logits_maxabs"] <= G["logits_maxabs"] and res["loss_abs"] <= G["loss_abs"], res
    assert res["worst_grad_rel"] <= G["worst_grad_rel"] and res["kink_flip_max_distance"] <= SRP.KINK, res
    assert res["after_adam_elements"] == sum(v.numel() for v in init_state.values())
    # After Adam: this tiny instance has 128 loss positions, so its gradients are LARGE (tensor maxima ~4e-3) and an element
    # whose gradient is ~1e-9 -- below Adam's eps, where the update lr g / (|g| + eps) has sensitivity 1 / eps -- moves by per
    # cent of lr for a gradient difference at fp32 round-off of the tensor's maximum (1.3e-7 of it here; the reference's own
    # fp32 step is 0.8e-2 lr from its fp64 evaluation on this instance).  So: every element whose reference gradient is above
    # 10 eps meets the 1e-2 lr gate, and the few over it are of that sign-like kind.
    assert res["after_adam_elements_over_gate"] <= 8 and res["after_adam_over_gate_max_abs_reference_grad"] < 1e-7, res
    assert res["after_adam_maxabs_over_lr"] <= 0.1, res


def test_split_contractions_by_the_fp16_pair_split_and_their_redo():
    """WN_FLAG_MM_F16PAIR (csrc/wn_gemm6.hip k_gemm6<.., F16>): the weights x activations contractions on the split matrix-core
    kernel -- skip sum, post-net (incl. the cross-entropy epilogue), their data gradients, the all-layer skip gradient -- with two
    fp16 pieces per operand and three products, each launch followed by ONE conditional six-product launch.  Forward, loss and
    every gradient against the oracle at the gates of every mode; against the six-product mode within 2e-6 of the largest value;
    an activation beyond fp16's range raises the overflow word and the redo launches produce the six-product mode's bits."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.engine import WaveNetEngine, load_state_into_flat
    cfg_t = (256, 4, 64, 256, 3, 1, 2, 16)
    cfg = O.OracleConfig(*cfg_t)
    B, T = 2, 48   # (few loss positions: with 512 ReLU inputs per position a kink-free instance must still exist)
    params, x, h, t, margin, sd = PC.pick_instance(cfg, B, T, 81, 0.1)
    loss_ref, logits_ref, grads_ref = O.train_step(cfg, params, None, x, h, t)
    eng = WaveNetEngine(*cfg_t, device="cpu", library=emu_library())
    load_state_into_flat(eng, params)
    base = _lib.FLAG_AUX_FUSED | _lib.FLAG_DW_F16PAIR
    res, logs = {}, {}
    for name, flags in (("six", base), ("f16", base | _lib.FLAG_MM_F16PAIR)):
        eng.flags = flags
        logits = eng.forward(x, h)
        assert float((logits.transpose(1, 2) - logits_ref).abs().max()) <= PC.TOL_LOGITS
        box = {}

        def run():
            box["loss"], box["dl"] = eng.forward_loss(x, h, t)
            eng.backward(box["dl"])
        logs[name] = PC.launch_log(emu_library(), run)
        assert abs(float(box["loss"]) - float(loss_ref)) <= PC.TOL_LOSS
        res[name] = (logits.clone(), eng.grads().clone())
        grads = PC.flat_to_state(eng, res[name][1], O.param_shapes(cfg))
        for k, ref in grads_ref.items():
            if ref is not None:
                assert PC.rel_to_max(grads[k], ref) <= PC.TOL_GRAD, (name, k)
    # forward + backward of the training step: skip sum, post1, post2 + CE, post2_dx, post1_dx, dz_skip_all
    assert "mm_redo_if_overflow" not in logs["six"] and logs["f16"].get("mm_redo_if_overflow") == 6, logs
    for a, b, what in ((res["f16"][0], res["six"][0], "logits"), (res["f16"][1], res["six"][1], "gradients")):
        err = float((a - b).abs().max()) / float(b.abs().max())
        assert 0.0 < err <= 2e-6, (what, err)
    # overflow: conv_post_1's input relu(skip sum) beyond 65504 (a bias of 1e5 on the skip connections): the fp16 launch of
    # conv_post_1 yields non-finite accumulators, raises the word, and every redo launch of the pass does the work
    big = {k: v.clone() for k, v in params.items()}
    big["skip_1x1.0.bias"] = big["skip_1x1.0.bias"] + 1.0e5
    load_state_into_flat(eng, big)
    eng.flags = base
    want = eng.forward(x, h).clone()
    assert bool(torch.isfinite(want).all())
    eng.flags = base | _lib.FLAG_MM_F16PAIR
    got = eng.forward(x, h).clone()
    assert torch.equal(got, want)


def test_fused_forward_block_on_the_block_scaled_fp16_pair_split():
    """WN_FLAG_FUSED_F16PAIR (csrc/wn_fused.hip k_resblock_fwd_h): the fused 64-channel residual block with two fp16 pieces per
    operand and three products, every weight image and every 64 x 32 operand tile scaled by the power of two that puts its
    maximum at 2^12 / 2^14.  Golden cases (kernel_size 2 and 3) and the oracle at the gates of every mode; against the six-product
    forward within 2e-6 of the largest logit; and MAGNITUDES: the same model with its front convolution scaled so that the
    residual stream is ~1e6, and ~1e-6, still meets the gates (nothing leaves fp16's range, nothing is flushed)."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.engine import SIX_PRODUCT_FLAGS, WaveNetEngine, load_state_into_flat
    F = SIX_PRODUCT_FLAGS | _lib.FLAG_FUSED_F16PAIR
    for name in ("r64_k2_up", "r64_k3_up"):
        PC.check_golden_case(GoldenCase(name), emu_library(), "cpu", flags=F)
    PC.run_oracle_vs_engine((64, 6, 64, 32, 3, 1, 1, 16), 2, 48, 41, emu_library(), "cpu", flags=F, scale=0.2)   # K = 1
    PC.run_oracle_vs_engine((64, 6, 64, 32, 2, 2, 2, 16), 3, 80, 42, emu_library(), "cpu", flags=F, scale=0.2)   # T % 32 == 16
    g = GoldenCase("r64_k2_up")
    eng = WaveNetEngine(*g.cfg.as_tuple(), device="cpu", library=emu_library())
    load_state_into_flat(eng, g.params)
    eng.flags = SIX_PRODUCT_FLAGS
    six = eng.forward(g.x, g.h).clone()
    eng.flags = F
    f16 = eng.forward(g.x, g.h).clone()
    err = float((f16 - six).abs().max()) / float(six.abs().max())
    assert 0.0 < err <= 2e-6, err
    for mag in (1.0e6, 1.0e-6):
        big = g.clone_params()
        big["causal.conv.weight"] = big["causal.conv.weight"] * mag
        big["causal.conv.bias"] = big["causal.conv.bias"] * mag
        ref = O.forward(g.cfg, big, g.x, g.h)
        load_state_into_flat(eng, big)
        got = eng.forward(g.x, g.h).transpose(1, 2)
        assert bool(torch.isfinite(got).all())
        assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), mag


def test_backward_chain_on_the_block_scaled_fp16_pair_split():
    """WN_FLAG_CHAIN_F16PAIR (csrc/wn_fused.hip k_chain64s<.., H16>): the fused backward data chain with two fp16 pieces per
    operand and three products; weight images, the dP operand of every tile (by the per-tile maxima the producing launch
    recorded) and the dX operand (by its own maximum) scaled by powers of two.  Golden gradients (kernel_size 2 and 3, with and
    without the aux partial sums), the oracle on a half-full last tile and kernel_size 1, against six products within 2e-6 of the
    largest gradient, and MAGNITUDES: the same loss scaled by 2^-60 and by 2^+40 gives the same gradients times that factor
    (nothing leaves fp16's range, nothing is flushed)."""
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.engine import SIX_PRODUCT_FLAGS, WaveNetEngine, load_state_into_flat
    F = SIX_PRODUCT_FLAGS | _lib.FLAG_CHAIN_F16PAIR
    for name in ("r64_k2_up", "r64_k3_up"):
        PC.check_golden_case(GoldenCase(name), emu_library(), "cpu", flags=F)
    PC.check_golden_case(GoldenCase("r64_k2_up"), emu_library(), "cpu", flags=F & ~_lib.FLAG_AUX_FUSED)
    PC.check_golden_case(GoldenCase("r64_k2_up"), emu_library(), "cpu", flags=F | _lib.FLAG_FUSED_F16PAIR | _lib.FLAG_MM_F16PAIR | _lib.FLAG_DW_F16PAIR)
    PC.run_oracle_vs_engine((64, 6, 64, 32, 3, 1, 1, 16), 2, 48, 41, emu_library(), "cpu", flags=F, scale=0.2)   # K = 1
    PC.run_oracle_vs_engine((64, 6, 64, 32, 2, 2, 2, 16), 3, 80, 42, emu_library(), "cpu", flags=F, scale=0.2)   # T % 32 == 16
    g = GoldenCase("r64_k2_up")
    eng = WaveNetEngine(*g.cfg.as_tuple(), device="cpu", library=emu_library())
    load_state_into_flat(eng, g.params)
    res = {}
    for name, flags in (("six", SIX_PRODUCT_FLAGS), ("f16", F)):
        eng.flags = flags
        loss, dl = eng.forward_loss(g.x, g.h, g.t)
        res[name] = eng.backward(dl).clone()
    scale = float(res["six"].abs().max())
    err = float((res["f16"] - res["six"]).abs().max()) / scale
    assert 0.0 < err <= 2e-6, err
    for e in (-60, 40):
        loss, dl = eng.forward_loss(g.x, g.h, g.t, grad_scale=2.0 ** e)
        got = eng.backward(dl).clone() * 2.0 ** -e
        assert bool(torch.isfinite(got).all())
        assert float((got - res["six"]).abs().max()) / scale <= 2e-6, e
