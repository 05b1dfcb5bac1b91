#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Benchmark of the WaveNet-vocoder TRAINING STEP on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = forward + softmax-CE on [:, rf:] + backward + gradient all-reduce (N>1) + Adam, on
one synthetic minibatch already resident in HBM (reference train.py:527-540), through the module's training half-step
(`WaveNet.loss_and_backward` = wn_forward_loss + wn_backward_window: loss as the epilogue of conv_post_2, post-net / skip
backward over the loss window; DESIGN.md 3.3c -- same loss and gradients as the separate entry points).  Workload =
BASELINE.json configs[1]: 30-layer WaveNet, 64 residual / 256 skip channels, 80-dim mel aux,
mu-law softmax, batch 8 x batch_len 20000 per GPU (-> T = 23040 model inputs, 19970 loss
positions per sequence).  Weak scaling: the per-GPU batch is fixed.

The K timed steps are measured `--repeats` times (default 5; each region bracketed by barrier + synchronize, MAX over
ranks); `ms_per_step` / `value` are the MEDIAN region, min / max are reported beside it.

Prints ONE JSON line (rank 0).  Extra blocks:
  roofline      bound = hbm (north_star's denominator).  `achieved` / `frac` are the WHOLE STEP against SURVEY.md 8(d):
                timesteps/s x 78 356 algorithmic bytes per timestep vs 8 TB/s.  `traffic` = HBM bytes the step really
                moved (rocprofv3 PMC passes of this launch mode, profiles/), `traffic_ratio` = traffic / algorithmic.
                `dominant_kernel`: the kernel with the largest share of HIP-event time -- average launch duration,
                its 8(d) algorithmic bytes per launch and the fraction of the HBM roof they amount to (`frac_alg`),
                its compulsory bytes (every operand of the launch once) as bandwidth utilisation (`bw_util`), PMC
                bytes per launch, and the matrix-core fraction of its FLOPs.
  cpu_baseline  the reference's OWN module (oracle/_ref/wavenet.py, a build-time copy; kind "reference" -- without the copy
                the restatement oracle/wavenet_oracle.py, kind "port") under its training loop train.py:527-540, timed on this
                host's cores on a bounded sample: B=1 windows (thread count calibrated, core count stated) and the benchmark's
                OWN B=8 minibatch and initial weights.
  parity        north_star's gate in the same job: the first B=8 reference step above runs on the SAME x, h, t and the SAME
                initialize()d state_dict as the timed GPU model; the HIP step (wn_forward logits, wn_forward_loss, wn_backward_window,
                wn_adam_step) is compared with it -- logits_maxabs, loss_abs, worst_grad_rel (+ key), after_adam_maxabs_over_lr --
                in the default arithmetic and with six bf16 products for the weight gradients (oracle/same_run_parity.py).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG2 = dict(n_quantize=256, n_aux=80, n_resch=64, n_skipch=256, dilation_depth=10, dilation_repeat=3,
            kernel_size=2, upsampling_factor=80)
BATCH_PER_GPU = 8
BATCH_LENGTH = 20000
# algorithmic HBM bytes and FLOPs of one training timestep of this model
ALG_BYTES_PER_TIMESTEP = 78356.0
ALG_FLOP_PER_TIMESTEP = 9.27e6
HBM_PEAK = 8.0e12          # B/s  (MI355X_MICROARCH.md)
F32_MFMA_PEAK = 157.3e12   # FLOP/s dense f32 matrix = f32 vector peak
BF16_MFMA_PEAK = 2.5e15     # FLOP/s dense bf16 matrix (MI355X_MICROARCH.md); the 3-way split spends 6 bf16
SPLIT_PRODUCTS = 6.0        # MFMAs per fp32-equivalent product -> 417 TFLOP/s of fp32-equivalent work
DW_PRODUCTS = 3.0           # ... of the weight-gradient contractions (a third of the step's FLOPs) with WN_FLAG_DW_F16PAIR / _3PRODUCT
# share of the forward (and of the data-gradient) multiplies that run in the fused 64-channel kernels (taps + res 1x1: 30 x 20480 of
# SURVEY 8d's 1 236 992 MACs per timestep without the frame-rate aux); the rest -- skip 1x1, post-net -- runs on k_gemm6
FUSED_SHARE_R64 = 30.0 * (2 * 64 * 64 * 2 + 64 * 64) / (30.0 * (2 * 64 * 64 * 2 + 64 * 64 + 64 * 256) + 2 * 256 * 256)


def products_per_multiply(flags, fused_share):
    """Average matrix-core products per fp32-equivalent multiply of one training step under the engine flags: forward and
    data-gradient contractions (a third of the FLOPs each) take 6 (three bf16 pieces) or 3 (two fp16 pieces) -- the fused forward
    block by WN_FLAG_FUSED_F16PAIR, the k_gemm6 launches by WN_FLAG_MM_F16PAIR, the fused backward chain always 6; the weight
    gradients 6 or 3 (WN_FLAG_DW_F16PAIR / _3PRODUCT)."""
    from pytorchwavenetvocoder_amd import _lib
    mm = DW_PRODUCTS if (flags & _lib.FLAG_MM_F16PAIR) else SPLIT_PRODUCTS
    dw = DW_PRODUCTS if (flags & (_lib.FLAG_DW_F16PAIR | _lib.FLAG_DW_3PRODUCT)) else SPLIT_PRODUCTS
    ff = DW_PRODUCTS if (flags & _lib.FLAG_FUSED_F16PAIR) else SPLIT_PRODUCTS     # fused FORWARD block; the backward chain keeps six
    fwd = fused_share * ff + (1.0 - fused_share) * mm
    bwd = fused_share * (DW_PRODUCTS if (flags & _lib.FLAG_CHAIN_F16PAIR) else SPLIT_PRODUCTS) + (1.0 - fused_share) * mm
    return (fwd + bwd + dw) / 3.0
LAYERS_PER_BUCKET = 30      # gradient buckets = weight-gradient launch groups; N = 1 runs the SAME launch structure as N > 1.
# 30 (= all layers of this model, GradientReducer's default) = [post-net] [skip_1x1 + all residual layers] [front + upsampling]: measured 11.64 vs 11.80 ms/step for groups of 10
# layers on the same box (profiles/r02/ab_probe.txt); the post-net bucket is exchanged under the whole backward chain, the
# rest under the front-conv / upsampling gradients
# SURVEY.md 8(d) algorithmic bytes per timestep of ONE launch of the per-layer chain kernels (R = 64 words of 4 B):
#   forward block  read x_l, write x_{l+1}, save s, g                   4R
#   gate'          read s, g and dx_{l+1}                                3R   (dSkip is on chip in 8(d)'s accounting)
#   dX             write dx_l (its inputs dP never leave the chip)       1R
ALG_BYTES_PER_TIMESTEP_OF = {"fused_resblock_fwd": 4 * 64 * 4, "fused_bwd_gate": 3 * 64 * 4, "fused_bwd_dx": 64 * 4,
                             "fused_bwd_chain": 4 * 64 * 4}
PMC_FILES = ["profiles/r06/pmc_traffic.json", "profiles/r05/pmc_traffic.json", "profiles/r04/pmc_traffic.json", "profiles/r03/pmc_traffic.json", "profiles/r02/pmc_traffic.json"]


def geometry(rf, batch_length, U):
    """reference train.py:106-110,202-232"""
    bl = batch_length - (rf + batch_length) % U
    frames = (rf + bl) // U
    return bl, frames, frames * U


def synthetic_minibatch(B, T, frames, rank):
    """The benchmark's synthetic minibatch (CPU tensors): x, t = consecutive tokens of one random sequence per window
    (train.py:223-224), h ~ N(0, 1) (standardised features, train.py:464-470); one generator per rank."""
    gen = torch.Generator().manual_seed(1234 + rank)
    xx = torch.randint(0, CFG2["n_quantize"], (B, T + 1), generator=gen)
    x = xx[:, :-1].contiguous()
    t = xx[:, 1:].contiguous()
    h = torch.randn(B, CFG2["n_aux"], frames, generator=gen)
    return x, h, t


def cpu_baseline(seconds_budget=30.0, bench_instance=None):
    """The reference's own CPU path on this host's cores.

    ``kind: "reference"``: the reference's WaveNet module itself (oracle/_ref/wavenet.py, a build-time copy made by
    oracle/build_ref.py; git-ignored, travels to the GPU box) under the reference's training loop (train.py:527-540 restated in
    oracle/ref_step.py: nn.CrossEntropyLoss + torch.optim.Adam).  Without the copy: the restatement (oracle/wavenet_oracle.py),
    ``kind: "port"`` -- bit-identical results, same torch ops.

    Bounded sample (~30 s): the thread count is calibrated on B=1 windows of the same model (oneDNN convs of this size
    get SLOWER with hundreds of threads), then B=1 and the benchmark's own B=8 minibatch are timed at that count
    (SURVEY.md 8d asks for both).  `value` is the better of the two rates.

    ``bench_instance`` = (init_state, x, h, t) of the GPU benchmark (CPU tensors): the B=8 leg then runs on EXACTLY that instance
    -- its first step (the warm-up of the timing) is the reference side of the same-run parity gate.  Returns (block, ref) with
    ref = oracle.same_run_parity.reference_step's dict (None without ``bench_instance``)."""
    from oracle import ref_step as RS
    from oracle import wavenet_oracle as O
    try:
        navail = len(os.sched_getaffinity(0))
    except AttributeError:
        navail = os.cpu_count() or 1
    cfg_t = tuple(CFG2[k] for k in ("n_quantize", "n_aux", "n_resch", "n_skipch", "dilation_depth",
                                     "dilation_repeat", "kernel_size", "upsampling_factor"))
    cfg = O.OracleConfig(*cfg_t)
    bl, frames, T = geometry(cfg.receptive_field, BATCH_LENGTH, cfg.upsampling_factor)
    x, h, t = O.synthetic_batch(cfg, 1, T, 1)
    kind = "reference" if RS.available() else "port"
    if kind == "reference":
        trainer = RS.ReferenceTrainer(cfg_t, lr=1e-4, seed=1)     # the reference's module, initialised as train.py does
        step = trainer.step
    else:
        params = O.init_params(cfg, generator=torch.Generator().manual_seed(1))
        opt = O.OracleAdam(lr=1e-4)
        step = lambda xb, hb, tb: O.train_step(cfg, params, opt, xb, hb, tb)   # noqa: E731
    t_begin = time.time()

    def timed(xb, hb, tb):
        t0 = time.time()
        step(xb, hb, tb)
        return time.time() - t0

    results = {}
    cands = sorted(set(min(navail, c) for c in (8, 16, 32, 64)))
    for nthr in cands:
        torch.set_num_threads(nthr)
        warm = timed(x, h, t)                        # warm-up at this thread count
        if time.time() - t_begin > 0.3 * seconds_budget:
            results.setdefault(nthr, []).append(warm)
            break
        results.setdefault(nthr, []).append(timed(x, h, t))
        if time.time() - t_begin > 0.3 * seconds_budget:
            break
    best_thr = min(results, key=lambda k: min(results[k]))
    torch.set_num_threads(best_thr)
    t1 = list(results[best_thr])
    while time.time() - t_begin < 0.3 * seconds_budget and len(t1) < 3:
        t1.append(timed(x, h, t))
    b1 = {"B": 1, "steps": len(t1), "best_s": min(t1), "median_s": sorted(t1)[len(t1) // 2],
          "value": (T - cfg.receptive_field) / min(t1)}
    # the benchmark's own minibatch (B = 8): a step takes ~15 - 20 s on this kind of host (the reference's CPU path does not
    # scale with the batch: 60x the B = 1 step): ONE warm-up step (first-touch of the 8x larger buffers, oneDNN primitives
    # of these shapes), then one timed step -- both reported, `value` of the block from the timed one
    ref = None
    if bench_instance is not None:
        from oracle import same_run_parity as SRP
        init_state, x8, h8, t8 = bench_instance
        # The benchmark's own weights and minibatch, ONE training step from the initial state, twice: the warm-up evaluation with
        # another thread count (its distance from the timed one = the reference's own reproducibility: parity.reference_self_noise),
        # then the TIMED evaluation at the calibrated thread count -- the reference side of the parity gate.
        alt_thr = max(1, best_thr // 2) if best_thr > 1 else 2
        ref_alt = SRP.reference_step(cfg_t, init_state, x8, h8, t8, lr=1e-4, threads=alt_thr, keep_forward=False)
        warm8 = ref_alt["seconds"]
        ref_alt["trainer"] = None
        ref = SRP.reference_step(cfg_t, init_state, x8, h8, t8, lr=1e-4, threads=best_thr)
        ref["trainer"] = None
        ref["self_noise"] = SRP.reference_self_noise(ref_alt, ref, lr=1e-4)
        del ref_alt
        t8s = [ref["seconds"]]
    else:
        x8, h8, t8 = O.synthetic_batch(cfg, BATCH_PER_GPU, T, 2)
        warm8 = timed(x8, h8, t8)
        t8s = [timed(x8, h8, t8)]
    B8 = int(x8.size(0))
    b8 = {"B": B8, "steps": len(t8s), "warmup_s": warm8, "best_s": min(t8s), "median_s": sorted(t8s)[len(t8s) // 2],
          "value": B8 * (T - cfg.receptive_field) / min(t8s),
          "instance": "the GPU benchmark's own x, h, t and initialize()d weights" if ref is not None else "synthetic_batch seed 2"}
    best = max((b1, b8), key=lambda b: b["value"])
    what = ("the reference's own WaveNet module (wavenet_vocoder/nets/wavenet.py, build-time copy in oracle/_ref) under its "
            "training loop train.py:527-540" if kind == "reference" else
            "CPU oracle (restatement of the reference's torch-CPU ops: train.py:527-540 on wavenet.py:212-241)")
    return {"value": best["value"], "unit": "audio-samples/sec", "cores": best_thr, "kind": kind,
            "host_logical_cpus": navail, "b1": b1, "b8": b8,
            "value_b1": b1["value"], "value_headline_batch": b8["value"],   # flat: the B = 1 rate and the headline's own B = 8
            "sample": "%s, same 30-layer model, "
                      "windows of T=%d; threads calibrated over %s -> %d; B=1: %d steps, best %.3f s; B=8 (the headline's minibatch; one warm-up step, one timed): %s; "
                      "value = the better rate (B=%d); %.0f s of CPU work" % (
                          what, T, cands, best_thr, b1["steps"], b1["best_s"], "%.3f s" % b8["best_s"], best["B"],
                          time.time() - t_begin)}, ref


def decode_report(model, device, with_cpu):
    """BASELINE configs[4]: sample-by-sample generation on the same 30-layer model (argmax mode, seed
    token 128), the decode kernel of csrc/wn_decode.hip; beside it the oracle's queue algorithm
    (reference wavenet.py:309-395 restated) on the host cores for a few samples."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import decode_bench
    out = {"workload": "BASELINE configs[4]: fast_generate / batch_fast_generate on the configs[1] model, argmax, "
                       "context = receptive field (3070 positions, built by one forward of the residual stack; context_s reported separately)"}
    for B, n in ((1, 2000), (256, 2000)):
        m = decode_bench.measure(model, B, n, device)
        out["batch%d" % B] = {k: m[k] for k in ("us_per_step", "samples_per_sec_per_utt", "samples_per_sec", "context_s")}
    # per step one workgroup streams the packed weights once: compulsory bytes = the stream
    eng = model.engine
    stream_bytes = float(eng.lib.wn_decode_stream_bytes(ctypes.byref(eng.cfg)))
    out["stream_bytes_per_step"] = stream_bytes
    out["stream_GBps_per_workgroup"] = stream_bytes / (out["batch1"]["us_per_step"] * 1e-6) / 1e9
    out["stream_note"] = "one CU sustains ~112 GB/s on a 5 MB cyclic read (tools/stream_probe.hip, profiles/r01/stream_probe.txt)"
    # the recipes' own model size (n_resch 512 / n_skipch 256, egs/arctic/sd/run.sh:46-52; decode.py:274-327): the any-size path --
    # ONE persistent launch per chunk of steps: 128 workgroups with fp32 VALU dot products for one utterance (csrc/wn_dlp.hip), 64
    # workgroups per block of 16 utterances with 16x16x4 matrix-core tiles from 2 to 64 (csrc/wn_dlpf.hip: plain vectors + one
    # flag per workgroup and stage, inputs by global -> LDS transfers; 64 = 256 workgroups, resident all at once: asked of the
    # device first); two groups up to 128, layer-wise launches above
    try:
        from pytorchwavenetvocoder_amd.nets import WaveNet, initialize
        torch.manual_seed(1)
        big = WaveNet(256, 80, 512, 256, 10, 3, 2, 80)
        big.apply(initialize)
        big.to(device)
        rs = {"model": "512/256, A=80, K=2, U=80, 30 layers (the recipes' default size)"}
        for B, n, lay in ((1, 400, True), (4, 400, True), (16, 300, True), (32, 300, True), (48, 300, True), (64, 200, True),
                          (64, 200, "launches")):
            m = decode_bench.measure(big, B, n, device, layered=lay)
            key = "batch%d" % B if lay is True else "batch%d_by_launches" % B
            rs[key] = {k: m[k] for k in ("us_per_step", "samples_per_sec_per_utt", "samples_per_sec", "context_s")}
            rs[key]["path"] = (("one persistent launch (wn_dlp)" if B <= 1 else
                                "one persistent launch (wn_dlpf: %d column blocks)" % ((B + 15) // 16) if B <= 64 else
                                "persistent launches (wn_dlpf), groups of 64 utterances")
                               if lay is True else "layer-wise launches")
        out["recipe_size"] = rs
        del big
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001 -- never takes the headline line down
        out["recipe_size"] = {"error": repr(e)}
    if with_cpu:
        # CPU leg: the reference's OWN fast_generate (wavenet.py:309-395, build-time copy in oracle/_ref -> kind "reference"),
        # else the restatement (kind "port"); a few samples after the context, 8 threads (the per-sample work is ~500 tiny ops)
        from oracle import ref_step as RS
        from oracle import wavenet_oracle as O
        cfg_t = tuple(CFG2[k] for k in ("n_quantize", "n_aux", "n_resch", "n_skipch", "dilation_depth",
                                         "dilation_repeat", "kernel_size", "upsampling_factor"))
        cfg = O.OracleConfig(*cfg_t)
        x = torch.full((1, 1), 128, dtype=torch.int64)
        n = 40
        h = torch.randn(1, cfg.n_aux, (n + 6 + 79) // 80)
        torch.set_num_threads(8)
        if RS.available():
            ref = RS.load_reference()
            torch.manual_seed(1)
            rmodel = ref.WaveNet(*cfg_t)
            rmodel.apply(ref.initialize)
            rmodel.eval()
            kind, what = "reference", "the reference's own WaveNet.fast_generate (wavenet.py:309-395, copy in oracle/_ref)"

            def gen(k):
                with torch.no_grad():
                    return rmodel.fast_generate(x, h, k, mode="argmax")
        else:
            params = O.init_params(cfg, generator=torch.Generator().manual_seed(1))
            kind, what = "port", "oracle fast_generate (restatement of wavenet.py:309-395)"

            def gen(k):
                return O.fast_generate(cfg, params, x, h, k)
        gen(1)                       # warm-up (first-touch allocations, oneDNN primitives)
        t0 = time.time()
        gen(5)
        t_a = time.time() - t0       # context pass + 5 samples
        t0 = time.time()
        gen(5 + n)
        t_b = time.time() - t0       # context pass + 5 + n samples
        out["cpu_baseline"] = {"value": n / max(t_b - t_a, 1e-9), "unit": "audio-samples/sec", "cores": 8,
                               "kind": kind, "context_s": t_a,
                               "sample": "%s, B=1, argmax: the time of %d further samples (two runs of 5 and %d samples after the "
                                         "same 3070-position context)" % (what, n, 5 + n)}
    return out


def extra_workloads(*release):
    """Two further training workloads of the same code, timed after the headline (N = 1 only; NOT the metric):
      * BASELINE configs[3] geometry -- LJSpeech recipe shape: kernel_size 3, upsampling 256, receptive field 6139, batch 8 x
        batch_len 20000 (T = 26112) -- with the softmax head the reference has (egs/ljspeech/sd-melspc/run.sh:29);
      * the recipe-size model (n_resch 512 / n_skipch 256, egs/arctic/sd/run.sh:46-57) on 4 windows of batch_len 20000:
        matrix-bound, so reported against the split-arithmetic matrix peak (2.5 PFLOP/s bf16 / 6 products = 417 TFLOP/s)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import recipe_bench
    out = {}
    try:
        c3 = recipe_bench.measure(resch=64, kernel_size=3, upsampling=256, T=26112, batch=8, steps=10, with_kernels=False)
        out["configs3_geometry"] = {k: c3[k] for k in ("model", "B", "T", "rf", "ms_per_step", "samples_per_sec")}
        # SURVEY 8(d)'s algorithmic bytes per timestep for this model: (9 R L + 7 S + 2 Q) * 4 + 16 + 4 A / U
        alg3 = (9 * 64 * 30 + 7 * 256 + 2 * 256) * 4 + 16 + 4.0 * 80 / 256
        bps3 = c3["B"] * c3["T"] * alg3 / (c3["ms_per_step"] * 1e-3)
        out["configs3_geometry"]["roofline"] = {"bound": "hbm", "achieved": bps3 / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                                "frac": bps3 / HBM_PEAK, "algorithmic_bytes_per_timestep": alg3}
        # the same geometry with the 10-component mixture-of-logistics head (BASELINE configs[3] as stated; the head is NOT in
        # the reference: parity unpinned, tests/test_gpu_fullsize.py checks it against this repo's own restatement)
        m3 = recipe_bench.measure(resch=64, kernel_size=3, upsampling=256, T=26112, batch=8, steps=10, with_kernels=False,
                                  n_mixture=10)
        out["configs3_mol"] = {k: m3[k] for k in ("model", "B", "T", "rf", "ms_per_step", "samples_per_sec")}
        out["configs3_mol"]["note"] = "mixture-of-logistics head: not in the reference (parity unpinned by it)"
        rs = recipe_bench.measure(resch=512, kernel_size=2, upsampling=80, T=23040, batch=4, steps=3, with_kernels=True)
        out["recipe_size"] = {k: rs[k] for k in ("model", "B", "T", "rf", "ms_per_step", "samples_per_sec", "approx_train_tflops")}
        # per-launch-tag HIP-event times of ONE extra step (serial, like `kernels` of the headline); the counters of this workload
        # (traffic, MFMA utilisation per kernel) are committed as profiles/r05/pmc_traffic_recipe.json / pmc_mfma_recipe.json
        out["recipe_size"]["kernels"] = {k: {"launches": v["launches"], "ms": v["ms"], "tflops": v["tflops"]}
                                         for k, v in list(rs.get("kernels", {}).items())[:10]}
        # products per multiply of this run's arithmetic: 6 (three bf16 pieces) for the forward / data-gradient contractions, 3 for the
        # weight gradients (a third of the FLOPs) in the engine's default mode (two fp16 pieces, WN_FLAG_DW_F16PAIR)
        from pytorchwavenetvocoder_amd import _lib
        from pytorchwavenetvocoder_amd.engine import DEFAULT_FLAGS
        eflags = int(os.environ.get("WN_ENGINE_FLAGS", str(DEFAULT_FLAGS)), 0)
        products = products_per_multiply(eflags, 0.0)   # the wide model has no fused kernels: every contraction is a k_gemm6 / k_gemm6_dw launch
        peak = BF16_MFMA_PEAK / products
        out["recipe_size"]["frac_of_split_matrix_peak"] = rs["approx_train_tflops"] * 1e12 / peak
        out["recipe_size"]["roofline"] = {"bound": "mfma", "achieved": rs["approx_train_tflops"], "peak": peak / 1e12,
                                          "unit": "TFLOP/s", "frac": rs["approx_train_tflops"] * 1e12 / peak,
                                          "products_per_multiply": products,
                                          "note": "fp32-equivalent work against the dense 16-bit MFMA peak / the products per multiply of "
                                                  "the operand split (6 for forward and data gradients, 3 for the weight gradients "
                                                  "in the default mode: %.1f on average)" % products}
    except Exception as e:  # noqa: BLE001 -- the extras never take the headline line down
        out["error"] = repr(e)
    return out


def stream_mode(flags):
    """Launch mode of the timed steps (include/wavenet_hip.h WN_FLAG_*).  The per-launch HIP-event table of the
    `kernels` / `roofline` blocks is always taken serially (wn_prof_enable keeps everything on one stream)."""
    from pytorchwavenetvocoder_amd import _lib
    n = (flags >> 8) & 0xff
    parts = []
    if flags & _lib.FLAG_BWD_OVERLAP:
        parts.append("weight gradients of every %d walked layers on the library's side stream beside the backward chain" % (n or 5))
    elif n:
        parts.append("weight gradients in groups of %d layers" % n)
    if flags & _lib.FLAG_FWD_OVERLAP:
        parts.append("skip-sum in 3 chunks on the side stream beside the residual stack")
    return "; ".join(parts) if parts else "serial (one stream)"


def check_aux_fused(model, x, h, t, tol=1e-5):
    """One backward with the separate wn_aux_bwd launch and one with WN_FLAG_AUX_FUSED (the engine's default) on the
    benchmark's own batch; the two gradient buffers must agree per parameter tensor to `tol` of the tensor's max (the
    modes only re-associate sums)."""
    from pytorchwavenetvocoder_amd import _lib
    eng = model.engine
    base = eng.flags
    eng.flags = base & ~_lib.FLAG_AUX_FUSED
    model.loss_and_backward(x, h, t)
    g0 = eng.grads().clone()
    eng.flags = base | _lib.FLAG_AUX_FUSED
    model.loss_and_backward(x, h, t)
    g1 = eng.grads().clone()
    eng.flags = base
    worst = 0.0
    for off, n, shape, dead in model._param_slices:
        if dead or n == 0:
            continue
        a, b = g0[off:off + n], g1[off:off + n]
        if not bool(torch.isfinite(b).all()):
            return False, float("inf")
        m = float(a.abs().max())
        worst = max(worst, float((a - b).abs().max()) / max(m, 1e-30))
    return worst <= tol, worst


def same_run_parity(model, ref, inst, layers_per_bucket):
    """north_star: "outputs match the reference PyTorch CPU forward on identical inputs within 1e-4 fp32, with the reference
    CPU path timed in the same run".  ``ref`` = the reference module's ONE training step (train.py:527-540) on the benchmark's
    own x, h, t from the benchmark's own initialize()d state_dict (the warm-up step of cpu_baseline's B=8 timing); the HIP
    step is repeated from the same state in the default arithmetic and with six bf16 products for the weight gradients
    (oracle/same_run_parity.py; gates: logits 1e-4, loss 1e-5, gradients 1e-4 of a tensor's maximum, weights after Adam 1e-2 lr)."""
    from oracle import same_run_parity as SRP
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.optim import FusedAdam
    init_state, x, h, t = inst
    base = int(model.engine.flags)
    six = base & ~_lib.NARROW_FLAGS
    out = {"reference": ref["kind"], "reference_step_s": ref["seconds"], "reference_threads": ref.get("threads"),
           "reference_self_noise": ref.get("self_noise"),
           "instance": "the timed model's initialize()d state_dict (seed 1) and the timed x, h, t (B=%d, T=%d), lr 1e-4, one Adam step"
                       % (x.size(0), x.size(1)),
           "method": "oracle/same_run_parity.py: HIP logits / loss / every gradient tensor / weights after Adam against the reference "
                     "module's own step; at ReLU ties (both evaluations within 1e-5 of the kink) the HIP backward takes the "
                     "reference's sub-gradient choice (kink_flips elements).  `pass` = all four gates as stated; "
                     "`reference_self_noise` = the same reference step at two thread counts (its own reproducibility in the gates' "
                     "units; host- and thread-count dependent: 0.004 - 0.04 lr after Adam).  The after-Adam gate of 1e-2 lr is below what "
                     "fp32 resolves at this size: against the fp64 evaluation of the same step the reference's own fp32 step is 0.016 - "
                     "0.024 lr (7 - 8 elements over the gate) and the HIP step 0.035 - 0.042 lr (11 - 17 elements), while the HIP step's "
                     "worst gradient tensor is CLOSER to fp64 than the reference's (6.2e-6 vs 8.3e-6; profiles/r06/adam_gate_study.txt) "
                     "-- the elements concerned have |gradient| < 1e-7, where Adam's first update lr g / (|g| + eps) is sign-like; "
                     "`after_adam_well_conditioned_pass` = every element over the gate is of that kind"}
    try:
        mk = lambda m, lr: FusedAdam(m, lr=lr)   # noqa: E731
        d = SRP.gpu_step_vs_reference(model, mk, ref, x, h, t, init_state, base, lr=1e-4, layers_per_bucket=layers_per_bucket)
        out.update(d)                              # flat: the default arithmetic (the metric's)
        if six != base:
            out["dw_six_products"] = SRP.gpu_step_vs_reference(model, mk, ref, x, h, t, init_state, six, lr=1e-4,
                                                               layers_per_bucket=layers_per_bucket)
    except Exception as e:  # noqa: BLE001 -- reported, never takes the headline line down
        out["error"] = repr(e)
        out["pass"] = False
    return out


def load_pmc_traffic(flags):
    """HBM traffic of a step / per kernel launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh), used only
    when they were taken in the launch mode of this run (`_engine_flags`)."""
    for rel in PMC_FILES:
        try:
            with open(os.path.join(ROOT, rel)) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        if int(d.get("_engine_flags", -1)) == int(flags):
            d["_file"] = rel
            try:   # was the counter pass taken on exactly the kernel sources this run's library was built from?
                from pytorchwavenetvocoder_amd.csrc import build as _b
                d["_same_build"] = d.get("_source_digest") == _b._digest()
            except Exception:  # noqa: BLE001
                d["_same_build"] = None
            return d
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="how many times the K timed steps are measured (median reported)")
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="sequences per GPU")
    ap.add_argument("--layers-per-bucket", type=int, default=LAYERS_PER_BUCKET,
                    help="gradient buckets = weight-gradient launch groups of this many layers (same for N = 1 and N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the configs[4] generation measurement")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the same-run parity gate (the first B=8 reference step on the benchmark's own instance is then not compared)")
    ap.add_argument("--no-fused", action="store_true", help="force the layered (any-size) kernels")
    ap.add_argument("--exact-mfma", action="store_true", help="every contraction on the exact f32-input MFMA")
    ap.add_argument("--no-aux-fused", action="store_true",
                    help="aux-path gradients by the separate wn_aux_bwd launch instead of the partial sums inside the gate "
                         "kernel (WN_FLAG_AUX_FUSED, the engine's default)")
    ap.add_argument("--aux-fused", action="store_true", help=argparse.SUPPRESS)   # round-1 spelling: now the default
    ap.add_argument("--profile-steps", type=int, default=2, help="extra untimed steps with per-launch HIP events")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the two extra (non-headline) workloads: BASELINE configs[3] geometry and the recipe-size model")
    args = ap.parse_args()

    import torch.distributed as dist
    from pytorchwavenetvocoder_amd import _lib
    from pytorchwavenetvocoder_amd.distributed import GradientReducer, rccl_footprint_defaults
    from pytorchwavenetvocoder_amd.nets import WaveNet, initialize
    from pytorchwavenetvocoder_amd.optim import FusedAdam

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # WN_BENCH_BACKEND=gloo lets the multi-rank control flow be exercised on a box with fewer GPUs than ranks
    # (ranks then share devices); the driver's runs use nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("WN_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            rccl_footprint_defaults()
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    torch.manual_seed(1)  # reference default seed (train.py:386)
    model = WaveNet(**CFG2)
    model.apply(initialize)
    model.to(device)
    # the weights the timed steps start from: the same-run parity gate hands exactly these to the reference's module
    want_parity = (world == 1 and not args.no_cpu_baseline and not args.no_parity)
    init_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()} if want_parity else None
    if args.no_fused:
        model.engine.flags |= _lib.FLAG_NO_FUSED
    if args.exact_mfma:
        model.engine.flags |= _lib.FLAG_EXACT_MFMA
    if args.no_aux_fused:
        model.engine.flags &= ~_lib.FLAG_AUX_FUSED
    rf = model.receptive_field
    bl, frames, T = geometry(rf, BATCH_LENGTH, CFG2["upsampling_factor"])
    B = args.batch
    x, h, t = (v.to(device) for v in synthetic_minibatch(B, T, frames, rank))
    if world > 1:  # identical initial weights on every rank (no per-step broadcast afterwards)
        dist.broadcast(model.engine.flat_params, src=0)

    aux_on = bool(model.engine.flags & _lib.FLAG_AUX_FUSED) and not (args.no_fused or args.exact_mfma)
    aux_mode = "partial sums inside the gate kernel (WN_FLAG_AUX_FUSED)" if aux_on else "separate wn_aux_bwd launch"
    opt = FusedAdam(model, lr=1e-4)
    red = GradientReducer(model, layers_per_bucket=args.layers_per_bucket)

    def step():
        loss = red.loss_and_backward(x, h, t)
        opt.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        loss = step()
    regions = []
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            el = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            elapsed = float(el.item())
        regions.append(elapsed)
    final_loss = float(loss.item())
    elapsed = sorted(regions)[len(regions) // 2]      # median region of K steps

    ms_per_step = elapsed / args.steps * 1e3
    samples_per_s = world * B * (T - rf) * args.steps / elapsed
    timesteps_per_s_gpu = B * T * args.steps / elapsed

    # ---- per-kernel timing with HIP events (untimed extra steps, rank 0) ----
    roofline = None
    kernels = None
    if args.profile_steps > 0:
        # every rank runs the extra steps (they contain the gradient all-reduce); only rank 0 records events
        lib = model.engine.lib
        torch.cuda.synchronize(device)
        if rank == 0:
            lib.wn_prof_enable(1)
        for _ in range(args.profile_steps):
            step()
        torch.cuda.synchronize(device)
        if rank == 0:
            lib.wn_prof_enable(0)
    if rank == 0:
        pmc = load_pmc_traffic(model.engine.flags)
        alg_step = B * T * ALG_BYTES_PER_TIMESTEP
        step_products = products_per_multiply(model.engine.flags, FUSED_SHARE_R64)
        step_traffic = pmc.get("_step_total_bytes") if pmc else None
        roofline = {
            "bound": "hbm", "achieved": timesteps_per_s_gpu * ALG_BYTES_PER_TIMESTEP / 1e9, "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": timesteps_per_s_gpu * ALG_BYTES_PER_TIMESTEP / HBM_PEAK,
            "scope": "whole training step per GPU: SURVEY.md 8(d) algorithmic bytes (78 356 B per timestep x %d timesteps "
                     "= %.2f GB per step) / median step time, vs the 8 TB/s HBM3E peak" % (B * T, alg_step / 1e9),
            "algorithmic_bytes_per_step": alg_step,
            "traffic": step_traffic,
            "traffic_ratio": (step_traffic / alg_step) if step_traffic else None,
            "traffic_note": ("HBM bytes of one step, (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over all launches, separate "
                             "rocprofv3 --pmc passes of this launch mode, NOT taken in this run: %s (commit %s; kernel sources "
                             "%s those of this run's library)" % (
                                 pmc["_file"], pmc.get("_commit"),
                                 "identical to" if pmc.get("_same_build") else "DIFFERENT from (or unknown vs)")) if pmc else
                            "no PMC passes committed for this launch mode",
            "traffic_same_build": pmc.get("_same_build") if pmc else None,
            "step_f32_flop_frac": timesteps_per_s_gpu * ALG_FLOP_PER_TIMESTEP / F32_MFMA_PEAK,
            # the OTHER roof of this arithmetic: 9.27 MFLOP per timestep of fp32-equivalent work on the bf16 matrix cores with six
            # products per multiply (2.5 PFLOP/s / 6 = 417 TFLOP/s) -- with the split in place the step cannot beat matrix_roof_ms,
            # i.e. `frac` cannot exceed hbm-roof time / matrix_roof_ms whatever the kernels do
            "matrix_roof_ms": B * T * ALG_FLOP_PER_TIMESTEP / (BF16_MFMA_PEAK / step_products) * 1e3,
            "matrix_roof_frac": (B * T * ALG_FLOP_PER_TIMESTEP / (BF16_MFMA_PEAK / step_products) * 1e3) / ms_per_step,
            "frac_ceiling_under_split": (alg_step / HBM_PEAK) / (B * T * ALG_FLOP_PER_TIMESTEP / (BF16_MFMA_PEAK / step_products)),
            "matrix_products_per_multiply": step_products,
            "matrix_roof_note": "fp32-equivalent FLOPs of the step (SURVEY 8d: 9.27 MFLOP per timestep) at 2.5 PFLOP/s dense 16-bit "
                                "MFMA / the products per multiply of the operand split: 3 (two fp16 pieces) for the fused forward block, "
                                "the k_gemm6 launches and the weight gradients (WN_FLAG_FUSED_F16PAIR / _MM_F16PAIR / _DW_F16PAIR), 6 "
                                "(three bf16 pieces) for the fused backward chain (%.0f %% of the data-gradient multiplies) "
                                "-> %.2f on average: the binding roof of this arithmetic is the matrix "
                                "pipe, not HBM; matrix_roof_frac = that time / the measured step" % (100.0 * FUSED_SHARE_R64, step_products),
            # the chip's streaming plateau, measured with 256 MiB tensors (beyond the 256 MiB Infinity Cache), float4 per lane,
            # 256 workgroups (tools/microbench/stream_big.hip, profiles/r06/stream_big.txt): numbers, not notes
            "stream_plateau_TBps": {"read_only": 6.3, "copy_1r_1w": 5.4, "fwd_mix_1r_3w": 5.9, "chain_mix_5r_2w": 5.5,
                                    "fwd_mix_tile_pattern_240wg": 4.8},
            "stream_plateau_note": "MI355X_MICROARCH.md's 6.29 TB/s is the READ rate (measured here: 5.9 - 6.9 TB/s read-only, 5.2 - 5.6 "
                                   "copy, 5.2 - 5.9 for the forward block's 1 read : 3 writes, 5.1 - 5.8 for the chain's 5 : 2; the fused "
                                   "kernels' tile pattern at 240 workgroups 4.6 - 5.1); `frac` stays priced against the 8 TB/s peak",
        }
    if rank == 0 and args.profile_steps > 0:
        need = lib.wn_prof_report(None, 0)
        buf = ctypes.create_string_buffer(max(need, 16))
        lib.wn_prof_report(buf, len(buf))
        prof = json.loads(buf.value.decode() or "{}")
        tot_ms = sum(v["ms"] for v in prof.values()) or 1.0
        kernels = {k: {"launches_per_step": v["count"] / args.profile_steps,
                       "ms_per_step": v["ms"] / args.profile_steps,
                       "share": v["ms"] / tot_ms,
                       "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["flops"] > 0 and v["ms"] > 0 else None}
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        for k, v in prof.items():
            kernels[k]["GBps_compulsory"] = (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["bytes"] > 0 and v["ms"] > 0 else None
        # Dominant kernel = largest share of the step's HIP-event time.
        dom = next((k for k, v in kernels.items() if v["tflops"] is not None), None)
        if dom is not None:
            v = prof[dom]
            sec_launch = v["ms"] * 1e-3 / v["count"]
            mfma_peak = F32_MFMA_PEAK if args.exact_mfma else BF16_MFMA_PEAK / SPLIT_PRODUCTS
            alg_launch = ALG_BYTES_PER_TIMESTEP_OF.get(dom)
            alg_launch = alg_launch * B * T if alg_launch else None
            roofline["dominant_kernel"] = {
                "kernel": dom, "launches_per_step": v["count"] / args.profile_steps, "avg_launch_ms": sec_launch * 1e3,
                "share_of_step": v["ms"] / tot_ms,
                "algorithmic_bytes_per_launch": alg_launch,
                "achieved_alg_GBps": (alg_launch / sec_launch / 1e9) if alg_launch else None,
                "frac_alg": (alg_launch / sec_launch / HBM_PEAK) if alg_launch else None,
                "compulsory_bytes_per_launch": v["bytes"] / v["count"],
                "bw_util": v["bytes"] / v["count"] / sec_launch / HBM_PEAK,
                "traffic": (pmc.get(dom) or {}).get("hbm_bytes_per_launch") if pmc else None,
                "flop_per_launch": v["flops"] / v["count"],
                "mfma_frac": v["flops"] / v["count"] / sec_launch / mfma_peak,
                "traffic_ratio": (((pmc.get(dom) or {}).get("hbm_bytes_per_launch") or 0) / alg_launch) if (pmc and alg_launch) else None,
                "notes": "algorithmic = SURVEY 8(d) bytes of this launch (skip-sum gradient and weight-gradient operands "
                         "on chip); compulsory = every operand tensor of the launch as built, once; traffic = PMC; "
                         "matrix peak = " + ("f32-input MFMA 157.3 TFLOP/s" if args.exact_mfma else
                                             "2.5 PFLOP/s dense bf16 / 6 products of the 3-way split = 417 TFLOP/s fp32-equivalent")}

    if rank == 0 and roofline is not None and roofline.get("dominant_kernel"):
        dk = roofline["dominant_kernel"]   # flat copies (a nested object does not survive every consumer of the line)
        roofline["dominant_kernel_name"] = dk["kernel"]
        roofline["dominant_kernel_us"] = dk["avg_launch_ms"] * 1e3
        roofline["dominant_kernel_frac_alg"] = dk["frac_alg"]
        roofline["dominant_kernel_traffic_ratio"] = dk.get("traffic_ratio") or None

    # ---- the exchange, as the ranks saw it (N > 1: a few extra untimed steps with the exposed part measured) ----
    comm = None
    if world > 1 and device.type == "cuda":
        red.measure_exposed = True
        for _ in range(5):
            step()
        barrier()
        red.measure_exposed = False
    if rank == 0:
        comm = red.comm_report()
        comm["ranks_reported_by_backend"] = dist.get_world_size() if world > 1 else 1
        comm["note"] = ("exposed_ms_per_step = event behind the last backward launch -> event after the caller's stream joined the "
                        "side stream (what the bucketed all-reduces did not hide under the backward kernels), rank 0, 5 untimed steps")
    if rank == 0:
        out = {
            "metric": "train audio-samples/sec, 30-layer WaveNet batch_len=20000",
            "value": samples_per_s, "unit": "audio-samples/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": len(regions), "ms_per_step_min": min(regions) / args.steps * 1e3,
            "ms_per_step_max": max(regions) / args.steps * 1e3,
            "config": {"workload": "BASELINE configs[1]: 30-layer WaveNet, 64 residual / 256 skip ch, 80-dim mel aux, "
                                   "mu-law softmax, K=2, U=80; batch %d x batch_len 20000 per GPU (T=%d inputs, %d loss "
                                   "positions per sequence), random-init weights, fwd+CE+bwd+allreduce+Adam" % (
                                       B, T, T - rf),
                       "global_batch": world * B, "parallelism": "dp%d" % world,
                       "kernels": "layered" if args.no_fused else "fused+gemm",
                       "streams": stream_mode(model.engine.flags), "aux_gradient": aux_mode,
                       "engine_flags": int(model.engine.flags),
                       "gradient_buckets": "post-net+skip | groups of %d layers | front+upsampling (weight gradients are "
                                           "launched per bucket; identical structure for N = 1 and N > 1)" % args.layers_per_bucket,
                       "loss_path": "wn_forward_loss + wn_backward_window: the residual stack computes every position in both "
                                    "directions; the loss of train.py:534-536 covers [:, rf:], so the skip-sum / post-net "
                                    "contractions of the forward pass and the post-net / skip part of the backward pass run over "
                                    "that window only (everything between skip sum and loss is pointwise in time: same loss, same "
                                    "gradients), and the cross-entropy is the epilogue of the conv_post_2 contraction (logits not "
                                    "materialised: same loss / dlogits as the separate kernel to 1e-7)",
                       "timing": "median of %d regions of %d steps" % (len(regions), args.steps),
                       "arithmetic": "fp32 storage and accumulation; contractions on the bf16 matrix cores with a 3-way "
                                     "operand split (6 products, fp32-equivalent to round-off); " +
                                     ("the weight-gradient contractions (leaf results: sums over every position of the minibatch) "
                                      "split their operands into two fp16 pieces and take 3 products on the fp16 matrix cores "
                                      "(WN_FLAG_DW_F16PAIR: 2^-22 per product; gradient operand scaled by a power of two from the "
                                      "MEASURED max |dlogits|, out-of-range gradients detected and redone with six products); "
                                      if (model.engine.flags & _lib.FLAG_DW_F16PAIR) else
                                      "the weight-gradient contractions (leaf results) take 3 of the 6 products (WN_FLAG_DW_3PRODUCT, "
                                      "opt-in: it misses the golden after-Adam gate); "
                                      if (model.engine.flags & _lib.FLAG_DW_3PRODUCT) else "") +
                                     ("the k_gemm6 contractions (skip sum, post-net + loss, their data gradients) take the same fp16 "
                                      "pair split (WN_FLAG_MM_F16PAIR), each launch with its conditional six-product redo; "
                                      if (model.engine.flags & _lib.FLAG_MM_F16PAIR) else "") +
                                     ("the fused 64-channel forward block takes it block-scaled (WN_FLAG_FUSED_F16PAIR: every weight image "
                                      "and operand tile by the power of two of its own maximum); the fused backward chain keeps six bf16 "
                                      "products; " if (model.engine.flags & _lib.FLAG_FUSED_F16PAIR) else "") +
                                     "WN_FLAG_EXACT_MFMA selects the f32 MFMA everywhere"},
            "timesteps_per_sec": world * timesteps_per_s_gpu, "final_loss": final_loss,
            "roofline": roofline, "comm": comm, "kernels": kernels,
        }
        if not args.no_cpu_baseline and world == 1:
            inst = (init_state, x.cpu(), h.cpu(), t.cpu()) if want_parity else None
            out["cpu_baseline"], ref = cpu_baseline(bench_instance=inst)
            if ref is not None:
                out["parity"] = same_run_parity(model, ref, inst, args.layers_per_bucket)
                del ref
        else:
            out["cpu_baseline"] = None
        if not args.no_decode and world == 1:
            out["decode"] = decode_report(model, device, not args.no_cpu_baseline)
        if not args.no_extras and world == 1:
            # the other arithmetic modes on the SAME step, beside the metric (every narrowed strand keeps its six-product
            # figure on the line): six bf16 products EVERYWHERE (fp32-equivalent to round-off: what every fp16 launch falls back to when
            # an operand leaves fp16's range), six products for the k_gemm6 contractions only (weight gradients stay fp16 pairs: round
            # 5's default), and the opt-in three bf16 products for the weight gradients (misses the golden after-Adam gate)
            base_flags = model.engine.flags
            six_flags = base_flags & ~_lib.NARROW_FLAGS

            def alt_mode(flags, note):
                try:
                    model.engine.flags = flags
                    for _ in range(3):
                        step()
                    reg = []
                    for _ in range(3):
                        barrier()
                        t0 = time.perf_counter()
                        for _ in range(args.steps):
                            step()
                        barrier()
                        reg.append((time.perf_counter() - t0) / args.steps * 1e3)
                    return {"ms_per_step": sorted(reg)[1], "samples_per_sec": B * (T - rf) / (sorted(reg)[1] * 1e-3),
                            "engine_flags": int(flags), "note": note}
                except Exception as e:  # noqa: BLE001
                    return {"error": repr(e)}
                finally:
                    model.engine.flags = base_flags
            dw6 = alt_mode(six_flags, "EVERY contraction with six bf16 products (engine.SIX_PRODUCT_FLAGS: no WN_FLAG_DW_F16PAIR, no "
                                      "WN_FLAG_MM_F16PAIR): fp32-equivalent to round-off; what each fp16-pair launch of the default redoes "
                                      "itself with when an operand leaves fp16's range; same gates (tests/test_gpu_fullsize.py, bench "
                                      "`parity.dw_six_products`)")
            mm6 = alt_mode(base_flags & ~(_lib.FLAG_MM_F16PAIR | _lib.FLAG_FUSED_F16PAIR),
                           "six bf16 products for every forward / data-gradient contraction (no WN_FLAG_MM_F16PAIR, no "
                           "WN_FLAG_FUSED_F16PAIR), fp16 pairs for the weight gradients: the default of round 5")
            dw3 = alt_mode(six_flags | _lib.FLAG_DW_3PRODUCT,
                           "WN_FLAG_DW_3PRODUCT (opt-in): two bf16 pieces, three products (2^-16 per product): meets the 3e-5 gradient "
                           "gate, misses the golden after-Adam gate (1e-2 lr) by 2x -- not a default, not the metric")
            out["extras"] = extra_workloads(model, opt, red)
            out["extras"]["dw_six_products"] = dw6          # (key kept from round 5: now "six products everywhere")
            out["extras"]["all_six_products"] = dw6
            out["extras"]["mm_six_products"] = mm6
            out["extras"]["dw_3product"] = dw3
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import recipe_bench
                prev_env = os.environ.get("WN_ENGINE_FLAGS")
                os.environ["WN_ENGINE_FLAGS"] = str(int(six_flags))
                r6 = recipe_bench.measure(resch=512, kernel_size=2, upsampling=80, T=23040, batch=4, steps=3, with_kernels=False)
                out["extras"]["recipe_size_dw_six_products"] = {k: r6[k] for k in ("model", "B", "T", "ms_per_step", "samples_per_sec")}
            except Exception as e:  # noqa: BLE001
                out["extras"]["recipe_size_dw_six_products"] = {"error": repr(e)}
            finally:
                if prev_env is None:
                    os.environ.pop("WN_ENGINE_FLAGS", None)
                else:
                    os.environ["WN_ENGINE_FLAGS"] = prev_env
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
