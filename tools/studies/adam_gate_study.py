#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""TEST INFRASTRUCTURE (a study, not collected by pytest).  How far is an fp32 evaluation of the BENCHMARK INSTANCE's training
step from its fp64 evaluation, in the units of the after-Adam gate (|dw| / lr after one Adam step, lr 1e-4)?

    python tools/studies/adam_gate_study.py truth  OUT.pt     # CPU, ~10 min / ~35 GB: fp64 oracle + the reference module in fp32
    python tools/studies/adam_gate_study.py hip    OUT.pt     # GPU box: the HIP step's gradients (default arithmetic, six products)
    python tools/studies/adam_gate_study.py compare TRUTH.pt HIP.pt

The instance is bench.py's: model.apply(initialize) under seed 1, synthetic_minibatch(8, 23040, 288, rank 0)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CFG_T = tuple(bench.CFG2[k] for k in ("n_quantize", "n_aux", "n_resch", "n_skipch", "dilation_depth", "dilation_repeat",
                                      "kernel_size", "upsampling_factor"))
LR, EPS = 1e-4, 1e-8


def instance():
    from oracle import ref_step as RS
    ref = RS.load_reference()
    torch.manual_seed(1)
    m = ref.WaveNet(*CFG_T)
    m.apply(ref.initialize)
    init = {k: v.detach().clone() for k, v in m.state_dict().items()}
    bl, frames, T = bench.geometry(m.receptive_field, bench.BATCH_LENGTH, CFG_T[7])
    return init, bench.synthetic_minibatch(bench.BATCH_PER_GPU, T, frames, 0)


def upd(g):
    """first Adam step with bias correction: m / (sqrt(v) + eps) = g / (|g| + eps)"""
    return g / (g.abs() + EPS)


def main():
    what = sys.argv[1]
    if what == "truth":
        from oracle import same_run_parity as SRP
        from oracle import wavenet_oracle as O
        init, (x, h, t) = instance()
        torch.set_num_threads(int(os.environ.get("WN_STUDY_THREADS", "8")))
        r32 = SRP.reference_step(CFG_T, init, x, h, t, lr=LR)
        masks = ((r32["relu_skip"] > 0).float(), (r32["relu_post1"] > 0).float())   # the reference's own sub-gradient choice
        g32 = {k: v for k, v in r32["grads"].items() if v is not None}
        del r32
        cfg = O.OracleConfig(*CFG_T)
        p64 = {k: v.double() for k, v in init.items()}
        _, _, g64 = O.train_step(cfg, p64, None, x, h.double(), t, relu_masks=(masks[0].double(), masks[1].double()))
        torch.save({"g32": g32, "g64": {k: v for k, v in g64.items() if v is not None}}, sys.argv[2])
    elif what == "hip":
        from oracle import same_run_parity as SRP
        from pytorchwavenetvocoder_amd import _lib as L
        from pytorchwavenetvocoder_amd.engine import DEFAULT_FLAGS
        from pytorchwavenetvocoder_amd.nets import WaveNet, initialize
        from pytorchwavenetvocoder_amd.optim import FusedAdam
        init, (x, h, t) = instance()
        torch.manual_seed(1)
        model = WaveNet(*CFG_T)
        model.apply(initialize)
        model.to("cuda:0")
        ref = SRP.reference_step(CFG_T, init, x, h, t, lr=LR, threads=16)
        out = {}
        for name, flags in (("default", DEFAULT_FLAGS), ("six", DEFAULT_FLAGS & ~(L.FLAG_DW_F16PAIR | L.FLAG_DW_3PRODUCT))):
            r = SRP.gpu_step_vs_reference(model, lambda m, lr: FusedAdam(m, lr=lr), ref, x, h, t, init, flags, lr=LR,
                                          layers_per_bucket=bench.LAYERS_PER_BUCKET)
            print(name, {k: r[k] for k in ("worst_grad_rel", "after_adam_maxabs_over_lr", "after_adam_elements_over_gate")})
            out[name] = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
        out["ref16"] = {k: v for k, v in ref["grads"].items() if v is not None}
        torch.save(out, sys.argv[2])
    elif what == "all":
        # GPU box, everything on ONE set of sub-gradient choices (the reference's, 16 threads): reference fp32, HIP (default / six
        # products / + WN_FLAG_MM_F16PAIR) and the fp64 evaluation of the same step, compared in the after-Adam gate's units
        avail = 0
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = int(line.split()[1]) // (1 << 20)
        print("host MemAvailable %d GiB, %d CPUs" % (avail, len(os.sched_getaffinity(0))))
        if avail < 120:
            raise SystemExit("not enough host memory for the fp64 evaluation (needs ~40 GiB; asks for 120 to be safe)")
        from oracle import same_run_parity as SRP
        from oracle import wavenet_oracle as O
        from pytorchwavenetvocoder_amd import _lib as L
        from pytorchwavenetvocoder_amd.engine import DEFAULT_FLAGS
        from pytorchwavenetvocoder_amd.nets import WaveNet, initialize
        from pytorchwavenetvocoder_amd.optim import FusedAdam
        init, (x, h, t) = instance()
        torch.manual_seed(1)
        model = WaveNet(*CFG_T)
        model.apply(initialize)
        model.to("cuda:0")
        ref = SRP.reference_step(CFG_T, init, x, h, t, lr=LR, threads=16)
        sets = {"reference fp32 (16 threads)": {k: v for k, v in ref["grads"].items() if v is not None}}
        six = DEFAULT_FLAGS & ~(L.FLAG_DW_F16PAIR | L.FLAG_DW_3PRODUCT)
        for name, flags in (("HIP default", DEFAULT_FLAGS), ("HIP six products", six), ("HIP default + MM_F16PAIR", DEFAULT_FLAGS | L.FLAG_MM_F16PAIR)):
            r = SRP.gpu_step_vs_reference(model, lambda m, lr: FusedAdam(m, lr=lr), ref, x, h, t, init, flags, lr=LR,
                                          layers_per_bucket=bench.LAYERS_PER_BUCKET)
            print(name, "vs reference:", {k: r[k] for k in ("worst_grad_rel", "after_adam_maxabs_over_lr", "after_adam_elements_over_gate", "kink_flips")})
            sets[name] = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
        masks = ((ref["relu_skip"] > 0).double(), (ref["relu_post1"] > 0).double())
        del ref, model
        torch.cuda.empty_cache()
        torch.set_num_threads(32)
        cfg = O.OracleConfig(*CFG_T)
        p64 = {k: v.double() for k, v in init.items()}
        _, _, g64 = O.train_step(cfg, p64, None, x, h.double(), t, relu_masks=masks)
        g64 = {k: v for k, v in g64.items() if v is not None}
        report(sets, g64)
    else:
        tr = torch.load(sys.argv[2])
        hp = torch.load(sys.argv[3]) if len(sys.argv) > 3 else {}
        sets = {"reference fp32 (8 threads, this host)": tr["g32"]}
        for k, v in hp.items():
            sets["HIP " + k if k != "ref16" else "reference fp32 (16 threads, GPU box host)"] = v
        report(sets, tr["g64"])


def report(sets, g64s):
    for name, gs in sets.items():
        worst, wk, n_over, wrel = 0.0, None, 0, 0.0
        for k, g64 in g64s.items():
            d = (upd(gs[k].double()) - upd(g64)).abs()
            n_over += int((d > 1e-2).sum())
            if float(d.max()) > worst:
                worst, wk = float(d.max()), k
            wrel = max(wrel, float((gs[k].double() - g64).abs().max() / g64.abs().max()))
        print("%-44s vs fp64: after-Adam %.4f lr (%s), %d elements over 1e-2 lr, worst gradient rel %.3g" % (name, worst, wk, n_over, wrel))


if __name__ == "__main__":
    main()
