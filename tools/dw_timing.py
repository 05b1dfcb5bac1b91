#!/usr/bin/env python
"""Cycle stamps of the k-loop of the weight-gradient kernel k_gemm6_dw at the recipes' size (n_resch 512: the 256 x 256 tile).

A -DWN_TIMING build of wn_gemm6.hip (tools/exp/libwn_dwtiming.so; the other objects are the product's) stamps, for the block
that owns logical tile (0, 0, 0) of the launches with the selected tag, per wave and k-step: 0 = step begins, 3 = the step's
MFMAs (with the loads / the operand split between them) issued, 5 = barrier passed.  s_memtime ticks (100 MHz).

    python tools/dw_timing.py --build-only        # here: cross-compile
    gpurun -- python tools/dw_timing.py [tag ...] # on the GPU box (default: dw_dilated dw_res)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "tools", "exp")
VARIANT = os.environ.get("WN_DWT_VARIANT", "")   # "", NOLOAD, NOSPLIT, NOPUT: what-if builds that drop one part of the k-step (timing only)
SO = os.path.join(EXP, "libwn_dwtiming%s.so" % VARIANT)
CSRC = os.path.join(ROOT, "pytorchwavenetvocoder_amd", "csrc")


def build():
    os.makedirs(EXP, exist_ok=True)
    obj = os.path.join(EXP, "wn_gemm6.dwtiming%s.o" % VARIANT)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-DWN_TIMING"] +
                          (["-DWN_DWX_" + VARIANT] if VARIANT else []) +
                          ["-c", os.path.join(CSRC, "wn_gemm6.hip"), "-o", obj])
    objs = [obj if n == "wn_gemm6" else os.path.join(CSRC, n + ".o")
            for n in ("wn_gemm", "wn_gemm6", "wn_elem", "wn_fused", "wn_decode", "wn_dlp", "wn_dlpm", "wn_dlpf", "wn_prof", "wn_api")]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)


if "--build-only" in sys.argv or not os.path.exists(SO):
    build()
    if "--build-only" in sys.argv:
        sys.exit(0)
sys.path.insert(0, ROOT)
os.environ["WN_LIB_PATH"] = SO
import torch  # noqa: E402

from pytorchwavenetvocoder_amd.nets import WaveNet, initialize  # noqa: E402

tags = [a for a in sys.argv[1:] if not a.startswith("-")] or ["dw_dilated", "dw_res"]
dev = "cuda:0"
torch.manual_seed(1)
m = WaveNet(256, 80, 512, 256, 10, 3, 2, 80); m.apply(initialize); m.to(dev)
B, T = 4, 23040
x = torch.randint(0, 256, (B, T), device=dev); h = torch.randn(B, 80, T // 80, device=dev)
t = torch.randint(0, 256, (B, T), device=dev)
lib = m.engine.lib
lib.lib.wn_debug_gemm6.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
for _ in range(2):
    m.loss_and_backward(x, h, t)
torch.cuda.synchronize()
for tag in tags:
    dbg = torch.zeros(4 * 256, dtype=torch.int64, device=dev)
    lib.lib.wn_debug_gemm6(ctypes.c_void_p(dbg.data_ptr()), tag.encode())
    m.loss_and_backward(x, h, t)
    torch.cuda.synchronize()
    lib.lib.wn_debug_gemm6(None, b"")
    d = dbg.cpu().view(4, 32, 8)
    print("== %s (engine flags %d) %s" % (tag, m.engine.flags, VARIANT and ("what-if build: " + VARIANT)))
    for w in range(4):
        rows = []
        for s in range(2, 22):
            e = [int(v) for v in d[w, s]]
            prev5 = int(d[w, s - 1, 5])
            if e[3] == 0 or e[5] == 0 or prev5 == 0:
                continue
            top = e[0] if e[0] else prev5
            rows.append((e[3] - top, e[5] - e[3], e[5] - prev5))
        if not rows:
            print("  wave %d: no stamps" % w)
            continue
        n = len(rows)
        mean = [sum(r[i] for r in rows) / n for i in range(3)]
        print("  wave %d (%2d steps): step body (MFMAs + loads + split) %6.1f | barrier %6.1f | step total %6.1f ticks" % ((w, n) + tuple(mean)))
    print("  steps of wave 0:", [int(d[0, s, 5] - d[0, s - 1, 5]) for s in range(3, 20) if d[0, s, 5] and d[0, s - 1, 5]])
