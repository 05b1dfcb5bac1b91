#!/usr/bin/env python
"""Cycle stamps of the k-loop phases of the split GEMM kernels (k_gemm6 / k_gemm6_dw) on the GPU.

Uses the instrumented build of tools/phase_timing.py (tools/exp/libwn_timing.so, -DWN_TIMING): the block that owns
logical tile (0, 0, 0) of every launch with the selected tag stamps, per wave and k-step, the cycle counter at
  0 loop top | 3 the step's MFMAs (with the loads and the split between them) issued |
  4 vmcnt wait (weight slab) done | 5 barrier passed

    python tools/phase_timing.py --build-only          # here: cross-compile
    gpurun -- python tools/gemm_timing.py [tag ...]    # on the GPU box (default tags below)
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "exp", "libwn_timing.so")
sys.path.insert(0, ROOT)
os.environ["WN_LIB_PATH"] = SO
import torch  # noqa: E402

from pytorchwavenetvocoder_amd.nets import WaveNet, initialize  # noqa: E402

tags = [a for a in sys.argv[1:] if not a.startswith("-")] or ["fwd_skip_sum", "bwd_dz_skip_all", "dw_dilated", "dw_skip", "dw_post1"]
dev = "cuda:0"
torch.manual_seed(1)
m = WaveNet(256, 80, 64, 256, 10, 3, 2, 80); m.apply(initialize); m.to(dev)
B, T = 8, 20000
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
    print("== %s" % tag)
    for w in range(4):
        rows = []
        for s in range(2, 22):
            e = [int(v) for v in d[w, s]]
            prev5 = int(d[w, s - 1, 5])
            if e[3] == 0 or e[5] == 0 or prev5 == 0:
                continue
            top = e[0] if e[0] else prev5
            rows.append((e[3] - top, (e[4] - e[3]) if e[4] else 0, e[5] - (e[4] if e[4] else e[3]), e[5] - prev5))
        if not rows:
            print("  wave %d: no stamps" % w)
            continue
        n = len(rows)
        mean = [sum(r[i] for r in rows) / n for i in range(4)]
        print("  wave %d (%2d steps): MFMAs with the loads and the split inside %5.0f | vmcnt(weight slab) %5.0f | barrier %5.0f | step total %5.0f cycles" % ((w, n) + tuple(mean)))
    print("  steps of wave 0:", [int(d[0, s, 5] - d[0, s - 1, 5]) for s in range(3, 20) if d[0, s, 5] and d[0, s - 1, 5]])
