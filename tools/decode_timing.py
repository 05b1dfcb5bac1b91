#!/usr/bin/env python
"""Cycle stamps inside one decode step (instrumented build of wn_decode.hip, -DWN_TIMING).

    python tools/decode_timing.py --build-only    # cross-compile here
    gpurun -- python tools/decode_timing.py       # run on the GPU box
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "tools", "exp")
SO = os.path.join(EXP, "libwn_timing.so")


def build():
    os.makedirs(EXP, exist_ok=True)
    csrc = os.path.join(ROOT, "pytorchwavenetvocoder_amd", "csrc")
    objs = []
    for name in ("wn_gemm", "wn_gemm6", "wn_elem", "wn_fused", "wn_decode", "wn_prof", "wn_api"):
        obj = os.path.join(EXP, name + ".timing.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DWN_TIMING", "-c",
                               os.path.join(csrc, name + ".hip"), "-o", obj])
        objs.append(obj)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)


if "--build-only" in sys.argv:
    build()
    sys.exit(0)
sys.path.insert(0, ROOT)
os.environ["WN_LIB_PATH"] = SO
import torch  # noqa: E402

from pytorchwavenetvocoder_amd.nets import WaveNet, initialize  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1)
m = WaveNet(256, 80, 64, 256, 10, 3, 2, 80)
m.apply(initialize)
m.to(dev)
dbg = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
lib = m.engine.lib
lib.lib.wn_decode_debug_set_buffer.argtypes = [ctypes.c_void_p]
lib.lib.wn_decode_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
x = torch.full((1, 1), 128, dtype=torch.int64, device=dev)
h = torch.randn(1, 80, 60, device=dev)
m.engine.decode(x, h, [600], chunk=1000)
torch.cuda.synchronize()
d = dbg.cpu().view(8, 64)
names = {0: "step start", 1: "phase A done (before barrier)", 2: "after barrier", 9: "layer 2 end", 10: "L3 dil dots done",
         11: "L3 dil group sums", 12: "L3 gate written", 13: "L3 after barrier", 14: "L3 res dots", 15: "L3 res group sum",
         16: "L3 skip dots", 17: "L3 after barrier", 3: "all layers done", 4: "post net done", 5: "token chosen", 6: "step end"}
order = [0, 1, 2, 9, 10, 11, 12, 13, 14, 15, 16, 17, 3, 4, 5, 6]
for w in (0, 3, 7):
    t0 = int(d[w, 0])
    prev = t0
    print("wave %d" % w)
    for i in order:
        v = int(d[w, i])
        print("   %-32s @%7d  (+%6d)" % (names[i], v - t0, v - prev))
        prev = v
