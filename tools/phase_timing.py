#!/usr/bin/env python
"""Per-phase cycle stamps of the fused forward kernel (k_resblock_fwd) on the GPU.

Builds an instrumented copy of the library (wn_fused.hip compiled with -DWN_TIMING, which adds
s_memtime / s_memrealtime stamps at the phase boundaries of block 0 and entry/exit stamps for every
block) into tools/exp/libwn_timing.so and runs cfg2-size forwards through it.

    python tools/phase_timing.py --build-only      # here (no GPU): cross-compile
    gpurun -- python tools/phase_timing.py         # on the GPU box: run (uses the prebuilt .so)
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
    procs = []
    for name in ("wn_gemm", "wn_gemm6", "wn_elem", "wn_fused", "wn_decode", "wn_dlp", "wn_dlpm", "wn_dlpf", "wn_prof", "wn_api"):
        obj = os.path.join(EXP, name + ".timing.o")
        procs.append(subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-DWN_TIMING", "-c",
                                       os.path.join(csrc, name + ".hip"), "-o", obj]))
        objs.append(obj)
    for pr in procs:
        if pr.wait() != 0:
            raise SystemExit("hipcc failed")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)


if "--build-only" in sys.argv or not os.path.exists(SO):
    build()
    if "--build-only" in sys.argv:
        sys.exit(0)
sys.path.insert(0, ROOT)
os.environ["WN_LIB_PATH"] = SO
import torch  # noqa: E402

from pytorchwavenetvocoder_amd.nets import WaveNet, initialize
dev = "cuda:0"
torch.manual_seed(1)
# python tools/phase_timing.py [kernel_size upsampling T]   (default 2 80 23040; configs[3]: 3 256 26112)
_a = [int(v) for v in sys.argv[1:] if v.isdigit()]
KS, UP, T = (_a + [2, 80, 23040])[:3] if len(_a) >= 3 else (2, 80, 23040)
m = WaveNet(256, 80, 64, 256, 10, 3, KS, UP); m.apply(initialize); m.to(dev)
B = 8
print("kernel_size %d, upsampling %d, B = %d, T = %d" % (KS, UP, B, T))
x = torch.randint(0, 256, (B, T), device=dev); h = torch.randn(B, 80, T // UP, device=dev)
dbg = torch.zeros(8 * 4 * 16 + 256 * 4, dtype=torch.int64, device=dev)
lib = m.engine.lib
lib.lib.wn_debug_set_buffer.argtypes = [ctypes.c_void_p]
for it in range(3):
    lib.lib.wn_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
    m.engine.forward(x, h)
    torch.cuda.synchronize()
blk = dbg.cpu()[512:].view(256, 4)
blk = blk[blk[:, 0] != 0]   # only the blocks that ran (the persistent grid is smaller than 256)
d = dbg.cpu()[:512].view(8, 4, 16)   # stamps of the LAST layer launch (each launch overwrites)
t0 = int(d[:, 0, 0].min())
names = ["start", "hist", "cur", "gate", "done"]   # WN_FWD_V2=1: tap 0 (both passes) | tap 1 (+ gate of channels 0..31) | res 1x1 (+ gate of 32..63) | stores
for w in range(8):
    for t in range(3):
        st = [int(d[w, t, i]) - t0 for i in range(5)]
        if d[w, t, 4] == 0: continue
        print("wave %d tile %d: start@%7d | hist %6d | cur %6d | gate %6d | res+store %6d | total %6d" % (
            w, t, st[0], st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[4] - st[0]))

for w in range(8):
    e = d[w, 0]
    cyc = int(e[9] - e[5]); rt = int(e[8] - e[7])
    if rt <= 0:   # (k_resblock_fwd_v2 keeps the per-tile stamps only)
        continue
    print("wave %d: prologue %6d cyc | entry->exit %7d cyc, %6d realtime ticks (100 MHz => %.1f us, clock %.2f GHz) | entry@%d" % (
        w, int(e[6] - e[5]), cyc, rt, rt / 100.0, cyc / (rt / 100.0) / 1e3, int(e[5]) - t0))

r0 = int(blk[:, 0].min())
ent = (blk[:, 0] - r0).float() / 100.0
ex0 = (blk[:, 1] - r0).float() / 100.0
ex7 = (blk[:, 2] - r0).float() / 100.0
print("block entry  us: min %.1f max %.1f mean %.1f" % (ent.min(), ent.max(), ent.mean()))
print("block exit0  us: min %.1f max %.1f mean %.1f" % (ex0.min(), ex0.max(), ex0.mean()))
print("block exit7  us: min %.1f max %.1f mean %.1f" % (ex7.min(), ex7.max(), ex7.mean()))
print("tiles by wave0 per block:", torch.bincount(blk[:, 3]).tolist())
dur = torch.maximum(ex0, ex7) - ent
print("block duration us: min %.1f max %.1f mean %.1f" % (dur.min(), dur.max(), dur.mean()))
order = torch.argsort(dur)
print("slowest blocks:", [(int(i), round(float(dur[i]), 1), round(float(ent[i]), 1)) for i in order[-8:]])
print("fastest blocks:", [(int(i), round(float(dur[i]), 1), round(float(ent[i]), 1)) for i in order[:8]])
