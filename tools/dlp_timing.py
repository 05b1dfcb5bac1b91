#!/usr/bin/env python
"""Phase stamps of one step of the persistent any-size decode (csrc/wn_dlp.hip, wn_dlpm.hip, wn_dlpf.hip -- whichever the batch
takes), recipe-size model.  Needs a timing build:
    bash tools/build_variant.sh dlptiming "-DWN_DLP_TIMING -fno-slp-vectorize" wn_dlp.hip wn_dlpm.hip wn_dlpf.hip
    WN_LIB_PATH=tools/exp/libwn_dlptiming.so python tools/dlp_timing.py [B]                      (GPU)
Prints, per stage of step p0 + 3 of unit 0: microseconds spent in gather / barrier / dot products (incl. the wait for the
stage's weights) / partial sums / epilogue + publish / closing barrier."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorchwavenetvocoder_amd.nets import WaveNet, initialize  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    m = WaveNet(256, 80, 512, 256, 10, 3, 2, 80)
    m.apply(initialize)
    m.to(dev)
    x = torch.full((B, 1), 128, dtype=torch.int64, device=dev)
    h = torch.randn(B, 80, 4, device=dev)
    m.engine.decode(x, h, [40] * B, layered=True)
    torch.cuda.synchronize()
    st = m.engine.last_decode_state
    eoff = m.engine.lib.wn_decode_layered_error_offset(ctypes.byref(m.engine.cfg), B, 0)
    stamps = st[eoff + 16:eoff + 16 + 2 * 8 * 40].view(torch.int64).view(40, 8).cpu()
    L = m.engine.n_layers
    print("B = %d; us per phase (100 MHz wall clock): stage | gather  barrier  dots(+weights)  partials  epilogue  barrier | total" % B)
    tot = 0.0
    for s in range(L + 1):
        r = stamps[s].tolist()
        d = [(r[i + 1] - r[i]) * 0.01 for i in range(6)]
        tot += (r[6] - r[0]) * 0.01
        print("%3d | %s | %.2f" % (s, "  ".join("%6.2f" % v for v in d), (r[6] - r[0]) * 0.01))
    print("stages: %.1f us; post net + token choice: %.1f us; whole step: %.1f us" % (
        tot, (stamps[L + 2][0] - stamps[L + 1][0]).item() * 0.01, (stamps[L + 2][0] - stamps[0][0]).item() * 0.01))


if __name__ == "__main__":
    main()
