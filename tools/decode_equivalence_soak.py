#!/usr/bin/env python
"""Long equivalence run of the persistent any-size decode against the layer-wise launches, recipe-size model (n_resch 512):
every logit of every step of every utterance (max abs difference; the two paths differ in summation order only), and the share
of equal tokens (random weights give near-uniform logits, so argmax ties flip: reported, not gated).

    python tools/decode_equivalence_soak.py [--steps 3000] [--batches 1,3,48]            (GPU)
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorchwavenetvocoder_amd.nets import WaveNet, initialize  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--batches", default="1,3,48")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    m = WaveNet(256, 80, 512, 256, 10, 3, 2, 80)
    m.apply(initialize)
    with torch.no_grad():   # sharper logits than the initialisation's: fewer argmax ties
        m.conv_post_2.weight.mul_(30.0)
    m.to(dev)
    for B in [int(v) for v in a.batches.split(",")]:
        n = a.steps
        x = torch.randint(0, 256, (B, 7), device=dev)
        h = torch.randn(B, 80, (n + 7) // 80 + 2, device=dev)
        ns = [n - 13 * (b % 4) for b in range(B)]
        tp, lp = m.engine.decode(x, h, ns, return_logits=True, layered=True)
        tl, ll = m.engine.decode(x, h, ns, return_logits=True, layered="launches")
        # teacher-force the launches' tokens? no: both generate freely; compare only up to the first differing token per utterance
        worst, equal, total, compared = 0.0, 0, 0, 0
        for b in range(B):
            same = (tp[b] == tl[b])
            first = int((~same).nonzero()[0]) if (~same).any() else len(same)
            equal += int(same.sum()); total += len(same)
            upto = min(first + 1, len(same))   # logits of step i depend on the tokens before i only
            compared += upto
            if upto > 0:
                worst = max(worst, float((lp[b][:upto] - ll[b][:upto]).abs().max()))
        print(json.dumps({"batch": B, "steps": n, "logit_rows_compared": compared, "max_abs_logit_diff": worst,
                          "tokens_equal_share": equal / total}), flush=True)
        assert worst <= 1e-3, worst


if __name__ == "__main__":
    main()
