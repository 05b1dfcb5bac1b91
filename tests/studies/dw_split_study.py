# -*- coding: utf-8 -*-
"""Numerics study behind WN_FLAG_DW_F16PAIR: how should the weight-gradient contractions split their operands?

Takes the ORACLE's own tensors (oracle/wavenet_oracle.py on CPU: test infrastructure, not the product path) of the 30-layer
64/256 model -- the gradient operand A and the activation operand B of every kind of weight-gradient contraction
(csrc/wn_api_backward.inl: dw_post2, dw_post1, dw_skip and, for three layers, dw_dilated / dw_res / dw_aux) -- and evaluates
G = sum_k A(m,k) B(n,k) in fp64 from (a) three bf16 pieces / six products, (b) two bf16 pieces / three products
(WN_FLAG_DW_3PRODUCT), (c) a plain fp32 running sum, (d) two fp16 pieces / three products with the gradient operand scaled by
2^E / (bound on dlogits) for three values of E.  Printed per contraction: the operands' ranges relative to the bound, the
maximum error relative to the largest element, and the quantity the golden after-Adam gate (1e-2 lr) sees --
max |dG| / (|G| + 1e-8), Adam's update being lr g / (|g| + eps).

    python tests/studies/dw_split_study.py [init_scale B T]      (CPU, ~1 min; output: profiles/r05/dw_split_study.txt)
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import wavenet_oracle as O
import torch.nn.functional as F
torch.manual_seed(0)
cfg_t = (256, 80, 64, 256, 10, 3, 2, 80)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
B, T = int(sys.argv[2]) if len(sys.argv) > 2 else 2, int(sys.argv[3]) if len(sys.argv) > 3 else 8000
cfg = O.OracleConfig(*cfg_t)
p = O.random_params(cfg, 5, scale=scale)
p = {k: v.clone().requires_grad_(True) for k, v in p.items()}
x, h, t = O.synthetic_batch(cfg, B, T, 7)
dtype = torch.float32
out = O.causal_conv1d(O.onehot(x, cfg.n_quantize, dtype).transpose(1, 2), p["causal.conv.weight"], p["causal.conv.bias"], 1)
hu = O.upsampling(h, p["upsampling.conv.weight"], p["upsampling.conv.bias"])
Xs, Ps, Zs, skips = [], [], [], []
for l, d in enumerate(cfg.dilations):
    xin = out; xin.retain_grad(); Xs.append(xin)
    os_ = O.causal_conv1d(xin, p["dil_sigmoid.%d.conv.weight" % l], p["dil_sigmoid.%d.conv.bias" % l], d) + F.conv1d(hu, p["aux_1x1_sigmoid.%d.weight" % l], p["aux_1x1_sigmoid.%d.bias" % l])
    ot_ = O.causal_conv1d(xin, p["dil_tanh.%d.conv.weight" % l], p["dil_tanh.%d.conv.bias" % l], d) + F.conv1d(hu, p["aux_1x1_tanh.%d.weight" % l], p["aux_1x1_tanh.%d.bias" % l])
    os_.retain_grad(); ot_.retain_grad(); Ps.append((os_, ot_))
    z = torch.sigmoid(os_) * torch.tanh(ot_); z.retain_grad(); Zs.append(z)
    skips.append(F.conv1d(z, p["skip_1x1.%d.weight" % l], p["skip_1x1.%d.bias" % l]))
    out = F.conv1d(z, p["res_1x1.%d.weight" % l], p["res_1x1.%d.bias" % l]) + xin
sk = sum(skips); sk.retain_grad()
o1 = F.relu(sk); o1.retain_grad()
q1 = F.conv1d(o1, p["conv_post_1.weight"], p["conv_post_1.bias"]); q1.retain_grad()
o2 = F.relu(q1)
lg = F.conv1d(o2, p["conv_post_2.weight"], p["conv_post_2.bias"]); lg.retain_grad()
rf = cfg.receptive_field
loss = F.cross_entropy(lg.transpose(1, 2)[:, rf:].contiguous().view(-1, cfg.n_quantize), t[:, rf:].contiguous().view(-1))
loss.backward()
gs = 1.0 / (B * (T - rf))
print("loss %.4f gs %.3g = 2^%.1f" % (float(loss), gs, math.log2(gs)))

def bf(x): return x.bfloat16().float()
def hf(x): return x.half().float()
def split_bf2(x):
    h_ = bf(x); m = bf(x - h_); return h_, m
def split_bf3(x):
    h_ = bf(x); r = x - h_; m = bf(r); l = bf(r - m); return h_, m, l
def split_h2(x, s):
    xs = x * s; h_ = hf(xs); l = hf(xs - h_); return h_, l
def contract(a, b):  # a (B,M,T), b (B,N,T) -> (M,N) in fp64
    return torch.einsum("bmt,bnt->mn", a.double(), b.double())
def study(name, A, Bm, E=8):   # E = WN_DW_F16_HEADROOM of include/wavenet_hip.h, then 64 x less and 64 x more
    A = A.detach(); Bm = Bm.detach()
    ex = contract(A, Bm)
    amax = float(A.abs().max()); arms = float(A.pow(2).mean().sqrt())
    print("%-10s A max/gs 2^%.1f rms/gs 2^%.1f  | B max %.3g rms %.3g | G max %.3g" % (name, math.log2(amax / gs), math.log2(arms / gs), float(Bm.abs().max()), float(Bm.pow(2).mean().sqrt()), float(ex.abs().max())))
    def rep(tag, g):
        d = (g - ex).abs()
        adam = (d / (ex.abs() + 1e-8)).max()
        print("    %-22s max err / max %.3g   adam-metric max %.3g  (rel to 1e-2: %.2f)" % (tag, float(d.max() / ex.abs().max()), float(adam), float(adam) / 1e-2))
    ah, am, al = split_bf3(A); bh, bm_, bl = split_bf3(Bm)
    rep("bf16 6-product", contract(ah, bh) + contract(ah, bm_) + contract(am, bh) + contract(am, bm_) + contract(ah, bl) + contract(al, bh))
    rep("bf16 3-product", contract(ah, bh) + contract(ah, bm_) + contract(am, bh))
    rep("fp32 (float sum)", torch.einsum("bmt,bnt->mn", A, Bm).double())
    for E_ in (E, E - 6, E + 6):
        s = 2.0 ** (E_ - math.floor(math.log2(gs)))
        fh, fl = split_h2(A, s); gh, gl = split_h2(Bm, 1.0)
        ok = bool(torch.isfinite(fh).all())
        rep("f16 pair E=%d%s" % (E_, "" if ok else " OVERFLOW"), (contract(fh, gh) + contract(fh, gl) + contract(fl, gh)) / s)
L = len(cfg.dilations)
study("dw_post2", lg.grad, o2)
study("dw_post1", q1.grad, o1)
study("dw_skip", sk.grad, Zs[L - 1])
for l in (0, 15, L - 2):
    dP = torch.cat([Ps[l][0].grad, Ps[l][1].grad], 1)
    study("dw_dil.%d" % l, dP, Xs[l])
    study("dw_res.%d" % l, Xs[l + 1].grad, Zs[l])
    study("dw_aux.%d" % l, dP, hu)
