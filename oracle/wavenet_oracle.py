# -*- coding: utf-8 -*-
"""CPU ORACLE for the WaveNet-vocoder training hot path.  ** TEST INFRASTRUCTURE ONLY **

This file is a CPU restatement (plain torch ops on CPU tensors + numpy) of the
reference algorithm in kan-bayashi/PytorchWaveNetVocoder for the path named by
BASELINE.json: model forward, softmax cross-entropy on [:, receptive_field:],
backward, and one Adam step.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / reported baseline.  The
product package (``pytorchwavenetvocoder_amd``) never imports it and has no CPU
fallback.

Parity pin: this oracle is checked against outputs of the REFERENCE ITSELF
(``/root/reference/wavenet_vocoder/nets/wavenet.py`` imported in the build
container) stored under ``tests/golden/*.npz`` by ``tests/golden/make_golden.py``
(see ``tests/test_oracle_golden.py``).  The arithmetic lives in the third-party
dependency ``torch`` (reference pins torch==1.0.1, ``tools/Makefile:8``); this
oracle runs the same torch ops of torch 2.10 CPU, fp32 (or fp64 for the noise
floor).

Every function cites the reference file:line it follows (paths relative to the
reference repository root).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# mu-law codec  (wavenet_vocoder/nets/wavenet.py:17-47)
# --------------------------------------------------------------------------
def encode_mu_law(x, mu=256):
    """wavenet.py:17-30  sign(x) ln(1+mu|x|)/ln(1+mu) -> floor((fx+1)/2*mu+0.5)."""
    mu = mu - 1
    fx = np.sign(x) * np.log(1 + mu * np.abs(x)) / np.log(1 + mu)
    return np.floor((fx + 1) / 2 * mu + 0.5).astype(np.int64)


def decode_mu_law(y, mu=256):
    """wavenet.py:33-47."""
    mu = mu - 1
    fx = (y - 0.5) / mu * 2 - 1
    x = np.sign(fx) / mu * ((1 + mu) ** np.abs(fx) - 1)
    return x


# --------------------------------------------------------------------------
# hyper-parameters / parameter inventory  (wavenet.py:172-210)
# --------------------------------------------------------------------------
class OracleConfig(object):
    """Same constructor arguments as WaveNet.__init__ (wavenet.py:172-173)."""

    def __init__(self, n_quantize=256, n_aux=28, n_resch=512, n_skipch=256,
                 dilation_depth=10, dilation_repeat=3, kernel_size=2, upsampling_factor=0, out_channels=0):
        # out_channels: NOT a reference argument; 3*n_mixture for the mixture-of-logistics head (0 = n_quantize)
        self.out_channels = out_channels if out_channels > 0 else n_quantize
        self.n_quantize = n_quantize
        self.n_aux = n_aux
        self.n_resch = n_resch
        self.n_skipch = n_skipch
        self.dilation_depth = dilation_depth
        self.dilation_repeat = dilation_repeat
        self.kernel_size = kernel_size
        self.upsampling_factor = upsampling_factor
        # wavenet.py:184-185
        self.dilations = [2 ** i for i in range(dilation_depth)] * dilation_repeat
        self.receptive_field = (kernel_size - 1) * sum(self.dilations) + 1

    def as_tuple(self):
        return (self.n_quantize, self.n_aux, self.n_resch, self.n_skipch, self.dilation_depth,
                self.dilation_repeat, self.kernel_size, self.upsampling_factor)


def param_shapes(cfg: OracleConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict keys/shapes in the reference's registration order (wavenet.py:187-210).

    causal.conv (R,Q,K); upsampling.conv (1,1,1,U)+(1,); per layer dil_sigmoid/dil_tanh.conv
    (R,R,K), aux_1x1_sigmoid/tanh (R,A,1), skip_1x1 (S,R,1), res_1x1 (R,R,1); conv_post_1
    (S,S,1), conv_post_2 (Q,S,1); every conv has a bias.
    """
    Q, A, R, S, K, U = (cfg.n_quantize, cfg.n_aux, cfg.n_resch, cfg.n_skipch,
                        cfg.kernel_size, cfg.upsampling_factor)
    L = len(cfg.dilations)
    d = OrderedDict()
    d["causal.conv.weight"] = (R, Q, K)
    d["causal.conv.bias"] = (R,)
    if U > 0:
        d["upsampling.conv.weight"] = (1, 1, 1, U)
        d["upsampling.conv.bias"] = (1,)
    for l in range(L):
        d["dil_sigmoid.%d.conv.weight" % l] = (R, R, K)
        d["dil_sigmoid.%d.conv.bias" % l] = (R,)
    for l in range(L):
        d["dil_tanh.%d.conv.weight" % l] = (R, R, K)
        d["dil_tanh.%d.conv.bias" % l] = (R,)
    for l in range(L):
        d["aux_1x1_sigmoid.%d.weight" % l] = (R, A, 1)
        d["aux_1x1_sigmoid.%d.bias" % l] = (R,)
    for l in range(L):
        d["aux_1x1_tanh.%d.weight" % l] = (R, A, 1)
        d["aux_1x1_tanh.%d.bias" % l] = (R,)
    for l in range(L):
        d["skip_1x1.%d.weight" % l] = (S, R, 1)
        d["skip_1x1.%d.bias" % l] = (S,)
    for l in range(L):
        d["res_1x1.%d.weight" % l] = (R, R, 1)
        d["res_1x1.%d.bias" % l] = (R,)
    d["conv_post_1.weight"] = (S, S, 1)
    d["conv_post_1.bias"] = (S,)
    d["conv_post_2.weight"] = (cfg.out_channels, S, 1)
    d["conv_post_2.bias"] = (cfg.out_channels,)
    return d


def init_params(cfg: OracleConfig, dtype=torch.float32, generator: Optional[torch.Generator] = None
                ) -> "OrderedDict[str, torch.Tensor]":
    """``model.apply(initialize)``  (wavenet.py:50-63): Conv1d xavier_uniform weight / zero
    bias; ConvTranspose2d weight 1 / bias 0."""
    params = OrderedDict()
    for k, shp in param_shapes(cfg).items():
        t = torch.zeros(shp, dtype=dtype)
        if k.startswith("upsampling"):
            if k.endswith("weight"):
                t.fill_(1.0)
        elif k.endswith("weight"):
            # xavier_uniform_: bound = sqrt(6 / (fan_in + fan_out)), fan = ch * kernel
            fan_out = shp[0] * shp[2]
            fan_in = shp[1] * shp[2]
            bound = math.sqrt(6.0 / (fan_in + fan_out))
            t.uniform_(-bound, bound, generator=generator)
        params[k] = t
    return params


def random_params(cfg: OracleConfig, seed: int, scale: float = 0.1, dtype=torch.float32
                  ) -> "OrderedDict[str, torch.Tensor]":
    """"trained-scale" weights for the harder parity set: N(0, scale) for every
    tensor incl. biases and a non-trivial upsampling kernel.  numpy RandomState so the stream is
    reproducible on any machine (fixtures store only the seed)."""
    rs = np.random.RandomState(seed)
    params = OrderedDict()
    for k, shp in param_shapes(cfg).items():
        a = rs.standard_normal(size=shp) * scale
        if k == "upsampling.conv.weight":
            a = 1.0 + a  # around the nearest-neighbour init
        params[k] = torch.from_numpy(a).to(dtype)
    return params


# --------------------------------------------------------------------------
# forward  (wavenet.py:212-241, 513-536)
# --------------------------------------------------------------------------
def causal_conv1d(x, weight, bias, dilation):
    """CausalConv1d.forward, wavenet.py:95-121: Conv1d(padding=(K-1)d, dilation=d) then drop the
    last (K-1)d outputs -> y[t] = b + sum_k W[:,:,k] x[t-(K-1-k)d], zero history."""
    K = weight.shape[2]
    pad = (K - 1) * dilation
    y = F.conv1d(x, weight, bias, padding=pad, dilation=dilation)
    if pad != 0:
        y = y[:, :, :-pad]
    return y


def onehot(x, depth, dtype):
    """OneHot.forward, wavenet.py:78-92 (x % depth, scatter 1)."""
    x = x % depth
    x = torch.unsqueeze(x, 2)
    oh = torch.zeros(x.size(0), x.size(1), depth, dtype=dtype)
    return oh.scatter_(2, x, 1)


def upsampling(h, weight, bias):
    """UpSampling.forward, wavenet.py:141-154: ConvTranspose2d(1,1,(1,U),stride (1,U))."""
    U = weight.shape[3]
    y = F.conv_transpose2d(h.unsqueeze(1), weight, bias, stride=(1, U))
    return y.squeeze(1)


def residual_forward(x, h, p, l, dilation):
    """WaveNet._residual_forward, wavenet.py:525-536."""
    out_s = causal_conv1d(x, p["dil_sigmoid.%d.conv.weight" % l], p["dil_sigmoid.%d.conv.bias" % l], dilation)
    out_t = causal_conv1d(x, p["dil_tanh.%d.conv.weight" % l], p["dil_tanh.%d.conv.bias" % l], dilation)
    aux_s = F.conv1d(h, p["aux_1x1_sigmoid.%d.weight" % l], p["aux_1x1_sigmoid.%d.bias" % l])
    aux_t = F.conv1d(h, p["aux_1x1_tanh.%d.weight" % l], p["aux_1x1_tanh.%d.bias" % l])
    z = torch.sigmoid(out_s + aux_s) * torch.tanh(out_t + aux_t)
    skip = F.conv1d(z, p["skip_1x1.%d.weight" % l], p["skip_1x1.%d.bias" % l])
    out = F.conv1d(z, p["res_1x1.%d.weight" % l], p["res_1x1.%d.bias" % l])
    return out + x, skip


class _ReluGivenSubgradient(torch.autograd.Function):
    """F.relu whose backward uses a GIVEN 0/1 sub-gradient instead of (x > 0).  The two coincide wherever x != 0 was
    decided the same way; an fp32 pre-activation within round-off of 0 can legitimately fall on either side in two
    correct evaluations, and a gradient comparison at sizes where such elements are certain to occur (config 2:
    8 x 19970 x 512 ReLU inputs) is only defined for a common choice -- the one of the implementation under test,
    whose every differing element the test then requires to be within round-off of the kink."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x.clamp(min=0)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


def forward(cfg: OracleConfig, p: Dict[str, torch.Tensor], x: torch.Tensor, h: torch.Tensor,
            return_intermediates: bool = False, relu_masks=None):
    """WaveNet.forward, wavenet.py:212-241.  x (B,T) int64, h (B,A,T) or (B,A,T/U) -> (B,T,Q).

    ``relu_masks = (m_skip, m_post1)`` (float 0/1 tensors (B,S,T); test option, not in the reference): the two ReLUs
    of _postprocess back-propagate with these sub-gradients (see _ReluGivenSubgradient); forward values are unchanged."""
    dtype = p["causal.conv.weight"].dtype
    # _preprocess, wavenet.py:513-516
    out = causal_conv1d(onehot(x, cfg.n_quantize, dtype).transpose(1, 2),
                        p["causal.conv.weight"], p["causal.conv.bias"], 1)
    if cfg.upsampling_factor > 0:
        h = upsampling(h, p["upsampling.conv.weight"], p["upsampling.conv.bias"])
    inter = {"x0": out, "h_up": h, "layer_out": [], "skip": []}
    skips = []
    for l, d in enumerate(cfg.dilations):
        out, skip = residual_forward(out, h, p, l, d)
        skips.append(skip)
        if return_intermediates:
            inter["layer_out"].append(out)
            inter["skip"].append(skip)
    out = sum(skips)
    inter["skip_sum"] = out          # pre-ReLU (the ReLU kinks matter for gradient comparisons)
    # _postprocess, wavenet.py:518-523
    out = F.relu(out) if relu_masks is None else _ReluGivenSubgradient.apply(out, relu_masks[0])
    out = F.conv1d(out, p["conv_post_1.weight"], p["conv_post_1.bias"])
    inter["post1_pre"] = out         # pre-ReLU
    out = F.relu(out) if relu_masks is None else _ReluGivenSubgradient.apply(out, relu_masks[1])
    out = F.conv1d(out, p["conv_post_2.weight"], p["conv_post_2.bias"]).transpose(1, 2)
    if return_intermediates:
        return out, inter
    return out


def loss_fn(cfg: OracleConfig, logits: torch.Tensor, t: torch.Tensor, start: Optional[int] = None):
    """wavenet_vocoder/bin/train.py:461,534-536: CrossEntropyLoss(mean) over
    out[:, rf:].contiguous().view(-1, Q) vs t[:, rf:].contiguous().view(-1)."""
    rf = cfg.receptive_field if start is None else start
    return F.cross_entropy(logits[:, rf:].contiguous().view(-1, cfg.n_quantize),
                           t[:, rf:].contiguous().view(-1))


# --------------------------------------------------------------------------
# Adam  (torch.optim.Adam as configured at train.py:457-460)
# --------------------------------------------------------------------------
class OracleAdam(object):
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=wd) restated: L2-in-grad
    weight decay, bias-corrected, params whose grad is None are skipped (train.py:537-539 with
    the dead last res_1x1)."""

    def __init__(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.state = {}

    def step(self, params: Dict[str, torch.Tensor], grads: Dict[str, Optional[torch.Tensor]]):
        b1, b2 = self.betas
        for k, p in params.items():
            g = grads.get(k)
            if g is None:
                continue
            st = self.state.setdefault(k, {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
            st["step"] += 1
            t = st["step"]
            if self.wd != 0:
                g = g + self.wd * p
            st["m"].mul_(b1).add_(g, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1 = 1 - b1 ** t
            bc2 = 1 - b2 ** t
            denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(st["m"], denom, value=-(self.lr / bc1))


# --------------------------------------------------------------------------
# one training step  (train.py:527-540)
# --------------------------------------------------------------------------
def train_step(cfg: OracleConfig, params: Dict[str, torch.Tensor], opt: Optional[OracleAdam],
               x: torch.Tensor, h: torch.Tensor, t: torch.Tensor, loss_start: Optional[int] = None, relu_masks=None,
               return_intermediates: bool = False):
    """forward -> CE on [:, rf:] -> backward -> Adam.  Returns (loss, logits, grads[, intermediates]).

    grads[k] is None for parameters that never receive a gradient (res_1x1.{L-1}.*, because the
    last layer's residual output is dead: wavenet.py:231-238)."""
    leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in params.items())
    res = forward(cfg, leaves, x, h, return_intermediates=return_intermediates, relu_masks=relu_masks)
    logits, inter = res if return_intermediates else (res, None)
    loss = loss_fn(cfg, logits, t, loss_start)
    gl = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    grads = OrderedDict((k, g) for k, g in zip(leaves.keys(), gl))
    if opt is not None:
        with torch.no_grad():
            opt.step(params, grads)
    if return_intermediates:
        inter = {k: ([u.detach() for u in v] if isinstance(v, list) else v.detach()) for k, v in inter.items()}
        return loss.detach(), logits.detach(), grads, inter
    return loss.detach(), logits.detach(), grads


# --------------------------------------------------------------------------
# synthetic inputs: x,t ~ randint(0,Q); h ~ N(0,1)
# --------------------------------------------------------------------------
def synthetic_batch(cfg: OracleConfig, B: int, T: int, seed: int, dtype=torch.float32):
    """numpy RandomState stream (machine independent).  T counts model inputs (x_[:-1]);
    targets are the next sample (x_[1:]), train.py:223-224."""
    rs = np.random.RandomState(seed)
    xx = rs.randint(0, cfg.n_quantize, size=(B, T + 1)).astype(np.int64)
    U = cfg.upsampling_factor
    if U > 0:
        assert T % U == 0, "T must be a multiple of the upsampling factor"
        hh = rs.standard_normal(size=(B, cfg.n_aux, T // U))
    else:
        hh = rs.standard_normal(size=(B, cfg.n_aux, T))
    x = torch.from_numpy(xx[:, :-1].copy())
    t = torch.from_numpy(xx[:, 1:].copy())
    h = torch.from_numpy(hh).to(dtype)
    return x, h, t


def relu_kink_margin(cfg: OracleConfig, params, x, h, t_start: Optional[int] = None) -> float:
    """min |pre-ReLU value| over the positions that carry loss (t >= t_start, default rf).

    The two ReLUs of _postprocess (wavenet.py:519,521) make the loss piecewise smooth: the gradient
    is discontinuous where a pre-activation crosses 0.  Two fp32 evaluations that differ by
    round-off (1e-6) can legitimately pick different sub-gradients for an element closer to 0 than
    that, which moves whole gradient rows by O(1/positions).  Gradient parity is therefore only
    defined for instances whose margin is well above round-off; tests select such instances."""
    rf = cfg.receptive_field if t_start is None else t_start
    with torch.no_grad():
        _, inter = forward(cfg, params, x, h, return_intermediates=True)
    m1 = float(inter["skip_sum"][:, :, rf:].abs().min())
    m2 = float(inter["post1_pre"][:, :, rf:].abs().min())
    return min(m1, m2)


def batch_geometry(receptive_field: int, batch_length: int, upsampling_factor: int):
    """train.py:106-110,202-232: batch_length is decreased by (rf+bl) % U; one window holds
    h_bs=(rf+bl)//U frames and x_bs=h_bs*U+1 samples -> T=h_bs*U model inputs."""
    bl = batch_length - (receptive_field + batch_length) % upsampling_factor
    h_bs = (receptive_field + bl) // upsampling_factor
    T = h_bs * upsampling_factor
    return {"batch_length": bl, "frames": h_bs, "T": T, "loss_positions": T - receptive_field}


# --------------------------------------------------------------------------
# generation  (wavenet.py:243-511, 538-549) -- BASELINE config 5
# --------------------------------------------------------------------------
def _pad_context(cfg: OracleConfig, p, x, h):
    """Shared prologue of generate / fast_generate / batch_fast_generate (wavenet.py:258-266,
    328-336, 417-425): up-sample h, then left-pad x with n_quantize//2 and h by replication up to
    the receptive field."""
    if cfg.upsampling_factor > 0:
        h = upsampling(h, p["upsampling.conv.weight"], p["upsampling.conv.bias"])
    n_pad = cfg.receptive_field - x.size(1)
    if n_pad > 0:
        x = F.pad(x, (n_pad, 0), "constant", cfg.n_quantize // 2)
        h = F.pad(h, (n_pad, 0), "replicate")
    return x, h


def _pick(logits_row, mode, generator=None):
    """wavenet.py:289-298 / 370-378: categorical sample of softmax, or argmax."""
    if mode == "argmax":
        return int(logits_row.argmax())
    if mode == "sampling":
        post = F.softmax(logits_row, dim=0)
        return int(torch.multinomial(post, 1, generator=generator))
    raise ValueError("mode should be sampling or argmax")


def generate(cfg: OracleConfig, p, x, h, n_samples, mode="argmax", return_logits=False):
    """WaveNet.generate, wavenet.py:243-307: every new sample is the last output of a full forward
    over the last receptive_field samples.  x (1,T) int64, h (1,A,(n_samples+T)[/U])."""
    x, h = _pad_context(cfg, p, x, h)
    rf = cfg.receptive_field
    samples = x[0].tolist()
    rows = []
    sub = OracleConfig(*cfg.as_tuple()[:7], 0)  # h is at sample rate from here on
    for _ in range(n_samples):
        n = len(samples)
        xw = torch.tensor(samples[-rf:]).long().view(1, -1)
        hw = h[:, :, n - rf:n]
        row = forward(sub, p, xw, hw)[0, -1]
        rows.append(row)
        samples.append(_pick(row, mode))
    out = np.array(samples[-n_samples:])
    return (out, torch.stack(rows)) if return_logits else out


def _one_step(cfg: OracleConfig, p, prev_tokens, h_col, queues):
    """One autoregressive step of the queue algorithm (wavenet.py:350-366, 538-549).

    prev_tokens: (B, >=K) last tokens (the newest last); h_col: (B,A,1) aux at this position;
    queues[l]: (B,R,(K-1)*d_l) = the input history of layer l (newest last).  Returns the logits
    (B,Q) and the updated queues.  The reference keeps the *outputs* of layer l with length
    (K-1)*d_{l+1} (buffer_size, wavenet.py:346-349); that is the same data indexed from the
    consumer's side."""
    K = cfg.kernel_size
    dtype = p["causal.conv.weight"].dtype
    # _preprocess on the last 2K-1 tokens, keep the newest column (wavenet.py:355-356)
    win = prev_tokens[:, -(2 * K - 1):]
    out = causal_conv1d(onehot(win, cfg.n_quantize, dtype).transpose(1, 2),
                        p["causal.conv.weight"], p["causal.conv.bias"], 1)[:, :, -1:]
    skips = None
    new_queues = []
    for l, d in enumerate(cfg.dilations):
        hist = torch.cat([queues[l], out], dim=2)                # (B,R,(K-1)d+1)
        new_queues.append(hist[:, :, 1:] if K > 1 else hist[:, :, :0])
        # _generate_residual_forward (wavenet.py:538-549): dilated conv, newest column only
        o_s = causal_conv1d(hist, p["dil_sigmoid.%d.conv.weight" % l], p["dil_sigmoid.%d.conv.bias" % l], d)[:, :, -1:]
        o_t = causal_conv1d(hist, p["dil_tanh.%d.conv.weight" % l], p["dil_tanh.%d.conv.bias" % l], d)[:, :, -1:]
        a_s = F.conv1d(h_col, p["aux_1x1_sigmoid.%d.weight" % l], p["aux_1x1_sigmoid.%d.bias" % l])
        a_t = F.conv1d(h_col, p["aux_1x1_tanh.%d.weight" % l], p["aux_1x1_tanh.%d.bias" % l])
        z = torch.sigmoid(o_s + a_s) * torch.tanh(o_t + a_t)
        skip = F.conv1d(z, p["skip_1x1.%d.weight" % l], p["skip_1x1.%d.bias" % l])
        out = F.conv1d(z, p["res_1x1.%d.weight" % l], p["res_1x1.%d.bias" % l]) + out
        skips = skip if skips is None else skips + skip
    o = F.relu(skips)
    o = F.relu(F.conv1d(o, p["conv_post_1.weight"], p["conv_post_1.bias"]))
    o = F.conv1d(o, p["conv_post_2.weight"], p["conv_post_2.bias"])
    return o[:, :, 0], new_queues


def _prefill_queues(cfg: OracleConfig, p, x, h):
    """'prepare buffer' (wavenet.py:338-349): a full forward over the context; queue l keeps the
    inputs of layer l at the last (K-1)*d_l positions BEFORE the newest one."""
    dtype = p["causal.conv.weight"].dtype
    K = cfg.kernel_size
    out = causal_conv1d(onehot(x, cfg.n_quantize, dtype).transpose(1, 2),
                        p["causal.conv.weight"], p["causal.conv.bias"], 1)
    hh = h[:, :, :x.size(1)]
    queues = []
    for l, d in enumerate(cfg.dilations):
        n = (K - 1) * d
        q = out[:, :, -n - 1:-1]
        if q.size(2) < n:   # context shorter than this queue: zero history (the conv's own padding)
            q = F.pad(q, (n - q.size(2), 0))
        queues.append(q)
        out, _ = residual_forward(out, hh, p, l, d)
    return queues


def batch_fast_generate(cfg: OracleConfig, p, x, h, n_samples_list, mode="argmax", return_logits=False):
    """WaveNet.batch_fast_generate, wavenet.py:397-511.  x (B,T), h (B,A,(max_n+T)[/U]).  Returns
    the list of generated token arrays in the reference's order (shortest first, ties by index)."""
    n_samples_list = list(n_samples_list)
    x, h = _pad_context(cfg, p, x, h)
    queues = _prefill_queues(cfg, p, x, h)
    samples = x
    B = x.size(0)
    alive = list(range(B))
    done = {}
    rows = [[] for _ in range(B)]
    for i in range(max(n_samples_list)):
        h_col = h[alive, :, samples.size(1) - 1].unsqueeze(-1)
        logits, queues = _one_step(cfg, p, samples, h_col, queues)
        new = torch.tensor([_pick(logits[j], mode) for j in range(len(alive))]).long().view(-1, 1)
        for j, b in enumerate(alive):
            rows[b].append(logits[j])
        samples = torch.cat([samples, new], dim=1)
        keep = [j for j, b in enumerate(alive) if n_samples_list[b] > i + 1]
        for j, b in enumerate(alive):
            if n_samples_list[b] == i + 1:
                done[b] = samples[j, -n_samples_list[b]:].numpy().copy()
        if len(keep) != len(alive):
            samples = samples[keep]
            queues = [q[keep] for q in queues]
            alive = [alive[j] for j in keep]
        if not alive:
            break
    order = sorted(range(B), key=lambda b: (n_samples_list[b], b))
    toks = [done[b] for b in order]
    if return_logits:
        return toks, [torch.stack(rows[b]) for b in order]
    return toks


def fast_generate(cfg: OracleConfig, p, x, h, n_samples, mode="argmax", return_logits=False):
    """WaveNet.fast_generate, wavenet.py:309-395 (the B=1 case of the queue algorithm)."""
    r = batch_fast_generate(cfg, p, x, h, [n_samples], mode, return_logits)
    return (r[0][0], r[1][0]) if return_logits else r[0]


# --------------------------------------------------------------------------
# mixture-of-logistics output head (BASELINE configs[3])
#
# NOT in the reference (its WaveNet has only the softmax head, wavenet.py:209-210,518-523): there is
# no reference code to restate and no reference output to pin this against -- PARITY UNPINNED BY THE
# REFERENCE.  What follows is the discretised mixture of logistics of PixelCNN++ (Salimans et al.,
# 2017, eq. 2-3 and its edge cases) as WaveNet vocoders use it for 16-bit audio, written from the
# published formulas; it is the checker of csrc/wn_elem.hip::k_mol_nll and of the decode sampler.
# --------------------------------------------------------------------------
def mol_nll(out, y, num_classes=65536, log_scale_min=-7.0, start=0):
    """out (B,T,3*nm): [logits | means | log-scales]; y (B,T) in [-1,1].  Mean negative
    log-likelihood over positions t >= start."""
    nm = out.size(-1) // 3
    out = out[:, start:]
    y = y[:, start:].unsqueeze(-1)
    logit, mean = out[..., :nm], out[..., nm:2 * nm]
    ls = torch.clamp(out[..., 2 * nm:3 * nm], min=log_scale_min)
    half = 1.0 / (num_classes - 1)
    inv = torch.exp(-ls)
    c = y - mean
    plus_in, min_in, mid_in = inv * (c + half), inv * (c - half), inv * c
    cdf_delta = torch.sigmoid(plus_in) - torch.sigmoid(min_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)               # log P(x < first bin edge)
    log_one_minus_cdf_min = -F.softplus(min_in)                # log P(x > last bin edge)
    log_pdf_mid = mid_in - ls - 2.0 * F.softplus(mid_in) - float(np.log((num_classes - 1) / 2.0))
    inner = torch.where(cdf_delta > 1e-5, torch.log(torch.clamp(cdf_delta, min=1e-12)), log_pdf_mid)
    ll = torch.where(y < -0.999, log_cdf_plus, torch.where(y > 0.999, log_one_minus_cdf_min, inner))
    lp = ll + F.log_softmax(logit, dim=-1)
    return -torch.logsumexp(lp, dim=-1).mean()


def mol_sample(out_row, u, log_scale_min=-7.0):
    """One draw from the mixture of a single position: out_row (3*nm,), u (nm+1,) uniforms in (0,1):
    component = argmax(logit - log(-log u_i)) (Gumbel max), x = mean + scale*(log u - log(1-u)), clipped."""
    nm = out_row.numel() // 3
    g = out_row[:nm] - torch.log(-torch.log(u[:nm]))
    k = int(g.argmax())
    mean = out_row[nm + k]
    ls = torch.clamp(out_row[2 * nm + k], min=log_scale_min)
    uu = u[nm]
    x = mean + torch.exp(ls) * (torch.log(uu) - torch.log(1.0 - uu))
    return float(torch.clamp(x, -1.0, 1.0))
