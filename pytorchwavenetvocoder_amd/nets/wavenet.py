# -*- coding: utf-8 -*-
"""MI355X-native drop-in for ``wavenet_vocoder.nets.wavenet`` (reference wavenet.py).

Same public surface as the reference module -- ``encode_mu_law``, ``decode_mu_law``,
``initialize``, ``OneHot``, ``CausalConv1d``, ``UpSampling``, ``WaveNet`` with the reference's
constructor, ``forward(x, h)`` contract and ``state_dict`` keys/shapes -- but ``WaveNet.forward``
and its backward run in the hand-written HIP kernels of ``libwavenet_hip.so`` (C ABI:
include/wavenet_hip.h).  The ``nn.Conv1d`` / ``nn.ConvTranspose2d`` sub-modules are parameter
containers only (so ``model.apply(initialize)``, ``load_state_dict`` and checkpoints behave like
the reference's, wavenet.py:57-63, train.py:325-331): their tensors are views into ONE flat fp32
buffer that the kernels, the fused Adam and the RCCL all-reduce operate on.

There is no CPU path: calling the model with CPU tensors raises.
"""
import logging
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import _lib
from ..engine import WaveNetEngine, _stream_handle, key_to_kind


def encode_mu_law(x, mu=256):
    """Mu-law encoding (reference wavenet.py:17-30).

    Args:
        x (ndarray): Audio signal with the range from -1 to 1.
        mu (int): Quantized level.

    Returns:
        ndarray: Quantized audio signal with the range from 0 to mu - 1.
    """
    mu = mu - 1
    fx = np.sign(x) * np.log(1 + mu * np.abs(x)) / np.log(1 + mu)
    return np.floor((fx + 1) / 2 * mu + 0.5).astype(np.int64)


def decode_mu_law(y, mu=256):
    """Mu-law decoding (reference wavenet.py:33-47)."""
    mu = mu - 1
    fx = (y - 0.5) / mu * 2 - 1
    x = np.sign(fx) / mu * ((1 + mu) ** np.abs(fx) - 1)
    return x


def initialize(m):
    """Xavier init for Conv1d, ones/zeros for ConvTranspose2d (reference wavenet.py:50-63).

    In-place, so it writes straight through the parameter views into the flat buffer.
    """
    if isinstance(m, nn.Conv1d):
        nn.init.xavier_uniform_(m.weight)
        nn.init.constant_(m.bias, 0.0)
    if isinstance(m, nn.ConvTranspose2d):
        nn.init.constant_(m.weight, 1.0)
        nn.init.constant_(m.bias, 0.0)


class OneHot(nn.Module):
    """One-hot conversion (reference wavenet.py:66-92).  Kept for API compatibility; the WaveNet
    forward never materialises it (the front convolution is a gather, csrc/wn_elem.hip)."""

    def __init__(self, depth):
        super(OneHot, self).__init__()
        self.depth = depth

    def forward(self, x):
        x = x % self.depth
        x = torch.unsqueeze(x, 2)
        x_onehot = x.new_zeros(x.size(0), x.size(1), self.depth).float()
        return x_onehot.scatter_(2, x, 1)


class CausalConv1d(nn.Module):
    """1D dilated causal convolution (reference wavenet.py:95-121).

    Stand-alone ``forward`` runs the HIP causal-conv op (taps as K-segments of one f32-MFMA
    contraction); inside ``WaveNet`` the module only holds the parameters.
    """

    def __init__(self, in_channels, out_channels, kernel_size, dilation=1, bias=True):
        super(CausalConv1d, self).__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.dilation = dilation
        self.padding = (kernel_size - 1) * dilation
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size,
                              padding=self.padding, dilation=dilation, bias=bias)

    def forward(self, x):
        """x (B, C, T) float -> (B, C', T) on the HIP op ``wn_op_causal_conv``.  Differentiable like the reference's module
        (wavenet.py:95-121 is an ordinary ``nn.Module``): when a gradient is asked for -- x or the conv parameters require grad
        under an enabled grad mode -- the call goes through ``_CausalConv1dFunction``, whose backward is the HIP op
        ``wn_op_causal_conv_backward`` (dx with the taps transposed, dW / db as fixed-order sums over time)."""
        if not x.is_cuda:
            raise _lib.WnError("CausalConv1d runs on the GPU HIP path only (no CPU fallback)")
        w, b = self.conv.weight, self.conv.bias
        if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad)):
            return _CausalConv1dFunction.apply(x, w, b, self.dilation)
        return _causal_conv_hip(x, w.detach(), None if b is None else b.detach(), self.dilation)


def _causal_conv_hip(x, w, b, dilation):
    import ctypes
    lib = _lib.load_library()
    if x.dtype != torch.float32 or w.dtype != torch.float32:
        raise _lib.WnError("CausalConv1d on the HIP path is fp32 (got %s / %s)" % (x.dtype, w.dtype))
    x = x.contiguous()
    B, C, T = x.shape
    Cout, Cin, K = w.shape
    if C != Cin:
        raise ValueError("expected %d input channels, got %d" % (Cin, C))
    w = w.contiguous()
    b = b.contiguous() if b is not None else torch.zeros(Cout, device=x.device)
    y = torch.empty(B, Cout, T, device=x.device, dtype=torch.float32)
    scratch = torch.empty(w.numel(), device=x.device, dtype=torch.float32)
    st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    rc = lib.wn_op_causal_conv(w.data_ptr(), b.data_ptr(), x.data_ptr(), y.data_ptr(), scratch.data_ptr(),
                               B, T, Cin, Cout, K, int(dilation), st)
    lib.check(rc, "wn_op_causal_conv")
    return y


class _CausalConv1dFunction(torch.autograd.Function):
    """``CausalConv1d.forward`` with a gradient: forward = wn_op_causal_conv, backward = wn_op_causal_conv_backward (autograd of
    the reference's Conv1d + slice, wavenet.py:105-121)."""

    @staticmethod
    def forward(ctx, x, w, b, dilation):
        ctx.save_for_backward(x, w)
        ctx.dilation, ctx.has_bias = int(dilation), b is not None
        return _causal_conv_hip(x.detach(), w.detach(), None if b is None else b.detach(), dilation)

    @staticmethod
    def backward(ctx, dy):
        import ctypes
        x, w = ctx.saved_tensors
        lib = _lib.load_library()
        x = x.detach().contiguous()
        w = w.detach().contiguous()
        dy = dy.contiguous().float()
        B, Cin, T = x.shape
        Cout, _, K = w.shape
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty_like(w) if need_w else None
        db = torch.empty(Cout, device=x.device, dtype=torch.float32) if need_b else None
        n = lib.wn_op_causal_conv_backward_scratch_floats(B, T, Cin, Cout, K)
        scratch = torch.empty(max(int(n), 1), device=x.device, dtype=torch.float32)
        st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        rc = lib.wn_op_causal_conv_backward(w.data_ptr(), x.data_ptr(), dy.data_ptr(), dx.data_ptr() if need_x else None,
                                            dw.data_ptr() if need_w else None, db.data_ptr() if need_b else None,
                                            scratch.data_ptr(), B, T, Cin, Cout, K, ctx.dilation, st)
        lib.check(rc, "wn_op_causal_conv_backward")
        return dx, dw, db, None


class UpSampling(nn.Module):
    """Upsampling with a (1, U) transposed convolution (reference wavenet.py:124-154).

    ``out[b, c, f*U + j] = h[b, c, f] * w[j] + bias`` -- inside ``WaveNet`` this is never
    materialised (applied at frame rate inside the gate, csrc/wn_elem.hip / wn_fused.hip).
    """

    def __init__(self, upsampling_factor, bias=True):
        super(UpSampling, self).__init__()
        self.upsampling_factor = upsampling_factor
        self.bias = bias
        self.conv = nn.ConvTranspose2d(1, 1, kernel_size=(1, self.upsampling_factor),
                                       stride=(1, self.upsampling_factor), bias=self.bias)

    def forward(self, x):
        """x (B, C, T) -> (B, C, T * upsampling_factor).  On the GPU: the HIP op ``wn_op_upsampling`` (what ``generate()``
        calls); when a gradient is asked for (x or the conv parameters require grad under an enabled grad mode) through
        ``_UpSamplingFunction``, whose backward is the closed form of the transposed convolution -- the stand-alone module stays
        differentiable as in the reference.  On the CPU the closed form below (shape tests of the reference)."""
        if x.is_cuda:
            w, b = self.conv.weight, self.conv.bias
            if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad)):
                return _UpSamplingFunction.apply(x, w, b, self.upsampling_factor)
            return _upsampling_hip(x, w.detach(), None if b is None else b.detach(), self.upsampling_factor)
        w = self.conv.weight.view(-1)
        y = x.unsqueeze(-1) * w
        if self.conv.bias is not None:
            y = y + self.conv.bias
        return y.reshape(x.size(0), x.size(1), -1)


def _upsampling_hip(x, w, b, U):
    import ctypes
    lib = _lib.load_library()
    x = x.contiguous().float()
    B, C, F_ = x.shape
    w = w.reshape(-1).contiguous()
    b = b.contiguous() if b is not None else None
    y = torch.empty(B, C, F_ * U, device=x.device, dtype=torch.float32)
    st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    rc = lib.wn_op_upsampling(w.data_ptr(), b.data_ptr() if b is not None else None, x.data_ptr(), y.data_ptr(), B, C, F_, U, st)
    lib.check(rc, "wn_op_upsampling")
    return y


class _UpSamplingFunction(torch.autograd.Function):
    """``UpSampling.forward`` with a gradient: forward = wn_op_upsampling; y[b, c, f U + j] = x[b, c, f] w[j] + bias, so
    dx = sum_j w[j] dy[.., f, j], dw[j] = sum x dy[.., j], dbias = sum dy (three reductions of (B, C, F, U) views)."""

    @staticmethod
    def forward(ctx, x, w, b, U):
        ctx.save_for_backward(x, w)
        ctx.U, ctx.has_bias = U, b is not None
        return _upsampling_hip(x.detach(), w.detach(), None if b is None else b.detach(), U)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy4 = dy.reshape(x.size(0), x.size(1), x.size(2), ctx.U)
        dx = (dy4 * w.reshape(-1)).sum(-1).to(x.dtype) if ctx.needs_input_grad[0] else None
        dw = (dy4 * x.float().unsqueeze(-1)).sum((0, 1, 2)).reshape(w.shape).to(w.dtype) if ctx.needs_input_grad[1] else None
        db = dy.sum().reshape(1).to(w.dtype) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None


class _WaveNetFunction(torch.autograd.Function):
    """autograd bridge: forward = wn_forward, backward = wn_backward (one flat gradient buffer)."""

    @staticmethod
    def forward(ctx, model, x, h, *params):
        eng = model._engine
        logits = eng.forward(x, h)
        model._fwd_serial += 1
        ctx.model = model
        ctx.serial = model._fwd_serial
        return logits.transpose(1, 2)

    @staticmethod
    def backward(ctx, grad_out):
        model = ctx.model
        if ctx.serial != model._fwd_serial:
            raise _lib.WnError("WaveNet backward after a newer forward: the HIP path keeps the activations "
                               "of the latest forward only")
        eng = model._engine
        if grad_out.is_contiguous() and grad_out.dtype == torch.float32:   # (B, T, Q) -> the kernels' (B, Q, T) on the HIP op
            B, T, Q = grad_out.shape
            dl = torch.empty((B, Q, T), dtype=torch.float32, device=grad_out.device)
            st = _stream_handle(grad_out.device)
            eng.lib.check(eng.lib.wn_op_transpose_last2(grad_out.data_ptr(), dl.data_ptr(), B, T, Q, st), "wn_op_transpose_last2")
        else:   # a view that already is (B, Q, T) underneath (e.g. the transposed logits themselves), or another dtype
            dl = grad_out.transpose(1, 2).contiguous().float()
        flat = eng.backward(dl).clone()
        grads = []
        for (off, n, shape, dead) in model._param_slices:
            grads.append(None if dead else flat[off:off + n].view(shape))
        return (None, None, None) + tuple(grads)


class WaveNet(nn.Module):
    """Conditional WaveNet (reference wavenet.py:157-210) on the MI355X HIP path.

    Args:
        n_quantize (int): Number of quantization.
        n_aux (int): Number of aux feature dimension.
        n_resch (int): Number of filter channels for residual block.
        n_skipch (int): Number of filter channels for skip connection.
        dilation_depth (int): Number of dilation depth (e.g. if set 10, max dilation = 2^(10-1)).
        dilation_repeat (int): Number of dilation repeat.
        kernel_size (int): Filter size of dilated causal convolution.
        upsampling_factor (int): Upsampling factor.
        n_mixture (int): NOT a reference argument.  0 (default) = the reference's n_quantize-way softmax head;
            > 0 = mixture-of-logistics head with that many components (BASELINE configs[3]): ``conv_post_2`` then
            has 3 * n_mixture channels (mixture logits, means, log-scales), the input stays the mu-law one-hot
            front end, the loss is ``mol_loss_and_backward`` and generation draws from the mixture.
        log_scale_min (float): mixture head only -- clamp of the predicted log-scales, used by the loss AND by sampling.
    """

    def __init__(self, n_quantize=256, n_aux=28, n_resch=512, n_skipch=256,
                 dilation_depth=10, dilation_repeat=3, kernel_size=2, upsampling_factor=0, n_mixture=0, _library=None,
                 log_scale_min=-7.0):
        super(WaveNet, self).__init__()
        self.n_mixture = n_mixture
        # mixture head: clamp of the predicted log-scales, ONE value for the likelihood (mol_loss_and_backward) and for
        # sampling (the decode kernels), so that training and generation cannot disagree.  A constructor argument, so it
        # lives in the model configuration (train.py writes it into model.conf, decode.py rebuilds the model from it) and
        # state_dict keeps exactly the reference's parameter keys.
        self.log_scale_min = float(log_scale_min)
        self.out_channels = 3 * n_mixture if n_mixture > 0 else n_quantize
        self.n_aux = n_aux
        self.n_quantize = n_quantize
        self.n_resch = n_resch
        self.n_skipch = n_skipch
        self.kernel_size = kernel_size
        self.dilation_depth = dilation_depth
        self.dilation_repeat = dilation_repeat
        self.upsampling_factor = upsampling_factor

        self.dilations = [2 ** i for i in range(self.dilation_depth)] * self.dilation_repeat
        self.receptive_field = (self.kernel_size - 1) * sum(self.dilations) + 1

        # parameter containers, registered in the reference's order (wavenet.py:187-210)
        self.onehot = OneHot(self.n_quantize)
        self.causal = CausalConv1d(self.n_quantize, self.n_resch, self.kernel_size)
        if self.upsampling_factor > 0:
            self.upsampling = UpSampling(self.upsampling_factor)
        self.dil_sigmoid = nn.ModuleList()
        self.dil_tanh = nn.ModuleList()
        self.aux_1x1_sigmoid = nn.ModuleList()
        self.aux_1x1_tanh = nn.ModuleList()
        self.skip_1x1 = nn.ModuleList()
        self.res_1x1 = nn.ModuleList()
        for d in self.dilations:
            self.dil_sigmoid += [CausalConv1d(self.n_resch, self.n_resch, self.kernel_size, d)]
            self.dil_tanh += [CausalConv1d(self.n_resch, self.n_resch, self.kernel_size, d)]
            self.aux_1x1_sigmoid += [nn.Conv1d(self.n_aux, self.n_resch, 1)]
            self.aux_1x1_tanh += [nn.Conv1d(self.n_aux, self.n_resch, 1)]
            self.skip_1x1 += [nn.Conv1d(self.n_resch, self.n_skipch, 1)]
            self.res_1x1 += [nn.Conv1d(self.n_resch, self.n_resch, 1)]
        self.conv_post_1 = nn.Conv1d(self.n_skipch, self.n_skipch, 1)
        self.conv_post_2 = nn.Conv1d(self.n_skipch, self.out_channels, 1)

        # the HIP engine (loads libwavenet_hip.so; raises when the extension is unavailable)
        object.__setattr__(self, "_engine", WaveNetEngine(
            n_quantize, n_aux, n_resch, n_skipch, dilation_depth, dilation_repeat, kernel_size,
            upsampling_factor, device="cpu", library=_library, out_channels=3 * n_mixture))
        assert self._engine.receptive_field == self.receptive_field
        self._fwd_serial = 0
        self._param_slices = []
        self._flatten()

    # ---- flat storage -------------------------------------------------------------------------
    def _flatten(self):
        """(Re)build the flat parameter buffer on the parameters' device and point every
        nn.Parameter at its slice."""
        eng = self._engine
        named = list(self.named_parameters())
        device = named[0][1].device
        for _, p in named:
            if p.dtype != torch.float32:
                raise _lib.WnError("the HIP WaveNet path is fp32 (parity gate 1e-4); got %s" % p.dtype)
            if p.device != device:
                raise _lib.WnError("all WaveNet parameters must live on one device")
        flat = torch.empty(eng.n_params, dtype=torch.float32, device=device)
        slices = []
        with torch.no_grad():
            for k, p in named:
                kind, layer = key_to_kind(k)
                off, n = eng.param_slice(kind, layer)
                assert n == p.numel(), k
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
                dead = (eng.dead_range[0] <= off < eng.dead_range[1])
                slices.append((off, n, tuple(p.shape), dead))
        eng.device = device
        eng.flat_params = flat
        eng.flat_grads = None
        eng._ws = None
        eng._ws_key = None
        eng._version_sources = tuple(p for _, p in named)
        eng._fwd_version = None
        self._param_slices = slices

    def _apply(self, fn, *args, **kwargs):
        out = super(WaveNet, self)._apply(fn, *args, **kwargs)
        self._flatten()
        return out

    @property
    def engine(self):
        return self._engine

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, x, h):
        """Forward calculation (reference wavenet.py:212-241).

        Args:
            x (Tensor): Long tensor variable with the shape (B, T).
            h (Tensor): Float tensor variable with the shape (B, n_aux, T)
                (or (B, n_aux, T / upsampling_factor) with the upsampling layer).

        Returns:
            Tensor: Float tensor variable with the shape (B, T, n_quantize)
                (a transposed view of a (B, n_quantize, T) buffer, like the reference's result).
        """
        params = tuple(self.parameters())
        return _WaveNetFunction.apply(self, x, h, *params)

    def loss_and_backward(self, x, h, t, t_start=None, grad_scale=1.0, events=None, layers_per_bucket=0):
        """Fused training half-step: forward -> CrossEntropy on ``[:, t_start:]`` -> backward
        (reference train.py:533-538) without an autograd graph.

        Leaves the gradients in the flat buffer (``p.grad`` of every parameter is a view of it;
        ``None`` for the dead last ``res_1x1``) and returns the mean loss as a 1-element device
        tensor (no host sync).  ``grad_scale`` multiplies the gradients (1/world_size for DP)."""
        eng = self._engine
        if t_start is None:
            t_start = eng.receptive_field
        # forward + loss in one call: the cross-entropy is the epilogue of conv_post_2 where the model allows it (the logits
        # never reach memory), and the backward pass runs its post-net part over the loss window only
        loss, dlogits = eng.forward_loss(x, h, t, t_start=t_start, grad_scale=grad_scale)
        self._fwd_serial += 1
        flat = eng.backward(dlogits, events=events, layers_per_bucket=layers_per_bucket, t_first=t_start)
        for p, (off, n, shape, dead) in zip(self.parameters(), self._param_slices):
            p.grad = None if dead else flat[off:off + n].view(shape)
        return loss

    def mol_loss_and_backward(self, x, h, y, t_start=None, grad_scale=1.0, num_classes=65536, log_scale_min=None,
                              events=None, layers_per_bucket=0):
        """Training half-step of the mixture-of-logistics head (``n_mixture > 0``): forward -> mean negative
        log-likelihood of the waveform ``y`` (B, T) in [-1, 1] (the value of the NEXT sample at every position,
        like ``t`` of the softmax head) on ``[:, t_start:]`` -> backward.  Gradients land as in
        ``loss_and_backward``."""
        if self.n_mixture <= 0:
            raise ValueError("this model has the softmax head (n_mixture = 0)")
        # ONE clamp for the likelihood and for sampling: the constructor's value (it travels in model.conf; a per-call
        # value would be lost with the checkpoint and generation would clamp differently from training)
        if log_scale_min is not None and float(log_scale_min) != self.log_scale_min:
            raise ValueError("log_scale_min=%g differs from the model's %g: pass it to the constructor "
                             "(train.py --log_scale_min), it is part of the model configuration"
                             % (float(log_scale_min), self.log_scale_min))
        log_scale_min = self.log_scale_min
        eng = self._engine
        out = eng.forward(x, h)
        self._fwd_serial += 1
        if t_start is None:
            t_start = eng.receptive_field
        loss, dout = eng.mol_loss(out, y, t_start=t_start, grad_scale=grad_scale, num_classes=num_classes,
                                  log_scale_min=log_scale_min)
        flat = eng.backward(dout, events=events, layers_per_bucket=layers_per_bucket, t_first=t_start)
        for p, (off, n, shape, dead) in zip(self.parameters(), self._param_slices):
            p.grad = None if dead else flat[off:off + n].view(shape)
        return loss

    # ---- generation (reference wavenet.py:243-511) --------------------------------------------
    def _window_logits(self, x, h_up):
        """Logits (T, Q) for ONE window with the aux features already at sample rate."""
        eng = self._gen_engine()
        logits = eng.forward(x, h_up)
        return logits[0].transpose(0, 1)

    def _gen_engine(self):
        """Engine that treats h as already up-sampled (the reference bypasses the upsampling layer
        inside generate(), wavenet.py:258-259,273-283).  The flat layout keeps the upsampling
        tensors at the very end, so the U=0 layout is a prefix of this model's buffer."""
        if self.upsampling_factor == 0:
            return self._engine
        eng = getattr(self, "_gen_eng", None)
        if eng is None or eng.device != self._engine.device or \
                eng.flat_params.data_ptr() != self._engine.flat_params.data_ptr():
            eng = WaveNetEngine(self.n_quantize, self.n_aux, self.n_resch, self.n_skipch, self.dilation_depth,
                                self.dilation_repeat, self.kernel_size, 0, device=self._engine.device,
                                library=self._engine.lib, out_channels=3 * self.n_mixture)
            eng.flat_params = self._engine.flat_params[:eng.n_params]
            object.__setattr__(self, "_gen_eng", eng)
        return eng

    def generate(self, x, h, n_samples, intervals=None, mode="sampling"):
        """Naive generation (reference wavenet.py:243-307): every new sample is the last output of
        a full forward over the newest ``receptive_field`` samples (activations left of the window
        count as zero, which is what distinguishes it from ``fast_generate``).  Every window runs
        through the HIP forward; the token buffer stays on the device.

        Args:
            x (Tensor): Long tensor variable with the shape (1, T).
            h (Tensor): Float tensor variable with the shape (1, n_aux, n_samples + T).
            n_samples (int): Number of samples to be generated.
            intervals (int): Log interval.
            mode (str): "sampling" or "argmax".

        Returns:
            ndarray: Generated quantized waveform (n_samples,).
        """
        if mode not in ("sampling", "argmax"):
            logging.error("mode should be sampling or argmax")
            sys.exit(1)
        if self.n_mixture > 0:
            # the 3 * n_mixture outputs are mixture parameters, not class logits: a softmax / argmax over them would
            # return meaningless tokens.  The queue-based generators draw from the mixture on the device.
            raise ValueError("generate() is the reference's softmax-head generator; a model with the mixture-of-logistics "
                             "head (n_mixture = %d) generates with fast_generate / batch_fast_generate" % self.n_mixture)
        rf = self.receptive_field
        with torch.no_grad():
            if self.upsampling_factor > 0:
                h = self.upsampling(h)
            n_pad = max(rf - x.size(1), 0)
            n_ctx = x.size(1) + n_pad
            tokens = torch.full((1, n_ctx + n_samples), self.n_quantize // 2, dtype=torch.long, device=h.device)
            tokens[:, n_pad:n_ctx] = x
            if n_pad > 0:
                h = F.pad(h, (n_pad, 0), "replicate")
            tick, done_at_tick = time.time(), 0
            for i in range(n_samples):
                end = n_ctx + i
                logits = self._window_logits(tokens[:, end - rf:end].contiguous(), h[:, :, end - rf:end].contiguous())[-1]
                if mode == "argmax":
                    tokens[0, end] = logits.argmax()
                else:
                    tokens[0, end] = torch.distributions.Categorical(F.softmax(logits, dim=0)).sample()
                if intervals is not None and (i + 1) % intervals == 0:
                    per = (time.time() - tick) / (i + 1 - done_at_tick)
                    logging.info("%d/%d estimated time = %.3f sec (%.3f sec / sample)" % (
                        i + 1, n_samples, (n_samples - i - 1) * per, per))
                    tick, done_at_tick = time.time(), i + 1
            return tokens[0, n_ctx:].cpu().numpy()

    def _decode(self, x, h, n_samples_list, intervals, mode):
        """Run the HIP decode kernel (csrc/wn_decode.hip); returns per-utterance LongTensors."""
        if mode not in ("sampling", "argmax"):
            logging.error("mode should be sampling or argmax")
            sys.exit(1)
        if self.n_mixture > 0:  # the mixture head has no argmax: both modes draw from the mixture
            mode = "mol"
        start = [time.time(), 0]

        def progress(done, total):
            if intervals is not None and done > start[1]:
                dt = (time.time() - start[0]) / (done - start[1])
                logging.info("%d/%d estimated time = %.3f sec (%.3f sec / sample)" % (
                    done, total, (total - done) * dt, dt))
                start[0], start[1] = time.time(), done

        with torch.no_grad():
            return self.engine.decode(x, h, list(n_samples_list), mode=mode,
                                      chunk=intervals if intervals else 4096, progress=progress,
                                      log_scale_min=self.log_scale_min)

    def fast_generate(self, x, h, n_samples, intervals=None, mode="sampling"):
        """Generate a waveform with the queue algorithm (reference wavenet.py:309-395) on the HIP
        decode path: one persistent workgroup per utterance for the model sizes the decode kernel
        covers, layer-wise launches on [channels x batch] operands for any other size; the context
        walk fills the dilation queues, then ``n_samples`` tokens are emitted; nothing but the result
        leaves the GPU.

        Args:
            x (tensor): Long tensor variable with the shape  (1, T).
            h (tensor): Float tensor variable with the shape  (1, n_aux, n_samples + T)
                (frames if the model up-samples).
            n_samples (int): Number of samples to be generated.
            intervals (int): Log interval.
            mode (str): "sampling" or "argmax".

        Returns:
            ndarray: Generated quantized waveform (n_samples,).
        """
        return self._decode(x, h, [n_samples], intervals, mode)[0].cpu().numpy()

    def batch_fast_generate(self, x, h, n_samples_list, intervals=None, mode="sampling"):
        """Batched queue-algorithm generation (reference wavenet.py:397-511): one workgroup per
        utterance, each stops at its own length.  Returns the list of generated waveforms in order
        of completion (shortest first), like the reference.

        Args:
            x (tensor): Long tensor variable with the shape (B, T).
            h (tensor): Float tensor variable with the shape (B, n_aux, max(n_samples_list) + T).
            n_samples_list (list): List of number of samples to be generated (B,).
            intervals (int): Log interval.
            mode (str): "sampling" or "argmax".
        """
        order = sorted(range(len(n_samples_list)), key=lambda i: (n_samples_list[i], i))
        toks = self._decode(x, h, n_samples_list, intervals, mode)
        return [toks[i].cpu().numpy() for i in order]
