# -*- coding: utf-8 -*-
"""Device-side state of one WaveNet replica: flat parameter / gradient / Adam buffers and the
workspace the HIP library needs, plus thin wrappers around the C-ABI calls.

PyTorch is used here only for device memory, streams and (in ``DataParallelReducer``)
``torch.distributed``; every FLOP of the path runs in libwavenet_hip.so.
"""
import ctypes
import os
import warnings
import weakref

import torch

from . import _lib

# Launch modes every engine starts with: everything on the caller's stream, and the aux-path gradients as partial sums
# inside the gate kernel (WN_FLAG_AUX_FUSED: -3.4 % step time, profiles/r01/aux_fused_probe.txt; it applies to the fused
# split kernels with U % 16 == 0 and is ignored elsewhere).  The opt-in overlap modes (_lib.FLAG_BWD_OVERLAP /
# FLAG_FWD_OVERLAP) measured no faster on MI355X (profiles/r01/overlap_probe.txt).
# Round 5: WN_FLAG_DW_3PRODUCT (opt-in, NOT in the defaults): the weight-gradient contractions -- LEAF results, sums over every
# position of the minibatch that nothing else consumes -- with three of the six products of the operand split.  It meets the
# gradient gates with room (worst gradient tensor against the oracle at the three timed sizes 6.3e-6 / 6.3e-6 / 1.5e-5 of its
# maximum; six products 4.9e-6 / 6.3e-6 / 1.5e-5) and buys 3 - 4 % (headline step 9.84 -> 9.46 ms, recipe size 126 -> 117 ms, same
# box, profiles/r05/), but NOT the golden after-Adam gate (weights after one Adam step within 1e-2 lr of the reference's): on the
# small golden cases elements whose gradient is ~1e-8 -- where Adam's update lr g / (|g| + eps) is sign-like -- move by up to
# 2e-2 lr (tests/test_gpu_parity.py, r64_k2_up).  Parity first: every default-mode contraction stays fp32-equivalent; the flag is
# for callers who accept that (engine.flags |= _lib.FLAG_DW_3PRODUCT, or WN_ENGINE_FLAGS=262176), bench.py reports its step time
# beside the headline (extras.dw_3product), tests/test_gpu_fullsize.py keeps its 3e-5 gradient gate.
# WN_FLAG_DW_F16PAIR (round 5, DEFAULT): the same contractions with TWO FP16 pieces per operand (11 + 11 significand bits) and the
# three products h h + h l + l h on v_mfma_f32_32x32x16_f16 -- 2^-22 per product instead of the 2^-16 of two bf16 pieces, below
# the rounding of an fp32 running sum over a minibatch's positions -- at the three-bf16-product mode's speed (headline step 9.54 ->
# 9.08 ms, recipe size 119.0 -> 104.9 ms, configs[3] geometry 12.70 -> 12.17 ms, same box, profiles/r05/abk_f16pair_headline.txt,
# dw3_probe_f16pair.txt).  It meets EVERY gate of the six-product mode: golden gradients and the golden weights after Adam
# (tests/test_gpu_dw_f16pair.py), the full-size gradient gates with the six-product mode's own worst tensor (4.87e-6 / 1.49e-5,
# tests/test_gpu_fullsize.py).  fp16's 5 exponent bits need the size of the gradient: backward() passes the bound of the
# tensor its own loss call returned (a mean cross-entropy: |dlogits| <= grad_scale / positions), the kernels scale the gradient
# operand by 2^(e + 8), and a gradient that still leaves fp16's range is DETECTED (non-finite block result) and redone by the
# six-product launch issued behind every fp16 launch (an empty launch otherwise: 0.04 ms per step).  Any other gradient (autograd's
# grad_output, the mixture-of-logistics head) keeps the six bf16 products unless the caller gives backward(dlogits_bound=...).
# WN_FLAG_MM_F16PAIR (round 6, DEFAULT): the weights x activations contractions on the split matrix-core kernel k_gemm6 -- skip sum,
# post-net with the cross-entropy epilogue, their data gradients, the all-layer skip gradient; the per-layer contractions of wide
# models -- take the same fp16 pair split (three products, ~2^-22 each: the rounding of an fp32 running sum over >= 64 terms), each
# launch followed by a conditional six-product redo (an operand outside fp16's range raises a workspace word).  Same box: headline
# step 9.21 -> 8.72 ms (fwd_skip_sum 0.68 -> 0.43, bwd_dz_skip_all 0.80 -> 0.62; profiles/r06/abk_mm_f16.txt).  Gates on the
# benchmark's own instance against the reference module (tests/test_gpu_fullsize.py, bench.py `parity`): logits 4.9e-6 (six
# products 5.5e-6), worst gradient 8.3e-6 (8.2e-6); against the fp64 evaluation of the same step the worst gradient is 6.2e-6 --
# closer than the reference's own fp32 step (8.3e-6; profiles/r06/adam_gate_study.txt).
# WN_FLAG_FUSED_F16PAIR (round 6, DEFAULT): the fused 64-channel FORWARD block (taps, gate, res 1x1: k_resblock_fwd_h) on the same
# split, block-scaled -- every weight image and every 64 x 32 operand tile by the power of two that puts its maximum at 2^12 /
# 2^14, so no magnitude leaves fp16's range and no redo exists.  Same box: 8.64 -> 8.43 ms per step (forward blocks 1.79 -> 1.55 ms,
# profiles/r06/abk_fused_f16.txt).  The backward data chain (k_chain64s) keeps six bf16 products.
DEFAULT_FLAGS = _lib.FLAG_AUX_FUSED | _lib.FLAG_DW_F16PAIR | _lib.FLAG_MM_F16PAIR | _lib.FLAG_FUSED_F16PAIR
SIX_PRODUCT_FLAGS = DEFAULT_FLAGS & ~_lib.NARROW_FLAGS   # every contraction fp32-equivalent (six bf16 products)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream_handle(device):
    if device.type == "cuda":
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    return None


class WaveNetEngine(object):
    """Owns the flat buffers of one model replica.

    Args mirror ``WaveNet.__init__`` (reference wavenet.py:172-173).
    """

    def __init__(self, n_quantize=256, n_aux=28, n_resch=512, n_skipch=256, dilation_depth=10,
                 dilation_repeat=3, kernel_size=2, upsampling_factor=0, device="cpu", library=None, out_channels=0):
        self.lib = library if library is not None else _lib.load_library()
        self.cfg = _lib.WnConfig(n_quantize, n_aux, n_resch, n_skipch, dilation_depth, dilation_repeat,
                                 kernel_size, upsampling_factor, out_channels)
        self.out_channels = out_channels if out_channels > 0 else n_quantize
        self.device = torch.device(device)
        n = self.lib.wn_param_count(ctypes.byref(self.cfg))
        if n <= 0:
            raise _lib.WnError("invalid WaveNet configuration: %s" % self.lib.wn_last_error().decode())
        self.n_params = int(n)
        self.n_layers = self.lib.wn_num_layers(ctypes.byref(self.cfg))
        self.receptive_field = self.lib.wn_receptive_field(ctypes.byref(self.cfg))
        self.flat_params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.flat_grads = None
        self._ws = None
        self._ws_key = None
        self._last_shape = None
        self._fwd_window = 0      # first loss position of the last forward_loss (0: a full forward)
        self._fwd_version = None  # parameter version the last forward packed its weight sets from
        self._fwd_flags = 0       # launch-mode flags of the last forward
        self._dlogits_bound = None  # (weak reference, version, max |dlogits|) of the gradient tensor the last loss call of this engine made
        self._params_epoch = 0    # bumped by every in-library parameter update (adam_step): torch cannot see those
        self._version_sources = ()  # tensors aliasing flat_params whose in-place version counters count as well
        self.ws_finite = True     # workspace() allocates zero-filled memory -> WN_FLAG_WS_FINITE (tests clear it)
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        self.lib.check(self.lib.wn_dead_param_range(ctypes.byref(self.cfg), ctypes.byref(lo), ctypes.byref(hi)),
                       "wn_dead_param_range")
        self.dead_range = (lo.value, hi.value)
        # launch-mode flags of wn_forward / wn_backward (include/wavenet_hip.h WN_FLAG_*); WN_ENGINE_FLAGS overrides
        # the default for A/B measurements (tools/overlap_probe.py)
        self.flags = int(os.environ.get("WN_ENGINE_FLAGS", str(DEFAULT_FLAGS)), 0)

    # ---- layout ---------------------------------------------------------------------------
    def param_slice(self, kind, layer=0):
        off, n = ctypes.c_int64(), ctypes.c_int64()
        self.lib.check(self.lib.wn_param_offset(ctypes.byref(self.cfg), kind, layer, ctypes.byref(off), ctypes.byref(n)),
                       "wn_param_offset")
        return off.value, n.value

    def bucket_ranges(self, layers_per_bucket):
        nb = self.lib.wn_num_buckets(ctypes.byref(self.cfg), layers_per_bucket)
        out = []
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        for i in range(nb):
            self.lib.check(self.lib.wn_bucket_range(ctypes.byref(self.cfg), layers_per_bucket, i, ctypes.byref(lo),
                                                    ctypes.byref(hi)), "wn_bucket_range")
            out.append((lo.value, hi.value))
        return out

    # ---- device management ----------------------------------------------------------------
    def to(self, device):
        device = torch.device(device)
        if device != self.device:
            self.flat_params = self.flat_params.to(device)
            self.flat_grads = None if self.flat_grads is None else self.flat_grads.to(device)
            self._ws = None
            self._ws_key = None
            self.device = device
        return self

    def _check_device(self, *tensors):
        if self.lib.is_emulator:
            for t in tensors:
                if t is not None and t.device.type != "cpu":
                    raise _lib.WnError("emulator binding takes CPU tensors")
            return
        if self.device.type != "cuda":
            raise _lib.WnError("the WaveNet HIP path runs on an MI355X: move the model to a GPU "
                               "(model.cuda()); there is no CPU fallback")
        for t in tensors:
            if t is not None and t.device != self.device:
                raise _lib.WnError("tensor on %s but the model lives on %s" % (t.device, self.device))

    def workspace(self, B, T):
        key = (B, T)
        if self._ws_key != key:
            nbytes = self.lib.wn_workspace_bytes(ctypes.byref(self.cfg), B, T)
            if nbytes == 0:
                raise _lib.WnError("wn_workspace_bytes: %s" % self.lib.wn_last_error().decode())
            self._ws = None  # free first
            # zero-filled: wn_forward_loss (FLAG_WS_FINITE) leaves the post-net columns in front of the loss window
            # untouched, and whatever is there must be finite (include/wavenet_hip.h)
            self._ws = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=self.device)
            self._ws_key = key
        return self._ws

    def params_version(self):
        """Changes whenever the flat parameter buffer is written: torch's in-place version counter (optimizers working
        on the nn.Parameter views, load_state_dict, ...) plus the engine's count of in-library updates."""
        v = self.flat_params._version
        for t in self._version_sources:   # nn.Parameter views keep their own counters (nets.WaveNet registers them)
            v += t._version
        return (self.flat_params.data_ptr(), v, self._params_epoch)

    def saved(self, kind):
        """View of a tensor the last forward / backward left in the workspace (``_lib.WS_*``; parity tests only)."""
        if self._last_shape is None:
            raise _lib.WnError("saved() without a preceding forward()")
        B, T = self._last_shape
        off, n = ctypes.c_int64(), ctypes.c_int64()
        self.lib.check(self.lib.wn_workspace_region(ctypes.byref(self.cfg), B, T, int(kind), ctypes.byref(off),
                                                    ctypes.byref(n)), "wn_workspace_region")
        R, S, L = self.cfg.n_resch, self.cfg.n_skipch, self.n_layers
        shape = {_lib.WS_RELU_SKIP: (B, S, T), _lib.WS_RELU_POST1: (B, S, T), _lib.WS_DSKIP: (B, S, T),
                 _lib.WS_DP: (L, B, 2 * R, T)}.get(int(kind), (L, B, R, T))
        return self.workspace(B, T)[off.value:off.value + n.value].view(shape)

    def grads(self):
        if self.flat_grads is None:
            self.flat_grads = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        return self.flat_grads

    # ---- C-ABI calls ------------------------------------------------------------------------
    def forward(self, x, h):
        """x (B,T) int64, h (B,A,T/U | T) fp32 -> logits (B,Q,T) fp32 (physical layout)."""
        self._check_device(x, h)
        if x.dtype != torch.int64 or x.dim() != 2:
            raise ValueError("x must be a LongTensor of shape (B, T)")
        B, T = x.shape
        U = self.cfg.upsampling_factor
        want = (B, self.cfg.n_aux, T // U if U > 0 else T)
        if U > 0 and T % U != 0:
            raise ValueError("T=%d is not a multiple of upsampling_factor=%d" % (T, U))
        if tuple(h.shape) != want:
            raise ValueError("h must have shape %s, got %s" % (want, tuple(h.shape)))
        x = x.contiguous()
        h = h.contiguous().float()
        ws = self.workspace(B, T)
        logits = torch.empty((B, self.out_channels, T), dtype=torch.float32, device=self.device)
        rc = self.lib.wn_forward(ctypes.byref(self.cfg), B, T, _ptr(self.flat_params), _ptr(x), _ptr(h), _ptr(logits),
                                 _ptr(ws), ws.numel() * 4, self.flags, _stream_handle(self.device))
        self.lib.check(rc, "wn_forward")
        self._last_shape = (B, T)
        self._last_inputs = (x, h)
        self._fwd_window = 0
        self._fwd_version = self.params_version()
        self._fwd_flags = self.flags
        return logits

    def forward_loss(self, x, h, target, t_start=None, grad_scale=1.0, loss_scale=1.0, want_grad=True):
        """``forward`` + ``loss`` of a training step in one call (reference train.py:533-536).  Softmax head with at most
        256 classes on the split contractions: the cross-entropy is the EPILOGUE of the conv_post_2 contraction, the
        (B, Q, T) logits never reach memory (``wn_forward_loss``); otherwise the two calls back to back.
        Returns (loss, dlogits) exactly as ``loss`` does."""
        self._check_device(x, h, target)
        if x.dtype != torch.int64 or x.dim() != 2:
            raise ValueError("x must be a LongTensor (B, T)")
        B, T = x.shape
        if t_start is None:
            t_start = self.receptive_field
        flen = T // self.cfg.upsampling_factor if self.cfg.upsampling_factor > 0 else T
        if h.dim() != 3 or h.size(0) != B or h.size(1) != self.cfg.n_aux or h.size(2) != flen:
            raise ValueError("h must be (B, n_aux, %d)" % flen)
        if self.cfg.upsampling_factor > 0 and T % self.cfg.upsampling_factor != 0:
            raise ValueError("T must be a multiple of the upsampling factor")
        x, h, target = x.contiguous(), h.contiguous().float(), target.contiguous()
        cfg = ctypes.byref(self.cfg)
        fused = bool(self.lib.wn_forward_loss_fused(cfg, B, T, self.flags))
        ws = self.workspace(B, T)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        dlogits = torch.empty((B, self.out_channels, T), dtype=torch.float32, device=self.device) if want_grad else None
        scratch = None if fused else torch.empty((B, self.out_channels, T), dtype=torch.float32, device=self.device)
        rc = self.lib.wn_forward_loss(cfg, B, T, _ptr(self.flat_params), _ptr(x), _ptr(h), _ptr(target), int(t_start),
                                      float(grad_scale), float(loss_scale), _ptr(loss), _ptr(dlogits), _ptr(scratch),
                                      _ptr(ws), ws.numel() * 4, self.flags | (_lib.FLAG_WS_FINITE if self.ws_finite else 0),
                                      _stream_handle(self.device))
        self.lib.check(rc, "wn_forward_loss")
        self._last_shape = (B, T)
        self._last_inputs = (x, h)
        # the fused form ran the skip sum / post-net over the loss window only (saved(WS_RELU_*) is valid from
        # t_start rounded down to a multiple of 128 on); backward() defaults its window to it
        self._fwd_window = int(t_start) if fused else 0
        self._fwd_version = self.params_version()
        self._fwd_flags = self.flags
        self._note_bound(dlogits, grad_scale, B * (T - int(t_start)))
        return loss, dlogits

    def _note_bound(self, dlogits, grad_scale, count):
        # The loss calls leave max |dlogits| of the tensor they wrote in the workspace (measured in their epilogue); backward() may
        # tell the library to use it (WN_FLAG_DW_F16_AMAX_WS) only for exactly that tensor in exactly that state, so the tensor
        # OBJECT is remembered, weakly (an address could be handed out again by the caching allocator), with its version counter.
        self._dlogits_bound = None
        if dlogits is not None and count > 0 and grad_scale != 0.0:
            self._dlogits_bound = (weakref.ref(dlogits), dlogits._version, self._ws_key)

    def _dw_mode_flags(self, flags, dlogits, dlogits_bound=None):
        """FLAG_DW_F16PAIR scales the gradient operand by max |dlogits| (include/wavenet_hip.h).  Where it comes from:
          * ``dlogits_bound`` given: the caller's promise, taken as given (a loose bound costs precision -- prefer None);
          * ``dlogits`` is the unmodified tensor the last loss call of this engine returned, same workspace: the maximum that call
            measured (| FLAG_DW_F16_AMAX_WS: free);
          * anything else (autograd's grad_output, a tensor modified in place, the mixture-of-logistics head): the library
            scans the tensor for its maximum first (the flag alone: one pass over the loss window).
        Never a silent default: an a-priori bound that is merely safe would underflow fp16's range."""
        if not (flags & _lib.FLAG_DW_F16PAIR):
            return flags
        flags &= ~((63 << _lib.DW_F16_EXP_SHIFT) | _lib.FLAG_DW_F16_EXP_VALID | _lib.FLAG_DW_F16_AMAX_WS)
        if dlogits_bound is not None:
            return flags | _lib.dw_f16_exp(float(dlogits_bound))
        nb = self._dlogits_bound
        if nb is not None and nb[0]() is dlogits and nb[1] == dlogits._version and nb[2] == self._ws_key:
            return flags | _lib.FLAG_DW_F16_AMAX_WS
        return flags

    def loss(self, logits, target, t_start=None, grad_scale=1.0, loss_scale=1.0, want_grad=True):
        """Softmax-CE over positions >= t_start (default: receptive field).  Returns (loss, dlogits)."""
        self._check_device(logits, target)
        B, Q, T = logits.shape
        if t_start is None:
            t_start = self.receptive_field
        target = target.contiguous()
        ws = self.workspace(B, T)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        dlogits = torch.empty_like(logits) if want_grad else None
        rc = self.lib.wn_softmax_ce_loss(ctypes.byref(self.cfg), B, T, _ptr(logits), _ptr(target), int(t_start),
                                         float(grad_scale), float(loss_scale), _ptr(loss), _ptr(dlogits), _ptr(ws),
                                         ws.numel() * 4, _stream_handle(self.device))
        self.lib.check(rc, "wn_softmax_ce_loss")
        self._note_bound(dlogits, grad_scale, B * (T - int(t_start)))
        return loss, dlogits

    def mol_loss(self, out, y, t_start=None, grad_scale=1.0, loss_scale=1.0, want_grad=True, num_classes=65536,
                 log_scale_min=-7.0):
        """Mixture-of-logistics head: mean NLL of the waveform ``y`` (B,T) in [-1,1] under ``out`` (B, 3*n_mix, T)
        over positions >= t_start, and d(loss)/d(out) for ``backward``.  Not in the reference (see include/)."""
        self._check_device(out, y)
        B, C, T = out.shape
        if C != self.out_channels or C % 3 != 0:
            raise ValueError("out must be (B, 3*n_mix = %d, T)" % self.out_channels)
        if t_start is None:
            t_start = self.receptive_field
        y = y.contiguous().float()
        ws = self.workspace(B, T)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        dout = torch.empty_like(out) if want_grad else None
        rc = self.lib.wn_mol_loss(ctypes.byref(self.cfg), B, T, _ptr(out), _ptr(y), int(t_start), float(grad_scale),
                                  float(loss_scale), int(num_classes), float(log_scale_min), _ptr(loss), _ptr(dout),
                                  _ptr(ws), ws.numel() * 4, _stream_handle(self.device))
        self.lib.check(rc, "wn_mol_loss")
        self._dlogits_bound = None   # (the mixture head's loss call measures no maximum: backward scans its gradient)
        return loss, dout

    def backward(self, dlogits, events=None, layers_per_bucket=0, t_first=None, repack=False, dlogits_bound=None):
        """Backward of the last ``forward`` / ``forward_loss`` call; fills ``self.grads()`` completely.  ``t_first``: the
        caller guarantees ``dlogits[:, :, :t_first] == 0`` (the training loss covers ``[:, receptive_field:]``,
        train.py:534-536): the post-net / skip part of the backward pass then runs over the loss window only
        (``wn_backward_window``).  Default: the window of the forward call (0 after ``forward``, its ``t_start`` after
        ``forward_loss``, whose ``dlogits`` is zero in front of it by construction).

        The parameters must not have changed since the forward call -- the workspace holds the weight sets that call
        packed from them next to its activations.  Like torch.autograd for a tensor modified in place between forward
        and backward this raises; ``repack=True`` instead rebuilds the weight sets from the current parameters
        (``WN_FLAG_REPACK``) and back-propagates through those.

        ``dlogits_bound``: with ``FLAG_DW_F16PAIR`` in ``self.flags`` the weight gradients take the fp16 pair split
        (include/wavenet_hip.h), scaled by max |dlogits|.  Default (None): MEASURED -- by the loss call for the unmodified
        tensor ``forward_loss`` / ``loss`` returned (free), by one pass over any other gradient (autograd's grad_output, the
        mixture-of-logistics head).  A number: the caller's own promise ``max |dlogits| <= dlogits_bound``, taken as given."""
        if self._last_shape is None:
            raise _lib.WnError("backward() without a preceding forward()")
        if t_first is None:
            t_first = self._fwd_window
        flags = self.flags
        family = _lib.FLAG_NO_FUSED | _lib.FLAG_EXACT_MFMA | _lib.FLAG_MM_F16PAIR | _lib.FLAG_CHAIN_F16PAIR
        if (flags ^ self._fwd_flags) & family and not repack:
            raise _lib.WnError("engine.flags changed the kernel family (NO_FUSED / EXACT_MFMA / MM_F16PAIR) since the forward call: "
                               "the families save different activations and weight sets -- run forward again")
        flags = self._dw_mode_flags(flags, dlogits, dlogits_bound)
        if repack:
            flags |= _lib.FLAG_REPACK
        elif self._fwd_version is not None and self._fwd_version != self.params_version():
            raise _lib.WnError("the WaveNet parameters were modified between forward and backward (in-place update, "
                               "optimizer step or load_state_dict): the gradient of the forward pass is no longer "
                               "defined -- run forward again, or pass repack=True to back-propagate through the "
                               "current weights")
        B, T = self._last_shape
        x, h = self._last_inputs
        self._check_device(dlogits)
        if tuple(dlogits.shape) != (B, self.out_channels, T) or not dlogits.is_contiguous():
            raise ValueError("dlogits must be a contiguous (B,Q,T) tensor")
        g = self.grads()
        ws = self.workspace(B, T)
        if events:
            arr = (ctypes.c_void_p * len(events))(*[ctypes.c_void_p(e) for e in events])
            n_ev = len(events)
        else:
            arr, n_ev = None, 0
        rc = self.lib.wn_backward_window(ctypes.byref(self.cfg), B, T, _ptr(self.flat_params), _ptr(x), _ptr(h),
                                         _ptr(dlogits), int(t_first), _ptr(g), _ptr(ws), ws.numel() * 4, arr, n_ev,
                                         int(layers_per_bucket), flags, _stream_handle(self.device))
        self.lib.check(rc, "wn_backward_window")
        return g

    def adam_step(self, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self._check_device(exp_avg, exp_avg_sq)
        rc = self.lib.wn_adam_step(_ptr(self.flat_params), _ptr(self.grads()), _ptr(exp_avg), _ptr(exp_avg_sq),
                                   self.n_params, int(step), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                   float(weight_decay), self.dead_range[0], self.dead_range[1],
                                   _stream_handle(self.device))
        self.lib.check(rc, "wn_adam_step")
        self._params_epoch += 1

    # ---- autoregressive decode (reference wavenet.py:309-511) ---------------------------------
    def decode_supported(self):
        return bool(self.lib.wn_decode_supported(ctypes.byref(self.cfg)))

    def decode(self, x, h, n_samples_list, mode="argmax", chunk=4096, return_logits=False, progress=None, layered=None,
               prefill="parallel", prefill_batch=32, log_scale_min=-7.0):
        """See ``_decode``.  ``layered="granules"``: the any-size path with its persistent launches handing their vectors over
        as 8-byte granules everywhere (csrc/wn_dlp.hip, wn_dlpm.hip) instead of plain vectors + flags where csrc/wn_dlpf.hip
        covers the model -- an A/B and test knob (the mode bit WN_DECODE_GRANULES of every library call of this decode).

        A persistent launch that gives up waiting between its workgroups (they were not all resident: another kernel held
        compute units) raises ``WnDecodeTimeout`` inside; the decode is then done again by layer-wise launches (HIP kernels
        as well; a warning says so) -- context, aux features and, in the sampling mode, the uniform draws are the same, so
        the tokens are those the persistent launch would have produced."""
        granules = layered == "granules"
        lay = True if granules else layered
        mbits = _lib.DECODE_GRANULES if granules else 0
        draws = None
        if mode == "sampling":   # drawn once for the whole batch: neither the grouping nor a fall-back changes the tokens
            n_pad = max(self.receptive_field - x.size(1), 0)
            draws = torch.rand((x.size(0), x.size(1) + n_pad + int(max(n_samples_list))), dtype=torch.float32, device=self.device)
        try:
            return self._decode_groups(x, h, n_samples_list, mode, chunk, return_logits, progress, lay, prefill, prefill_batch,
                                       log_scale_min, mbits, draws)
        except _lib.WnDecodeTimeout as e:
            warnings.warn("%s -- decoding again by layer-wise launches" % e, RuntimeWarning)
            return self._decode_groups(x, h, n_samples_list, mode, chunk, return_logits, progress, "launches", prefill,
                                       prefill_batch, log_scale_min, 0, draws)

    def _decode_groups(self, x, h, n_samples_list, mode, chunk, return_logits, progress, lay, prefill, prefill_batch, log_scale_min,
                       mbits, draws):
        groups = self._persistent_groups(x.size(0), lay, mode, mbits)
        if groups is None:
            return self._decode(x, h, n_samples_list, mode, chunk, return_logits, progress, lay, prefill, prefill_batch,
                                log_scale_min, _uniforms=draws, _mbits=mbits)
        # More utterances than ONE persistent launch takes (64 with the flag hand-off: four column blocks = 256 workgroups), up to
        # two launches' worth: the batch goes through it in two groups -- the utterances are independent.
        T0 = x.size(1)
        n_pad = max(self.receptive_field - T0, 0)
        toks, lgs = [], []
        total = sum(int(max(n_samples_list[g0:g1])) for g0, g1 in groups)
        done = 0
        for g0, g1 in groups:
            ns = list(n_samples_list[g0:g1])
            u = None if draws is None else draws[g0:g1, :T0 + n_pad + int(max(ns))].contiguous()
            prog = None if progress is None else (lambda d, n, base=done: progress(base + d, total))   # steps over all groups
            r = self._decode(x[g0:g1].contiguous(), h[g0:g1].contiguous(), ns, mode, chunk, return_logits, prog, lay, prefill,
                             prefill_batch, log_scale_min, _uniforms=u, _mbits=mbits)
            done += int(max(ns))
            toks += r[0] if return_logits else r
            if return_logits:
                lgs += r[1]
        self.last_uniforms = draws
        return (toks, lgs) if return_logits else toks

    def decode_residency(self, B, layered=None):
        """(persistent, workgroups, capacity) of the any-size decode of B utterances on this device: whether it runs as ONE
        persistent launch, the workgroups that launch needs resident at once and how many the device keeps
        (``wn_decode_layered_residency``)."""
        wg, cap = ctypes.c_int(0), ctypes.c_int(0)
        rc = self.lib.wn_decode_layered_residency(ctypes.byref(self.cfg), int(B), _lib.DECODE_GRANULES if layered == "granules" else 0,
                                                  ctypes.byref(wg), ctypes.byref(cap))
        if rc < 0:
            raise _lib.WnError("wn_decode_layered_residency: %s" % self.lib.wn_last_error().decode())
        return bool(rc), wg.value, cap.value

    def _persistent_groups(self, B, layered, mode, mbits=0):
        """[(begin, end)] when a batch of B utterances should go through the persistent any-size launch in groups, else None: the
        any-size path is asked for (or chosen because the one-workgroup kernel does not cover the model), B is more than one
        launch takes, and a launch of the group size exists."""
        if layered == "launches" or mode not in ("argmax", "sampling") or B <= 1:
            return None
        if layered is None and self.decode_supported():
            return None
        if layered is not None and not layered:
            return None
        cfg = ctypes.byref(self.cfg)
        if self.lib.wn_decode_layered_error_offset(cfg, B, mbits) >= 0:
            return None   # one launch takes the whole batch
        gs = 0
        for cand in (64, 48, 32, 16, 8, 4, 2):   # whole column blocks of 16 first; what the device keeps resident decides (wn_dlp.h)
            if self.lib.wn_decode_layered_error_offset(cfg, cand, mbits) >= 0:
                gs = cand
                break
        if gs <= 1 or B <= gs or B > 2 * gs:
            return None   # (three groups and more: the layer-wise launches on the whole batch are as fast or faster -- n_resch 512:
                          # 777 against 751 us per step at 128 utterances, 1575 against 837 at 256; 509 against 710 at 64)
        return [(g0, min(g0 + gs, B)) for g0 in range(0, B, gs)]

    def _decode(self, x, h, n_samples_list, mode="argmax", chunk=4096, return_logits=False, progress=None, layered=None,
                prefill="parallel", prefill_batch=32, log_scale_min=-7.0, _uniforms=None, _mbits=0):
        """Queue-based sample-by-sample generation on the HIP decode kernel.

        x (B,T0) int64 context, h (B, n_aux, frames | samples) aux features covering T0 + max(n)
        samples, n_samples_list: samples to generate per utterance.  Follows the reference's prologue
        (wavenet.py:328-336, 417-425): the context is left-padded with n_quantize//2 up to the
        receptive field and the aux features by replicating their first column.  Returns the
        generated tokens as a list of LongTensors (utterance order) [and the per-step logits].

        Two kernels implement it: the persistent one-workgroup-per-utterance kernel (model sizes covered by
        ``decode_supported()``) and the any-size path (``layered=True``; chosen automatically when the first does not
        apply, e.g. the n_resch = 512 recipe default).  The any-size path is itself ONE persistent launch per chunk of
        steps where csrc/wn_dlp.hip / wn_dlpm.hip / wn_dlpf.hip cover the model and the batch (up to 64 utterances: workgroups
        handing their vectors to each other) and layer-wise launches otherwise; ``layered="launches"`` forces the launches (independent check, A/B).

        ``prefill``: how the dilation queues of the context are built.  "parallel" (default) does what the
        reference does (wavenet.py:338-349): one forward of the residual stack over the whole padded context
        (``prefill_batch`` utterances at a time), then decoding starts at the last context position.  "walk"
        steps the decode kernel through the context sample by sample (teacher forced; receptive-field steps
        before the first new sample) -- kept as the independent check of the former.
        ``log_scale_min`` (mode "mol"): clamp of the mixture log-scales, the value the model was trained with.
        """
        self._check_device(x, h)
        if x.dtype != torch.int64 or x.dim() != 2 or h.dim() != 3:
            raise ValueError("x must be a LongTensor (B, T) and h a FloatTensor (B, n_aux, F)")
        if mode not in ("argmax", "sampling", "mol"):
            raise ValueError("mode should be sampling, argmax or mol")
        if mode == "mol" and (self.out_channels % 3 != 0 or self.out_channels == self.cfg.n_quantize):
            raise ValueError("mode mol needs a model with the mixture-of-logistics head (n_mixture)")
        n_mix = self.out_channels // 3 if mode == "mol" else 0
        cfg = ctypes.byref(self.cfg)
        B, T0 = x.shape
        if len(n_samples_list) != B or h.size(0) != B or h.size(1) != self.cfg.n_aux:
            raise ValueError("batch mismatch between x, h and n_samples_list")
        U = self.cfg.upsampling_factor
        n_max = int(max(n_samples_list))
        need = T0 + n_max - 1   # last aux column read: sample index T0 + n_max - 2
        F = h.size(2)
        if (F * U if U > 0 else F) < need:
            raise ValueError("h covers %d samples, %d needed" % (F * U if U > 0 else F, need))
        n_pad = max(self.receptive_field - T0, 0)
        Tctx = T0 + n_pad
        Ttot = Tctx + n_max
        if layered is None:
            layered = not self.decode_supported()
        mbits = (_lib.DECODE_BY_LAUNCHES if layered == "launches" else 0) | int(_mbits)   # the same bits to every call of this decode
        layered = bool(layered)
        if prefill not in ("parallel", "walk"):
            raise ValueError("prefill should be parallel or walk")
        st = _stream_handle(self.device)
        dev = self.device
        h = h.contiguous().float()
        nG = self.n_layers * 2 * self.cfg.n_resch
        # Aux projections G (B, columns, L*2R).  With the upsampling layer the columns are frames (small): all of them
        # up front.  Without it (U = 0, decode.py's extend_time path) a column is a SAMPLE -- 5 s at 16 kHz with the
        # recipe-size model would be 9.8 GB per utterance -- so G only ever holds the columns of the chunk of steps
        # being decoded (at most ~1 GiB), recomputed per chunk from the matching slice of h.
        windowed = (U == 0)
        if windowed:
            chunk = max(1, min(int(chunk), (1 << 30) // (4 * nG * B)))
        wpack = None
        if layered:
            nst = self.lib.wn_decode_layered_state_floats(cfg, B, mbits)
            if nst <= 0:
                raise _lib.WnError("wn_decode_layered_state_floats: %s" % self.lib.wn_last_error().decode())
            state = torch.zeros(nst, dtype=torch.float32, device=dev)
        else:
            npk = self.lib.wn_decode_pack_floats(cfg)
            if npk <= 0:
                raise _lib.WnError("wn_decode_pack_floats: %s" % self.lib.wn_last_error().decode())
            wpack = torch.empty(npk, dtype=torch.float32, device=dev)
            self.lib.check(self.lib.wn_decode_pack(cfg, _ptr(self.flat_params), _ptr(wpack), st), "wn_decode_pack")
            state = torch.zeros((B, self.lib.wn_decode_state_floats(cfg)), dtype=torch.float32, device=dev)

        packed = [False]

        def aux_columns(c0, c1):
            """G of aux columns [c0, c1).  The layered variant packs the weights into its state on the first call only
            (params = NULL afterwards: a windowed decode comes here once per chunk of steps)."""
            hw = h if (c0 == 0 and c1 == F) else h[:, :, c0:c1].contiguous()
            Gw = torch.empty((B, c1 - c0, nG), dtype=torch.float32, device=dev)
            if layered:
                self.lib.check(self.lib.wn_decode_layered_prepare(cfg, B, c1 - c0, None if packed[0] else _ptr(self.flat_params),
                                                                  _ptr(hw), _ptr(Gw), _ptr(state), nst, mbits, st),
                               "wn_decode_layered_prepare")
                packed[0] = True
            else:
                self.lib.check(self.lib.wn_decode_aux(cfg, B, c1 - c0, _ptr(wpack), _ptr(hw), _ptr(Gw), st), "wn_decode_aux")
            return Gw

        G = None if windowed else aux_columns(0, F)
        samples = torch.full((B, Ttot), self.cfg.n_quantize // 2, dtype=torch.int64, device=dev)
        samples[:, n_pad:Tctx] = x
        t_forced = torch.full((B,), Tctx, dtype=torch.int32, device=dev)
        t_end = torch.tensor([Tctx + int(n) for n in n_samples_list], dtype=torch.int32, device=dev)
        if mode == "mol":   # n_mix Gumbel draws + one logistic draw per position, kept away from 0 and 1
            uniforms = torch.rand((B, Ttot, n_mix + 1), dtype=torch.float32, device=dev).clamp_(1e-5, 1.0 - 1e-5)
            wave = torch.zeros((B, Ttot), dtype=torch.float32, device=dev)
        else:
            uniforms = None
            if mode == "sampling":   # (a group of a larger batch brings its rows of the whole batch's draws)
                uniforms = _uniforms if _uniforms is not None else torch.rand((B, Ttot), dtype=torch.float32, device=dev)
                if tuple(uniforms.shape) != (B, Ttot):
                    raise ValueError("uniforms must be (B, Ttot)")
            wave = None
        logits = torch.zeros((B, Ttot, self.out_channels), dtype=torch.float32, device=dev) if return_logits else None
        eoff = -1   # float offset of the persistent launch's error word in the state (-1: layer-wise launches / other kernel)
        if layered and not (mbits & _lib.DECODE_BY_LAUNCHES) and mode != "mol":
            eoff = self.lib.wn_decode_layered_error_offset(cfg, B, mbits)
        p = 0
        if prefill == "parallel" and Tctx >= 2:
            self._decode_prefill(samples, h, F, Tctx, n_pad, state, (1 | int(_mbits)) if layered else 0, int(prefill_batch), st)
            p = Tctx - 1
        while p < Ttot - 1:
            p1 = min(p + chunk, Ttot - 1)
            if windowed:
                # steps [p, p1) read the aux columns max(q - n_pad, 0) for q in [p, p1): compute that window and shift
                # the kernel's padding origin by its first column (a position left of it can only be inside the left
                # padding, which replicates column 0 = the window's first column then)
                c0 = min(max(p - n_pad, 0), F - 1)
                c1 = min(max(p1 - 1 - n_pad, 0), F - 1) + 1
                G = None  # release the previous window first
                G = aux_columns(c0, c1)
                Fw, pad_w = c1 - c0, n_pad + c0
            else:
                Fw, pad_w = F, n_pad
            if layered:
                rc = self.lib.wn_decode_layered_steps(cfg, B, _ptr(self.flat_params), _ptr(G), Fw, pad_w, _ptr(samples), Ttot,
                                                      _ptr(t_forced), _ptr(t_end), p, p1, _ptr(state), state.numel(),
                                                      _ptr(uniforms), _ptr(logits),
                                                      {"argmax": 0, "sampling": 1, "mol": 2}[mode] | mbits,
                                                      _ptr(wave), float(log_scale_min), st)
                if rc == 4 and eoff >= 0:   # the persistent grid would not be resident at once (the device changed under a cached
                    # layout, a partitioned GPU): the same fall-back as a time-out
                    raise _lib.WnDecodeTimeout("the persistent decode launch was refused: its %s" % self.lib.wn_last_error().decode())
                self.lib.check(rc, "wn_decode_layered_steps")
                # the persistent launch bounds every wait between its workgroups and reports a time-out in the state; a launch
                # that finds the word set returns at once, so it is looked at after every chunk (one word, once per chunk)
                if eoff >= 0 and int(state[eoff:eoff + 1].view(torch.int32).item()) != 0:
                    raise _lib.WnDecodeTimeout("the persistent decode launch timed out waiting between its workgroups (not all "
                                               "of them were resident at once: another kernel on the device?)")
            else:
                rc = self.lib.wn_decode_steps(cfg, B, _ptr(self.flat_params), _ptr(wpack), _ptr(G), Fw, pad_w, _ptr(samples),
                                              Ttot, _ptr(t_forced), _ptr(t_end), p, p1, _ptr(state), _ptr(uniforms),
                                              _ptr(logits), {"argmax": 0, "sampling": 1, "mol": 2}[mode], _ptr(wave),
                                              float(log_scale_min), st)
                self.lib.check(rc, "wn_decode_steps")
            p = p1
            if progress is not None:
                progress(max(p + 1 - Tctx, 0), n_max)
        self.last_decode_state = state if layered else None   # (tools/dlp_timing.py reads a timing build's stamps from it)
        out = [samples[b, Tctx:Tctx + int(n)] for b, n in enumerate(n_samples_list)]
        self.last_wave = None if wave is None else [wave[b, Tctx:Tctx + int(n)] for b, n in enumerate(n_samples_list)]
        self.last_uniforms = uniforms
        if return_logits:   # row Tctx-1+i holds the logits that chose generated sample i
            return out, [logits[b, Tctx - 1:Tctx - 1 + int(n)] for b, n in enumerate(n_samples_list)]
        return out

    def _decode_prefill(self, samples, h, F, Tctx, n_pad, state, layered, nbatch, st):
        """Dilation queues of the context from one forward of the residual stack (wn_decode_prefill)."""
        cfg = ctypes.byref(self.cfg)
        B = samples.size(0)
        dev = self.device
        # Only the newest rf + K - 1 positions reach the queues (rf - 1 through the dilated stack, K - 1 more
        # through the causal front conv): a longer context is passed as its tail.
        pos0 = max(0, Tctx - (self.receptive_field + self.cfg.kernel_size - 1))
        Tpre = Tctx - pos0
        # groups of utterances with a bounded workspace (the training workspace of nb x Tpre inputs)
        per_utt = self.lib.wn_decode_prefill_workspace_bytes(cfg, 1, Tpre)
        if per_utt == 0:
            raise _lib.WnError("wn_decode_prefill_workspace_bytes: %s" % self.lib.wn_last_error().decode())
        nbatch = max(1, min(nbatch, (8 << 30) // per_utt))
        ws = None
        for b0 in range(0, B, nbatch):
            nb = min(nbatch, B - b0)
            x_ctx = samples[b0:b0 + nb, pos0:Tctx].contiguous()
            h_ctx = torch.empty((nb, self.cfg.n_aux, Tpre), dtype=torch.float32, device=dev)
            self.lib.check(self.lib.wn_decode_ctx_aux(cfg, nb, F, Tpre, n_pad, pos0, _ptr(self.flat_params),
                                                      _ptr(h[b0:b0 + nb]), _ptr(h_ctx), st), "wn_decode_ctx_aux")
            nbytes = self.lib.wn_decode_prefill_workspace_bytes(cfg, nb, Tpre)
            if ws is None or ws.numel() * 4 < nbytes:
                ws = None
                ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
            rc = self.lib.wn_decode_prefill(cfg, nb, Tpre, pos0, _ptr(self.flat_params), _ptr(x_ctx), _ptr(h_ctx), _ptr(ws),
                                            ws.numel() * 4, _ptr(state), state.numel(), B, b0, int(layered), self.flags,
                                            st)
            self.lib.check(rc, "wn_decode_prefill")



# ---- reference state_dict keys <-> flat buffer ------------------------------------------------
_LIST_KINDS = {
    "dil_sigmoid": (_lib.P_DSIG_W, _lib.P_DSIG_B, True),
    "dil_tanh": (_lib.P_DTANH_W, _lib.P_DTANH_B, True),
    "aux_1x1_sigmoid": (_lib.P_ASIG_W, _lib.P_ASIG_B, False),
    "aux_1x1_tanh": (_lib.P_ATANH_W, _lib.P_ATANH_B, False),
    "skip_1x1": (_lib.P_SKIP_W, _lib.P_SKIP_B, False),
    "res_1x1": (_lib.P_RES_W, _lib.P_RES_B, False),
}
_SINGLE_KINDS = {
    "causal.conv.weight": _lib.P_CAUSAL_W, "causal.conv.bias": _lib.P_CAUSAL_B,
    "upsampling.conv.weight": _lib.P_UP_W, "upsampling.conv.bias": _lib.P_UP_B,
    "conv_post_1.weight": _lib.P_POST1_W, "conv_post_1.bias": _lib.P_POST1_B,
    "conv_post_2.weight": _lib.P_POST2_W, "conv_post_2.bias": _lib.P_POST2_B,
}


def key_to_kind(key):
    """Reference state_dict key (wavenet.py:187-210) -> (tensor kind, layer)."""
    if key in _SINGLE_KINDS:
        return _SINGLE_KINDS[key], 0
    parts = key.split(".")
    name, layer = parts[0], int(parts[1])
    wk, bk, has_conv = _LIST_KINDS[name]
    expect = 4 if has_conv else 3
    if len(parts) != expect or (has_conv and parts[2] != "conv"):
        raise KeyError(key)
    return (wk if parts[-1] == "weight" else bk), layer


def state_keys(cfg):
    """All state_dict keys in the reference's registration order."""
    L = cfg.dilation_depth * cfg.dilation_repeat
    keys = ["causal.conv.weight", "causal.conv.bias"]
    if cfg.upsampling_factor > 0:
        keys += ["upsampling.conv.weight", "upsampling.conv.bias"]
    for name in ("dil_sigmoid", "dil_tanh"):
        for l in range(L):
            keys += ["%s.%d.conv.weight" % (name, l), "%s.%d.conv.bias" % (name, l)]
    for name in ("aux_1x1_sigmoid", "aux_1x1_tanh", "skip_1x1", "res_1x1"):
        for l in range(L):
            keys += ["%s.%d.weight" % (name, l), "%s.%d.bias" % (name, l)]
    keys += ["conv_post_1.weight", "conv_post_1.bias", "conv_post_2.weight", "conv_post_2.bias"]
    return keys


def load_state_into_flat(engine, state):
    """Copy a reference-layout state dict {key: tensor} into the engine's flat parameter buffer."""
    for k in state_keys(engine.cfg):
        kind, layer = key_to_kind(k)
        off, n = engine.param_slice(kind, layer)
        v = state[k]
        if v.numel() != n:
            raise ValueError("%s: expected %d elements, got %s" % (k, n, tuple(v.shape)))
        engine.flat_params[off:off + n].copy_(v.reshape(-1).to(engine.flat_params.dtype))


def flat_to_state(engine, flat, shapes):
    """Views of a flat buffer keyed like the reference state dict (shapes: {key: shape})."""
    out = {}
    for k in state_keys(engine.cfg):
        kind, layer = key_to_kind(k)
        off, n = engine.param_slice(kind, layer)
        out[k] = flat[off:off + n].view(shapes[k])
    return out
