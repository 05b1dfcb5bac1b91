# -*- coding: utf-8 -*-
"""ctypes binding of libwavenet_hip.so (C ABI: include/wavenet_hip.h).

The product path loads the gfx950 library built in-tree at ``csrc/libwavenet_hip.so`` and FAILS
LOUDLY when it is missing or cannot be built -- there is no CPU or eager-PyTorch fallback.
(``load_library(path, _test_emulator=True)`` exists only so that tests/ can point the same binding
at the host-compiled kernel emulator; nothing in this package ever passes that flag.)
"""
import contextlib
import ctypes
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libwavenet_hip.so")

c_float_p = ctypes.c_void_p
c_i64_p = ctypes.c_void_p


class WnConfig(ctypes.Structure):
    """Mirror of ``struct WnConfig`` == WaveNet.__init__ arguments (reference wavenet.py:172-173)."""
    _fields_ = [("n_quantize", ctypes.c_int32), ("n_aux", ctypes.c_int32), ("n_resch", ctypes.c_int32),
                ("n_skipch", ctypes.c_int32), ("dilation_depth", ctypes.c_int32),
                ("dilation_repeat", ctypes.c_int32), ("kernel_size", ctypes.c_int32),
                ("upsampling_factor", ctypes.c_int32), ("out_channels", ctypes.c_int32)]


class WnGemmArgs(ctypes.Structure):
    """Mirror of ``struct WnGemmArgs`` (csrc/wn_gemm.h); used by the op-level parity tests."""
    _fields_ = [
        ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int),
        ("A", ctypes.c_void_p), ("lda", ctypes.c_long), ("a_zstride", ctypes.c_long), ("a_kmajor", ctypes.c_int),
        ("B", ctypes.c_void_p), ("ldb", ctypes.c_long), ("b_zstride", ctypes.c_long), ("b_kmajor", ctypes.c_int),
        ("b_seg_len", ctypes.c_int), ("b_seg_stride", ctypes.c_long), ("b_shift0", ctypes.c_int),
        ("b_shift_step", ctypes.c_int), ("b_clen", ctypes.c_int), ("b_relu", ctypes.c_int),
        ("b_index", ctypes.c_void_p), ("b_index_zstride", ctypes.c_long), ("b_index_mod", ctypes.c_int),
        ("C", ctypes.c_void_p), ("ldc", ctypes.c_long), ("c_zstride", ctypes.c_long),
        ("bias", ctypes.c_void_p),
        ("D", ctypes.c_void_p), ("ldd", ctypes.c_long), ("d_zstride", ctypes.c_long),
        ("E", ctypes.c_void_p), ("lde", ctypes.c_long), ("e_zstride", ctypes.c_long),
        ("relu", ctypes.c_int), ("accumulate", ctypes.c_int),
        ("nbatch", ctypes.c_int), ("ksplit", ctypes.c_int), ("kchunk", ctypes.c_int),
        ("a_rowsum", ctypes.c_void_p),
        ("tag", ctypes.c_char_p),
        ("nlayer", ctypes.c_int), ("a_lstride", ctypes.c_long), ("b_lstride", ctypes.c_long),
        ("b_dil_depth", ctypes.c_int), ("b_layer0", ctypes.c_int),
    ]

    @classmethod
    def default(cls):
        g = cls()
        g.b_seg_len = 0x7fffffff
        g.b_index_mod = 1
        g.nbatch = 1
        g.ksplit = 1
        g.kchunk = 0x7fffffff
        g.nlayer = 1
        return g


# tensor kinds (include/wavenet_hip.h)
(P_CAUSAL_W, P_CAUSAL_B, P_UP_W, P_UP_B, P_DSIG_W, P_DSIG_B, P_DTANH_W, P_DTANH_B, P_ASIG_W, P_ASIG_B,
 P_ATANH_W, P_ATANH_B, P_SKIP_W, P_SKIP_B, P_RES_W, P_RES_B, P_POST1_W, P_POST1_B, P_POST2_W, P_POST2_B) = range(20)

# wn_workspace_region kinds (include/wavenet_hip.h WN_WS_*)
WS_X, WS_SIGMOID, WS_TANH, WS_Z, WS_RELU_SKIP, WS_RELU_POST1, WS_DSKIP, WS_DP, WS_DX = range(9)

FLAG_NO_FUSED = 1
FLAG_EXACT_MFMA = 2
FLAG_BWD_OVERLAP = 4  # wn_backward: weight gradients on the library's side stream beside the gate'/dX chain (opt-in)
FLAG_NO_CHAIN = 64  # wn_backward: the former gate' + dX launch pair per layer instead of the fused chain kernel
FLAG_AUX_FUSED = 32  # wn_backward: aux-gradient partial sums in the gate kernel (dP not re-read by aux_bwd); the engine's default
FLAG_BWD_OVERLAP_HEAD = 16  # with FLAG_BWD_OVERLAP: only the post-net / skip weight gradients on the side stream
FLAG_FWD_OVERLAP = 8  # wn_forward: chunked skip-sum on the side stream beside the residual stack (opt-in)
FLAG_WS_FINITE = 1 << 16  # wn_forward_loss: the workspace holds only finite values (the engine allocates it zero-filled)
DECODE_BY_LAUNCHES = 256  # wn_decode_layered_*: mode bit that keeps the layer-wise launches (csrc/wn_dlp.hip otherwise)
DECODE_GRANULES = 512  # wn_decode_layered_*: mode bit, the persistent launches hand over 8-byte granules everywhere (A/B, tests)
FLAG_DW_3PRODUCT = 1 << 18  # wn_backward: weight gradients (leaf results) with 3 of the 6 products of the operand split (opt-in)
FLAG_DW_F16PAIR = 1 << 19  # wn_backward: weight gradients by the fp16 pair split; its scale from max |dlogits|: measured by a scan of the
#                            given tensor (flag alone), as the loss call of the workspace left it (| FLAG_DW_F16_AMAX_WS), or the caller's
#                            promise (| dw_f16_exp(bound), which carries FLAG_DW_F16_EXP_VALID)
DW_F16_EXP_SHIFT = 20
DW_F16_HEADROOM = 8
FLAG_DW_F16_EXP_VALID = 1 << 26
FLAG_DW_F16_AMAX_WS = 1 << 27
FLAG_MM_F16PAIR = 1 << 28  # forward / data-gradient split contractions (k_gemm6) by the fp16 pair split too, each with its conditional redo


def dw_f16_exp(bound):
    """WN_FLAG_DW_F16_EXP_VALID | WN_FLAG_DW_F16_EXP(e) for a gradient with max |dlogits| <= bound: the largest e in [0, 63] with
    bound <= 2^-e."""
    import math
    if not (bound > 0.0) or math.isinf(bound):
        raise ValueError("bound must be a positive finite number")
    e = int(math.floor(-math.log2(bound)))
    while e > 0 and bound > 2.0 ** -e:    # (floating-point log2 at an exact power of two)
        e -= 1
    return ((max(0, min(63, e)) & 63) << DW_F16_EXP_SHIFT) | FLAG_DW_F16_EXP_VALID


FLAG_FUSED_F16PAIR = 1 << 29  # the fused 64-channel forward block on the block-scaled fp16 pair split (k_resblock_fwd_h)
FLAG_CHAIN_F16PAIR = 1 << 30  # the fused backward chain on the block-scaled fp16 pair split (k_chain64s<.., H16>)
# every flag that narrows a contraction below the six bf16 products (engine.SIX_PRODUCT_FLAGS = DEFAULT_FLAGS without them)
NARROW_FLAGS = FLAG_DW_3PRODUCT | FLAG_DW_F16PAIR | FLAG_MM_F16PAIR | FLAG_FUSED_F16PAIR | FLAG_CHAIN_F16PAIR
FLAG_REPACK = 1 << 17  # wn_backward: rebuild the packed / pre-split weight sets from the params given to that call


def flag_dw_flush(n):
    """WN_FLAG_DW_FLUSH(n): weight gradients of at most n walked layers per launch group (wn_backward)."""
    return (int(n) & 0xff) << 8


ABI_VERSION = 9

# every symbol include/wavenet_hip.h declares
EXPORTS = [
    "wn_abi_version", "wn_last_error", "wn_receptive_field", "wn_num_layers", "wn_param_count", "wn_param_offset",
    "wn_num_buckets", "wn_bucket_range", "wn_dead_param_range", "wn_workspace_bytes", "wn_workspace_region", "wn_forward",
    "wn_softmax_ce_loss", "wn_forward_loss_fused", "wn_forward_loss", "wn_backward", "wn_backward_window", "wn_adam_step", "wn_op_front", "wn_op_causal_conv", "wn_op_causal_conv_backward", "wn_op_causal_conv_backward_scratch_floats", "wn_op_upsampling", "wn_op_transpose_last2", "wn_decode_layered_residency", "wn_op_gemm", "wn_prof_enable", "wn_prof_report", "wn_prof_sequence",
    "wn_decode_supported", "wn_decode_pack_floats", "wn_decode_state_floats", "wn_decode_pack", "wn_decode_aux",
    "wn_decode_steps", "wn_decode_stream_bytes",
    "wn_decode_layered_state_floats", "wn_decode_layered_error_offset", "wn_decode_layered_prepare", "wn_decode_layered_steps", "wn_mol_loss",
    "wn_decode_ctx_aux", "wn_decode_prefill_workspace_bytes", "wn_decode_prefill",
]


class WnError(RuntimeError):
    pass


class WnDecodeTimeout(WnError):
    """The persistent decode launch gave up waiting between its workgroups (they were not all resident at once)."""


class WnLibrary(object):
    def __init__(self, path, is_emulator=False):
        self.path = path
        self.is_emulator = is_emulator
        if not is_emulator:
            # PyTorch-ROCm ships its own HIP runtime: it must be in the process BEFORE this library is loaded,
            # otherwise libwavenet_hip.so binds the system runtime, the process ends up with two runtimes and
            # every launch of ours fails with "no ROCm-capable device is detected".
            import torch  # noqa: F401
        self.lib = ctypes.CDLL(path)
        L = self.lib
        vp, i, i64, f, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
        cfgp = ctypes.POINTER(WnConfig)
        L.wn_abi_version.restype = i
        L.wn_last_error.restype = ctypes.c_char_p
        L.wn_receptive_field.argtypes = [cfgp]
        L.wn_num_layers.argtypes = [cfgp]
        L.wn_param_count.argtypes = [cfgp]
        L.wn_param_count.restype = i64
        L.wn_param_offset.argtypes = [cfgp, i, i, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        L.wn_num_buckets.argtypes = [cfgp, i]
        L.wn_bucket_range.argtypes = [cfgp, i, i, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        L.wn_dead_param_range.argtypes = [cfgp, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        L.wn_workspace_bytes.argtypes = [cfgp, i, i]
        L.wn_workspace_bytes.restype = sz
        L.wn_workspace_region.argtypes = [cfgp, i, i, i, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        L.wn_forward.argtypes = [cfgp, i, i, vp, vp, vp, vp, vp, sz, i, vp]
        L.wn_softmax_ce_loss.argtypes = [cfgp, i, i, vp, vp, i, f, f, vp, vp, vp, sz, vp]
        L.wn_forward_loss_fused.argtypes = [cfgp, i, i, i]
        L.wn_forward_loss.argtypes = [cfgp, i, i, vp, vp, vp, vp, i, f, f, vp, vp, vp, vp, sz, i, vp]
        L.wn_backward.argtypes = [cfgp, i, i, vp, vp, vp, vp, vp, vp, sz, ctypes.POINTER(vp), i, i, i, vp]
        L.wn_backward_window.argtypes = [cfgp, i, i, vp, vp, vp, vp, i, vp, vp, sz, ctypes.POINTER(vp), i, i, i, vp]
        L.wn_adam_step.argtypes = [vp, vp, vp, vp, i64, i64, f, f, f, f, f, i64, i64, vp]
        L.wn_op_front.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, vp]
        L.wn_op_causal_conv.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, vp]
        L.wn_op_causal_conv_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, vp]
        L.wn_op_causal_conv_backward_scratch_floats.argtypes = [i, i, i, i, i]
        L.wn_op_causal_conv_backward_scratch_floats.restype = i64
        L.wn_op_upsampling.argtypes = [vp, vp, vp, vp, i, i, i, i, vp]
        L.wn_op_transpose_last2.argtypes = [vp, vp, i, i, i, vp]
        L.wn_op_gemm.argtypes = [ctypes.POINTER(WnGemmArgs), vp]
        L.wn_prof_enable.argtypes = [i]
        L.wn_prof_report.argtypes = [ctypes.c_char_p, sz]
        L.wn_prof_sequence.argtypes = [ctypes.c_char_p, sz]
        L.wn_decode_supported.argtypes = [cfgp]
        L.wn_decode_pack_floats.argtypes = [cfgp]
        L.wn_decode_pack_floats.restype = i64
        L.wn_decode_state_floats.argtypes = [cfgp]
        L.wn_decode_state_floats.restype = i64
        L.wn_decode_stream_bytes.argtypes = [cfgp]
        L.wn_decode_stream_bytes.restype = i64
        L.wn_decode_pack.argtypes = [cfgp, vp, vp, vp]
        L.wn_decode_aux.argtypes = [cfgp, i, i, vp, vp, vp, vp]
        L.wn_decode_steps.argtypes = [cfgp, i, vp, vp, vp, i, i, vp, i64, vp, vp, i, i, vp, vp, vp, i, vp, f, vp]
        L.wn_mol_loss.argtypes = [cfgp, i, i, vp, vp, i, f, f, i, f, vp, vp, vp, sz, vp]
        L.wn_decode_layered_state_floats.argtypes = [cfgp, i, i]
        L.wn_decode_layered_state_floats.restype = i64
        L.wn_decode_layered_error_offset.argtypes = [cfgp, i, i]
        L.wn_decode_layered_error_offset.restype = i64
        L.wn_decode_layered_residency.argtypes = [cfgp, i, i, ctypes.POINTER(i), ctypes.POINTER(i)]
        L.wn_decode_layered_prepare.argtypes = [cfgp, i, i, vp, vp, vp, vp, i64, i, vp]
        L.wn_decode_layered_steps.argtypes = [cfgp, i, vp, vp, i, i, vp, i64, vp, vp, i, i, vp, i64, vp, vp, i, vp, f, vp]
        L.wn_decode_ctx_aux.argtypes = [cfgp, i, i, i, i, i, vp, vp, vp, vp]
        L.wn_decode_prefill_workspace_bytes.argtypes = [cfgp, i, i]
        L.wn_decode_prefill_workspace_bytes.restype = ctypes.c_size_t
        L.wn_decode_prefill.argtypes = [cfgp, i, i, i, vp, vp, vp, vp, ctypes.c_size_t, vp, i64, i, i, i, i, vp]
        if L.wn_abi_version() != ABI_VERSION:
            raise WnError("ABI mismatch: library %d, binding %d" % (L.wn_abi_version(), ABI_VERSION))

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.wn_last_error()
            raise WnError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))

    def __getattr__(self, name):
        return getattr(self.lib, name)


_lock = threading.Lock()
_cached = None


@contextlib.contextmanager
def _build_lock():
    """Inter-process lock around the in-tree build (flock on csrc/.build.lock)."""
    import fcntl
    fh = open(os.path.join(CSRC, ".build.lock"), "w")
    try:
        fcntl.flock(fh, fcntl.LOCK_EX)
        yield
    finally:
        fcntl.flock(fh, fcntl.LOCK_UN)
        fh.close()


def load_library(path=None, _test_emulator=False):
    """Load (building if necessary) the gfx950 library.  Raises if that is impossible."""
    global _cached
    if path is None and os.environ.get("WN_LIB_PATH"):  # tuning experiments: an alternative gfx950 build
        return WnLibrary(os.environ["WN_LIB_PATH"])
    if path is not None:
        return WnLibrary(path, is_emulator=_test_emulator)
    with _lock:
        if _cached is None:
            # Always go through build(): it compares the digest of the sources with the stamp written next to the .so
            # and returns at once when they match, so a stale library beside edited kernels is never loaded silently.
            # It cross-compiles with hipcc (no GPU needed) and raises when hipcc is absent.  The file lock serialises
            # the ranks of a multi-process launch (train.py --n_gpus N) that would otherwise compile into the same paths.
            from .csrc import build as _build
            try:
                # digest first, WITHOUT the lock: a valid prebuilt library loads from a read-only install too (the lock
                # file cannot be created there); only a rebuild takes the lock
                if not _build.up_to_date():
                    with _build_lock():
                        _build.build(verbose=False)
            except Exception as e:  # noqa: BLE001 -- reported with the product's own error type
                raise WnError("libwavenet_hip.so is missing or stale and could not be built (%s): the MI355X HIP "
                              "extension is required (there is no CPU / eager fallback)" % e)
            if not os.path.exists(LIB_PATH):
                raise WnError("libwavenet_hip.so is missing and could not be built: the MI355X HIP "
                              "extension is required (there is no CPU / eager fallback)")
            _cached = WnLibrary(LIB_PATH)
        return _cached
