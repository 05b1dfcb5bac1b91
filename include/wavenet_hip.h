/* wavenet_hip.h -- C ABI of libwavenet_hip.so (MI355X / gfx950 WaveNet-vocoder training path).
 *
 * Drop-in boundary.  The reference (kan-bayashi/PytorchWaveNetVocoder) has no FFI: its boundary
 * is the Python API of wavenet_vocoder.nets.  This library is what a binding of
 * that API calls instead of torch.nn ops; every entry point names the reference code it replaces.
 *
 * Conventions
 *   - plain C types only: device pointers + sizes; the caller (PyTorch-ROCm, or any HIP program)
 *     owns every buffer, the library allocates no device memory and keeps no state between calls
 *     (one exception: the opt-in overlap modes WN_FLAG_BWD_OVERLAP / WN_FLAG_FWD_OVERLAP create ONE internal
 *     non-blocking stream and a few events per device on first use);
 *   - activations are channel-major fp32 (B, C, T) exactly like the reference's tensors; sample
 *     indices are int64 (torch.long); logits are written physically as (B, Q, T) -- the
 *     reference's `(B, T, Q)` result is the `.transpose(1, 2)` view of that buffer
 *     (wavenet.py:522 does the same);
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream); no hidden synchronisation; re-entrant per stream;
 *   - return value 0 = ok; non-zero = error, text via wn_last_error() (thread local).
 *
 * Parameters live in ONE flat fp32 buffer (and gradients / Adam moments in buffers of the same
 * shape).  The flat order is the order in which the backward pass finishes gradients
 *      [conv_post_2, conv_post_1, skip_1x1.0..L-1] [layer L-1] ... [layer 0] [causal, upsampling]
 * so that data-parallel gradient buckets are contiguous ranges that become ready front to back.
 * Inside the buffer every tensor keeps the reference's own layout (Conv1d weight (Cout,Cin,K)
 * etc.), so nn.Parameter views of it carry the reference's state_dict keys and shapes
 * (wavenet.py:187-210).  wn_param_offset() is the single source of truth for the offsets.
 */
#ifndef WAVENET_HIP_H_
#define WAVENET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#include "wavenet_hip_gemm.h" /* struct WnGemmArgs of wn_op_gemm */

#ifdef __cplusplus
extern "C" {
#endif

#define WN_ABI_VERSION 9

/* Same fields as the constructor WaveNet(n_quantize, n_aux, n_resch, n_skipch, dilation_depth,
 * dilation_repeat, kernel_size, upsampling_factor)  -- reference wavenet.py:172-173. */
typedef struct WnConfig {
    int32_t n_quantize;
    int32_t n_aux;
    int32_t n_resch;
    int32_t n_skipch;
    int32_t dilation_depth;
    int32_t dilation_repeat;
    int32_t kernel_size;
    int32_t upsampling_factor; /* 0 = no upsampling layer (h arrives at sample rate) */
    int32_t out_channels;      /* channels of conv_post_2; 0 = n_quantize (the reference's softmax head).  A mixture-of-
                                * logistics head uses 3 * n_mixture (BASELINE configs[3]); the front end stays the
                                * n_quantize-class one-hot causal conv */
} WnConfig;

/* Tensor kinds for wn_param_offset (reference state_dict key in the comment). */
enum {
    WN_P_CAUSAL_W = 0, /* causal.conv.weight        (R,Q,K)   wavenet.py:189 */
    WN_P_CAUSAL_B,     /* causal.conv.bias          (R)                      */
    WN_P_UP_W,         /* upsampling.conv.weight    (1,1,1,U) wavenet.py:136 */
    WN_P_UP_B,         /* upsampling.conv.bias      (1)                      */
    WN_P_DSIG_W,       /* dil_sigmoid.l.conv.weight (R,R,K)   wavenet.py:201 */
    WN_P_DSIG_B,       /* dil_sigmoid.l.conv.bias   (R)                      */
    WN_P_DTANH_W,      /* dil_tanh.l.conv.weight    (R,R,K)   wavenet.py:202 */
    WN_P_DTANH_B,      /* dil_tanh.l.conv.bias      (R)                      */
    WN_P_ASIG_W,       /* aux_1x1_sigmoid.l.weight  (R,A,1)   wavenet.py:203 */
    WN_P_ASIG_B,       /* aux_1x1_sigmoid.l.bias    (R)                      */
    WN_P_ATANH_W,      /* aux_1x1_tanh.l.weight     (R,A,1)   wavenet.py:204 */
    WN_P_ATANH_B,      /* aux_1x1_tanh.l.bias       (R)                      */
    WN_P_SKIP_W,       /* skip_1x1.l.weight         (S,R,1)   wavenet.py:205 */
    WN_P_SKIP_B,       /* skip_1x1.l.bias           (S)                      */
    WN_P_RES_W,        /* res_1x1.l.weight          (R,R,1)   wavenet.py:206 */
    WN_P_RES_B,        /* res_1x1.l.bias            (R)                      */
    WN_P_POST1_W,      /* conv_post_1.weight        (S,S,1)   wavenet.py:209 */
    WN_P_POST1_B,      /* conv_post_1.bias          (S)                      */
    WN_P_POST2_W,      /* conv_post_2.weight        (Q,S,1)   wavenet.py:210 */
    WN_P_POST2_B,      /* conv_post_2.bias          (Q)                      */
    WN_P_NKINDS
};

/* flags for wn_forward */
#define WN_FLAG_NO_FUSED 1 /* force the any-size layered path even when the fused R=64 kernels apply */
#define WN_FLAG_EXACT_MFMA 2 /* every contraction on the exact f32-input MFMA (default: the skip-sum / post-net
                              * contractions run on the bf16 matrix cores with a 3-way operand split whose six
                              * products reproduce fp32 to round-off; csrc/wn_gemm6.hip) */
/* Opt-in stream-overlap modes (fused kernels' data chain on the caller's stream, independent contractions on an
 * internal low-priority side stream, fork/join with events inside the call; the caller sees plain stream semantics).
 * Measured on MI355X they do not pay today (the persistent chain kernels hold every wave slot, so a
 * concurrent contraction either starves them or waits), hence not the default; per-launch profiling
 * (wn_prof_enable) always runs serially. */
#define WN_FLAG_BWD_OVERLAP 4 /* wn_backward: weight-gradient contractions on the side stream beside the gate'/dX chain;
                               * same kernels, same reduction order: bit-identical to the serial mode for the same
                               * launch-group size */
#define WN_FLAG_AUX_FUSED 32  /* wn_backward (fused split kernels, upsampling_factor % 16 == 0): the gate kernel also writes the
                               * partial sums of the aux-path gradients (frame-rate aux gradient, upsampling weight), a
                               * small kernel finishes them and dP is not re-read by wn_aux_bwd.  What WaveNetEngine passes by
                               * default since round 2 (measured -3 %); a flag of the C ABI because the sums re-associate
                               * (~1e-7 relative) */
#define WN_FLAG_NO_CHAIN 64   /* wn_backward (fused split kernels, kernel_size <= 2): since ABI v4 the data chain runs as ONE launch
                               * per layer (dX_l and gate'_{l-1} fused, the skip part of dZ pre-contracted for all layers by
                               * one matrix-bound launch; csrc/wn_fused.hip k_chain64s).  This flag restores the former
                               * gate' + dX launch pair per layer (kept for A/B measurements and as an independent check) */
#define WN_FLAG_BWD_OVERLAP_HEAD 16 /* with WN_FLAG_BWD_OVERLAP: only the post-net / skip weight gradients run on the side
                               * stream; the per-layer groups stay on the caller's stream */
#define WN_FLAG_FWD_OVERLAP 8 /* wn_forward (fused kernels): the skip-sum contraction is issued in three chunks of layers on
                               * the side stream while the residual stack is still running (the partial sums round
                               * differently from the single contraction: ~1e-7 relative on the logits) */
#define WN_FLAG_WS_FINITE (1 << 16) /* wn_forward_loss (since ABI v7): the caller vouches that the workspace holds only FINITE
                               * values (it was allocated zero-filled, or an earlier full wn_forward of the same shape wrote it).
                               * wn_forward_loss runs the skip-sum / post-net over the loss window only and leaves the columns
                               * of relu(skip) / relu(post1) in front of it untouched; without this flag it zero-fills them
                               * (two small fills per call) so that a later wn_backward with an earlier window start cannot
                               * multiply dlogits == 0 with NaN / Inf bit patterns of uninitialised memory */
#define WN_FLAG_REPACK (1 << 17)    /* wn_backward / wn_backward_window (since ABI v7): re-build the re-laid-out / pre-split weight
                               * sets in the workspace from the `params` given to THIS call before using them (for callers that
                               * modified params after the forward call; costs one pack pass, ~0.05 ms at the BASELINE size) */
#define WN_FLAG_DW_3PRODUCT (1 << 18) /* wn_backward / wn_backward_window (since ABI v8, opt-in): the weight-gradient contractions -- LEAF results:
                               * sums over every position of the minibatch that no other layer consumes -- take three of the six
                               * products of the 3-way operand split (h h + h m + m h; two bf16 pieces per operand: relative error
                               * ~2^-16 per product, random in sign, instead of 2^-24).  Never applied to a contraction whose
                               * output feeds another layer.  Default off: every contraction fp32-equivalent. */
#define WN_FLAG_DW_F16PAIR (1 << 19) /* wn_backward / wn_backward_window (since ABI v8; opt-in at this boundary, where flags = 0 means six
                               * bf16 products everywhere -- the Python engine sets it by default): the weight-gradient contractions split
                               * their operands into TWO fp16 pieces (11 + 11 significand bits) and take the three products
                               * h h + h l + l h on v_mfma_f32_32x32x16_f16: ~2^-22 relative per product -- below the rounding of an
                               * fp32 running sum over a minibatch's positions -- at half the matrix work of the six bf16 products.
                               * fp16 has 5 exponent bits, so the SIZE of the gradient decides the scale: the gradient operand of
                               * every weight-gradient contraction is multiplied by 2^(e + WN_DW_F16_HEADROOM) before the split and
                               * the result by its inverse (activations are taken as they are), where 2^-e bounds max |dlogits|.
                               * Since ABI v9 the library MEASURES that maximum (a bound that is merely safe -- the a-priori
                               * grad_scale / positions of a mean cross-entropy on a well-fitted model, a forgotten exponent -- would
                               * push the scaled operand into fp16's subnormals and silently cost precision):
                               *   WN_FLAG_DW_F16PAIR alone                      one pass over the `dlogits` given to this call (its loss
                               *                                                  window; ~35 us at the BASELINE size) finds the maximum
                               *   | WN_FLAG_DW_F16_AMAX_WS                       the caller vouches that `dlogits` is, unmodified, what the
                               *                                                  last wn_forward_loss / wn_softmax_ce_loss call ON THIS
                               *                                                  WORKSPACE wrote: those calls leave its maximum in the
                               *                                                  workspace (their epilogue computes it for free)
                               *   | WN_FLAG_DW_F16_EXP_VALID | WN_FLAG_DW_F16_EXP(e)   the caller's own promise max |dlogits| <= 2^-e
                               *                                                  (0 <= e <= 63), taken as given (v8 took the exponent
                               *                                                  field without a valid bit: e = 0 was a silent default)
                               * The scale is decided on the device (no host synchronisation).  Values down to 2^-(e+11) keep all 22
                               * bits.  A back-propagated gradient more than 2^(16 - WN_DW_F16_HEADROOM) times the maximum leaves fp16's
                               * range: the launch detects it (non-finite result) and a six-product launch issued right behind every
                               * fp16 launch, which otherwise returns at once, redoes the contraction -- the result is then the
                               * default mode's; the same redo is forced when no usable maximum exists (all-zero or non-finite
                               * gradient, WN_FLAG_DW_F16_AMAX_WS without such a loss call).  Wins over WN_FLAG_DW_3PRODUCT. */
#define WN_FLAG_DW_F16_EXP_SHIFT 20
#define WN_FLAG_DW_F16_EXP(e) (((e) & 63) << WN_FLAG_DW_F16_EXP_SHIFT)
#define WN_DW_F16_HEADROOM 8
#define WN_FLAG_DW_F16_EXP_VALID (1 << 26) /* since ABI v9: the WN_FLAG_DW_F16_EXP(e) field is the caller's promise (e = 0 included) */
#define WN_FLAG_DW_F16_AMAX_WS (1 << 27)   /* since ABI v9: max |dlogits| as the last loss call of this workspace measured it */
#define WN_FLAG_MM_F16PAIR (1 << 28) /* since ABI v9, opt-in: wn_forward / wn_forward_loss / wn_backward(_window) -- the weights x activations
                               * contractions on the split matrix-core kernel (skip sum, post-net, their data gradients, the per-layer
                               * contractions of wide models) take the fp16 pair split too (two fp16 pieces per operand, three products,
                               * ~2^-22 per product: the rounding of an fp32 running sum over >= 64 terms) instead of the six bf16
                               * products.  Activations are taken as they are, back-propagated gradients scaled by the measured maximum
                               * like the weight gradients' (the same three sources, see WN_FLAG_DW_F16PAIR); every such launch is
                               * followed by a conditional six-product launch that redoes it if an operand left fp16's range.  The same
                               * flag must be given to the forward and the backward call of a step (the workspace holds the weight
                               * images the forward call packed), or wn_backward gets WN_FLAG_REPACK. */
#define WN_FLAG_FUSED_F16PAIR (1 << 29) /* since ABI v9, opt-in: wn_forward / wn_forward_loss -- the fused 64-channel residual block (taps, gate, res
                               * 1x1) on the fp16 pair split as well (three products per multiply instead of six), BLOCK-SCALED: every
                               * weight image and every 64 x 32 operand tile is multiplied by the power of two that puts its maximum
                               * at 2^12 / 2^14, so no magnitude can leave fp16's range and no redo is needed (csrc/wn_fused.hip
                               * k_resblock_fwd_h).  Forward only: the backward chain keeps six bf16 products. */
#define WN_FLAG_CHAIN_F16PAIR (1 << 30) /* since ABI v9, opt-in: wn_forward(_loss) packs and wn_backward(_window) uses -- the fused backward data chain
                               * (dX_l from dP_l, gate' of layer l-1) on the block-scaled fp16 pair split (k_chain64s<.., H16>): weight
                               * images by the power of two of their maximum, the dP operand of a tile by the maximum of the producer
                               * tiles it reads (every chain launch records max |dP| per 32-sample tile beside dP), the dX operand by
                               * its own maximum.  Same flag for the forward and the backward call of a step (the images are packed by
                               * the forward call), or WN_FLAG_REPACK. */
#define WN_FLAG_DW_FLUSH(n) (((n) & 0xff) << 8) /* wn_backward: issue the weight gradients of at most n walked layers per
                               * launch group (0 = default: a whole gradient bucket; 5 layers with WN_FLAG_BWD_OVERLAP).
                               * Groups never straddle a bucket.  The split-K plan of a group depends on its size, so
                               * different n round differently (~1e-7 relative) */

int wn_abi_version(void);
const char* wn_last_error(void);

/* receptive_field = (K-1)*sum(dilations)+1, dilations = [2^i for i<depth]*repeat (wavenet.py:184-185) */
int wn_receptive_field(const WnConfig* cfg);
int wn_num_layers(const WnConfig* cfg);

/* Number of fp32 elements of the flat parameter buffer (== sum of numel of the reference state_dict);
 * -1 for an invalid configuration (wn_last_error() says which field). */
int64_t wn_param_count(const WnConfig* cfg);
/* Offset (in floats) and numel of tensor `kind` of layer `layer` (ignored for non per-layer kinds)
 * inside the flat parameter / gradient buffers.  Returns non-zero if the tensor does not exist
 * (upsampling tensors when upsampling_factor == 0). */
int wn_param_offset(const WnConfig* cfg, int kind, int layer, int64_t* offset, int64_t* numel);

/* Gradient buckets for data-parallel all-reduce: bucket 0 = post-net + all skip_1x1; then groups of
 * `layers_per_bucket` residual layers from the last layer down; the final bucket = causal +
 * upsampling.  With ONE layer group (layers_per_bucket <= 0 or >= the number of layers) the skip_1x1
 * tensors belong to that group's bucket instead (bucket 0 = post-net only): their gradients may then
 * be produced behind the data chain, together with the res_1x1 gradients.  The buckets are contiguous,
 * disjoint and cover the buffer; [lo, hi) are float offsets into the flat gradient buffer. */
int wn_num_buckets(const WnConfig* cfg, int layers_per_bucket);
int wn_bucket_range(const WnConfig* cfg, int layers_per_bucket, int bucket, int64_t* lo, int64_t* hi);
/* [lo, hi) of parameters that never receive a gradient (res_1x1 of the last layer; reference:
 * wavenet.py:231-238 leaves its output unused, so torch reports grad None and Adam skips it). */
int wn_dead_param_range(const WnConfig* cfg, int64_t* lo, int64_t* hi);

/* Bytes of caller-provided scratch for batch B x T model inputs (forward + backward); 0 for an invalid
 * configuration or shape (wn_last_error()). */
size_t wn_workspace_bytes(const WnConfig* cfg, int B, int T);

/* Introspection for parity tests (since ABI v4; the training path never calls it): offset and length, in floats, of
 * a tensor that wn_forward / wn_backward leave in the workspace of a (B, T) call.  Layouts (time contiguous):
 *   X, SIGMOID, TANH, Z  (L, B, R, T)  layer input x_l, sigmoid(.) and tanh(.) of the gate, z_l = their product
 *                                      (reference wavenet.py:525-536; x_0 = the front conv's output).  TANH is only
 *                                      written by the any-size kernels: the fused n_resch = 64 kernels save the sigmoid
 *                                      half and z, and rebuild tanh = z / sigmoid in the backward pass
 *   RELU_SKIP, RELU_POST1 (B, S, T)    relu(sum of skips), relu(conv_post_1(.))  (wavenet.py:519-521): their
 *                                      positivity is the ReLU sub-gradient wn_backward uses
 *   DSKIP (B, S, T), DP (L, B, 2R, T), DX (L, B, R, T)   after wn_backward: dL/d(skip sum), dL/d(gate pre-activations
 *                                      [sigmoid rows ; tanh rows]), dL/dx_l */
enum { WN_WS_X = 0, WN_WS_SIGMOID = 1, WN_WS_TANH = 2, WN_WS_Z = 3, WN_WS_RELU_SKIP = 4, WN_WS_RELU_POST1 = 5,
       WN_WS_DSKIP = 6, WN_WS_DP = 7, WN_WS_DX = 8 };
int wn_workspace_region(const WnConfig* cfg, int B, int T, int kind, int64_t* offset_floats, int64_t* n_floats);

/* WaveNet.forward(x, h)  -- reference wavenet.py:212-241 (+ _preprocess :513-516, UpSampling
 * :141-154, _residual_forward :525-536, _postprocess :518-523).
 *   params : flat parameter buffer                         (device, wn_param_count floats)
 *   x      : (B, T) int64 sample indices, taken modulo n_quantize like OneHot (wavenet.py:88)
 *   h      : (B, n_aux, T / upsampling_factor) fp32, or (B, n_aux, T) when upsampling_factor == 0
 *   logits : (B, n_quantize, T) fp32 out  [view it as (B, T, Q) via transpose(1,2)]
 * Everything the backward pass needs is kept in `ws`. */
int wn_forward(const WnConfig* cfg, int B, int T, const float* params, const int64_t* x, const float* h,
               float* logits, void* ws, size_t ws_bytes, int flags, void* stream);

/* nn.CrossEntropyLoss()(out[:, t_start:].reshape(-1,Q), target[:, t_start:].reshape(-1))
 *   -- reference train.py:461,534-536 (t_start = receptive_field there).
 *   loss     : device scalar out = mean over B*(T-t_start) positions, times loss_scale
 *   dlogits  : (B, Q, T) out = d(mean loss)/d(logits) * grad_scale  (zeros for t < t_start); may be NULL */
int wn_softmax_ce_loss(const WnConfig* cfg, int B, int T, const float* logits, const int64_t* target, int t_start,
                       float grad_scale, float loss_scale, float* loss, float* dlogits, void* ws, size_t ws_bytes,
                       void* stream);

/* wn_forward + wn_softmax_ce_loss in one call, for the training step (since ABI v5; reference train.py:533-536:
 * batch_output = model(x, h); loss = CrossEntropyLoss()(batch_output[:, rf:], batch_t[:, rf:])).  When
 * wn_forward_loss_fused(cfg, B, T, flags) == 1 the loss is the EPILOGUE of the conv_post_2 contraction: a workgroup holds all
 * n_quantize <= 256 classes of its 128 positions on chip, so the (B, Q, T) logits are never written to or read back from
 * memory -- `loss` and `dlogits` (nullable) come out exactly as wn_softmax_ce_loss defines them, logits_scratch is not
 * touched (may be NULL).  Otherwise (exact-MFMA mode, more than 256 classes, mixture head) the call runs the two entry
 * points back to back and needs logits_scratch (B, Q, T).
 * Workspace after the call: as wn_forward leaves it, EXCEPT that the fused form computes the skip sum and the post-net over
 * the loss window only -- columns [t0, T) with t0 = t_start rounded down to a multiple of 128: relu(skip) and relu(post1)
 * (WN_WS_RELU_SKIP / WN_WS_RELU_POST1) are valid from t0 on; in front of t0 they hold zeros (or, with WN_FLAG_WS_FINITE,
 * whatever finite values were there).  dlogits is exactly zero there, so a backward pass over ANY window start t_first <= t_start
 * (wn_backward included) yields the same gradients; wn_backward_window(t_first = t_start) is the cheapest. */
int wn_forward_loss_fused(const WnConfig* cfg, int B, int T, int flags);
int wn_forward_loss(const WnConfig* cfg, int B, int T, const float* params, const int64_t* x, const float* h,
                    const int64_t* target, int t_start, float grad_scale, float loss_scale, float* loss, float* dlogits,
                    float* logits_scratch, void* ws, size_t ws_bytes, int flags, void* stream);

/* Backward of wn_forward (what autograd does for train.py:538): writes EVERY element of the flat
 * gradient buffer `grads` (the dead range gets zeros).  `ws` must still hold the matching
 * wn_forward call, made with the same WN_FLAG_NO_FUSED / WN_FLAG_EXACT_MFMA choice (the two kernel
 * families save different activations), and `params` must be UNCHANGED since that call: the workspace also holds the
 * re-laid-out / pre-split weight sets wn_forward packed from them (the post-net and skip weights of the backward
 * contractions among them), and wn_backward does not re-pack unless WN_FLAG_REPACK is given.  The training loop satisfies
 * this by construction (the optimizer step comes after the backward pass, train.py:533-539); the Python engine tracks a
 * parameter version and raises, like torch.autograd does for a tensor modified in place between forward and backward.  If events != NULL, hipEvent_t events[i] is recorded on
 * `stream` as soon as bucket i (wn_bucket_range) is final, so the caller can all-reduce it on another stream. */
int wn_backward(const WnConfig* cfg, int B, int T, const float* params, const int64_t* x, const float* h,
                const float* dlogits, float* grads, void* ws, size_t ws_bytes, void* const* events, int n_events,
                int layers_per_bucket, int flags, void* stream);

/* wn_backward for a loss that covers positions [t_first, T) only (since ABI v5) -- the reference's training loss,
 * train.py:534-536: CrossEntropyLoss on batch_output[:, receptive_field:].  The caller guarantees
 * dlogits[:, :, :t_first] == 0 (wn_softmax_ce_loss / wn_mol_loss with t_start = t_first write exactly that).  Between the
 * logits and the residual stack everything is pointwise in time (wavenet.py:518-523,533), so dO2, dSkip and the skip part
 * of every layer's dZ are zero in front of t_first as well: their contractions and the post-net / skip weight gradients
 * run over the window only (from t_first rounded down to a multiple of 128; the skipped columns of dSkip are zero-filled
 * for the residual chain, which needs every position).  Same gradients as wn_backward up to the rounding of a different
 * split-K plan; t_first = 0 IS wn_backward. */
int wn_backward_window(const WnConfig* cfg, int B, int T, const float* params, const int64_t* x, const float* h,
                       const float* dlogits, int t_first, float* grads, void* ws, size_t ws_bytes, void* const* events,
                       int n_events, int layers_per_bucket, int flags, void* stream);

/* torch.optim.Adam step over the flat buffers (reference train.py:457-460,539): L2-in-gradient
 * weight decay, bias correction with `step` (1-based); [skip_lo, skip_hi) is left untouched. */
int wn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                 float lr, float beta1, float beta2, float eps, float weight_decay, int64_t skip_lo, int64_t skip_hi,
                 void* stream);

/* ---- op-level entry points (used by the composite calls above; exported for parity tests) ---- */

/* OneHot + CausalConv1d(Q->R,K) as a gather (wavenet.py:78-92,513-516).  weight (R,Q,K), bias (R). */
int wn_op_front(const float* weight, const float* bias, const int64_t* x, float* out /*(B,R,T)*/, float* scratch /*K*Q*R*/,
                int B, int T, int Q, int R, int K, void* stream);

/* CausalConv1d forward (wavenet.py:95-121): y[b,:,t] = bias + sum_k W[:,:,k] x[b,:,t-(K-1-k)d].
 * weight (Cout,Cin,K) natural layout; scratch >= Cout*Cin*K floats. */
int wn_op_causal_conv(const float* weight, const float* bias, const float* x /*(B,Cin,T)*/, float* y /*(B,Cout,T)*/,
                      float* scratch, int B, int T, int Cin, int Cout, int K, int dilation, void* stream);

/* Backward of CausalConv1d (since ABI v9; the reference's module is an ordinary differentiable nn.Module, wavenet.py:95-121 --
 * autograd of Conv1d + slice): dx (B,Cin,T), dw (Cout,Cin,K), db (Cout) from dy (B,Cout,T); any of the three may be NULL.
 * Exact f32 matrix-core contractions, weight gradient as per-(sequence, time-chunk) partials summed in a fixed order.
 * scratch >= wn_op_causal_conv_backward_scratch_floats(B, T, Cin, Cout, K) floats. */
long wn_op_causal_conv_backward_scratch_floats(int B, int T, int Cin, int Cout, int K);
int wn_op_causal_conv_backward(const float* weight, const float* x, const float* dy, float* dx, float* dw, float* db,
                               float* scratch, int B, int T, int Cin, int Cout, int K, int dilation, void* stream);

/* UpSampling.forward (wavenet.py:124-154, since ABI v7): y (B, C, F*U) = x (B, C, F) through the (1, U) transposed
 * convolution with ONE kernel `weight` [U] and scalar `bias` (nullable) shared by all channels. */
int wn_op_upsampling(const float* weight, const float* bias, const float* x, float* y, int B, int C, int F, int U, void* stream);
/* dst (B, C, R) = src (B, R, C) with the last two axes swapped (LDS-tiled): the reference's logits are (B, T, n_quantize)
 * (wavenet.py:522), the kernels' (B, n_quantize, T) -- used for a gradient handed back by an external loss (train.py:534-538
 * through autograd).  src != dst. */
int wn_op_transpose_last2(const float* src, float* dst, int B, int R, int C, void* stream);

/* Generic C[z] = A.B contraction on the f32 matrix cores; argument block: wavenet_hip_gemm.h (struct WnGemmArgs,
 * the inline helper wn_gemm_default fills the neutral values). */
int wn_op_gemm(const struct WnGemmArgs* args, void* stream);

/* Mixture-of-logistics output head (BASELINE configs[3]).  NOT part of the reference (its WaveNet only has the
 * softmax head, wavenet.py:209-210,518-523): parity is therefore pinned to this repo's own CPU restatement of
 * the discretised mixture of logistics (PixelCNN++, Salimans et al. 2017) in oracle/, not to the reference.
 * out (B, out_channels = 3*n_mix, T) from wn_forward; y (B, T) fp32 target waveform in [-1, 1];
 * loss = mean negative log-likelihood over t >= t_start; dout goes to wn_backward as `dlogits`. */
int wn_mol_loss(const WnConfig* cfg, int B, int T, const float* out, const float* y, int t_start, float grad_scale,
                float loss_scale, int num_classes, float log_scale_min, float* loss, float* dout, void* workspace,
                size_t workspace_bytes, void* stream);

/* ---- autoregressive decode (BASELINE config 5) -------------------------------------------------
 * Replaces WaveNet.fast_generate / batch_fast_generate / _generate_residual_forward
 * (wavenet_vocoder/nets/wavenet.py:309-395, 397-511, 538-549): one persistent workgroup per
 * utterance, the dilation queues in a caller-owned state buffer, tokens chosen on-chip.
 *
 * Positions index the left-padded token buffer `samples` (B, Ttot) (wavenet.py:331-336: the
 * context is padded with n_quantize/2 up to the receptive field, the aux features by replicating
 * their first column: pass the pad length as n_pad).  Step p reads the tokens at p-K+1..p and the
 * aux column of position p and produces the logits of position p+1.  Positions below t_forced[b]
 * are the context (teacher forced; running them from step 0 over a zeroed state IS the reference's
 * "prepare buffer" pass, wavenet.py:338-349); from t_forced[b] on, step p writes samples[b][p+1].
 * Utterance b stops at t_end[b] positions.  All pointers are device pointers; nothing is allocated;
 * asynchronous on `stream`.  The compiled kernel classes cover n_resch <= 64, n_skipch <= 256,
 * n_quantize <= 256, kernel_size <= 3: wn_decode_supported() tells, everything else returns an
 * error (callers fall back to full-window forwards, wavenet.py:243-307). */
int wn_decode_supported(const WnConfig* cfg);
int64_t wn_decode_pack_floats(const WnConfig* cfg);   /* floats of the packed decode weights, <0: unsupported */
int64_t wn_decode_state_floats(const WnConfig* cfg);  /* floats of queue state per utterance */
int64_t wn_decode_stream_bytes(const WnConfig* cfg);  /* weight bytes one workgroup streams per step */
/* Re-pack the flat parameter buffer into the per-thread weight stream + side tables. */
int wn_decode_pack(const WnConfig* cfg, const float* params, float* wpack, void* stream);
/* Aux projections of all layers at the aux rate: G[b][f][l*2R+o'] = Waux_l h[b][:, f] (wavenet.py:541-542).
 * h is (B, n_aux, F): frames if upsampling_factor > 0 (the kernel applies the transposed-conv taps
 * of wavenet.py:136 per sample), samples otherwise. */
int wn_decode_aux(const WnConfig* cfg, int B, int F, const float* wpack, const float* h, float* G, void* stream);
/* mode: 0 argmax, 1 categorical sampling with the caller's uniform draws `uniforms` (B, Ttot)
 * (draw [b][p+1] picks the token of position p+1), 2 mixture-of-logistics head (out_channels = 3*n_mix, n_mix <= 64;
 * not in the reference): uniforms is (B, Ttot, n_mix+1), the drawn value goes to wave_out (B, Ttot) (nullable) and,
 * mu-law encoded with n_quantize levels, to samples.  logits_out (B, Ttot, out_channels) is optional (row p = the
 * network output computed by step p).  `state` (B, wn_decode_state_floats) must be zero before step 0. */
int wn_decode_steps(const WnConfig* cfg, int B, const float* params, const float* wpack, const float* G, int F, int n_pad,
                    int64_t* samples, int64_t Ttot, const int32_t* t_forced, const int32_t* t_end, int p0, int p1,
                    float* state, const float* uniforms, float* logits_out, int mode, float* wave_out, float log_scale_min,
                    void* stream);

/* Any-size variant of the same decode (layer-wise launches of the contraction kernels on [channels x B]
 * operands: one pass over the weights per step serves the whole batch).  Same positions / teacher forcing
 * / mode / uniforms / logits_out conventions as wn_decode_steps.  `state` is ONE caller-owned buffer of
 * wn_decode_layered_state_floats(cfg, B, mode) floats, zero-filled before wn_decode_layered_prepare, which packs
 * the weights into it and computes G (B, F, L*2R) from h (B, n_aux, F).  A later wn_decode_layered_prepare on the
 * same state with params == NULL keeps the packed weights and only projects the given window of h (a decode
 * without an upsampling layer projects one window of aux columns per chunk of steps). */
int64_t wn_decode_layered_state_floats(const WnConfig* cfg, int B, int mode);
/* Since ABI v7: for models the plan of csrc/wn_dlp.h covers (n_resch % 32 == 0, kernel_size 2 or 3, B <= 64 (48 with the granule hand-off), softmax head)
 * wn_decode_layered_steps runs the whole range of steps as ONE launch of workgroups that hand their vectors to each other as
 * 8-byte {value, tag} granules or as plain vectors + one flag per workgroup and stage (the recipes' n_resch = 512 model: 66
 * dependent launches per step before): n_resch / 4 workgroups with fp32 VALU dot products for one utterance (wn_dlp.hip),
 * n_resch / 8 workgroups per block of 16 utterances with v_mfma_f32_16x16x4_f32 tiles up to 64 (wn_dlpf.hip; wn_dlpm.hip: 48).
 * Every workgroup of such a launch waits for the others, so ALL of them must be resident at once: the library asks the device
 * (occupancy of the chosen kernel x compute units, since ABI v8) when it chooses the path, and a grid that does not fit -- a
 * partitioned GPU, a part with fewer CUs -- decodes by layer-wise launches; wn_decode_layered_residency() reports the numbers.
 * Mode bits (the SAME bits go to every call of one decode -- state_floats / error_offset / prepare / steps, and OR-ed into
 * `layered` of wn_decode_prefill -- because the layout of `state` depends on them):
 *   WN_DECODE_BY_LAUNCHES  wn_decode_layered_steps keeps the layer-wise launches (independent check, A/B, fall-back);
 *   WN_DECODE_GRANULES     (since ABI v8; replaces the process-wide wn_decode_set_handoff of v7) the persistent launches hand
 *                          their vectors over as 8-byte granules everywhere instead of plain vectors + one flag per workgroup
 *                          and stage where wn_dlpf.hip covers the plan (A/B, tests).
 * The persistent launch bounds every wait; wn_decode_layered_error_offset() is the float offset in `state` of an int that is
 * non-zero afterwards if a wait timed out (-1: this model / B / device decodes by launches).  A launch that finds the word
 * non-zero returns at once, so a caller may check it after any chunk of steps. */
#define WN_DECODE_BY_LAUNCHES 256
#define WN_DECODE_GRANULES 512
int64_t wn_decode_layered_error_offset(const WnConfig* cfg, int B, int mode);
/* 1: wn_decode_layered_steps(cfg, B, mode) runs as one persistent launch on the current device, 0: as layer-wise launches;
 * *workgroups (nullable) = the grid the plan asks for (0: no plan covers this model / B), *capacity (nullable) = workgroups of
 * that kernel the device keeps resident at once.  < 0: bad argument. */
int wn_decode_layered_residency(const WnConfig* cfg, int B, int mode, int* workgroups, int* capacity);
int wn_decode_layered_prepare(const WnConfig* cfg, int B, int F, const float* params, const float* h, float* G, float* state,
                              int64_t state_floats, int mode, void* stream);
int wn_decode_layered_steps(const WnConfig* cfg, int B, const float* params, const float* G, int F, int n_pad,
                            int64_t* samples, int64_t Ttot, const int32_t* t_forced, const int32_t* t_end, int p0, int p1,
                            float* state, int64_t state_floats, const float* uniforms, float* logits_out, int mode,
                            float* wave_out, float log_scale_min, void* stream);
/* mode 2 (mixture-of-logistics head, out_channels = 3*n_mix): uniforms is (B, Ttot, n_mix+1); the drawn value is
 * written to wave_out (B, Ttot) (nullable) and, mu-law encoded with n_quantize levels, to samples.  log_scale_min
 * (since ABI v4; ignored by modes 0/1) is the clamp of the log-scales -- pass the value the model was trained with
 * (wn_mol_loss's log_scale_min) so that sampling and the likelihood agree. */

/* Parallel context walk.  The reference builds the generation buffers with ONE full forward over the padded
 * context (wavenet.py:338-349, 427-441); stepping the decode kernel through those >= receptive-field positions
 * instead costs rf steps before the first new sample.  These three entry points do what the reference does:
 *   wn_decode_ctx_aux   h (B, n_aux, F) -> h_ctx (B, n_aux, Tctx): the UPSAMPLED aux feature of every context
 *                       position (ConvTranspose2d of wavenet.py:141-154), the n_pad left-padding positions
 *                       replicating the first upsampled column (wavenet.py:336, 425);
 *   wn_decode_prefill   runs the residual stack of the training forward on x_ctx (B, Tctx) / h_ctx and copies
 *                       the newest (K-1)*d_l layer inputs of every layer into the dilation queues of `state`
 *                       (layered = 0: the (B, wn_decode_state_floats) state of wn_decode_steps; 1 [| WN_DECODE_GRANULES
 *                       when the decode runs with that bit]: the state of wn_decode_layered_steps, after
 *                       wn_decode_layered_prepare).  The B utterances of the call
 *                       are utterances [state_b0, state_b0 + B) of a state built for state_B utterances, so a
 *                       large batch can be walked in groups with a bounded workspace.  Decoding then resumes
 *                       with p0 = Tctx-1, the last context position, whose logits choose the first new sample.
 *                       `ws`: wn_decode_prefill_workspace_bytes(cfg, B, Tctx) bytes; flags as wn_forward.
 * Tctx >= receptive field (the caller pads: wavenet.py:328-336).  Only the newest receptive field + kernel_size - 1
 * positions of a context reach the queues (the dilated stack plus the causal front conv), so a long context may be
 * passed as its tail of at least that many positions: x_ctx / h_ctx then hold positions [pos0, pos0 + Tctx) of
 * the padded context and decoding resumes with p0 = pos0 + Tctx - 1. */
int wn_decode_ctx_aux(const WnConfig* cfg, int B, int F, int Tctx, int n_pad, int pos0, const float* params, const float* h,
                      float* h_ctx, void* stream);
size_t wn_decode_prefill_workspace_bytes(const WnConfig* cfg, int B, int Tctx);
int wn_decode_prefill(const WnConfig* cfg, int B, int Tctx, int pos0, const float* params, const int64_t* x_ctx,
                      const float* h_ctx,
                      void* ws, size_t ws_bytes, float* state, int64_t state_floats, int state_B, int state_b0, int layered,
                      int flags, void* stream);

/* ---- diagnostics: opt-in per-launch timing with HIP events (used by bench.py's roofline block) ----
 * wn_prof_enable(1) clears and starts recording {kernel tag, algorithmic flops/bytes, start/stop
 * event} for every launch; after synchronising, wn_prof_report writes a JSON object
 * {"tag": {"count", "ms", "flops", "bytes"}} into buf (returns the needed size when buf == NULL). */
int wn_prof_enable(int on);
int wn_prof_report(char* buf, size_t n);
/* the recorded tags in issue order, comma separated; "bucket_event" marks where wn_backward recorded a gradient-bucket
 * event (tests assert that a bucket's event follows the last launch that writes into the bucket).  Same buffer protocol. */
int wn_prof_sequence(char* buf, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* WAVENET_HIP_H_ */
