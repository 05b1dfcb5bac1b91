// wn_fused.h -- fused residual-block kernels for n_resch == 64 (wn_fused.hip).
#pragma once
#include "wn_device.h"

// The fused kernels need R == 64 (two 32-row MFMA tiles per gate half), S % 32 == 0 and weights
// that fit the 160 KB LDS of a CU.
int wn_fused_supported(int R, int K, int S);

// One residual block forward (reference wavenet.py:525-536) for all (b, t):
//   P = sum_tap Wd_tap x[t-(K-1-tap)d] + w[t%U] G[:, t/U] + c ; s = sigmoid(P[:R]) ; g = tanh(P[R:])
//   z = s*g ; x_next = Wres z + b_res + x      (x_next == NULL: dead output of the last layer)
// wd_f  : [(tap*R + i)*2R + o']   packed dilated weights (sigmoid rows then tanh rows)
// wres_f: [i*R + o]               packed (transposed) res_1x1 weight
int wn_fused_resblock_fwd(const float* wd_f, const float* wres_f, const float* cvec, const float* res_bias,
                          const float* X, const float* G, long g_bstride, const float* upw, float* Xnext, float* S,
                          float* Gt, float* Z, int B, int T, int K, int dilation, int U, int F, int split, const float* wimg,
                          wn_stream_t st);
// wimg (split kernels, K <= 2; nullable): this layer's pre-built LDS weight image, see wn_fused_pack_images

// dZ = Wskip^T dSkip (+ Wres^T dXn) ; dP = [dZ*g*s*(1-s) ; dZ*s*(1-g^2)]
// wskip : natural skip_1x1 weight [S][R] ; wres : natural res_1x1 weight [R][R] ; dXn may be NULL.
// gt_is_z != 0 (all three gate' launchers): `Gt` holds the saved product z = s * tanh instead of the tanh half, which the
// fused forward then does not store at all (Gt == NULL there); the kernels rebuild g = z / s (s = sigmoid > 0).
int wn_fused_bwd_gate(const float* wskip, const float* wres, const float* dSk, const float* dXn, const float* S,
                      const float* Gt, int gt_is_z, float* dP, int B, int T, int Sch, int split, wn_stream_t st);

// The same plus the partial sums of the aux-path gradients (split kernels only; U % 16 == 0, T == U * F), so that dP
// is not re-read for them (wn_aux_bwd):
//   dGp[b][row][h] = sum_{t in [16h, 16h+16)} w[t % U] dP[b][row][t]        (a frame is U/16 consecutive groups)
//   qp[b][t]       = sum_row dP[b][row][t] * G[b][row][t / U]               (-> d up_w[t % U] after summing over b, frames)
// A tile's dP sits in the accumulator layout (lane = time): 16-sample groups are DPP rows, and because U % 16 == 0 a
// frame boundary never cuts one.  wn_aux_finish (wn_elem.h) turns the partials into what wn_aux_bwd produces.
int wn_fused_bwd_gate_aux(const float* wskip, const float* wres, const float* dSk, const float* dXn, const float* S,
                          const float* Gt, int gt_is_z, float* dP, const float* G, long g_bstride, const float* upw, int U, int F,
                          float* dGp, float* qp, int B, int T, int Sch, wn_stream_t st);

// dX[t] = (dXn[t]) + sum_tap Wd_tap^T dP[t + (K-1-tap) d]
// wd_b : [(tap*2R + o')*R + i] packed weights ; dXn may be NULL.
// split != 0: bf16 matrix cores with the 3-way operand split (fp32-equivalent), else the exact f32 MFMA
int wn_fused_bwd_dx(const float* wd_b, const float* dP, const float* dXn, float* dX, int B, int T, int K, int dilation,
                    int split, wn_stream_t st);

// One launch per layer of the backward data chain (split arithmetic, K <= 2; wn_fused_chain_supported):
//   dX_l = (dXn) + sum_tap Wd_tap^T dP_l[t + (K-1-tap) d]           -> dX
//   dZ_{l-1} = dZs_{l-1} + Wres_{l-1}^T dX_l ; dP_{l-1} = gate'(dZ_{l-1}; S, Gt of layer l-1)   -> dP_prev
// dZs (row stride T, batch stride zs_bstride) is the pre-contracted skip part Wskip_{l-1}^T dSkip.  dX_l goes from the
// accumulators of the first half straight into the MFMAs of the second.  dGp != NULL adds the aux-gradient partial sums
// of wn_fused_bwd_gate_aux for layer l-1 (G = that layer's rows of the frame-rate projection).
int wn_fused_chain_supported(int R, int K, int S);
int wn_fused_bwd_chain(const float* wd_b, const float* dP, const float* dXn, float* dX, const float* wres_prev, const float* dZs,
                       long zs_bstride, const float* S, const float* Gt, int gt_is_z, float* dP_prev, const float* G, long g_bstride,
                       const float* upw, int U, int F, float* dGp, float* qp, int B, int T, int K, int dilation,
                       const float* img_taps, const float* img_res, int zs_t0, const float* amaxP, float* amaxPm, wn_stream_t st);
// zs_t0: dZs[.., t < zs_t0] is taken as zero and never read (loss window, wn_backward_window)
// amaxP != NULL: the fp16 pair split (k_chain64s<.., H16>): img_taps / img_res are the two-piece images of wn_fused_pack_images16,
// amaxP (B, ceil(T/32)) holds max |dP| per 32-sample tile as the launch that wrote dP left it, amaxPm receives it for dP_prev
// top of the chain: dP_{L-1} = gate'(dZs_{L-1}) alone (the last layer has no residual gradient); same aux outputs
int wn_fused_bwd_chain_head(const float* dZs, long zs_bstride, const float* S, const float* Gt, int gt_is_z, float* dP_prev,
                            const float* G, long g_bstride, const float* upw, int U, int F, float* dGp, float* qp, int B, int T,
                            int zs_t0, float* amaxPm, wn_stream_t st);

// LDS weight images of the split kernels, built ONCE per step for all L layers instead of once per workgroup per launch:
//   which = 0  forward block of layer l (taps + res 1x1)      from wd_f, wres_f
//           1  tap weights of the chain kernel's dX half       from wd_b
//           2  Wres_l^T of the chain kernel's gate half        from the natural res_1x1 weight in the flat parameter buffer
// wn_fused_image_floats: floats of region `which` for L layers (0 when K is outside the split kernels' range).
// (img_taps of layer l and img_res of layer l-1 belong to the chain launch of layer l.)
long wn_fused_image_floats(int K, int L, int which);
// Two-piece fp16 images of the forward block (k_resblock_fwd_h: `split` = 2 in wn_fused_resblock_fwd, `wimg` = layer l's image):
// [K*4 blocks][2 pieces][128 rows][16 k] taps, [4][2][64][16] res 1x1, block-scaled by powers of two, inverse scales in the tail
// which = 0 forward block, 1 chain taps, 2 chain Wres^T (k_chain64s<.., H16>); img_taps16 == NULL: the forward image only
long wn_fused_image16_floats(int K, int L, int which);
int wn_fused_pack_images16(const float* wd_f, const float* wres_f, const float* wd_b, const float* params, long res_off,
                           long res_lstride, float* img_fwd16, float* img_taps16, float* img_res16, int K, int L, wn_stream_t st);
int wn_fused_pack_images(const float* wd_f, const float* wres_f, const float* wd_b, const float* params, long res_off,
                         long res_lstride, float* img_fwd, float* img_taps, float* img_res, int K, int L, wn_stream_t st);
