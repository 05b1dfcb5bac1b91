// wn_fused.h -- fused residual-block kernels for n_resch == 64 (wn_fused.hip).
#pragma once
#include "wn_device.h"

int wn_fused_supported(int R, int K);
long wn_fused_fwd_weight_floats(int R, int K);
long wn_fused_bwd_weight_floats(int R, int K, int S);

// Builds the per-layer LDS weight images from the flat parameter buffer.
int wn_fused_pack_weights(const float* params, long off_dsig_w0, long off_dtanh_w0, long off_res_w0, long layer_step,
                          long off_skip_w0, long skip_step, int L, int R, int K, int S, float* fw_fwd, float* fw_bwd,
                          wn_stream_t st);

// One residual block forward (reference wavenet.py:525-536) for all (b, t):
//   P = sum_tap Wd_tap x[t-(K-1-tap)d] + w[t%U] G[:, t/U] + c ; s = sigmoid(P[:R]) ; g = tanh(P[R:])
//   z = s*g ; x_next = Wres z + b_res + x      (x_next == NULL: dead output of the last layer)
int wn_fused_resblock_fwd(const float* fw, const float* X, const float* G, long g_bstride, const float* upw,
                          const float* cvec, const float* res_bias, float* Xnext, float* S, float* Gt, float* Z, int B,
                          int T, int R, int K, int dilation, int U, int F, wn_stream_t st);

// dZ = Wskip^T dSkip (+ Wres^T dXn) ; dP = [dZ*g*s*(1-s) ; dZ*s*(1-g^2)]
int wn_fused_resblock_bwd_gate(const float* fw, const float* dSk, const float* dXn, const float* S, const float* Gt,
                               float* dP, int B, int T, int R, int Sch, wn_stream_t st);
