// wn_gemm6.h -- forward-type contraction C = W . X on the bf16 matrix cores with fp32-equivalent
// accuracy (wn_gemm6.hip): every fp32 operand is split into three bf16 pieces (hi, mid, lo; 24
// significand bits) and the six largest cross products are accumulated in fp32.
#pragma once
#include "wn_device.h"

#define WN_G6_BM 256
#define WN_G6_BN 128

typedef struct WnGemm6Args {
    int M, N, K;
    const unsigned short* Apk;  // split weights [(K+15)/16][3][Mpad][16] bf16, Mpad = roundup(M, 256)
    int Mpad;
    const float* B;             // B[k][n]: segment = k / b_seg_len (base seg*b_seg_stride), row (k % b_seg_len)*ldb
    long ldb;
    long b_zstride;
    int b_seg_len;              // multiple of 16, or >= K
    long b_seg_stride;
    int b_shift0, b_shift_step; // element (k, n) reads column n - (b_shift0 + seg*b_shift_step), 0 outside [0, b_clen)
    int b_clen;
    float* C;                   // C[z][m][n]
    long ldc;
    long c_zstride;
    const float* bias;          // [M] or null
    const float* E;             // mask source: result *= (E > 0), indexed like C, or null
    long lde;
    long e_zstride;
    const float* D;             // residual added before relu/mask, indexed like C, or null
    long ldd;
    long d_zstride;
    int relu;
    int accumulate;             // C += result
    int nbatch;
    const char* tag;
    int no_interior;            // set by wn_gemm6_launch (tuning knob WN_G6_INTERIOR=0)
} WnGemm6Args;

static inline long wn_gemm6_apk_elems(int M, int K) {
    const long Mpad = ((long)M + WN_G6_BM - 1) / WN_G6_BM * WN_G6_BM;
    return ((long)K + 15) / 16 * 3 * Mpad * 16;
}
// src holds A(m,k) = src[k*lda + m] (fp32) -> Apk
int wn_gemm6_pack(const float* src, long lda, int M, int K, unsigned short* Apk, wn_stream_t st);
int wn_gemm6_launch(const WnGemm6Args* g, wn_stream_t st);

// Weight-gradient type contraction (both operands k-major, k = time; csrc/wn_gemm.h semantics of the
// a_kmajor = b_kmajor = 1 mode incl. segments, shifts, split-K, layers, a_rowsum) on the same 3-way split.
struct WnGemmArgs;
int wn_gemm6_dw_eligible(const struct WnGemmArgs* g);
int wn_gemm6_dw_tall(int M, int N);   // 256 x 128 tiles for this shape (split path)
int wn_gemm6_dw_launch(const struct WnGemmArgs* g, wn_stream_t st);
