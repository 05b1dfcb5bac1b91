// wn_gemm6.h -- forward-type contraction C = W . X on the bf16 matrix cores with fp32-equivalent
// accuracy (wn_gemm6.hip): every fp32 operand is split into three bf16 pieces (hi, mid, lo; 24
// significand bits) and the six largest cross products are accumulated in fp32.
#pragma once
#include "wn_device.h"

#define WN_G6_F16_WSCALE 64   // power of two the fp16 pair images of the weights are scaled by (wn_gemm6_pack_batch, f16 jobs)
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
    int n_phase;                // parity of the 128-column tile that column 0 of this launch is in the caller's full tensor: the sign
                                // of a column tile's arithmetic (k_gemm6) then does not depend on where a column WINDOW starts
    int stagger;                // set by wn_gemm6_launch (WN_G6_STAGGER): head start, in s_sleep(127) units, of the first block of a CU over its co-resident
    // Gate epilogues of the any-size residual block (R % 128 == 0): the contraction's result never goes to memory.
    // gate_S != NULL (forward, reference wavenet.py:529-532): M = 2R rows packed with wn_gemm6_pack(..., gate_R = R), so that
    //   a lane holds the sigmoid and the tanh pre-activation of the same channel; P = acc + w[t%U] G[:, t/U] + cvec;
    //   S = sigmoid(P[:R]), Gt = tanh(P[R:]), Z = S * Gt are written as (B, R, T); C is not touched.
    // gbw_dP != NULL (backward): M = R rows of dZ (+ the previous C when `accumulate`); dP = [dZ g s (1-s) ; dZ s (1-g^2)]
    //   is written as (B, 2R, T) from the saved halves gbw_S, gbw_Gt (B, R, T); C is read (accumulate) but not written.
    int gate_R;
    float* gate_S;
    float* gate_Gt;
    float* gate_Z;
    const float* gate_G;        // frame-rate aux projection of this layer: rows [0, 2R), row stride gate_F, batch stride gate_gb
    long gate_gb;
    int gate_F, gate_U;
    const float* gate_upw;      // [U]
    const float* gate_cvec;     // [2R]
    const float* gbw_S;
    const float* gbw_Gt;
    float* gbw_dP;
    // Softmax cross-entropy epilogue (ce_target != NULL; the post-net's last contraction, reference wavenet.py:522 +
    // train.py:461,534-536): the block holds ALL M <= 256 rows (classes) of its 128 columns (positions), so
    //   logit = acc + bias never goes to memory; C (nullable) receives d(mean loss)/d(logit) * ce_gs = (softmax - onehot) * ce_gs
    //   for columns >= ce_t_start and 0 in front of them; ce_partial[z * gridDim.x + x] = sum over the block's columns
    //   >= ce_t_start of (logsumexp - logit[target]).  target[z * ce_tstride + column] is taken modulo M.
    const long long* ce_target;
    long ce_tstride;
    int ce_t_start;
    float ce_gs;
    float* ce_partial;
    float* ce_amax;             // nullable, indexed like ce_partial: the block's max |d(mean loss)/d(logit) * ce_gs|
    // fp16 pair split (k_gemm6<.., F16>): f16 != 0 -> Apk is the two-piece fp16 image (wn_gemm6_pack_batch, f16 job), B is
    // multiplied by b_mul at the split (> 0: that power of two; < 0: the one at ((const float*)ovf)[1]) and the accumulators by
    // its inverse, *ovf := 1 when a block's accumulators are not finite.  f16 == 0 with ovf != NULL: the conditional six-product
    // redo behind such a launch (returns at once unless *ovf != 0).
    int f16;
    float b_mul;
    int* ovf;
} WnGemm6Args;

static inline void wn_gemm6_no_gate(WnGemm6Args* a) {
    a->gate_R = 0; a->gate_S = 0; a->gate_Gt = 0; a->gate_Z = 0; a->gate_G = 0; a->gate_gb = 0; a->gate_F = 0; a->gate_U = 1;
    a->gate_upw = 0; a->gate_cvec = 0; a->gbw_S = 0; a->gbw_Gt = 0; a->gbw_dP = 0; a->no_interior = 0; a->stagger = 0; a->n_phase = 0;
    a->ce_target = 0; a->ce_tstride = 0; a->ce_t_start = 0; a->ce_gs = 0.f; a->ce_partial = 0; a->ce_amax = 0;
    a->f16 = 0; a->b_mul = 0.f; a->ovf = 0;
}

static inline long wn_gemm6_apk_elems(int M, int K) {
    const long Mpad = ((long)M + WN_G6_BM - 1) / WN_G6_BM * WN_G6_BM;
    return ((long)K + 15) / 16 * 3 * Mpad * 16;
}
// src holds A(m,k) = src[k*lda + m] (fp32) -> Apk.  gate_R > 0 (M = 2 gate_R, gate_R % 128 == 0): packed row p takes
// source row wn_gemm6_gate_row(p, gate_R) -- every 256-row block holds 128 channels, each wave's 128 rows = 64 sigmoid
// rows followed by the 64 tanh rows of the same channels (the pairing of the forward gate epilogue).
int wn_gemm6_pack(const float* src, long lda, int M, int K, unsigned short* Apk, int gate_R, wn_stream_t st);
// several weight sets in ONE launch (the six of a training step are split once per step: wn_api.hip pack_weights)
#define WN_G6_PACK_MAXJOBS 24
typedef struct WnGemm6PackJobs {
    int njobs;
    int blk0[WN_G6_PACK_MAXJOBS + 1];   // first block of job j (filled by wn_gemm6_pack_batch)
    const float* src[WN_G6_PACK_MAXJOBS];
    long lda[WN_G6_PACK_MAXJOBS];
    int M[WN_G6_PACK_MAXJOBS], K[WN_G6_PACK_MAXJOBS];
    unsigned short* dst[WN_G6_PACK_MAXJOBS];
    // a job may cover `nl` weight sets of the same shape (the layers of a wide model): set i reads src + i * src_lstride
    // (floats) and writes dst + i * dst_lstride (bf16 elements); gate_R as wn_gemm6_pack
    int nl[WN_G6_PACK_MAXJOBS];
    long src_lstride[WN_G6_PACK_MAXJOBS], dst_lstride[WN_G6_PACK_MAXJOBS];
    int gate_R[WN_G6_PACK_MAXJOBS];
    int f16[WN_G6_PACK_MAXJOBS];   // != 0: the two-piece fp16 image [kb][2][Mpad][16] of k_gemm6<.., F16> instead of the three bf16 pieces
} WnGemm6PackJobs;
static inline void wn_gemm6_pack_job_single(WnGemm6PackJobs* j, int i) {
    j->nl[i] = 1; j->src_lstride[i] = 0; j->dst_lstride[i] = 0; j->gate_R[i] = 0; j->f16[i] = 0;
}
int wn_gemm6_pack_batch(WnGemm6PackJobs* jobs, wn_stream_t st);
static __host__ __device__ inline int wn_gemm6_gate_row(int p, int R) {
    const int mb = p >> 8, wmi = (p >> 7) & 1, i = (p >> 5) & 3, rr = p & 31;
    const int c = mb * 128 + wmi * 64 + (i & 1) * 32 + rr;
    return (i < 2) ? c : R + c;
}
int wn_gemm6_launch(const WnGemm6Args* g, wn_stream_t st);

// Weight-gradient type contraction (both operands k-major, k = time; csrc/wn_gemm.h semantics of the
// a_kmajor = b_kmajor = 1 mode incl. segments, shifts, split-K, layers, a_rowsum) on the same 3-way split.
struct WnGemmArgs;
int wn_gemm6_dw_eligible(const struct WnGemmArgs* g);
// The split-K plan of the caller (wn_api.hip dw_plan) must count tiles with the same two rules the launcher applies:
int wn_gemm6_dw_big(int M, int N);    // 1: 256 x 256 tiles, one wave per SIMD (k_gemm6_dw<4,4>); checked first
int wn_gemm6_dw_tall(int M, int N);   // 1: 256 x 128 tiles (k_gemm6_dw<4,2>) for this output shape
int wn_gemm6_dw_tn(int M, int N);            // otherwise: 3 = one 192-column tile (N = 192), 2 = 128-column tiles, 1 = 64-column tiles
// products: 6, or 3 for leaf results (weight gradients); f16_mul != 0: the fp16 pair split (A times a power of two; raises
// *ovf on a non-finite result) -- f16_mul > 0: that power of two; f16_mul < 0: the kernel reads it from ((const float*)ovf)[1],
// where wn_dw_prepare (wn_elem.h) left it; products 6 with ovf != NULL and f16_mul == 0: conditional redo (works only if *ovf != 0)
int wn_gemm6_dw_launch(const struct WnGemmArgs* g, int products, float f16_mul, int* ovf, wn_stream_t st);

// Skip + residual 1x1 weight gradients of a run of layers in ONE launch (k_dw_skipres, fp16 pair split only):
//   dWskip_i[s][c] = sum_{b,t} dS[b][s][t] z_i[b][c][t]        dWres_i[m][c] = sum_{b,t} dX_{i+1}[b][m][t] z_i[b][c][t]
// Both contract against z_i (64 channels, every position): as two launches z of every layer was read from HBM twice
// (2 x 1.4 GB of the benchmark's 36 GB per step).  A block takes 256 rows of dS, the 128 z rows of two layers and the 64 dX rows
// of each of those layers over one k-chunk (one workgroup of 512 threads per CU); partial sums go to the buffers the two separate launches would have filled
// (same layout, so the conditional six-product redo launches and the reductions behind them are the existing ones):
//   Cskip[zr][S][64 nl]   Cres[i][zr][64][64]   rs_skip[zr][S] (row sums of dS: column block 0 only)   rs_res[i][zr][64]
// zr = b * ksplit + ks.  Layers i >= n_res have no live res_1x1 (the last layer of the stack): nothing is stored for them.
typedef struct WnDwSkipRes {
    int S, nl, n_res, K;
    int nbatch, ksplit, kchunk;   // kchunk % 32 == 0
    const float* dS; long ds_ld, ds_zstride;
    const float* Z; long z_ld, z_zstride, z_lstride;
    const float* dX; long dx_ld, dx_zstride, dx_lstride;   // dX + i dx_lstride = the gradient at the OUTPUT of layer i's res_1x1
    float* Cskip; float* Cres; float* rs_skip; float* rs_res;
} WnDwSkipRes;
int wn_dw_skipres_supported(int S, int R, int nl, int n_res);
// f16_mul / ovf as wn_gemm6_dw_launch (fp16 pair split; *ovf := 1 on a non-finite result)
int wn_dw_skipres_launch(const WnDwSkipRes* a, float f16_mul, int* ovf, wn_stream_t st);
