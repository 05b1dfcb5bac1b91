/* wavenet_hip_gemm.h -- argument block of wn_op_gemm (include/wavenet_hip.h), the generic C[z] = A.B contraction
 * the composite entry points are built from.  Plain C; part of the public C ABI (the op-level parity tests and
 * any external caller of wn_op_gemm fill it; ctypes mirror: pytorchwavenetvocoder_amd/_lib.py::WnGemmArgs). */
#ifndef WAVENET_HIP_GEMM_H
#define WAVENET_HIP_GEMM_H
#include <stdint.h>

// C[z][m][n] (+)= sum_k A(m,k) * B(k,n)   for z in [0, nbatch*ksplit)
//
// All activations of the WaveNet path are channel-major (B, C, T) like the reference's
// tensors, so "time" is always the contiguous axis of the B operand:
//   b_kmajor = 0 : forward / dX type.  B row r = k (a channel), contiguous axis c = n (time).
//   b_kmajor = 1 : dW type.            B row r = n (a channel), contiguous axis c = k (time).
// A B row can belong to a *segment* (seg = r / b_seg_len): segments have their own base offset
// (seg * b_seg_stride) and their own shift along the contiguous axis
// (shift = b_shift0 + seg * b_shift_step); element (r, c) reads
//      Bz[seg*b_seg_stride + (r % b_seg_len)*ldb + (c - shift)]   if 0 <= c-shift < b_clen else 0.
// This one mechanism expresses the dilated causal taps (x[t-(K-1-k)d], zero history,
// reference wavenet.py:118-121), the transposed taps of the backward pass (dP[t+(K-1-k)d]) and
// the layer-stacked skip operand (segment = layer).
//   a_kmajor = 0 : A(m,k) = Az[k*lda + m]     (packed / transposed weights)
//   a_kmajor = 1 : A(m,k) = Az[m*lda + k]     (k = time; dW type)
typedef struct WnGemmArgs {
    int M, N, K;
    const float* A;
    long lda;
    long a_zstride;  // per batch index b = z / ksplit
    int a_kmajor;
    const float* B;
    long ldb;
    long b_zstride;
    int b_kmajor;
    int b_seg_len;
    long b_seg_stride;
    int b_shift0;
    int b_shift_step;
    int b_clen;
    int b_relu;  // apply max(.,0) to B elements on load
    // one-hot B operand (front-conv weight gradient): if b_index != null the B element is
    //   (b_index[b*b_index_zstride + (c - shift)] mod b_index_mod == r % b_seg_len) ? 1 : 0
    const int64_t* b_index;
    long b_index_zstride;
    int b_index_mod;
    float* C;
    long ldc;
    long c_zstride;     // per z
    const float* bias;  // [M] or null
    const float* D;     // residual add source (same indexing as C, per batch b) or null
    long ldd;
    long d_zstride;
    const float* E;  // mask source: result *= (E > 0)   (per batch b) or null
    long lde;
    long e_zstride;
    int relu;
    int accumulate;
    int nbatch;
    int ksplit;
    int kchunk;       // k range of split ks: [ks*kchunk, min(K, (ks+1)*kchunk))
    float* a_rowsum;  // optional [nz][M]: sum_k A(m,k) over this z's k range (a_kmajor=1 only)
    const char* tag;  // static string naming the call site (profiling); may be null
    // optional outer "layer" dimension: z = (layer*nbatch + b)*ksplit + ks.  Layer li adds
    // li*a_lstride / li*b_lstride to the operand bases; if b_dil_depth > 0 the shifts are scaled by
    // the layer's dilation 2^((b_layer0 + li) % b_dil_depth)  (reference wavenet.py:184).
    int nlayer;
    long a_lstride;
    long b_lstride;
    int b_dil_depth;
    int b_layer0;
} WnGemmArgs;

static inline WnGemmArgs wn_gemm_default(void) {
    WnGemmArgs g;
    g.M = g.N = g.K = 0;
    g.A = 0; g.lda = 0; g.a_zstride = 0; g.a_kmajor = 0;
    g.B = 0; g.ldb = 0; g.b_zstride = 0; g.b_kmajor = 0;
    g.b_seg_len = 0x7fffffff; g.b_seg_stride = 0; g.b_shift0 = 0; g.b_shift_step = 0; g.b_clen = 0; g.b_relu = 0;
    g.b_index = 0; g.b_index_zstride = 0; g.b_index_mod = 1;
    g.C = 0; g.ldc = 0; g.c_zstride = 0;
    g.bias = 0; g.D = 0; g.ldd = 0; g.d_zstride = 0; g.E = 0; g.lde = 0; g.e_zstride = 0;
    g.relu = 0; g.accumulate = 0;
    g.nbatch = 1; g.ksplit = 1; g.kchunk = 0x7fffffff;
    g.a_rowsum = 0;
    g.tag = 0;
    g.nlayer = 1; g.a_lstride = 0; g.b_lstride = 0; g.b_dil_depth = 0; g.b_layer0 = 0;
    return g;
}

#endif /* WAVENET_HIP_GEMM_H */
