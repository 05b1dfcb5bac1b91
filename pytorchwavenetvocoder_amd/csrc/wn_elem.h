// wn_elem.h -- launchers of the element-wise / reduction kernels of the WaveNet path (wn_elem.hip).
// All tensors are channel-major (B, C, T) fp32 like the reference's; indices are int64.
#pragma once
#include "wn_device.h"

// x0[b][r][t] = bc[r] + sum_tap Wc_f[tap][x[b][t-(K-1-tap)] mod Q][r]   (zero history)
// == reference OneHot + CausalConv1d(Q->R,K)  (wavenet.py:88-92,513-516,118-121)
int wn_front_gather(const int64_t* x, const float* wc_f /*[K][Q][R]*/, const float* bias /*[R]*/,
                    float* x0 /*[B][R][T]*/, int B, int T, int Q, int R, int K, wn_stream_t st);

// Gate: a = P[b][r][t] + w[t%U]*G[b][r][t/U] + c[r];  g likewise with r+R;
//       s = sigmoid(a), gt = tanh(g), z = s*gt        (wavenet.py:529-532 with the upsampling of
//       wavenet.py:152-154 applied at frame rate: aux(up(h)) = w[j]*(Waux.h[f]) + (b_up*rowsum(Waux)+b_aux))
// U = 1, w = {1} expresses "no upsampling layer" (h at sample rate).
int wn_gate_fwd(const float* P /*[B][2R][T]*/, const float* G /*[B][2R][F]*/, long g_bstride,
                const float* upw /*[U]*/, const float* cvec /*[2R]*/, float* S, float* Gt, float* Z,
                int B, int T, int R, int U, int F, wn_stream_t st);

// dP[b][r][t] = dZ*gt*s*(1-s);  dP[b][R+r][t] = dZ*s*(1-gt^2)
int wn_gate_bwd(const float* dZ, const float* S, const float* Gt, float* dP, int B, int T, int R, wn_stream_t st);

// Softmax cross entropy over logits (B,Q,T) for positions t >= t_start (train.py:534-536):
//   loss = mean_{b,t>=t_start} ( logsumexp_q - logit[target] );  dlogits = (softmax - onehot) * grad_scale
// (zero for t < t_start).  loss_partial has one float per launched block; wn_loss_finalize sums
// them in a fixed order and multiplies by loss_scale.
// amax_partial (nullable): one float per launched block = the block's max |dlogits| (for the measured scale of the fp16 pair
// split of the weight gradients, WN_FLAG_DW_F16PAIR).
int wn_softmax_ce(const float* logits, const int64_t* target, float* dlogits /*nullable*/,
                  float* loss_partial, int* n_partial /*out: host*/, int B, int T, int Q, int t_start,
                  float grad_scale, float* amax_partial, wn_stream_t st);
// out[0] = scale * sum(partial[0..n)); amax_out (nullable): amax_out[0] = max(amax_partial[0..n)) (0 without amax_partial)
int wn_sum_partials(const float* partial, int n, float scale, float* out /*device scalar*/, const float* amax_partial,
                    float* amax_out, wn_stream_t st);
// Scale of the fp16 pair split (WN_FLAG_DW_F16PAIR), decided ON THE DEVICE before the weight-gradient launches of a backward call.
// words (workspace, >= 3 words): [0] overflow flag := 0; [1] a_mul := 2^(floor(-log2 amax) + headroom) (float); [2] amax (float).
//   host_mul > 0 : the caller's promise: words[1] := host_mul
//   scan != NULL : amax := max over scan[0..n_scan) (partial maxima of wn_absmax_rows), stored to words[2]
//   otherwise    : amax = words[2] as a loss call of this workspace left it
// amax == 0 or not finite: a_mul := 1 and words[0] := 1 -- the six-product redo behind every fp16 launch then does the work.
int wn_dw_prepare(float* words, float host_mul, const float* scan, int n_scan, int headroom, wn_stream_t st);
// partial[r * nchunk + c] = max |p[r * ld + c0 + 4096 c + j]|, j < min(4096, ncols - 4096 c); nchunk = ceil(ncols / 4096)
int wn_absmax_rows(const float* p, long rows, long ld, int c0, int ncols, float* partial, wn_stream_t st);
int wn_softmax_ce_nblocks(int B, int T);

// Adam over a flat fp32 buffer (torch.optim.Adam semantics, train.py:457-460): elements in
// [skip_lo, skip_hi) are left untouched (parameters that never receive a gradient).
int wn_adam(float* p, const float* g, float* m, float* v, long n, float lr_over_bc1, float inv_sqrt_bc2,
            float beta1, float beta2, float eps, float weight_decay, long skip_lo, long skip_hi, wn_stream_t st);

// dst[d_off + i0*d0 + i1*d1 + i2*d2 + l*dl] = src[s_off + i0*s0 + i1*s1 + i2*s2 + l*sl]
typedef struct WnCopy4 {
    int n0, n1, n2, nl;
    long d0, d1, d2, dl, s0, s1, s2, sl;
} WnCopy4;
int wn_copy4(float* dst, const float* src, const WnCopy4* c, wn_stream_t st);
// several independent copies in ONE launch (the weight packing of a forward is ten of them)
#define WN_COPY4_MAXJOBS 12
typedef struct WnCopy4Batch {
    int njobs;
    int blk0[WN_COPY4_MAXJOBS + 1];  // first block of job j (filled by wn_copy4_batch)
    float* dst[WN_COPY4_MAXJOBS];
    const float* src[WN_COPY4_MAXJOBS];
    WnCopy4 c[WN_COPY4_MAXJOBS];
} WnCopy4Batch;
static inline int wn_copy4_batch_add(WnCopy4Batch* b, float* dst, const float* src, const WnCopy4* c) {
    if (b->njobs >= WN_COPY4_MAXJOBS) return 1;
    b->dst[b->njobs] = dst; b->src[b->njobs] = src; b->c[b->njobs] = *c;
    b->njobs++;
    return 0;
}
int wn_copy4_batch(WnCopy4Batch* b, wn_stream_t st);

// cvec[l][o'] = b_dil[o'] + b_aux[o'] + b_up * sum_a Waux[o'][a]   (o' in [0,2R): sigmoid then tanh)
// rowsum_aux[l][o'] = sum_a Waux[o'][a]
typedef struct WnCvecArgs {
    const float* params;
    long off_dsig_b, off_dtanh_b, off_asig_w, off_atanh_w, off_asig_b, off_atanh_b;  // layer 0
    long ls_dil, ls_aux;                                                           // layer strides
    long off_up_b;                                                                 // -1: none
    int L, R, A;
    float* cvec;
    float* rowsum_aux;
} WnCvecArgs;
int wn_cvec(const WnCvecArgs* a, wn_stream_t st);
// out[s] = sum_l params[off + l*ls + s]
int wn_sum_layers(const float* params, long off, long ls, int L, int n, float* out, wn_stream_t st);

// Aux backward for one layer (upsampling layer present):
//   dG[b][o'][f] = sum_j w[j] dP[b][o'][fU+j];   dw_partial[(b*2R+o')][j] = sum_f dP[b][o'][fU+j] G[b][o'][f]
// Batched over nl layers (grid.z): dP += l*dp_lstride, G += l*R2*F (layer rows inside the per-batch G
// block), dG += l*B*R2*F, dw_partial += l*B*R2*U.
int wn_aux_bwd(const float* dP, long dp_lstride, const float* G, long g_bstride, const float* upw, float* dG,
               float* dw_partial, int B, int T, int R2, int U, int F, int nl, wn_stream_t st);
// The same outputs from the partial sums the gate kernel leaves behind (wn_fused_bwd_gate_aux), nl layers per launch:
//   dG[l][b][o'][f]            = sum_{i < U/16} dGp[l][b][o'][f*(U/16) + i]
//   dw_partial[l][b][0][j]     = sum_f qp[l][b][fU + j]      (rows o' > 0 of the block are zero: qp is already summed over o')
// dGp += l*dgp_lstride, qp += l*qp_lstride; dG / dw_partial laid out as wn_aux_bwd writes them.
int wn_aux_finish(const float* dGp, long dgp_lstride, const float* qp, long qp_lstride, float* dG, float* dw_partial, int B,
                  int T, int R2, int U, int F, int nl, wn_stream_t st);

// out[map(m,n)] (=|+=) scale * sum_z partial[z][m*N+n] (+ addend_m[m]*addend_scale)
// map(m,n) = (m/m_seg)*m_seg_stride + (m%m_seg)*m_stride + (n/n_seg)*n_seg_stride + (n%n_seg)*n_stride
typedef struct WnReduceArgs {
    const float* partial;
    int nz, M, N;
    float* out;
    int m_seg, n_seg;
    long m_seg_stride, m_stride, n_seg_stride, n_stride;
    float scale;
    int accumulate;
    const float* addend_m;  // nullable, [M]
    const float* addend_scale_ptr;  // nullable device scalar multiplied onto addend_m
    float* scratch;        // nullable: second-level buffer for two-level reductions
    long scratch_floats;
    int nl;                // layers (grid.z): partial += l*nz*M*N, out += l*out_lstride, addend_m += l*addend_lstride
    long out_lstride;
    long addend_lstride;
} WnReduceArgs;
int wn_reduce(const WnReduceArgs* a, wn_stream_t st);

// out[0] (=|+=) sum_i a[i]*b[i]
int wn_dot(const float* a, const float* b, long n, float* out, int accumulate, wn_stream_t st);
int wn_fill(float* p, float v, long n, wn_stream_t st);
// p[r * stride + c] = 0 for r < rows, c < ncols (ncols % 4 == 0, rows 16-byte aligned)
int wn_fill_cols(float* p, long rows, long stride, int ncols, wn_stream_t st);

// Front-conv weight / bias gradient as a scatter (reference: autograd of OneHot + CausalConv1d,
// wavenet.py:78-92,513-516): dW[c][q][k] = sum_{b,t} dX0[b][c][t] * [x[b][t-(K-1-k)] mod Q == q]  (zero history),
// db[c] = sum_{b,t} dX0[b][c][t].  One LDS table [R][K*Q] per workgroup, every wave owns its own channel
// rows and walks time 64 steps at a time (lane = time step: coalesced loads, in-order LDS adds), per-block
// tables are summed in block order -> deterministic.  Needs R*K*Q*4 <= 150 KB of LDS.
int wn_front_dw_supported(int R, int K, int Q);
long wn_front_dw_partial_floats(int B, int T, int R, int K, int Q);
int wn_front_dw(const float* dX0, const int64_t* x, float* partial, float* dW, float* db, int B, int T, int R, int K, int Q,
                wn_stream_t st);

// ---- any-size decode (layer-wise launches; utterances are the contiguous axis of every matrix) ----
// Step inputs for all layers: history taps from the queue rings into the per-layer operand windows
// xin[l][tap][R][nb], aux pre-activations Gstep[l*2R+o][u] = upw[j]*G[u][f][l*2R+o], and the front conv of
// the newest tokens into xin[0][K-1] (reference wavenet.py:355-356, 513-516).
typedef struct WnDlArgs {
    int nb, L, K, R, Q, depth, nG;  // nG = L*2R
    int p, n_pad, Ue, F;
    const float* params;
    long off_causal_w, off_causal_b;
    const float* upw;
    const float* G;       // (nb, F, nG)
    const int64_t* samples;  // (nb, Ttot)
    long Ttot;
    float* queues;        // layer l at qoff(l)*nb: [slot][R][nb]
    float* xin;           // [L][K][R][nb]
    float* gstep;         // [nG][nb]
} WnDlArgs;
int wn_dl_inputs(const WnDlArgs* a, wn_stream_t st);
// After the step: push the layer inputs of position p into the queue rings.
int wn_dl_push(const WnDlArgs* a, wn_stream_t st);
// Token choice per utterance from logits [Q][nb]: argmax or inverse-CDF draw; teacher forcing below t_forced.
int wn_dl_select(const float* logits, int Q, int nb, int64_t* samples, long Ttot, const int* t_forced, const int* t_end,
                 int p, const float* uniforms, float* logits_out, int mode, wn_stream_t st);

// Skinny contraction of the any-size decode: C[z][m][u] = epi( sum_k A[z](m,k) * B[z][k][u] ), u < nb (tens of
// utterances), A(m,k) = Az[k*lda + m] (the packed transposed weights).  One workgroup = 32 output rows x 32
// columns; its 4 waves split K and stream their weight rows straight from global memory into the A operand of
// the f32 MFMA (no LDS staging: every weight is used once), partial tiles are summed through LDS.
// epi: + bias[m] + D[m][u], relu.
typedef struct WnDlMmArgs {
    int M, K, nb;
    const float* A; long lda; long a_zstride;
    const float* B; long ldb; long b_zstride;
    float* C; long ldc; long c_zstride;
    const float* bias;
    const float* D; long ldd;
    int relu;
    int nz;
    const char* tag;
    // Gate epilogue (gate_R > 0, M = 2*gate_R, gate_R % 16 == 0): a tile holds rows c0..c0+15 (sigmoid half) and
    // gate_R+c0.. (tanh half) of the same 16 channels and writes C[c][u] = sigmoid(acc_s + (gate_g[c][u] + gate_c[c])) *
    // tanh(acc_t + (gate_g[R+c][u] + gate_c[R+c]))  (wavenet.py:542-544) instead of the 2R pre-activations.
    int gate_R;
    const float* gate_g;  // [2R][nb] aux pre-activations of this layer and step
    const float* gate_c;  // [2R] constant part
} WnDlMmArgs;
int wn_dl_mm(const WnDlMmArgs* a, wn_stream_t st);
// out[m][u] = relu?( sum_z part[z][m][u] + bias[m] )
int wn_dl_sum(const float* part, int nz, long zstride, int M, int nb, const float* bias, int relu, float* out, wn_stream_t st);

// ---- mixture-of-logistics output head (BASELINE configs[3]; NOT in the reference: the formulas are the
// discretised mixture of logistics of PixelCNN++, Salimans et al. 2017, as used by WaveNet vocoders) ----
// out (B, 3*nm, T): rows [0,nm) mixture logits, [nm,2nm) means, [2nm,3nm) log-scales (clamped at log_scale_min);
// y (B, T) target waveform in [-1, 1]; num_classes = quantisation levels of the waveform (65536 for 16 bit).
// loss_partial gets one partial sum of the negative log-likelihood per block over t >= t_start; dout (nullable)
// gets d(sum nll * grad_scale)/d(out), zero for t < t_start.
int wn_mol_nll(const float* out, const float* y, float* dout, float* loss_partial, int* n_partial, int B, int T, int nm,
               int t_start, float grad_scale, int num_classes, float log_scale_min, wn_stream_t st);
// Decode-side draw of the MoL head for every utterance from out [3*nm][nb]: component by Gumbel max with the
// uniforms u[0..nm), value = mean + scale*(log u_nm - log(1-u_nm)) clipped to [-1,1]; the value is mu-law
// encoded (levels Q) into the token fed back to the one-hot front end; wave_out (nb, Ttot) keeps the float.
int wn_dl_select_mol(const float* out, int nm, int nb, int Q, int64_t* samples, float* wave_out, long Ttot, const int* t_forced,
                     const int* t_end, int p, const float* uniforms /* (nb, Ttot, nm+1) */, float* out_copy, float log_scale_min,
                     wn_stream_t st);

// ---- parallel context walk of the decode path (reference wavenet.py:338-349: the "prepare buffer" pass IS a
// full forward over the padded context) ----
// Sample-rate aux features of the padded context: out (B, A, T); column p is position pos0 + p of the padded
// context and reads the upsampled feature of sample t = max(pos0 + p - n_pad, 0) (the left padding replicates the first upsampled column, wavenet.py:336):
// U > 0: upw[t % U] * h[b][a][min(t / U, F-1)] + upb[0]  (ConvTranspose2d (1,U)/(1,U), wavenet.py:141-154);
// U == 0: h[b][a][min(t, F-1)].
int wn_decode_ctx_aux_rows(const float* h, const float* upw, const float* upb, float* out, int B, int A, int F, int U, int T,
                           int n_pad, int pos0, wn_stream_t st);
// Dilation queues after a context of P0 positions from the layer inputs X [L][B][R][T] of a training forward:
// for layer l the Dq = (K-1)*d_l newest positions q in [P0-Dq, P0) go to slot (pos0 + q) % Dq (pos0 = absolute
// position of column 0 of X), dst[(queue_off(l) + slot*R + c) * elem_stride + b * utt_stride] = X[l][b][c][q].
int wn_decode_fill_queues(const float* X, float* dst, int L, int B, int R, int T, int K, int depth, int P0, int pos0,
                          long elem_stride, long utt_stride, wn_stream_t st);
// (B, R, C) -> (B, C, R)
int wn_transpose_last2(const float* src, float* dst, int B, int R, int C, wn_stream_t st);
