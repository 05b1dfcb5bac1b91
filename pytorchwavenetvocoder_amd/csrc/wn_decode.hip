// wn_decode.hip -- autoregressive sample-by-sample decode kernel (BASELINE config 5).
//
// Reference: WaveNet.fast_generate / batch_fast_generate / _generate_residual_forward
// (wavenet_vocoder/nets/wavenet.py:309-511, 538-549).  One persistent workgroup of 512 threads
// per utterance; step p consumes the tokens at positions p-K+1..p and produces the logits for
// position p+1.  Positions below t_forced are teacher forced (the context), which also builds
// the dilation queues from zero history exactly like the reference's "prepare buffer" pass
// (wavenet.py:338-349: a full forward with zero left padding).
//
// Per step the whole network is a chain of matrix-vector products: the packed weights are
// streamed from L2 through two register rings (one layer ahead for the residual stack, 16 units
// ahead for the post net), the activations of the step live in LDS, cross-lane sums are DPP
// moves, and the block barrier waits for LDS traffic only so the weight stream never drains.
#include "wn_decode.h"

#include <string.h>

#include "wn_prof.h"

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------
static int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
static int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
static long round4(long v) { return (v + 3) & ~3L; }

// compiled unit classes {UD, UR, US, UP1, UP2}
static const int kClasses[][5] = {
    {8, 2, 8, 32, 32},   // R<=64, S<=256, Q<=256, K<=2  (the BASELINE model)
    {12, 2, 8, 32, 32},  // ... K<=3
};
static const int kNumClasses = (int)(sizeof(kClasses) / sizeof(kClasses[0]));

void wn_decode_make_plan(int Q, int A, int R, int S, int L, int K, int depth, WnDecodePlan* pl) {
    memset(pl, 0, sizeof(*pl));
    const int Cpad = pow2_ceil(R) < 8 ? 8 : pow2_ceil(R);
    const int Spad = pow2_ceil(S) < 8 ? 8 : pow2_ceil(S);
    const int Qpad = pow2_ceil(Q) < 8 ? 8 : pow2_ceil(Q);
    if (Cpad > WN_DT || Spad > WN_DT || Qpad > WN_DT || K > 8) return;
    pl->R4 = (R + 3) / 4;
    pl->S4 = (S + 3) / 4;
    pl->lg_pd = pl->lg_pr = ilog2(WN_DT / Cpad);
    pl->lg_ps = pl->lg_p1 = ilog2(WN_DT / Spad);
    pl->lg_p2 = ilog2(WN_DT / Qpad);
    const int pd = 1 << pl->lg_pd, ps = 1 << pl->lg_ps, p2 = 1 << pl->lg_p2;
    const int need[5] = {2 * ((K * pl->R4 + pd - 1) / pd), (pl->R4 + pd - 1) / pd, (pl->R4 + ps - 1) / ps,
                         (pl->S4 + ps - 1) / ps, (pl->S4 + p2 - 1) / p2};
    pl->cls = -1;
    for (int c = 0; c < kNumClasses && pl->cls < 0; ++c) {
        bool fits = true;
        for (int i = 0; i < 5; ++i) fits = fits && need[i] <= kClasses[c][i];
        if (fits) pl->cls = c;
    }
    if (pl->cls < 0) return;
    pl->UD = kClasses[pl->cls][0];
    pl->UR = kClasses[pl->cls][1];
    pl->US = kClasses[pl->cls][2];
    pl->UP1 = kClasses[pl->cls][3];
    pl->UP2 = kClasses[pl->cls][4];
    // the post net rides the same register ring as the layers: its UP1+UP2 units are cut into
    // pseudo-layers of UL units (the last one zero padded)
    const int UL = pl->UD + pl->UR + pl->US;
    const int NPL = (pl->UP1 + pl->UP2 + UL - 1) / UL;
    pl->stream_f4 = (long)(L + NPL) * UL * WN_DT;
    pl->off_cvec = pl->stream_f4 * 4;
    pl->off_bskip = pl->off_cvec + round4((long)L * 2 * R);
    pl->off_wauxf = pl->off_bskip + round4(S);
    pl->off_one = pl->off_wauxf + round4((long)A * L * 2 * R);
    pl->total_floats = pl->off_one + 64;
    long sumd = 0;
    for (int l = 0; l < L; ++l) sumd += 1L << (l % depth);
    pl->queue_floats = (long)(K - 1) * sumd * R;
    // step state (inputs, aux, gate, skip, post, logits, token history) + the bias tables (res_1x1 of every
    // layer, summed skip, post 1, post 2): a bias read from global memory inside the layer loop would make the
    // consumer wait for every weight unit in flight (vmcnt is in-order)
    const long lds_f = (long)L * K * pl->R4 * 4 + round4((long)L * 2 * R) + pl->R4 * 4 + 2L * pl->S4 * 4 + Qpad + 16 +
                       round4((long)L * R) + 2L * pl->S4 * 4 + Qpad + round4((long)L * 2 * R) /* cvec */ + pl->R4 * 4 /* causal bias */;
    pl->lds_bytes = (size_t)lds_f * 4;
    pl->ok = pl->lds_bytes <= 160 * 1024 ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// stream packing: one thread per float4 unit
// ------------------------------------------------------------------------------------------
__global__ void k_decode_pack(WnDecodePackArgs a) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const WnDecodePlan& pl = a.plan;
    if (g >= pl.stream_f4) return;
    const int tid = (int)(g % WN_DT);
    const long u = g / WN_DT;
    const int UL = pl.UD + pl.UR + pl.US;
    const int R = a.R, S = a.S, Q = a.Q, K = a.K;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (u < (long)a.L * UL) {
        const int l = (int)(u / UL), j = (int)(u % UL);
        const long base = a.lb0 + (long)l * a.lstep;
        if (j < pl.UD) {  // dilated conv, sigmoid / tanh row pair of channel c  (wavenet.py:527-528)
            // unit pair (2r, 2r+1) = inputs 4*i4..+3 of both rows, interleaved for packed fma:
            //   unit 2r = [Ws.x, Wg.x, Ws.y, Wg.y], unit 2r+1 = [Ws.z, Wg.z, Ws.w, Wg.w]
            const int r = j >> 1, half = j & 1;
            const int c = tid >> pl.lg_pd, i4 = (tid & ((1 << pl.lg_pd) - 1)) + (r << pl.lg_pd);
            if (c < R && i4 < K * pl.R4) {
                const int tap = i4 / pl.R4, ib = 4 * (i4 % pl.R4) + 2 * half;
                for (int e = 0; e < 4; ++e) {
                    const int i = ib + (e >> 1);
                    const long w0 = base + ((e & 1) ? a.o_dtanh_w : a.o_dsig_w);
                    if (i < R) v[e] = a.params[w0 + ((long)c * R + i) * K + tap];
                }
            }
        } else if (j < pl.UD + pl.UR) {  // res_1x1  (wavenet.py:534)
            const int r = j - pl.UD;
            const int o = tid >> pl.lg_pr, i4 = (tid & ((1 << pl.lg_pr) - 1)) + (r << pl.lg_pr);
            if (o < R && i4 < pl.R4)
                for (int e = 0; e < 4; ++e)
                    if (4 * i4 + e < R) v[e] = a.params[base + a.o_res_w + (long)o * R + 4 * i4 + e];
        } else {  // skip_1x1  (wavenet.py:533)
            const int r = j - pl.UD - pl.UR;
            const int o = tid >> pl.lg_ps, i4 = (tid & ((1 << pl.lg_ps) - 1)) + (r << pl.lg_ps);
            if (o < S && i4 < pl.R4)
                for (int e = 0; e < 4; ++e)
                    if (4 * i4 + e < R) v[e] = a.params[a.skip0 + (long)l * a.ls_skip + (long)o * R + 4 * i4 + e];
        }
    } else {
        int j = (int)(u - (long)a.L * UL);
        if (j >= pl.UP1 + pl.UP2) {  // padding of the last pseudo-layer
        } else if (j < pl.UP1) {  // conv_post_1  (wavenet.py:520)
            const int o = tid >> pl.lg_p1, i4 = (tid & ((1 << pl.lg_p1) - 1)) + (j << pl.lg_p1);
            if (o < S && i4 < pl.S4)
                for (int e = 0; e < 4; ++e)
                    if (4 * i4 + e < S) v[e] = a.params[a.post1_w + (long)o * S + 4 * i4 + e];
        } else {  // conv_post_2  (wavenet.py:522)
            j -= pl.UP1;
            const int q = tid >> pl.lg_p2, i4 = (tid & ((1 << pl.lg_p2) - 1)) + (j << pl.lg_p2);
            if (q < Q && i4 < pl.S4)
                for (int e = 0; e < 4; ++e)
                    if (4 * i4 + e < S) v[e] = a.params[a.post2_w + (long)q * S + 4 * i4 + e];
        }
    }
    float* dst = a.stream + 4 * g;
    dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
}

int wn_decode_pack_stream(const WnDecodePackArgs* a, wn_stream_t st) {
    WN_PROF("decode_pack", 0.0, 0.0, st);
    const long n = a->plan.stream_f4;
    WN_LAUNCH(k_decode_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *a);
    return 0;
}

// ------------------------------------------------------------------------------------------
// decode kernel
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
// single-row unit: two interleaved partial sums (even / odd inputs)
static __device__ __forceinline__ f32x2 dot4p(const float4& w, const float4& x, f32x2 acc) {
    acc = wn_pk_fma(f32x2{w.x, w.y}, f32x2{x.x, x.y}, acc);
    acc = wn_pk_fma(f32x2{w.z, w.w}, f32x2{x.z, x.w}, acc);
    return acc;
}
// sum over the 2^lg adjacent lanes that share an output (lg is workgroup-uniform)
static __device__ __forceinline__ float group_sum(float v, int lg) {
    if (lg > 0) {
        v = wn_xor_add(v, 1);
        if (lg > 1) {
            v = wn_xor_add(v, 2);
            if (lg > 2) {
                v = wn_xor_add(v, 4);
                if (lg > 3) {
                    v = wn_xor_add(v, 8);
                    if (lg > 4) {
                        v = wn_xor_add(v, 16);
                        if (lg > 5) v = wn_xor_add(v, 32);
                    }
                }
            }
        }
    }
    return v;
}
static __device__ __forceinline__ void group_sum2(float& a, float& b, int lg) {
    if (lg > 0) {
        a = wn_xor_add(a, 1); b = wn_xor_add(b, 1);
        if (lg > 1) {
            a = wn_xor_add(a, 2); b = wn_xor_add(b, 2);
            if (lg > 2) {
                a = wn_xor_add(a, 4); b = wn_xor_add(b, 4);
                if (lg > 3) {
                    a = wn_xor_add(a, 8); b = wn_xor_add(b, 8);
                    if (lg > 4) {
                        a = wn_xor_add(a, 16); b = wn_xor_add(b, 16);
                        if (lg > 5) { a = wn_xor_add(a, 32); b = wn_xor_add(b, 32); }
                    }
                }
            }
        }
    }
}
// float offset of layer l's queue: R*(K-1)*sum_{l'<l} d_l'   (dilations 1,2,..,2^(depth-1) repeated)
static __device__ __forceinline__ long queue_off(int l, int depth, int K, int R) {
    const long cyc = l / depth, in = l % depth;
    return (long)R * (K - 1) * (cyc * ((1L << depth) - 1) + ((1L << in) - 1));
}

#ifdef WN_TIMING
// Experimental build only (tools/decode_timing.py): cycle stamps of workgroup 0, wave 0 / wave 7.
static long long* g_dec_dbg = nullptr;
extern "C" void wn_decode_debug_set_buffer(void* p) { g_dec_dbg = (long long*)p; }
#define DSTAMP(i) do { if (a.dbg && b == 0 && (tid & 63) == 0 && p == a.p1 - 1) a.dbg[(tid >> 6) * 64 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define DSTAMP(i)
#endif

// The LDS addresses of the matrix-vector units are loop invariant; hoisted out of the step loop they would pin
// ~90 registers next to the weight ring.  Passing the lane's role through an empty asm once per layer / per step
// makes the compiler recompute them (two VALU ops each) where they are used.
#ifdef WN_EMU
#define WN_OPAQUE(x) (x)
#else
static __device__ __forceinline__ int wn_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
#define WN_OPAQUE(x) wn_opaque(x)
#endif

template <int UD, int UR, int US, int UP1, int UP2>
__global__ __launch_bounds__(WN_DT) void k_decode(WnDecodeArgs a) {
    constexpr int UL = UD + UR + US;
    constexpr int UP = UP1 + UP2;
    constexpr int NPL = (UP + UL - 1) / UL;  // pseudo-layers of the post net in the ring
    WN_DYN_SMEM(smem_raw);
    float* lds = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, b = blockIdx.x;
    const int R = a.R, S = a.S, Q = a.Q, Qo = a.Qo, L = a.L, K = a.K, depth = a.depth;
    const WnDecodePlan& pl = a.plan;
    const int R4 = pl.R4, S4 = pl.S4;
    const int XS = K * R4 * 4;         // floats of one layer's input window [tap][R4*4]
    const int CUR = (K - 1) * R4 * 4;  // offset of the newest tap
    const int lgQ = 9 - pl.lg_p2, Qpad = 1 << lgQ;
    const int nG = L * 2 * R;
    float* xin = lds;
    float* aux = xin + (long)L * XS;
    float* zbuf = aux + ((nG + 3) & ~3);
    float* sk = zbuf + R4 * 4;
    float* p1v = sk + S4 * 4;
    float* lgt = p1v + S4 * 4;
    int* tokh = reinterpret_cast<int*>(lgt + Qpad);
    float* b_res = lgt + Qpad + 16;                 // [L][R] res_1x1 biases
    float* b_sk = b_res + (((long)L * R + 3) & ~3L);  // [S] sum of the skip_1x1 biases
    float* b_p1 = b_sk + S4 * 4;                    // [S] conv_post_1 bias
    float* b_p2 = b_p1 + S4 * 4;                    // [Qpad] conv_post_2 bias
    float* c_lds = b_p2 + Qpad;                     // [L][2R] constant part of the aux pre-activations
    float* b_front = c_lds + ((nG + 3) & ~3);       // [R] causal conv bias
    const int lds_floats = (int)(pl.lds_bytes / 4);
    for (int i = tid; i < lds_floats; i += WN_DT) lds[i] = 0.f;

    // thread roles
    const int pd = 1 << pl.lg_pd, pr = 1 << pl.lg_pr, ps = 1 << pl.lg_ps, pp1 = 1 << pl.lg_p1, pp2 = 1 << pl.lg_p2;
    const int c_d = tid >> pl.lg_pd, part_d0 = tid & (pd - 1);
    const int o_r = tid >> pl.lg_pr, part_r0 = tid & (pr - 1);
    const int o_s = tid >> pl.lg_ps, part_s0 = tid & (ps - 1);
    const int o_1 = tid >> pl.lg_p1, part_10 = tid & (pp1 - 1);
    const int o_2 = tid >> pl.lg_p2, part_20 = tid & (pp2 - 1);

    // weight stream: unit u of this thread sits at byte u*WN_DT*16 + tid*16
    const wn_rsrc_t stream = wn_make_buf(a.wpack, (unsigned)(pl.stream_f4 * 16));
    const int voff = tid * 16;
    constexpr unsigned UB = WN_DT * 16;  // bytes per unit
    constexpr unsigned LB = UL * UB;  // bytes per (pseudo-)layer
    const float* cvec = a.wpack + pl.off_cvec;
    const float* bskip = a.wpack + pl.off_bskip;
    const float* P = a.params;

    const int t_end = a.t_end[b], t_forced = a.t_forced[b];
    const int pend = imin(a.p1, t_end - 1);
    int64_t* smp = a.samples + (long)b * a.s_bstride;
    float* qb = a.queues + (long)b * a.q_bstride;

    WN_LDS_BARRIER();
    for (int i = tid; i < L * R; i += WN_DT) b_res[i] = P[a.off_res_b0 + (long)(i / R) * a.res_b_lstride + (i % R)];
    for (int i = tid; i < S; i += WN_DT) {
        b_sk[i] = bskip[i];
        b_p1[i] = P[a.off_post1_b + i];
    }
    for (int i = tid; i < Qo; i += WN_DT) b_p2[i] = P[a.off_post2_b + i];
    for (int i = tid; i < nG; i += WN_DT) c_lds[i] = cvec[i];
    for (int i = tid; i < R; i += WN_DT) b_front[i] = P[a.off_causal_b + i];
    if (tid < 8) {
        const int q = a.p0 - tid;
        if (q >= 0) {
            const long v = (long)(smp[q] % Q);
            tokh[q & 7] = (int)(v < 0 ? v + Q : v);
        }
    }
    // ring prologue: layer 0 and the first PR post units
    float4 W[UL];
    WN_UNROLL
    for (int j = 0; j < UL; ++j) W[j] = wn_buf_load4(stream, voff, j * UB);
    f32x2 acc_sk = f32x2{0.f, 0.f};
    WN_LDS_BARRIER();

    for (int p = a.p0; p < pend; ++p) {
        DSTAMP(0);
        // ---- phase A: history taps, aux pre-activations, front conv -----------------------------
        {
            const int nh = L * (K - 1) * R;
            for (int i0 = tid; i0 < nh; i0 += 4 * WN_DT) {  // four loads in flight, then the LDS writes
                float hv[4];
                int dst[4];
                WN_UNROLL
                for (int u = 0; u < 4; ++u) {
                    const int idx = i0 + u * WN_DT;
                    dst[u] = -1;
                    hv[u] = 0.f;
                    if (idx < nh) {
                        const int l = idx / ((K - 1) * R);
                        const int rem = idx - l * (K - 1) * R;
                        const int j = rem / R, c = rem - j * R;
                        const int d = 1 << (l % depth), Dq = (K - 1) * d;
                        int slot = (p - (K - 1 - j) * d) % Dq;
                        if (slot < 0) slot += Dq;  // not written yet in this run: zero history
                        dst[u] = l * XS + j * R4 * 4 + c;
                        hv[u] = wn_ld_coherent(qb + queue_off(l, depth, K, R) + (long)slot * R + c);
                    }
                }
                WN_UNROLL
                for (int u = 0; u < 4; ++u)
                    if (dst[u] >= 0) xin[dst[u]] = hv[u];
            }
            // aux(t) = upw[t % U] * (Waux h[:, t / U]) + (b_up rowsum(Waux) + b_aux + b_dil); positions
            // inside the left padding replicate the first aux column (wavenet.py:266,336)
            const int t = p > a.n_pad ? p - a.n_pad : 0;
            int f = t / a.Ue;
            const float uw = a.upw[t - f * a.Ue];
            f = imin(f, a.F - 1);
            const float* g = a.G + (long)b * a.g_bstride + (long)f * nG;
            for (int i0 = tid; i0 < nG; i0 += 8 * WN_DT) {
                float gv[8];
                WN_UNROLL
                for (int u = 0; u < 8; ++u) {
                    const int idx = i0 + u * WN_DT;
                    gv[u] = idx < nG ? g[idx] : 0.f;
                }
                WN_UNROLL
                for (int u = 0; u < 8; ++u) {
                    const int idx = i0 + u * WN_DT;
                    if (idx < nG) aux[idx] = fmaf(uw, gv[u], c_lds[idx]);
                }
            }
        }
        float x0 = 0.f;
        if (tid < R) {  // front conv as a gather (wavenet.py:513-516); the taps are requested together
            float wv[4];
            WN_UNROLL
            for (int k = 0; k < 4; ++k) {
                const int kc = imin(k, K - 1);
                const int q = p - (K - 1 - kc);
                wv[k] = P[a.off_causal_w + ((long)tid * Q + tokh[q & 7]) * K + kc];
            }
            x0 = b_front[tid];
            WN_UNROLL
            for (int k = 0; k < 4; ++k)
                if (k < K && p - (K - 1 - k) >= 0) x0 += wv[k];
            for (int k = 4; k < K; ++k) {
                const int q = p - (K - 1 - k);
                if (q >= 0) x0 += P[a.off_causal_w + ((long)tid * Q + tokh[q & 7]) * K + k];
            }
            xin[CUR + tid] = x0;
        }
        DSTAMP(1);
        WN_LDS_BARRIER();
        DSTAMP(2);
        if (K > 1 && tid < R) qb[(long)(p % (K - 1)) * R + tid] = x0;

        // ---- residual stack (wavenet.py:538-549) ------------------------------------------------
        for (int l = 0; l < L; ++l) {
            const float* xl = xin + l * XS;
            const int part_d = WN_OPAQUE(part_d0), part_r = WN_OPAQUE(part_r0), part_s = WN_OPAQUE(part_s0);
            const unsigned nxt = (unsigned)(l + 1) * LB;  // after the last layer: post pseudo-layer 0
            f32x2 sg = f32x2{0.f, 0.f};  // (sigmoid row, tanh row) of channel c_d
            WN_UNROLL
            for (int r = 0; r < UD / 2; ++r) {
                const int i4 = imin(part_d + pd * r, K * R4 - 1);
                const float4 x = *reinterpret_cast<const float4*>(xl + 4 * i4);
                sg = wn_pk_fma(f32x2{W[2 * r].x, W[2 * r].y}, f32x2{x.x, x.x}, sg);
                sg = wn_pk_fma(f32x2{W[2 * r].z, W[2 * r].w}, f32x2{x.y, x.y}, sg);
                W[2 * r] = wn_buf_load4(stream, voff, nxt + (2 * r) * UB);
                sg = wn_pk_fma(f32x2{W[2 * r + 1].x, W[2 * r + 1].y}, f32x2{x.z, x.z}, sg);
                sg = wn_pk_fma(f32x2{W[2 * r + 1].z, W[2 * r + 1].w}, f32x2{x.w, x.w}, sg);
                W[2 * r + 1] = wn_buf_load4(stream, voff, nxt + (2 * r + 1) * UB);
            }
            float as = sg.x, ag = sg.y;
            if (l == 3) DSTAMP(10);
            group_sum2(as, ag, pl.lg_pd);
            if (l == 3) DSTAMP(11);
            if (part_d == 0 && c_d < R)
                zbuf[c_d] = wn_sigmoid(as + aux[l * 2 * R + c_d]) * wn_tanh(ag + aux[l * 2 * R + R + c_d]);
            if (l == 3) DSTAMP(12);
            WN_LDS_BARRIER();
            if (l == 3) DSTAMP(13);
            f32x2 ar2 = f32x2{0.f, 0.f};
            WN_UNROLL
            for (int r = 0; r < UR; ++r) {
                const int i4 = imin(part_r + pr * r, R4 - 1);
                const float4 x = *reinterpret_cast<const float4*>(zbuf + 4 * i4);
                ar2 = dot4p(W[UD + r], x, ar2);
                W[UD + r] = wn_buf_load4(stream, voff, nxt + (UD + r) * UB);
            }
            float ar = ar2.x + ar2.y;
            if (l == 3) DSTAMP(14);
            ar = group_sum(ar, pl.lg_pr);
            if (l == 3) DSTAMP(15);
            if (part_r == 0 && o_r < R && l + 1 < L) {  // the last layer's residual output is dead
                const float xn = ar + b_res[l * R + o_r] + xl[CUR + o_r];
                xin[(l + 1) * XS + CUR + o_r] = xn;
                if (K > 1) {
                    const int Dq = (K - 1) << ((l + 1) % depth);
                    qb[queue_off(l + 1, depth, K, R) + (long)(p % Dq) * R + o_r] = xn;
                }
            }
            WN_UNROLL
            for (int r = 0; r < US; ++r) {
                const int i4 = imin(part_s + ps * r, R4 - 1);
                const float4 x = *reinterpret_cast<const float4*>(zbuf + 4 * i4);
                acc_sk = dot4p(W[UD + UR + r], x, acc_sk);
                W[UD + UR + r] = wn_buf_load4(stream, voff, nxt + (UD + UR + r) * UB);
            }
            if (l == 3) DSTAMP(16);
            WN_LDS_BARRIER();
            if (l == 3) DSTAMP(17);
            if (l == 2) DSTAMP(9);
        }
        DSTAMP(3);

        // ---- post net (wavenet.py:518-523) -------------------------------------------------------
        const int part_s = WN_OPAQUE(part_s0), part_1 = WN_OPAQUE(part_10), part_2 = WN_OPAQUE(part_20);
        {
            const float v = group_sum(acc_sk.x + acc_sk.y, pl.lg_ps);
            acc_sk = f32x2{0.f, 0.f};
            if (part_s == 0 && o_s < S) sk[o_s] = fmaxf(v + b_sk[o_s], 0.f);
        }
        WN_LDS_BARRIER();
        {
            f32x2 acc2 = f32x2{0.f, 0.f};
            WN_UNROLL
            for (int j = 0; j < UP1; ++j) {
                const int i4 = imin(part_1 + pp1 * j, S4 - 1);
                const float4 x = *reinterpret_cast<const float4*>(sk + 4 * i4);
                acc2 = dot4p(W[j % UL], x, acc2);
                W[j % UL] = wn_buf_load4(stream, voff, ((j / UL + 1 < NPL) ? (unsigned)(L + j / UL + 1) * LB : 0u) + (j % UL) * UB);
            }
            const float acc = group_sum(acc2.x + acc2.y, pl.lg_p1);
            if (part_1 == 0 && o_1 < S) p1v[o_1] = fmaxf(acc + b_p1[o_1], 0.f);
        }
        WN_LDS_BARRIER();
        {
            f32x2 acc2 = f32x2{0.f, 0.f};
            WN_UNROLL
            for (int j = UP1; j < UP; ++j) {
                const int i4 = imin(part_2 + pp2 * (j - UP1), S4 - 1);
                const float4 x = *reinterpret_cast<const float4*>(p1v + 4 * i4);
                acc2 = dot4p(W[j % UL], x, acc2);
                W[j % UL] = wn_buf_load4(stream, voff, ((j / UL + 1 < NPL) ? (unsigned)(L + j / UL + 1) * LB : 0u) + (j % UL) * UB);
            }
            const float acc = group_sum(acc2.x + acc2.y, pl.lg_p2);
            if (part_2 == 0) lgt[o_2] = o_2 < Qo ? acc + b_p2[o_2] : -3.0e38f;
            // slots of the padded tail are never consumed: refill them with layer 0 directly
            WN_UNROLL
            for (int j = UP; j < NPL * UL; ++j) W[j % UL] = wn_buf_load4(stream, voff, (j % UL) * UB);
        }
        WN_LDS_BARRIER();

        DSTAMP(4);
        // ---- next token: argmax or categorical draw (wavenet.py:370-378) -------------------------
        if (a.logits_out != nullptr) {
            float* lo = a.logits_out + (long)b * a.lo_bstride + (long)p * Qo;
            for (int q = tid; q < Qo; q += WN_DT) lo[q] = lgt[q];
        }
        if (a.mode == 2) {
            // mixture-of-logistics head (not in the reference; same draw as k_dl_select_mol): component by Gumbel
            // max over the nm mixture logits, value = mean + scale * logit(u), clipped, mu-law token for the front end
            if (tid < 64) {
                const int nm = Qo / 3;
                const bool gen = p + 1 >= t_forced;
                const float* un = a.uniforms + ((long)b * a.u_bstride + p + 1) * (nm + 1);
                float best = -3.0e38f;
                int bi = 0x7fffffff;
                if (gen && tid < nm) {
                    best = lgt[tid] - logf(-logf(un[tid]));
                    bi = tid;
                }
                for (int m = 1; m < 64; m <<= 1) {
                    const float ov = __shfl_xor(best, m, 64);
                    const int oi = __shfl_xor(bi, m, 64);
                    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
                }
                if (tid == 0) {
                    int next;
                    if (!gen) {
                        const long v = (long)(smp[p + 1] % Q);
                        next = (int)(v < 0 ? v + Q : v);
                    } else {
                        const float mean = lgt[nm + bi];
                        const float ls = fmaxf(lgt[2 * nm + bi], a.log_scale_min);
                        const float uu = un[nm];
                        float xv = mean + expf(ls) * (logf(uu) - logf(1.0f - uu));
                        xv = fminf(fmaxf(xv, -1.0f), 1.0f);
                        if (a.wave_out) a.wave_out[(long)b * a.w_bstride + p + 1] = xv;
                        const float mu = (float)(Q - 1);
                        const float fx = copysignf(logf(1.0f + mu * fabsf(xv)) / logf(1.0f + mu), xv);
                        next = (int)floorf((fx + 1.0f) * 0.5f * mu + 0.5f);
                        smp[p + 1] = next;
                    }
                    tokh[(p + 1) & 7] = next;
                }
            }
        } else if (tid < 64) {
            const int per = Qpad >= 64 ? (Qpad >> 6) : 1;
            const int q0 = tid * per;
            float best = -3.0e38f;
            int bi = 0x7fffffff;
            for (int i = 0; i < per; ++i) {
                const int q = q0 + i;
                if (q < Qo) {
                    const float v = lgt[q];
                    if (v > best || bi == 0x7fffffff) { best = v; bi = q; }
                }
            }
            for (int m = 1; m < 64; m <<= 1) {
                const float ov = __shfl_xor(best, m, 64);
                const int oi = __shfl_xor(bi, m, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            int chosen = bi;  // argmax: first maximal index
            if (a.mode == 1 && a.uniforms != nullptr) {
                float lsum = 0.f;
                for (int i = 0; i < per; ++i) {
                    const int q = q0 + i;
                    if (q < Qo) lsum += expf(lgt[q] - best);
                }
                float incl = lsum;  // inclusive scan over lanes
                for (int off = 1; off < 64; off <<= 1) {
                    const float o = __shfl(incl, tid >= off ? tid - off : tid, 64);
                    if (tid >= off) incl += o;
                }
                const float total = __shfl(incl, 63, 64);
                const float target = a.uniforms[(long)b * a.u_bstride + p + 1] * total;
                int cand = 0x7fffffff;
                float run = incl - lsum;
                for (int i = 0; i < per; ++i) {
                    const int q = q0 + i;
                    if (q < Qo) {
                        run += expf(lgt[q] - best);
                        if (cand == 0x7fffffff && run >= target) cand = q;
                    }
                }
                for (int m = 1; m < 64; m <<= 1) {
                    const int oc = __shfl_xor(cand, m, 64);
                    cand = oc < cand ? oc : cand;
                }
                if (cand != 0x7fffffff) chosen = cand;
            }
            if (tid == 0) {
                int next = chosen;
                if (p + 1 < t_forced) {
                    const long v = (long)(smp[p + 1] % Q);
                    next = (int)(v < 0 ? v + Q : v);
                } else {
                    smp[p + 1] = chosen;
                }
                tokh[(p + 1) & 7] = next;
            }
        }
        DSTAMP(5);
        WN_LDS_BARRIER();
        DSTAMP(6);
    }
}

template <int UD, int UR, int US, int UP1, int UP2>
static int launch_cls(const WnDecodeArgs& a, int B, wn_stream_t st) {
#ifndef WN_EMU
    if (a.plan.lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_decode<UD, UR, US, UP1, UP2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.plan.lds_bytes) != hipSuccess)
        return 1;
#endif
#ifdef WN_TIMING
    WnDecodeArgs a2 = a;
    a2.dbg = g_dec_dbg;
    WN_LAUNCH((k_decode<UD, UR, US, UP1, UP2>), dim3((unsigned)B), dim3(WN_DT), a.plan.lds_bytes, st, a2);
    return 0;
#endif
    WN_LAUNCH((k_decode<UD, UR, US, UP1, UP2>), dim3((unsigned)B), dim3(WN_DT), a.plan.lds_bytes, st, a);
    return 0;
}

int wn_decode_launch(const WnDecodeArgs* ap, int B, wn_stream_t st) {
    const WnDecodeArgs& a = *ap;
    if (!a.plan.ok || B <= 0) return 1;
    const double steps = (double)(a.p1 - a.p0) * B;
    WN_PROF("decode_steps", steps * 8.0 * a.plan.stream_f4, steps * 16.0 * a.plan.stream_f4, st);
    switch (a.plan.cls) {
        case 0: return launch_cls<8, 2, 8, 32, 32>(a, B, st);
        case 1: return launch_cls<12, 2, 8, 32, 32>(a, B, st);
        default: return 2;
    }
}
