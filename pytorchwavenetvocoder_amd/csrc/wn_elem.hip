// wn_elem.hip -- element-wise / reduction kernels of the WaveNet training path (gfx950).
// These are the HBM-bound parts: every kernel walks the time axis with consecutive lanes on
// consecutive samples (coalesced 256 B per wave-instruction) and all reductions are two-stage
// and order-deterministic (no float atomics).
#include "wn_elem.h"
#include "wn_prof.h"

#define WN_TPB 256

static __device__ __forceinline__ float block_reduce_sum(float v, float* red /*[4]*/) {
    v = wave_reduce_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.0f;
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// max(a, |v|) on the bit patterns: a and |v| are sign-free, so integer order = float order with inf and NaN on top (kept)
static __device__ __forceinline__ float wn_absmax_keep_nan(float a, float v) {
    const int ai = __builtin_bit_cast(int, a), vi = __builtin_bit_cast(int, v) & 0x7fffffff;
    return __builtin_bit_cast(float, vi > ai ? vi : ai);
}
static __device__ __forceinline__ float block_reduce_max(float v, float* red /*[4]*/) {
    v = wave_reduce_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WN_TPB) void k_front_gather(const int64_t* __restrict__ x, const float* __restrict__ wc_f,
                                                         const float* __restrict__ bias, float* __restrict__ x0, int T,
                                                         int Q, int R, int K) {
    const int t = blockIdx.x * WN_TPB + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= T) return;
    int q[8];
    for (int tap = 0; tap < K; ++tap) {
        const int idx = t - (K - 1 - tap);
        int v = -1;
        if (idx >= 0) {
            long long xv = x[(long)b * T + idx];
            if ((unsigned long long)xv >= (unsigned long long)Q) {   // OneHot's modulo (wavenet.py:88): 64-bit division, rare
                xv %= Q;
                if (xv < 0) xv += Q;
            }
            v = (int)xv;
        }
        q[tap] = v;
    }
    for (int r = 0; r < R; ++r) {
        float v = bias[r];
        for (int tap = 0; tap < K; ++tap)
            if (q[tap] >= 0) v += wc_f[((long)tap * Q + q[tap]) * R + r];
        x0[((long)b * R + r) * T + t] = v;
    }
}

// The same gather with the weight table in LDS.  k_front_gather reads wc_f[(tap Q + q) R + r] with a different q in every
// lane: 64 cache lines per load instruction, K R of them per thread -- bound by the texture addresser (63 us for the
// benchmark's 47 MB of output).  Here one 1024-thread workgroup per CU copies the table (K Q R floats: 128 KB for the
// benchmark's model) into LDS once, rows padded to R + 1 words so that the lanes' rows fall into different banks, and walks
// a chunk of one sequence.  Same additions in the same order: bit-identical output.  Tables too large for one CU's LDS (the
// recipes' n_resch = 512: 1 MB) are cut into groups of RG output rows on blockIdx.z (a [K Q][RG + 1] slice each).
#define FG_T 1024
__global__ __launch_bounds__(FG_T) void k_front_gather_lds(const int64_t* __restrict__ x, const float* __restrict__ wc_f,
                                                           const float* __restrict__ bias, float* __restrict__ x0, int T,
                                                           int Q, int R, int K, int chunk, int RG) {
    WN_DYN_SMEM(smem_raw);
    float* tab = reinterpret_cast<float*>(smem_raw);   // [K * Q][RG + 1]
    const int RP = RG + 1;
    const int r_lo = (int)blockIdx.z * RG;
    const int tid = threadIdx.x;
    const int b = blockIdx.y, t0 = blockIdx.x * chunk;
    const int t1 = (t0 + chunk < T) ? t0 + chunk : T;
    {   // table copy: 8 independent 16-byte loads in flight per thread (a dependent load per iteration made it latency bound)
        const int n4 = (K * Q * RG) >> 2;   // RG % 4 == 0 (launcher): the 4 elements of a load share a table row
        for (int base = 0; base < n4; base += 8 * FG_T) {
            float4 v[8];
            WN_UNROLL
            for (int u = 0; u < 8; ++u) {
                const int i4 = base + u * FG_T + tid;
                const int e = i4 << 2, row = e / RG, r = e - row * RG;
                v[u] = i4 < n4 ? *reinterpret_cast<const float4*>(wc_f + (long)row * R + r_lo + r) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            WN_UNROLL
            for (int u = 0; u < 8; ++u) {
                const int i4 = base + u * FG_T + tid;
                if (i4 < n4) {
                    const int e = i4 << 2, row = e / RG, r = e - row * RG;
                    float* d = tab + row * RP + r;
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                }
            }
        }
    }
    __syncthreads();
    for (int t = t0 + tid; t < t1; t += FG_T) {
        int q[8];
        for (int tap = 0; tap < K; ++tap) {
            const int idx = t - (K - 1 - tap);
            int v = -1;
            if (idx >= 0) {
                long long xv = x[(long)b * T + idx];
                if ((unsigned long long)xv >= (unsigned long long)Q) {
                    xv %= Q;
                    if (xv < 0) xv += Q;
                }
                v = (tap * Q + (int)xv) * RP;
            }
            q[tap] = v;
        }
        for (int r0 = 0; r0 < RG; r0 += 4) {   // RG % 4 == 0; 4 K LDS reads in flight
            float v[4];
            WN_UNROLL
            for (int u = 0; u < 4; ++u) v[u] = bias[r_lo + r0 + u];
            for (int tap = 0; tap < K; ++tap) {
                if (q[tap] >= 0) {
                    WN_UNROLL
                    for (int u = 0; u < 4; ++u) v[u] += tab[q[tap] + r0 + u];
                }
            }
            WN_UNROLL
            for (int u = 0; u < 4; ++u) x0[((long)b * R + r_lo + r0 + u) * T + t] = v[u];
        }
    }
}

int wn_front_gather(const int64_t* x, const float* wc_f, const float* bias, float* x0, int B, int T, int Q, int R, int K,
                    wn_stream_t st) {
    WN_PROF("front_gather", 0.0, 0.0, st);
    if (K > 8 || K < 1) return 1;
    int RG = R;   // output rows per workgroup: all of them if the table fits one CU's LDS, else groups of 64 / 32 / 16
    while (RG > 16 && ((size_t)K * Q * (RG + 1) * 4 > 150 * 1024 || R % RG != 0)) RG = RG > 64 ? 64 : RG / 2;
    const size_t lds = (size_t)K * Q * (RG + 1) * 4;
    if (lds <= 150 * 1024 && R % RG == 0 && RG % 4 == 0 && (long)B * T >= 16384) {   // (small calls: the table copy would dominate)
        int nc = 256 / (B * (R / RG));
        if (nc < 1) nc = 1;
        int ch = (T + nc - 1) / nc;
        ch = (ch + 63) / 64 * 64;
        nc = (T + ch - 1) / ch;
#ifndef WN_EMU
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_gather_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return 2;
#endif
        WN_LAUNCH(k_front_gather_lds, dim3((unsigned)nc, (unsigned)B, (unsigned)(R / RG)), dim3(FG_T), lds, st, x, wc_f, bias, x0, T, Q, R, K, ch, RG);
        return 0;
    }
    dim3 grid((T + WN_TPB - 1) / WN_TPB, B);
    WN_LAUNCH(k_front_gather, grid, dim3(WN_TPB), 0, st, x, wc_f, bias, x0, T, Q, R, K);
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WN_TPB) void k_gate_fwd(const float* __restrict__ P, const float* __restrict__ G, long g_bstride,
                                                     const float* __restrict__ upw, const float* __restrict__ cvec,
                                                     float* __restrict__ S, float* __restrict__ Gt, float* __restrict__ Z, int T,
                                                     int R, int U, int F) {
    const int t = blockIdx.x * WN_TPB + threadIdx.x;
    const int r = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const int f = t / U, j = t - f * U;
    const float w = upw[j];
    const float* Gb = G + (long)b * g_bstride;
    const float a = P[((long)b * 2 * R + r) * T + t] + (w * Gb[(long)r * F + f] + cvec[r]);
    const float g = P[((long)b * 2 * R + R + r) * T + t] + (w * Gb[(long)(R + r) * F + f] + cvec[R + r]);
    const float s = wn_sigmoid(a), gt = wn_tanh(g);
    const long o = ((long)b * R + r) * T + t;
    S[o] = s;
    Gt[o] = gt;
    Z[o] = s * gt;
}

int wn_gate_fwd(const float* P, const float* G, long g_bstride, const float* upw, const float* cvec, float* S, float* Gt,
                float* Z, int B, int T, int R, int U, int F, wn_stream_t st) {
    WN_PROF("gate_fwd", 0.0, 0.0, st);
    if (U < 1) return 1;
    dim3 grid((T + WN_TPB - 1) / WN_TPB, R, B);
    WN_LAUNCH(k_gate_fwd, grid, dim3(WN_TPB), 0, st, P, G, g_bstride, upw, cvec, S, Gt, Z, T, R, U, F);
    return 0;
}

__global__ __launch_bounds__(WN_TPB) void k_gate_bwd(const float* __restrict__ dZ, const float* __restrict__ S,
                                                     const float* __restrict__ Gt, float* __restrict__ dP, int T, int R) {
    const int t = blockIdx.x * WN_TPB + threadIdx.x;
    const int r = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const long o = ((long)b * R + r) * T + t;
    const float dz = dZ[o], s = S[o], g = Gt[o];
    dP[((long)b * 2 * R + r) * T + t] = dz * g * (s * (1.0f - s));
    dP[((long)b * 2 * R + R + r) * T + t] = dz * s * (1.0f - g * g);
}

int wn_gate_bwd(const float* dZ, const float* S, const float* Gt, float* dP, int B, int T, int R, wn_stream_t st) {
    WN_PROF("gate_bwd", 0.0, 0.0, st);
    dim3 grid((T + WN_TPB - 1) / WN_TPB, R, B);
    WN_LAUNCH(k_gate_bwd, grid, dim3(WN_TPB), 0, st, dZ, S, Gt, dP, T, R);
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WN_TPB) void k_softmax_ce(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                       float* __restrict__ dlogits, float* __restrict__ loss_partial, int T,
                                                       int Q, int t_start, float grad_scale, float* __restrict__ amax_partial) {
    __shared__ float red[4];
    float my_amax = 0.0f;
    const int t = blockIdx.x * WN_TPB + threadIdx.x;
    const int b = blockIdx.y;
    float my_loss = 0.0f;
    const bool live = (t < T) && (t >= t_start);
    const float* lg = logits + (long)b * Q * T + t;
    if (live) {
        // online softmax in chunks of 8 channels: the 8 strided loads of a chunk are independent and in
        // flight together (one dependent load per iteration made this kernel latency bound)
        float mx = -3.0e38f, sum = 0.0f;
        for (int q0 = 0; q0 < Q; q0 += 8) {
            float v[8];
            WN_UNROLL
            for (int u = 0; u < 8; ++u) v[u] = (q0 + u < Q) ? lg[(long)(q0 + u) * T] : -3.0e38f;
            float cm = v[0];
            WN_UNROLL
            for (int u = 1; u < 8; ++u) cm = fmaxf(cm, v[u]);
            if (cm > mx) {
                sum *= expf(mx - cm);
                mx = cm;
            }
            WN_UNROLL
            for (int u = 0; u < 8; ++u) sum += expf(v[u] - mx);
        }
        long long tg = target[(long)b * T + t] % Q;
        if (tg < 0) tg += Q;
        const float lse = logf(sum) + mx;
        my_loss = lse - lg[(long)tg * T];
        if (dlogits != nullptr) {
            float* dl = dlogits + (long)b * Q * T + t;
            for (int q0 = 0; q0 < Q; q0 += 8) {
                float v[8];
                WN_UNROLL
                for (int u = 0; u < 8; ++u) v[u] = (q0 + u < Q) ? lg[(long)(q0 + u) * T] : 0.0f;
                WN_UNROLL
                for (int u = 0; u < 8; ++u) {
                    if (q0 + u < Q) {
                        float pq = expf(v[u] - lse);
                        if (q0 + u == (int)tg) pq -= 1.0f;
                        dl[(long)(q0 + u) * T] = pq * grad_scale;
                        my_amax = fmaxf(my_amax, fabsf(pq * grad_scale));
                    }
                }
            }
        }
    } else if (t < T && dlogits != nullptr) {
        float* dl = dlogits + (long)b * Q * T + t;
        for (int q = 0; q < Q; ++q) dl[(long)q * T] = 0.0f;
    }
    const float tot = block_reduce_sum(my_loss, red);
    if (threadIdx.x == 0) loss_partial[blockIdx.y * gridDim.x + blockIdx.x] = tot;
    if (amax_partial != nullptr) {
        const float am = block_reduce_max(my_amax, red);
        if (threadIdx.x == 0) amax_partial[blockIdx.y * gridDim.x + blockIdx.x] = am;
    }
}

int wn_softmax_ce_nblocks(int B, int T) { return ((T + WN_TPB - 1) / WN_TPB) * B; }

int wn_softmax_ce(const float* logits, const int64_t* target, float* dlogits, float* loss_partial, int* n_partial, int B,
                  int T, int Q, int t_start, float grad_scale, float* amax_partial, wn_stream_t st) {
    WN_PROF("softmax_ce", 0.0, 0.0, st);
    dim3 grid((T + WN_TPB - 1) / WN_TPB, B);
    if (n_partial) *n_partial = (int)(grid.x * grid.y);
    WN_LAUNCH(k_softmax_ce, grid, dim3(WN_TPB), 0, st, logits, target, dlogits, loss_partial, T, Q, t_start, grad_scale, amax_partial);
    return 0;
}

__global__ __launch_bounds__(WN_TPB) void k_sum_partials(const float* __restrict__ partial, int n, float scale,
                                                         float* __restrict__ out, const float* __restrict__ amax_partial,
                                                         float* __restrict__ amax_out) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += WN_TPB) s += partial[i];
    const float tot = block_reduce_sum(s, red);
    if (threadIdx.x == 0) out[0] = tot * scale;
    if (amax_out != nullptr) {
        float a = 0.0f;
        if (amax_partial != nullptr)
            for (int i = threadIdx.x; i < n; i += WN_TPB) a = fmaxf(a, amax_partial[i]);
        a = block_reduce_max(a, red);
        if (threadIdx.x == 0) amax_out[0] = a;
    }
}

int wn_sum_partials(const float* partial, int n, float scale, float* out, const float* amax_partial, float* amax_out, wn_stream_t st) {
    WN_PROF("sum_partials", 0.0, 0.0, st);
    WN_LAUNCH(k_sum_partials, dim3(1), dim3(WN_TPB), 0, st, partial, n, scale, out, amax_partial, amax_out);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// measured scale of the fp16 pair split of the weight gradients (wn_elem.h)
__global__ __launch_bounds__(WN_TPB) void k_dw_prepare(float* __restrict__ words, float host_mul, const float* __restrict__ scan,
                                                       int n_scan, int headroom) {
    __shared__ float red[4];
    float a;
    if (scan != nullptr) {   // maxima as bit patterns: an inf / NaN partial stays on top (-> a_mul = 1, the overflow redo takes over)
        a = 0.0f;
        for (int i = threadIdx.x; i < n_scan; i += WN_TPB) a = wn_absmax_keep_nan(a, scan[i]);
        int ai = __builtin_bit_cast(int, a);
        WN_UNROLL
        for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(ai, m, 64); ai = o > ai ? o : ai; }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = __builtin_bit_cast(float, ai);
        __syncthreads();
        ai = __builtin_bit_cast(int, red[0]);
        for (int i = 1; i < (WN_TPB >> 6); ++i) { const int o = __builtin_bit_cast(int, red[i]); ai = o > ai ? o : ai; }
        a = __builtin_bit_cast(float, ai);
    } else {
        a = words[2];
    }
    if (threadIdx.x != 0) return;
    int force_redo = 0;
    if (scan != nullptr) words[2] = a;
    float mul = host_mul;
    if (!(host_mul > 0.0f)) {
        const unsigned bits = __builtin_bit_cast(unsigned, a);
        const int ex = (int)((bits >> 23) & 0xffu);
        if (a > 0.0f && ex != 0xff) {
            // 2^(ex - 127) <= a < 2^(ex - 126): e = floor(-log2 a) = 126 - ex for a power of two above... take the safe side:
            // a < 2^(ex - 126) =: 2^-e  ->  e = 126 - ex  (denormals: ex = 0 -> e = 126)
            int e = 126 - ex + headroom;
            e = e > 120 ? 120 : (e < -120 ? -120 : e);
            mul = __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
        } else {
            // no usable maximum (all-zero gradient, inf / NaN, or no loss call of this workspace measured one): the overflow word
            // is raised up front, so the six-product launch behind every fp16 launch does the work -- never a silent underflow
            mul = 1.0f;
            force_redo = 1;
        }
    }
    reinterpret_cast<int*>(words)[0] = force_redo;
    words[1] = mul;
}

int wn_dw_prepare(float* words, float host_mul, const float* scan, int n_scan, int headroom, wn_stream_t st) {
    WN_PROF("dw_prepare", 0.0, 0.0, st);
    WN_LAUNCH(k_dw_prepare, dim3(1), dim3(WN_TPB), 0, st, words, host_mul, scan, n_scan, headroom);
    return 0;
}

__global__ __launch_bounds__(WN_TPB) void k_absmax_rows(const float* __restrict__ p, long rows, long ld, int c0, int ncols,
                                                        float* __restrict__ partial) {
    __shared__ float red[4];
    for (long r = blockIdx.y; r < rows; r += gridDim.y) {   // (more rows than a grid has y blocks: a block walks several)
    __syncthreads();
    const int lo = (int)blockIdx.x * 4096, hi = lo + 4096 < ncols ? lo + 4096 : ncols;
    const float* row = p + r * ld + c0;
    float a = 0.0f;
    if (((reinterpret_cast<uintptr_t>(row + lo) & 15) == 0)) {
        const int n4 = (hi - lo) >> 2;
        const float4* q = reinterpret_cast<const float4*>(row + lo);
        for (int i = threadIdx.x; i < n4; i += WN_TPB) {
            const float4 v = q[i];
            // a NaN must not be dropped: |v| as an integer compare keeps inf / NaN bit patterns on top
            a = wn_absmax_keep_nan(a, v.x); a = wn_absmax_keep_nan(a, v.y); a = wn_absmax_keep_nan(a, v.z); a = wn_absmax_keep_nan(a, v.w);
        }
        for (int i = lo + 4 * n4 + threadIdx.x; i < hi; i += WN_TPB) a = wn_absmax_keep_nan(a, row[i]);
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += WN_TPB) a = wn_absmax_keep_nan(a, row[i]);
    }
    // non-negative floats (and the inf / NaN patterns above them) order like their bit patterns
    int ai = __builtin_bit_cast(int, a);
    WN_UNROLL
    for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(ai, m, 64); ai = o > ai ? o : ai; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = __builtin_bit_cast(float, ai);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = __builtin_bit_cast(int, red[0]);
        for (int i = 1; i < (WN_TPB >> 6); ++i) { const int o = __builtin_bit_cast(int, red[i]); t = o > t ? o : t; }
        partial[r * gridDim.x + blockIdx.x] = __builtin_bit_cast(float, t);
    }
    }
}

int wn_absmax_rows(const float* p, long rows, long ld, int c0, int ncols, float* partial, wn_stream_t st) {
    WN_PROF("dw_absmax_scan", 0.0, (double)rows * ncols * 4.0, st);
    if (rows < 1 || ncols < 1) return 1;
    WN_LAUNCH(k_absmax_rows, dim3((unsigned)((ncols + 4095) / 4096), (unsigned)(rows < 32768 ? rows : 32768)), dim3(WN_TPB), 0, st, p,
              rows, ld, c0, ncols, partial);
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WN_TPB) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, long n, float lr_over_bc1, float sqrt_bc2, float beta1,
                                                 float beta2, float eps, float wd, long skip_lo, long skip_hi) {
    const long stride = (long)gridDim.x * WN_TPB;
    for (long i = (long)blockIdx.x * WN_TPB + threadIdx.x; i < n; i += stride) {
        if (i >= skip_lo && i < skip_hi) continue;
        const float pv = p[i];
        float gv = g[i];
        if (wd != 0.0f) gv += wd * pv;
        const float mv = beta1 * m[i] + (1.0f - beta1) * gv;
        const float vv = beta2 * v[i] + (1.0f - beta2) * gv * gv;
        m[i] = mv;
        v[i] = vv;
        const float denom = sqrtf(vv) / sqrt_bc2 + eps;
        p[i] = pv - lr_over_bc1 * (mv / denom);
    }
}

int wn_adam(float* p, const float* g, float* m, float* v, long n, float lr_over_bc1, float sqrt_bc2, float beta1, float beta2,
            float eps, float weight_decay, long skip_lo, long skip_hi, wn_stream_t st) {
    WN_PROF("adam", 0.0, 0.0, st);
    long nb = (n + WN_TPB - 1) / WN_TPB;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    WN_LAUNCH(k_adam, dim3((unsigned)nb), dim3(WN_TPB), 0, st, p, g, m, v, n, lr_over_bc1, sqrt_bc2, beta1, beta2, eps,
              weight_decay, skip_lo, skip_hi);
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WN_TPB) void k_copy4(float* __restrict__ dst, const float* __restrict__ src, WnCopy4 c) {
    const long total = (long)c.nl * c.n0 * c.n1 * c.n2;
    const long i = (long)blockIdx.x * WN_TPB + threadIdx.x;
    if (i >= total) return;
    const int i2 = (int)(i % c.n2);
    long r = i / c.n2;
    const int i1 = (int)(r % c.n1);
    r /= c.n1;
    const int i0 = (int)(r % c.n0);
    const int l = (int)(r / c.n0);
    dst[i0 * c.d0 + i1 * c.d1 + i2 * c.d2 + l * c.dl] = src[i0 * c.s0 + i1 * c.s1 + i2 * c.s2 + l * c.sl];
}

static __device__ __forceinline__ void copy4_elem(float* __restrict__ dst, const float* __restrict__ src, const WnCopy4& c, long i) {
    const int i2 = (int)(i % c.n2);
    long r = i / c.n2;
    const int i1 = (int)(r % c.n1);
    r /= c.n1;
    const int i0 = (int)(r % c.n0);
    const int l = (int)(r / c.n0);
    dst[i0 * c.d0 + i1 * c.d1 + i2 * c.d2 + l * c.dl] = src[i0 * c.s0 + i1 * c.s1 + i2 * c.s2 + l * c.sl];
}

__global__ __launch_bounds__(WN_TPB) void k_copy4_batch(WnCopy4Batch a) {
    int j = 0;
    while (j + 1 < a.njobs && (int)blockIdx.x >= a.blk0[j + 1]) ++j;  // block-uniform
    const WnCopy4& c = a.c[j];
    const long total = (long)c.nl * c.n0 * c.n1 * c.n2;
    const long i = (long)((int)blockIdx.x - a.blk0[j]) * WN_TPB + threadIdx.x;
    if (i < total) copy4_elem(a.dst[j], a.src[j], c, i);
}

int wn_copy4_batch(WnCopy4Batch* b, wn_stream_t st) {
    WN_PROF("copy4", 0.0, 0.0, st);
    int nblk = 0;
    for (int j = 0; j < b->njobs; ++j) {
        const long total = (long)b->c[j].nl * b->c[j].n0 * b->c[j].n1 * b->c[j].n2;
        b->blk0[j] = nblk;
        nblk += (int)((total + WN_TPB - 1) / WN_TPB);
    }
    b->blk0[b->njobs] = nblk;
    if (nblk <= 0) return 0;
    WN_LAUNCH(k_copy4_batch, dim3((unsigned)nblk), dim3(WN_TPB), 0, st, *b);
    return 0;
}

int wn_copy4(float* dst, const float* src, const WnCopy4* c, wn_stream_t st) {
    WN_PROF("copy4", 0.0, 0.0, st);
    const long total = (long)c->nl * c->n0 * c->n1 * c->n2;
    if (total <= 0) return 0;
    WN_LAUNCH(k_copy4, dim3((unsigned)((total + WN_TPB - 1) / WN_TPB)), dim3(WN_TPB), 0, st, dst, src, *c);
    return 0;
}

__global__ __launch_bounds__(WN_TPB) void k_cvec(WnCvecArgs a) {
    const int l = blockIdx.x;
    const float bup = (a.off_up_b >= 0) ? a.params[a.off_up_b] : 0.0f;
    for (int o = threadIdx.x; o < 2 * a.R; o += WN_TPB) {
        const bool th = o >= a.R;
        const int r = th ? o - a.R : o;
        const float* w = a.params + (th ? a.off_atanh_w : a.off_asig_w) + (long)l * a.ls_aux + (long)r * a.A;
        float rs = 0.0f;
        for (int k = 0; k < a.A; ++k) rs += w[k];
        const float bd = a.params[(th ? a.off_dtanh_b : a.off_dsig_b) + (long)l * a.ls_dil + r];
        const float ba = a.params[(th ? a.off_atanh_b : a.off_asig_b) + (long)l * a.ls_aux + r];
        a.cvec[(long)l * 2 * a.R + o] = (bd + ba) + bup * rs;
        a.rowsum_aux[(long)l * 2 * a.R + o] = rs;
    }
}

int wn_cvec(const WnCvecArgs* a, wn_stream_t st) {
    WN_PROF("cvec", 0.0, 0.0, st);
    WN_LAUNCH(k_cvec, dim3((unsigned)a->L), dim3(WN_TPB), 0, st, *a);
    return 0;
}

__global__ __launch_bounds__(WN_TPB) void k_sum_layers(const float* __restrict__ params, long off, long ls, int L, int n,
                                                       float* __restrict__ out) {
    const int i = blockIdx.x * WN_TPB + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int l = 0; l < L; ++l) s += params[off + (long)l * ls + i];
    out[i] = s;
}

int wn_sum_layers(const float* params, long off, long ls, int L, int n, float* out, wn_stream_t st) {
    WN_PROF("sum_layers", 0.0, 0.0, st);
    WN_LAUNCH(k_sum_layers, dim3((unsigned)((n + WN_TPB - 1) / WN_TPB)), dim3(WN_TPB), 0, st, params, off, ls, L, n, out);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// grid.x: blocks [0, nbg) sum the 16-sample groups of a frame (one thread per (o', f)), block nbg builds
// the dw_partial block of this (l, b): row 0 = sum over frames of qp, the other rows zero.
__global__ __launch_bounds__(WN_TPB) void k_aux_finish(const float* __restrict__ dGp, long dgp_lstride,
                                                       const float* __restrict__ qp, long qp_lstride, float* __restrict__ dG,
                                                       float* __restrict__ dw_partial, int T, int R2, int U, int F, int nbg) {
    const int b = blockIdx.y, l = blockIdx.z, nb = gridDim.y;
    const int H = T >> 4, per = U >> 4;
    if ((int)blockIdx.x < nbg) {
        const int idx = blockIdx.x * WN_TPB + threadIdx.x;
        if (idx >= R2 * F) return;
        const int o = idx / F, f = idx - o * F;
        const float* src = dGp + (long)l * dgp_lstride + ((long)b * R2 + o) * H + (long)f * per;
        float s = 0.0f;
        for (int i = 0; i < per; ++i) s += src[i];
        dG[(long)l * nb * R2 * F + ((long)b * R2 + o) * F + f] = s;
    } else {
        // ONE workgroup per (l, b): row 0 of its dw_partial block = sum over the frames of qp (per phase j), the other rows
        // zero.  The frames are dealt to WN_TPB / U thread groups in contiguous ranges (8 loads in flight each) and the range
        // sums added in group order: one thread per phase walking all F frames was 36 dependent rounds of loads on 80
        // threads -- the critical path of this launch.
        __shared__ float part[WN_TPB];
        const int tid = threadIdx.x;
        float* dst = dw_partial + (long)l * nb * R2 * U + (long)b * R2 * U;
        const float* src0 = qp + (long)l * qp_lstride + (long)b * T;
        if (U <= WN_TPB) {
            const int np = WN_TPB / U, p = tid / U, j = tid - p * U;
            const int Fp = (F + np - 1) / np;
            const int f_lo = p * Fp, f_hi = (f_lo + Fp < F) ? f_lo + Fp : F;
            float s = 0.0f;
            if (p < np) {
                const float* src = src0 + j;
                for (int f0 = f_lo; f0 < f_hi; f0 += 8) {  // summed in frame order
                    float v[8];
                    WN_UNROLL
                    for (int u = 0; u < 8; ++u) v[u] = (f0 + u < f_hi) ? src[(long)(f0 + u) * U] : 0.0f;
                    WN_UNROLL
                    for (int u = 0; u < 8; ++u) s += v[u];
                }
            }
            part[tid] = s;
            __syncthreads();
            if (p == 0) {
                float t = part[j];
                for (int q = 1; q < np; ++q) t += part[q * U + j];
                dst[j] = t;
            }
        } else {
            for (int j = tid; j < U; j += WN_TPB) {
                const float* src = src0 + j;
                float s = 0.0f;
                for (int f0 = 0; f0 < F; f0 += 8) {
                    float v[8];
                    WN_UNROLL
                    for (int u = 0; u < 8; ++u) v[u] = (f0 + u < F) ? src[(long)(f0 + u) * U] : 0.0f;
                    WN_UNROLL
                    for (int u = 0; u < 8; ++u) s += v[u];
                }
                dst[j] = s;
            }
        }
        for (int idx = U + tid; idx < R2 * U; idx += WN_TPB) dst[idx] = 0.0f;
    }
}

int wn_aux_finish(const float* dGp, long dgp_lstride, const float* qp, long qp_lstride, float* dG, float* dw_partial, int B,
                  int T, int R2, int U, int F, int nl, wn_stream_t st) {
    WN_PROF("aux_finish", 0.0, (double)nl * B * ((double)R2 * (T / 16) + T) * 4.0, st);
    if (U < 16 || (U & 15) || (long)U * F != T) return 1;
    const int nbg = (R2 * F + WN_TPB - 1) / WN_TPB, nbw = 1;
    WN_LAUNCH(k_aux_finish, dim3((unsigned)(nbg + nbw), (unsigned)B, (unsigned)nl), dim3(WN_TPB), 0, st, dGp, dgp_lstride, qp,
              qp_lstride, dG, dw_partial, T, R2, U, F, nbg);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// NJ = phases per lane (U <= 64*NJ), UNR = frames in flight per wave
template <int NJ, int UNR>
__global__ __launch_bounds__(WN_TPB) void k_aux_bwd(const float* __restrict__ dP, long dp_lstride, const float* __restrict__ G,
                                                    long g_bstride, const float* __restrict__ upw, float* __restrict__ dG,
                                                    float* __restrict__ dw_partial, int T, int R2, int U, int F) {
    constexpr int nj = NJ;
    // One WAVE per (row o', batch b, layer l): lanes run along the phase j = t % U (coalesced row
    // segments of U floats per frame), UNR frames are in flight at a time.  Per frame the wave
    // reduces sum_j w[j] dP[fU+j] (-> dG[f]); per lane it accumulates dP[fU+j] * G[f] over the frames
    // (-> dw[j]).  No LDS, no block barriers: the kernel is a pure HBM stream of dP.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o = blockIdx.x * 4 + wave;
    const int b = blockIdx.y, l = blockIdx.z, nb = gridDim.y;
    if (o >= R2) return;  // whole wave leaves together (o is wave-uniform)
    const float* row = dP + (long)l * dp_lstride + ((long)b * R2 + o) * T;
    const float* grow = G + (long)b * g_bstride + ((long)l * R2 + o) * F;
    float* dgrow = dG + (long)l * nb * R2 * F + ((long)b * R2 + o) * F;
    float* dwrow = dw_partial + (long)l * nb * R2 * U + ((long)b * R2 + o) * U;
    float wj[NJ], accw[NJ];
    WN_UNROLL
    for (int i = 0; i < NJ; ++i) {
        const int j = lane + 64 * i;
        wj[i] = (i < nj && j < U) ? upw[j] : 0.0f;
        accw[i] = 0.0f;
    }
    for (int f0 = 0; f0 < F; f0 += UNR) {
        float v[UNR][NJ];
        float gf[UNR];
        WN_UNROLL
        for (int u = 0; u < UNR; ++u) {
            const int f = f0 + u;
            const bool fok = f < F;
            gf[u] = fok ? grow[f] : 0.0f;
            WN_UNROLL
            for (int i = 0; i < NJ; ++i) {
                const int j = lane + 64 * i;
                v[u][i] = (i < nj && fok && j < U) ? row[(long)f * U + j] : 0.0f;
            }
        }
        WN_UNROLL
        for (int u = 0; u < UNR; ++u) {
            float sdot = 0.0f;
            WN_UNROLL
            for (int i = 0; i < NJ; ++i) {
                if (i < nj) {
                    accw[i] += v[u][i] * gf[u];
                    sdot += wj[i] * v[u][i];
                }
            }
            sdot = wave_reduce_sum(sdot);
            if (lane == 0 && f0 + u < F) dgrow[f0 + u] = sdot;
        }
    }
    WN_UNROLL
    for (int i = 0; i < NJ; ++i) {
        const int j = lane + 64 * i;
        if (i < nj && j < U) dwrow[j] = accw[i];
    }
}

int wn_aux_bwd(const float* dP, long dp_lstride, const float* G, long g_bstride, const float* upw, float* dG, float* dw_partial,
               int B, int T, int R2, int U, int F, int nl, wn_stream_t st) {
    WN_PROF("aux_bwd", 0.0, (double)nl * B * R2 * T * 4.0, st);
    if ((long)U * F != T) return 1;
    const int nj = (U + 63) / 64;
    if (nj > 16) return 2;
    const dim3 grid((unsigned)((R2 + 3) / 4), (unsigned)B, (unsigned)nl), block(WN_TPB);
    if (nj <= 1) {
        WN_LAUNCH((k_aux_bwd<1, 16>), grid, block, 0, st, dP, dp_lstride, G, g_bstride, upw, dG, dw_partial, T, R2, U, F);
    } else if (nj <= 2) {
        WN_LAUNCH((k_aux_bwd<2, 8>), grid, block, 0, st, dP, dp_lstride, G, g_bstride, upw, dG, dw_partial, T, R2, U, F);
    } else if (nj <= 4) {
        WN_LAUNCH((k_aux_bwd<4, 4>), grid, block, 0, st, dP, dp_lstride, G, g_bstride, upw, dG, dw_partial, T, R2, U, F);
    } else {
        WN_LAUNCH((k_aux_bwd<16, 2>), grid, block, 0, st, dP, dp_lstride, G, g_bstride, upw, dG, dw_partial, T, R2, U, F);
    }
    return 0;
}

// 32 outputs x 8 z-lanes per workgroup; the 8 partial sums are combined in a fixed order, so the
// result is deterministic.  grid.y > 1 = first level of a two-level reduction (raw sums of one
// z-chunk each into scratch[chunk][MN]); the mapped/scaled write happens in the last level.
__global__ __launch_bounds__(WN_TPB) void k_reduce(WnReduceArgs a, const float* __restrict__ src, int nz, int zchunk,
                                                   float* __restrict__ raw_out) {
    __shared__ float red[8][33];
    const long mn = (long)a.M * a.N;
    const int il = threadIdx.x & 31, zg = threadIdx.x >> 5;
    const int l = blockIdx.z;
    src += (long)l * nz * mn;
    const long i = (long)blockIdx.x * 32 + il;
    const int zbeg = blockIdx.y * zchunk;
    int zend = zbeg + zchunk;
    if (zend > nz) zend = nz;
    float s = 0.0f;
    if (i < mn)
        for (int z = zbeg + zg; z < zend; z += 8) s += src[(long)z * mn + i];
    red[zg][il] = s;
    __syncthreads();
    if (zg != 0 || i >= mn) return;
    s = 0.0f;
    WN_UNROLL
    for (int q = 0; q < 8; ++q) s += red[q][il];
    if (raw_out != nullptr) {
        raw_out[(long)blockIdx.y * mn + i] = s;
        return;
    }
    s *= a.scale;
    const int m = (int)(i / a.N), n = (int)(i % a.N);
    if (a.addend_m != nullptr) {
        const float sc = a.addend_scale_ptr ? a.addend_scale_ptr[0] : 1.0f;
        s += a.addend_m[(long)l * a.addend_lstride + m] * sc;
    }
    const long o = (long)l * a.out_lstride + (long)(m / a.m_seg) * a.m_seg_stride + (long)(m % a.m_seg) * a.m_stride +
                   (long)(n / a.n_seg) * a.n_seg_stride + (long)(n % a.n_seg) * a.n_stride;
    if (a.accumulate) s += a.out[o];
    a.out[o] = s;
}

int wn_reduce(const WnReduceArgs* a, wn_stream_t st) {
    WN_PROF("reduce_partials", 0.0, (double)(a->nl > 0 ? a->nl : 1) * a->nz * a->M * a->N * 4.0, st);
    const long mn = (long)a->M * a->N;
    if (mn <= 0 || a->m_seg <= 0 || a->n_seg <= 0) return 1;
    const unsigned gx = (unsigned)((mn + 31) / 32);
    // two levels when there are few outputs but very many partials (keeps the machine busy)
    const int nl = a->nl > 0 ? a->nl : 1;
    if (nl == 1 && a->scratch != nullptr && a->nz >= 512 && gx < 512) {
        int zchunk = 128;
        const int nchunk = (a->nz + zchunk - 1) / zchunk;
        if ((long)nchunk * mn <= a->scratch_floats) {
            WN_LAUNCH(k_reduce, dim3(gx, (unsigned)nchunk), dim3(WN_TPB), 0, st, *a, a->partial, a->nz, zchunk, a->scratch);
            WN_LAUNCH(k_reduce, dim3(gx, 1), dim3(WN_TPB), 0, st, *a, (const float*)a->scratch, nchunk, nchunk, (float*)nullptr);
            return 0;
        }
    }
    WN_LAUNCH(k_reduce, dim3(gx, 1, (unsigned)nl), dim3(WN_TPB), 0, st, *a, a->partial, a->nz, a->nz, (float*)nullptr);
    return 0;
}

__global__ __launch_bounds__(WN_TPB) void k_dot(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                float* __restrict__ out, int accumulate) {
    __shared__ float red[4];
    float s = 0.0f;
    for (long i = threadIdx.x; i < n; i += WN_TPB) s += a[i] * b[i];
    const float tot = block_reduce_sum(s, red);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + tot : tot;
}

int wn_dot(const float* a, const float* b, long n, float* out, int accumulate, wn_stream_t st) {
    WN_PROF("dot", 0.0, 0.0, st);
    WN_LAUNCH(k_dot, dim3(1), dim3(WN_TPB), 0, st, a, b, n, out, accumulate);
    return 0;
}

__global__ __launch_bounds__(WN_TPB) void k_fill(float* __restrict__ p, float v, long n) {
    const long i = (long)blockIdx.x * WN_TPB + threadIdx.x;
    if (i < n) p[i] = v;
}

int wn_fill(float* p, float v, long n, wn_stream_t st) {
    WN_PROF("fill", 0.0, 0.0, st);
    if (n <= 0) return 0;
    WN_LAUNCH(k_fill, dim3((unsigned)((n + WN_TPB - 1) / WN_TPB)), dim3(WN_TPB), 0, st, p, v, n);
    return 0;
}

// zero the first `ncols` (a multiple of 4) columns of `rows` rows that are `stride` floats apart: 16 bytes per lane
__global__ __launch_bounds__(WN_TPB) void k_fill_cols(float* __restrict__ p, long rows, long stride, int ncols4) {
    const long i = (long)blockIdx.x * WN_TPB + threadIdx.x;
    if (i >= rows * ncols4) return;
    const long r = i / ncols4;
    const int c = (int)(i - r * ncols4);
    float* q = p + r * stride + 4 * c;
    q[0] = 0.0f; q[1] = 0.0f; q[2] = 0.0f; q[3] = 0.0f;
}

int wn_fill_cols(float* p, long rows, long stride, int ncols, wn_stream_t st) {
    WN_PROF("fill_cols", 0.0, (double)rows * ncols * 4, st);
    if (rows <= 0 || ncols <= 0) return 0;
    if (ncols % 4 != 0) return 1;
    const long n = rows * (ncols / 4);
    WN_LAUNCH(k_fill_cols, dim3((unsigned)((n + WN_TPB - 1) / WN_TPB)), dim3(WN_TPB), 0, st, p, rows, stride, ncols / 4);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// front-conv weight gradient as a scatter (see wn_elem.h)
// ---------------------------------------------------------------------------------------------
static bool front_dw_mfma_ok(int R, int K, int Q);
// row groups of the matrix-core kernel: a workgroup contracts 64 rows of dX0 (all of them up to 64 channels)
static int front_dw_row_groups(int R, int K, int Q) { return (front_dw_mfma_ok(R, K, Q) && R > 64) ? R / 64 : 1; }
static void front_dw_grid(int B, int T, int RG, int* nchunk, int* chunk) {
    int nc = 256 / ((B > 0 ? B : 1) * RG);  // one workgroup per CU (the table fills its LDS)
    if (nc < 1) nc = 1;
    int ch = (T + nc - 1) / nc;
    ch = (ch + 63) / 64 * 64;
    if (ch < 64) ch = 64;
    *chunk = ch;
    *nchunk = (T + ch - 1) / ch;
}

// the matrix-core kernel (small LDS), or the scatter kernel with its [R][K*Q] table in LDS
int wn_front_dw_supported(int R, int K, int Q) {
    return (front_dw_mfma_ok(R, K, Q) || ((long)R * K * Q + R) * 4 <= 150 * 1024) && K <= 8;
}

long wn_front_dw_partial_floats(int B, int T, int R, int K, int Q) {
    int nc, ch;
    front_dw_grid(B, T, front_dw_row_groups(R, K, Q), &nc, &ch);
    return (long)B * nc * ((long)R * K * Q + R);
}

#define FD_T 1024            // threads of k_front_dw_scatter
#define FD_W (FD_T / 64)      // its waves
__global__ __launch_bounds__(FD_T) void k_front_dw_scatter(const float* __restrict__ dX0, const int64_t* __restrict__ x,
                                                          float* __restrict__ partial, int T, int R, int K, int Q, int chunk) {
    WN_DYN_SMEM(smem_raw);
    float* acc = reinterpret_cast<float*>(smem_raw);  // [R][K*Q]
    const int KQ = K * Q;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * chunk;
    const int t1 = (t0 + chunk < T) ? t0 + chunk : T;
    const int ntab = R * KQ + R, ntab4 = ntab >> 2;
    for (int i = tid; i < ntab4; i += FD_T) reinterpret_cast<float4*>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 4 * ntab4 + tid; i < ntab; i += FD_T) acc[i] = 0.0f;
    __syncthreads();
    const int64_t* xb = x + (long)b * T;
    const float* db = dX0 + (long)b * R * T;
    float* out = partial + ((long)b * gridDim.x + blockIdx.x) * ((long)R * KQ + R);
    // wave w owns channels w, w+FD_W, w+2 FD_W, ...: no two waves ever touch the same table row (and the order of the
    // additions into one table entry -- time strips in sequence, lanes in hardware order -- does not depend on the
    // number of waves).  16 waves per CU instead of 4 hide the LDS-atomic latency.  The token
    // columns of a 64-step strip are looked up once and reused for all channels of the wave.
    float* rsum = acc + R * KQ;  // [R] row sums (bias gradient)
    // FD_S strips of 64 time steps are in flight together: their token columns and their (up to 8) channel rows are all
    // requested before the first LDS atomic -- one memory latency per FD_S strips instead of two per strip (the atomics
    // order the loop, so the compiler cannot overlap strips on its own).  The bias gradient (row sums of dX0) is
    // accumulated from the same registers: per lane and channel slot, reduced across the wave at the end.
    constexpr int FD_S = 4;
    float rs[8];
    WN_UNROLL
    for (int u = 0; u < 8; ++u) rs[u] = 0.0f;
    for (int ts = t0; ts < t1; ts += 64 * FD_S) {
        int col[FD_S][8];
        bool okS[FD_S];
        WN_UNROLL
        for (int sidx = 0; sidx < FD_S; ++sidx) {
            const int t = ts + 64 * sidx + lane;
            const bool ok = t < t1;
            okS[sidx] = ok;
            WN_UNROLL
            for (int k = 0; k < 8; ++k) {
                col[sidx][k] = -1;
                const int tq = t - (K - 1 - k);
                if (k < K && ok && tq >= 0) {
                    long long q = xb[tq];
                    if ((unsigned long long)q >= (unsigned long long)Q) {   // OneHot takes indices modulo Q (wavenet.py:88); rare
                        q %= Q;
                        if (q < 0) q += Q;
                    }
                    col[sidx][k] = k * Q + (int)q;
                }
            }
        }
        for (int c0 = wave; c0 < R; c0 += 8 * FD_W) {  // up to 8 channels of this wave per step
            float v[FD_S][8];
            WN_UNROLL
            for (int sidx = 0; sidx < FD_S; ++sidx) {
                const int t = ts + 64 * sidx + lane;
                WN_UNROLL
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + FD_W * u;
                    v[sidx][u] = (okS[sidx] && c < R) ? db[(long)c * T + t] : 0.0f;
                }
            }
            WN_UNROLL
            for (int sidx = 0; sidx < FD_S; ++sidx) {
                WN_UNROLL
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + FD_W * u;
                    if (c < R) {
                        if (c0 == wave) rs[u] += v[sidx][u];   // R <= 8 FD_W: one slot per channel (else the second pass below)
                        WN_UNROLL
                        for (int k = 0; k < 8; ++k)
                            if (col[sidx][k] >= 0) atomicAdd(&acc[c * KQ + col[sidx][k]], v[sidx][u]);
                    }
                }
            }
        }
    }
    if (R <= 8 * FD_W) {
        WN_UNROLL
        for (int u = 0; u < 8; ++u) {
            const int c = wave + FD_W * u;
            const float tot = wave_reduce_sum(rs[u]);
            if (lane == 0 && c < R) rsum[c] = tot;
        }
    } else {
        // bias gradient: row sums of this block's time range (second pass over the same cache lines)
        for (int c = wave; c < R; c += FD_W) {
            float r1 = 0.0f;
            for (int ts = t0; ts < t1; ts += 256) {
                float v[4];
                WN_UNROLL
                for (int u = 0; u < 4; ++u) {
                    const int t = ts + 64 * u + lane;
                    v[u] = t < t1 ? db[(long)c * T + t] : 0.0f;
                }
                r1 += (v[0] + v[1]) + (v[2] + v[3]);
            }
            r1 = wave_reduce_sum(r1);
            if (lane == 0) rsum[c] = r1;
        }
    }
    __syncthreads();
    if ((((long)R * KQ + R) & 3) == 0) {  // per-block slabs stay 16-byte aligned
        for (int i = tid; i < ntab4; i += FD_T) reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(acc)[i];
    } else {
        for (int i = tid; i < ntab; i += FD_T) out[i] = acc[i];
    }
}

// ---------------------------------------------------------------------------------------------
// The same partial tables on the matrix cores, without atomics (R = 32 or a multiple of 64 -- blockIdx.z = group of 64 rows --,
// K Q <= 1024: the benchmark's front conv, and the recipes' n_resch = 512, where the one-hot contraction through k_gemm6_dw took
// 0.9 ms per step).
// LDS float atomics retire about one lane per 3 cycles on gfx950: the 1 536 wave-atomics of a k_front_dw_scatter workgroup
// cost 125 us, two thirds of that launch (profiles/r02/front_dw_probe.txt).  Here the table of a workgroup's time chunk is
//      acc[c][(k, q)] = sum_t dX0[c][t] * onehot(x[t - (K-1-k)])[q]
// as a contraction over t: A = dX0 (R x 32 time steps per iteration) in the 3-way bf16 split -- every thread splits ONE pair
// of values and leaves the pieces in LDS in the A-fragment layout -- and B = the one-hot columns, which a lane builds in
// registers from the chunk's token indices (kept in LDS as ints): 1.0 is exact in bf16, so three MFMAs per fragment
// (h, m, l pieces of A) give the fp32 result; wave w owns the 32 columns [32 w, 32 w + 32) of the table (all of one tap) in
// 2 accumulator tiles.  Deterministic (no atomics), same partial layout as the scatter kernel.
#define FM_T 1024
__global__ __launch_bounds__(FM_T) void k_front_dw_mfma(const float* __restrict__ dX0, const int64_t* __restrict__ x,
                                                       float* __restrict__ partial, int T, int R, int K, int Q, int chunk) {
    WN_DYN_SMEM(smem_raw);
    // stage[buf][kblock][piece][64 rows][32 B]  (2 x 2 x 3 x 2 KB = 24 KB), then the chunk's tokens as ints
    char* stage = smem_raw;
    int* tok = reinterpret_cast<int*>(smem_raw + 2 * 2 * 3 * 2048);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * chunk;
    const int t1 = (t0 + chunk < T) ? t0 + chunk : T;
    const int KQ = K * Q;
    const int r0 = 64 * (int)blockIdx.z;   // this workgroup's rows of dX0 / of the table
    const int64_t* xb = x + (long)b * T;
    const float* db = dX0 + ((long)b * R + r0) * T;
    float* out = partial + ((long)b * gridDim.x + blockIdx.x) * ((long)R * KQ + R);
    // tokens of positions [t0 - (K-1), t1): index i <-> position t0 - (K-1) + i; -1 in front of the sequence
    for (int i = tid; i < (t1 - t0) + K - 1; i += FM_T) {
        const int tq = t0 - (K - 1) + i;
        int v = -1;
        if (tq >= 0) {
            long long q = xb[tq];
            if ((unsigned long long)q >= (unsigned long long)Q) {   // OneHot takes indices modulo Q (wavenet.py:88); rare
                q %= Q;
                if (q < 0) q += Q;
            }
            v = (int)q;
        }
        tok[i] = v;
    }
    // phase-1 role of this thread: one pair of time steps of one row per iteration
    const int prow = tid >> 4, pkq = tid & 15;
    const int pkb = pkq >> 3, phi = (pkq & 7) >> 2, pdw = pkq & 3;
    const int pofs = pkb * (3 * 2048) + (prow * 2 + phi) * 16 + pdw * 4;   // + piece * 2048 + buf * 2 * 3 * 2048
    const float* prowp = db + (long)prow * T;
    float rsum = 0.0f;
    // phase-2 role of this wave: column tiles `wave` and `wave + 16` (32 columns of one tap each), if the table has that many
    // (K * Q <= 512: one tile per wave; kernel_size 3 with 256 classes: 24 tiles, the first 8 waves take two)
    constexpr int NW = FM_T / 64;
    bool active[2];
    int col[2], q[2], tshift[2];
    WN_UNROLL
    for (int ct = 0; ct < 2; ++ct) {
        active[ct] = (wave + NW * ct) * 32 < KQ;
        col[ct] = (wave + NW * ct) * 32 + li;
        const int tap = col[ct] / Q;
        q[ct] = col[ct] - tap * Q;
        tshift[ct] = tap;   // token index of position t with this tap: (t - t0) + (K-1) - (K-1-tap) = (t - t0) + tap
    }
    f32x16 acc[2][2];
    WN_UNROLL
    for (int ct = 0; ct < 2; ++ct) {
        acc[ct][0] = f32x16_zero();
        acc[ct][1] = f32x16_zero();
    }
    const int nrt = (R - r0) >= 64 ? 2 : (R - r0) >> 5;
    auto stage_pair = [&](int ts, int buf) {
        const int t = ts + 2 * pkq;
        const float x0 = (r0 + prow < R && t < t1) ? prowp[t] : 0.0f;
        const float x1 = (r0 + prow < R && t + 1 < t1) ? prowp[t + 1] : 0.0f;
        rsum += x0 + x1;
        const unsigned h = wn_pk_bf16(x0, x1);
        const float r0 = x0 - wn_bits_f32(h << 16), r1 = x1 - wn_bits_f32(h & 0xffff0000u);
        const unsigned m = wn_pk_bf16(r0, r1);
        const unsigned l = wn_pk_bf16(r0 - wn_bits_f32(m << 16), r1 - wn_bits_f32(m & 0xffff0000u));
        char* d = stage + buf * (2 * 3 * 2048) + pofs;
        *reinterpret_cast<unsigned*>(d) = h;
        *reinterpret_cast<unsigned*>(d + 2048) = m;
        *reinterpret_cast<unsigned*>(d + 4096) = l;
    };
    stage_pair(t0, 0);
    __syncthreads();   // tokens and the first stage
    int buf = 0;
    for (int ts = t0; ts < t1; ts += 32, buf ^= 1) {
        if (ts + 32 < t1) stage_pair(ts + 32, buf ^ 1);
        WN_UNROLL
        for (int ct = 0; ct < 2; ++ct) {
            if (!active[ct]) continue;   // (wave-uniform)
            WN_UNROLL
            for (int kb = 0; kb < 2; ++kb) {
                // one-hot B fragment: this lane's column q against the tokens of its 8 time steps
                const int* tk = tok + (ts - t0) + 16 * kb + 8 * hi + tshift[ct];
                unsigned bq[4];
                WN_UNROLL
                for (int e = 0; e < 4; ++e)
                    bq[e] = (tk[2 * e] == q[ct] ? 0x3F80u : 0u) | (tk[2 * e + 1] == q[ct] ? 0x3F800000u : 0u);
                wn_f4 bf;
                bf.x = wn_bits_f32(bq[0]); bf.y = wn_bits_f32(bq[1]); bf.z = wn_bits_f32(bq[2]); bf.w = wn_bits_f32(bq[3]);
                const char* sa = stage + buf * (2 * 3 * 2048) + kb * (3 * 2048) + (li * 2 + hi) * 16;
                WN_UNROLL
                for (int rt = 0; rt < 2; ++rt) {
                    if (rt < nrt) {
                        const wn_f4 al = *reinterpret_cast<const wn_f4*>(sa + 4096 + rt * 1024);
                        const wn_f4 am = *reinterpret_cast<const wn_f4*>(sa + 2048 + rt * 1024);
                        const wn_f4 ah = *reinterpret_cast<const wn_f4*>(sa + rt * 1024);
                        acc[ct][rt] = mfma_bf16(al, bf, acc[ct][rt]);   // small pieces first
                        acc[ct][rt] = mfma_bf16(am, bf, acc[ct][rt]);
                        acc[ct][rt] = mfma_bf16(ah, bf, acc[ct][rt]);
                    }
                }
            }
        }
        __syncthreads();
    }
    WN_UNROLL
    for (int ct = 0; ct < 2; ++ct) {
        if (!active[ct]) continue;
        WN_UNROLL
        for (int rt = 0; rt < 2; ++rt) {
            if (rt < nrt) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) out[(long)(r0 + 32 * rt + mfma32_row(r, hi)) * KQ + col[ct]] = acc[ct][rt][r];
            }
        }
    }
    // bias gradient: the 16 threads of a row are adjacent lanes
    for (int m = 1; m < 16; m <<= 1) rsum += __shfl_xor(rsum, m, 64);
    if (pkq == 0 && r0 + prow < R) out[(long)R * KQ + r0 + prow] = rsum;
}

static bool front_dw_mfma_ok(int R, int K, int Q) {
    // (other shapes keep the LDS-atomic scatter kernel: 0.18 vs 0.07 ms at the benchmark's size, profiles/r02/front_dw_probe.txt)
    return (R == 32 || R % 64 == 0) && R <= 64 * 1024 && Q % 32 == 0 && K * Q <= 1024 && K <= 8;
}

// dW[c][q][k] = sum_blk partial[blk][c][k*Q+q] ; db[c] = sum_blk partial[blk][R*KQ + c]
// 64 outputs per workgroup (one per lane: coalesced rows of the partial slabs), the partial slabs dealt to the 4 waves in
// contiguous ranges, 8 loads in flight per thread; the 4 range sums are added in wave order -- a fixed order, whatever the
// grid.  (One thread per output walking all 256 slabs was latency bound on half the CUs: 129 workgroups x 32 dependent
// rounds of loads, most of the 0.18 ms of the front-conv weight gradient.)
__global__ __launch_bounds__(256) void k_front_dw_reduce(const float* __restrict__ partial, int nblk, float* __restrict__ dW,
                                                         float* __restrict__ db, int R, int K, int Q) {
    __shared__ float part[4][64];
    const int KQ = K * Q;
    const long per = (long)R * KQ + R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + lane;
    const int per_w = (nblk + 3) / 4;
    const int b_lo = wave * per_w, b_hi = (b_lo + per_w < nblk) ? b_lo + per_w : nblk;
    float s = 0.0f;
    if (i < per) {
        for (int b0 = b_lo; b0 < b_hi; b0 += 8) {  // 8 independent loads in flight, summed in slab order
            float v[8];
            WN_UNROLL
            for (int u = 0; u < 8; ++u) v[u] = (b0 + u < b_hi) ? partial[(long)(b0 + u) * per + i] : 0.0f;
            WN_UNROLL
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave != 0 || i >= per) return;
    s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (i < (long)R * KQ) {
        const int c = (int)(i / KQ), r = (int)(i % KQ);
        const int k = r / Q, q = r % Q;
        dW[((long)c * Q + q) * K + k] = s;
    } else {
        db[i - (long)R * KQ] = s;
    }
}

int wn_front_dw(const float* dX0, const int64_t* x, float* partial, float* dW, float* db, int B, int T, int R, int K, int Q,
                wn_stream_t st) {
    WN_PROF("dw_front_scatter", 0.0, (double)B * R * T * 4.0, st);
    if (!wn_front_dw_supported(R, K, Q)) return 1;
    int nc, ch;
    const int RG = front_dw_row_groups(R, K, Q);
    front_dw_grid(B, T, RG, &nc, &ch);
    const size_t lds = ((size_t)R * K * Q + R) * 4;
#ifndef WN_EMU
    if (!front_dw_mfma_ok(R, K, Q) && lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_dw_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return 2;
#endif
    if (front_dw_mfma_ok(R, K, Q)) {
        const size_t lds_m = 2 * 2 * 3 * 2048 + (size_t)(ch + K) * 4;
        WN_LAUNCH(k_front_dw_mfma, dim3((unsigned)nc, (unsigned)B, (unsigned)RG), dim3(FM_T), lds_m, st, dX0, x, partial, T, R, K, Q, ch);
    } else
    WN_LAUNCH(k_front_dw_scatter, dim3((unsigned)nc, (unsigned)B), dim3(FD_T), lds, st, dX0, x, partial, T, R, K, Q, ch);
    const long per = (long)R * K * Q + R;
    WN_LAUNCH(k_front_dw_reduce, dim3((unsigned)((per + 63) / 64)), dim3(256), 0, st, partial, nc * B, dW, db, R, K, Q);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// (B, R, C) -> (B, C, R): 32 x 32 tiles through LDS (rows padded to 33 words), both sides in full 128-byte segments.  The
// autograd bridge's gradient arrives as (B, T, Q) (the reference's logits layout, wavenet.py:522); the kernels take (B, Q, T).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose_last2(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const long zb = (long)blockIdx.z * R * C;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    WN_UNROLL
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < R && c < C) ? src[zb + (long)r * C + c] : 0.0f;
    }
    __syncthreads();
    WN_UNROLL
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < R && c < C) dst[zb + (long)c * R + r] = tile[tx][ty + 8 * i];
    }
}

int wn_transpose_last2(const float* src, float* dst, int B, int R, int C, wn_stream_t st) {
    WN_PROF("transpose", 0.0, (double)B * R * C * 8.0, st);
    WN_LAUNCH(k_transpose_last2, dim3((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32), (unsigned)B), dim3(256), 0, st, src, dst, R, C);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// any-size decode helpers (see wn_elem.h)
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ long dl_queue_off(int l, int depth, int K, int R) {
    const long cyc = l / depth, in = l % depth;
    return (long)R * (K - 1) * (cyc * ((1L << depth) - 1) + ((1L << in) - 1));
}

__global__ __launch_bounds__(WN_TPB) void k_dl_inputs(WnDlArgs a) {
    const int nb = a.nb, R = a.R, K = a.K;
    const long n_hist = (long)a.L * (K - 1) * R * nb, n_aux = (long)a.nG * nb, n_front = (long)R * nb;
    const long i = (long)blockIdx.x * WN_TPB + threadIdx.x;
    if (i < n_hist) {
        const int u = (int)(i % nb);
        long r = i / nb;
        const int c = (int)(r % R);
        r /= R;
        const int j = (int)(r % (K - 1)), l = (int)(r / (K - 1));
        const int d = 1 << (l % a.depth), Dq = (K - 1) * d;
        int slot = (a.p - (K - 1 - j) * d) % Dq;
        if (slot < 0) slot += Dq;  // not written yet in this run: zero history
        a.xin[(((long)l * K + j) * R + c) * nb + u] = a.queues[(dl_queue_off(l, a.depth, K, R) + (long)slot * R + c) * nb + u];
    } else if (i < n_hist + n_aux) {
        const long q = i - n_hist;
        const int u = (int)(q % nb), row = (int)(q / nb);
        const int t = a.p > a.n_pad ? a.p - a.n_pad : 0;  // replicated first column inside the left padding
        int f = t / a.Ue;
        const float w = a.upw[t - f * a.Ue];
        if (f > a.F - 1) f = a.F - 1;
        a.gstep[q] = w * a.G[((long)u * a.F + f) * a.nG + row];
    } else if (i < n_hist + n_aux + n_front) {
        const long q = i - n_hist - n_aux;
        const int u = (int)(q % nb), c = (int)(q / nb);
        float x0 = a.params[a.off_causal_b + c];
        for (int k = 0; k < K; ++k) {
            const int tq = a.p - (K - 1 - k);
            if (tq >= 0) {
                long long tok = a.samples[(long)u * a.Ttot + tq] % a.Q;
                if (tok < 0) tok += a.Q;
                x0 += a.params[a.off_causal_w + ((long)c * a.Q + tok) * K + k];
            }
        }
        a.xin[((long)(K - 1) * R + c) * nb + u] = x0;
    }
}

int wn_dl_inputs(const WnDlArgs* a, wn_stream_t st) {
    WN_PROF("dl_inputs", 0.0, 0.0, st);
    const long n = (long)a->L * (a->K - 1) * a->R * a->nb + (long)a->nG * a->nb + (long)a->R * a->nb;
    WN_LAUNCH(k_dl_inputs, dim3((unsigned)((n + WN_TPB - 1) / WN_TPB)), dim3(WN_TPB), 0, st, *a);
    return 0;
}

__global__ __launch_bounds__(WN_TPB) void k_dl_push(WnDlArgs a) {
    const int nb = a.nb, R = a.R, K = a.K;
    const long n = (long)a.L * R * nb;
    const long i = (long)blockIdx.x * WN_TPB + threadIdx.x;
    if (i >= n || K < 2) return;
    const int u = (int)(i % nb);
    long r = i / nb;
    const int c = (int)(r % R), l = (int)(r / R);
    const int Dq = (K - 1) << (l % a.depth);
    a.queues[(dl_queue_off(l, a.depth, K, R) + (long)(a.p % Dq) * R + c) * nb + u] = a.xin[(((long)l * K + (K - 1)) * R + c) * nb + u];
}

int wn_dl_push(const WnDlArgs* a, wn_stream_t st) {
    WN_PROF("dl_push", 0.0, 0.0, st);
    const long n = (long)a->L * a->R * a->nb;
    WN_LAUNCH(k_dl_push, dim3((unsigned)((n + WN_TPB - 1) / WN_TPB)), dim3(WN_TPB), 0, st, *a);
    return 0;
}

// one thread per utterance (Q <= a few hundred): first-max argmax or inverse-CDF draw
__global__ __launch_bounds__(WN_TPB) void k_dl_select(const float* __restrict__ logits, int Q, int nb, int64_t* samples, long Ttot,
                                                      const int* t_forced, const int* t_end, int p, const float* uniforms,
                                                      float* logits_out, int mode) {
    const int u = blockIdx.x * WN_TPB + threadIdx.x;
    if (u >= nb) return;
    float best = -3.0e38f;
    int bi = 0;
    for (int q = 0; q < Q; ++q) {
        const float v = logits[(long)q * nb + u];
        if (logits_out) logits_out[((long)u * Ttot + p) * Q + q] = v;
        if (v > best) { best = v; bi = q; }
    }
    int chosen = bi;
    if (mode == 1 && uniforms != nullptr) {
        float total = 0.0f;
        for (int q = 0; q < Q; ++q) total += expf(logits[(long)q * nb + u] - best);
        const float target = uniforms[(long)u * Ttot + p + 1] * total;
        float run = 0.0f;
        int cand = -1;
        for (int q = 0; q < Q; ++q) {
            run += expf(logits[(long)q * nb + u] - best);
            if (cand < 0 && run >= target) cand = q;
        }
        if (cand >= 0) chosen = cand;
    }
    if (p + 1 >= t_forced[u] && p + 1 < t_end[u]) samples[(long)u * Ttot + p + 1] = chosen;
}

int wn_dl_select(const float* logits, int Q, int nb, int64_t* samples, long Ttot, const int* t_forced, const int* t_end, int p,
                 const float* uniforms, float* logits_out, int mode, wn_stream_t st) {
    WN_PROF("dl_select", 0.0, 0.0, st);
    WN_LAUNCH(k_dl_select, dim3((unsigned)((nb + WN_TPB - 1) / WN_TPB)), dim3(WN_TPB), 0, st, logits, Q, nb, samples, Ttot,
              t_forced, t_end, p, uniforms, logits_out, mode);
    return 0;
}

// NW waves split the K range of a 32 x 32 output tile; every wave requests the operands of up to four 16-k
// steps before its first MFMA (a launch is a handful of dependent memory round trips, so what counts is how few
// of them are exposed), partial tiles are summed through LDS in a fixed order.
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_dl_mm(WnDlMmArgs a) {
    __shared__ float red[NW][32 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32, z = blockIdx.z;
    const float* __restrict__ Az = a.A + (long)z * a.a_zstride;
    const float* __restrict__ Bz = a.B + (long)z * a.b_zstride;
    // this wave's k range (multiples of 2)
    const int kq = ((a.K + 2 * NW - 1) / (2 * NW)) * 2;
    const int k0 = wave * kq, k1 = (k0 + kq < a.K) ? k0 + kq : a.K;
    // row of the weight matrix behind tile row li: plain tiles are 32 consecutive rows; gate tiles pair the
    // sigmoid and the tanh row of 16 channels
    const int mrow = a.gate_R > 0 ? ((li < 16) ? blockIdx.x * 16 + li : a.gate_R + blockIdx.x * 16 + (li - 16)) : m0 + li;
    const bool m_ok = mrow < a.M, u_ok = (n0 + li) < a.nb;
    const float* pa = Az + (m_ok ? mrow : 0);
    const float* pb = Bz + (u_ok ? n0 + li : 0);
    f32x16 acc = f32x16_zero();
    for (int kg = k0; kg < k1; kg += 64) {
        float av[4][8], bv[4][8];
        WN_UNROLL
        for (int g = 0; g < 4; ++g) {
            WN_UNROLL
            for (int s = 0; s < 8; ++s) {
                const int kk = kg + 16 * g + 2 * s + hi;
                const bool ok = kk < k1;
                const int kc = ok ? kk : k0;   // k0 < k1 here: a valid address for the dead lanes
                av[g][s] = pa[(long)kc * a.lda];
                bv[g][s] = pb[(long)kc * a.ldb];
                if (!ok || !m_ok) av[g][s] = 0.0f;
                if (!ok || !u_ok) bv[g][s] = 0.0f;
            }
        }
        WN_SCHED_BARRIER();
        WN_UNROLL
        for (int g = 0; g < 4; ++g) {
            if (kg + 16 * g < k1) {
                WN_UNROLL
                for (int s = 0; s < 8; ++s) acc = mfma32(av[g][s], bv[g][s], acc);
            }
        }
    }
    WN_UNROLL
    for (int r = 0; r < 16; ++r) red[wave][mfma32_row(r, hi) * 32 + li] = acc[r];
    __syncthreads();
    float* Cz = a.C + (long)z * a.c_zstride;
    if (a.gate_R > 0) {
        const int R = a.gate_R;
        for (int i = tid; i < 16 * 32; i += NW * 64) {
            const int row = i >> 5, col = i & 31;
            const int c = blockIdx.x * 16 + row, u = n0 + col;
            if (c < R && u < a.nb) {
                float vs = 0.0f, vt = 0.0f;
                WN_UNROLL
                for (int w = 0; w < NW; w += 4) {
                    vs += (red[w][i] + red[w + 1][i]) + (red[w + 2][i] + red[w + 3][i]);
                    vt += (red[w][i + 512] + red[w + 1][i + 512]) + (red[w + 2][i + 512] + red[w + 3][i + 512]);
                }
                const float ps = vs + (a.gate_g[(long)c * a.nb + u] + a.gate_c[c]);
                const float pt = vt + (a.gate_g[(long)(R + c) * a.nb + u] + a.gate_c[R + c]);
                Cz[(long)c * a.ldc + u] = wn_sigmoid(ps) * wn_tanh(pt);
            }
        }
        return;
    }
    for (int i = tid; i < 32 * 32; i += NW * 64) {
        const int row = i >> 5, col = i & 31;
        const int m = m0 + row, u = n0 + col;
        if (m < a.M && u < a.nb) {
            float v = 0.0f;
            WN_UNROLL
            for (int w = 0; w < NW; w += 4) v += (red[w][i] + red[w + 1][i]) + (red[w + 2][i] + red[w + 3][i]);
            if (a.bias) v += a.bias[m];
            if (a.D) v += a.D[(long)m * a.ldd + u];
            if (a.relu) v = fmaxf(v, 0.0f);
            Cz[(long)m * a.ldc + u] = v;
        }
    }
}

int wn_dl_mm(const WnDlMmArgs* a, wn_stream_t st) {
    WN_PROF(a->tag ? a->tag : "dl_mm", 2.0 * a->M * (double)a->K * a->nb * a->nz, (double)a->M * a->K * 4.0 * a->nz, st);
    if (a->M <= 0 || a->K <= 0 || a->nb <= 0 || a->nz <= 0) return 1;
    if (a->gate_R > 0 && (a->gate_R % 16 != 0 || a->M != 2 * a->gate_R || !a->gate_g || !a->gate_c)) return 1;
    dim3 grid((unsigned)(a->gate_R > 0 ? a->gate_R / 16 : (a->M + 31) / 32), (unsigned)((a->nb + 31) / 32), (unsigned)a->nz);
    // few tiles and a long K: 16 waves per tile keep every wave's range at one group of requests
    const long tiles = (long)grid.x * grid.y * grid.z;
    if (a->K >= 512 && tiles <= 512) {
        WN_LAUNCH(k_dl_mm<16>, grid, dim3(1024), 0, st, *a);
    } else {
        WN_LAUNCH(k_dl_mm<4>, grid, dim3(256), 0, st, *a);
    }
    return 0;
}

__global__ __launch_bounds__(WN_TPB) void k_dl_sum(const float* __restrict__ part, int nz, long zstride, int M, int nb,
                                                   const float* __restrict__ bias, int relu, float* __restrict__ out) {
    const long i = (long)blockIdx.x * WN_TPB + threadIdx.x;
    if (i >= (long)M * nb) return;
    float s = 0.0f;
    for (int z0 = 0; z0 < nz; z0 += 8) {
        float v[8];
        WN_UNROLL
        for (int u = 0; u < 8; ++u) v[u] = (z0 + u < nz) ? part[(long)(z0 + u) * zstride + i] : 0.0f;
        WN_UNROLL
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    if (bias) s += bias[i / nb];
    if (relu) s = fmaxf(s, 0.0f);
    out[i] = s;
}

int wn_dl_sum(const float* part, int nz, long zstride, int M, int nb, const float* bias, int relu, float* out, wn_stream_t st) {
    WN_PROF("dl_sum", 0.0, 0.0, st);
    const long n = (long)M * nb;
    WN_LAUNCH(k_dl_sum, dim3((unsigned)((n + WN_TPB - 1) / WN_TPB)), dim3(WN_TPB), 0, st, part, nz, zstride, M, nb, bias, relu, out);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// mixture-of-logistics negative log-likelihood, forward + gradient (see wn_elem.h)
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ float mol_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
static __device__ __forceinline__ float mol_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// log-likelihood of one mixture component and its derivatives w.r.t. mean and (clamped) log-scale
static __device__ __forceinline__ float mol_component(float y, float mean, float ls, float half_bin, float log_half_classes,
                                                      float* dll_dm, float* dll_dls) {
    const float inv = expf(-ls), c = y - mean;
    const float plus_in = inv * (c + half_bin), min_in = inv * (c - half_bin), mid_in = inv * c;
    if (y < -0.999f) {  // left edge: everything below the first bin
        const float sg = mol_sigmoid(-plus_in);
        *dll_dm = -inv * sg;
        *dll_dls = -plus_in * sg;
        return plus_in - mol_softplus(plus_in);
    }
    if (y > 0.999f) {  // right edge
        const float sg = mol_sigmoid(min_in);
        *dll_dm = inv * sg;
        *dll_dls = min_in * sg;
        return -mol_softplus(min_in);
    }
    // mass of the bin, sigmoid(plus_in) - sigmoid(min_in), WITHOUT the subtraction: with 65536 classes the two sigmoids are one
    // part in 1e5 apart and their fp32 difference carries per cent of rounding error (6e-8 / 1e-5), which goes straight into
    // the gradient.  With a = mid_in, h = inv * half_bin:  sigmoid(a + h) - sigmoid(a - h) = sinh(h) / (cosh(a) + cosh(h)),
    // written on u = exp(-|a|), q = exp(-h) (both <= 1) so that nothing overflows:
    const float hb = inv * half_bin;
    const float u = expf(-fabsf(mid_in)), q = expf(-hb);
    const float delta = u * (-expm1f(-2.0f * hb)) / (q * (1.0f + u * u) + u * (1.0f + q * q));
    if (delta > 1e-5f) {
        const float cp = mol_sigmoid(plus_in), cm = mol_sigmoid(min_in);
        const float dp = cp * mol_sigmoid(-plus_in), dm = cm * mol_sigmoid(-min_in);
        // (dp - dm) / delta = 1 - cp - cm exactly (dp - dm = (cp - cm)(1 - cp - cm))
        const float w = mol_sigmoid(-plus_in) - cm;
        *dll_dm = -inv * w;
        *dll_dls = -mid_in * w - hb * (dp + dm) / delta;
        return logf(delta);
    }
    // extremely narrow component: density at the bin centre times the bin width
    const float sm = mol_sigmoid(mid_in);
    *dll_dm = -inv * (1.0f - 2.0f * sm);
    *dll_dls = -mid_in * (1.0f - 2.0f * sm) - 1.0f;
    return mid_in - ls - 2.0f * mol_softplus(mid_in) - log_half_classes;
}

__global__ __launch_bounds__(WN_TPB) void k_mol_nll(const float* __restrict__ out, const float* __restrict__ yv,
                                                    float* __restrict__ dout, float* __restrict__ loss_partial, int T, int nm,
                                                    int t_start, float grad_scale, float half_bin, float log_half_classes,
                                                    float log_scale_min) {
    __shared__ float red[4];
    const int t = blockIdx.x * WN_TPB + threadIdx.x;
    const int b = blockIdx.y;
    float my = 0.0f;
    const bool live = (t < T) && (t >= t_start);
    const float* o = out + (long)b * 3 * nm * T + t;
    float* d = dout ? dout + (long)b * 3 * nm * T + t : nullptr;
    if (live) {
        const float y = yv[(long)b * T + t];
        // log-softmax of the mixture logits
        float m1 = -3.0e38f;
        for (int i = 0; i < nm; ++i) m1 = fmaxf(m1, o[(long)i * T]);
        float s1 = 0.0f;
        for (int i = 0; i < nm; ++i) s1 += expf(o[(long)i * T] - m1);
        const float lse_p = m1 + logf(s1);
        // log-sum-exp over components of (component log-likelihood + log prior)
        float m2 = -3.0e38f;
        for (int i = 0; i < nm; ++i) {
            float a, c;
            const float ls = fmaxf(o[(long)(2 * nm + i) * T], log_scale_min);
            const float lp = mol_component(y, o[(long)(nm + i) * T], ls, half_bin, log_half_classes, &a, &c) + o[(long)i * T] - lse_p;
            m2 = fmaxf(m2, lp);
        }
        float s2 = 0.0f;
        for (int i = 0; i < nm; ++i) {
            float a, c;
            const float ls = fmaxf(o[(long)(2 * nm + i) * T], log_scale_min);
            const float lp = mol_component(y, o[(long)(nm + i) * T], ls, half_bin, log_half_classes, &a, &c) + o[(long)i * T] - lse_p;
            s2 += expf(lp - m2);
        }
        const float lse = m2 + logf(s2);
        my = -lse;
        if (d) {
            for (int i = 0; i < nm; ++i) {
                float dm, dls;
                const float raw = o[(long)(2 * nm + i) * T];
                const float ls = fmaxf(raw, log_scale_min);
                const float logit = o[(long)i * T];
                const float lp = mol_component(y, o[(long)(nm + i) * T], ls, half_bin, log_half_classes, &dm, &dls) + logit - lse_p;
                const float r = expf(lp - lse);           // posterior responsibility
                const float pi = expf(logit - lse_p);     // prior
                d[(long)i * T] = (pi - r) * grad_scale;
                d[(long)(nm + i) * T] = -r * dm * grad_scale;
                d[(long)(2 * nm + i) * T] = raw >= log_scale_min ? -r * dls * grad_scale : 0.0f;
            }
        }
    } else if (t < T && d) {
        for (int i = 0; i < 3 * nm; ++i) d[(long)i * T] = 0.0f;
    }
    const float tot = block_reduce_sum(my, red);
    if (threadIdx.x == 0) loss_partial[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

int wn_mol_nll(const float* out, const float* y, float* dout, float* loss_partial, int* n_partial, int B, int T, int nm,
               int t_start, float grad_scale, int num_classes, float log_scale_min, wn_stream_t st) {
    WN_PROF("mol_nll", 0.0, 0.0, st);
    if (nm < 1 || num_classes < 2) return 1;
    dim3 grid((T + WN_TPB - 1) / WN_TPB, B);
    if (n_partial) *n_partial = (int)(grid.x * grid.y);
    WN_LAUNCH(k_mol_nll, grid, dim3(WN_TPB), 0, st, out, y, dout, loss_partial, T, nm, t_start, grad_scale,
              1.0f / (float)(num_classes - 1), logf((float)(num_classes - 1) * 0.5f), log_scale_min);
    return 0;
}

__global__ __launch_bounds__(WN_TPB) void k_dl_select_mol(const float* __restrict__ out, int nm, int nb, int Q, int64_t* samples,
                                                          float* wave_out, long Ttot, const int* t_forced, const int* t_end,
                                                          int p, const float* __restrict__ uniforms, float* out_copy,
                                                          float log_scale_min) {
    const int u = blockIdx.x * WN_TPB + threadIdx.x;
    if (u >= nb) return;
    if (out_copy) {
        for (int q = 0; q < 3 * nm; ++q) out_copy[((long)u * Ttot + p) * (3 * nm) + q] = out[(long)q * nb + u];
    }
    if (!(p + 1 >= t_forced[u] && p + 1 < t_end[u])) return;
    const float* un = uniforms + ((long)u * Ttot + p + 1) * (nm + 1);
    float best = -3.0e38f;
    int k = 0;
    for (int i = 0; i < nm; ++i) {
        const float g = out[(long)i * nb + u] - logf(-logf(un[i]));
        if (g > best) { best = g; k = i; }
    }
    const float mean = out[(long)(nm + k) * nb + u];
    const float ls = fmaxf(out[(long)(2 * nm + k) * nb + u], log_scale_min);
    const float uu = un[nm];
    float x = mean + expf(ls) * (logf(uu) - logf(1.0f - uu));
    x = fminf(fmaxf(x, -1.0f), 1.0f);
    if (wave_out) wave_out[(long)u * Ttot + p + 1] = x;
    // mu-law encode (reference wavenet.py:17-30): sign(x) ln(1+mu|x|)/ln(1+mu) -> floor((fx+1)/2*mu + 0.5)
    const float mu = (float)(Q - 1);
    const float fx = copysignf(logf(1.0f + mu * fabsf(x)) / logf(1.0f + mu), x);
    samples[(long)u * Ttot + p + 1] = (int64_t)floorf((fx + 1.0f) * 0.5f * mu + 0.5f);
}

int wn_dl_select_mol(const float* out, int nm, int nb, int Q, int64_t* samples, float* wave_out, long Ttot, const int* t_forced,
                     const int* t_end, int p, const float* uniforms, float* out_copy, float log_scale_min, wn_stream_t st) {
    WN_PROF("dl_select_mol", 0.0, 0.0, st);
    WN_LAUNCH(k_dl_select_mol, dim3((unsigned)((nb + WN_TPB - 1) / WN_TPB)), dim3(WN_TPB), 0, st, out, nm, nb, Q, samples, wave_out,
              Ttot, t_forced, t_end, p, uniforms, out_copy, log_scale_min);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// parallel context walk of the decode path (see wn_elem.h)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WN_TPB) void k_decode_ctx_aux(const float* __restrict__ h, const float* __restrict__ upw,
                                                           const float* __restrict__ upb, float* __restrict__ out, int A, int F,
                                                           int U, int T, int n_pad, int pos0) {
    const int p = blockIdx.x * WN_TPB + threadIdx.x;
    const int a = blockIdx.y, b = blockIdx.z;
    if (p >= T) return;
    const int t = pos0 + p > n_pad ? pos0 + p - n_pad : 0;
    const float* hr = h + ((long)b * A + a) * F;
    float v;
    if (U > 0) {
        int f = t / U;
        const float w = upw[t - f * U];
        if (f > F - 1) f = F - 1;
        v = w * hr[f] + (upb ? upb[0] : 0.0f);
    } else {
        v = hr[t < F ? t : F - 1];
    }
    out[((long)b * A + a) * T + p] = v;
}

int wn_decode_ctx_aux_rows(const float* h, const float* upw, const float* upb, float* out, int B, int A, int F, int U, int T,
                           int n_pad, int pos0, wn_stream_t st) {
    WN_PROF("decode_ctx_aux", 0.0, 0.0, st);
    if (B < 1 || A < 1 || F < 1 || T < 1 || A > 65535 || B > 65535 || pos0 < 0) return 1;
    WN_LAUNCH(k_decode_ctx_aux, dim3((unsigned)((T + WN_TPB - 1) / WN_TPB), (unsigned)A, (unsigned)B), dim3(WN_TPB), 0, st, h, upw,
              upb, out, A, F, U, T, n_pad, pos0);
    return 0;
}

__global__ __launch_bounds__(WN_TPB) void k_decode_fill_queues(const float* __restrict__ X, float* __restrict__ dst, int B, int R,
                                                               int T, int K, int depth, int P0, int pos0, long elem_stride,
                                                               long utt_stride) {
    const int l = blockIdx.x, b = blockIdx.y;
    const int Dq = (K - 1) << (l % depth);
    const long qoff = dl_queue_off(l, depth, K, R);
    const float* Xl = X + ((long)l * B + b) * R * T;
    for (int i = threadIdx.x; i < Dq * R; i += WN_TPB) {  // consecutive threads read consecutive positions of one channel
        const int c = i / Dq, q = P0 - Dq + (i - c * Dq);
        if (q < 0) continue;  // zero history (the state buffer is zero initialised)
        dst[(qoff + (long)((pos0 + q) % Dq) * R + c) * elem_stride + (long)b * utt_stride] = Xl[(long)c * T + q];
    }
}

int wn_decode_fill_queues(const float* X, float* dst, int L, int B, int R, int T, int K, int depth, int P0, int pos0,
                          long elem_stride, long utt_stride, wn_stream_t st) {
    WN_PROF("decode_fill_queues", 0.0, 0.0, st);
    if (K < 2) return 0;  // no history taps, no queues
    if (L < 1 || B < 1 || B > 65535 || P0 < 0 || P0 > T || pos0 < 0) return 1;
    WN_LAUNCH(k_decode_fill_queues, dim3((unsigned)L, (unsigned)B), dim3(WN_TPB), 0, st, X, dst, B, R, T, K, depth, P0, pos0,
              elem_stride, utt_stride);
    return 0;
}
