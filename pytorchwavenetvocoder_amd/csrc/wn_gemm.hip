// wn_gemm.hip -- generic fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Why f32-input MFMA: the parity gate is 1e-4 on fp32 logits after 30 residual layers; the
// f32 MFMA is bit-for-bit a k-ordered fmaf chain (exact f32) and runs at the f32 vector peak
// (157 TFLOP/s) while leaving the VALU free for loaders/epilogues.  This kernel is the
// work-horse for everything on the path that is a plain contraction:
//   * skip-sum   (reference wavenet.py:533,238):  [S x L*R] . [L*R x T]      (segment = layer)
//   * post-net   (wavenet.py:518-523):            [S x S].[S x T], [Q x S].[S x T]
//   * weight gradients (contraction over time)    dW[o][i] = sum_t dY[o][t] X[i][t-shift]
//   * the any-size "layered" path (R != 64): dilated taps as K-segments with per-segment shift.
//
// Tiling: 256 threads = 4 waves (2x2), block tile (64*TM) x (64*TN), BK = 32; each wave owns
// TM x TN 32x32 MFMA tiles.  Global -> registers -> LDS with the next tile's loads in flight
// during the MFMAs of the current one.  All LDS traffic is ds_read/write_b32 and conflict free:
//   k-minor tiles  [kk][mn]      : lanes read 32 consecutive floats of one row
//   k-major tiles  [mn][kk + 1]  : row stride 33 dwords -> bank = (mn + kk) mod 32
// Lane l of a wave feeds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; accumulator register r
// of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]  (see wn_device.h).
#include "wn_gemm.h"
#include "wn_prof.h"

#define WN_BK 32
#define WN_GEMM_THREADS 256

template <int BMN, int KMAJ>
struct TileGeom {
    static constexpr int LD = KMAJ ? (WN_BK + 1) : BMN;
    static constexpr int SIZE = KMAJ ? BMN * (WN_BK + 1) : WN_BK * BMN;
    static constexpr int NE = BMN * WN_BK / WN_GEMM_THREADS;
    static __device__ __forceinline__ void coord(int e, int tid, int& kk, int& mn) {
        int idx = e * WN_GEMM_THREADS + tid;
        if (KMAJ) {
            mn = idx / WN_BK;
            kk = idx % WN_BK;
        } else {
            kk = idx / BMN;
            mn = idx % BMN;
        }
    }
    static __device__ __forceinline__ int soff(int kk, int mn) { return KMAJ ? mn * LD + kk : kk * LD + mn; }
};

template <int TM, int TN, int KMAJ, int ONEHOT>
__global__ __launch_bounds__(WN_GEMM_THREADS, 3) void wn_gemm_kernel(WnGemmArgs g) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    typedef TileGeom<BM, KMAJ> GA;
    typedef TileGeom<BN, KMAJ> GB;
    __shared__ __attribute__((aligned(16))) float As[GA::SIZE];
    __shared__ __attribute__((aligned(16))) float Bs[GB::SIZE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z;
    const int zl = z / (g.nbatch * g.ksplit);            // layer (outermost)
    const int zr = z - zl * (g.nbatch * g.ksplit);
    const int b = zr / g.ksplit;
    const int ks = zr - b * g.ksplit;
    const int dmul = (g.b_dil_depth > 0) ? (1 << ((g.b_layer0 + zl) % g.b_dil_depth)) : 1;
    const int sh0 = g.b_shift0 * dmul, shstep = g.b_shift_step * dmul;
    const int kbeg = ks * g.kchunk;
    const int kend = (g.K - kbeg > g.kchunk) ? (kbeg + g.kchunk) : g.K;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const float* __restrict__ Az = g.A + (long)zl * g.a_lstride + (long)b * g.a_zstride;
    const float* __restrict__ Bz = g.B + (long)zl * g.b_lstride + (long)b * g.b_zstride;
    const bool one_seg = (g.b_seg_len >= (KMAJ ? g.N : g.K));

    float ra[GA::NE], rb[GB::NE];
    f32x16 acc[TM][TN];
    WN_UNROLL
    for (int i = 0; i < TM; ++i) {
        WN_UNROLL
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x16_zero();
    }
    float rowsum = 0.0f;

    // ---- operand loaders -----------------------------------------------------------------
    // All per-row work (segment lookup = integer division, shifts, 64-bit row offsets) is hoisted
    // out of the per-element path: the loaders are VALU work that competes with the MFMA issue of
    // the co-resident waves.
    //   k-minor tiles: a thread owns ONE column (m or n) and walks the tile's k rows; when segments
    //     are BK-aligned a whole tile lies in one segment (block-uniform scalar math);
    //   k-major tiles: a thread owns ONE k column and walks rows that never change across k-tiles;
    //     their (offset, shift, index) are computed once per block into an LDS table.
    __shared__ long b_rowoff[KMAJ ? BN : 1];
    __shared__ int b_rowshift[KMAJ ? BN : 1];
    __shared__ int b_rowrr[KMAJ ? BN : 1];
    __shared__ int b_kvec_flag;
    if (KMAJ) {
        if (tid == 0) b_kvec_flag = 1;
        __syncthreads();
        for (int i = tid; i < BN; i += WN_GEMM_THREADS) {
            const int n = n0 + i;
            int seg = 0, rr = n;
            if (!one_seg) {
                seg = n / g.b_seg_len;
                rr = n - seg * g.b_seg_len;
            }
            const long off = (n < g.N) ? ((long)seg * g.b_seg_stride + (long)rr * g.ldb) : -1;
            const int sh = sh0 + seg * shstep;
            b_rowoff[i] = off;
            b_rowshift[i] = sh;
            b_rowrr[i] = rr;
            if ((sh & 3) != 0 || (off >= 0 && (off & 3) != 0)) b_kvec_flag = 0;  // benign race: only 0 is ever written
        }
        __syncthreads();
    }
    // 16-byte loads along k (time) for the k-major (dW type) tiles: block-uniform eligibility
    const bool k_base_ok = KMAJ && !ONEHOT && (kbeg % 4 == 0) && (kend % 4 == 0);
    const bool a_kvec = k_base_ok && (g.lda % 4 == 0) && (m0 + BM <= g.M) && ((reinterpret_cast<uintptr_t>(Az) & 15) == 0);
    const bool b_kvec_static = k_base_ok && (b_kvec_flag != 0) && (n0 + BN <= g.N) &&
                               ((reinterpret_cast<uintptr_t>(Bz) & 15) == 0);
    // shift range over the rows of this B tile (shift is affine in the segment index)
    int b_shmin = sh0, b_shmax = sh0;
    if (KMAJ && !one_seg) {
        const int s_lo = sh0 + (n0 / g.b_seg_len) * shstep, s_hi = sh0 + ((n0 + BN - 1) / g.b_seg_len) * shstep;
        b_shmin = s_lo < s_hi ? s_lo : s_hi;
        b_shmax = s_lo < s_hi ? s_hi : s_lo;
    }
    const bool seg_aligned = one_seg || (g.b_seg_len % WN_BK == 0 && kbeg % WN_BK == 0);
    constexpr int A_RPP = KMAJ ? (WN_GEMM_THREADS / WN_BK) : (WN_GEMM_THREADS / BM);  // rows per pass
    constexpr int B_RPP = KMAJ ? (WN_GEMM_THREADS / WN_BK) : (WN_GEMM_THREADS / BN);
    const int a_col = KMAJ ? (tid % WN_BK) : (tid % BM);  // fixed column of this thread
    const int a_row0 = KMAJ ? (tid / WN_BK) : (tid / BM);
    const int b_col = KMAJ ? (tid % WN_BK) : (tid % BN);
    const int b_row0 = KMAJ ? (tid / WN_BK) : (tid / BN);

    // 16-byte staging path for k-minor tiles (forward / dX type): a thread owns 4 consecutive columns,
    // loads them with one global_load_dwordx4 and stores them with one ds_write_b128 -> 4x fewer VMEM,
    // LDS-write and address instructions than the scalar path.  Used when the tile is interior and
    // 16-byte aligned (block-uniform decisions), otherwise the scalar path below handles edges/shifts.
    constexpr int A_TPR = BM / 4, B_TPR = BN / 4;                          // threads per tile row
    constexpr int A_VRPP = WN_GEMM_THREADS / A_TPR, B_VRPP = WN_GEMM_THREADS / B_TPR;  // rows per pass
    const bool a_vec = !KMAJ && (g.lda % 4 == 0) && (m0 + BM <= g.M) &&
                       ((reinterpret_cast<uintptr_t>(Az) & 15) == 0);
    const bool b_vec_static = !KMAJ && !ONEHOT && seg_aligned && (g.ldb % 4 == 0) && (g.b_seg_stride % 4 == 0) &&
                              (n0 + BN <= g.N) && ((reinterpret_cast<uintptr_t>(Bz) & 15) == 0);
    bool rb_vec = false;  // layout of the B registers currently held (set by fetch, used by the LDS store)

    auto fetch = [&](int k0) {
        if (!KMAJ && a_vec) {
            const int c4 = (tid % A_TPR) * 4, r0 = tid / A_TPR;
            WN_UNROLL
            for (int p4 = 0; p4 < GA::NE / 4; ++p4) {
                const int k = k0 + r0 + p4 * A_VRPP;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < kend) v = *reinterpret_cast<const float4*>(Az + (long)k * g.lda + m0 + c4);
                ra[4 * p4 + 0] = v.x; ra[4 * p4 + 1] = v.y; ra[4 * p4 + 2] = v.z; ra[4 * p4 + 3] = v.w;
            }
        }
        if (!KMAJ) {
            rb_vec = false;
            if (b_vec_static) {
                int seg = 0, rr0 = k0;
                if (!one_seg) {
                    seg = k0 / g.b_seg_len;
                    rr0 = k0 - seg * g.b_seg_len;
                }
                const int cc0 = n0 - (sh0 + seg * shstep);
                if ((cc0 % 4 == 0) && cc0 >= 0 && cc0 + BN <= g.b_clen) {
                    rb_vec = true;
                    const int c4 = (tid % B_TPR) * 4, r0 = tid / B_TPR;
                    const float* base = Bz + (long)seg * g.b_seg_stride + cc0 + c4;
                    WN_UNROLL
                    for (int p4 = 0; p4 < GB::NE / 4; ++p4) {
                        const int kr = r0 + p4 * B_VRPP;
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (k0 + kr < kend) v = *reinterpret_cast<const float4*>(base + (long)(rr0 + kr) * g.ldb);
                        if (g.b_relu) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                        rb[4 * p4 + 0] = v.x; rb[4 * p4 + 1] = v.y; rb[4 * p4 + 2] = v.z; rb[4 * p4 + 3] = v.w;
                    }
                }
            }
        }
        if (KMAJ && a_kvec) {
            // A[m][k..k+3]: 8 threads per row, 32 rows per pass
            const int kq = (tid & 7) * 4, r0 = tid >> 3;
            WN_UNROLL
            for (int p4 = 0; p4 < GA::NE / 4; ++p4) {
                const int m = m0 + r0 + 32 * p4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + kq < kend) v = *reinterpret_cast<const float4*>(Az + (long)m * g.lda + k0 + kq);  // (kend % 4 == 0)
                ra[4 * p4 + 0] = v.x; ra[4 * p4 + 1] = v.y; ra[4 * p4 + 2] = v.z; ra[4 * p4 + 3] = v.w;
            }
        }
        if (KMAJ) {
            // interior tile (every row's shifted window lies inside [0, clen) and the k-tile is full):
            // unconditional 16-byte loads; any other tile takes the scalar path
            rb_vec = b_kvec_static && (k0 - b_shmax >= 0) && (k0 + WN_BK - b_shmin <= g.b_clen) && (k0 + WN_BK <= kend);
        }
        if (KMAJ && rb_vec) {
            const int kq = (tid & 7) * 4, r0 = tid >> 3;
            WN_UNROLL
            for (int p4 = 0; p4 < GB::NE / 4; ++p4) {
                const int i = r0 + 32 * p4;
                float4 v = *reinterpret_cast<const float4*>(Bz + b_rowoff[i] + (k0 + kq - b_rowshift[i]));
                if (g.b_relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                rb[4 * p4 + 0] = v.x; rb[4 * p4 + 1] = v.y; rb[4 * p4 + 2] = v.z; rb[4 * p4 + 3] = v.w;
            }
        }
        if (KMAJ) {
          if (!a_kvec) {
            // A[m][k]: column = k (fixed), rows = m
            const int k = k0 + a_col;
            const bool kok = k < kend;
            const float* pa = Az + (long)(m0 + a_row0) * g.lda + k;
            const long astep = (long)A_RPP * g.lda;
            WN_UNROLL
            for (int e = 0; e < GA::NE; ++e) {
                const bool ok = kok && (m0 + a_row0 + e * A_RPP) < g.M;
                ra[e] = ok ? *pa : 0.0f;
                pa += astep;
            }
          }
          if (!rb_vec) {
            // B[n][k]: column = k (fixed), rows = n (table)
            const int kb = k0 + b_col;
            const bool kbok = kb < kend;
            WN_UNROLL
            for (int e = 0; e < GB::NE; ++e) {
                const int i = b_row0 + e * B_RPP;
                const long off = b_rowoff[i];
                const int cc = kb - b_rowshift[i];
                const bool ok = kbok && off >= 0 && cc >= 0 && cc < g.b_clen;
                float v = 0.0f;
                if (ONEHOT) {
                    if (ok) {
                        long long q = g.b_index[(long)b * g.b_index_zstride + cc] % g.b_index_mod;
                        if (q < 0) q += g.b_index_mod;
                        v = ((int)q == b_rowrr[i]) ? 1.0f : 0.0f;
                    }
                } else {
                    v = ok ? Bz[off + cc] : 0.0f;
                    if (g.b_relu) v = fmaxf(v, 0.0f);
                }
                rb[e] = v;
            }
          }
        } else {
          if (!a_vec) {
            // A[k][m]: column = m (fixed), rows = k
            const bool mok = (m0 + a_col) < g.M;
            const float* pa = Az + (long)(k0 + a_row0) * g.lda + (m0 + a_col);
            const long astep = (long)A_RPP * g.lda;
            WN_UNROLL
            for (int e = 0; e < GA::NE; ++e) {
                const bool ok = mok && (k0 + a_row0 + e * A_RPP) < kend;
                ra[e] = ok ? *pa : 0.0f;
                pa += astep;
            }
          }
          if (!rb_vec) {
            // B[k][n]: column = n (fixed), rows = k
            const int n = n0 + b_col;
            const bool nok = n < g.N;
            if (seg_aligned) {
                int seg = 0, rr0 = k0;
                if (!one_seg) {
                    seg = k0 / g.b_seg_len;
                    rr0 = k0 - seg * g.b_seg_len;
                }
                const int cc = n - (sh0 + seg * shstep);
                const bool cok = nok && cc >= 0 && cc < g.b_clen;
                const float* pb = Bz + (long)seg * g.b_seg_stride + (long)(rr0 + b_row0) * g.ldb + cc;
                const long bstep = (long)B_RPP * g.ldb;
                WN_UNROLL
                for (int e = 0; e < GB::NE; ++e) {
                    const bool ok = cok && (k0 + b_row0 + e * B_RPP) < kend;
                    float v = ok ? *pb : 0.0f;
                    if (g.b_relu) v = fmaxf(v, 0.0f);
                    rb[e] = v;
                    pb += bstep;
                }
            } else {
                WN_UNROLL
                for (int e = 0; e < GB::NE; ++e) {
                    const int k = k0 + b_row0 + e * B_RPP;
                    const int seg = k / g.b_seg_len;
                    const int rr = k - seg * g.b_seg_len;
                    const int cc = n - (sh0 + seg * shstep);
                    const bool ok = nok && k < kend && cc >= 0 && cc < g.b_clen;
                    float v = ok ? Bz[(long)seg * g.b_seg_stride + (long)rr * g.ldb + cc] : 0.0f;
                    if (g.b_relu) v = fmaxf(v, 0.0f);
                    rb[e] = v;
                }
            }
          }
        }
    };

    const int nk = (kend > kbeg) ? (kend - kbeg + WN_BK - 1) / WN_BK : 0;
    if (nk > 0) fetch(kbeg);
    for (int kt = 0; kt < nk; ++kt) {
        if (KMAJ && a_kvec) {
            const int kq = (tid & 7) * 4, r0 = tid >> 3;
            WN_UNROLL
            for (int p4 = 0; p4 < GA::NE / 4; ++p4) {
                WN_UNROLL
                for (int i = 0; i < 4; ++i) As[GA::soff(kq + i, r0 + 32 * p4)] = ra[4 * p4 + i];
            }
        } else if (!KMAJ && a_vec) {
            const int c4 = (tid % A_TPR) * 4, r0 = tid / A_TPR;
            WN_UNROLL
            for (int p4 = 0; p4 < GA::NE / 4; ++p4)
                *reinterpret_cast<float4*>(&As[(r0 + p4 * A_VRPP) * BM + c4]) =
                    make_float4(ra[4 * p4 + 0], ra[4 * p4 + 1], ra[4 * p4 + 2], ra[4 * p4 + 3]);
        } else {
            WN_UNROLL
            for (int e = 0; e < GA::NE; ++e) {
                int kk, mn;
                GA::coord(e, tid, kk, mn);
                As[GA::soff(kk, mn)] = ra[e];
            }
        }
        if (KMAJ && rb_vec) {
            const int kq = (tid & 7) * 4, r0 = tid >> 3;
            WN_UNROLL
            for (int p4 = 0; p4 < GB::NE / 4; ++p4) {
                WN_UNROLL
                for (int i = 0; i < 4; ++i) Bs[GB::soff(kq + i, r0 + 32 * p4)] = rb[4 * p4 + i];
            }
        } else if (!KMAJ && rb_vec) {
            const int c4 = (tid % B_TPR) * 4, r0 = tid / B_TPR;
            WN_UNROLL
            for (int p4 = 0; p4 < GB::NE / 4; ++p4)
                *reinterpret_cast<float4*>(&Bs[(r0 + p4 * B_VRPP) * BN + c4]) =
                    make_float4(rb[4 * p4 + 0], rb[4 * p4 + 1], rb[4 * p4 + 2], rb[4 * p4 + 3]);
        } else {
            WN_UNROLL
            for (int e = 0; e < GB::NE; ++e) {
                int kk, mn;
                GB::coord(e, tid, kk, mn);
                Bs[GB::soff(kk, mn)] = rb[e];
            }
        }
        __syncthreads();
        if (kt + 1 < nk) fetch(kbeg + (kt + 1) * WN_BK);

        {
            // LDS operands of k-step s+1 are read before the MFMAs of k-step s (measured on MI355X with
            // tools/mfma_probe.hip: 98% of the f32 MFMA peak vs 84-92% when each step waits for its own reads)
            float an[TM], bn[TN];
            WN_UNROLL
            for (int i = 0; i < TM; ++i) an[i] = As[GA::soff(hi, (wm * TM + i) * 32 + li)];
            WN_UNROLL
            for (int j = 0; j < TN; ++j) bn[j] = Bs[GB::soff(hi, (wn * TN + j) * 32 + li)];
            WN_UNROLL
            for (int s = 0; s < WN_BK / 2; ++s) {
                float a[TM], bb[TN];
                WN_UNROLL
                for (int i = 0; i < TM; ++i) a[i] = an[i];
                WN_UNROLL
                for (int j = 0; j < TN; ++j) bb[j] = bn[j];
                if (s + 1 < WN_BK / 2) {
                    const int kk = 2 * (s + 1) + hi;
                    WN_UNROLL
                    for (int i = 0; i < TM; ++i) an[i] = As[GA::soff(kk, (wm * TM + i) * 32 + li)];
                    WN_UNROLL
                    for (int j = 0; j < TN; ++j) bn[j] = Bs[GB::soff(kk, (wn * TN + j) * 32 + li)];
                }
                WN_UNROLL
                for (int i = 0; i < TM; ++i) {
                    WN_UNROLL
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i], bb[j], acc[i][j]);
                }
            }
        }
        if (KMAJ && g.a_rowsum != nullptr && tid < BM) {
            float s = 0.0f;
            for (int kk = 0; kk < WN_BK; ++kk) s += As[GA::soff(kk, tid)];
            rowsum += s;
        }
        __syncthreads();
    }

    if (KMAJ && g.a_rowsum != nullptr && blockIdx.x == 0 && tid < BM && (m0 + tid) < g.M)
        g.a_rowsum[(long)z * g.M + m0 + tid] = rowsum;

    // epilogue: lane (col = li, hi) holds rows mfma32_row(r, hi) of each 32x32 tile.  All side inputs
    // of a tile (bias, residual D, mask E, old C) are loaded first with clamped (always valid)
    // indices, so the loads are issued back to back instead of one wait per element.
    float* __restrict__ Cz = g.C + (long)z * g.c_zstride;
    const float* __restrict__ Dz = g.D ? g.D + (long)b * g.d_zstride : nullptr;
    const float* __restrict__ Ez = g.E ? g.E + (long)b * g.e_zstride : nullptr;
    WN_UNROLL
    for (int i = 0; i < TM; ++i) {
        WN_UNROLL
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + li;
            const int nc = n < g.N ? n : g.N - 1;
            WN_UNROLL
            for (int h8 = 0; h8 < 4; ++h8) {  // 4 rows at a time keeps the register peak of the epilogue low
                float bv[4], dv[4], ev[4], cv[4];
                WN_UNROLL
                for (int rr = 0; rr < 4; ++rr) {
                    const int m = m0 + (wm * TM + i) * 32 + mfma32_row(4 * h8 + rr, hi);
                    const int mc = m < g.M ? m : g.M - 1;
                    bv[rr] = g.bias ? g.bias[mc] : 0.0f;
                    // 32-bit element offsets from block-uniform bases (a per-batch slab is < 2^32 elements)
                    dv[rr] = Dz ? Dz[(unsigned)mc * (unsigned)g.ldd + (unsigned)nc] : 0.0f;
                    ev[rr] = Ez ? Ez[(unsigned)mc * (unsigned)g.lde + (unsigned)nc] : 1.0f;
                    cv[rr] = g.accumulate ? Cz[(unsigned)mc * (unsigned)g.ldc + (unsigned)nc] : 0.0f;
                }
                WN_UNROLL
                for (int rr = 0; rr < 4; ++rr) {
                    const int m = m0 + (wm * TM + i) * 32 + mfma32_row(4 * h8 + rr, hi);
                    float v = acc[i][j][4 * h8 + rr] + bv[rr] + dv[rr];
                    if (g.relu) v = fmaxf(v, 0.0f);
                    v = (ev[rr] > 0.0f) ? v : 0.0f;
                    v += cv[rr];
                    if (m < g.M && n < g.N) Cz[(unsigned)m * (unsigned)g.ldc + (unsigned)n] = v;
                }
                WN_SCHED_BARRIER();  // keep the next chunk's loads from being hoisted above these stores
            }
        }
    }
}

template <int TM, int TN, int KMAJ, int ONEHOT>
static void launch_variant(const WnGemmArgs& g, wn_stream_t stream) {
    dim3 grid((unsigned)((g.N + 64 * TN - 1) / (64 * TN)), (unsigned)((g.M + 64 * TM - 1) / (64 * TM)),
              (unsigned)(g.nlayer * g.nbatch * g.ksplit));
    dim3 block(WN_GEMM_THREADS);
    WN_LAUNCH((wn_gemm_kernel<TM, TN, KMAJ, ONEHOT>), grid, block, 0, stream, g);
}

int wn_gemm_launch(const WnGemmArgs* gp, wn_stream_t stream) {
    const WnGemmArgs& g = *gp;
    if (g.M <= 0 || g.N <= 0 || g.K < 0 || g.nbatch <= 0 || g.ksplit <= 0 || g.nlayer <= 0) return 1;
    if (g.a_kmajor != g.b_kmajor) return 2;
    if (g.b_seg_len <= 0 || g.kchunk <= 0) return 3;
    const int tm = g.M > 64 ? 2 : 1, tn = g.N > 64 ? 2 : 1;
    // compulsory bytes: both operands and the result once (a one-hot operand is 8-byte indices)
    WN_PROF(g.tag ? g.tag : "gemm", 2.0 * g.M * g.N * (double)g.K * g.nbatch * g.nlayer,
            ((double)g.M * g.K * 4.0 + (double)g.K * (g.b_index ? 8.0 : 4.0 * g.N) + (double)g.M * g.N * 4.0) * g.nbatch * g.nlayer,
            stream);
    if (g.b_index != nullptr) {
        if (!g.a_kmajor) return 4;  // the one-hot operand exists for the dW (k = time) mode only
        if (tm == 2 && tn == 2) launch_variant<2, 2, 1, 1>(g, stream);
        else if (tm == 2) launch_variant<2, 1, 1, 1>(g, stream);
        else if (tn == 2) launch_variant<1, 2, 1, 1>(g, stream);
        else launch_variant<1, 1, 1, 1>(g, stream);
        return 0;
    }
    const int key = (g.a_kmajor ? 4 : 0) | (tm == 2 ? 2 : 0) | (tn == 2 ? 1 : 0);
    switch (key) {
        case 0: launch_variant<1, 1, 0, 0>(g, stream); break;
        case 1: launch_variant<1, 2, 0, 0>(g, stream); break;
        case 2: launch_variant<2, 1, 0, 0>(g, stream); break;
        case 3: launch_variant<2, 2, 0, 0>(g, stream); break;
        case 4: launch_variant<1, 1, 1, 0>(g, stream); break;
        case 5: launch_variant<1, 2, 1, 0>(g, stream); break;
        case 6: launch_variant<2, 1, 1, 0>(g, stream); break;
        default: launch_variant<2, 2, 1, 0>(g, stream); break;
    }
    return 0;
}
