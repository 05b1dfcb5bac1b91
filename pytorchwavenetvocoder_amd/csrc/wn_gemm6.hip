// wn_gemm6.hip -- forward / dX type GEMM (weights x activations) on v_mfma_f32_32x32x16_bf16 with
// fp32-equivalent results.
//
// The f32-input MFMA (wn_gemm.hip) runs at 1/16 of the bf16 rate, and the skip-sum / post-net
// contractions of the WaveNet path (reference wavenet.py:533,238,518-523) are bound by it.  Here
// every fp32 value is split exactly into three bf16 pieces x = h + m + l (8 + 8 + 8 significand
// bits, each piece the round-to-nearest of the remainder) and
//      a*b ~= h_a h_b + (h_a m_b + m_a h_b) + (h_a l_b + m_a m_b + l_a h_b)
// -- the six products above 2^-24 |ab| -- is accumulated in the fp32 accumulator of the bf16 MFMA
// (products of bf16 pairs are exact in fp32).  The dropped terms are below fp32 round-off, so the
// result differs from an fp32 fma chain by summation-order noise only (tests: same 1e-4 gates).
// Six bf16 MFMAs (6 x 32 cycles per 16-k step) replace eight f32 MFMAs (8 x 64): 2.7x less matrix
// time, which moves these kernels to the HBM roof.
//
// Tile: 256 (all output channels of the skip/post nets) x 128 time steps per block, so every
// activation is read from HBM exactly once; 256 threads = 4 waves (2 x 2), wave tile 128 x 64
// = 8 accumulator tiles.  The weights arrive pre-split (wn_gemm6_pack, once per step); the
// activation slab of a 16-k step is loaded as 8 dwords per lane (lane = time step: coalesced),
// split in registers and written as three 16-byte LDS rows [n][16 k], exactly the B-fragment
// layout, so fragment reads are conflict-free ds_read_b128.  Double-buffered LDS (72 KB), the
// loads of step s+1 are in flight during the 48 MFMAs of step s.
#include "wn_gemm6.h"

#include "wn_prof.h"

#define G6_T 256

__global__ void k_gemm6_pack(const float* src, long lda, int M, int K, int Mpad, unsigned short* Apk) {
    // one thread per (kb, m): 16 k values -> 3 x 16 bf16
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int nkb = (K + 15) / 16;
    if (idx >= (long)nkb * Mpad) return;
    const int kb = (int)(idx / Mpad), m = (int)(idx % Mpad);
    for (int e = 0; e < 16; e += 2) {
        float x[2];
        for (int u = 0; u < 2; ++u) {
            const int k = kb * 16 + e + u;
            x[u] = (m < M && k < K) ? src[(long)k * lda + m] : 0.f;
        }
        const unsigned h = wn_pk_bf16(x[0], x[1]);
        const float r0 = x[0] - wn_bits_f32(h << 16), r1 = x[1] - wn_bits_f32(h & 0xffff0000u);
        const unsigned md = wn_pk_bf16(r0, r1);
        const unsigned lo = wn_pk_bf16(r0 - wn_bits_f32(md << 16), r1 - wn_bits_f32(md & 0xffff0000u));
        unsigned* d = reinterpret_cast<unsigned*>(Apk);
        d[(((long)kb * 3 + 0) * Mpad + m) * 8 + e / 2] = h;
        d[(((long)kb * 3 + 1) * Mpad + m) * 8 + e / 2] = md;
        d[(((long)kb * 3 + 2) * Mpad + m) * 8 + e / 2] = lo;
    }
}

int wn_gemm6_pack(const float* src, long lda, int M, int K, unsigned short* Apk, wn_stream_t st) {
    WN_PROF("gemm6_pack", 0.0, 0.0, st);
    const int Mpad = (M + WN_G6_BM - 1) / WN_G6_BM * WN_G6_BM;
    const long n = (long)((K + 15) / 16) * Mpad;
    WN_LAUNCH(k_gemm6_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, lda, M, K, Mpad, Apk);
    return 0;
}

__global__ __launch_bounds__(G6_T, 2) void k_gemm6(WnGemm6Args g) {
    WN_DYN_SMEM(smem_raw);
    // stage s: A pieces [3][256][16] bf16 (24 KB) then B pieces [3][128][16] bf16 (12 KB)
    constexpr int A_BYTES = 3 * WN_G6_BM * 32, B_BYTES = 3 * WN_G6_BN * 32, ST_BYTES = A_BYTES + B_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int b = blockIdx.z;
    const int m0 = blockIdx.y * WN_G6_BM, n0 = blockIdx.x * WN_G6_BN;
    const float* __restrict__ Bz = g.B + (long)b * g.b_zstride;
    const char* Ab = reinterpret_cast<const char*>(g.Apk);
    const int nk = (g.K + 15) / 16;
    const bool one_seg = g.b_seg_len >= g.K;

    // staging registers
    wn_f4 ra[6];
    float rb[8];
    const int bn = tid & 127, bkh = tid >> 7;  // this thread's B column and k half (8 k values)
    const bool n_ok = (n0 + bn) < g.N;
    auto fetch = [&](int kb) {
        WN_UNROLL
        for (int p = 0; p < 3; ++p) {
            const char* src = Ab + (((long)kb * 3 + p) * g.Mpad + m0) * 32 + tid * 16;
            ra[2 * p] = *reinterpret_cast<const wn_f4*>(src);
            ra[2 * p + 1] = *reinterpret_cast<const wn_f4*>(src + 4096);
        }
        const int k0 = kb * 16;
        int seg = 0, rr0 = k0;
        if (!one_seg) {
            seg = k0 / g.b_seg_len;
            rr0 = k0 - seg * g.b_seg_len;
        }
        // buffer loads: rows past K / columns past N get an out-of-range offset and read as 0 without
        // a branch (a predicated load becomes a branch with its own vmcnt(0) and serialises the loads)
        const int rows = one_seg ? g.K : g.b_seg_len;
        const wn_rsrc_t Br = wn_make_buf(Bz + (long)seg * g.b_seg_stride, (unsigned)((long)rows * g.ldb * 4));
        const int krem = g.K - (k0 + 8 * bkh);  // valid rows of this thread's 8
        const int base = ((rr0 + 8 * bkh) * (int)g.ldb + n0 + bn) * 4;
        WN_UNROLL
        for (int e = 0; e < 8; ++e) rb[e] = wn_buf_load(Br, (n_ok && e < krem) ? base + e * (int)g.ldb * 4 : 0x7ffffff0, 0);
    };
    auto stage = [&](int st) {
        char* sa = smem_raw + st * ST_BYTES;
        WN_UNROLL
        for (int p = 0; p < 3; ++p) {
            *reinterpret_cast<wn_f4*>(sa + p * (WN_G6_BM * 32) + tid * 16) = ra[2 * p];
            *reinterpret_cast<wn_f4*>(sa + p * (WN_G6_BM * 32) + 4096 + tid * 16) = ra[2 * p + 1];
        }
        unsigned h[4], md[4], lo[4];
        WN_UNROLL
        for (int q = 0; q < 4; ++q) {
            const float x0 = rb[2 * q], x1 = rb[2 * q + 1];
            h[q] = wn_pk_bf16(x0, x1);
            const float r0 = x0 - wn_bits_f32(h[q] << 16), r1 = x1 - wn_bits_f32(h[q] & 0xffff0000u);
            md[q] = wn_pk_bf16(r0, r1);
            lo[q] = wn_pk_bf16(r0 - wn_bits_f32(md[q] << 16), r1 - wn_bits_f32(md[q] & 0xffff0000u));
        }
        char* sb = sa + A_BYTES + bn * 32 + bkh * 16;
        wn_f4 v;
        v.x = wn_bits_f32(h[0]); v.y = wn_bits_f32(h[1]); v.z = wn_bits_f32(h[2]); v.w = wn_bits_f32(h[3]);
        *reinterpret_cast<wn_f4*>(sb) = v;
        v.x = wn_bits_f32(md[0]); v.y = wn_bits_f32(md[1]); v.z = wn_bits_f32(md[2]); v.w = wn_bits_f32(md[3]);
        *reinterpret_cast<wn_f4*>(sb + WN_G6_BN * 32) = v;
        v.x = wn_bits_f32(lo[0]); v.y = wn_bits_f32(lo[1]); v.z = wn_bits_f32(lo[2]); v.w = wn_bits_f32(lo[3]);
        *reinterpret_cast<wn_f4*>(sb + 2 * WN_G6_BN * 32) = v;
    };

    f32x16 acc[4][2];
    WN_UNROLL
    for (int i = 0; i < 4; ++i) {
        acc[i][0] = f32x16_zero();
        acc[i][1] = f32x16_zero();
    }
    if (nk > 0) {
        fetch(0);
        stage(0);
    }
    __syncthreads();
    for (int kb = 0; kb < nk; ++kb) {
        const bool more = kb + 1 < nk;
        if (more) fetch(kb + 1);
        WN_SCHED_BARRIER();  // the loads of the next step stay in flight during the MFMAs of this one
        const char* sa = smem_raw + (kb & 1) * ST_BYTES;
        const char* sb = sa + A_BYTES;
        wn_f4 bf[3][2];
        WN_UNROLL
        for (int p = 0; p < 3; ++p) {
            WN_UNROLL
            for (int j = 0; j < 2; ++j)
                bf[p][j] = *reinterpret_cast<const wn_f4*>(sb + p * (WN_G6_BN * 32) + (64 * wn + 32 * j + li) * 32 + hi * 16);
        }
        WN_UNROLL
        for (int i = 0; i < 4; ++i) {
            wn_f4 af[3];
            WN_UNROLL
            for (int p = 0; p < 3; ++p)
                af[p] = *reinterpret_cast<const wn_f4*>(sa + p * (WN_G6_BM * 32) + (128 * wm + 32 * i + li) * 32 + hi * 16);
            WN_UNROLL
            for (int j = 0; j < 2; ++j) {
                f32x16 c = acc[i][j];
                c = mfma_bf16(af[0], bf[2][j], c);  // small terms first
                c = mfma_bf16(af[2], bf[0][j], c);
                c = mfma_bf16(af[1], bf[1][j], c);
                c = mfma_bf16(af[0], bf[1][j], c);
                c = mfma_bf16(af[1], bf[0][j], c);
                c = mfma_bf16(af[0], bf[0][j], c);
                acc[i][j] = c;
            }
        }
        WN_SCHED_BARRIER();
        if (more) stage((kb + 1) & 1);
        __syncthreads();
    }

    // epilogue: bias, mask, relu; rows of a lane are (r&3) + 8*(r>>2) + 4*hi, its column is li.
    // Buffer accesses with out-of-range offsets for the ragged edges (reads give 0, writes are dropped).
    const wn_rsrc_t Cr = wn_make_buf(g.C + (long)b * g.c_zstride, (unsigned)((long)g.M * g.ldc * 4));
    const wn_rsrc_t Er = wn_make_buf(g.E ? g.E + (long)b * g.e_zstride : g.C, g.E ? (unsigned)((long)g.M * g.lde * 4) : 0u);
    const wn_rsrc_t Biasr = wn_make_buf(g.bias ? g.bias : g.C, g.bias ? (unsigned)(g.M * 4) : 0u);
    WN_UNROLL
    for (int i = 0; i < 4; ++i) {
        float bv[16];  // bias of this lane's 16 rows (0 when absent: out-of-range reads)
        WN_UNROLL
        for (int r = 0; r < 16; ++r) bv[r] = wn_buf_load(Biasr, (m0 + 128 * wm + 32 * i + mfma32_row(r, hi)) * 4, 0);
        WN_UNROLL
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + li;
            float ev[16];
            if (g.E) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 128 * wm + 32 * i + mfma32_row(r, hi);
                    ev[r] = wn_buf_load(Er, (row < g.M && col < g.N) ? (row * (int)g.lde + col) * 4 : 0x7ffffff0, 0);
                }
            }
            WN_SCHED_BARRIER();
            WN_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 128 * wm + 32 * i + mfma32_row(r, hi);
                float v = acc[i][j][r];
                v += bv[r];
                if (g.E) v = (ev[r] > 0.f) ? v : 0.f;
                if (g.relu) v = fmaxf(v, 0.f);
                wn_buf_store(Cr, v, (row < g.M && col < g.N) ? (row * (int)g.ldc + col) * 4 : 0x7ffffff0, 0);
            }
        }
    }
}

int wn_gemm6_launch(const WnGemm6Args* gp, wn_stream_t st) {
    const WnGemm6Args& g = *gp;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.nbatch <= 0) return 1;
    if (g.b_seg_len < g.K && (g.b_seg_len % 16) != 0) return 2;
    constexpr int lds = 2 * (3 * WN_G6_BM * 32 + 3 * WN_G6_BN * 32);
#ifndef WN_EMU
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm6), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess)
            return 3;
        attr_set = true;
    }
#endif
    WN_PROF(g.tag ? g.tag : "gemm6", 2.0 * g.M * g.N * (double)g.K * g.nbatch,
            ((double)g.M * g.K * 6.0 + (double)g.K * g.N * 4.0 + (double)g.M * g.N * (g.E ? 8.0 : 4.0)) * g.nbatch, st);
    dim3 grid((unsigned)((g.N + WN_G6_BN - 1) / WN_G6_BN), (unsigned)(g.Mpad / WN_G6_BM), (unsigned)g.nbatch);
    WN_LAUNCH(k_gemm6, grid, dim3(G6_T), lds, st, g);
    return 0;
}
