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

#include <stdlib.h>

#include "wn_gemm.h"
#include "wn_prof.h"

#define G6_T 256

static __device__ __forceinline__ void gemm6_pack_elem(const float* src, long lda, int M, int K, int Mpad, unsigned short* Apk,
                                                       int gate_R, long idx, int f16 = 0);
__global__ void k_gemm6_pack(const float* src, long lda, int M, int K, int Mpad, unsigned short* Apk, int gate_R) {
    gemm6_pack_elem(src, lda, M, K, Mpad, Apk, gate_R, (long)blockIdx.x * 256 + threadIdx.x);
}
__global__ void k_gemm6_pack_batch(WnGemm6PackJobs a) {
    int j = 0;
    while (j + 1 < a.njobs && (int)blockIdx.x >= a.blk0[j + 1]) ++j;   // block-uniform
    const int Mpad = (a.M[j] + WN_G6_BM - 1) / WN_G6_BM * WN_G6_BM;
    const int per = (int)(((long)((a.K[j] + 15) / 16) * Mpad + 255) / 256);   // blocks per weight set of this job
    const int rel = (int)blockIdx.x - a.blk0[j], li = rel / per;
    gemm6_pack_elem(a.src[j] + (long)li * a.src_lstride[j], a.lda[j], a.M[j], a.K[j], Mpad, a.dst[j] + (long)li * a.dst_lstride[j],
                    a.gate_R[j], (long)(rel - li * per) * 256 + threadIdx.x, a.f16[j]);
}
// f16 != 0: the fp16 pair split of k_gemm6<.., F16> -- TWO pieces per value (x ~ h + l, 11 + 11 significand bits), laid out
// [kb][2][Mpad][16] (two thirds of the bf16 image)
static __device__ __forceinline__ void gemm6_pack_elem(const float* src, long lda, int M, int K, int Mpad, unsigned short* Apk,
                                                       int gate_R, long idx, int f16) {
    // one thread per (kb, m): 16 k values -> 3 x 16 bf16
    const int nkb = (K + 15) / 16;
    if (idx >= (long)nkb * Mpad) return;
    const int kb = (int)(idx / Mpad), m = (int)(idx % Mpad);
    const int ms = (gate_R > 0 && m < M) ? wn_gemm6_gate_row(m, gate_R) : m;   // source row of packed row m
    // the row's 3 x 32 bytes are collected in registers and written as 16-byte stores: with one 4-byte store per pair a wave wrote
    // 4 bytes into each of 64 rows per instruction (3.5 GB of traffic for 190 MB of weights at n_resch 512, profiles/r05)
    unsigned hh[8], mm[8], ll[8];
    WN_UNROLL
    for (int e = 0; e < 16; e += 2) {
        float x[2];
        WN_UNROLL
        for (int u = 0; u < 2; ++u) {
            const int k = kb * 16 + e + u;
            x[u] = (m < M && k < K) ? src[(long)k * lda + ms] : 0.f;
        }
        if (f16) {   // weights times 2^6: a second piece stays a normal fp16 number down to |w| = 2^-9 (full 22 bits); the kernel divides it out
            x[0] *= (float)WN_G6_F16_WSCALE;
            x[1] *= (float)WN_G6_F16_WSCALE;
            const unsigned h = wn_pk_f16(x[0], x[1]);
            hh[e / 2] = h;
            mm[e / 2] = wn_pk_f16(x[0] - wn_f16lo_f32(h), x[1] - wn_f16hi_f32(h));
            ll[e / 2] = 0u;
            continue;
        }
        const unsigned h = wn_pk_bf16(x[0], x[1]);
        const float r0 = x[0] - wn_bits_f32(h << 16), r1 = x[1] - wn_bits_f32(h & 0xffff0000u);
        const unsigned md = wn_pk_bf16(r0, r1);
        hh[e / 2] = h;
        mm[e / 2] = md;
        ll[e / 2] = wn_pk_bf16(r0 - wn_bits_f32(md << 16), r1 - wn_bits_f32(md & 0xffff0000u));
    }
    unsigned* d = reinterpret_cast<unsigned*>(Apk);
    const unsigned* src3[3] = {hh, mm, ll};
    const int np = f16 ? 2 : 3;
    for (int p = 0; p < np; ++p) {
        wn_f4* dst = reinterpret_cast<wn_f4*>(d + (((long)kb * np + p) * Mpad + m) * 8);   // 32-byte rows: 16-byte aligned
        WN_UNROLL
        for (int q = 0; q < 2; ++q) {
            wn_f4 v;
            v.x = wn_bits_f32(src3[p][4 * q]); v.y = wn_bits_f32(src3[p][4 * q + 1]);
            v.z = wn_bits_f32(src3[p][4 * q + 2]); v.w = wn_bits_f32(src3[p][4 * q + 3]);
            dst[q] = v;
        }
    }
}

int wn_gemm6_pack_batch(WnGemm6PackJobs* jobs, wn_stream_t st) {
    WN_PROF("gemm6_pack", 0.0, 0.0, st);
    int nblk = 0;
    for (int j = 0; j < jobs->njobs; ++j) {
        const int Mpad = (jobs->M[j] + WN_G6_BM - 1) / WN_G6_BM * WN_G6_BM;
        jobs->blk0[j] = nblk;
        nblk += (int)(((long)((jobs->K[j] + 15) / 16) * Mpad + 255) / 256) * jobs->nl[j];
    }
    jobs->blk0[jobs->njobs] = nblk;
    if (nblk <= 0) return 0;
    WN_LAUNCH(k_gemm6_pack_batch, dim3((unsigned)nblk), dim3(256), 0, st, *jobs);
    return 0;
}

int wn_gemm6_pack(const float* src, long lda, int M, int K, unsigned short* Apk, int gate_R, wn_stream_t st) {
    WN_PROF("gemm6_pack", 0.0, 0.0, st);
    if (gate_R > 0 && (M != 2 * gate_R || gate_R % 128 != 0)) return 1;
    const int Mpad = (M + WN_G6_BM - 1) / WN_G6_BM * WN_G6_BM;
    const long n = (long)((K + 15) / 16) * Mpad;
    WN_LAUNCH(k_gemm6_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, lda, M, K, Mpad, Apk, gate_R);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// block order.  Workgroups are dealt to the 8 XCDs round-robin by their linear id (x fastest), and every XCD has its own
// L2.  The blocks that read the SAME operand tile (the M-tiles of one time tile in k_gemm6, the N- and M-tiles of one
// k-chunk in k_gemm6_dw) differ in ONE grid coordinate, so with the plain order they land on different XCDs -- or on the
// same one a whole grid row later -- and each fetches the shared tile from HBM again (bwd_dz_skip_all: dSkip 8 times).
// `wn_block_order` renumbers: XCD j gets the contiguous range [j * per, (j + 1) * per) of a logical order in which the
// sharing coordinate runs fastest, so the sharers are neighbours in time on one XCD and the tile comes out of its L2.
//   order 0: plain blockIdx; 1: logical order x, y, z (x fastest); 2: logical order y, x, z (y fastest).
// ---------------------------------------------------------------------------------------------
struct WnBlock { int x, y, z; };
__device__ inline WnBlock wn_block_order(int order) {
    WnBlock r;
    r.x = (int)blockIdx.x; r.y = (int)blockIdx.y; r.z = (int)blockIdx.z;
    if (order == 0) return r;
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned per = total >> 3;
    unsigned p = lin;
    if (lin < (per << 3)) p = (lin & 7u) * per + (lin >> 3);   // the last total % 8 blocks keep their id
    if (order == 1) {
        r.x = (int)(p % gx); p /= gx;
        r.y = (int)(p % gy); r.z = (int)(p / gy);
    } else {
        r.y = (int)(p % gy); p /= gy;
        r.x = (int)(p % gx); r.z = (int)(p / gx);
    }
    return r;
}
static int xcd_block_order() { return 1; }   // (measured against plain blockIdx: PMC 43.3 -> 38.9 GB per step, profiles/r02)

// Timing builds only (tools/gemm_timing.py, -DWN_TIMING): cycle stamps of the k-loop phases, waves of the block that
// gets logical tile (0, 0, 0), first 24 steps, for the launches whose tag was selected with wn_debug_gemm6().
#ifdef WN_TIMING
#include <string.h>
static long long* g6_dbg_buf = nullptr;
static char g6_dbg_tag[64] = "";
extern "C" void wn_debug_gemm6(void* buf, const char* tag) {
    g6_dbg_buf = (long long*)buf;
    strncpy(g6_dbg_tag, tag ? tag : "", sizeof(g6_dbg_tag) - 1);
}
static long long* g6_dbg_for(const char* tag) { return (g6_dbg_buf && tag && strcmp(tag, g6_dbg_tag) == 0) ? g6_dbg_buf : nullptr; }
#define G6_DBG_PARAM , long long* dbg
#define G6_DBG_ARG(tag) , g6_dbg_for(tag)
#define G6_STAMP(step, slot)                                                                                          \
    do {                                                                                                              \
        if (dbg && dbg_blk && (threadIdx.x & 63) == 0 && (step) < 24)                                                  \
            dbg[(threadIdx.x >> 6) * 256 + (step) * 8 + (slot)] = (long long)__builtin_readcyclecounter();            \
    } while (0)
#else
#define G6_DBG_PARAM
#define G6_DBG_ARG(tag)
#define G6_STAMP(step, slot)
#endif

// CE = true: the instantiation with the softmax cross-entropy epilogue (its own kernel, so that the register allocation of
// the plain one does not depend on it)
// F16 (round 6, WN_FLAG_MM_F16PAIR): TWO fp16 pieces per operand (x ~ h + l, 11 + 11 significand bits) and the three products
// h h + h l + l h on v_mfma_f32_32x32x16_f16 -- ~2^-22 |a b| per product, the rounding of an fp32 running sum over >= 64 terms -- at
// half the matrix work and two thirds of the LDS traffic of the six bf16 products.  fp16 has 5 exponent bits: the B operand (an
// activation, or a gradient ~1e-5 ... 1e-8) is multiplied by g.b_mul (a power of two; < 0: the scale wn_dw_prepare left at
// ((float*)g.ovf)[1], i.e. 2^8 / max |dlogits| measured) at the split and the accumulators by its inverse; a value beyond fp16's
// range makes the block's accumulators non-finite: the epilogue raises *g.ovf, and the six-product launch issued behind every
// fp16 launch (this kernel with F16 = false and g.ovf set: it returns at its first instruction otherwise) redoes the contraction.
template <bool CE, bool F16 = false>
__global__ __launch_bounds__(G6_T, 2) void k_gemm6(WnGemm6Args g, int order G6_DBG_PARAM) {
    WN_DYN_SMEM(smem_raw);
    if (!F16 && g.ovf != nullptr && wn_load_coherent_int(g.ovf) == 0) return;   // the conditional redo behind an fp16 launch
    constexpr int NP = F16 ? 2 : 3;          // pieces per operand
    constexpr int NPROD = F16 ? 3 : 6;       // products per multiply
    // stage s: A pieces [NP][256][16] (24 / 16 KB) then B pieces [NP][128][16] (12 / 8 KB)
    constexpr int A_BYTES = NP * WN_G6_BM * 32, B_BYTES = NP * WN_G6_BN * 32, ST_BYTES = A_BYTES + B_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const WnBlock blk = wn_block_order(order);
    const int b = WN_UNIFORM(blk.z);
    const int m0 = WN_UNIFORM(blk.y) * WN_G6_BM, n0 = WN_UNIFORM(blk.x) * WN_G6_BN;
#ifdef WN_TIMING
    const bool dbg_blk = (blk.x | blk.y | blk.z) == 0;
#endif
    const float* __restrict__ Bz = g.B + (long)b * g.b_zstride;
    const int nk = (g.K + 15) / 16;
    const bool one_seg = g.b_seg_len >= g.K;
    // Sign of this column tile's arithmetic: odd column tiles contract -B and negate the result.  The matrix core aligns the
    // products of a step against the accumulator by truncation, which leaves a small bias of CONSTANT (negative) sign on every
    // output -- measured on the recipe-size backward pass: -1e-7 of the mean magnitude after a K = 2048 contraction, -1.4e-8
    // more per layer, which a bias-type gradient (a sum over every position) turned into 6.2e-5 of its maximum
    // (profiles/r04/wide_drift_*.txt).  With the operand negated the bias changes sign with it, so over the positions of a
    // sequence it cancels instead of adding up (same probe: 9.0e-6).  Exact in every other respect: -x splits into the
    // negated pieces of x.
    const float fs = ((blk.x + g.n_phase) & 1) ? -1.0f : 1.0f;
    float b_mul = 1.0f;
    if (F16) b_mul = g.b_mul < 0.0f ? reinterpret_cast<const float*>(g.ovf)[1] : g.b_mul;   // (one uniform load)
    const float fsm = fs * b_mul;            // what a B element is multiplied by at the split
    const float fsc = F16 ? fs / (b_mul * (float)WN_G6_F16_WSCALE) : fs; // ... and an accumulator in the epilogue (powers of two)
    // De-phase the two blocks that share a CU (WN_G6_STAGGER, A/B knob).  They start together, do the same work and so
    // stay in lock step: both in their prologue (HBM latency) and both in their epilogue (stores) at the same time, with the
    // matrix pipe idle.  The second resident of the first round -- its waves sit in wave slot 1 of their SIMDs -- starts
    // late by a fraction of a block; every later block inherits the offset of the slot it takes over.
    if (g.stagger > 0 && blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z) < 512u && (WN_HW_WAVE_SLOT() & 1))
        for (int i = 0; i < g.stagger; ++i) WN_SLEEP(127);

    // the split weights go global -> LDS directly (their packed layout IS the LDS layout)
    const wn_rsrc_t Ar = wn_make_buf(g.Apk, (unsigned)((long)nk * NP * g.Mpad * 32));
    const int wave_u = WN_UNIFORM(wave);
    float rb0[8], rb1[8];  // activations are fetched two steps ahead (HBM latency), weights one (L2)
    const int bn = tid & 127, bkh = tid >> 7;  // this thread's B column and k half (8 k values)
    const bool n_ok = (n0 + bn) < g.N;
    const int a_voff = (tid >> 1) * 32 + (wn_frag_off(tid >> 1, tid & 1) & 16);   // global byte offset of that k half
    // Loads never leave the operands: a step index past the end is clamped to the last step (its data is then simply not
    // used), a row past K to row K - 1 (the packed weights are zero there, k_gemm6_pack), so nothing in the loop depends on
    // how the hardware range-checks the scalar offset.  Only the per-lane COLUMN offset carries an out-of-range marker
    // (zero history / columns past N read as 0).
    auto fetch_a = [&](int kb, int st) {
        char* sa = smem_raw + st * ST_BYTES + wave_u * 1024;
        const int kc = kb < nk ? kb : nk - 1;
        WN_UNROLL
        for (int p = 0; p < NP; ++p) {
            const unsigned src = (unsigned)((kc * NP + p) * g.Mpad + m0) * 32u;
            // slot tid of the piece = row tid >> 1, stored half tid & 1, which holds the k half wn_frag_off says
            wn_buf_load_lds16(Ar, sa + p * (WN_G6_BM * 32), a_voff, src);
            wn_buf_load_lds16(Ar, sa + p * (WN_G6_BM * 32) + 4096, a_voff, src + 4096u);
        }
    };
    const int bkh_u = WN_UNIFORM(bkh);
    // fetch_b is called for the steps 0, 1, 2, ... in order: the segment and the row inside it advance with the calls
    // (no division, no branch: the step function must stay ONE basic block for its scheduling fences to mean anything)
    const int seg_rows = one_seg ? nk * 16 : g.b_seg_len;
    int fb_k0 = 0, fb_seg = 0, fb_rr = 0;
    auto fetch_b = [&](float (&rb)[8]) {
        const wn_rsrc_t Br = wn_make_buf(Bz + (long)fb_seg * g.b_seg_stride, (unsigned)((long)(one_seg ? g.K : g.b_seg_len) * g.ldb * 4));
        const int cc = n0 + bn - (g.b_shift0 + fb_seg * g.b_shift_step);  // shifted column: zero history outside [0, clen)
        const int voff = (n_ok && cc >= 0 && cc < g.b_clen) ? cc * 4 : 0x7ffffff0;
        const int rlast = g.K - 1 - (fb_k0 - fb_rr);   // last valid row of this segment (>= 0: the step starts inside K)
        WN_UNROLL
        for (int e = 0; e < 8; ++e) {   // the row offsets are wave-uniform: scalar registers, no vector arithmetic per load
            int r = fb_rr + 8 * bkh_u + e;
            r = r < rlast ? r : rlast;
            rb[e] = wn_buf_load(Br, voff, r * (int)g.ldb * 4);
        }
        // next step; a step past the end stays on the last one (its data is not used)
        const bool more = fb_k0 + 16 < nk * 16;
        const bool wrap = fb_rr + 16 >= seg_rows;
        fb_k0 = more ? fb_k0 + 16 : fb_k0;
        fb_seg = (more && wrap) ? fb_seg + 1 : fb_seg;
        fb_rr = more ? (wrap ? 0 : fb_rr + 16) : fb_rr;
    };
    // split of one pair of this thread's 8 activations into its three bf16 pieces (9 VALU instructions)
    auto split_pair = [&](int q, const float (&rb)[8], unsigned (&h)[4], unsigned (&md)[4], unsigned (&lo)[4]) {
        const float x0 = rb[2 * q] * fsm, x1 = rb[2 * q + 1] * fsm;
        if constexpr (F16) {
            h[q] = wn_pk_f16(x0, x1);
            md[q] = wn_pk_f16(x0 - wn_f16lo_f32(h[q]), x1 - wn_f16hi_f32(h[q]));
            lo[q] = 0u;
            return;
        }
        h[q] = wn_pk_bf16(x0, x1);
        const float r0 = x0 - wn_bits_f32(h[q] << 16), r1 = x1 - wn_bits_f32(h[q] & 0xffff0000u);
        md[q] = wn_pk_bf16(r0, r1);
        lo[q] = wn_pk_bf16(r0 - wn_bits_f32(md[q] << 16), r1 - wn_bits_f32(md[q] & 0xffff0000u));
    };
    auto write_pieces = [&](int st, const unsigned (&h)[4], const unsigned (&md)[4], const unsigned (&lo)[4]) {
        char* sb = smem_raw + st * ST_BYTES + A_BYTES + wn_frag_off(bn, bkh);
        wn_f4 v;
        v.x = wn_bits_f32(h[0]); v.y = wn_bits_f32(h[1]); v.z = wn_bits_f32(h[2]); v.w = wn_bits_f32(h[3]);
        *reinterpret_cast<wn_f4*>(sb) = v;
        v.x = wn_bits_f32(md[0]); v.y = wn_bits_f32(md[1]); v.z = wn_bits_f32(md[2]); v.w = wn_bits_f32(md[3]);
        *reinterpret_cast<wn_f4*>(sb + WN_G6_BN * 32) = v;
        if constexpr (!F16) {
            v.x = wn_bits_f32(lo[0]); v.y = wn_bits_f32(lo[1]); v.z = wn_bits_f32(lo[2]); v.w = wn_bits_f32(lo[3]);
            *reinterpret_cast<wn_f4*>(sb + 2 * WN_G6_BN * 32) = v;
        }
    };
    auto stage = [&](int st, const float (&rb)[8]) {
        unsigned h[4], md[4], lo[4];
        WN_UNROLL
        for (int q = 0; q < 4; ++q) split_pair(q, rb, h, md, lo);
        write_pieces(st, h, md, lo);
    };

    f32x16 acc[4][2];
    WN_UNROLL
    for (int i = 0; i < 4; ++i) {
        acc[i][0] = f32x16_zero();
        acc[i][1] = f32x16_zero();
    }
    // One k-step.  The 48 MFMAs on LDS stage `st` carry everything else of the step in their shadow: a wave's own
    // independent VALU / memory instructions issue between its MFMAs for free (measured, tools/microbench/mfma_valu.hip: up
    // to 6 VALU per 32x32x16 MFMA at no cost), whereas as separate phases before / after the MFMAs the same instructions
    // cost the wave ~1000 (loads) + ~1500 (split) cycles per step in which it issued no MFMA (tools/gemm_timing.py).
    //   after row tile 0: the weight slab of step `ka` -> LDS stage stn (6 LDS-DMA instructions), split of pair 0
    //   after row tile 1: the activations of the step after next -> registers rbn (8 loads),     split of pair 1
    //   after row tiles 2, 3: split of pairs 2, 3;   then the three LDS writes of the split pieces (stage stn)
    // The fences pin this order for VALU, MFMA and memory instructions; LDS reads (the next row tile's fragments) and scalar
    // instructions may cross them.  The slab is issued BEFORE the activation loads, so "at most 8 loads outstanding" at the
    // end of the step == "the weight slab has landed in LDS" (vmcnt retires in order).
    // One k-step.  The 48 MFMAs on LDS stage `st` carry everything else of the step in their shadow, ONE piece after every pair
    // of MFMAs (24 slots per step).  A wave's own independent VALU / memory instructions issue between its MFMAs for free
    // (tools/microbench/mfma_valu.hip), but only if they really sit BETWEEN them: with one slice per row tile (round 2/3) the
    // ISA showed the 12 MFMAs of a tile issued as a burst and the slice's 20 - 50 instructions behind them with the matrix pipe
    // draining (round 4, same box: fwd_skip_sum 0.78 -> 0.70 ms, recipe-size step 134.8 -> 128.7 ms,
    // profiles/r04/ab_gemm6_fine_noslp.txt).  The fences pin the order for VALU, MFMA and memory instructions; LDS reads (the
    // next row tile's fragments) and scalar instructions may cross them.  The weight slab is issued BEFORE the activation
    // loads, so "at most 8 loads outstanding" at the end of the step == "the slab has landed in LDS" (vmcnt retires in order).
    //   slots 0-5: the weight slab (6 LDS-DMA pieces)   6-13: the 8 activation loads   14-21: the operand split, half a pair
    //   per slot   22: the three LDS writes of the split pieces
    auto step = [&](int st, int stn, const float (&rb)[8], float (&rbn)[8], int ka, bool last) {
        const char* sa = smem_raw + st * ST_BYTES;
        const char* sb = sa + A_BYTES;
        unsigned h[4], md[4], lo[4];
        float r0[4], r1[4];
        wn_f4 bf[NP][2];
        WN_UNROLL
        for (int p = 0; p < NP; ++p) {
            WN_UNROLL
            for (int j = 0; j < 2; ++j)
                bf[p][j] = *reinterpret_cast<const wn_f4*>(sb + p * (WN_G6_BN * 32) + wn_frag_off(64 * wn + 32 * j + li, hi));
        }
        // state of the activation loads of this step (see fetch_b)
        const wn_rsrc_t Br = wn_make_buf(Bz + (long)fb_seg * g.b_seg_stride, (unsigned)((long)(one_seg ? g.K : g.b_seg_len) * g.ldb * 4));
        const int cc = n0 + bn - (g.b_shift0 + fb_seg * g.b_shift_step);
        const int voff = (n_ok && cc >= 0 && cc < g.b_clen) ? cc * 4 : 0x7ffffff0;
        const int rlast = g.K - 1 - (fb_k0 - fb_rr);
        char* sda = smem_raw + stn * ST_BYTES + wave_u * 1024;
        const int kc = ka < nk ? ka : nk - 1;
        WN_UNROLL
        for (int i = 0; i < 4; ++i) {
            wn_f4 af[NP];
            WN_UNROLL
            for (int p = 0; p < NP; ++p)
                af[p] = *reinterpret_cast<const wn_f4*>(sa + p * (WN_G6_BM * 32) + wn_frag_off(128 * wm + 32 * i + li, hi));
            // small terms first; F16: h l, l h, h h
            constexpr int PA[6] = {0, F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0}, PB[6] = {F16 ? 1 : 2, 0, F16 ? 0 : 1, 1, 0, 0};
            WN_UNROLL
            for (int t = 0; t < NPROD; ++t) {
                WN_UNROLL
                for (int j = 0; j < 2; ++j) {
                    if constexpr (F16) acc[i][j] = mfma_f16(af[PA[t]], bf[PB[t]][j], acc[i][j]);
                    else acc[i][j] = mfma_bf16(af[PA[t]], bf[PB[t]][j], acc[i][j]);
                }
                if (last) continue;   // the final step has nothing to prepare
                const int sl = i * NPROD + t;
#ifdef WN_G6_NO_PAIR_INTERLEAVE   // (round 4's order: the pair of MFMAs, then its piece)
                WN_SCHED_FENCE_ALU();
#endif
                if constexpr (F16) {
                    // 12 slots: 0-3 the weight slab (4 LDS-DMA pieces), 4-7 the 8 activation loads (two per slot), 8-11 the operand
                    // split, one pair per slot, and the two LDS writes of the pieces behind the last pair
                    if (sl < 4) {
                        const int p = sl >> 1;
                        const unsigned src = (unsigned)((kc * NP + p) * g.Mpad + m0) * 32u + ((sl & 1) ? 4096u : 0u);
                        wn_buf_load_lds16(Ar, sda + p * (WN_G6_BM * 32) + ((sl & 1) ? 4096 : 0), a_voff, src);
                    } else if (sl < 8) {
                        WN_UNROLL
                        for (int u = 0; u < 2; ++u) {
                            const int e = 2 * (sl - 4) + u;
                            int r = fb_rr + 8 * bkh_u + e;
                            r = r < rlast ? r : rlast;
                            rbn[e] = wn_buf_load(Br, voff, r * (int)g.ldb * 4);
                        }
                    } else {
                        const int q = sl - 8;
                        const float x0 = rb[2 * q] * fsm, x1 = rb[2 * q + 1] * fsm;
                        h[q] = wn_pk_f16(x0, x1);
                        md[q] = wn_pk_f16(x0 - wn_f16lo_f32(h[q]), x1 - wn_f16hi_f32(h[q]));
                        if (sl == 11) write_pieces(stn, h, md, lo);
                    }
                } else if (sl < 6) {
                    const int p = sl >> 1;
                    const unsigned src = (unsigned)((kc * 3 + p) * g.Mpad + m0) * 32u + ((sl & 1) ? 4096u : 0u);
                    wn_buf_load_lds16(Ar, sda + p * (WN_G6_BM * 32) + ((sl & 1) ? 4096 : 0), a_voff, src);
                } else if (sl < 14) {
                    const int e = sl - 6;
                    int r = fb_rr + 8 * bkh_u + e;
                    r = r < rlast ? r : rlast;
                    rbn[e] = wn_buf_load(Br, voff, r * (int)g.ldb * 4);
                } else if (sl < 22) {
                    const int q = (sl - 14) >> 1;
                    if (((sl - 14) & 1) == 0) {
                        const float x0 = rb[2 * q] * fs, x1 = rb[2 * q + 1] * fs;
                        h[q] = wn_pk_bf16(x0, x1);
                        r0[q] = x0 - wn_bits_f32(h[q] << 16);
                        r1[q] = x1 - wn_bits_f32(h[q] & 0xffff0000u);
                    } else {
                        md[q] = wn_pk_bf16(r0[q], r1[q]);
                        lo[q] = wn_pk_bf16(r0[q] - wn_bits_f32(md[q] << 16), r1[q] - wn_bits_f32(md[q] & 0xffff0000u));
                    }
                } else if (sl == 22) {
                    write_pieces(stn, h, md, lo);
                }
#ifndef WN_G6_NO_PAIR_INTERLEAVE   // the pair of MFMAs and its piece as MFMA, <= 6 VALU, MFMA, the rest (round 5: -1 % on the long
                                   // contractions of the recipe-size model, profiles/r05/ab_gemm6_pair_interleave.txt)
                WN_SGB_MFMA(1);
                WN_SGB_VALU(6);
                WN_SGB_MFMA(1);
#endif
                WN_SCHED_FENCE_ALU();
            }
        }
        if (!last) {   // next step of the activation loads; a step past the end stays on the last one (its data is not used)
            const bool more = fb_k0 + 16 < nk * 16;
            const bool wrap = fb_rr + 16 >= seg_rows;
            fb_k0 = more ? fb_k0 + 16 : fb_k0;
            fb_seg = (more && wrap) ? fb_seg + 1 : fb_seg;
            fb_rr = more ? (wrap ? 0 : fb_rr + 16) : fb_rr;
        }
    };
    // Steps are processed in pairs with the two register sets swapping roles.  Entering a pair (kb, kb + 1): LDS stage 0
    // holds step kb (weights and split activations), rb1 the activations of step kb + 1, rb0 is free.
    fetch_a(0, 0);
    fetch_b(rb0);
    fetch_b(rb1);
    stage(0, rb0);
    WN_WAIT_VMCNT(8);
    __syncthreads();
    const int npair = nk >> 1;
    for (int kp = 0; kp < npair; ++kp) {
        const int kb = 2 * kp;
        G6_STAMP(kb, 0);
        step(0, 1, rb1, rb0, kb + 1, false);
        G6_STAMP(kb, 3);
        WN_WAIT_VMCNT(8);
        G6_STAMP(kb, 4);
        __syncthreads();
        G6_STAMP(kb, 5);
        step(1, 0, rb0, rb1, kb + 2, false);
        G6_STAMP(kb + 1, 3);
        WN_WAIT_VMCNT(8);
        G6_STAMP(kb + 1, 4);
        __syncthreads();
        G6_STAMP(kb + 1, 5);
    }
    if (nk & 1) step(0, 1, rb1, rb0, 0, true);   // odd number of steps: the last one is staged in LDS stage 0
    WN_WAIT_VMCNT(0);
    if constexpr (F16) {   // an operand left fp16's range: inf pieces -> inf / NaN accumulators -> the six-product launch behind this one redoes it
        unsigned nonfinite = 0u;
        WN_UNROLL
        for (int i = 0; i < 4; ++i) {
            WN_UNROLL
            for (int j = 0; j < 2; ++j) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r)
                    nonfinite |= (~__builtin_bit_cast(unsigned, (float)acc[i][j][r]) & 0x7f800000u) == 0u ? 1u : 0u;
            }
        }
        if (nonfinite) wn_store_coherent_int(g.ovf, 1);
    }

    // epilogue: bias, mask, relu; rows of a lane are (r&3) + 8*(r>>2) + 4*hi, its column is li.
    // Buffer accesses with out-of-range offsets for the ragged edges (reads give 0, writes are dropped).
    const wn_rsrc_t Cr = wn_make_buf(g.C + (long)b * g.c_zstride, (unsigned)((long)g.M * g.ldc * 4));
    const wn_rsrc_t Er = wn_make_buf(g.E ? g.E + (long)b * g.e_zstride : g.C, g.E ? (unsigned)((long)g.M * g.lde * 4) : 0u);
    const wn_rsrc_t Dr = wn_make_buf(g.D ? g.D + (long)b * g.d_zstride : g.C, g.D ? (unsigned)((long)g.M * g.ldd * 4) : 0u);
    const wn_rsrc_t Biasr = wn_make_buf(g.bias ? g.bias : g.C, g.bias ? (unsigned)(g.M * 4) : 0u);
    if (g.gate_S != nullptr) {
        // Forward gate (wavenet.py:529-532) on the accumulators: row tiles i = 0, 1 of this wave are the sigmoid rows of
        // channels ch0 + 32 i + row, tiles 2, 3 the tanh rows of the same channels (wn_gemm6_gate_row).
        const int R = g.gate_R;
        const int ch0 = (m0 >> 1) + 64 * wm;
        const int F4 = g.gate_F * 4, T4 = (int)g.ldc * 4;
        const wn_rsrc_t Gr = wn_make_buf(g.gate_G + (long)b * g.gate_gb, (unsigned)(2 * R * F4));
        const wn_rsrc_t Sr = wn_make_buf(g.gate_S + (long)b * R * g.ldc, (unsigned)(R * T4));
        const wn_rsrc_t Tr = wn_make_buf(g.gate_Gt + (long)b * R * g.ldc, (unsigned)(R * T4));
        const wn_rsrc_t Zr = wn_make_buf(g.gate_Z + (long)b * R * g.ldc, (unsigned)(R * T4));
        WN_UNROLL
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + li;
            const bool ok = col < g.N;
            const int tc = ok ? col : g.N - 1;
            const int fr = tc / g.gate_U;
            const float wj = g.gate_upw[tc - fr * g.gate_U];
            WN_UNROLL
            for (int i = 0; i < 2; ++i) {
                const int cb = ch0 + 32 * i + 4 * hi;   // + mfma32_row(r, 0)
                float ga[16], gg[16];
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    ga[r] = wn_buf_load(Gr, (cb * g.gate_F + fr) * 4, mfma32_row(r, 0) * F4);
                    gg[r] = wn_buf_load(Gr, ((R + cb) * g.gate_F + fr) * 4, mfma32_row(r, 0) * F4);
                }
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int c = cb + mfma32_row(r, 0);
                    const float pa = acc[i][j][r] * fsc + (wj * ga[r] + g.gate_cvec[c]);
                    const float pg = acc[i + 2][j][r] * fsc + (wj * gg[r] + g.gate_cvec[R + c]);
                    const float sv = wn_sigmoid(pa), gv = wn_tanh(pg);
                    const int off = ok ? (cb * (int)g.ldc + col) * 4 : 0x7ffffff0;
                    wn_buf_store(Sr, sv, off, mfma32_row(r, 0) * T4);
                    wn_buf_store(Tr, gv, off, mfma32_row(r, 0) * T4);
                    wn_buf_store(Zr, sv * gv, off, mfma32_row(r, 0) * T4);
                }
            }
        }
        return;
    }
    if (g.gbw_dP != nullptr) {
        // Backward gate (wavenet.py:529-532 reversed): dZ = acc (+ the partial sum already in C) never leaves the chip
        const int R = g.M, T4 = (int)g.ldc * 4;
        const wn_rsrc_t Sr = wn_make_buf(g.gbw_S + (long)b * R * g.ldc, (unsigned)(R * T4));
        const wn_rsrc_t Tr = wn_make_buf(g.gbw_Gt + (long)b * R * g.ldc, (unsigned)(R * T4));
        const wn_rsrc_t Pr = wn_make_buf(g.gbw_dP + (long)b * 2 * R * g.ldc, (unsigned)(2 * R * T4));
        WN_UNROLL
        for (int i = 0; i < 4; ++i) {
            WN_UNROLL
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + 64 * wn + 32 * j + li;
                const int rb = m0 + 128 * wm + 32 * i + 4 * hi;
                float sv[16], gv[16], cv[16];
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row = rb + mfma32_row(r, 0);
                    const int off = (row < R && col < g.N) ? (row * (int)g.ldc + col) * 4 : 0x7ffffff0;
                    sv[r] = wn_buf_load(Sr, off, 0);
                    gv[r] = wn_buf_load(Tr, off, 0);
                    cv[r] = g.accumulate ? wn_buf_load(Cr, off, 0) : 0.f;
                }
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row = rb + mfma32_row(r, 0);
                    const int off = (row < R && col < g.N) ? (row * (int)g.ldc + col) * 4 : 0x7ffffff0;
                    const float dz = acc[i][j][r] * fsc + cv[r];
                    wn_buf_store(Pr, dz * gv[r] * (sv[r] * (1.0f - sv[r])), off, 0);
                    wn_buf_store(Pr, dz * sv[r] * (1.0f - gv[r] * gv[r]), off, R * T4);
                }
            }
        }
        return;
    }
    if (CE) {
        // Softmax cross-entropy on the accumulators (see WnGemm6Args).  A column's M classes are spread over the two lane
        // halves (rows 4 hi + ...) of the two waves wm = 0, 1 with the same wn: per-lane reduction over its 64 values, one
        // lane-half exchange, one exchange through LDS -- for the maximum, then for the sum of exponentials.  The logits stay
        // in the accumulator registers (overwritten by exp(logit - max)).
        float* red = reinterpret_cast<float*>(smem_raw);   // [max | sum][wave][64 columns], then [4] loss
        __syncthreads();                                   // every wave is done with the operand stages
        const float NEG = -3.0e38f;
        float mx[2], sm[2], vt[2];
        int tq[2];
        bool okc[2], live[2];
        WN_UNROLL
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + li;
            okc[j] = col < g.N;
            live[j] = okc[j] && col >= g.ce_t_start;
            // class index modulo M like k_softmax_ce; the 64-bit division only runs for a target outside [0, M)
            long long tg = okc[j] ? g.ce_target[(long)b * g.ce_tstride + col] : 0;
            if ((unsigned long long)tg >= (unsigned long long)g.M) {
                tg %= g.M;
                if (tg < 0) tg += g.M;
            }
            tq[j] = (int)tg;
            mx[j] = NEG;
        }
        WN_UNROLL
        for (int i = 0; i < 4; ++i) {
            float bv[16];
            WN_UNROLL
            for (int r = 0; r < 16; ++r)
                bv[r] = wn_buf_load(Biasr, (128 * wm + 32 * i + 4 * hi) * 4, mfma32_row(r, 0) * 4);   // no bias: empty descriptor, reads 0
            WN_UNROLL
            for (int j = 0; j < 2; ++j) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row = 128 * wm + 32 * i + mfma32_row(r, hi);
                    float v = acc[i][j][r] * fsc + bv[r];
                    v = row < g.M ? v : NEG;
                    acc[i][j][r] = v;
                    mx[j] = fmaxf(mx[j], v);
                }
            }
        }
        WN_UNROLL
        for (int j = 0; j < 2; ++j) {
            mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], 32, 64));
            red[wave * 64 + 32 * j + li] = mx[j];   // both lane halves write the same value
        }
        __syncthreads();
        WN_UNROLL
        for (int j = 0; j < 2; ++j) {
            const float m = fmaxf(mx[j], red[(wave ^ 2) * 64 + 32 * j + li]);
            mx[j] = m;
            float s = 0.f, t = 0.f;
            WN_UNROLL
            for (int i = 0; i < 4; ++i) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row = 128 * wm + 32 * i + mfma32_row(r, hi);
                    const float v = acc[i][j][r];
                    t = (row == tq[j]) ? v : t;   // the target's logit, in the one lane that holds its row
                    const float e = wn_exp2((v - m) * 1.4426950408889634f);
                    acc[i][j][r] = e;
                    s += e;
                }
            }
            vt[j] = t;
            s += __shfl_xor(s, 32, 64);
            sm[j] = s;
            red[256 + wave * 64 + 32 * j + li] = s;
        }
        __syncthreads();
        const wn_rsrc_t Dl = wn_make_buf(g.C ? g.C + (long)b * g.c_zstride : g.B, g.C ? (unsigned)((long)g.M * g.ldc * 4) : 0u);
        float my_loss = 0.f, my_amax = 0.f;
        WN_UNROLL
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + li;
            const float s = sm[j] + red[256 + (wave ^ 2) * 64 + 32 * j + li];
            const float lse = wn_log2(s) * 0.6931471805599453f + mx[j];
            const float scale = live[j] ? wn_rcp(s) * g.ce_gs : 0.f;
            const float hot = live[j] ? g.ce_gs : 0.f;
            const int tl = tq[j] - 128 * wm - 4 * hi;   // row == tq  <=>  32 i + mfma32_row(r, 0) == tl
            my_loss += (live[j] && tl >= 0 && tl < 128 && ((tl >> 2) & 1) == 0) ? (lse - vt[j]) : 0.f;
            const int vC = okc[j] ? ((128 * wm + 4 * hi) * (int)g.ldc + col) * 4 : 0x7ffffff0;
            WN_UNROLL
            for (int i = 0; i < 4; ++i) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int rl = 32 * i + mfma32_row(r, 0);
                    const float d = acc[i][j][r] * scale - (rl == tl ? hot : 0.f);
                    my_amax = fmaxf(my_amax, fabsf(d));               // (rows >= M hold exp(NEG - max) = 0)
                    wn_buf_store(Dl, d, vC, rl * (int)g.ldc * 4);   // rows >= M fall outside the descriptor: dropped
                }
            }
        }
        my_loss = wave_reduce_sum(my_loss);
        my_amax = wave_reduce_max(my_amax);
        __syncthreads();
        if (lane == 0) { red[wave] = my_loss; red[4 + wave] = my_amax; }
        __syncthreads();
        if (tid == 0) {
            g.ce_partial[(long)blk.z * gridDim.x + blk.x] = (red[0] + red[1]) + (red[2] + red[3]);
            if (g.ce_amax) g.ce_amax[(long)blk.z * gridDim.x + blk.x] = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        }
        return;
    }
    // Interior blocks (the whole 256 x 128 tile inside C; every block of the benchmark's launches): the row part of an
    // address is a wave-uniform scalar offset, the lane keeps ONE byte offset per tensor -- no per-element range selects.
    const bool interior = (m0 + WN_G6_BM <= g.M) && (n0 + WN_G6_BN <= g.N) && !g.no_interior &&
                          (long)g.M * g.ldc * 4 < 0x7fffffffL;
    if (interior) {
        const int rl = m0 + 128 * wm + 4 * hi, cl = n0 + 64 * wn + li;
        const int vC = (rl * (int)g.ldc + cl) * 4, vE = (rl * (int)g.lde + cl) * 4, vD = (rl * (int)g.ldd + cl) * 4;
        WN_UNROLL
        for (int i = 0; i < 4; ++i) {
            float bv[16];
            WN_UNROLL
            for (int r = 0; r < 16; ++r) bv[r] = 0.f;
            if (g.bias) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) bv[r] = wn_buf_load(Biasr, (rl + 32 * i) * 4, mfma32_row(r, 0) * 4);
            }
            WN_UNROLL
            for (int j = 0; j < 2; ++j) {
                float ev[16], dv[16];
                if (g.E) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r)
                        ev[r] = wn_buf_load(Er, vE + (32 * i * (int)g.lde + 32 * j) * 4, mfma32_row(r, 0) * (int)g.lde * 4);
                }
                WN_UNROLL
                for (int r = 0; r < 16; ++r) dv[r] = 0.f;
                if (g.D) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r)
                        dv[r] = wn_buf_load(Dr, vD + (32 * i * (int)g.ldd + 32 * j) * 4, mfma32_row(r, 0) * (int)g.ldd * 4);
                }
                float cv[16];   // C += result: the previous value of this lane's 16 elements (0 otherwise)
                WN_UNROLL
                for (int r = 0; r < 16; ++r) cv[r] = 0.f;
                if (g.accumulate) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r)
                        cv[r] = wn_buf_load(Cr, vC + (32 * i * (int)g.ldc + 32 * j) * 4, mfma32_row(r, 0) * (int)g.ldc * 4);
                }
                WN_SCHED_BARRIER();
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] * fsc;
                    v += bv[r] + dv[r];
                    if (g.relu) v = fmaxf(v, 0.f);
                    if (g.E) v = (ev[r] > 0.f) ? v : 0.f;
                    v += cv[r];
                    wn_buf_store(Cr, v, vC + (32 * i * (int)g.ldc + 32 * j) * 4, mfma32_row(r, 0) * (int)g.ldc * 4);
                }
            }
        }
        return;
    }
    WN_UNROLL
    for (int i = 0; i < 4; ++i) {
        float bv[16];  // bias of this lane's 16 rows (0 when absent: out-of-range reads)
        WN_UNROLL
        for (int r = 0; r < 16; ++r) bv[r] = wn_buf_load(Biasr, (m0 + 128 * wm + 32 * i + mfma32_row(r, hi)) * 4, 0);
        WN_UNROLL
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + li;
            float ev[16], dv[16];
            if (g.E) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 128 * wm + 32 * i + mfma32_row(r, hi);
                    ev[r] = wn_buf_load(Er, (row < g.M && col < g.N) ? (row * (int)g.lde + col) * 4 : 0x7ffffff0, 0);
                }
            }
            WN_UNROLL
            for (int r = 0; r < 16; ++r) dv[r] = 0.f;
            if (g.D) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 128 * wm + 32 * i + mfma32_row(r, hi);
                    dv[r] = wn_buf_load(Dr, (row < g.M && col < g.N) ? (row * (int)g.ldd + col) * 4 : 0x7ffffff0, 0);
                }
            }
            WN_SCHED_BARRIER();
            WN_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 128 * wm + 32 * i + mfma32_row(r, hi);
                float v = acc[i][j][r] * fsc;
                v += bv[r] + dv[r];
                if (g.relu) v = fmaxf(v, 0.f);
                if (g.E) v = (ev[r] > 0.f) ? v : 0.f;
                const int coff = (row < g.M && col < g.N) ? (row * (int)g.ldc + col) * 4 : 0x7ffffff0;
                if (g.accumulate) v += wn_buf_load(Cr, coff, 0);
                wn_buf_store(Cr, v, coff, 0);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_gemm6n -- the fp16 pair contraction for SHORT contractions (K <= 512: 16 - 32 k-steps per block) with MORE than 256 output
// rows: the all-layer skip gradient (M = 64 L).  At 256 x 128 the blocks of k_gemm6 spend two thirds of their life in their
// prologue (operand latency) and their 128 KB epilogue, two per CU: the matrix pipe idles while a block stores, the store path
// idles while it contracts (bwd_dz_skip_all: 0.3 ms of MFMAs at full rate, 0.28 ms of HBM time, 0.62 ms measured), and 1920 rows are
// 7.5 blocks of 256.  Here a block owns 128 x 128 (wave tile 64 x 64: 64 accumulator registers), 32 KB of LDS and <= 168 registers:
// THREE blocks per CU are in different phases at any time: 0.59 ms (profiles/r06/abk_gemm6_narrow.txt; the 256-row post-net launches
// measured 0.004 ms slower each on it and stay on k_gemm6).  Same
// operand layouts, same pre-split fp16 images (a block takes rows m0 .. m0 + 127 of the [kb][2][Mpad][16] image), same
// alternating tile signs, same overflow word; plain epilogue only (bias, residual D, relu, mask E): no gate / loss epilogues, no
// C += result.
// ---------------------------------------------------------------------------------------------
#define WN_G6N_BM 128
__global__ __launch_bounds__(G6_T, 3) void k_gemm6n(WnGemm6Args g, int order) {
    WN_DYN_SMEM(smem_raw);
    constexpr int NP = 2;
    constexpr int A_BYTES = NP * WN_G6N_BM * 32, B_BYTES = NP * WN_G6_BN * 32, ST_BYTES = A_BYTES + B_BYTES;   // 8 + 8 KB per stage
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const WnBlock blk = wn_block_order(order);
    const int b = WN_UNIFORM(blk.z);
    const int m0 = WN_UNIFORM(blk.y) * WN_G6N_BM, n0 = WN_UNIFORM(blk.x) * WN_G6_BN;
    const float* __restrict__ Bz = g.B + (long)b * g.b_zstride;
    const int nk = (g.K + 15) / 16;
    const bool one_seg = g.b_seg_len >= g.K;
    const float fs = ((blk.x + g.n_phase) & 1) ? -1.0f : 1.0f;
    const float b_mul = g.b_mul < 0.0f ? reinterpret_cast<const float*>(g.ovf)[1] : g.b_mul;
    const float fsm = fs * b_mul;
    const float fsc = fs / (b_mul * (float)WN_G6_F16_WSCALE);
    const wn_rsrc_t Ar = wn_make_buf(g.Apk, (unsigned)((long)nk * NP * g.Mpad * 32));
    const int wave_u = WN_UNIFORM(wave);
    float rb0[8], rb1[8];
    const int bn = tid & 127, bkh = tid >> 7;
    const bool n_ok = (n0 + bn) < g.N;
    const int a_voff = (tid >> 1) * 32 + (wn_frag_off(tid >> 1, tid & 1) & 16);
    const int bkh_u = WN_UNIFORM(bkh);
    const int seg_rows = one_seg ? nk * 16 : g.b_seg_len;
    int fb_k0 = 0, fb_seg = 0, fb_rr = 0;
    auto fetch_a = [&](int kb, int st) {   // 128 rows x 32 bytes per piece = 4 KB = one 16-byte slot per thread
        char* sa = smem_raw + st * ST_BYTES + wave_u * 1024;
        const int kc = kb < nk ? kb : nk - 1;
        WN_UNROLL
        for (int p = 0; p < NP; ++p)
            wn_buf_load_lds16(Ar, sa + p * (WN_G6N_BM * 32), a_voff, (unsigned)((kc * NP + p) * g.Mpad + m0) * 32u);
    };
    auto advance_b = [&]() {
        const bool more = fb_k0 + 16 < nk * 16;
        const bool wrap = fb_rr + 16 >= seg_rows;
        fb_k0 = more ? fb_k0 + 16 : fb_k0;
        fb_seg = (more && wrap) ? fb_seg + 1 : fb_seg;
        fb_rr = more ? (wrap ? 0 : fb_rr + 16) : fb_rr;
    };
    auto fetch_b = [&](float (&rb)[8]) {
        const wn_rsrc_t Br = wn_make_buf(Bz + (long)fb_seg * g.b_seg_stride, (unsigned)((long)(one_seg ? g.K : g.b_seg_len) * g.ldb * 4));
        const int cc = n0 + bn - (g.b_shift0 + fb_seg * g.b_shift_step);
        const int voff = (n_ok && cc >= 0 && cc < g.b_clen) ? cc * 4 : 0x7ffffff0;
        const int rlast = g.K - 1 - (fb_k0 - fb_rr);
        WN_UNROLL
        for (int e = 0; e < 8; ++e) {
            int r = fb_rr + 8 * bkh_u + e;
            r = r < rlast ? r : rlast;
            rb[e] = wn_buf_load(Br, voff, r * (int)g.ldb * 4);
        }
        advance_b();
    };
    auto write_pieces = [&](int st, const unsigned (&h)[4], const unsigned (&md)[4]) {
        char* sb = smem_raw + st * ST_BYTES + A_BYTES + wn_frag_off(bn, bkh);
        wn_f4 v;
        v.x = wn_bits_f32(h[0]); v.y = wn_bits_f32(h[1]); v.z = wn_bits_f32(h[2]); v.w = wn_bits_f32(h[3]);
        *reinterpret_cast<wn_f4*>(sb) = v;
        v.x = wn_bits_f32(md[0]); v.y = wn_bits_f32(md[1]); v.z = wn_bits_f32(md[2]); v.w = wn_bits_f32(md[3]);
        *reinterpret_cast<wn_f4*>(sb + WN_G6_BN * 32) = v;
    };
    auto split_pair = [&](int q, const float (&rb)[8], unsigned (&h)[4], unsigned (&md)[4]) {
        const float x0 = rb[2 * q] * fsm, x1 = rb[2 * q + 1] * fsm;
        h[q] = wn_pk_f16(x0, x1);
        md[q] = wn_pk_f16(x0 - wn_f16lo_f32(h[q]), x1 - wn_f16hi_f32(h[q]));
    };
    auto stage = [&](int st, const float (&rb)[8]) {
        unsigned h[4], md[4];
        WN_UNROLL
        for (int q = 0; q < 4; ++q) split_pair(q, rb, h, md);
        write_pieces(st, h, md);
    };
    f32x16 acc[2][2];
    WN_UNROLL
    for (int i = 0; i < 2; ++i) {
        acc[i][0] = f32x16_zero();
        acc[i][1] = f32x16_zero();
    }
    // One k-step: 12 MFMAs (2 row tiles x 3 products x 2 column tiles) in 6 pairs; behind pair s:
    //   0, 1: one LDS-DMA piece of the next weight slab + two activation loads each    2: the other four loads
    //   3, 4: two pair splits each        5: the two LDS writes of the split pieces
    auto step = [&](int st, int stn, const float (&rb)[8], float (&rbn)[8], int ka, bool last) {
        const char* sa = smem_raw + st * ST_BYTES;
        const char* sb = sa + A_BYTES;
        unsigned h[4], md[4];
        wn_f4 bf[NP][2];
        WN_UNROLL
        for (int p = 0; p < NP; ++p) {
            WN_UNROLL
            for (int j = 0; j < 2; ++j)
                bf[p][j] = *reinterpret_cast<const wn_f4*>(sb + p * (WN_G6_BN * 32) + wn_frag_off(64 * wn + 32 * j + li, hi));
        }
        const wn_rsrc_t Br = wn_make_buf(Bz + (long)fb_seg * g.b_seg_stride, (unsigned)((long)(one_seg ? g.K : g.b_seg_len) * g.ldb * 4));
        const int cc = n0 + bn - (g.b_shift0 + fb_seg * g.b_shift_step);
        const int voff = (n_ok && cc >= 0 && cc < g.b_clen) ? cc * 4 : 0x7ffffff0;
        const int rlast = g.K - 1 - (fb_k0 - fb_rr);
        char* sda = smem_raw + stn * ST_BYTES + wave_u * 1024;
        const int kc = ka < nk ? ka : nk - 1;
        auto load_b = [&](int e) {
            int r = fb_rr + 8 * bkh_u + e;
            r = r < rlast ? r : rlast;
            rbn[e] = wn_buf_load(Br, voff, r * (int)g.ldb * 4);
        };
        WN_UNROLL
        for (int i = 0; i < 2; ++i) {
            wn_f4 af[NP];
            WN_UNROLL
            for (int p = 0; p < NP; ++p)
                af[p] = *reinterpret_cast<const wn_f4*>(sa + p * (WN_G6N_BM * 32) + wn_frag_off(64 * wm + 32 * i + li, hi));
            constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};   // small terms first: h l, l h, h h
            WN_UNROLL
            for (int t = 0; t < 3; ++t) {
                WN_UNROLL
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f16(af[PA[t]], bf[PB[t]][j], acc[i][j]);
                if (last) continue;
                const int sl = i * 3 + t;
                if (sl == 0) {   // the slab BEFORE the activation loads: "at most 8 loads outstanding" == "the slab has landed"
                    wn_buf_load_lds16(Ar, sda, a_voff, (unsigned)((kc * NP) * g.Mpad + m0) * 32u);
                    wn_buf_load_lds16(Ar, sda + WN_G6N_BM * 32, a_voff, (unsigned)((kc * NP + 1) * g.Mpad + m0) * 32u);
                } else if (sl < 3) {
                    load_b(4 * sl - 4); load_b(4 * sl - 3); load_b(4 * sl - 2); load_b(4 * sl - 1);
                } else if (sl < 5) {
                    split_pair(2 * (sl - 3), rb, h, md);
                    split_pair(2 * (sl - 3) + 1, rb, h, md);
                } else {
                    write_pieces(stn, h, md);
                }
                WN_SGB_MFMA(1);
                WN_SGB_VALU(8);
                WN_SGB_MFMA(1);
                WN_SCHED_FENCE_ALU();
            }
        }
        if (!last) advance_b();
    };
    fetch_a(0, 0);
    fetch_b(rb0);
    fetch_b(rb1);
    stage(0, rb0);
    WN_WAIT_VMCNT(8);
    __syncthreads();
    const int npair = nk >> 1;
    for (int kp = 0; kp < npair; ++kp) {
        const int kb = 2 * kp;
        step(0, 1, rb1, rb0, kb + 1, false);
        WN_WAIT_VMCNT(8);
        __syncthreads();
        step(1, 0, rb0, rb1, kb + 2, false);
        WN_WAIT_VMCNT(8);
        __syncthreads();
    }
    if (nk & 1) step(0, 1, rb1, rb0, 0, true);
    WN_WAIT_VMCNT(0);
    {   // an operand left fp16's range -> the six-product launch behind this one redoes the contraction
        unsigned nonfinite = 0u;
        WN_UNROLL
        for (int i = 0; i < 2; ++i) {
            WN_UNROLL
            for (int j = 0; j < 2; ++j) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r)
                    nonfinite |= (~__builtin_bit_cast(unsigned, (float)acc[i][j][r]) & 0x7f800000u) == 0u ? 1u : 0u;
            }
        }
        if (nonfinite) wn_store_coherent_int(g.ovf, 1);
    }
    // epilogue: bias, residual, relu, mask.  A lane's rows are m0 + 64 wm + 32 i + (r & 3) + 8 (r >> 2) + 4 hi (all inside M: whole
    // row blocks only), its columns n0 + 64 wn + 32 j + li: the row part of an address is a wave-uniform scalar offset, a column
    // past N carries an out-of-range lane offset (loads give 0, stores are dropped)
    const wn_rsrc_t Cr = wn_make_buf(g.C + (long)b * g.c_zstride, (unsigned)((long)g.M * g.ldc * 4));
    const wn_rsrc_t Er = wn_make_buf(g.E ? g.E + (long)b * g.e_zstride : g.C, g.E ? (unsigned)((long)g.M * g.lde * 4) : 0u);
    const wn_rsrc_t Dr = wn_make_buf(g.D ? g.D + (long)b * g.d_zstride : g.C, g.D ? (unsigned)((long)g.M * g.ldd * 4) : 0u);
    const wn_rsrc_t Biasr = wn_make_buf(g.bias ? g.bias : g.C, g.bias ? (unsigned)(g.M * 4) : 0u);
    const int rl = m0 + 64 * wm + 4 * hi;
    WN_UNROLL
    for (int i = 0; i < 2; ++i) {
        float bv[16];
        WN_UNROLL
        for (int r = 0; r < 16; ++r) bv[r] = 0.f;
        if (g.bias) {
            WN_UNROLL
            for (int r = 0; r < 16; ++r) bv[r] = wn_buf_load(Biasr, (rl + 32 * i) * 4, mfma32_row(r, 0) * 4);
        }
        WN_UNROLL
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + li;
            const bool cok = col < g.N;
            const int vC = cok ? ((rl + 32 * i) * (int)g.ldc + col) * 4 : 0x7ffffff0;
            const int vE = cok ? ((rl + 32 * i) * (int)g.lde + col) * 4 : 0x7ffffff0;
            const int vD = cok ? ((rl + 32 * i) * (int)g.ldd + col) * 4 : 0x7ffffff0;
            float ev[16], dv[16];
            if (g.E) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) ev[r] = wn_buf_load(Er, vE, mfma32_row(r, 0) * (int)g.lde * 4);
            }
            WN_UNROLL
            for (int r = 0; r < 16; ++r) dv[r] = 0.f;
            if (g.D) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) dv[r] = wn_buf_load(Dr, vD, mfma32_row(r, 0) * (int)g.ldd * 4);
            }
            WN_SCHED_BARRIER();
            WN_UNROLL
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] * fsc;
                v += bv[r] + dv[r];
                if (g.relu) v = fmaxf(v, 0.f);
                if (g.E) v = (ev[r] > 0.f) ? v : 0.f;
                wn_buf_store(Cr, v, vC, mfma32_row(r, 0) * (int)g.ldc * 4);
            }
        }
    }
}

// 1: this launch takes k_gemm6n (fp16 pair, short contraction, plain epilogue, whole 128-row blocks)
static bool gemm6n_applies(const WnGemm6Args& g) {
    static int on = -1;   // WN_G6_NARROW=0: A/B knob
    if (on < 0) { const char* e = getenv("WN_G6_NARROW"); on = e ? atoi(e) : 1; }
    return on && g.f16 && g.K <= 512 && g.M > WN_G6_BM && (g.M % WN_G6N_BM) == 0 && !g.ce_target && !g.gate_S && !g.gbw_dP && !g.accumulate && !g.no_interior &&
           (long)g.M * g.ldc * 4 < 0x7ffffff0L && (!g.E || (long)g.M * g.lde * 4 < 0x7ffffff0L) && (!g.D || (long)g.M * g.ldd * 4 < 0x7ffffff0L);
}

int wn_gemm6_launch(const WnGemm6Args* gp, wn_stream_t st) {
    WnGemm6Args g = *gp;
    g.no_interior = 0;
    {   // head start of a CU's first block over its co-resident: 50 % of a block's k-loop, see k_gemm6.  Only for short contractions
        // (K <= 512: the post-net and the all-layer skip gradient, 16 steps per block), where a block's prologue and epilogue
        // are a third of its life; measured on MI355X (profiles/r02/ab_probe_loss_window_stagger.txt): bwd_post{1,2}_dx
        // 0.172 -> 0.147 ms each, while the long skip-sum (120 steps per block) only pays for the late start (+0.04 ms).
        static int pct_env = -1;   // WN_G6_STAGGER_PCT: A/B knob
        if (pct_env < 0) { const char* e = getenv("WN_G6_STAGGER_PCT"); pct_env = e ? atoi(e) : 50; }
        const int pct = pct_env;
        // ~4000 cycles per 16-k step with two blocks per CU (six products; ~2000 with the three of the fp16 pair split); one
        // s_sleep(127) = 8128 cycles
        g.stagger = g.K <= 512 ? (int)(((long)pct * ((g.K + 15) / 16) * (g.f16 ? 2000L : 4000L)) / (100L * 8128L)) : 0;
    }
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.nbatch <= 0) return 1;
    if (g.b_seg_len < g.K && (g.b_seg_len % 16) != 0) return 2;
    if (g.ce_target && (g.Mpad != WN_G6_BM || !g.ce_partial || g.gate_S || g.gbw_dP || g.E || g.D || g.accumulate || g.relu)) return 4;
    constexpr int lds = 2 * (3 * WN_G6_BM * 32 + 3 * WN_G6_BN * 32);
    constexpr int lds16 = 2 * (2 * WN_G6_BM * 32 + 2 * WN_G6_BN * 32);
    if (g.f16 && (!g.ovf || g.b_mul == 0.0f)) return 5;
#ifndef WN_EMU
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm6<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
                hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm6<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
                hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm6<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds16) !=
                hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm6<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds16) !=
                hipSuccess)
            return 3;
        attr_set = true;
    }
#endif
    const bool redo = !g.f16 && g.ovf != nullptr;   // the conditional redo: no work unless an fp16 launch overflowed
    WN_PROF(redo ? "mm_redo_if_overflow" : (g.tag ? g.tag : "gemm6"), redo ? 0.0 : 2.0 * g.M * g.N * (double)g.K * g.nbatch,
            redo ? 0.0 : ((double)g.M * g.K * (g.f16 ? 4.0 : 6.0) + (double)g.K * g.N * 4.0 + (double)g.M * g.N * (g.E ? 8.0 : 4.0)) * g.nbatch, st);
    dim3 grid((unsigned)((g.N + WN_G6_BN - 1) / WN_G6_BN), (unsigned)(g.Mpad / WN_G6_BM), (unsigned)g.nbatch);
    if (g.f16 && gemm6n_applies(g)) {
        constexpr int ldsn = 2 * (2 * WN_G6N_BM * 32 + 2 * WN_G6_BN * 32);
        dim3 gridn((unsigned)((g.N + WN_G6_BN - 1) / WN_G6_BN), (unsigned)(g.M / WN_G6N_BM), (unsigned)g.nbatch);
        WN_LAUNCH(k_gemm6n, gridn, dim3(G6_T), ldsn, st, g, xcd_block_order() ? 2 : 0);
    } else if (g.f16) {
        if (g.ce_target)
            WN_LAUNCH((k_gemm6<true, true>), grid, dim3(G6_T), lds16, st, g, xcd_block_order() ? 2 : 0 G6_DBG_ARG(g.tag));
        else
            WN_LAUNCH((k_gemm6<false, true>), grid, dim3(G6_T), lds16, st, g, xcd_block_order() ? 2 : 0 G6_DBG_ARG(g.tag));
    } else if (g.ce_target)
        WN_LAUNCH(k_gemm6<true>, grid, dim3(G6_T), lds, st, g, xcd_block_order() ? 2 : 0 G6_DBG_ARG(g.tag));
    else
        WN_LAUNCH(k_gemm6<false>, grid, dim3(G6_T), lds, st, g, xcd_block_order() ? 2 : 0 G6_DBG_ARG(g.tag));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// weight-gradient type: C[z][m][n] = sum_k A(m,k) B(n, k - shift_n), k = time (contiguous in both)
// ---------------------------------------------------------------------------------------------
// Block tile (64 TM) x (64 TN), 16 k per step, 4 waves (2 x 2).  A thread owns 8 (or 4) consecutive
// k of one operand row: one or two 16-byte global loads (the shifted taps only need 4-byte
// alignment), split into the three bf16 pieces in registers, one 16 (8) byte LDS write per piece
// into the fragment layout [piece][row][16 k].  The loads run two steps ahead of the MFMAs (two
// register sets), LDS is double buffered.
// NP = bf16 pieces per operand: 3 (x = h + m + l, the six products above 2^-24 |ab|: fp32-equivalent) or 2 (x ~ h + m, the three
// products h h + h m + m h: relative error ~2^-16 per product, random in sign).  NP = 2 is for LEAF results only -- a weight
// gradient is a sum over every position of the minibatch that feeds nothing downstream --, opt-in (WN_FLAG_DW_3PRODUCT),
// never for a contraction whose output another layer consumes.
// TM x TN = 4 x 4 (round 5): a 256 x 256 block tile, ONE wave per SIMD (its 128 x 128 wave tile is 256 accumulator registers: the
// unified 512-register file of a lone wave), for the square weight gradients of wide models (n_resch 512: 1024 x 1024 and
// 512 x 512 outputs) -- every operand row is read once per 256 rows of the other operand (the 256 x 128 tile moved 50.8 GB per
// launch for 22.6 GB of operands, profiles/r05/pmc_traffic_recipe.json) and a wave reads 1.5 x fewer fragment bytes per MFMA.
// F16 (round 5, NP = 2; WN_FLAG_DW_F16PAIR): the two pieces are fp16 -- 11 + 11 significand bits, the three products leave
// ~2^-22 |a b| (64 x less than two bf16 pieces, less than the rounding of an fp32 running sum over the minibatch's positions).
// fp16 has 5 exponent bits where bf16 has fp32's 8: the A operand (a gradient: <= 2^-e by the caller's word, ~1e-5 at the
// benchmark's size) is multiplied by a_mul = 2^(e + WN_DW_F16_HEADROOM) at the split and the result by 1 / a_mul; B (an
// activation, O(1)) is taken as it is.  A value beyond fp16's range becomes inf, its second piece -inf or NaN, the block's result
// non-finite: the epilogue then raises *ovf and the caller's conditional six-product launch (below) redoes the contraction.
template <bool F16>
static __device__ __forceinline__ f32x16 mfma_piece(wn_f4 a, wn_f4 b, f32x16 c) {
    if constexpr (F16) return mfma_f16(a, b, c);
    else return mfma_bf16(a, b, c);
}
template <int TM, int TN, int NP, bool F16 = false>
__global__ __launch_bounds__(G6_T, (TM * TN > 8 ? 1 : (TM * TN > 4 ? 2 : 3))) void k_gemm6_dw(WnGemmArgs g, int order, float a_mul, int* ovf G6_DBG_PARAM) {
    static_assert(NP == 2 || NP == 3, "two or three pieces per operand");
    static_assert(!F16 || NP == 2, "the fp16 split has two pieces");
    constexpr int NPROD = NP == 3 ? 6 : 3;
    // the six-product launch behind an fp16-pair launch: runs only if that one overflowed (ovf is never written by this kernel)
    if (!F16 && ovf != nullptr && wn_load_coherent_int(ovf) == 0) return;
    // the scale decided on the device (wn_dw_prepare: the measured max |dlogits| of this backward call), one uniform load
    if (F16 && a_mul < 0.0f) a_mul = reinterpret_cast<const float*>(ovf)[1];
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int AE = BM / 16, BE = BN / 16;            // fp32 elements per thread and step
    // 192-column tiles (TN = 3: kernel_size 3 at 64 channels, N = 3 x 64) give a thread three B rows of 4 consecutive k each
    // (rows r, r + 64, r + 128) instead of one row of BE
    // Round 5: a thread takes 4 consecutive k of TM A rows (rows r, r + 64, ...) and of TN B rows instead of AE (BE) consecutive k of
    // ONE row: 4 lanes then cover 64 contiguous bytes of a row and a wave's load instruction 16 rows x 64 B -- with one row per lane
    // (TM = 4) or two lanes per row (TM = 2) every 16-byte load was its own cache line (64 / 32 lines per instruction: the vector L1
    // serialises them, and with 512 rows x 128 B per step the 32 KB L1 lost each line before its second half was used).  Same
    // box: recipe size dw_dilated 20.0 -> 17.2 ms (256 x 256 tile), 26.9 -> 17.5 (256 x 128); headline 9.18 -> 9.00 ms per step with
    // the 128-row tiles (profiles/r05/dw3_probe_*.txt, abk_headline_coalesced.txt).
    constexpr int AR = TM, AEr = AE / AR;   // = 4
    constexpr int BR = TN, BEr = BE / BR;   // = 4
    static_assert(AEr == 4 && BEr == 4, "4 consecutive k per row and thread");
    constexpr int A_BYTES = NP * BM * 32, B_BYTES = NP * BN * 32, ST_BYTES = A_BYTES + B_BYTES;
    WN_DYN_SMEM(smem_raw);
    __shared__ long b_rowoff[BN];
    __shared__ int b_rowshift[BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const WnBlock blk = wn_block_order(order);
    const int z = WN_UNIFORM(blk.z);
#ifdef WN_TIMING
    const bool dbg_blk = (blk.x | blk.y | blk.z) == 0;
#endif
    const int zl = z / (g.nbatch * g.ksplit);
    const int zr = z - zl * (g.nbatch * g.ksplit);
    const int b = zr / g.ksplit;
    const int ks = zr - b * g.ksplit;
    const int dmul = (g.b_dil_depth > 0) ? (1 << ((g.b_layer0 + zl) % g.b_dil_depth)) : 1;
    const int sh0 = g.b_shift0 * dmul, shstep = g.b_shift_step * dmul;
    const int kbeg = ks * g.kchunk;
    const int kend = (g.K - kbeg > g.kchunk) ? (kbeg + g.kchunk) : g.K;
    const int bx = WN_UNIFORM(blk.x);
    const int m0 = WN_UNIFORM(blk.y) * BM, n0 = bx * BN;
    const float* __restrict__ Az = g.A + (long)zl * g.a_lstride + (long)b * g.a_zstride;
    const float* __restrict__ Bz = g.B + (long)zl * g.b_lstride + (long)b * g.b_zstride;
    const bool one_seg = g.b_seg_len >= g.N;

    for (int i = tid; i < BN; i += G6_T) {
        const int n = n0 + i;
        int seg = 0, rr = n;
        if (!one_seg) {
            seg = n / g.b_seg_len;
            rr = n - seg * g.b_seg_len;
        }
        b_rowoff[i] = (n < g.N) ? ((long)seg * g.b_seg_stride + (long)rr * g.ldb) : -1;
        b_rowshift[i] = sh0 + seg * shstep;
    }
    int b_shmin = sh0, b_shmax = sh0;
    if (!one_seg) {
        const int s_lo = sh0 + (n0 / g.b_seg_len) * shstep, s_hi = sh0 + ((n0 + BN - 1) / g.b_seg_len) * shstep;
        b_shmin = s_lo < s_hi ? s_lo : s_hi;
        b_shmax = s_lo < s_hi ? s_hi : s_lo;
    }
    __syncthreads();

    // this thread's slice of the operand tiles
    const int a_row = tid / (16 / AEr), a_k = (tid % (16 / AEr)) * AEr;   // (AR > 1: the first of the thread's rows a_row + 64 u)
    const int b_row = tid / (16 / BEr), b_k = (tid % (16 / BEr)) * BEr;   // (TN = 3: the first of the thread's three rows)
    long a_off[AR];
    bool a_row_ok[AR];
    WN_UNROLL
    for (int u = 0; u < AR; ++u) {
        a_off[u] = (long)(m0 + a_row + 64 * u) * g.lda;
        a_row_ok[u] = (m0 + a_row + 64 * u) < g.M;
    }
    long b_off[BR];
    int b_sh[BR];
    WN_UNROLL
    for (int u = 0; u < BR; ++u) {
        b_off[u] = b_rowoff[b_row + 64 * u];
        b_sh[u] = b_rowshift[b_row + 64 * u];
    }
    const bool a_tile_ok = (m0 + BM) <= g.M, b_tile_ok = (n0 + BN) <= g.N;
    float rowsum[AR];
    WN_UNROLL
    for (int u = 0; u < AR; ++u) rowsum[u] = 0.f;
    // (b_relu is not taken here: wn_gemm6_dw_eligible -- no caller of the weight-gradient path uses it, and a floor applied to every
    // B element cost 2 VALU instructions per element in the k-step whether it was on or not)
    // the bias row sums of A are written by the blocks of column tile 0 only: the others do not add them up either
    const bool do_rowsum = g.a_rowsum != nullptr && bx == 0;
    // Odd (batch, k-chunk) partials contract -A and are stored negated: the matrix core's truncation bias (see k_gemm6)
    // changes sign with the operand, so it cancels in the fixed-order sum of the partials instead of adding up over time.
    // (The bias row sums are fp32 VALU sums of the un-negated values.)
    const unsigned a_sign = WN_UNIFORM((zr & 1) ? 0x80008000u : 0u);   // applied to the three bf16 pieces of A at their LDS write

    auto fetch = [&](int k0, float (&ra)[AE], float (&rb)[BE]) {
        const bool full = (k0 + 16) <= kend;
        if (a_tile_ok && full) {
            WN_UNROLL
            for (int q = 0; q < AE / 4; ++q) {
                const int u = q / (AEr / 4), qq = q % (AEr / 4);
                const wn_f4 v = wn_ld4_unaligned(Az + a_off[u] + k0 + a_k + 4 * qq);
                ra[4 * q] = v.x; ra[4 * q + 1] = v.y; ra[4 * q + 2] = v.z; ra[4 * q + 3] = v.w;
            }
        } else {
            WN_UNROLL
            for (int e = 0; e < AE; ++e) {
                const int u = e / AEr;
                const int k = k0 + a_k + e % AEr;
                ra[e] = (a_row_ok[u] && k < kend) ? Az[a_off[u] + k] : 0.f;
            }
        }
        if (b_tile_ok && full && (k0 - b_shmax) >= 0 && (k0 + 16 - b_shmin) <= g.b_clen) {
            WN_UNROLL
            for (int q = 0; q < BE / 4; ++q) {
                const int u = q / (BEr / 4), qq = q % (BEr / 4);
                const wn_f4 v = wn_ld4_unaligned(Bz + b_off[u] + (k0 + b_k + 4 * qq - b_sh[u]));
                rb[4 * q] = v.x; rb[4 * q + 1] = v.y; rb[4 * q + 2] = v.z; rb[4 * q + 3] = v.w;
            }
        } else {
            WN_UNROLL
            for (int e = 0; e < BE; ++e) {
                const int u = e / BEr;
                const int k = k0 + b_k + e % BEr, cc = k - b_sh[u];
                rb[e] = (b_off[u] >= 0 && k < kend && cc >= 0 && cc < g.b_clen) ? Bz[b_off[u] + cc] : 0.f;
            }
        }
    };
    // split E consecutive-k values and write the three pieces of row `row` at k offset `kofs`
    // one pair of consecutive-k values -> its pieces (packed lo | hi << 16)
    auto split_pair = [&](float x0, float x1, bool is_a, unsigned& h, unsigned& md, unsigned& lo) {
        if constexpr (F16) {
            if (is_a) {
                x0 *= a_mul;
                x1 *= a_mul;
            }
            h = wn_pk_f16(x0, x1);
            md = wn_pk_f16(x0 - wn_f16lo_f32(h), x1 - wn_f16hi_f32(h));
            lo = 0u;
        } else {
            h = wn_pk_bf16(x0, x1);
            const float r0 = x0 - wn_bits_f32(h << 16), r1 = x1 - wn_bits_f32(h & 0xffff0000u);
            md = wn_pk_bf16(r0, r1);
            lo = NP == 3 ? wn_pk_bf16(r0 - wn_bits_f32(md << 16), r1 - wn_bits_f32(md & 0xffff0000u)) : 0u;
        }
    };
    auto split_store = [&](char* base, int rows, int row, int kofs, const float* v, int E, bool is_a, unsigned sign = 0u) {
        unsigned h[8], md[8], lo[8];  // E <= 16
        for (int q = 0; q < E / 2; ++q) split_pair(v[2 * q], v[2 * q + 1], is_a, h[q], md[q], lo[q]);
        const unsigned* src[3] = {h, md, lo};
        for (int p = 0; p < NP; ++p) {
            char* d = base + p * rows * 32;
            for (int q = 0; q < E / 2; ++q) {   // dword kq of the row: k half kq >> 2 (placement: wn_frag_off), dword kq & 3 of it
                const int kq = (kofs >> 1) + q;
                *reinterpret_cast<unsigned*>(d + wn_frag_off(row, kq >> 2) + (kq & 3) * 4) = src[p][q] ^ sign;
            }
        }
    };
    auto stage = [&](int st, const float (&ra)[AE], const float (&rb)[BE], bool counted = true) {
        char* sa = smem_raw + st * ST_BYTES;
        if (do_rowsum) {
            WN_UNROLL
            for (int u = 0; u < AR; ++u) {
                float rs = 0.f;
                WN_UNROLL
                for (int e = 0; e < AEr; ++e) rs += ra[u * AEr + e];
                rowsum[u] += counted ? rs : 0.f;
            }
        }
        WN_UNROLL
        for (int u = 0; u < AR; ++u) split_store(sa, BM, a_row + 64 * u, a_k, ra + u * AEr, AEr, true, a_sign);
        WN_UNROLL
        for (int u = 0; u < BR; ++u) split_store(sa + A_BYTES, BN, b_row + 64 * u, b_k, rb + u * BEr, BEr, false);
    };

    f32x16 acc[TM][TN];
    WN_UNROLL
    for (int i = 0; i < TM; ++i) {
        WN_UNROLL
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x16_zero();
    }
    auto compute = [&](int st) {
        const char* sa = smem_raw + st * ST_BYTES;
        const char* sb = sa + A_BYTES;
        wn_f4 bf[NP][TN];
        WN_UNROLL
        for (int p = 0; p < NP; ++p) {
            WN_UNROLL
            for (int j = 0; j < TN; ++j)
                bf[p][j] = *reinterpret_cast<const wn_f4*>(sb + p * (BN * 32) + wn_frag_off((wn * TN + j) * 32 + li, hi));
        }
        WN_UNROLL
        for (int i = 0; i < TM; ++i) {
            wn_f4 af[NP];
            WN_UNROLL
            for (int p = 0; p < NP; ++p)
                af[p] = *reinterpret_cast<const wn_f4*>(sa + p * (BM * 32) + wn_frag_off((wm * TM + i) * 32 + li, hi));
            // small terms first; NP = 2: h m, m h, h h
            constexpr int PA[6] = {0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0};
            constexpr int PB[6] = {NP == 3 ? 2 : 1, 0, NP == 3 ? 1 : 0, 1, 0, 0};
            WN_UNROLL
            for (int t = 0; t < NPROD; ++t) {
                WN_UNROLL
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma_piece<F16>(af[PA[t]], bf[PB[t]][j], acc[i][j]);
            }
        }
    };

    // Branch-free loads of an INTERIOR step (whole tile inside M x N, 16 k inside this k-chunk, every shifted tap inside its
    // sequence): two or four 16-byte loads per thread, nothing conditional -- so the interior pass below has no control
    // flow between a load and its use and the compiler keeps counted vmcnt waits.  (With the conditional fetch of the
    // general pass inside the pipelined loop it merged the two register sets through copies at the loop header and put
    // an `s_waitcnt vmcnt(0)` in front of every iteration: the loads were never in flight under the MFMAs.)
    auto fetch_fast = [&](int k0, float (&ra)[AE], float (&rb)[BE]) {
        WN_UNROLL
        for (int q = 0; q < AE / 4; ++q) {
            const int u = q / (AEr / 4), qq = q % (AEr / 4);
            const wn_f4 v = wn_ld4_unaligned(Az + a_off[u] + k0 + a_k + 4 * qq);
            ra[4 * q] = v.x; ra[4 * q + 1] = v.y; ra[4 * q + 2] = v.z; ra[4 * q + 3] = v.w;
        }
        WN_UNROLL
        for (int q = 0; q < BE / 4; ++q) {
            const int u = q / (BEr / 4), qq = q % (BEr / 4);
            const wn_f4 v = wn_ld4_unaligned(Bz + b_off[u] + (k0 + b_k + 4 * qq - b_sh[u]));
            rb[4 * q] = v.x; rb[4 * q + 1] = v.y; rb[4 * q + 2] = v.z; rb[4 * q + 3] = v.w;
        }
    };
    // One pipelined pass over the steps [k_lo, k_lo + 16 n): loads two steps ahead of the MFMAs in two register sets, LDS
    // double buffered; the accumulators carry over between passes.  FAST: n is even and every step is interior; the
    // loads past the last step re-read the last step (harmless, never staged) so that nothing in the loop is conditional.
    auto pass_general = [&](int k_lo, int n) {
        float ra0[AE], rb0[BE], ra1[AE], rb1[BE];
        if (n > 0) fetch(k_lo, ra0, rb0);
        if (n > 1) fetch(k_lo + 16, ra1, rb1);
        if (n > 0) stage(0, ra0, rb0);
        __syncthreads();
        for (int kb = 0; kb < n; kb += 2) {
            // even step: registers set 0 is free, set 1 holds step kb+1
            if (kb + 2 < n) fetch(k_lo + (kb + 2) * 16, ra0, rb0);
            WN_SCHED_BARRIER();
            compute(0);
            WN_SCHED_BARRIER();
            if (kb + 1 < n) stage(1, ra1, rb1);
            __syncthreads();
            if (kb + 1 >= n) break;
            // odd step
            if (kb + 3 < n) fetch(k_lo + (kb + 3) * 16, ra1, rb1);
            WN_SCHED_BARRIER();
            compute(1);
            WN_SCHED_BARRIER();
            if (kb + 2 < n) stage(0, ra0, rb0);
            __syncthreads();
        }
    };
    // One interior k-step: the TM x TN x 6 MFMAs on LDS stage `st` with everything else of the step in their shadow (see
    // k_gemm6: a wave's own independent VALU / memory instructions issue between its MFMAs for free, as separate phases they
    // cost the wave more cycles than the MFMAs themselves).  After every group of TN MFMAs comes one slice of the rest:
    // first the loads of the step after next, then the split of the next step's operands pair by pair (A, then B), each
    // operand's three LDS writes right after its last pair.  The fences pin this order for VALU, MFMA and memory
    // instructions; LDS reads (the next row tile's fragments) and scalar instructions may cross them.
    auto step_fast = [&](int st, int stn, const float (&ra)[AE], const float (&rb)[BE], float (&ran)[AE], float (&rbn)[BE],
                         int k_next, bool counted) {
        const char* sa = smem_raw + st * ST_BYTES;
        const char* sb = sa + A_BYTES;
        char* da = smem_raw + stn * ST_BYTES;
        constexpr int NPA = AE / 2, NPB = BE / 2, NSL = TM * NPROD;         // pairs of A, of B; slices
        constexpr int PPS = (NPA + NPB + NSL - 2) / (NSL - 1);              // pairs per slice (slice 0 is the loads)
        unsigned ha[NPA], ma[NPA], la[NPA], hb[NPB], mb[NPB], lb[NPB];
        auto put = [&](char* base, int rows, int row, int kofs, const unsigned* h, const unsigned* md, const unsigned* lo, int np,
                       unsigned sign) {
#ifdef WN_DWX_NOPUT
            return;
#endif
            const unsigned* src[3] = {h, md, lo};
            for (int p = 0; p < NP; ++p) {
                char* d = base + p * rows * 32;
                for (int q = 0; q < np; ++q) {
                    const int kq = (kofs >> 1) + q;
                    *reinterpret_cast<unsigned*>(d + wn_frag_off(row, kq >> 2) + (kq & 3) * 4) = src[p][q] ^ sign;
                }
            }
        };
        auto slice = [&](int sl) {
            if (sl == 0) {
#ifndef WN_DWX_NOLOAD   // (timing experiments only: tools/dw_timing.py)
                fetch_fast(k_next, ran, rbn);
#endif
                return;
            }
#ifdef WN_DWX_NOSPLIT
            return;
#endif
            WN_UNROLL
            for (int u = 0; u < PPS; ++u) {
                const int q = (sl - 1) * PPS + u;
                if (q < NPA) {
                    constexpr int PRA = AEr / 2;   // pairs per A row of this thread
                    split_pair(ra[2 * q], ra[2 * q + 1], true, ha[q], ma[q], la[q]);
                    if (do_rowsum) rowsum[q / PRA] += counted ? ra[2 * q] + ra[2 * q + 1] : 0.f;
                    if ((q + 1) % PRA == 0) {
                        const int u = q / PRA;
                        put(da, BM, a_row + 64 * u, a_k, ha + u * PRA, ma + u * PRA, la + u * PRA, PRA, a_sign);
                    }
                } else if (q < NPA + NPB) {
                    const int qb = q - NPA;
                    split_pair(rb[2 * qb], rb[2 * qb + 1], false, hb[qb], mb[qb], lb[qb]);
                    constexpr int PR = BEr / 2;   // pairs per B row of this thread
                    if ((qb + 1) % PR == 0) {
                        const int u = qb / PR;
                        put(da + A_BYTES, BN, b_row + 64 * u, b_k, hb + u * PR, mb + u * PR, lb + u * PR, PR, 0u);
                    }
                }
            }
        };
        wn_f4 bf[NP][TN];
        WN_UNROLL
        for (int p = 0; p < NP; ++p) {
            WN_UNROLL
            for (int j = 0; j < TN; ++j)
                bf[p][j] = *reinterpret_cast<const wn_f4*>(sb + p * (BN * 32) + wn_frag_off((wn * TN + j) * 32 + li, hi));
        }
        WN_UNROLL
        for (int i = 0; i < TM; ++i) {
            wn_f4 af[NP];
            WN_UNROLL
            for (int p = 0; p < NP; ++p)
                af[p] = *reinterpret_cast<const wn_f4*>(sa + p * (BM * 32) + wn_frag_off((wm * TM + i) * 32 + li, hi));
            // small terms first; NP = 2: h m, m h, h h
            constexpr int PA[6] = {0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0};
            constexpr int PB[6] = {NP == 3 ? 2 : 1, 0, NP == 3 ? 1 : 0, 1, 0, 0};
            WN_UNROLL
            for (int t = 0; t < NPROD; ++t) {
                WN_UNROLL
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma_piece<F16>(af[PA[t]], bf[PB[t]][j], acc[i][j]);
                slice(i * NPROD + t);
                // One scheduling region = the TN MFMAs + their slice, issued as MFMA, <= 6 VALU, MFMA, <= 6 VALU, ...: a wave's own
                // VALU instructions are free under an MFMA only up to 6 per MFMA (tools/microbench/mfma_valu.hip: 34.5 cycles per
                // slot with 6, 47.8 with 8) -- issued as a block BEHIND the TN MFMAs (round 4) only the last MFMA covered them: the
                // 256 x 256 tile's step body took 2650 cycles for 1536 of MFMAs (profiles/r05/dw_timing_big_tile_before_interleave.txt).
                WN_UNROLL
                for (int j = 0; j < TN; ++j) {
                    WN_SGB_MFMA(1);
                    WN_SGB_VALU(6);
                }
                WN_SCHED_FENCE_ALU();
            }
        }
    };
    auto pass_fast = [&](int k_lo, int n) {   // n even, >= 2
        float ra0[AE], rb0[BE], ra1[AE], rb1[BE];
        const int k_last = k_lo + (n - 1) * 16;
        fetch_fast(k_lo, ra0, rb0);
        fetch_fast(k_lo + 16, ra1, rb1);
        stage(0, ra0, rb0);
        __syncthreads();
        for (int kb = 0; kb < n; kb += 2) {
            const int ka = k_lo + (kb + 2) * 16, kc = k_lo + (kb + 3) * 16;
            G6_STAMP(kb, 0);
            step_fast(0, 1, ra1, rb1, ra0, rb0, ka < k_last ? ka : k_last, true);
            G6_STAMP(kb, 3);
            __syncthreads();
            G6_STAMP(kb, 5);
            // past the end the staged step is a copy of the last one that nobody reads (nor counts)
            step_fast(1, 0, ra0, rb0, ra1, rb1, kc < k_last ? kc : k_last, kb + 2 < n);
            G6_STAMP(kb + 1, 3);
            __syncthreads();
            G6_STAMP(kb + 1, 5);
        }
    };
    // interior steps of this block's k-chunk: [k_a, k_b) in units of 16 from kbeg
    const int nk = (kend > kbeg) ? (kend - kbeg + 15) / 16 : 0;
    int s_a = 0, s_b = 0;
    if (a_tile_ok && b_tile_ok && nk > 0) {
        // step s (k0 = kbeg + 16 s) is interior iff k0 + 16 <= kend, k0 - b_shmax >= 0 and k0 + 16 - b_shmin <= b_clen
        int lo = b_shmax - kbeg;                 // k0 >= b_shmax
        lo = lo > 0 ? (lo + 15) / 16 : 0;
        int hi_k = kend < g.b_clen + b_shmin ? kend : g.b_clen + b_shmin;   // k0 + 16 <= hi_k
        int hi_s = (hi_k - kbeg) / 16;           // steps [0, hi_s) satisfy k0 + 16 <= hi_k
        if (hi_s > nk) hi_s = nk;
        if (hi_s - lo >= 4) {
            s_a = lo;
            s_b = lo + ((hi_s - lo) & ~1);
        }
    }
    // the 256 x 256 tile has no registers for the general pass's second operand set beside its 256 accumulators (it spilled 540
    // of them): its few non-interior steps (zero history of the shifted taps, the ragged end) go one at a time
    auto pass_simple = [&](int k_lo, int n) {
        for (int kb = 0; kb < n; ++kb) {
            float ra[AE], rb[BE];
            fetch(k_lo + kb * 16, ra, rb);
            stage(0, ra, rb);
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    };
    if (TM * TN > 8) {
        if (s_b > s_a) {
            pass_simple(kbeg, s_a);
            pass_fast(kbeg + 16 * s_a, s_b - s_a);
            pass_simple(kbeg + 16 * s_b, nk - s_b);
        } else {
            pass_simple(kbeg, nk);
        }
    } else if (s_b > s_a) {
        pass_general(kbeg, s_a);
        pass_fast(kbeg + 16 * s_a, s_b - s_a);
        pass_general(kbeg + 16 * s_b, nk - s_b);
    } else {
        pass_general(kbeg, nk);
    }

    if (do_rowsum) {
        // the 16 / AEr threads of a row are adjacent lanes
        WN_UNROLL
        for (int u = 0; u < AR; ++u) {
            float rs = rowsum[u];
            for (int m = 1; m < 16 / AEr; m <<= 1) rs += __shfl_xor(rs, m, 64);
            if (a_k == 0 && a_row_ok[u]) g.a_rowsum[(long)z * g.M + m0 + a_row + 64 * u] = rs;
        }
    }
    const wn_rsrc_t Cr = wn_make_buf(g.C + (long)z * g.c_zstride, (unsigned)((long)g.M * g.ldc * 4));
    const float c_mul = F16 ? 1.0f / a_mul : 1.0f;   // a power of two
    unsigned nonfinite = 0u;
    WN_UNROLL
    for (int i = 0; i < TM; ++i) {
        WN_UNROLL
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + li;
            WN_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + mfma32_row(r, hi);
                float v = wn_bits_f32(__builtin_bit_cast(unsigned, (float)acc[i][j][r]) ^ (a_sign & 0x80000000u));   // * (+-1)
                if (F16) {
                    nonfinite |= (~__builtin_bit_cast(unsigned, v) & 0x7f800000u) == 0u ? 1u : 0u;
                    v *= c_mul;
                }
                wn_buf_store(Cr, v, (m < g.M && n < g.N) ? (m * (int)g.ldc + n) * 4 : 0x7ffffff0, 0);
            }
        }
    }
    if (F16 && nonfinite) wn_store_coherent_int(ovf, 1);
}

// ---------------------------------------------------------------------------------------------
// k_dw_skipres8 -- skip_1x1 and res_1x1 weight gradients against ONE read of z (WnDwSkipRes, wn_gemm6.h).  The 256 x 128 tile of
// k_gemm6_dw<4, 2, 2, true> (dS rows x the z rows of two layers; same split, same alternating sign of the partials) plus the 64 dX
// rows of each of the two layers as a fifth row tile.  EIGHT waves (4 x 2, wave tile 64 x 64): two per SIMD at <= 256 registers,
// so that one wave's splits, LDS writes and barrier waits sit under the other's MFMAs -- four waves with 128 x 64 wave tiles
// needed 346 registers (one wave per SIMD) and took 0.84 - 0.92 ms where this takes 0.80 - 0.83 (the two separate launches:
// 0.97 - 1.03; profiles/r06/abk_dw_skipres.txt).  The dX row tile goes to the waves wm = 0, 1 (SIMD = wave % 4: each SIMD hosts one
// wave with and one without it).  A thread stages 4 consecutive k of two dS rows, one z row and one dX row.  No shifted taps, whole
// 64-row segments: every full 16-position step takes the branch-free pipelined pass; a tile half past the last layer (nl odd)
// reads the last layer again and stores nothing.
// ---------------------------------------------------------------------------------------------
#define WN_SR8_T 512
__global__ __launch_bounds__(WN_SR8_T) void k_dw_skipres8(WnDwSkipRes g, int order, float a_mul, int* ovf) {
    constexpr int NP = 2, NPROD = 3, TMW = 2, TN = 2;
    if (a_mul < 0.0f) a_mul = reinterpret_cast<const float*>(ovf)[1];
    constexpr int BM = 256, BN = 128;
    constexpr int AE = 8, BE = 4;               // fp32 elements per thread and step: 4 consecutive k of two dS rows / one z / one dX row
    constexpr int A_BYTES = NP * BM * 32, B_BYTES = NP * BN * 32, ST_BYTES = A_BYTES + 2 * B_BYTES;   // dS, z, dX
    WN_DYN_SMEM(smem_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = WN_UNIFORM(wave >> 1), wn = WN_UNIFORM(wave & 1);
    const bool has_r = wm < 2;
    const WnBlock blk = wn_block_order(order);
    const int zr = WN_UNIFORM(blk.z);
    const int b = zr / g.ksplit;
    const int ks = zr - b * g.ksplit;
    const int kbeg = ks * g.kchunk;
    const int kend = (g.K - kbeg > g.kchunk) ? (kbeg + g.kchunk) : g.K;
    const int bx = WN_UNIFORM(blk.x);
    const int m0 = WN_UNIFORM(blk.y) * BM;
    const int l0 = bx * TN;
    const bool with_res = blk.y == 0;
    const int N = 64 * g.nl;
    const float* __restrict__ Az = g.dS + (long)b * g.ds_zstride;
    const float* __restrict__ Bz = g.Z + (long)b * g.z_zstride;
    const float* __restrict__ Rz = g.dX + (long)b * g.dx_zstride;

    const int a_row = tid >> 2, a_k = (tid & 3) * 4;     // dS rows a_row, a_row + 128; tile row a_row of z and dX = row a_row & 63 of layer l0 + (a_row >> 6)
    long a_off[2], b_off, r_off;
    a_off[0] = (long)(m0 + a_row) * g.ds_ld;
    a_off[1] = (long)(m0 + a_row + 128) * g.ds_ld;
    {
        const int lt = l0 + (a_row >> 6);
        const int lb = lt < g.nl ? lt : g.nl - 1;                        // (a layer past the last: valid addresses, no stores)
        const int lr = lb < g.n_res ? lb : (g.n_res > 0 ? g.n_res - 1 : 0);
        b_off = (long)lb * g.z_lstride + (long)(a_row & 63) * g.z_ld;
        r_off = (long)lr * g.dx_lstride + (long)(a_row & 63) * g.dx_ld;
    }
    float rowsum[2] = {0.f, 0.f}, rowsum_r = 0.f;
    const bool do_rowsum = g.rs_skip != nullptr && bx == 0;
    const unsigned a_sign = WN_UNIFORM((zr & 1) ? 0x80008000u : 0u);

    auto fetch_a = [&](int k0, float (&ra)[AE]) {
        WN_UNROLL
        for (int u = 0; u < 2; ++u) {
            const wn_f4 v = wn_ld4_unaligned(Az + a_off[u] + k0 + a_k);
            ra[4 * u] = v.x; ra[4 * u + 1] = v.y; ra[4 * u + 2] = v.z; ra[4 * u + 3] = v.w;
        }
    };
    auto fetch_br = [&](int k0, float (&rb)[BE], float (&rr)[BE]) {
        const wn_f4 v = wn_ld4_unaligned(Bz + b_off + k0 + a_k);
        rb[0] = v.x; rb[1] = v.y; rb[2] = v.z; rb[3] = v.w;
        const wn_f4 w = wn_ld4_unaligned(Rz + r_off + k0 + a_k);
        rr[0] = w.x; rr[1] = w.y; rr[2] = w.z; rr[3] = w.w;
    };
    auto fetch_edge = [&](int k0, float (&ra)[AE], float (&rb)[BE], float (&rr)[BE]) {   // the ragged end of the last k-chunk
        WN_UNROLL
        for (int e = 0; e < AE; ++e) {
            const int k = k0 + a_k + (e & 3);
            ra[e] = k < kend ? Az[a_off[e >> 2] + k] : 0.f;
        }
        WN_UNROLL
        for (int e = 0; e < BE; ++e) {
            const int k = k0 + a_k + e;
            rb[e] = k < kend ? Bz[b_off + k] : 0.f;
            rr[e] = k < kend ? Rz[r_off + k] : 0.f;
        }
    };
    auto split_pair = [&](float x0, float x1, bool scaled, unsigned& h, unsigned& md) {
        if (scaled) {
            x0 *= a_mul;
            x1 *= a_mul;
        }
        h = wn_pk_f16(x0, x1);
        md = wn_pk_f16(x0 - wn_f16lo_f32(h), x1 - wn_f16hi_f32(h));
    };
    auto put = [&](char* base, int rows, int row, const unsigned* h, const unsigned* md, unsigned sign) {
        const unsigned* src[2] = {h, md};
        WN_UNROLL
        for (int p = 0; p < NP; ++p) {
            char* d = base + p * rows * 32;
            WN_UNROLL
            for (int q = 0; q < 2; ++q) {
                const int kq = (a_k >> 1) + q;
                *reinterpret_cast<unsigned*>(d + wn_frag_off(row, kq >> 2) + (kq & 3) * 4) = src[p][q] ^ sign;
            }
        }
    };
    // row jobs of a step: 0, 1 = this thread's dS rows, 2 = its z row, 3 = its dX row
    auto row_job = [&](int q, char* sa, const float (&ra)[AE], const float (&rb)[BE], const float (&rr)[BE], bool counted) {
        unsigned h[2], md[2];
        if (q < 2) {
            rowsum[q] += counted ? (ra[4 * q] + ra[4 * q + 1]) + (ra[4 * q + 2] + ra[4 * q + 3]) : 0.f;
            split_pair(ra[4 * q], ra[4 * q + 1], true, h[0], md[0]);
            split_pair(ra[4 * q + 2], ra[4 * q + 3], true, h[1], md[1]);
            put(sa, BM, a_row + 128 * q, h, md, a_sign);
        } else if (q == 2) {
            split_pair(rb[0], rb[1], false, h[0], md[0]);
            split_pair(rb[2], rb[3], false, h[1], md[1]);
            put(sa + A_BYTES, BN, a_row, h, md, 0u);
        } else {
            rowsum_r += counted ? (rr[0] + rr[1]) + (rr[2] + rr[3]) : 0.f;
            split_pair(rr[0], rr[1], true, h[0], md[0]);
            split_pair(rr[2], rr[3], true, h[1], md[1]);
            put(sa + A_BYTES + B_BYTES, BN, a_row, h, md, a_sign);
        }
    };
    auto stage = [&](int st, const float (&ra)[AE], const float (&rb)[BE], const float (&rr)[BE]) {
        WN_UNROLL
        for (int q = 0; q < 4; ++q) row_job(q, smem_raw + st * ST_BYTES, ra, rb, rr, true);
    };
    f32x16 acc[TMW][TN], acc_r[TN];
    WN_UNROLL
    for (int j = 0; j < TN; ++j) {
        acc[0][j] = f32x16_zero();
        acc[1][j] = f32x16_zero();
        acc_r[j] = f32x16_zero();
    }
    const int r_frag_row = wn * 64 + (wm & 1) * 32 + li;   // (waves wm = 0, 1 only)
    constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};   // small terms first: h m, m h, h h
    // One step: 12 MFMAs in six groups with one job behind each (0: the loads of a later step; 1 - 4: the row jobs of the step being
    // split into stage `stn`; pipelined = false: a plain step on stage `st`), then the six MFMAs of the dX row tile (waves wm = 0, 1)
    auto step = [&](int st, int stn, const float (&ra)[AE], const float (&rb)[BE], const float (&rr)[BE], float (&ran)[AE],
                    float (&rbn)[BE], float (&rrn)[BE], int ka_next, int kb_next, bool counted, bool pipelined) {
        const char* sa = smem_raw + st * ST_BYTES;
        const char* sb = sa + A_BYTES;
        char* da = smem_raw + stn * ST_BYTES;
        wn_f4 bf[NP][TN];
        WN_UNROLL
        for (int p = 0; p < NP; ++p) {
            WN_UNROLL
            for (int j = 0; j < TN; ++j)
                bf[p][j] = *reinterpret_cast<const wn_f4*>(sb + p * (BN * 32) + wn_frag_off((wn * TN + j) * 32 + li, hi));
        }
        WN_UNROLL
        for (int i = 0; i < TMW; ++i) {
            wn_f4 af[NP];
            WN_UNROLL
            for (int p = 0; p < NP; ++p)
                af[p] = *reinterpret_cast<const wn_f4*>(sa + p * (BM * 32) + wn_frag_off((wm * TMW + i) * 32 + li, hi));
            WN_UNROLL
            for (int t = 0; t < NPROD; ++t) {
#ifndef WN_SRX_NOMFMA   // (WN_SRX_*: what-if builds that drop one part of the k-step -- timing only, tools/build_variant.sh)
                WN_UNROLL
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma_f16(af[PA[t]], bf[PB[t]][j], acc[i][j]);
#endif
                if (!pipelined) continue;
                const int sl = i * NPROD + t;
                if (sl == 0) {
#ifndef WN_SRX_NOLOAD
                    fetch_a(ka_next, ran);
                    fetch_br(kb_next, rbn, rrn);
#endif
                } else if (sl <= 4) {
#ifndef WN_SRX_NOJOB
                    row_job(sl - 1, da, ra, rb, rr, counted);
#endif
                }
                WN_UNROLL
                for (int j = 0; j < TN; ++j) {
                    WN_SGB_MFMA(1);
                    WN_SGB_VALU(12);
                }
                WN_SCHED_FENCE_ALU();
            }
        }
        if (has_r) {
            wn_f4 af[NP];
            WN_UNROLL
            for (int p = 0; p < NP; ++p)
                af[p] = *reinterpret_cast<const wn_f4*>(sb + B_BYTES + p * (BN * 32) + wn_frag_off(r_frag_row, hi));
            WN_UNROLL
            for (int t = 0; t < NPROD; ++t) {
#ifndef WN_SRX_NOMFMA
                WN_UNROLL
                for (int j = 0; j < TN; ++j) acc_r[j] = mfma_f16(af[PA[t]], bf[PB[t]][j], acc_r[j]);
#endif
            }
        }
    };
    // The pipelined pass: THREE LDS stages.  Step k runs its MFMAs on stage k % 3, splits the operands of step k + 2 into stage
    // (k + 2) % 3 and loads step k + 4 into the register set step k + 1 left: the barrier behind a step never waits for LDS writes
    // its next step reads (with two stages every step began with that wait, one workgroup per CU: nothing else to run meanwhile).
    const int nfull = (kend > kbeg) ? (kend - kbeg) / 16 : 0;   // whole 16-position steps of this chunk
    const int nfast = nfull >= 9 ? nfull - nfull % 3 : 0;
    if (nfast > 0) {
        float ra0[AE], rb0[BE], rr0[BE], ra1[AE], rb1[BE], rr1[BE], ra2[AE], rb2[BE], rr2[BE];
        const int k_last = kbeg + (nfast - 1) * 16;
        auto kk = [&](int s_) { const int k = kbeg + s_ * 16; return k < k_last ? k : k_last; };   // past the end: the last step again (staged, never read)
        fetch_a(kbeg, ra0);
        fetch_br(kbeg, rb0, rr0);
        fetch_a(kbeg + 16, ra1);
        fetch_br(kbeg + 16, rb1, rr1);
        fetch_a(kbeg + 32, ra2);
        fetch_br(kbeg + 32, rb2, rr2);
        stage(0, ra0, rb0, rr0);
        fetch_a(kbeg + 48, ra0);
        fetch_br(kbeg + 48, rb0, rr0);
        stage(1, ra1, rb1, rr1);
        __syncthreads();
        for (int kb = 0; kb < nfast; kb += 3) {
            step(0, 2, ra2, rb2, rr2, ra1, rb1, rr1, kk(kb + 4), kk(kb + 4), kb + 2 < nfast, true);
            __syncthreads();
            step(1, 0, ra0, rb0, rr0, ra2, rb2, rr2, kk(kb + 5), kk(kb + 5), kb + 3 < nfast, true);
            __syncthreads();
            step(2, 1, ra1, rb1, rr1, ra0, rb0, rr0, kk(kb + 6), kk(kb + 6), kb + 4 < nfast, true);
            __syncthreads();
        }
    }
    for (int k0 = kbeg + 16 * nfast; k0 < kend; k0 += 16) {   // what the pipelined pass left (<= two whole steps, or a chunk below nine) and the ragged end: one step at a time
        float ra[AE], rb[BE], rr[BE];
        fetch_edge(k0, ra, rb, rr);
        stage(0, ra, rb, rr);
        __syncthreads();
        step(0, 1, ra, rb, rr, ra, rb, rr, 0, 0, false, false);
        __syncthreads();
    }

    // row sums (bias gradients): the 4 threads of a row are adjacent lanes
    WN_UNROLL
    for (int u = 0; u < 2; ++u) {
        float rs = rowsum[u];
        rs += __shfl_xor(rs, 1, 64);
        rs += __shfl_xor(rs, 2, 64);
        if (do_rowsum && a_k == 0) g.rs_skip[(long)zr * g.S + m0 + a_row + 128 * u] = rs;
    }
    const long nz = (long)g.nbatch * g.ksplit;
    {
        float rs = rowsum_r;
        rs += __shfl_xor(rs, 1, 64);
        rs += __shfl_xor(rs, 2, 64);
        const int lt = l0 + (a_row >> 6);
        if (with_res && a_k == 0 && lt < g.n_res) g.rs_res[((long)lt * nz + zr) * 64 + (a_row & 63)] = rs;
    }
    const float c_mul = 1.0f / a_mul;   // a power of two
    const unsigned osign = a_sign & 0x80000000u;
    unsigned nonfinite = 0u;
    const wn_rsrc_t Cr = wn_make_buf(g.Cskip + (long)zr * g.S * N, (unsigned)((long)g.S * N * 4));
    WN_UNROLL
    for (int i = 0; i < TMW; ++i) {
        WN_UNROLL
        for (int j = 0; j < TN; ++j) {
            const int n = 64 * l0 + (wn * TN + j) * 32 + li;
            WN_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TMW + i) * 32 + mfma32_row(r, hi);
                float v = wn_bits_f32(__builtin_bit_cast(unsigned, (float)acc[i][j][r]) ^ osign);
                nonfinite |= (~__builtin_bit_cast(unsigned, v) & 0x7f800000u) == 0u ? 1u : 0u;
                v *= c_mul;
                wn_buf_store(Cr, v, n < N ? (m * N + n) * 4 : 0x7ffffff0, 0);
            }
        }
    }
    const int lw = l0 + wn;   // the layer of this wave's columns
    if (has_r && with_res && lw < g.n_res) {
        float* Cw = g.Cres + ((long)lw * nz + zr) * 64 * 64;
        WN_UNROLL
        for (int j = 0; j < TN; ++j) {
            const int n = j * 32 + li;   // z channel inside the layer
            WN_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int m = (wm & 1) * 32 + mfma32_row(r, hi);
                float v = wn_bits_f32(__builtin_bit_cast(unsigned, (float)acc_r[j][r]) ^ osign);
                nonfinite |= (~__builtin_bit_cast(unsigned, v) & 0x7f800000u) == 0u ? 1u : 0u;
                Cw[m * 64 + n] = v * c_mul;
            }
        }
    }
    if (nonfinite) wn_store_coherent_int(ovf, 1);
}

int wn_dw_skipres_supported(int S, int R, int nl, int n_res) {
    static int on = -1;   // WN_DW_SKIPRES=0: A/B knob (the two separate launches)
    if (on < 0) { const char* e = getenv("WN_DW_SKIPRES"); on = e ? atoi(e) : 1; }
    return on && R == 64 && S >= 256 && S % 256 == 0 && nl >= 1 && n_res >= 1 && n_res <= nl;
}
int wn_dw_skipres_launch(const WnDwSkipRes* ap, float f16_mul, int* ovf, wn_stream_t st) {
    const WnDwSkipRes& a = *ap;
    if (f16_mul == 0.0f || !ovf || !wn_dw_skipres_supported(a.S, 64, a.nl, a.n_res)) return 2;
    if (a.K <= 0 || a.nbatch <= 0 || a.ksplit <= 0 || a.kchunk <= 0 || a.kchunk % 32 != 0) return 2;
    if ((long)a.S * 64 * a.nl * 4 >= 0x7ffffff0L) return 2;
    const double cols = 64.0 * a.nl, rows = (double)a.S + 64.0;
    WN_PROF("dw_skip_res", 2.0 * rows * cols * (double)a.K * a.nbatch,
            ((double)a.S * a.K + 2.0 * cols * a.K) * 4.0 * a.nbatch, st);
    dim3 grid((unsigned)((a.nl + 1) / 2), (unsigned)(a.S / 256), (unsigned)(a.nbatch * a.ksplit));
    constexpr int lds = 3 * (2 * 256 * 32 + 2 * 2 * 128 * 32);   // three stages of dS, z, dX pieces: 96 KB
#ifndef WN_EMU
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_dw_skipres8), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 3;
        attr_set = true;
    }
#endif
    WN_LAUNCH(k_dw_skipres8, grid, dim3(WN_SR8_T), lds, st, a, xcd_block_order(), f16_mul, ovf);
    return 0;
}

// 256-row tiles (k_gemm6_dw<4,2>): every B row is read once per 256 A rows.  Round 2 took them for >= 512 output rows only -- at
// 256 rows (the skip gradient: 256 x 1920, k = every position) they measured 0.06 - 0.1 ms SLOWER than 128 x 128 tiles at 3
// workgroups per CU, and no different with this round's operand mapping under six products (0.80 ms either way): the launch was
// matrix-bound.  With the three products of the fp16 pair split it is not: two 128-row tiles read z of every layer twice
// (~3.2 GB at 5.5 TB/s), one 256-row tile once -- dw_skip 0.563 -> 0.501 ms, recipe size 2.38 -> 1.85, configs[3] geometry
// 0.57 -> 0.49 (same box, profiles/r05/abk_tall256_f16.txt, dw3_probe_tall256_f16.txt).  The split-K plan of the caller must use
// the same rule (-DWN_DW_TALL_MIN_M=512: A/B builds).
#ifndef WN_DW_TALL_MIN_M
#define WN_DW_TALL_MIN_M 256
#endif
int wn_gemm6_dw_tall(int M, int N) { return M >= WN_DW_TALL_MIN_M && (M % 256 == 0) && N >= 512; }
// 256 x 256 tiles, one wave per SIMD (k_gemm6_dw<4,4>): square weight gradients of wide models (-DWN_DW_BIG=0: A/B builds)
#ifndef WN_DW_BIG
#define WN_DW_BIG 1
#endif
int wn_gemm6_dw_big(int M, int N) { return WN_DW_BIG && M >= 512 && N >= 512 && (M % 256 == 0) && (N % 256 == 0); }

// Column tiles of the weight-gradient kernel: 128 wide unless the last one would be at most half full -- kernel_size 3 has
// N = 3 * 64 = 192 columns: with 128-wide tiles the second one is ragged, i.e. never takes the branch-free interior pass
// (5.05 ms for dw_dilated of the configs[3] geometry against 1.1 ms at kernel_size 2, profiles/r03) -- then 64 wide.
int wn_gemm6_dw_tn(int M, int N) {
    // N = 192 (kernel_size 3 at 64 channels: three taps of 64 rows): ONE 192-column tile, so that the A operand (dP, two thirds
    // of the launch's bytes) is read once -- with three 64-column tiles it was read three times (PMC: 11.1 GB per launch for
    // 4.8 GB of operands, profiles/r04/pmc_traffic_config4_before.json)
    if (N == 192 && M > 64) return 3;
    return (N > 64 && (N % 128 == 0 || N % 128 > 64)) ? 2 : 1;
}

int wn_gemm6_dw_eligible(const WnGemmArgs* g) {
    return g->a_kmajor && g->b_kmajor && !g->b_index && !g->b_relu && !g->bias && !g->D && !g->E && !g->relu && !g->accumulate &&
           (long)g->M * g->ldc * 4 < 0x7ffffff0L && g->M > 0 && g->N > 0;
}

template <int TM, int TN>
static int launch_dw(const WnGemmArgs& g, int products, float f16_mul, int* ovf, wn_stream_t st) {
    dim3 grid((unsigned)((g.N + 64 * TN - 1) / (64 * TN)), (unsigned)((g.M + 64 * TM - 1) / (64 * TM)),
              (unsigned)(g.nlayer * g.nbatch * g.ksplit));
    if (f16_mul != 0.0f) {
        constexpr int lds = 2 * (2 * 64 * TM * 32 + 2 * 64 * TN * 32);
#ifndef WN_EMU
        static bool attr_set = false;
        if (lds > 65536 && !attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm6_dw<TM, TN, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 3;
            attr_set = true;
        }
#endif
        WN_LAUNCH((k_gemm6_dw<TM, TN, 2, true>), grid, dim3(G6_T), lds, st, g, xcd_block_order(), f16_mul, ovf G6_DBG_ARG(g.tag));
    } else if (products == 3) {
        constexpr int lds = 2 * (2 * 64 * TM * 32 + 2 * 64 * TN * 32);
#ifndef WN_EMU
        static bool attr_set = false;
        if (lds > 65536 && !attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm6_dw<TM, TN, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 3;
            attr_set = true;
        }
#endif
        WN_LAUNCH((k_gemm6_dw<TM, TN, 2>), grid, dim3(G6_T), lds, st, g, xcd_block_order(), 1.0f, static_cast<int*>(nullptr) G6_DBG_ARG(g.tag));
    } else {
        constexpr int lds = 2 * (3 * 64 * TM * 32 + 3 * 64 * TN * 32);
#ifndef WN_EMU
        static bool attr_set = false;
        if (lds > 65536 && !attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm6_dw<TM, TN, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 3;
            attr_set = true;
        }
#endif
        WN_LAUNCH((k_gemm6_dw<TM, TN, 3>), grid, dim3(G6_T), lds, st, g, xcd_block_order(), 1.0f, ovf G6_DBG_ARG(g.tag));
    }
    return 0;
}

// products: 6 (default: fp32-equivalent) or 3 (h h + h m + m h of bf16 pieces: leaf results only, see k_gemm6_dw).
// f16_mul > 0: the fp16 pair split instead (three products, A scaled by f16_mul = a power of two; *ovf := 1 if a block's result is
// not finite).  f16_mul == 0, products == 6 and ovf != NULL: the launch does its work only if *ovf != 0 (the exact redo).
int wn_gemm6_dw_launch(const WnGemmArgs* gp, int products, float f16_mul, int* ovf, wn_stream_t st) {
    const WnGemmArgs& g = *gp;
    if (products != 3 && products != 6) return 2;
    if ((f16_mul != 0.0f && !ovf) || (ovf && f16_mul == 0.0f && products != 6)) return 2;
    if (!wn_gemm6_dw_eligible(gp)) return 1;
    if (g.K < 0 || g.nbatch <= 0 || g.ksplit <= 0 || g.nlayer <= 0 || g.b_seg_len <= 0 || g.kchunk <= 0) return 2;
    const bool redo = ovf && f16_mul == 0.0f;   // the conditional redo: no work unless an fp16 launch overflowed
    WN_PROF(redo ? "dw_redo_if_overflow" : (g.tag ? g.tag : "gemm6_dw"), redo ? 0.0 : 2.0 * g.M * g.N * (double)g.K * g.nbatch * g.nlayer,
            redo ? 0.0 : ((double)g.M * g.K * 4.0 + (double)g.K * 4.0 * g.N + (double)g.M * g.N * 4.0) * g.nbatch * g.nlayer, st);
    const int tm = g.M > 64 ? 2 : 1, tn = wn_gemm6_dw_tn(g.M, g.N);
    if (wn_gemm6_dw_big(g.M, g.N)) return launch_dw<4, 4>(g, products, f16_mul, ovf, st);     // 256 x 256 tiles, one wave per SIMD
    if (wn_gemm6_dw_tall(g.M, g.N)) return launch_dw<4, 2>(g, products, f16_mul, ovf, st);   // 256 x 128 tiles: every B row is read once per 256 A rows
    if (tm == 2 && tn == 3) return launch_dw<2, 3>(g, products, f16_mul, ovf, st);
    if (tm == 2 && tn == 2) return launch_dw<2, 2>(g, products, f16_mul, ovf, st);
    if (tm == 2) return launch_dw<2, 1>(g, products, f16_mul, ovf, st);
    if (tn == 2) return launch_dw<1, 2>(g, products, f16_mul, ovf, st);
    return launch_dw<1, 1>(g, products, f16_mul, ovf, st);
}
