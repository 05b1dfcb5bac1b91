// wn_fused.hip -- fused residual-block kernels for n_resch == 64 on CDNA4 (gfx950).
//
// Design (MI355X-first, not a translation of the reference's conv calls):
//   * time runs across the 32 columns of a v_mfma_f32_32x32x2_f32 tile, channels down its rows;
//     activations stay channel-major (B, C, T), so every B-operand load and every store of a wave
//     is 32 consecutive samples of one channel (128 B segments, two per instruction);
//   * the whole weight set of one residual block (dilated taps 2R x K*R, res 1x1, biases) is staged
//     ONCE per workgroup into LDS (80 KB of the 160 KB) and re-used by all tiles the persistent
//     workgroup walks; the A operand is a conflict-free ds_read_b32 (32 consecutive floats);
//   * one wave owns one 32-sample tile end to end: dilated conv (256 MFMAs) -> gate in registers
//     -> the gate output z is fed STRAIGHT from the accumulator registers into the res 1x1 MFMAs
//     (64 MFMAs): the contraction order k of an MFMA is free, so k-step (tile, r) takes channel
//     kappa = 32*tile + (r&3) + 8*(r>>2) + 4*(lane>>5) -- exactly the channel accumulator
//     register r of that lane already holds.  No LDS round trip, no shuffles;
//   * the upsampled aux features are never materialised: P = conv + w[t%U] * G[:, t/U] + c with
//     G = Waux.h at FRAME rate (computed once per step for all layers);
//   * all global traffic goes through buffer instructions: one per-lane byte offset (time) in a
//     VGPR, the channel row offset in an SGPR -> no per-access 64-bit address registers;
//   * memory latency is hidden by explicit software pipelining, not by occupancy (80 KB of LDS
//     weights allow 2 waves per SIMD): the operands of the NEXT tile are issued before the res
//     MFMAs of the current one, the aux/gate inputs before the current-tap MFMAs, and
//     sched_barrier pins that issue order (hipcc otherwise sinks every load to its first use and
//     waits vmcnt(0) on each, and both waves of a SIMD stall in lock step).
// The f32-input MFMA is an exact fp32 fma chain, so parity with the fp32 reference is kept.
#include "wn_fused.h"
#include <type_traits>


#include "wn_prof.h"

#include <stdlib.h>
#include <string.h>

// -DWN_EXP_NPROD=3: TIMING-ONLY experiment builds (tools/build_variant.sh): the split fused kernels issue 3 of their 6 products per
// multiply -- numerically WRONG, what the matrix work of a two-piece operand split would cost.  The product is built with 6.
#ifndef WN_EXP_NPROD
#define WN_EXP_NPROD 6
#endif

#ifdef WN_TIMING
// Experimental build only (tools/exp): per-phase cycle stamps of block 0, lane 0 of every wave.
static long long* g_dbg = nullptr;
extern "C" void wn_debug_set_buffer(void* p) { g_dbg = (long long*)p; }
#define WN_STAMP(i) do { if (a.dbg && a.Xnext != nullptr && blockIdx.x == 0 && lane == 0 && tcount < 4) a.dbg[(wave * 4 + tcount) * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define WN_STAMP(i)
#endif

// tuning knob (A/B on hardware): WN_CHAIN_BLOCKS=<n> caps the persistent grid of the fused kernels (default 256 = one
// workgroup per CU); measures how many CUs the HBM-bound chain kernels need to keep their rate
static long chain_blocks() {
    static long v = -1;
    if (v < 0) {
        const char* e = getenv("WN_CHAIN_BLOCKS");
        v = e ? atol(e) : 256;
        if (v < 8) v = 256;
    }
    return v;
}

// -DWN_PRIO_PHASES (A/B builds): a wave raises its issue priority for its MFMA phases and drops it for its gate phase
#ifndef WN_PRIO_MFMA
#define WN_PRIO_MFMA 3
#endif
#ifndef WN_PRIO_GATE
#define WN_PRIO_GATE 0
#endif
// Issue priority by phase (default; -DWN_NO_PRIO_PHASES builds without it).  The two waves of a SIMD compete for one issue
// port; a wave in its MFMA phase issues one instruction every ~32 cycles and is otherwise idle, a wave in its gate phase issues
// VALU work back to back -- and tools/microbench/mfma_valu.hip shows that VALU work of the neighbour slows a wave's MFMAs
// down 3x when both have the same priority.  A wave therefore raises its priority (s_setprio) for its MFMA phases and drops it
// for its gate phase: the MFMAs go out when their operands are ready, the neighbour's gate math fills the gaps.  Same box:
// forward blocks 1.89 -> 1.82 ms per step, backward chain 2.94 -> 2.91 (profiles/r02/ab_probe_prio_phases.txt); the opposite
// assignment is slower than none (1.94).
#if !defined(WN_NO_PRIO_PHASES) && !defined(WN_EMU)
#ifdef WN_PRIO_GATE_BY_WAVE   // experiment: the second wave of a SIMD keeps priority 1 in its gate phase
#define WN_PRIO(n) do { if ((n) == WN_PRIO_GATE && (threadIdx.x >> 8)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(n); } while (0)
#else
#define WN_PRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
#else
#define WN_PRIO(n)
#endif
// Loads of data a launch reads exactly once (saved gate activations, the pre-contracted skip gradient, the residual input of
// a chain launch) are NON-TEMPORAL: they no longer displace the lines that are read twice (the shifted taps) from the XCD's
// 4 MB L2.  Same box: 10.40 -> 10.16 ms per step (chain 2.95 -> 2.71 ms; profiles/r03/visit6..8: the same for the operands
// of the split contractions / weight gradients changed nothing, non-temporal STORES cost 0.1 - 0.2 ms, a tile walk that
// alternates direction from layer to layer gained 0.04 ms without and nothing with the non-temporal loads).
#define WN_LD_DXN wn_buf_load_once
#ifndef WN_FT
#define WN_FT 512  // threads per workgroup (8 waves = 2 per SIMD)
#endif
#define WN_FW (WN_FT / 64)  // waves per workgroup
#ifndef WN_LB
#define WN_LB WN_FT   // launch bound of the fused kernels (experiments: fewer threads under the same register cap)
#endif

// Persistent grid of the fused kernels: the smallest multiple of 8 workgroups (XCD-aware walk) that needs no more
// rounds of tiles than one workgroup per CU would.  Config 2 has 5 760 tiles: 256 x 8 waves walk them in 2.81 -> 3
// rounds, 240 x 8 in exactly 3 -- same time (12.95 vs 13.05 ms/step measured, profiles/r01/cosched_probe.txt), and 16
// CUs stay free for whatever runs beside the chain (the RCCL kernels of the gradient all-reduce).
static long balanced_blocks(long ntiles, long fw = WN_FW) {
    const long cap = chain_blocks();
    long nblk = (ntiles + fw - 1) / fw;
    if (nblk <= cap) return nblk;
    const long rounds = (ntiles + cap * fw - 1) / (cap * fw);
    long nb = (ntiles + rounds * fw - 1) / (rounds * fw);
    nb = (nb + 7) / 8 * 8;
    return nb < cap ? nb : cap;
}

// channel handled by k-step s (0..31 within a 64-channel group) for lane-half hi
static __device__ __forceinline__ int kappa64(int s, int hi) {
    return 32 * (s >> 4) + ((s & 15) & 3) + 8 * ((s & 15) >> 2) + 4 * hi;
}

int wn_fused_supported(int R, int K, int S) {
    if (R != 64 || K < 1 || K > 3 || S % 32 != 0) return 0;
    const long fwd = ((long)K * 64 * 128 + 64 * 64 + 192) * 4;
    const long gate = ((long)S * 64 + 64 * 64) * 4;
    const long dx = ((long)K * 128 * 64) * 4;
    const long lim = 160 * 1024;
    return fwd <= lim && gate <= lim && dx <= lim;
}

#ifndef WN_EMU
template <class Kern>
static int set_lds(Kern kern, size_t bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return e == hipSuccess ? 0 : 1;
}
#else
template <class Kern>
static int set_lds(Kern, size_t) { return 0; }
#endif

// XCD-aware tile mapping.  Workgroup b is dispatched to XCD b % 8 (each XCD has its own 4 MB L2): the tile
// space is cut into 8 contiguous time ranges, one per XCD, and the 256 waves of an XCD walk their range
// together, so the shifted taps of a tile (up to 512 samples = 16 tiles back / ahead) were just read by a
// neighbouring wave of the SAME XCD and hit its L2 instead of going out to the fabric.
struct TileWalk {
    int first, end, step;
};

static __device__ __forceinline__ TileWalk tile_walk(int ntiles, int wave, int fw = WN_FW) {   // fw: tile-walking waves per workgroup
    TileWalk w;
    if ((gridDim.x & 7) == 0) {
        const int per = (ntiles + 7) >> 3, x = blockIdx.x & 7;
        const int lo = x * per;
        w.end = lo + per < ntiles ? lo + per : ntiles;
        w.first = lo + (blockIdx.x >> 3) * fw + wave;
        w.step = (gridDim.x >> 3) * fw;
    } else {
        w.end = ntiles;
        w.first = blockIdx.x * fw + wave;
        w.step = gridDim.x * fw;
    }
    return w;
}

static __device__ __forceinline__ void stage_copy(float* dst, const float* __restrict__ src, int n) {
    // n is a multiple of 4; dst is 16-byte aligned; src usually is (checked, block-uniform branch)
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int i = threadIdx.x; i < n / 4; i += WN_FT) d4[i] = s4[i];
    } else {
        for (int i = threadIdx.x; i < n; i += WN_FT) dst[i] = src[i];
    }
}

// ---------------------------------------------------------------------------------------------
struct FwdArgs {
    const float* wimg;   // split kernels: pre-built LDS weight image of this layer (wn_fused_pack_images) or NULL
    const float* wd_f;
    const float* wres_f;
    const float* cvec;
    const float* res_bias;
    const float* X;
    const float* G;
    long g_bstride;
    const float* upw;
    float* Xnext;
    float* S;
    float* Gt;
    float* Z;
    int B, T, dil, U, F;
#ifdef WN_TIMING
    long long* dbg;
#endif
};

template <int K>
__global__ __launch_bounds__(WN_LB) void k_resblock_fwd(FwdArgs a) {
    WN_DYN_SMEM(smem_raw);
    float* Wd = reinterpret_cast<float*>(smem_raw);  // [K*64][128]
    float* Wr = Wd + K * 64 * 128;                   // [64][64]
    float* cv = Wr + 64 * 64;                        // [128]
    float* rb = cv + 128;                            // [64]
#ifdef WN_TIMING
    if (a.dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        a.dbg[((threadIdx.x >> 6) * 4) * 16 + 5] = (long long)__builtin_readcyclecounter();
        a.dbg[((threadIdx.x >> 6) * 4) * 16 + 7] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    if (a.dbg && threadIdx.x == 0) a.dbg[512 + blockIdx.x * 4 + 0] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    stage_copy(Wd, a.wd_f, K * 64 * 128);
    stage_copy(Wr, a.wres_f, 64 * 64);
    if (threadIdx.x < 128) cv[threadIdx.x] = a.cvec[threadIdx.x];
    if (threadIdx.x < 64) rb[threadIdx.x] = a.res_bias[threadIdx.x];
    __syncthreads();
#ifdef WN_TIMING
    if (a.dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0)
        a.dbg[((threadIdx.x >> 6) * 4) * 16 + 6] = (long long)__builtin_readcyclecounter();
#endif
    // De-phase the two waves that share a SIMD (waves w and w+4): started together they would run
    // their MFMA phases and their gate/store phases in lock step and leave the matrix pipe idle
    // during the latter; half a tile of head start makes one wave's VALU/VMEM phase coincide with
    // the other's MFMA phase.

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int T = a.T;
    const int T4 = T * 4;  // bytes per channel row
    const int F4 = a.F * 4;
    const int tiles_per_b = (T + 31) >> 5;
    const int ntiles = a.B * tiles_per_b;
    const unsigned slab = (unsigned)(64 * T4);
    const TileWalk walk = tile_walk(ntiles, threadIdx.x >> 6);
    const int step = walk.step, tile_end = walk.end;
    constexpr int KH = (K > 1) ? (K - 1) : 1;  // history taps (shift > 0)

    // Software pipeline (per wave, per 32-sample tile):
    //   history-tap operands xh : issued before the res MFMAs of the PREVIOUS tile (cross-tile prefetch)
    //   current-tap operands xc : issued at tile start, land under the history-tap MFMAs
    //   aux/gate inputs         : first half issued before the current-tap MFMAs, second half
    //                             before the gate math of the first half
    // Operands are consumed in place (zero history / dead lanes selected at use).
    float xh[KH][32];
    bool okh[KH];
    auto issue_hist = [&](int tl_v) {
        const int tl = WN_UNIFORM(tl_v);
        const int b = tl / tiles_per_b;
        const int t = (tl - b * tiles_per_b) * 32 + li;
        const wn_rsrc_t Xr = wn_make_buf(a.X + (long)b * 64 * T, slab);
        WN_UNROLL
        for (int tap = 0; tap + 1 < K; ++tap) {
            const int ts = t - (K - 1 - tap) * a.dil;
            const bool ok = (t < T) && ts >= 0;
            okh[tap] = ok;
            const int vt = ok ? (4 * hi * T + ts) * 4 : 0;  // dead lanes read a valid dummy address
            WN_UNROLL
            for (int s = 0; s < 32; ++s) xh[tap][s] = wn_buf_load(Xr, vt, kappa64(s, 0) * T4);
        }
    };

    int tile_v = walk.first;
    int tcount = 0;
    (void)tcount;
    if (K > 1 && tile_v < tile_end) issue_hist(tile_v);
    while (tile_v < tile_end) {
        WN_STAMP(0);
        const int tile = WN_UNIFORM(tile_v);
        const int b = tile / tiles_per_b;
        const int t = (tile - b * tiles_per_b) * 32 + li;
        const bool inb = t < T;
        const int tc = inb ? t : T - 1;
        const int vcur = inb ? (4 * hi * T + t) * 4 : 0;

        // current tap (shift 0): raw loads now, consumed after the history taps
        float xc[32];
        {
            const wn_rsrc_t Xr = wn_make_buf(a.X + (long)b * 64 * T, slab);
            WN_UNROLL
            for (int s = 0; s < 32; ++s) xc[s] = wn_buf_load(Xr, vcur, kappa64(s, 0) * T4);
        }
        WN_SCHED_BARRIER();
        f32x16 acc[4];
        WN_UNROLL
        for (int q = 0; q < 4; ++q) acc[q] = f32x16_zero();
        // dilated taps with history (shift > 0)
        WN_UNROLL
        for (int tap = 0; tap + 1 < K; ++tap) {
            const float* Wt = Wd + tap * 64 * 128 + 4 * hi * 128 + li;
            // ping-pong LDS operand sets: the reads of k-step s+1 are in flight during the MFMAs of step s
            float a0[4], a1[4];
            WN_UNROLL
            for (int q = 0; q < 4; ++q) a0[q] = Wt[kappa64(0, 0) * 128 + 32 * q];
            WN_SGB_DS(2);  // prologue group: from here on every [DS][MFMA] pair = (next operands, current MFMAs)
            WN_UNROLL
            for (int s = 0; s < 32; s += 2) {
                WN_UNROLL
                for (int q = 0; q < 4; ++q) a1[q] = Wt[kappa64(s + 1, 0) * 128 + 32 * q];
                {
                    const float xv = okh[tap] ? xh[tap][s] : 0.0f;
                    WN_UNROLL
                    for (int q = 0; q < 4; ++q) acc[q] = mfma32(a0[q], xv, acc[q]);
                }
                WN_SGB_DS(2);
                WN_SGB_MFMA(4);
                if (s + 2 < 32) {
                    WN_UNROLL
                    for (int q = 0; q < 4; ++q) a0[q] = Wt[kappa64(s + 2, 0) * 128 + 32 * q];
                }
                {
                    const float xv = okh[tap] ? xh[tap][s + 1] : 0.0f;
                    WN_UNROLL
                    for (int q = 0; q < 4; ++q) acc[q] = mfma32(a1[q], xv, acc[q]);
                }
                WN_SGB_DS(2);
                WN_SGB_MFMA(4);
            }
        }
        WN_STAMP(1);  // after history-tap MFMAs
        // aux / gate inputs (frame rate, L2 resident), first 32 gate channels
        const int fr = tc / a.U;
        const float upw_j = a.upw[tc - fr * a.U];
        const wn_rsrc_t Gr = wn_make_buf(a.G + (long)b * a.g_bstride, (unsigned)(128 * F4));
        const int vg = (4 * hi * a.F + fr) * 4;
        // one register set for both 32-channel halves: an element of the second half is requested right after the
        // element of the first half in the same register has been consumed (64 registers less across the gate phase)
        float ga[16], gg[16];
        WN_UNROLL
        for (int r = 0; r < 16; ++r) {
            ga[r] = wn_buf_load(Gr, vg, mfma32_row(r, 0) * F4);
            gg[r] = wn_buf_load(Gr, vg, (mfma32_row(r, 0) + 64) * F4);
        }
        WN_SCHED_BARRIER();
        // current tap; xc is also the residual input, already in D layout
        {
            const float* Wt = Wd + (K - 1) * 64 * 128 + 4 * hi * 128 + li;
            // ping-pong LDS operand sets: the reads of k-step s+1 are in flight during the MFMAs of step s
            float a0[4], a1[4];
            WN_UNROLL
            for (int q = 0; q < 4; ++q) a0[q] = Wt[kappa64(0, 0) * 128 + 32 * q];
            WN_SGB_DS(2);  // prologue group: from here on every [DS][MFMA] pair = (next operands, current MFMAs)
            WN_UNROLL
            for (int s = 0; s < 32; s += 2) {
                WN_UNROLL
                for (int q = 0; q < 4; ++q) a1[q] = Wt[kappa64(s + 1, 0) * 128 + 32 * q];
                {
                    const float xv = inb ? xc[s] : 0.0f;
                    WN_UNROLL
                    for (int q = 0; q < 4; ++q) acc[q] = mfma32(a0[q], xv, acc[q]);
                }
                WN_SGB_DS(2);
                WN_SGB_MFMA(4);
                if (s + 2 < 32) {
                    WN_UNROLL
                    for (int q = 0; q < 4; ++q) a0[q] = Wt[kappa64(s + 2, 0) * 128 + 32 * q];
                }
                {
                    const float xv = inb ? xc[s + 1] : 0.0f;
                    WN_UNROLL
                    for (int q = 0; q < 4; ++q) acc[q] = mfma32(a1[q], xv, acc[q]);
                }
                WN_SGB_DS(2);
                WN_SGB_MFMA(4);
            }
        }
        WN_SCHED_BARRIER();
        WN_STAMP(2);  // after current-tap MFMAs
        // Prefetch the history-tap operands of this wave's next tile NOW, i.e. before the stores of the
        // gate phase: vmcnt is one in-order counter for loads AND stores, so loads issued behind the
        // 96 S/Gt/Z stores could only be waited for together with those stores' acknowledgements.
        const int next_v = tile_v + step;
        if (K > 1 && next_v < tile_end) issue_hist(next_v);
        // gate (reference wavenet.py:529-532): P = conv + w[j]*G[row][f] + c[row]; saved for backward
        const wn_rsrc_t Sr = wn_make_buf(a.S + (long)b * 64 * T, slab);
        const bool keep_g = a.Gt != nullptr;   // NULL: the tanh half is not saved (backward rebuilds it as z / s)
        const wn_rsrc_t Gtr = wn_make_buf((keep_g ? a.Gt : a.S) + (long)b * 64 * T, slab);
        const wn_rsrc_t Zr = wn_make_buf(a.Z + (long)b * 64 * T, slab);
        const float* cvl = cv + 4 * hi;
        f32x16 z[2];
        const int vst = inb ? vcur : WN_VOFF_DEAD;
        auto gate_phase = [&](auto keep_tag) {
        constexpr bool KEEP_G = decltype(keep_tag)::value;
        WN_UNROLL
        for (int q = 0; q < 2; ++q) {
            WN_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row0 = 32 * q + mfma32_row(r, 0);  // + 4*hi is in the per-lane offsets
                const float pa = acc[q][r] + (upw_j * ga[r] + cvl[row0]);
                const float pg = acc[q + 2][r] + (upw_j * gg[r] + cvl[row0 + 64]);
                if (q == 0) {
                    ga[r] = wn_buf_load(Gr, vg, (32 + mfma32_row(r, 0)) * F4);
                    gg[r] = wn_buf_load(Gr, vg, (96 + mfma32_row(r, 0)) * F4);
                }
                const float s = wn_sigmoid(pa);
                const float g = wn_tanh(pg);
                const float zz = s * g;
                z[q][r] = zz;
                // unconditional stores: lanes past T carry an out-of-range offset (dropped by the buffer range check), and
                // the tanh half goes through the same instruction stream only when it is kept -- a lane- or kernel-
                // conditional store here would cut the gate phase into one basic block per element (no overlap of the
                // exp / rcp chains of different elements: measured 11000 cycles for ~4500 cycles of arithmetic)
                wn_buf_store(Sr, s, vst, row0 * T4);
                if (KEEP_G) wn_buf_store(Gtr, g, vst, row0 * T4);
                wn_buf_store(Zr, zz, vst, row0 * T4);
            }
        }
        };
        if (keep_g) gate_phase(std::true_type{});
        else gate_phase(std::false_type{});
        WN_STAMP(3);  // after gate math + S/Gt/Z stores issued
        f32x16 racc[2];
        if (a.Xnext != nullptr) {
            const float* rbl = rb + 4 * hi;
            WN_UNROLL
            for (int q = 0; q < 2; ++q) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r)
                    racc[q][r] = (inb ? xc[16 * q + r] : 0.0f) + rbl[32 * q + mfma32_row(r, 0)];
            }
        }
        WN_SCHED_BARRIER();
        // res 1x1 + residual; z is consumed straight from the accumulator registers
        if (a.Xnext != nullptr) {
            const float* Wrl = Wr + 4 * hi * 64 + li;
            // stage = two k-steps (4 LDS operands, 4 MFMAs); ping-pong operand sets as above
            float a0[4], a1[4];
            WN_UNROLL
            for (int q = 0; q < 4; ++q) a0[q] = Wrl[kappa64(q >> 1, 0) * 64 + 32 * (q & 1)];
            WN_SGB_DS(2);
            WN_UNROLL
            for (int s = 0; s < 32; s += 4) {
                WN_UNROLL
                for (int q = 0; q < 4; ++q) a1[q] = Wrl[kappa64(s + 2 + (q >> 1), 0) * 64 + 32 * (q & 1)];
                racc[0] = mfma32(a0[0], z[s >> 4][s & 15], racc[0]);
                racc[1] = mfma32(a0[1], z[s >> 4][s & 15], racc[1]);
                racc[0] = mfma32(a0[2], z[(s + 1) >> 4][(s + 1) & 15], racc[0]);
                racc[1] = mfma32(a0[3], z[(s + 1) >> 4][(s + 1) & 15], racc[1]);
                WN_SGB_DS(2);
                WN_SGB_MFMA(4);
                if (s + 4 < 32) {
                    WN_UNROLL
                    for (int q = 0; q < 4; ++q) a0[q] = Wrl[kappa64(s + 4 + (q >> 1), 0) * 64 + 32 * (q & 1)];
                }
                racc[0] = mfma32(a1[0], z[(s + 2) >> 4][(s + 2) & 15], racc[0]);
                racc[1] = mfma32(a1[1], z[(s + 2) >> 4][(s + 2) & 15], racc[1]);
                racc[0] = mfma32(a1[2], z[(s + 3) >> 4][(s + 3) & 15], racc[0]);
                racc[1] = mfma32(a1[3], z[(s + 3) >> 4][(s + 3) & 15], racc[1]);
                WN_SGB_DS(2);
                WN_SGB_MFMA(4);
            }
            {
                const wn_rsrc_t Xn = wn_make_buf(a.Xnext + (long)b * 64 * T, slab);
                WN_UNROLL
                for (int q = 0; q < 2; ++q) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) wn_buf_store(Xn, racc[q][r], vst, (32 * q + mfma32_row(r, 0)) * T4);
                }
            }
        }
        WN_STAMP(4);  // tile done
        ++tcount;
        tile_v = next_v;
    }
#ifdef WN_TIMING
    if (a.dbg && blockIdx.x == 0 && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        a.dbg[(wave * 4) * 16 + 8] = (long long)__builtin_amdgcn_s_memrealtime();
        a.dbg[(wave * 4) * 16 + 9] = (long long)__builtin_readcyclecounter();
    }
    if (a.dbg && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        if (wave == 0) a.dbg[512 + blockIdx.x * 4 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
        if (wave == 7) a.dbg[512 + blockIdx.x * 4 + 2] = (long long)__builtin_amdgcn_s_memrealtime();
        if (wave == 0) a.dbg[512 + blockIdx.x * 4 + 3] = tcount;
    }
#endif
}

// The same residual block on the bf16 matrix cores (3-way operand split, six products, fp32
// accumulate: fp32-equivalent, see wn_gemm6.hip).  The register layouts of the activation operands
// (xh, xc, z: k-step s of lane half hi = channel kappa64(s, hi)) are unchanged; 8 consecutive steps
// form the lane's share of one 16-k block, i.e. block kb holds channels 16 kb .. 16 kb + 15 and
// position (hi, e) of the block is channel 16 kb + (e&3) + 8 (e>>2) + 4 hi.  The weights are split
// once per launch into LDS with exactly that k order: [block][piece][row][hi*8 + e].
static __device__ __forceinline__ void split8(const float (&x)[8], wn_f4 (&bf)[3]) {
    unsigned hq[4], mq[4], lq[4];
    WN_UNROLL
    for (int e = 0; e < 4; ++e) {
        const float x0 = x[2 * e], x1 = x[2 * e + 1];
        hq[e] = wn_pk_bf16(x0, x1);
        const float r0 = x0 - wn_bits_f32(hq[e] << 16), r1 = x1 - wn_bits_f32(hq[e] & 0xffff0000u);
        mq[e] = wn_pk_bf16(r0, r1);
        lq[e] = wn_pk_bf16(r0 - wn_bits_f32(mq[e] << 16), r1 - wn_bits_f32(mq[e] & 0xffff0000u));
    }
    bf[0].x = wn_bits_f32(hq[0]); bf[0].y = wn_bits_f32(hq[1]); bf[0].z = wn_bits_f32(hq[2]); bf[0].w = wn_bits_f32(hq[3]);
    bf[1].x = wn_bits_f32(mq[0]); bf[1].y = wn_bits_f32(mq[1]); bf[1].z = wn_bits_f32(mq[2]); bf[1].w = wn_bits_f32(mq[3]);
    bf[2].x = wn_bits_f32(lq[0]); bf[2].y = wn_bits_f32(lq[1]); bf[2].z = wn_bits_f32(lq[2]); bf[2].w = wn_bits_f32(lq[3]);
}
// 8 source values (stride `st` floats) of one (row, block half) -> the three LDS pieces
static __device__ __forceinline__ void split_to_lds(const float* src, int st, char* dst, int piece_bytes) {
    float x[8];
    WN_UNROLL
    for (int e = 0; e < 8; ++e) x[e] = src[((e & 3) + 8 * (e >> 2)) * st];
    wn_f4 bf[3];
    split8(x, bf);
    WN_UNROLL
    for (int p = 0; p < 3; ++p) *reinterpret_cast<wn_f4*>(dst + p * piece_bytes) = bf[p];
}


// ---------------------------------------------------------------------------------------------
// LDS weight images of the split kernels.  A kernel either builds its image itself (every workgroup splits the whole
// weight set again: ~5 us of latency-bound prologue per launch) or copies an image that wn_fused_pack_images built ONCE
// per step with the SAME functions: the copy is 16 bytes per lane global -> LDS without registers.
// ---------------------------------------------------------------------------------------------
static __host__ __device__ constexpr int fwd_image_bytes(int K) { return K * 4 * (3 * 128 * 32) + 4 * (3 * 64 * 32); }
static __host__ __device__ constexpr int chain_taps_bytes(int K) { return K * 8 * 6144; }
#define WN_RES_T_BYTES (4 * 6144)

// wd_f[(tap*64 + i)*128 + o'] , wres_f[i*64 + o]: channel i of (block kb, half h, e) = 16 kb + 4 h + (e&3) + 8 (e>>2)
template <int K>
static __device__ __forceinline__ void fill_fwd_image(char* Wd, char* Wr, const float* wd_f, const float* wres_f, int tid, int nthr) {
    constexpr int WD_BLK = 3 * 128 * 32, WR_BLK = 3 * 64 * 32;
    for (int idx = tid; idx < K * 4 * 2 * 128; idx += nthr) {
        const int o = idx & 127, h = (idx >> 7) & 1, blk = idx >> 8;  // blk = tap*4 + kb
        const int tap = blk >> 2, kb = blk & 3;
        split_to_lds(wd_f + (long)(tap * 64 + 16 * kb + 4 * h) * 128 + o, 128, Wd + blk * WD_BLK + wn_frag_off(o, h), 128 * 32);
    }
    for (int idx = tid; idx < 4 * 2 * 64; idx += nthr) {
        const int o = idx & 63, h = (idx >> 6) & 1, kb = idx >> 7;
        split_to_lds(wres_f + (long)(16 * kb + 4 * h) * 64 + o, 64, Wr + kb * WR_BLK + wn_frag_off(o, h), 64 * 32);
    }
}
// tap blocks of the chain kernel: chunk q (32 channels) = tap q % K, channel group q / K; two 6 KB blocks per chunk
static __device__ __forceinline__ void fill_chain_taps(char* W, const float* wd_b, int K, int tid, int nthr) {
    for (int idx = tid; idx < K * 4 * 2 * 128; idx += nthr) {
        const int o = idx & 63, h = (idx >> 6) & 1, kbg = idx >> 7;
        const int q = kbg >> 1;
        const int tap = q % K;
        const int c = (q / K) * 32 + (kbg & 1) * 16 + 8 * h;
        const float* src = wd_b + ((long)tap * 128 + c) * 64 + o;
        float x[8];
        WN_UNROLL
        for (int e = 0; e < 8; ++e) x[e] = src[e * 64];
        wn_f4 bf[3];
        split8(x, bf);
        WN_UNROLL
        for (int p = 0; p < 3; ++p) *reinterpret_cast<wn_f4*>(W + kbg * 6144 + p * 2048 + wn_frag_off(o, h)) = bf[p];
    }
}
// Wres^T of the gate half: natural res_1x1 weight [k = o][row = i], k order = accumulator register order
static __device__ __forceinline__ void fill_res_t(char* Wr, const float* wres, int tid, int nthr) {
    for (int idx = tid; idx < 4 * 2 * 64; idx += nthr) {
        const int o = idx & 63, h = (idx >> 6) & 1, kb = idx >> 7;
        split_to_lds(wres + (long)(16 * kb + 4 * h) * 64 + o, 64, Wr + kb * 6144 + wn_frag_off(o, h), 2048);
    }
}
// `bytes` (a multiple of 1024) from a 16-byte aligned global image to the start of the dynamic LDS, 16 bytes per lane
static __device__ __forceinline__ void copy_image_to_lds(char* lds, const float* img, int bytes) {
    const wn_rsrc_t Ir = wn_make_buf(img, (unsigned)bytes);
    const int lane = threadIdx.x & 63, wave_u = WN_UNIFORM((int)(threadIdx.x >> 6));
    for (int off = wave_u * 1024; off < bytes; off += WN_FW * 1024) wn_buf_load_lds16(Ir, lds + off, lane * 16, (unsigned)off);
}

struct PackImgArgs {
    const float* wd_f;    // [L][K*64][128]
    const float* wres_f;  // [L][64][64]
    const float* wd_b;    // [L][K][128][64]
    const float* params;  // flat parameter buffer; natural res_1x1 weight of layer l at params + res_off + l * res_lstride
    long res_off, res_lstride;
    float* img_fwd;       // [L][fwd_image_bytes / 4]
    float* img_taps;      // [L][chain_taps_bytes / 4]
    float* img_res;       // [L][WN_RES_T_BYTES / 4]
    int K;
};
template <int K>
__global__ __launch_bounds__(WN_LB) void k_fused_pack_images(PackImgArgs a) {
    const int l = blockIdx.x, kind = blockIdx.y;
    if (kind == 0) {
        char* img = reinterpret_cast<char*>(a.img_fwd) + (long)l * fwd_image_bytes(K);
        fill_fwd_image<K>(img, img + K * 4 * (3 * 128 * 32), a.wd_f + (long)l * K * 64 * 128, a.wres_f + (long)l * 64 * 64,
                          threadIdx.x, WN_FT);
    } else if (kind == 1) {
        fill_chain_taps(reinterpret_cast<char*>(a.img_taps) + (long)l * chain_taps_bytes(K), a.wd_b + (long)l * K * 128 * 64, K,
                        threadIdx.x, WN_FT);
    } else {
        fill_res_t(reinterpret_cast<char*>(a.img_res) + (long)l * WN_RES_T_BYTES, a.params + a.res_off + (long)l * a.res_lstride,
                   threadIdx.x, WN_FT);
    }
}

long wn_fused_image_floats(int K, int L, int which) {
    if (K < 1 || K > 3) return 0;
    const long per = which == 0 ? fwd_image_bytes(K) : which == 1 ? chain_taps_bytes(K) : WN_RES_T_BYTES;
    return (long)L * per / 4;
}

int wn_fused_pack_images(const float* wd_f, const float* wres_f, const float* wd_b, const float* params, long res_off,
                         long res_lstride, float* img_fwd, float* img_taps, float* img_res, int K, int L, wn_stream_t st) {
    WN_PROF("fused_pack_images", 0.0, 0.0, st);
    if (K < 1 || K > 3) return 1;
    PackImgArgs a;
    a.wd_f = wd_f; a.wres_f = wres_f; a.wd_b = wd_b; a.params = params; a.res_off = res_off; a.res_lstride = res_lstride;
    a.img_fwd = img_fwd; a.img_taps = img_taps; a.img_res = img_res; a.K = K;
    if (K == 1) WN_LAUNCH((k_fused_pack_images<1>), dim3((unsigned)L, 3), dim3(WN_FT), 0, st, a);
    else if (K == 2) WN_LAUNCH((k_fused_pack_images<2>), dim3((unsigned)L, 3), dim3(WN_FT), 0, st, a);
    else WN_LAUNCH((k_fused_pack_images<3>), dim3((unsigned)L, 3), dim3(WN_FT), 0, st, a);
    return 0;
}

template <int K>
__global__ __launch_bounds__(WN_LB) void k_resblock_fwd_s(FwdArgs a) {
    WN_DYN_SMEM(smem_raw);
    constexpr int WD_BLK = 3 * 128 * 32, WR_BLK = 3 * 64 * 32;  // bytes of one 16-k block
    char* Wd = smem_raw;                                         // [K*4 blocks][piece][128 rows][16 k] bf16
    // K = 3: the three taps alone fill the LDS (144 KB), so the res-1x1 fragments (24 KB per layer, L2 resident) are read
    // from the layer's pre-split image in global memory, one 16-k block ahead of their MFMAs (RG = "res from global")
    constexpr bool RG = (K >= 3);
    char* Wr = Wd + K * 4 * WD_BLK;                              // [4 blocks][piece][64 rows][16 k] bf16   (not with RG)
    float* cv = reinterpret_cast<float*>(Wr + (RG ? 0 : 4 * WR_BLK));   // [128]
    float* rb = cv + 128;                                        // [64]
#ifdef WN_TIMING
    if (a.dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        a.dbg[((threadIdx.x >> 6) * 4) * 16 + 5] = (long long)__builtin_readcyclecounter();
        a.dbg[((threadIdx.x >> 6) * 4) * 16 + 7] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    if (a.dbg && threadIdx.x == 0) a.dbg[512 + blockIdx.x * 4 + 0] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    if (RG) {   // (the launcher only takes this kernel with an image)
        copy_image_to_lds(smem_raw, a.wimg, K * 4 * WD_BLK);
        WN_WAIT_VMCNT(0);
    } else if (a.wimg != nullptr) {
        copy_image_to_lds(smem_raw, a.wimg, fwd_image_bytes(K));   // pre-split once per step (wn_fused_pack_images)
        WN_WAIT_VMCNT(0);
    } else {
        fill_fwd_image<K>(Wd, Wr, a.wd_f, a.wres_f, threadIdx.x, WN_FT);
    }
    if (threadIdx.x < 128) cv[threadIdx.x] = a.cvec[threadIdx.x];
    if (threadIdx.x < 64) rb[threadIdx.x] = a.res_bias[threadIdx.x];
    __syncthreads();
#ifdef WN_TIMING
    if (a.dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0)
        a.dbg[((threadIdx.x >> 6) * 4) * 16 + 6] = (long long)__builtin_readcyclecounter();
#endif
    // De-phase the two waves that share a SIMD (waves w and w+4): started together they would run
    // their MFMA phases and their gate/store phases in lock step and leave the matrix pipe idle
    // during the latter; half a tile of head start makes one wave's VALU/VMEM phase coincide with
    // the other's MFMA phase.

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int T = a.T;
    const int T4 = T * 4;  // bytes per channel row
    const int F4 = a.F * 4;
    const int tiles_per_b = (T + 31) >> 5;
    const int ntiles = a.B * tiles_per_b;
    const unsigned slab = (unsigned)(64 * T4);
    const TileWalk walk = tile_walk(ntiles, threadIdx.x >> 6);
    const int step = WN_UNIFORM(walk.step), tile_end = WN_UNIFORM(walk.end);
    constexpr int KH = (K > 1) ? (K - 1) : 1;  // history taps (shift > 0)
    // history taps requested one tile ahead (across the gate phase of the previous tile).  K = 3 keeps only the oldest tap
    // that way and requests the middle one at tile start with the current tap: 32 registers less across the gate phase.
    constexpr int KP = (K >= 3) ? 1 : K - 1;

    // Software pipeline (per wave, per 32-sample tile):
    //   history-tap operands xh : issued before the res MFMAs of the PREVIOUS tile (cross-tile prefetch)
    //   current-tap operands xc : issued at tile start, land under the history-tap MFMAs
    //   aux/gate inputs         : first half issued before the current-tap MFMAs, second half
    //                             before the gate math of the first half
    // Operands are consumed in place (zero history / dead lanes selected at use).
    float xh[KH][32];
    bool okh[KH];
    auto issue_hist = [&](int tl_v) {
        const int tl = WN_UNIFORM(tl_v);
        const int b = tl / tiles_per_b;
        const int t = (tl - b * tiles_per_b) * 32 + li;
        const wn_rsrc_t Xr = wn_make_buf(a.X + (long)b * 64 * T, slab);
        WN_UNROLL
        for (int tap = 0; tap < KP; ++tap) {
            const int ts = t - (K - 1 - tap) * a.dil;
            const bool ok = (t < T) && ts >= 0;
            okh[tap] = ok;
            const int vt = ok ? (4 * hi * T + ts) * 4 : 0;  // dead lanes read a valid dummy address
            WN_UNROLL
            for (int s = 0; s < 32; ++s) xh[tap][s] = wn_buf_load(Xr, vt, kappa64(s, 0) * T4);
        }
    };

    int tile_v = WN_UNIFORM(walk.first);
    int tcount = 0;
    (void)tcount;
    if (K > 1 && tile_v < tile_end) issue_hist(tile_v);
    while (tile_v < tile_end) {
        WN_STAMP(0);
        const int tile = WN_UNIFORM(tile_v);
        const int b = tile / tiles_per_b;
        const int t = (tile - b * tiles_per_b) * 32 + li;
        const bool inb = t < T;
        const int tc = inb ? t : T - 1;
        const int vcur = inb ? (4 * hi * T + t) * 4 : 0;

        // current tap (shift 0): raw loads now, consumed after the history taps
        float xc[32];
        {
            const wn_rsrc_t Xr = wn_make_buf(a.X + (long)b * 64 * T, slab);
            WN_UNROLL
            for (int tap = KP; tap + 1 < K; ++tap) {   // (K = 3) the history taps that are not requested a tile ahead
                const int ts = t - (K - 1 - tap) * a.dil;
                const bool ok = inb && ts >= 0;
                okh[tap] = ok;
                const int vt = ok ? (4 * hi * T + ts) * 4 : 0;
                WN_UNROLL
                for (int s = 0; s < 32; ++s) xh[tap][s] = wn_buf_load(Xr, vt, kappa64(s, 0) * T4);
            }
            WN_UNROLL
            for (int s = 0; s < 32; ++s) xc[s] = wn_buf_load(Xr, vcur, kappa64(s, 0) * T4);
        }
        WN_SCHED_BARRIER();
        WN_PRIO(WN_PRIO_MFMA);
        f32x16 acc[4];
        WN_UNROLL
        for (int q = 0; q < 4; ++q) acc[q] = f32x16_zero();
        // dilated taps with history (shift > 0)
        WN_UNROLL
        for (int tap = 0; tap + 1 < K; ++tap) {
            WN_UNROLL
            for (int kb = 0; kb < 4; ++kb) {
                float x8[8];
                WN_UNROLL
                for (int e = 0; e < 8; ++e) x8[e] = okh[tap] ? xh[tap][8 * kb + e] : 0.0f;
                wn_f4 bf[3];
                split8(x8, bf);
                const char* Wl = Wd + (tap * 4 + kb) * WD_BLK + wn_frag_off(li, hi);
                constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};  // small terms first
                WN_UNROLL
                for (int qh = 0; qh < 4; qh += 2) {  // two row tiles at a time (register budget)
                    wn_f4 af[2][3];
                    WN_UNROLL
                    for (int q = 0; q < 2; ++q) {
                        WN_UNROLL
                        for (int p = 0; p < 3; ++p)
                            af[q][p] = *reinterpret_cast<const wn_f4*>(Wl + p * (128 * 32) + (qh + q) * 1024);
                    }
                    WN_UNROLL
                    for (int t6 = 0; t6 < WN_EXP_NPROD; ++t6) {
                        acc[qh] = mfma_bf16(af[0][PA[t6]], bf[PB[t6]], acc[qh]);
                        acc[qh + 1] = mfma_bf16(af[1][PA[t6]], bf[PB[t6]], acc[qh + 1]);
                    }
                }
            }
        }
        WN_STAMP(1);  // after history-tap MFMAs
        // aux / gate inputs (frame rate, L2 resident), first 32 gate channels
        const int fr = tc / a.U;
        const float upw_j = a.upw[tc - fr * a.U];
        const wn_rsrc_t Gr = wn_make_buf(a.G + (long)b * a.g_bstride, (unsigned)(128 * F4));
        const int vg = (4 * hi * a.F + fr) * 4;
        // one register set for both 32-channel halves: an element of the second half is requested right after the
        // element of the first half in the same register has been consumed (64 registers less across the gate phase)
        float ga[16], gg[16];
        WN_UNROLL
        for (int r = 0; r < 16; ++r) {
            ga[r] = wn_buf_load(Gr, vg, mfma32_row(r, 0) * F4);
            gg[r] = wn_buf_load(Gr, vg, (mfma32_row(r, 0) + 64) * F4);
        }
        WN_SCHED_BARRIER();
        // current tap; xc is also the residual input, already in D layout
        {
            WN_UNROLL
            for (int kb = 0; kb < 4; ++kb) {
                float x8[8];
                WN_UNROLL
                for (int e = 0; e < 8; ++e) x8[e] = inb ? xc[8 * kb + e] : 0.0f;
                wn_f4 bf[3];
                split8(x8, bf);
                const char* Wl = Wd + ((K - 1) * 4 + kb) * WD_BLK + wn_frag_off(li, hi);
                constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
                WN_UNROLL
                for (int qh = 0; qh < 4; qh += 2) {  // two row tiles at a time (register budget)
                    wn_f4 af[2][3];
                    WN_UNROLL
                    for (int q = 0; q < 2; ++q) {
                        WN_UNROLL
                        for (int p = 0; p < 3; ++p)
                            af[q][p] = *reinterpret_cast<const wn_f4*>(Wl + p * (128 * 32) + (qh + q) * 1024);
                    }
                    WN_UNROLL
                    for (int t6 = 0; t6 < WN_EXP_NPROD; ++t6) {
                        acc[qh] = mfma_bf16(af[0][PA[t6]], bf[PB[t6]], acc[qh]);
                        acc[qh + 1] = mfma_bf16(af[1][PA[t6]], bf[PB[t6]], acc[qh + 1]);
                    }
                }
            }
        }
        WN_SCHED_BARRIER();
        WN_PRIO(WN_PRIO_GATE);
        WN_STAMP(2);  // after current-tap MFMAs
        // Prefetch the history-tap operands of this wave's next tile NOW, i.e. before the stores of the
        // gate phase: vmcnt is one in-order counter for loads AND stores, so loads issued behind the
        // 96 S/Gt/Z stores could only be waited for together with those stores' acknowledgements.
        // residual input + bias = the initial value of the res-1x1 accumulators, formed NOW: xc dies here, so that in a
        // chain its registers simply become the next tile's history operands (no second copy of the tile is ever live)
        // (Round 3: the sums are ADDED to the res products after the last MFMA instead of being the accumulators' initial
        // value -- products aligned against a large initial value lose their low bits with a consistent sign, see k_chain64s.)
        f32x16 xb2[2];
        if (a.Xnext != nullptr) {
            const float* rbl = rb + 4 * hi;
            WN_UNROLL
            for (int q = 0; q < 2; ++q) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r)
                    xb2[q][r] = (inb ? xc[16 * q + r] : 0.0f) + rbl[32 * q + mfma32_row(r, 0)];
            }
#ifndef WN_EMU
            // pin: the sums exist from here on (machine sinking would otherwise move them below the branch and keep xc alive)
            asm volatile("" : "+v"(xb2[0]), "+v"(xb2[1]));
#endif
        }
        // next tile of this wave: its history taps are requested NOW, before the stores of the gate phase.  (A variant that
        // kept the history tap of a d >= 32 layer in registers across "tile chains" measured slower -- 60 B/lane of scratch and
        // 16-tile runs instead of one contiguous span: profiles/r02/ab_probe_fwd_chain.txt -- and was removed in round 3.)
        const int next_v = tile_v + step;
        if (K > 1 && next_v < tile_end) issue_hist(next_v);
        // gate (reference wavenet.py:529-532): P = conv + w[j]*G[row][f] + c[row]; saved for backward
        const wn_rsrc_t Sr = wn_make_buf(a.S + (long)b * 64 * T, slab);
        const bool keep_g = a.Gt != nullptr;   // NULL: the tanh half is not saved (backward rebuilds it as z / s)
        const wn_rsrc_t Gtr = wn_make_buf((keep_g ? a.Gt : a.S) + (long)b * 64 * T, slab);
        const wn_rsrc_t Zr = wn_make_buf(a.Z + (long)b * 64 * T, slab);
        const float* cvl = cv + 4 * hi;
        f32x16 z[2];
        const int vst = inb ? vcur : WN_VOFF_DEAD;
        auto gate_phase = [&](auto keep_tag) {
        constexpr bool KEEP_G = decltype(keep_tag)::value;
        WN_UNROLL
        for (int q = 0; q < 2; ++q) {
            WN_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row0 = 32 * q + mfma32_row(r, 0);  // + 4*hi is in the per-lane offsets
                const float pa = acc[q][r] + (upw_j * ga[r] + cvl[row0]);
                const float pg = acc[q + 2][r] + (upw_j * gg[r] + cvl[row0 + 64]);
                if (q == 0) {
                    ga[r] = wn_buf_load(Gr, vg, (32 + mfma32_row(r, 0)) * F4);
                    gg[r] = wn_buf_load(Gr, vg, (96 + mfma32_row(r, 0)) * F4);
                }
                const float s = wn_sigmoid(pa);
                const float g = wn_tanh(pg);
                const float zz = s * g;
                z[q][r] = zz;
                // unconditional stores: lanes past T carry an out-of-range offset (dropped by the buffer range check), and
                // the tanh half goes through the same instruction stream only when it is kept -- a lane- or kernel-
                // conditional store here would cut the gate phase into one basic block per element (no overlap of the
                // exp / rcp chains of different elements: measured 11000 cycles for ~4500 cycles of arithmetic)
                wn_buf_store(Sr, s, vst, row0 * T4);
                if (KEEP_G) wn_buf_store(Gtr, g, vst, row0 * T4);
                wn_buf_store(Zr, zz, vst, row0 * T4);
            }
        }
        };
        if (keep_g) gate_phase(std::true_type{});
        else gate_phase(std::false_type{});
        WN_STAMP(3);  // after gate math + S/Gt/Z stores issued
        WN_SCHED_BARRIER();
        WN_PRIO(WN_PRIO_MFMA);
        // res 1x1 + residual; z is consumed straight from the accumulator registers
        if (a.Xnext != nullptr) {
            f32x16 racc[2];
            racc[0] = f32x16_zero();
            racc[1] = f32x16_zero();
            const wn_rsrc_t Ir = wn_make_buf(a.wimg, (unsigned)(RG ? fwd_image_bytes(K) : 16));
            const int vfrag = wn_frag_off(li, hi);
            auto res_frags = [&](int kb, wn_f4 (&af)[2][3]) {
                WN_UNROLL
                for (int q = 0; q < 2; ++q) {
                    WN_UNROLL
                    for (int p = 0; p < 3; ++p) {
                        if (RG) {
                            const float4 f = wn_buf_load4(Ir, vfrag, (unsigned)(K * 4 * WD_BLK + kb * WR_BLK + p * (64 * 32) + q * 1024));
                            af[q][p] = wn_f4{f.x, f.y, f.z, f.w};
                        } else {
                            af[q][p] = *reinterpret_cast<const wn_f4*>(Wr + kb * WR_BLK + vfrag + p * (64 * 32) + q * 1024);
                        }
                    }
                }
            };
            wn_f4 afr[2][2][3];   // two sets: with RG the fragments of block kb + 1 are in flight under the MFMAs of block kb
            res_frags(0, afr[0]);
            WN_UNROLL
            for (int kb = 0; kb < 4; ++kb) {
                float x8[8];
                WN_UNROLL
                for (int e = 0; e < 8; ++e) x8[e] = z[(8 * kb + e) >> 4][(8 * kb + e) & 15];
                wn_f4 bf[3];
                split8(x8, bf);
                if (kb + 1 < 4) res_frags(kb + 1, afr[(kb + 1) & 1]);
                constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
                WN_UNROLL
                for (int t6 = 0; t6 < WN_EXP_NPROD; ++t6) {
                    racc[0] = mfma_bf16(afr[kb & 1][0][PA[t6]], bf[PB[t6]], racc[0]);
                    racc[1] = mfma_bf16(afr[kb & 1][1][PA[t6]], bf[PB[t6]], racc[1]);
                }
            }
            WN_UNROLL
            for (int q = 0; q < 2; ++q) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) racc[q][r] += xb2[q][r];
            }
            {
                const wn_rsrc_t Xn = wn_make_buf(a.Xnext + (long)b * 64 * T, slab);
                WN_UNROLL
                for (int q = 0; q < 2; ++q) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) wn_buf_store(Xn, racc[q][r], vst, (32 * q + mfma32_row(r, 0)) * T4);
                }
            }
        }
        WN_PRIO(WN_PRIO_GATE);
        WN_STAMP(4);  // tile done
        ++tcount;
        tile_v = next_v;
    }
#ifdef WN_TIMING
    if (a.dbg && blockIdx.x == 0 && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        a.dbg[(wave * 4) * 16 + 8] = (long long)__builtin_amdgcn_s_memrealtime();
        a.dbg[(wave * 4) * 16 + 9] = (long long)__builtin_readcyclecounter();
    }
    if (a.dbg && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        if (wave == 0) a.dbg[512 + blockIdx.x * 4 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
        if (wave == 7) a.dbg[512 + blockIdx.x * 4 + 2] = (long long)__builtin_amdgcn_s_memrealtime();
        if (wave == 0) a.dbg[512 + blockIdx.x * 4 + 3] = tcount;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// k_resblock_fwd_h -- the same residual block on the FP16 matrix cores with a two-piece operand split (round 6,
// WN_FLAG_FUSED_F16PAIR): x ~ h + l with two fp16 pieces (11 + 11 significand bits), the three products h h + h l + l h on
// v_mfma_f32_32x32x16_f16, fp32 accumulate -- ~2^-22 |w x| per product (the rounding of an fp32 running sum over the 128 - 192
// terms of a tap contraction) at HALF the matrix work of the six bf16 products, a weight image of 80 KB instead of 120 KB
// (K = 3: 112 KB, so the res-1x1 fragments are back in the LDS), and 6 instead of 11 VALU instructions per operand pair.
// fp16 has 5 exponent bits, so everything is BLOCK-SCALED by powers of two (exact):
//   * each weight image by 2^ew with max |w| 2^ew in [2^11, 2^12) (wn_fused_pack_images measures the maximum per layer);
//   * each 64-channel x 32-sample operand tile by 2^e with max |x| 2^e in [2^13, 2^14) -- the wave reduces the maximum of the tile
//     it holds in registers; the taps of a tile share ONE scale (the smallest so far: the accumulators are multiplied down when a
//     later tap has larger values), so nothing can leave fp16's range whatever the magnitudes are, and an element 2^-18 below the
//     tile's maximum still has all 22 bits (smaller ones keep 2^-25 of the maximum in absolute terms);
//   * z in (-1, 1) by 2^13.
// The inverse scales are folded into the additions that follow the MFMAs (gate pre-activation, residual): no extra pass.
// Pipeline, registers, stores: k_resblock_fwd_s's.
// ---------------------------------------------------------------------------------------------
static __host__ __device__ constexpr int fwd16_image_bytes(int K) { return K * 4 * (2 * 128 * 32) + 4 * (2 * 64 * 32) + 64; }
// split of 8 values times s into the two fp16 pieces of the lane's share of a 16-k block
static __device__ __forceinline__ void split8h(const float (&x)[8], float s, wn_f4 (&bf)[2]) {
    unsigned hq[4], lq[4];
    WN_UNROLL
    for (int e = 0; e < 4; ++e) {
        const float x0 = x[2 * e] * s, x1 = x[2 * e + 1] * s;
        hq[e] = wn_pk_f16(x0, x1);
        lq[e] = wn_pk_f16(x0 - wn_f16lo_f32(hq[e]), x1 - wn_f16hi_f32(hq[e]));
    }
    bf[0].x = wn_bits_f32(hq[0]); bf[0].y = wn_bits_f32(hq[1]); bf[0].z = wn_bits_f32(hq[2]); bf[0].w = wn_bits_f32(hq[3]);
    bf[1].x = wn_bits_f32(lq[0]); bf[1].y = wn_bits_f32(lq[1]); bf[1].z = wn_bits_f32(lq[2]); bf[1].w = wn_bits_f32(lq[3]);
}
// power of two s with amax * s in [2^(E-1), 2^E)  (amax == 0 or tiny / huge: clamped, s stays a normal float with a normal inverse)
static __host__ __device__ __forceinline__ float wn_pow2_scale(float amax, int E) {
    unsigned bits;
#ifdef __HIP_DEVICE_COMPILE__
    bits = __builtin_bit_cast(unsigned, amax);
#else
    memcpy(&bits, &amax, 4);
#endif
    int ex = (int)((bits >> 23) & 0xffu);              // amax in [2^(ex-127), 2^(ex-126))
    ex = ex < 27 ? 27 : (ex > 254 ? 254 : ex);         // (zero / denormal tiles: a harmless finite scale; s and 1 / s stay normal)
    const unsigned sb = (unsigned)(E + 253 - ex) << 23;   // 2^(E - (ex - 126))
    float sc;
#ifdef __HIP_DEVICE_COMPILE__
    sc = __builtin_bit_cast(float, sb);
#else
    memcpy(&sc, &sb, 4);
#endif
    return sc;
}
static __device__ __forceinline__ float wave_amax32(const float (&x)[32], bool ok) {
    float m = 0.0f;
    WN_UNROLL
    for (int e = 0; e < 32; e += 2) m = fmaxf(m, fmaxf(fabsf(x[e]), fabsf(x[e + 1])));
    m = ok ? m : 0.0f;
    return wave_reduce_max(m);
}

// image: [K*4 blocks][2 pieces][128 rows][16 k] fp16 taps, [4 blocks][2][64][16] res 1x1, then {1 / 2^ew_taps, 1 / 2^ew_res} floats
template <int K>
static __device__ __forceinline__ void fill_fwd16_image(char* img, const float* wd_f, const float* wres_f, float* red, int tid, int nthr) {
    constexpr int WD_BLK = 2 * 128 * 32, WR_BLK = 2 * 64 * 32;
    // maxima of the two weight sets (block-wide)
    float m0 = 0.0f, m1 = 0.0f;
    for (int i = tid; i < K * 64 * 128; i += nthr) m0 = fmaxf(m0, fabsf(wd_f[i]));
    for (int i = tid; i < 64 * 64; i += nthr) m1 = fmaxf(m1, fabsf(wres_f[i]));
    m0 = wave_reduce_max(m0);
    m1 = wave_reduce_max(m1);
    if ((tid & 63) == 0) { red[tid >> 6] = m0; red[16 + (tid >> 6)] = m1; }
    __syncthreads();
    m0 = 0.0f; m1 = 0.0f;
    for (int i = 0; i < (nthr >> 6); ++i) { m0 = fmaxf(m0, red[i]); m1 = fmaxf(m1, red[16 + i]); }
    const float s0 = wn_pow2_scale(m0, 12), s1 = wn_pow2_scale(m1, 12);
    char* Wd = img;
    char* Wr = img + K * 4 * WD_BLK;
    for (int idx = tid; idx < K * 4 * 2 * 128; idx += nthr) {
        const int o = idx & 127, h = (idx >> 7) & 1, blk = idx >> 8;  // blk = tap*4 + kb
        const int tap = blk >> 2, kb = blk & 3;
        const float* src = wd_f + (long)(tap * 64 + 16 * kb + 4 * h) * 128 + o;
        float x[8];
        WN_UNROLL
        for (int e = 0; e < 8; ++e) x[e] = src[((e & 3) + 8 * (e >> 2)) * 128];
        wn_f4 bf[2];
        split8h(x, s0, bf);
        WN_UNROLL
        for (int p = 0; p < 2; ++p) *reinterpret_cast<wn_f4*>(Wd + blk * WD_BLK + p * (128 * 32) + wn_frag_off(o, h)) = bf[p];
    }
    for (int idx = tid; idx < 4 * 2 * 64; idx += nthr) {
        const int o = idx & 63, h = (idx >> 6) & 1, kb = idx >> 7;
        const float* src = wres_f + (long)(16 * kb + 4 * h) * 64 + o;
        float x[8];
        WN_UNROLL
        for (int e = 0; e < 8; ++e) x[e] = src[((e & 3) + 8 * (e >> 2)) * 64];
        wn_f4 bf[2];
        split8h(x, s1, bf);
        WN_UNROLL
        for (int p = 0; p < 2; ++p) *reinterpret_cast<wn_f4*>(Wr + kb * WR_BLK + p * (64 * 32) + wn_frag_off(o, h)) = bf[p];
    }
    if (tid == 0) {
        float* tail = reinterpret_cast<float*>(Wr + 4 * WR_BLK);
        tail[0] = 1.0f / s0;
        tail[1] = 1.0f / s1;
    }
}

// block-wide maximum of |w[0 .. n)| (every thread returns it); red: 16 floats of shared memory
static __device__ __forceinline__ float block_amax(const float* w, int n, float* red, int tid, int nthr) {
    float m = 0.0f;
    for (int i = tid; i < n; i += nthr) m = fmaxf(m, fabsf(w[i]));
    m = wave_reduce_max(m);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = 0.0f;
    for (int i = 0; i < (nthr >> 6); ++i) m = fmaxf(m, red[i]);
    return m;
}
// two-piece fp16 images of the backward chain (k_chain64s<.., H16>): the tap blocks in fill_chain_taps' order, 4 KB each, and
// Wres^T (4 blocks); each image is followed by the inverse of the power of two it was scaled by (64-byte tail)
static __host__ __device__ constexpr int chain_taps16_bytes(int K) { return K * 8 * 4096 + 64; }
#define WN_RES_T16_BYTES (4 * 4096 + 64)
static __device__ __forceinline__ void fill_chain_taps16(char* W, const float* wd_b, int K, float* red, int tid, int nthr) {
    const float sc = wn_pow2_scale(block_amax(wd_b, K * 128 * 64, red, tid, nthr), 12);
    for (int idx = tid; idx < K * 4 * 2 * 128; idx += nthr) {
        const int o = idx & 63, h = (idx >> 6) & 1, kbg = idx >> 7;
        const int q = kbg >> 1;
        const int tap = q % K;
        const int c = (q / K) * 32 + (kbg & 1) * 16 + 8 * h;
        const float* src = wd_b + ((long)tap * 128 + c) * 64 + o;
        float x[8];
        WN_UNROLL
        for (int e = 0; e < 8; ++e) x[e] = src[e * 64];
        wn_f4 bf[2];
        split8h(x, sc, bf);
        WN_UNROLL
        for (int p = 0; p < 2; ++p) *reinterpret_cast<wn_f4*>(W + kbg * 4096 + p * 2048 + wn_frag_off(o, h)) = bf[p];
    }
    if (tid == 0) *reinterpret_cast<float*>(W + K * 8 * 4096) = 1.0f / sc;
}
static __device__ __forceinline__ void fill_res_t16(char* Wr, const float* wres, float* red, int tid, int nthr) {
    const float sc = wn_pow2_scale(block_amax(wres, 64 * 64, red, tid, nthr), 12);
    for (int idx = tid; idx < 4 * 2 * 64; idx += nthr) {
        const int o = idx & 63, h = (idx >> 6) & 1, kb = idx >> 7;
        const float* src = wres + (long)(16 * kb + 4 * h) * 64 + o;
        float x[8];
        WN_UNROLL
        for (int e = 0; e < 8; ++e) x[e] = src[((e & 3) + 8 * (e >> 2)) * 64];
        wn_f4 bf[2];
        split8h(x, sc, bf);
        WN_UNROLL
        for (int p = 0; p < 2; ++p) *reinterpret_cast<wn_f4*>(Wr + kb * 4096 + p * 2048 + wn_frag_off(o, h)) = bf[p];
    }
    if (tid == 0) *reinterpret_cast<float*>(Wr + 4 * 4096) = 1.0f / sc;
}

struct PackImg16Args {
    const float* wd_f;    // [L][K*64][128]
    const float* wres_f;  // [L][64][64]
    const float* wd_b;    // [L][K][128][64]
    const float* params;  // natural res_1x1 weight of layer l at params + res_off + l * res_lstride
    long res_off, res_lstride;
    float* img_fwd16;     // [L][fwd16_image_bytes / 4] or NULL
    float* img_taps16;    // [L][chain_taps16_bytes / 4] or NULL
    float* img_res16;     // [L][WN_RES_T16_BYTES / 4]
};
template <int K>
__global__ __launch_bounds__(WN_LB) void k_fused_pack_images16(PackImg16Args a) {
    __shared__ float red[32];
    const int l = blockIdx.x, kind = blockIdx.y;
    if (kind == 0) {
        if (a.img_fwd16)
            fill_fwd16_image<K>(reinterpret_cast<char*>(a.img_fwd16) + (long)l * fwd16_image_bytes(K), a.wd_f + (long)l * K * 64 * 128,
                                a.wres_f + (long)l * 64 * 64, red, threadIdx.x, WN_FT);
    } else if (a.img_taps16) {
        if (kind == 1)
            fill_chain_taps16(reinterpret_cast<char*>(a.img_taps16) + (long)l * chain_taps16_bytes(K), a.wd_b + (long)l * K * 128 * 64, K,
                              red, threadIdx.x, WN_FT);
        else
            fill_res_t16(reinterpret_cast<char*>(a.img_res16) + (long)l * WN_RES_T16_BYTES, a.params + a.res_off + (long)l * a.res_lstride,
                         red, threadIdx.x, WN_FT);
    }
}

long wn_fused_image16_floats(int K, int L, int which) {
    if (K < 1 || K > 3) return 0;
    const long per = which == 0 ? fwd16_image_bytes(K) : which == 1 ? chain_taps16_bytes(K) : WN_RES_T16_BYTES;
    return (long)L * per / 4;
}

int wn_fused_pack_images16(const float* wd_f, const float* wres_f, const float* wd_b, const float* params, long res_off,
                           long res_lstride, float* img_fwd16, float* img_taps16, float* img_res16, int K, int L, wn_stream_t st) {
    WN_PROF("fused_pack_images16", 0.0, 0.0, st);
    PackImg16Args a;
    a.wd_f = wd_f; a.wres_f = wres_f; a.wd_b = wd_b; a.params = params; a.res_off = res_off; a.res_lstride = res_lstride;
    a.img_fwd16 = img_fwd16; a.img_taps16 = img_taps16; a.img_res16 = img_res16;
    const dim3 grid((unsigned)L, img_taps16 ? 3u : 1u);
    if (K == 1) WN_LAUNCH((k_fused_pack_images16<1>), grid, dim3(WN_FT), 0, st, a);
    else if (K == 2) WN_LAUNCH((k_fused_pack_images16<2>), grid, dim3(WN_FT), 0, st, a);
    else if (K == 3) WN_LAUNCH((k_fused_pack_images16<3>), grid, dim3(WN_FT), 0, st, a);
    else return 1;
    return 0;
}

template <int K>
__global__ __launch_bounds__(WN_LB) void k_resblock_fwd_h(FwdArgs a) {
    WN_DYN_SMEM(smem_raw);
    constexpr int WD_BLK = 2 * 128 * 32, WR_BLK = 2 * 64 * 32;  // bytes of one 16-k block (two fp16 pieces)
    char* Wd = smem_raw;                                         // [K*4 blocks][piece][128 rows][16 k]
    char* Wr = Wd + K * 4 * WD_BLK;                              // [4 blocks][piece][64 rows][16 k]
    float* tail = reinterpret_cast<float*>(Wr + 4 * WR_BLK);     // {1 / 2^ew_taps, 1 / 2^ew_res} (+ padding to 64 bytes)
    float* cv = tail + 16;                                       // [128]
    float* rb = cv + 128;                                        // [64]
#ifdef WN_TIMING
    if (a.dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        a.dbg[((threadIdx.x >> 6) * 4) * 16 + 5] = (long long)__builtin_readcyclecounter();
        a.dbg[((threadIdx.x >> 6) * 4) * 16 + 7] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    if (a.dbg && threadIdx.x == 0) a.dbg[512 + blockIdx.x * 4 + 0] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    copy_image_to_lds(smem_raw, a.wimg, fwd16_image_bytes(K) - 64);   // (a multiple of 1024)
    if (threadIdx.x < 2) tail[threadIdx.x] = a.wimg[(fwd16_image_bytes(K) - 64) / 4 + threadIdx.x];
    if (threadIdx.x < 128) cv[threadIdx.x] = a.cvec[threadIdx.x];
    if (threadIdx.x < 64) rb[threadIdx.x] = a.res_bias[threadIdx.x];
    WN_WAIT_VMCNT(0);
    __syncthreads();
    const float inv_wd = tail[0], inv_wr = tail[1];
#ifdef WN_TIMING
    if (a.dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0)
        a.dbg[((threadIdx.x >> 6) * 4) * 16 + 6] = (long long)__builtin_readcyclecounter();
    const int wave = threadIdx.x >> 6;
#endif
    const int lane = threadIdx.x & 63;
    const int li = lane & 31, hi = lane >> 5;
    const int T = a.T;
    const int T4 = T * 4;  // bytes per channel row
    const int F4 = a.F * 4;
    const int tiles_per_b = (T + 31) >> 5;
    const int ntiles = a.B * tiles_per_b;
    const unsigned slab = (unsigned)(64 * T4);
    const TileWalk walk = tile_walk(ntiles, threadIdx.x >> 6);
    const int step = WN_UNIFORM(walk.step), tile_end = WN_UNIFORM(walk.end);
    constexpr int KH = (K > 1) ? (K - 1) : 1;  // history taps (shift > 0)
    constexpr int KP = (K >= 3) ? 1 : K - 1;   // ... requested one tile ahead (see k_resblock_fwd_s)

    float xh[KH][32];
    bool okh[KH];
    auto issue_hist = [&](int tl_v) {
        const int tl = WN_UNIFORM(tl_v);
        const int b = tl / tiles_per_b;
        const int t = (tl - b * tiles_per_b) * 32 + li;
        const wn_rsrc_t Xr = wn_make_buf(a.X + (long)b * 64 * T, slab);
        WN_UNROLL
        for (int tap = 0; tap < KP; ++tap) {
            const int ts = t - (K - 1 - tap) * a.dil;
            const bool ok = (t < T) && ts >= 0;
            okh[tap] = ok;
            const int vt = ok ? (4 * hi * T + ts) * 4 : 0;  // dead lanes read a valid dummy address
            WN_UNROLL
            for (int s = 0; s < 32; ++s) xh[tap][s] = wn_buf_load(Xr, vt, kappa64(s, 0) * T4);
        }
    };
    constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};   // small terms first: h l, l h, h h
    f32x16 acc[4];
    // one tap: the 4 k-blocks of 16 channels against the tap's 128 weight rows
    auto tap_mfmas = [&](const float (&xr)[32], bool ok, int tapblk, float sc) {
        WN_UNROLL
        for (int kb = 0; kb < 4; ++kb) {
            float x8[8];
            WN_UNROLL
            for (int e = 0; e < 8; ++e) x8[e] = ok ? xr[8 * kb + e] : 0.0f;
            wn_f4 bf[2];
            split8h(x8, sc, bf);
            const char* Wl = Wd + (tapblk * 4 + kb) * WD_BLK + wn_frag_off(li, hi);
            WN_UNROLL
            for (int qh = 0; qh < 4; qh += 2) {  // two row tiles at a time (register budget)
                wn_f4 af[2][2];
                WN_UNROLL
                for (int q = 0; q < 2; ++q) {
                    WN_UNROLL
                    for (int p = 0; p < 2; ++p) af[q][p] = *reinterpret_cast<const wn_f4*>(Wl + p * (128 * 32) + (qh + q) * 1024);
                }
                WN_UNROLL
                for (int t3 = 0; t3 < 3; ++t3) {
                    acc[qh] = mfma_f16(af[0][PA[t3]], bf[PB[t3]], acc[qh]);
                    acc[qh + 1] = mfma_f16(af[1][PA[t3]], bf[PB[t3]], acc[qh + 1]);
                }
            }
        }
    };

    int tile_v = WN_UNIFORM(walk.first);
    int tcount = 0;
    (void)tcount;
    if (K > 1 && tile_v < tile_end) issue_hist(tile_v);
    while (tile_v < tile_end) {
        WN_STAMP(0);
        const int tile = WN_UNIFORM(tile_v);
        const int b = tile / tiles_per_b;
        const int t = (tile - b * tiles_per_b) * 32 + li;
        const bool inb = t < T;
        const int tc = inb ? t : T - 1;
        const int vcur = inb ? (4 * hi * T + t) * 4 : 0;

        float xc[32];
        {
            const wn_rsrc_t Xr = wn_make_buf(a.X + (long)b * 64 * T, slab);
            WN_UNROLL
            for (int tap = KP; tap + 1 < K; ++tap) {   // (K = 3) the history taps that are not requested a tile ahead
                const int ts = t - (K - 1 - tap) * a.dil;
                const bool ok = inb && ts >= 0;
                okh[tap] = ok;
                const int vt = ok ? (4 * hi * T + ts) * 4 : 0;
                WN_UNROLL
                for (int s = 0; s < 32; ++s) xh[tap][s] = wn_buf_load(Xr, vt, kappa64(s, 0) * T4);
            }
            WN_UNROLL
            for (int s = 0; s < 32; ++s) xc[s] = wn_buf_load(Xr, vcur, kappa64(s, 0) * T4);
        }
        WN_SCHED_BARRIER();
        WN_PRIO(WN_PRIO_MFMA);
        WN_UNROLL
        for (int q = 0; q < 4; ++q) acc[q] = f32x16_zero();
        // The tile's block scale: the smallest of its taps' scales so far; the accumulators follow it down (exact: powers of two).
        float sc = 0.0f;
        WN_UNROLL
        for (int tap = 0; tap + 1 < K; ++tap) {
            const float st = wn_pow2_scale(wave_amax32(xh[tap], okh[tap]), 14);
            if (tap == 0) {
                sc = st;
            } else {
                const float sn = fminf(sc, st);
                const float ratio = sn / sc;     // <= 1
                WN_UNROLL
                for (int q = 0; q < 4; ++q) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) acc[q][r] *= ratio;
                }
                sc = sn;
            }
            tap_mfmas(xh[tap], okh[tap], tap, sc);
        }
        WN_STAMP(1);  // after history-tap MFMAs
        // aux / gate inputs (frame rate, L2 resident), first 32 gate channels
        const int fr = tc / a.U;
        const float upw_j = a.upw[tc - fr * a.U];
        const wn_rsrc_t Gr = wn_make_buf(a.G + (long)b * a.g_bstride, (unsigned)(128 * F4));
        const int vg = (4 * hi * a.F + fr) * 4;
        float ga[16], gg[16];
        WN_UNROLL
        for (int r = 0; r < 16; ++r) {
            ga[r] = wn_buf_load(Gr, vg, mfma32_row(r, 0) * F4);
            gg[r] = wn_buf_load(Gr, vg, (mfma32_row(r, 0) + 64) * F4);
        }
        WN_SCHED_BARRIER();
        {   // current tap; xc is also the residual input, already in D layout
            const float st = wn_pow2_scale(wave_amax32(xc, inb), 14);
            if (K > 1) {
                const float sn = fminf(sc, st);
                const float ratio = sn / sc;
                WN_UNROLL
                for (int q = 0; q < 4; ++q) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) acc[q][r] *= ratio;
                }
                sc = sn;
            } else {
                sc = st;
            }
            tap_mfmas(xc, inb, K - 1, sc);
        }
        const float inv = inv_wd / sc;   // (both powers of two) accumulators -> true pre-activations
        WN_SCHED_BARRIER();
        WN_PRIO(WN_PRIO_GATE);
        WN_STAMP(2);  // after current-tap MFMAs
        f32x16 xb2[2];
        if (a.Xnext != nullptr) {
            const float* rbl = rb + 4 * hi;
            WN_UNROLL
            for (int q = 0; q < 2; ++q) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r)
                    xb2[q][r] = (inb ? xc[16 * q + r] : 0.0f) + rbl[32 * q + mfma32_row(r, 0)];
            }
#ifndef WN_EMU
            asm volatile("" : "+v"(xb2[0]), "+v"(xb2[1]));
#endif
        }
        const int next_v = tile_v + step;
        if (K > 1 && next_v < tile_end) issue_hist(next_v);
        // gate (reference wavenet.py:529-532): P = conv + w[j]*G[row][f] + c[row]; saved for backward
        const wn_rsrc_t Sr = wn_make_buf(a.S + (long)b * 64 * T, slab);
        const bool keep_g = a.Gt != nullptr;
        const wn_rsrc_t Gtr = wn_make_buf((keep_g ? a.Gt : a.S) + (long)b * 64 * T, slab);
        const wn_rsrc_t Zr = wn_make_buf(a.Z + (long)b * 64 * T, slab);
        const float* cvl = cv + 4 * hi;
        f32x16 z[2];
        const int vst = inb ? vcur : WN_VOFF_DEAD;
        auto gate_phase = [&](auto keep_tag) {
        constexpr bool KEEP_G = decltype(keep_tag)::value;
        WN_UNROLL
        for (int q = 0; q < 2; ++q) {
            WN_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row0 = 32 * q + mfma32_row(r, 0);  // + 4*hi is in the per-lane offsets
                const float pa = fmaf(acc[q][r], inv, upw_j * ga[r] + cvl[row0]);
                const float pg = fmaf(acc[q + 2][r], inv, upw_j * gg[r] + cvl[row0 + 64]);
                if (q == 0) {
                    ga[r] = wn_buf_load(Gr, vg, (32 + mfma32_row(r, 0)) * F4);
                    gg[r] = wn_buf_load(Gr, vg, (96 + mfma32_row(r, 0)) * F4);
                }
                const float s = wn_sigmoid(pa);
                const float g = wn_tanh(pg);
                const float zz = s * g;
                z[q][r] = zz;
                wn_buf_store(Sr, s, vst, row0 * T4);
                if (KEEP_G) wn_buf_store(Gtr, g, vst, row0 * T4);
                wn_buf_store(Zr, zz, vst, row0 * T4);
            }
        }
        };
        if (keep_g) gate_phase(std::true_type{});
        else gate_phase(std::false_type{});
        WN_STAMP(3);  // after gate math + S/Gt/Z stores issued
        WN_SCHED_BARRIER();
        WN_PRIO(WN_PRIO_MFMA);
        // res 1x1 + residual; z (in (-1, 1): scaled by 2^13) is consumed straight from the accumulator registers
        if (a.Xnext != nullptr) {
            f32x16 racc[2];
            racc[0] = f32x16_zero();
            racc[1] = f32x16_zero();
            const int vfrag = wn_frag_off(li, hi);
            WN_UNROLL
            for (int kb = 0; kb < 4; ++kb) {
                float x8[8];
                WN_UNROLL
                for (int e = 0; e < 8; ++e) x8[e] = z[(8 * kb + e) >> 4][(8 * kb + e) & 15];
                wn_f4 bf[2];
                split8h(x8, 8192.0f, bf);
                wn_f4 afr[2][2];
                WN_UNROLL
                for (int q = 0; q < 2; ++q) {
                    WN_UNROLL
                    for (int p = 0; p < 2; ++p) afr[q][p] = *reinterpret_cast<const wn_f4*>(Wr + kb * WR_BLK + vfrag + p * (64 * 32) + q * 1024);
                }
                WN_UNROLL
                for (int t3 = 0; t3 < 3; ++t3) {
                    racc[0] = mfma_f16(afr[0][PA[t3]], bf[PB[t3]], racc[0]);
                    racc[1] = mfma_f16(afr[1][PA[t3]], bf[PB[t3]], racc[1]);
                }
            }
            const float invr = inv_wr * (1.0f / 8192.0f);
            {
                const wn_rsrc_t Xn = wn_make_buf(a.Xnext + (long)b * 64 * T, slab);
                WN_UNROLL
                for (int q = 0; q < 2; ++q) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r)
                        wn_buf_store(Xn, fmaf(racc[q][r], invr, xb2[q][r]), vst, (32 * q + mfma32_row(r, 0)) * T4);
                }
            }
        }
        WN_PRIO(WN_PRIO_GATE);
        WN_STAMP(4);  // tile done
        ++tcount;
        tile_v = next_v;
    }
#ifdef WN_TIMING
    if (a.dbg && blockIdx.x == 0 && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        a.dbg[(wave * 4) * 16 + 8] = (long long)__builtin_amdgcn_s_memrealtime();
        a.dbg[(wave * 4) * 16 + 9] = (long long)__builtin_readcyclecounter();
    }
    if (a.dbg && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        if (wave == 0) a.dbg[512 + blockIdx.x * 4 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
        if (wave == 7) a.dbg[512 + blockIdx.x * 4 + 2] = (long long)__builtin_amdgcn_s_memrealtime();
        if (wave == 0) a.dbg[512 + blockIdx.x * 4 + 3] = tcount;
    }
#endif
}

template <int K>
static int launch_fwd(const FwdArgs& a, int split, wn_stream_t st) {
    const long ntiles = (long)a.B * ((a.T + 31) / 32);
    const long nblk = balanced_blocks(ntiles);
    if (split == 2) {   // fp16 pair split (needs the two-piece image of wn_fused_pack_images16)
        if (a.wimg == nullptr) return 1;
        const size_t lds_h = (size_t)fwd16_image_bytes(K) + 192 * sizeof(float);
        if (set_lds(k_resblock_fwd_h<K>, lds_h)) return 1;
        WN_LAUNCH((k_resblock_fwd_h<K>), dim3((unsigned)nblk), dim3(WN_FT), lds_h, st, a);
        return 0;
    }
    // split arithmetic: taps (+ res 1x1 for K <= 2; K = 3 reads those fragments from the global image) + cvec / bias
    const size_t lds_s = (size_t)K * 4 * (3 * 128 * 32) + (K >= 3 ? 0 : 4 * (3 * 64 * 32)) + 192 * sizeof(float);
    if (split && lds_s <= 160 * 1024 && (K < 3 || a.wimg != nullptr)) {  // (K = 3 without a weight image: the f32 MFMA kernel)
        if (set_lds(k_resblock_fwd_s<K>, lds_s)) return 1;
        WN_LAUNCH((k_resblock_fwd_s<K>), dim3((unsigned)nblk), dim3(WN_FT), lds_s, st, a);
        return 0;
    }
    const size_t lds = ((size_t)K * 64 * 128 + 64 * 64 + 192) * sizeof(float);
    if (set_lds(k_resblock_fwd<K>, lds)) return 1;
    WN_LAUNCH((k_resblock_fwd<K>), dim3((unsigned)nblk), dim3(WN_FT), lds, st, a);
    return 0;
}

int wn_fused_resblock_fwd(const float* wd_f, const float* wres_f, const float* cvec, const float* res_bias, const float* X,
                          const float* G, long g_bstride, const float* upw, float* Xnext, float* S, float* Gt, float* Z, int B,
                          int T, int K, int dilation, int U, int F, int split, const float* wimg, wn_stream_t st) {
    WN_PROF("fused_resblock_fwd", 2.0 * (double)B * T * (K * 64.0 * 128.0 + (Xnext ? 64.0 * 64.0 : 0.0)),
            4.0 * (double)B * T * 64.0 * ((Xnext ? 4.0 : 3.0) + (Gt ? 1.0 : 0.0)), st);  // X in; S, (Gt,) Z (, Xnext) out
    FwdArgs a;
    a.wimg = split ? wimg : nullptr;
    a.wd_f = wd_f; a.wres_f = wres_f; a.cvec = cvec; a.res_bias = res_bias;
    a.X = X; a.G = G; a.g_bstride = g_bstride; a.upw = upw;
    a.Xnext = Xnext; a.S = S; a.Gt = Gt; a.Z = Z;
    a.B = B; a.T = T; a.dil = dilation; a.U = U; a.F = F;
#ifdef WN_TIMING
    a.dbg = g_dbg;
#endif
    switch (K) {
        case 1: return launch_fwd<1>(a, split, st);
        case 2: return launch_fwd<2>(a, split, st);
        case 3: return launch_fwd<3>(a, split, st);
        default: return 1;
    }
}

// ---------------------------------------------------------------------------------------------
// conv64: out[64][t] = sum over segments, channels k:  W_seg[k][0..63] * src_seg[k][t - shift_seg]
// ---------------------------------------------------------------------------------------------
struct ConvSeg {
    const float* src;  // (B, nch, T)
    const float* w;    // [nch][64] (global); staged to LDS at float offset woff
    int nch;           // multiple of 32
    int shift;
    int woff;
};
struct ConvArgs {
    ConvSeg seg[3];
    int nseg;
    int nchunks;  // sum of nch/32
    int wfloats;  // total LDS floats
    int B, T;
    const float* S;      // MODE 0
    const float* Gt;     // MODE 0: the saved tanh half -- or, with gz != 0, the saved product z = s * tanh (g = z / s)
    int gz;
    const float* resid;  // MODE 1 (nullable)
    float* out;          // MODE 0: dP (B,128,T) ; MODE 1: dX (B,64,T)
    int interleave;      // k_conv64s: chunk q -> segment q % nseg (equal-size segments): the taps of one channel group back to back
    // MODE 2 (gate' + aux-gradient partials, see wn_fused.h)
    const float* G;      // (B, g_bstride) frame-rate aux projection of this layer, rows [0,128)
    long g_bstride;
    const float* upw;    // [U]
    int U, F;
    float* dGp;          // (B, 128, T/16)
    float* qp;           // (B, T)
};

template <int MODE>
__global__ __launch_bounds__(WN_LB) void k_conv64(ConvArgs a) {
    WN_DYN_SMEM(smem_raw);
    float* W = reinterpret_cast<float*>(smem_raw);
    for (int sg = 0; sg < a.nseg; ++sg) stage_copy(W + a.seg[sg].woff, a.seg[sg].w, a.seg[sg].nch * 64);
    __syncthreads();
    // De-phase the two waves that share a SIMD (waves w and w+4): started together they would run
    // their MFMA phases and their gate/store phases in lock step and leave the matrix pipe idle
    // during the latter; half a tile of head start makes one wave's VALU/VMEM phase coincide with
    // the other's MFMA phase.

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int T = a.T;
    const int T4 = T * 4;
    const int tiles_per_b = (T + 31) >> 5;
    const int ntiles = a.B * tiles_per_b;
    const TileWalk walk = tile_walk(ntiles, threadIdx.x >> 6);
    const int step = walk.step, tile_end = walk.end;
    const int NCH = a.nchunks;

    // Operand chunks (32 channels = 16 k-steps) are double buffered in registers: chunk q+2 is in
    // flight while the 32 MFMAs of chunks q and q+1 run.
    float xa[16], xb[16];
    bool oka = false, okb = false;
    auto locate = [&](int q, int& sg, int& c0) {
        sg = 0;
        c0 = q * 32;
        while (sg + 1 < a.nseg && c0 >= a.seg[sg].nch) {
            c0 -= a.seg[sg].nch;
            ++sg;
        }
    };
    auto issue = [&](int tl_v, int q, float (&xr)[16], bool& okr) {
        const int tl = WN_UNIFORM(tl_v);
        const int b = tl / tiles_per_b;
        const int t = (tl - b * tiles_per_b) * 32 + li;
        int sg, c0;
        locate(q, sg, c0);
        const ConvSeg& g = a.seg[sg];
        const int ts = t - g.shift;
        const bool ok = (t < T) && ts >= 0 && ts < T;
        okr = ok;
        const wn_rsrc_t Sr = wn_make_buf(g.src + (long)b * g.nch * T, (unsigned)(g.nch * T4));
        const int vt = ok ? (hi * T + ts) * 4 : 0;
        WN_UNROLL
        for (int s = 0; s < 16; ++s) xr[s] = wn_buf_load(Sr, vt, (c0 + 2 * s) * T4);
    };
    f32x16 acc[2];
    auto consume = [&](int q, const float (&xr)[16], bool okr) {
        int sg, c0;
        locate(q, sg, c0);
        const float* Wl = W + a.seg[sg].woff + (c0 + hi) * 64 + li;
        // stage = two k-steps (4 LDS operands, 4 MFMAs); ping-pong operand sets: the reads of the next
        // stage are in flight during the MFMAs of the current one
        float a0[4], a1[4];
        WN_UNROLL
        for (int q = 0; q < 4; ++q) a0[q] = Wl[(2 * (q >> 1)) * 64 + 32 * (q & 1)];
        WN_SGB_DS(2);
        WN_UNROLL
        for (int s = 0; s < 16; s += 4) {
            WN_UNROLL
            for (int q = 0; q < 4; ++q) a1[q] = Wl[(2 * (s + 2 + (q >> 1))) * 64 + 32 * (q & 1)];
            {
                const float xv0 = okr ? xr[s] : 0.0f, xv1 = okr ? xr[s + 1] : 0.0f;
                acc[0] = mfma32(a0[0], xv0, acc[0]);
                acc[1] = mfma32(a0[1], xv0, acc[1]);
                acc[0] = mfma32(a0[2], xv1, acc[0]);
                acc[1] = mfma32(a0[3], xv1, acc[1]);
            }
            WN_SGB_DS(2);
            WN_SGB_MFMA(4);
            if (s + 4 < 16) {
                WN_UNROLL
                for (int q = 0; q < 4; ++q) a0[q] = Wl[(2 * (s + 4 + (q >> 1))) * 64 + 32 * (q & 1)];
            }
            {
                const float xv0 = okr ? xr[s + 2] : 0.0f, xv1 = okr ? xr[s + 3] : 0.0f;
                acc[0] = mfma32(a1[0], xv0, acc[0]);
                acc[1] = mfma32(a1[1], xv0, acc[1]);
                acc[0] = mfma32(a1[2], xv1, acc[0]);
                acc[1] = mfma32(a1[3], xv1, acc[1]);
            }
            WN_SGB_DS(2);
            WN_SGB_MFMA(4);
        }
    };

    int tile_v = walk.first;
    if (tile_v < tile_end) {
        issue(tile_v, 0, xa, oka);
        if (NCH > 1) issue(tile_v, 1, xb, okb);
    }
    while (tile_v < tile_end) {
        const int tile = WN_UNIFORM(tile_v);
        const int b = tile / tiles_per_b;
        const int t = (tile - b * tiles_per_b) * 32 + li;
        const bool inb = t < T;
        const int vcur = inb ? (4 * hi * T + t) * 4 : 0;
        const int vst = inb ? vcur : WN_VOFF_DEAD;   // stores of lanes past T are dropped by the range check
        const int next_v = tile_v + step;

        // epilogue inputs: issued now, consumed after all MFMAs of the tile
        float e0[2][16], e1[2][16];
        if (MODE == 0) {
            const wn_rsrc_t Sr = wn_make_buf(a.S + (long)b * 64 * T, (unsigned)(64 * T4));
            const wn_rsrc_t Gr = wn_make_buf(a.Gt + (long)b * 64 * T, (unsigned)(64 * T4));
            WN_UNROLL
            for (int q = 0; q < 2; ++q) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int so = (32 * q + mfma32_row(r, 0)) * T4;
                    e0[q][r] = wn_buf_load(Sr, vcur, so);
                    e1[q][r] = wn_buf_load(Gr, vcur, so);
                }
            }
        } else if (a.resid != nullptr) {
            const wn_rsrc_t Rr = wn_make_buf(a.resid + (long)b * 64 * T, (unsigned)(64 * T4));
            WN_UNROLL
            for (int q = 0; q < 2; ++q) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) e0[q][r] = wn_buf_load(Rr, vcur, (32 * q + mfma32_row(r, 0)) * T4);
            }
        }
        WN_SCHED_BARRIER();
        acc[0] = f32x16_zero();
        acc[1] = f32x16_zero();
        for (int q = 0; q < NCH; q += 2) {
            consume(q, xa, oka);
            if (q + 2 < NCH) issue(tile_v, q + 2, xa, oka);
            else if (next_v < tile_end) issue(next_v, 0, xa, oka);
            WN_SCHED_BARRIER();
            if (q + 1 < NCH) {
                consume(q + 1, xb, okb);
                if (q + 3 < NCH) issue(tile_v, q + 3, xb, okb);
                else if (next_v < tile_end && NCH > 1) issue(next_v, 1, xb, okb);
                WN_SCHED_BARRIER();
            }
        }
        {
            if (MODE == 0) {
                // gate backward: dP = [dZ*g*s*(1-s) ; dZ*s*(1-g^2)]
                const wn_rsrc_t Or = wn_make_buf(a.out + (long)b * 128 * T, (unsigned)(128 * T4));
                WN_UNROLL
                for (int q = 0; q < 2; ++q) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) {
                        const int so = (32 * q + mfma32_row(r, 0)) * T4;
                        const float s = e0[q][r], g = a.gz ? e1[q][r] * wn_rcp(e0[q][r]) : e1[q][r], dz = acc[q][r];
                        wn_buf_store(Or, dz * g * (s * (1.0f - s)), vst, so);
                        wn_buf_store(Or, dz * s * (1.0f - g * g), vst, so + 64 * T4);
                    }
                }
            } else {
                const wn_rsrc_t Or = wn_make_buf(a.out + (long)b * 64 * T, (unsigned)(64 * T4));
                WN_UNROLL
                for (int q = 0; q < 2; ++q) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[q][r];
                        if (a.resid != nullptr) v += e0[q][r];
                        wn_buf_store(Or, v, vst, (32 * q + mfma32_row(r, 0)) * T4);
                    }
                }
            }
        }
        tile_v = next_v;
    }
}

// Same tile loop on the bf16 matrix cores: every fp32 operand is split into three bf16 pieces and the
// six significant cross products are accumulated in fp32 (see wn_gemm6.hip for the error analysis).
// Weights are split once per launch into LDS in fragment layout [16-k block][piece][64 rows][16 k];
// the activation chunk of a lane (2 x 8 consecutive channels of its time step) is split in registers.
// 24 bf16 MFMAs (768 cycles) replace the 32 f32 MFMAs (2048 cycles) of a 32-channel chunk.
template <int MODE>
__global__ __launch_bounds__(WN_LB) void k_conv64s(ConvArgs a) {
    WN_DYN_SMEM(smem_raw);
    char* W = smem_raw;  // 6 KB per 16-k block: [piece][row][16 k] bf16
    for (int idx = threadIdx.x; idx < a.nchunks * 2 * 128; idx += WN_FT) {
        const int o = idx & 63, h = (idx >> 6) & 1, kbg = idx >> 7;
        int sg = 0, c = kbg * 16 + 8 * h;
        if (a.interleave) {
            const int q = kbg >> 1;
            sg = q % a.nseg;
            c = (q / a.nseg) * 32 + (kbg & 1) * 16 + 8 * h;
        } else {
            while (sg + 1 < a.nseg && c >= a.seg[sg].nch) {
                c -= a.seg[sg].nch;
                ++sg;
            }
        }
        const float* src = a.seg[sg].w + (long)c * 64 + o;
        unsigned hq[4], mq[4], lq[4];
        WN_UNROLL
        for (int q = 0; q < 4; ++q) {
            const float x0 = src[(2 * q) * 64], x1 = src[(2 * q + 1) * 64];
            hq[q] = wn_pk_bf16(x0, x1);
            const float r0 = x0 - wn_bits_f32(hq[q] << 16), r1 = x1 - wn_bits_f32(hq[q] & 0xffff0000u);
            mq[q] = wn_pk_bf16(r0, r1);
            lq[q] = wn_pk_bf16(r0 - wn_bits_f32(mq[q] << 16), r1 - wn_bits_f32(mq[q] & 0xffff0000u));
        }
        unsigned* d = reinterpret_cast<unsigned*>(W + kbg * 6144 + wn_frag_off(o, h));
        WN_UNROLL
        for (int q = 0; q < 4; ++q) {
            d[q] = hq[q];
            d[512 + q] = mq[q];
            d[1024 + q] = lq[q];
        }
    }
    __syncthreads();
    // De-phase the two waves that share a SIMD (waves w and w+4): started together they would run
    // their MFMA phases and their gate/store phases in lock step and leave the matrix pipe idle
    // during the latter; half a tile of head start makes one wave's VALU/VMEM phase coincide with
    // the other's MFMA phase.

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int T = a.T;
    const int T4 = T * 4;
    const int tiles_per_b = (T + 31) >> 5;
    const int ntiles = a.B * tiles_per_b;
    const TileWalk walk = tile_walk(ntiles, threadIdx.x >> 6);
    const int step = walk.step, tile_end = walk.end;
    const int NCH = a.nchunks;

    // Operand chunks (32 channels = 16 k-steps) are double buffered in registers: chunk q+2 is in
    // flight while the 32 MFMAs of chunks q and q+1 run.
    float xa[16], xb[16];
    bool oka = false, okb = false;
    auto locate = [&](int q, int& sg, int& c0) {
        if (a.interleave) {
            sg = q % a.nseg;
            c0 = (q / a.nseg) * 32;
            return;
        }
        sg = 0;
        c0 = q * 32;
        while (sg + 1 < a.nseg && c0 >= a.seg[sg].nch) {
            c0 -= a.seg[sg].nch;
            ++sg;
        }
    };
    auto issue = [&](int tl_v, int q, float (&xr)[16], bool& okr) {
        const int tl = WN_UNIFORM(tl_v);
        const int b = tl / tiles_per_b;
        const int t = (tl - b * tiles_per_b) * 32 + li;
        int sg, c0;
        locate(q, sg, c0);
        const ConvSeg& g = a.seg[sg];
        const int ts = t - g.shift;
        const bool ok = (t < T) && ts >= 0 && ts < T;
        okr = ok;
        const wn_rsrc_t Sr = wn_make_buf(g.src + (long)b * g.nch * T, (unsigned)(g.nch * T4));
        const int vt = ok ? (8 * hi * T + ts) * 4 : 0;  // lane half hi owns channels 8*hi .. 8*hi+7 of each 16-k block
        WN_UNROLL
        for (int s = 0; s < 16; ++s) xr[s] = wn_buf_load(Sr, vt, (c0 + 16 * (s >> 3) + (s & 7)) * T4);
    };
    f32x16 acc[2];
    auto consume = [&](int q, const float (&xr)[16], bool okr) {
        WN_UNROLL
        for (int blk = 0; blk < 2; ++blk) {
            unsigned hq[4], mq[4], lq[4];
            WN_UNROLL
            for (int e = 0; e < 4; ++e) {
                const float x0 = okr ? xr[8 * blk + 2 * e] : 0.0f, x1 = okr ? xr[8 * blk + 2 * e + 1] : 0.0f;
                hq[e] = wn_pk_bf16(x0, x1);
                const float r0 = x0 - wn_bits_f32(hq[e] << 16), r1 = x1 - wn_bits_f32(hq[e] & 0xffff0000u);
                mq[e] = wn_pk_bf16(r0, r1);
                lq[e] = wn_pk_bf16(r0 - wn_bits_f32(mq[e] << 16), r1 - wn_bits_f32(mq[e] & 0xffff0000u));
            }
            wn_f4 bf[3];
            bf[0].x = wn_bits_f32(hq[0]); bf[0].y = wn_bits_f32(hq[1]); bf[0].z = wn_bits_f32(hq[2]); bf[0].w = wn_bits_f32(hq[3]);
            bf[1].x = wn_bits_f32(mq[0]); bf[1].y = wn_bits_f32(mq[1]); bf[1].z = wn_bits_f32(mq[2]); bf[1].w = wn_bits_f32(mq[3]);
            bf[2].x = wn_bits_f32(lq[0]); bf[2].y = wn_bits_f32(lq[1]); bf[2].z = wn_bits_f32(lq[2]); bf[2].w = wn_bits_f32(lq[3]);
            const char* Wl = W + (2 * q + blk) * 6144 + wn_frag_off(li, hi);
            wn_f4 af[2][3];
            WN_UNROLL
            for (int rt = 0; rt < 2; ++rt) {
                WN_UNROLL
                for (int p = 0; p < 3; ++p) af[rt][p] = *reinterpret_cast<const wn_f4*>(Wl + p * 2048 + rt * 1024);
            }
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};  // small terms first
            WN_UNROLL
            for (int t = 0; t < 6; ++t) {
                acc[0] = mfma_bf16(af[0][PA[t]], bf[PB[t]], acc[0]);
                acc[1] = mfma_bf16(af[1][PA[t]], bf[PB[t]], acc[1]);
            }
        }
    };

    int tile_v = walk.first;
    if (tile_v < tile_end) {
        issue(tile_v, 0, xa, oka);
        if (NCH > 1) issue(tile_v, 1, xb, okb);
    }
    while (tile_v < tile_end) {
        const int tile = WN_UNIFORM(tile_v);
        const int b = tile / tiles_per_b;
        const int t = (tile - b * tiles_per_b) * 32 + li;
        const bool inb = t < T;
        const int vcur = inb ? (4 * hi * T + t) * 4 : 0;
        const int vst = inb ? vcur : WN_VOFF_DEAD;   // stores of lanes past T are dropped by the range check
        const int next_v = tile_v + step;

        // epilogue inputs: issued now, consumed after all MFMAs of the tile
        float e0[2][16], e1[2][16];
        if (MODE != 1) {
            const wn_rsrc_t Sr = wn_make_buf(a.S + (long)b * 64 * T, (unsigned)(64 * T4));
            const wn_rsrc_t Gr = wn_make_buf(a.Gt + (long)b * 64 * T, (unsigned)(64 * T4));
            WN_UNROLL
            for (int q = 0; q < 2; ++q) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int so = (32 * q + mfma32_row(r, 0)) * T4;
                    e0[q][r] = wn_buf_load(Sr, vcur, so);
                    e1[q][r] = wn_buf_load(Gr, vcur, so);
                }
            }
        } else if (a.resid != nullptr) {
            const wn_rsrc_t Rr = wn_make_buf(a.resid + (long)b * 64 * T, (unsigned)(64 * T4));
            WN_UNROLL
            for (int q = 0; q < 2; ++q) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) e0[q][r] = wn_buf_load(Rr, vcur, (32 * q + mfma32_row(r, 0)) * T4);
            }
        }
        WN_SCHED_BARRIER();
        acc[0] = f32x16_zero();
        acc[1] = f32x16_zero();
        for (int q = 0; q < NCH; q += 2) {
            consume(q, xa, oka);
            if (q + 2 < NCH) issue(tile_v, q + 2, xa, oka);
            else if (next_v < tile_end) issue(next_v, 0, xa, oka);
            WN_SCHED_BARRIER();
            if (q + 1 < NCH) {
                consume(q + 1, xb, okb);
                if (q + 3 < NCH) issue(tile_v, q + 3, xb, okb);
                else if (next_v < tile_end && NCH > 1) issue(next_v, 1, xb, okb);
                WN_SCHED_BARRIER();
            }
        }
        if (MODE == 2) {
            // gate backward + the partial sums of the aux gradient (wn_fused.h): every lane takes part in the
            // cross-lane sums, lanes past T carry dP = 0 (their operands were zero).
            const int tc = inb ? t : 0;
            const int fr = tc / a.U;
            const float wj = inb ? a.upw[tc - fr * a.U] : 0.0f;
            const int F4 = a.F * 4;
            const int H = T >> 4;  // 16-sample groups per channel row
            const wn_rsrc_t Or = wn_make_buf(a.out + (long)b * 128 * T, (unsigned)(128 * T4));
            const wn_rsrc_t Gr = wn_make_buf(a.G + (long)b * a.g_bstride, (unsigned)(128 * F4));
            const wn_rsrc_t Dr = wn_make_buf(a.dGp + (long)b * 128 * H, (unsigned)(128 * H * 4));
            const int vg = (4 * hi * a.F + fr) * 4;
            const int l16 = li & 15;
            const int h16 = (((tile - b * tiles_per_b) * 32) >> 4) + (li >> 4);
            float qsum = 0.0f;
            WN_UNROLL
            for (int q = 0; q < 2; ++q) {
                float ga[16], gg[16];
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    ga[r] = wn_buf_load(Gr, vg, (32 * q + mfma32_row(r, 0)) * F4);
                    gg[r] = wn_buf_load(Gr, vg, (64 + 32 * q + mfma32_row(r, 0)) * F4);
                }
                float keep_a = 0.0f, keep_g = 0.0f;
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int so = (32 * q + mfma32_row(r, 0)) * T4;
                    const float s = e0[q][r], g = a.gz ? e1[q][r] * wn_rcp(e0[q][r]) : e1[q][r], dz = inb ? acc[q][r] : 0.0f;
                    const float dpa = dz * g * (s * (1.0f - s)), dpg = dz * s * (1.0f - g * g);
                    wn_buf_store(Or, dpa, vst, so);   // lanes past T: out-of-range offset, dropped (no branch per element)
                    wn_buf_store(Or, dpg, vst, so + 64 * T4);
                    qsum += dpa * ga[r] + dpg * gg[r];
                    const float ra = wn_row16_sum(wj * dpa), rg = wn_row16_sum(wj * dpg);
                    keep_a = (l16 == r) ? ra : keep_a;
                    keep_g = (l16 == r) ? rg : keep_g;
                }
                // lane l16 of each 16-lane group keeps the sums of channel row 32q + mfma32_row(l16, hi)
                const int row = 32 * q + mfma32_row(l16, hi);
                const int doff = (h16 < H) ? (row * H + h16) * 4 : 0x7ffffff0;
                wn_buf_store(Dr, keep_a, doff, 0);
                wn_buf_store(Dr, keep_g, doff, (h16 < H) ? 64 * H * 4 : 0);
            }
            qsum += __shfl_xor(qsum, 32, 64);  // the two lane halves hold complementary channel rows
            if (hi == 0 && inb) a.qp[(long)b * T + t] = qsum;
        } else {
            if (MODE == 0) {
                // gate backward: dP = [dZ*g*s*(1-s) ; dZ*s*(1-g^2)]
                const wn_rsrc_t Or = wn_make_buf(a.out + (long)b * 128 * T, (unsigned)(128 * T4));
                WN_UNROLL
                for (int q = 0; q < 2; ++q) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) {
                        const int so = (32 * q + mfma32_row(r, 0)) * T4;
                        const float s = e0[q][r], g = a.gz ? e1[q][r] * wn_rcp(e0[q][r]) : e1[q][r], dz = acc[q][r];
                        wn_buf_store(Or, dz * g * (s * (1.0f - s)), vst, so);
                        wn_buf_store(Or, dz * s * (1.0f - g * g), vst, so + 64 * T4);
                    }
                }
            } else {
                const wn_rsrc_t Or = wn_make_buf(a.out + (long)b * 64 * T, (unsigned)(64 * T4));
                WN_UNROLL
                for (int q = 0; q < 2; ++q) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[q][r];
                        if (a.resid != nullptr) v += e0[q][r];
                        wn_buf_store(Or, v, vst, (32 * q + mfma32_row(r, 0)) * T4);
                    }
                }
            }
        }
        tile_v = next_v;
    }
}

template <int MODE>
static int launch_conv64(const ConvArgs& a, int split, wn_stream_t st) {
    const long ntiles = (long)a.B * ((a.T + 31) / 32);
    const long nblk = balanced_blocks(ntiles);
    if (split) {
        const size_t lds = (size_t)a.nchunks * 2 * 6144;
        if (lds > 160 * 1024 || set_lds(k_conv64s<MODE>, lds)) return 1;
        WN_LAUNCH((k_conv64s<MODE>), dim3((unsigned)nblk), dim3(WN_FT), lds, st, a);
        return 0;
    }
    if constexpr (MODE != 2) {  // the aux-partial epilogue exists in the split kernel only
        const size_t lds = (size_t)a.wfloats * sizeof(float);
        if (set_lds(k_conv64<MODE>, lds)) return 1;
        WN_LAUNCH((k_conv64<MODE>), dim3((unsigned)nblk), dim3(WN_FT), lds, st, a);
        return 0;
    }
    return 1;
}

int wn_fused_bwd_gate(const float* wskip, const float* wres, const float* dSk, const float* dXn, const float* S, const float* Gt,
                      int gt_is_z, float* dP, int B, int T, int Sch, int split, wn_stream_t st) {
    WN_PROF("fused_bwd_gate", 2.0 * (double)B * T * 64.0 * (Sch + (dXn ? 64.0 : 0.0)),
            4.0 * (double)B * T * (Sch + (dXn ? 64.0 : 0.0) + 4.0 * 64.0), st);  // dSk (, dXn), S, Gt in; dP out
    ConvArgs a;
    a.nseg = 1;
    a.seg[0].src = dSk; a.seg[0].w = wskip; a.seg[0].nch = Sch; a.seg[0].shift = 0; a.seg[0].woff = 0;
    a.wfloats = Sch * 64;
    a.nchunks = Sch / 32;
    if (dXn) {
        a.seg[1].src = dXn; a.seg[1].w = wres; a.seg[1].nch = 64; a.seg[1].shift = 0; a.seg[1].woff = Sch * 64;
        a.nseg = 2;
        a.wfloats += 64 * 64;
        a.nchunks += 2;
    }
    a.B = B; a.T = T; a.S = S; a.Gt = Gt; a.gz = gt_is_z; a.resid = nullptr; a.out = dP;
    a.interleave = 0;
    a.G = nullptr; a.g_bstride = 0; a.upw = nullptr; a.U = 0; a.F = 0; a.dGp = nullptr; a.qp = nullptr;
    return launch_conv64<0>(a, split, st);
}

int wn_fused_bwd_gate_aux(const float* wskip, const float* wres, const float* dSk, const float* dXn, const float* S,
                          const float* Gt, int gt_is_z, float* dP, const float* G, long g_bstride, const float* upw, int U, int F,
                          float* dGp, float* qp, int B, int T, int Sch, wn_stream_t st) {
    WN_PROF("fused_bwd_gate", 2.0 * (double)B * T * 64.0 * (Sch + (dXn ? 64.0 : 0.0)),
            4.0 * (double)B * T * (Sch + (dXn ? 64.0 : 0.0) + 4.0 * 64.0 + 9.0), st);  // + dGp (8/timestep) and qp (1) out
    if (U < 16 || (U & 15) || (T & 15) || (long)U * F != T) return 1;
    ConvArgs a;
    a.nseg = 1;
    a.seg[0].src = dSk; a.seg[0].w = wskip; a.seg[0].nch = Sch; a.seg[0].shift = 0; a.seg[0].woff = 0;
    a.wfloats = Sch * 64;
    a.nchunks = Sch / 32;
    if (dXn) {
        a.seg[1].src = dXn; a.seg[1].w = wres; a.seg[1].nch = 64; a.seg[1].shift = 0; a.seg[1].woff = Sch * 64;
        a.nseg = 2;
        a.wfloats += 64 * 64;
        a.nchunks += 2;
    }
    a.B = B; a.T = T; a.S = S; a.Gt = Gt; a.gz = gt_is_z; a.resid = nullptr; a.out = dP;
    a.interleave = 0;
    a.G = G; a.g_bstride = g_bstride; a.upw = upw; a.U = U; a.F = F; a.dGp = dGp; a.qp = qp;
    return launch_conv64<2>(a, 1, st);
}

int wn_fused_bwd_dx(const float* wd_b, const float* dP, const float* dXn, float* dX, int B, int T, int K, int dilation,
                    int split, wn_stream_t st) {
    WN_PROF("fused_bwd_dx", 2.0 * (double)B * T * 64.0 * K * 128.0,
            4.0 * (double)B * T * 64.0 * (dXn ? 4.0 : 3.0), st);  // dP (, dXn) in; dX out
    if (K > 3) return 1;
    ConvArgs a;
    a.nseg = K;
    for (int tap = 0; tap < K; ++tap) {
        a.seg[tap].src = dP;
        a.seg[tap].w = wd_b + (long)tap * 128 * 64;
        a.seg[tap].nch = 128;
        a.seg[tap].shift = -(K - 1 - tap) * dilation;
        a.seg[tap].woff = tap * 128 * 64;
    }
    a.wfloats = K * 128 * 64;
    a.nchunks = K * 4;
    a.B = B; a.T = T; a.S = nullptr; a.Gt = nullptr; a.gz = 0; a.resid = dXn; a.out = dX;
    // The taps of one 32-channel group are consumed back to back: position p is read as tap K-1 by the wave of its
    // own tile and as an earlier tap by the wave d samples away, and with [tap][channel] order those two reads of the
    // same lines were half a tile period apart -- longer than a line survives in the XCD's 4 MB L2 under the
    // kernel's write stream, so the second tap came over the fabric again (PMC: 253 MB per launch for 189 compulsory).
    // measured: 56.3 -> 53.0 us per launch (profiles/r01/tap_probe.txt)
    a.interleave = (K > 1) ? 1 : 0;
    a.G = nullptr; a.g_bstride = 0; a.upw = nullptr; a.U = 0; a.F = 0; a.dGp = nullptr; a.qp = nullptr;
    return launch_conv64<1>(a, split, st);
}

// ---------------------------------------------------------------------------------------------
// k_chain64s -- one launch per layer of the backward data chain (split arithmetic, K <= 2):
//     dX_l[t]     = dX_{l+1}[t] + sum_tap Wd_tap^T dP_l[t + (K-1-tap) d]                    (wavenet.py:527-528 reversed)
//     dZ_{l-1}[t] = dZs_{l-1}[t] + Wres_{l-1}^T dX_l[t]                                      (wavenet.py:533-535 reversed)
//     dP_{l-1}[t] = [dZ g s (1-s) ; dZ s (1-g^2)]                                            (wavenet.py:529-532 reversed)
// dX_l and gate'_{l-1} meet at the SAME time index, so the tile of dX_l a wave has just accumulated goes from its
// accumulator registers straight into the MFMAs of the res-1x1 transpose -- the chaining the forward kernel uses for
// z -> res 1x1 (k-step order = accumulator register order) -- and never comes back from HBM: one launch and 64 words
// per timestep less than the k_conv64s<0/2> + k_conv64s<1> pair.  What makes the weights of both halves fit the LDS in
// split form is that the skip part of dZ arrives pre-contracted: dZs = Wskip_l^T dSkip for ALL layers is one matrix-bound
// contraction per step (wn_api.hip), so this kernel reads 64 channels of it instead of the 256 of dSkip and needs no
// Wskip: taps K x 128 x 64 + Wres^T 64 x 64 = 120 KB of bf16 fragments for K = 2.
// AUX = 1 adds the aux-gradient partial sums of wn_fused_bwd_gate_aux to the gate epilogue (same arithmetic, same order).
// ---------------------------------------------------------------------------------------------
struct ChainArgs {
    const float* img_taps;  // pre-built LDS images of the tap weights (layer l) and of Wres^T (layer l-1), or NULL
    const float* img_res;
    const float* wd_b;   // [tap][128][64]: W_tap^T, row = dP channel, col = dX channel   (layer l)
    const float* dP;     // (B, 128, T) of layer l
    const float* dXn;    // (B, 64, T) dX_{l+1}; NULL for the last layer (its residual output is dead)
    float* dX;           // (B, 64, T) out
    const float* wres;   // natural res_1x1 weight of layer l-1: [o][i] = [k][row]
    const float* dZs;    // skip part of dZ_{l-1}: rows [0, 64) at dZs + b * zs_bstride, row stride T
    long zs_bstride;
    int zs_t0;           // dZs is zero (and was never written) in front of this position: the loss window of wn_backward_window
    const float* S;      // (B, 64, T) sigmoid / tanh halves of layer l-1 (saved by the forward)
    const float* Gt;     // with gz != 0: the saved product z = s * tanh instead of the tanh half (g = z / s)
    int gz;
    float* dPm;          // (B, 128, T) out: dP_{l-1}
    int B, T, K, dil;
    // AUX
    const float* G;      // (B, g_bstride) frame-rate aux projection of layer l-1, rows [0,128)
    long g_bstride;
    const float* upw;    // [U]
    int U, F;
    float* dGp;          // (B, 128, T/16)
    float* qp;           // (B, T)
    // H16 (fp16 pair split, block-scaled): img_taps / img_res are the two-piece images of wn_fused_pack_images16
    const float* amaxP;  // (B, tiles) max |dP_l| of every 32-sample tile, written by the launch that produced dP_l
    float* amaxPm;       // (B, tiles) out: max |dP_{l-1}| per tile (nullable)
};

// 8 fp32 values -> the three bf16 pieces of the lane's share of a 16-k block
static __device__ __forceinline__ void split8v(const float (&x)[8], bool ok, wn_f4 (&bf)[3]) {
    float y[8];
    WN_UNROLL
    for (int e = 0; e < 8; ++e) y[e] = ok ? x[e] : 0.0f;
    split8(y, bf);
}

// HEAD = true: the top of the chain.  The last layer's residual output is dead (wavenet.py:231-238), so dP_{L-1} is the gate'
// epilogue alone on dZs_{L-1} (which bwd_dz_skip_all now produces for ALL layers): no taps, no Wres^T, no dX -- its own
// instantiation, so that the main one compiles exactly as before.
// H16 (round 6, WN_FLAG_CHAIN_F16PAIR): the fp16 pair split, block-scaled like k_resblock_fwd_h -- two fp16 pieces per operand, three
// products; the weight images by the power of two of their own maximum (wn_fused_pack_images16); the dP operand of a tile by the
// power of two that puts the maximum of the producer tiles it reads at 2^14 (every launch leaves max |dP| per 32-sample tile beside
// the tensor it writes: `amaxPm`; the consumer looks up the tiles its shifted taps cover: `amaxP`), the dX operand of the res-1x1
// transpose by its own wave-wide maximum.  Nothing can leave fp16's range, whatever the size of the gradients.  Images: 64 + 16 KB
// for K = 2, 96 + 16 KB for K = 3 (the Wres^T fragments are back in the LDS).
template <int AUX, int K, bool HEAD = false, bool H16 = false>
__global__ __launch_bounds__(WN_LB) void k_chain64s(ChainArgs a) {
    WN_DYN_SMEM(smem_raw);
    char* W = smem_raw;                      // tap blocks: chunk q (32 channels) = tap q % K, channel group q / K; 2 blocks of 6 KB each
    constexpr int NCH = K * 4;               // chunks of the dX part
    constexpr int BLKB = H16 ? 4096 : 6144;  // bytes of one [piece][64 rows][16 k] block
    constexpr int PCB = 2048;                // ... of one piece of it
    char* Wr = W + NCH * 2 * BLKB;           // Wres^T: 4 blocks [piece][64 rows][16 k], k order = accumulator register order
    // K = 3: the taps alone fill the LDS (144 KB); the Wres^T fragments (24 KB, L2 resident) are read from the pre-split image
    // in global memory, one 16-k block ahead of their MFMAs (the launcher only takes K = 3 with images)
    constexpr bool RG = (K >= 3) && !H16;
    float inv_wd = 1.0f, inv_wr = 1.0f;
    (void)inv_wd; (void)inv_wr; (void)PCB;
    if (HEAD) {
        // no weights
    } else if (H16) {
        copy_image_to_lds(W, a.img_taps, NCH * 2 * BLKB);
        copy_image_to_lds(Wr, a.img_res, 4 * BLKB);
        inv_wd = a.img_taps[NCH * 2 * BLKB / 4];
        inv_wr = a.img_res[4 * BLKB / 4];
        WN_WAIT_VMCNT(0);
    } else if (RG) {
        copy_image_to_lds(W, a.img_taps, chain_taps_bytes(K));
        WN_WAIT_VMCNT(0);
    } else if (a.img_taps != nullptr) {   // pre-split once per step (wn_fused_pack_images): two straight global -> LDS copies
        copy_image_to_lds(W, a.img_taps, chain_taps_bytes(K));
        copy_image_to_lds(Wr, a.img_res, WN_RES_T_BYTES);
        WN_WAIT_VMCNT(0);
    } else {
        fill_chain_taps(W, a.wd_b, K, threadIdx.x, WN_FT);
        fill_res_t(Wr, a.wres, threadIdx.x, WN_FT);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int li = lane & 31, hi = lane >> 5;
    const int T = a.T;
    const int T4 = T * 4;
    const int tiles_per_b = (T + 31) >> 5;
    const int ntiles = a.B * tiles_per_b;
    const TileWalk walk = tile_walk(ntiles, threadIdx.x >> 6);
    const int step = walk.step, tile_end = walk.end;

    float xa[16], xb[16];
    bool oka = false, okb = false;
    auto issue = [&](int tl_v, int q, float (&xr)[16], bool& okr) {
        const int tl = WN_UNIFORM(tl_v);
        const int b = tl / tiles_per_b;
        const int t = (tl - b * tiles_per_b) * 32 + li;
        const int tap = q % K, c0 = (q / K) * 32;
        const int ts = t + (K - 1 - tap) * a.dil;
        const bool ok = (t < T) && ts < T;
        okr = ok;
        const wn_rsrc_t Sr = wn_make_buf(a.dP + (long)b * 128 * T, (unsigned)(128 * T4));
        const int vt = ok ? (8 * hi * T + ts) * 4 : 0;
        WN_UNROLL
        for (int s = 0; s < 16; ++s) xr[s] = wn_buf_load(Sr, vt, (c0 + 16 * (s >> 3) + (s & 7)) * T4);
    };
    f32x16 acc[2];
    float s_dp = 1.0f;   // H16: block scale of this tile's dP operand
    auto consume = [&](int q, const float (&xr)[16], bool okr) {
        WN_UNROLL
        for (int blk = 0; blk < 2; ++blk) {
            float x8[8];
            WN_UNROLL
            for (int e = 0; e < 8; ++e) x8[e] = H16 ? (okr ? xr[8 * blk + e] : 0.0f) : xr[8 * blk + e];
            const char* Wl = W + (2 * q + blk) * BLKB + wn_frag_off(li, hi);
            if constexpr (H16) {
                wn_f4 bf[2];
                split8h(x8, s_dp, bf);
                wn_f4 af[2][2];
                WN_UNROLL
                for (int rt = 0; rt < 2; ++rt) {
                    WN_UNROLL
                    for (int p = 0; p < 2; ++p) af[rt][p] = *reinterpret_cast<const wn_f4*>(Wl + p * PCB + rt * 1024);
                }
                constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};   // small terms first: h l, l h, h h
                WN_UNROLL
                for (int t3 = 0; t3 < 3; ++t3) {
                    acc[0] = mfma_f16(af[0][PA[t3]], bf[PB[t3]], acc[0]);
                    acc[1] = mfma_f16(af[1][PA[t3]], bf[PB[t3]], acc[1]);
                }
            } else {
                wn_f4 bf[3];
                split8v(x8, okr, bf);
                wn_f4 af[2][3];
                WN_UNROLL
                for (int rt = 0; rt < 2; ++rt) {
                    WN_UNROLL
                    for (int p = 0; p < 3; ++p) af[rt][p] = *reinterpret_cast<const wn_f4*>(Wl + p * 2048 + rt * 1024);
                }
                constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};  // small terms first
                WN_UNROLL
                for (int t6 = 0; t6 < WN_EXP_NPROD; ++t6) {
                    acc[0] = mfma_bf16(af[0][PA[t6]], bf[PB[t6]], acc[0]);
                    acc[1] = mfma_bf16(af[1][PA[t6]], bf[PB[t6]], acc[1]);
                }
            }
        }
    };

    // H16: block scale of a tile's dP operand = the largest |dP_l| over the producer tiles its taps read (shift (K - 1 - tap) d
    // ahead: at most two tiles each), as the launch that wrote dP_l recorded them.  Looked up together with the tile's first
    // operand chunks, i.e. a tile ahead: the loads are off the critical path of the tile's first MFMA.
    auto tile_amax = [&](int tl_v) -> float {
        const int tl = WN_UNIFORM(tl_v);
        const int b = tl / tiles_per_b, tb = tl - b * tiles_per_b;
        const float* am = a.amaxP + (long)b * tiles_per_b;
        float m = 0.0f;
        WN_UNROLL
        for (int tap = 0; tap < K; ++tap) {
            const int sh = (K - 1 - tap) * a.dil;
            const int t_lo = tb * 32 + sh, t_hi = t_lo + 31;
            const int i0 = t_lo >> 5, i1 = t_hi >> 5;
            if (i0 < tiles_per_b) m = fmaxf(m, am[i0]);
            if (i1 < tiles_per_b && i1 != i0) m = fmaxf(m, am[i1]);
        }
        return m;
    };
    float m_tile = 0.0f, m_next = 0.0f;
    (void)m_tile; (void)m_next;
    int tile_v = walk.first;
    if (!HEAD && tile_v < tile_end) {
        issue(tile_v, 0, xa, oka);
        issue(tile_v, 1, xb, okb);
        if constexpr (H16) m_next = tile_amax(tile_v);
    }
    while (tile_v < tile_end) {
        const int tile = WN_UNIFORM(tile_v);
        const int b = tile / tiles_per_b;
        const int t = (tile - b * tiles_per_b) * 32 + li;
        const bool inb = t < T;
        const int vcur = inb ? (4 * hi * T + t) * 4 : 0;
        const int vst = inb ? vcur : WN_VOFF_DEAD;   // stores of lanes past T are dropped by the range check
        // positions in front of the loss window have no skip gradient: their dZs loads carry an out-of-range offset, which the
        // range check answers with 0 without touching memory (no branch, and the region needs no zero-fill)
        const int vzs = (t >= a.zs_t0) ? vst : WN_VOFF_DEAD;
        const int next_v = tile_v + step;

        // The residual input dX_{l+1} is the INITIAL VALUE of the tap accumulators (loaded straight into them), and the
        // pre-contracted skip part of dZ that of the second accumulator pair (requested in the middle of the taps).
        // Lanes past T read a valid dummy address; their columns never leave the wave.
        f32x16 dz[2];
        const wn_rsrc_t Ssr = wn_make_buf(a.S + (long)b * 64 * T, (unsigned)(64 * T4));
        const wn_rsrc_t Gsr = wn_make_buf(a.Gt + (long)b * 64 * T, (unsigned)(64 * T4));
        float e0[2][16], e1[2][16];
        if (HEAD) {
            const wn_rsrc_t Zr = wn_make_buf(a.dZs + (long)b * a.zs_bstride, (unsigned)(64 * T4));
            WN_UNROLL
            for (int qq = 0; qq < 2; ++qq) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int so = (32 * qq + mfma32_row(r, 0)) * T4;
                    dz[qq][r] = wn_buf_load_once(Zr, vzs, so);
                    e0[qq][r] = wn_buf_load_once(Ssr, vcur, so);
                    e1[qq][r] = wn_buf_load_once(Gsr, vcur, so);
                }
            }
        } else {
            if constexpr (H16) {
                m_tile = m_next;
                s_dp = wn_pow2_scale(m_tile, 14);   // (looked up a tile ahead: tile_amax above)
            }
            WN_PRIO(WN_PRIO_MFMA);
            // The tap products are accumulated FROM ZERO and the residual input dX_{l+1} is added once, in fp32 VALU
            // arithmetic, after the last tap.  (Round 2 loaded it straight into the accumulators as their initial value, which
            // saved its 32 registers -- but every one of the 96 MFMA steps per accumulator then aligned its small products
            // against the large residual value and lost their low bits with a consistent sign: dX drifted by -1.3e-8 of its
            // mean magnitude PER LAYER, 6 x the launch pair's drift, and the bias-type gradients (sums over 184 320 positions)
            // showed it: profiles/r03/chain_pair_diff.txt, grad_gap_probe.txt.)
            acc[0] = f32x16_zero();
            acc[1] = f32x16_zero();
            float rx[2][16];
            const bool have_res = a.dXn != nullptr;
            WN_SCHED_BARRIER();
            // saved gate halves of layer l-1: the first 32 channels are requested half way through the taps, the second 32
            // at their end (their registers are the operand buffers the taps no longer need)
            WN_UNROLL
            for (int q = 0; q < NCH; q += 2) {
                consume(q, xa, oka);
                if (q + 2 < NCH) issue(tile_v, q + 2, xa, oka);
                WN_SCHED_BARRIER();
                consume(q + 1, xb, okb);
                if (q + 3 < NCH) issue(tile_v, q + 3, xb, okb);
                if (q == 0 && have_res) {   // the residual input: needed after the last tap
                    const wn_rsrc_t Rr = wn_make_buf(a.dXn + (long)b * 64 * T, (unsigned)(64 * T4));
                    WN_UNROLL
                    for (int qq = 0; qq < 2; ++qq) {
                        WN_UNROLL
                        for (int r = 0; r < 16; ++r) rx[qq][r] = WN_LD_DXN(Rr, vcur, (32 * qq + mfma32_row(r, 0)) * T4);
                    }
                }
                if (q == (NCH >> 1) - 2) {  // once, in the middle of the tap loop
                    const wn_rsrc_t Zr = wn_make_buf(a.dZs + (long)b * a.zs_bstride, (unsigned)(64 * T4));
                    WN_UNROLL
                    for (int qq = 0; qq < 2; ++qq) {
                        WN_UNROLL
                        for (int r = 0; r < 16; ++r) dz[qq][r] = wn_buf_load_once(Zr, vzs, (32 * qq + mfma32_row(r, 0)) * T4);
                    }
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) {
                        const int so = mfma32_row(r, 0) * T4;
                        e0[0][r] = wn_buf_load_once(Ssr, vcur, so);
                        e1[0][r] = wn_buf_load_once(Gsr, vcur, so);
                    }
                }
                WN_SCHED_BARRIER();
            }
            WN_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int so = (32 + mfma32_row(r, 0)) * T4;
                e0[1][r] = wn_buf_load_once(Ssr, vcur, so);
                e1[1][r] = wn_buf_load_once(Gsr, vcur, so);
            }
            if constexpr (H16) {   // accumulators -> true values (the inverse scales are powers of two), then the residual
                const float inv = inv_wd / s_dp;
                WN_UNROLL
                for (int qq = 0; qq < 2; ++qq) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) acc[qq][r] = have_res ? fmaf(acc[qq][r], inv, rx[qq][r]) : acc[qq][r] * inv;
                }
            } else if (have_res) {
                WN_UNROLL
                for (int qq = 0; qq < 2; ++qq) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) acc[qq][r] += rx[qq][r];
                }
            }
            // dX_l of this tile is finished in the accumulator layout: register r of lane (li, hi) = channel 32q + row(r, hi)
            WN_SCHED_BARRIER();
            // dZ += Wres^T dX: k-step e of block kb takes accumulator register 8 (kb & 1) + e of row tile kb >> 1
            const wn_rsrc_t Ir = wn_make_buf(a.img_res, (unsigned)(RG ? WN_RES_T_BYTES : 16));
            const int vfrag = wn_frag_off(li, hi);
            auto res_frags = [&](int kb, wn_f4 (&af)[2][3]) {
                WN_UNROLL
                for (int rt = 0; rt < 2; ++rt) {
                    WN_UNROLL
                    for (int p = 0; p < 3; ++p) {
                        if (RG) {
                            const float4 f = wn_buf_load4(Ir, vfrag, (unsigned)(kb * 6144 + p * 2048 + rt * 1024));
                            af[rt][p] = wn_f4{f.x, f.y, f.z, f.w};
                        } else {
                            af[rt][p] = *reinterpret_cast<const wn_f4*>(Wr + kb * 6144 + vfrag + p * 2048 + rt * 1024);
                        }
                    }
                }
            };
            if constexpr (H16) {
                // dX operand: its own wave-wide maximum decides the block scale; the skip part dZs (true scale, loaded into dz) is
                // added by the fma that scales the products back
                float m = 0.0f;
                WN_UNROLL
                for (int qq = 0; qq < 2; ++qq) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(acc[qq][r]));
                }
                m = inb ? m : 0.0f;
                const float s_dx = wn_pow2_scale(wave_reduce_max(m), 14);
                f32x16 dzm[2];
                dzm[0] = f32x16_zero();
                dzm[1] = f32x16_zero();
                constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
                WN_UNROLL
                for (int kb = 0; kb < 4; ++kb) {
                    float x8[8];
                    WN_UNROLL
                    for (int e = 0; e < 8; ++e) x8[e] = inb ? acc[kb >> 1][8 * (kb & 1) + e] : 0.0f;
                    wn_f4 bf[2];
                    split8h(x8, s_dx, bf);
                    wn_f4 af[2][2];
                    WN_UNROLL
                    for (int rt = 0; rt < 2; ++rt) {
                        WN_UNROLL
                        for (int p = 0; p < 2; ++p) af[rt][p] = *reinterpret_cast<const wn_f4*>(Wr + kb * BLKB + vfrag + p * PCB + rt * 1024);
                    }
                    WN_UNROLL
                    for (int t3 = 0; t3 < 3; ++t3) {
                        dzm[0] = mfma_f16(af[0][PA[t3]], bf[PB[t3]], dzm[0]);
                        dzm[1] = mfma_f16(af[1][PA[t3]], bf[PB[t3]], dzm[1]);
                    }
                }
                const float invz = inv_wr / s_dx;
                WN_UNROLL
                for (int qq = 0; qq < 2; ++qq) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) dz[qq][r] = fmaf(dzm[qq][r], invz, dz[qq][r]);
                }
            } else {
            wn_f4 afr[2][2][3];
            res_frags(0, afr[0]);
            WN_UNROLL
            for (int kb = 0; kb < 4; ++kb) {
                float x8[8];
                WN_UNROLL
                for (int e = 0; e < 8; ++e) x8[e] = acc[kb >> 1][8 * (kb & 1) + e];
                wn_f4 bf[3];
                split8(x8, bf);
                if (kb + 1 < 4) res_frags(kb + 1, afr[(kb + 1) & 1]);
                constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
                WN_UNROLL
                for (int t6 = 0; t6 < WN_EXP_NPROD; ++t6) {
                    dz[0] = mfma_bf16(afr[kb & 1][0][PA[t6]], bf[PB[t6]], dz[0]);
                    dz[1] = mfma_bf16(afr[kb & 1][1][PA[t6]], bf[PB[t6]], dz[1]);
                }
            }
            }
            WN_SCHED_BARRIER();
            // the next tile's first operand chunks go out before this tile's stores (vmcnt counts loads and stores in order)
            if (next_v < tile_end) {
                issue(next_v, 0, xa, oka);
                issue(next_v, 1, xb, okb);
                if constexpr (H16) m_next = tile_amax(next_v);
            }
            {
                const wn_rsrc_t Xr = wn_make_buf(a.dX + (long)b * 64 * T, (unsigned)(64 * T4));
                WN_UNROLL
                for (int q = 0; q < 2; ++q) {
                    WN_UNROLL
                    for (int r = 0; r < 16; ++r) wn_buf_store(Xr, acc[q][r], vst, (32 * q + mfma32_row(r, 0)) * T4);
                }
            }
        }
        WN_SCHED_BARRIER();  // the dX registers are free from here on
        WN_PRIO(WN_PRIO_GATE);
        const wn_rsrc_t Or = wn_make_buf(a.dPm + (long)b * 128 * T, (unsigned)(128 * T4));
        if (AUX) {
            // gate backward + the partial sums of the aux gradient (as k_conv64s<2>): every lane takes part in the
            // cross-lane sums, lanes past T carry dP = 0
            const int tc = inb ? t : 0;
            const int fr = tc / a.U;
            const float wj = inb ? a.upw[tc - fr * a.U] : 0.0f;
            const int F4 = a.F * 4;
            const int H = T >> 4;  // 16-sample groups per channel row
            const wn_rsrc_t Gr = wn_make_buf(a.G + (long)b * a.g_bstride, (unsigned)(128 * F4));
            const wn_rsrc_t Dr = wn_make_buf(a.dGp + (long)b * 128 * H, (unsigned)(128 * H * 4));
            const int vg = (4 * hi * a.F + fr) * 4;
            const int l16 = li & 15;
            const int h16 = (((tile - b * tiles_per_b) * 32) >> 4) + (li >> 4);
            float qsum = 0.0f;
            float pmax = 0.0f;
            (void)pmax;
            WN_UNROLL
            for (int q = 0; q < 2; ++q) {
                float ga[16], gg[16];
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    ga[r] = wn_buf_load(Gr, vg, (32 * q + mfma32_row(r, 0)) * F4);
                    gg[r] = wn_buf_load(Gr, vg, (64 + 32 * q + mfma32_row(r, 0)) * F4);
                }
                float keep_a = 0.0f, keep_g = 0.0f;
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int so = (32 * q + mfma32_row(r, 0)) * T4;
                    const float s = e0[q][r], g = a.gz ? e1[q][r] * wn_rcp(e0[q][r]) : e1[q][r], dzv = inb ? dz[q][r] : 0.0f;
                    const float dpa = dzv * g * (s * (1.0f - s)), dpg = dzv * s * (1.0f - g * g);
                    wn_buf_store(Or, dpa, vst, so);   // lanes past T: out-of-range offset, dropped (no branch per element)
                    wn_buf_store(Or, dpg, vst, so + 64 * T4);
                    if (H16) pmax = fmaxf(pmax, fmaxf(fabsf(dpa), fabsf(dpg)));
                    qsum += dpa * ga[r] + dpg * gg[r];
                    const float ra = wn_row16_sum(wj * dpa), rg = wn_row16_sum(wj * dpg);
                    keep_a = (l16 == r) ? ra : keep_a;
                    keep_g = (l16 == r) ? rg : keep_g;
                }
                const int row = 32 * q + mfma32_row(l16, hi);
                const int doff = (h16 < H) ? (row * H + h16) * 4 : 0x7ffffff0;
                wn_buf_store(Dr, keep_a, doff, 0);
                wn_buf_store(Dr, keep_g, doff, (h16 < H) ? 64 * H * 4 : 0);
                WN_SCHED_BARRIER();  // one 32-row half at a time (register budget)
            }
            qsum += __shfl_xor(qsum, 32, 64);
            if (hi == 0 && inb) a.qp[(long)b * T + t] = qsum;
            if (H16 && a.amaxPm != nullptr) {
                pmax = wave_reduce_max(pmax);
                if (lane == 0) a.amaxPm[(long)b * tiles_per_b + (tile - b * tiles_per_b)] = pmax;
            }
        } else {
            float pmax = 0.0f;
            (void)pmax;
            WN_UNROLL
            for (int q = 0; q < 2; ++q) {
                WN_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int so = (32 * q + mfma32_row(r, 0)) * T4;
                    const float s = e0[q][r], g = a.gz ? e1[q][r] * wn_rcp(e0[q][r]) : e1[q][r], dzv = H16 ? (inb ? dz[q][r] : 0.0f) : dz[q][r];
                    const float dpa = dzv * g * (s * (1.0f - s)), dpg = dzv * s * (1.0f - g * g);
                    wn_buf_store(Or, dpa, vst, so);
                    wn_buf_store(Or, dpg, vst, so + 64 * T4);
                    if (H16) pmax = fmaxf(pmax, fmaxf(fabsf(dpa), fabsf(dpg)));
                }
            }
            if (H16 && a.amaxPm != nullptr) {
                pmax = wave_reduce_max(pmax);
                if (lane == 0) a.amaxPm[(long)b * tiles_per_b + (tile - b * tiles_per_b)] = pmax;
            }
        }
        tile_v = next_v;
    }
}

int wn_fused_chain_supported(int R, int K, int S) {
    return wn_fused_supported(R, K, S) && K >= 1 && K <= 3 && (size_t)(K * 8 + (K >= 3 ? 0 : 4)) * 6144 <= 160 * 1024;
}

int wn_fused_bwd_chain(const float* wd_b, const float* dP, const float* dXn, float* dX, const float* wres_prev, const float* dZs,
                       long zs_bstride, const float* S, const float* Gt, int gt_is_z, float* dP_prev, const float* G, long g_bstride,
                       const float* upw, int U, int F, float* dGp, float* qp, int B, int T, int K, int dilation,
                       const float* img_taps, const float* img_res, int zs_t0, const float* amaxP, float* amaxPm, wn_stream_t st) {
    // dP (, dXn), dZs, S, Gt in; dX, dP_prev out (+ dGp 8 / qp 1 words per timestep with the aux partials)
    const bool aux = dGp != nullptr;
    WN_PROF("fused_bwd_chain", 2.0 * (double)B * T * 64.0 * (K * 128.0 + 64.0),
            4.0 * (double)B * T * (128.0 + (dXn ? 64.0 : 0.0) + 64.0 + 128.0 + 64.0 + 128.0 + (aux ? 9.0 : 0.0)), st);
    if (K < 1 || K > 3) return 1;
    if (aux && (U < 16 || (U & 15) || (T & 15) || (long)U * F != T)) return 1;
    if (K == 3 && !(img_taps && img_res)) return 1;   // K = 3 reads the Wres^T fragments from the image
    ChainArgs a;
    a.img_taps = (img_taps && img_res) ? img_taps : nullptr; a.img_res = img_res;
    a.wd_b = wd_b; a.dP = dP; a.dXn = dXn; a.dX = dX;
    a.wres = wres_prev; a.dZs = dZs; a.zs_bstride = zs_bstride; a.zs_t0 = zs_t0; a.S = S; a.Gt = Gt; a.gz = gt_is_z; a.dPm = dP_prev;
    a.B = B; a.T = T; a.K = K; a.dil = dilation;
    a.G = G; a.g_bstride = g_bstride; a.upw = upw; a.U = U; a.F = F; a.dGp = dGp; a.qp = qp;
    a.amaxP = amaxP; a.amaxPm = amaxPm;
    const long ntiles = (long)B * ((T + 31) / 32);
    const long nblk = balanced_blocks(ntiles);
    if (amaxP != nullptr) {   // fp16 pair split: img_taps / img_res are the two-piece images (wn_fused_pack_images16)
        if (!img_taps || !img_res) return 1;
        const size_t lds16 = (size_t)(K * 8 + 4) * 4096;
#define WN_CHAIN_LAUNCH16(AUXV, KV)                                                                               \
    do {                                                                                                          \
        if (set_lds(k_chain64s<AUXV, KV, false, true>, lds16)) return 1;                                          \
        WN_LAUNCH((k_chain64s<AUXV, KV, false, true>), dim3((unsigned)nblk), dim3(WN_FT), lds16, st, a);          \
    } while (0)
        if (aux) {
            if (K == 1) WN_CHAIN_LAUNCH16(1, 1);
            else if (K == 2) WN_CHAIN_LAUNCH16(1, 2);
            else WN_CHAIN_LAUNCH16(1, 3);
        } else {
            if (K == 1) WN_CHAIN_LAUNCH16(0, 1);
            else if (K == 2) WN_CHAIN_LAUNCH16(0, 2);
            else WN_CHAIN_LAUNCH16(0, 3);
        }
#undef WN_CHAIN_LAUNCH16
        return 0;
    }
    const size_t lds = (size_t)(K * 8 + (K >= 3 ? 0 : 4)) * 6144;
#define WN_CHAIN_LAUNCH(AUXV, KV)                                                                       \
    do {                                                                                                \
        if (set_lds(k_chain64s<AUXV, KV>, lds)) return 1;                                               \
        WN_LAUNCH((k_chain64s<AUXV, KV>), dim3((unsigned)nblk), dim3(WN_FT), lds, st, a);               \
    } while (0)
    if (aux) {
        if (K == 1) WN_CHAIN_LAUNCH(1, 1);
        else if (K == 2) WN_CHAIN_LAUNCH(1, 2);
        else WN_CHAIN_LAUNCH(1, 3);
    } else {
        if (K == 1) WN_CHAIN_LAUNCH(0, 1);
        else if (K == 2) WN_CHAIN_LAUNCH(0, 2);
        else WN_CHAIN_LAUNCH(0, 3);
    }
#undef WN_CHAIN_LAUNCH
    return 0;
}

// Top of the chain: dP_{L-1} = gate'(dZs_{L-1}) (k_chain64s<AUX, K, true>: no weights, no LDS, the epilogue only)
int wn_fused_bwd_chain_head(const float* dZs, long zs_bstride, const float* S, const float* Gt, int gt_is_z, float* dP_prev,
                            const float* G, long g_bstride, const float* upw, int U, int F, float* dGp, float* qp, int B, int T,
                            int zs_t0, float* amaxPm, wn_stream_t st) {
    const bool aux = dGp != nullptr;
    WN_PROF("fused_bwd_gate", 0.0, 4.0 * (double)B * T * (64.0 + 128.0 + 128.0 + (aux ? 9.0 : 0.0)), st);
    if (aux && (U < 16 || (U & 15) || (T & 15) || (long)U * F != T)) return 1;
    ChainArgs a;
    a.img_taps = nullptr; a.img_res = nullptr; a.wd_b = nullptr; a.dP = nullptr; a.dXn = nullptr; a.dX = nullptr; a.wres = nullptr;
    a.dZs = dZs; a.zs_bstride = zs_bstride; a.zs_t0 = zs_t0; a.S = S; a.Gt = Gt; a.gz = gt_is_z; a.dPm = dP_prev;
    a.B = B; a.T = T; a.K = 1; a.dil = 1;
    a.G = G; a.g_bstride = g_bstride; a.upw = upw; a.U = U; a.F = F; a.dGp = dGp; a.qp = qp;
    a.amaxP = nullptr; a.amaxPm = amaxPm;
    const long ntiles = (long)B * ((T + 31) / 32);
    long nblk = (ntiles + WN_FW - 1) / WN_FW;   // one tile per wave up to 1024 workgroups (HBM-bound, nothing to amortise)
    if (nblk > 1024) nblk = 1024;
    if (nblk >= 8) nblk &= ~7L;                 // whole XCD rounds for the tile walk
    if (amaxPm != nullptr) {   // the fp16 pair chain follows: also record max |dP| per tile
        if (aux)
            WN_LAUNCH((k_chain64s<1, 1, true, true>), dim3((unsigned)nblk), dim3(WN_FT), 0, st, a);
        else
            WN_LAUNCH((k_chain64s<0, 1, true, true>), dim3((unsigned)nblk), dim3(WN_FT), 0, st, a);
    } else if (aux)
        WN_LAUNCH((k_chain64s<1, 1, true>), dim3((unsigned)nblk), dim3(WN_FT), 0, st, a);
    else
        WN_LAUNCH((k_chain64s<0, 1, true>), dim3((unsigned)nblk), dim3(WN_FT), 0, st, a);
    return 0;
}
