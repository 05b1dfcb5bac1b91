// wn_dlpf.hip -- the matrix-core persistent decode of wn_dlpm.hip (same algorithm, stage decomposition, packed weight images
// and tiles: reference wavenet.py:355-385, 538-549, 518-523) with a different hand-off between the workgroups.
//
// wn_dlpm.hip hands a stage's vectors over as self-validating 8-byte granules {value, tag}: every consumer polls every element
// through its registers -- a stage's inputs arrive as three dependent memory round trips (older taps, z, x; 16 granules in
// flight per thread) and ~6 of its 11 us are that.  Here a stage's outputs are PLAIN vectors written with agent-scope
// (write-through) stores, and a unit publishes ONE flag per stage: every wave waits for its own stores, a workgroup barrier,
// then one thread stores the flag.  A consumer's first wave polls the flags of the block's units (one lane per unit), a
// barrier, and then the whole stage's input -- z, x and the older taps -- comes in as ONE batch of global -> LDS transfers with
// the agent-scope cache policy (buffer_load_dwordx4 ... lds sc1: 16 bytes per lane, no registers, no per-element arithmetic).
// No fence on either side: tools/microbench/handoff.hip, mode "sc1 stores + flag, sc1 loads", 0 stale words across XCDs at
// 1.5 us per hop (profiles/r04/handoff_microbench.txt); the release / acquire fences of the memory model's plain-store protocol
// cost the same for two workgroups but write back / invalidate a whole XCD's L2 each time -- with 24 units per XCD the step
// went from 318 us (16 utterances, 8 units per XCD) to 506 (48).
// With plain vectors the dilation queues need no private copies either: `queues` (the shared rings the context pass fills)
// is read by everyone and written by the owner of a channel -- one stage LATER than the value is computed, after the flags of
// that stage have been seen, because for kernel_size 2 the slot of position p is the slot of the tap p - d that the other
// units are still reading during the stage itself.
// LDS: the stage's inputs in the order the tiles read them -- tile step i of wave w contracts the k of its four lane groups
// q, k = (4 w + q) NS + i, and finds their 16 columns at rows 4 (w NS + i) + q: 64 consecutive floats, one conflict-free read per
// lane, and 16 such rows are what one transfer instruction writes.  The x / skip row set has its own copy of the z part in ITS
// order where the LDS has room (kernel_size 2 class, NSP = 48: 148 KB); the kernel_size 3 class (NSP = 64) reads z where the gate
// set's order put it.
#include "wn_dlp.h"

#include <type_traits>

#include "wn_prof.h"

typedef unsigned long long u64;

static __device__ __forceinline__ long dlpf_queue_off(int l, int depth, int K, int R) {
    const long cyc = l / depth, in = l % depth;
    return (long)R * (K - 1) * (cyc * ((1L << depth) - 1) + ((1L << in) - 1));
}
#define DLPF_SPIN_MAX (1 << 22)
#ifndef WN_DLPF_TAP_PREFETCH
// 1: the older taps of a stage are requested a stage early (behind the previous stage's publish and weight requests) instead of
// with the stage's other inputs.  Measured SLOWER on MI355X (profiles/r05/decode_tap_ab.txt: kernel_size 2, 16 utterances 244 ->
// 264 us per step; kernel_size 3: 275 -> 305): the flag poll of the next stage waits for every outstanding load of the wave
// (vmcnt is in order), so the early requests sit in front of the poll instead of beside it.  Kept as an A/B switch, default off.
#define WN_DLPF_TAP_PREFETCH 0
#endif
#ifdef WN_DLP_TIMING   // stamps of wn_dlp.hip / wn_dlpm.hip (tools/dlp_timing.py): unit 0 of block 0, step p0 + 3
#define DLPF_STAMP(stage, ph)                                                                                        \
    do {                                                                                                             \
        if (tid == 0 && blockIdx.x == 0 && p == a.p0 + 3 && (stage) < 40)                                            \
            reinterpret_cast<long long*>(a.err + 16)[(stage) * 8 + (ph)] = (long long)wall_clock64();                \
    } while (0)
#else
#define DLPF_STAMP(stage, ph)
#endif

template <int NSP, int NSX>
__global__ __launch_bounds__(WN_DLP_T, 2) void k_dlpf(WnDlpArgs a) {
    WN_DYN_SMEM(smem_raw);
    if (wn_load_coherent_int(a.err) != 0) return;   // an earlier launch on this state timed out (or this one already has): nothing to continue from
    constexpr int CG = 8, SL = 32, CB = WN_DLPM_CB;
    constexpr int KPAD = SL * NSP, XPAD = SL * NSX;
    // XCOPY: the x / skip row set has its own copy of the z part in ITS tile order (kernel_size 2 class: 148 KB of LDS).  The
    // kernel_size 3 class (NSP = 64: 128 KB of inputs) has no room for it: that set reads the z part where the gate set's order
    // put it -- the k of its lane group q sit 64 floats apart there as well (NSP = 4 NSX), but the four groups are 16 rows apart
    // in every tile step: 4-way bank conflicts on 16 of the 80 tile steps.
    constexpr bool XCOPY = NSP * 3 <= 160;
    static_assert(XCOPY || NSP == 4 * NSX, "without its own copy the x / skip set reads the gate set's order: NSP = 4 NSX");
    float* s_p = reinterpret_cast<float*>(smem_raw);   // [KPAD][CB] inputs in the order of the gate row set's tile steps
    float* s_x = s_p + KPAD * CB;                      // [XPAD][CB] the z part (post net: the vector) in the x / skip set's order
    float* s_red = s_x + (XCOPY ? XPAD * CB : 0);      // partial tiles [2 sets][8 waves][16 rows][16 columns]
    float* s_xown = s_red + 4096;                      // [8][CB] x of the unit's own channels (previous stage)
    float* s_sk = s_xown + 8 * CB;                     // [8][CB] skip accumulators of the unit's rows
    int* s_tok = reinterpret_cast<int*>(s_sk + 8 * CB);   // [3][CB] the newest K tokens of the block's utterances
    int* s_flag = s_tok + 4 * CB;                      // [0] a poll timed out

    const int tid = threadIdx.x, lane = tid & 63, wave = WN_UNIFORM(tid >> 6);
    const int lc = lane & 15, q = lane >> 4;     // transfers and tiles: column, lane group
    const int col = tid & 15, kk = tid >> 4;     // epilogues: utterance column of the block, output row 0 .. 31
    const int R = a.R, S = a.S, L = a.L, K = a.K, B = a.B, Qo = a.Qo, Bp = a.Bp;
    const int NU = a.plan.NU;
    const int u = (int)blockIdx.x % NU, cblk = (int)blockIdx.x / NU;
    const int c0 = u * CG;
    const int SU = a.plan.SU, QU = a.plan.QU, KP = a.plan.KP;
    const int nbc = (B - cblk * CB) < CB ? (B - cblk * CB) : CB;
    const int b = cblk * CB + col;
    const bool live = col < nbc;
    float* zx = reinterpret_cast<float*>(a.gz);   // [2][2 R][Bp]: z rows [0, R), x rows [R, 2 R) of a stage, two stages alive
    float* vs = reinterpret_cast<float*>(a.gs);   // [S][Bp] relu(skip sum)
    float* vo = reinterpret_cast<float*>(a.go);   // [S][Bp] relu(post1)
    float* vl = reinterpret_cast<float*>(a.gl);   // [Qo][Bp] logits
    u64* flags = a.flags + (long)cblk * NU;
    const wn_rsrc_t rZX = wn_make_buf(zx, (unsigned)((long)2 * 2 * R * Bp * 4));
    const wn_rsrc_t rQ = wn_make_buf(a.queues, (unsigned)(a.qfloats * B * 4 + 64));   // (+ the 16 floats a ragged block's last row reads past B; the region is carved with 64)
    auto posP = [&](int k) -> int { const int g = k / NSP, i = k - g * NSP; return (((g >> 2) * NSP + i) * 4 + (g & 3)) * CB; };

    for (int i = tid; i < KPAD * CB + (XCOPY ? XPAD * CB : 0); i += WN_DLP_T) s_p[i] = 0.0f;
    for (int i = tid; i < 8 * CB; i += WN_DLP_T) { s_xown[i] = 0.0f; s_sk[i] = 0.0f; }
    for (int i = tid; i < K * CB; i += WN_DLP_T) {
        const int j = i / CB, c = i % CB;
        const long pos = (long)a.p0 - (K - 1 - j);
        long long tok = (pos >= 0 && c < nbc) ? a.samples[(long)(cblk * CB + c) * a.Ttot + pos] % a.Q : 0;
        if (tok < 0) tok += a.Q;
        s_tok[j * CB + c] = pos >= 0 ? (int)tok : -1;
    }
    if (tid == 0) s_flag[0] = 0;
    __syncthreads();

    // x_0[c][column] of the current step: front conv as a gather of K weight columns (wavenet.py:355-356, 513-516)
    auto x0_of = [&](int c, int cl) -> float {
        float v = a.params[a.off_causal_b + c];
        for (int k = 0; k < K; ++k) {
            const int tok = s_tok[k * CB + cl];
            if (tok >= 0) v += a.params[a.off_causal_w + ((long)c * a.Q + tok) * K + k];
        }
        return v;
    };
    // wait until every unit of the block has published `tag` (flags only grow); the first wave polls, one lane per unit
    auto wait_flags = [&](unsigned tag) {
        if (wave == 0) {
            for (int u0 = 0; u0 < NU; u0 += 64) {
                const int uu = u0 + lane;
                int spin = 0;
                while (uu < NU && (unsigned)(wn_granule_load(flags + uu) >> 32) < tag) {
                    if (++spin > DLPF_SPIN_MAX || s_flag[0]) { s_flag[0] = 1; break; }
                    WN_SLEEP(1);
                }
            }
        }
        __syncthreads();
    };
    auto publish = [&](unsigned tag) {   // after the workgroup's plain stores
        WN_WAIT_VMCNT(0);   // every wave: its own stores have reached the L2 before the barrier
        __syncthreads();
        if (tid == 0) wn_granule_store(flags + u, 0.0f, tag);   // {0, tag}: the tag is the upper word
    };
    auto load_weights = [&](auto& w, auto ns_c, const float* img) {
        constexpr int ns = decltype(ns_c)::value;
        const wn_f4* src = reinterpret_cast<const wn_f4*>(img) + (long)wave * (ns / 4) * 64 + lane;
        WN_UNROLL
        for (int t4 = 0; t4 < ns / 4; ++t4) {
            const wn_f4 v = wn_ld4_stream(src + (long)t4 * 64);
            w[4 * t4] = v.x; w[4 * t4 + 1] = v.y; w[4 * t4 + 2] = v.z; w[4 * t4 + 3] = v.w;
        }
    };
    // one row set times the staged column block: tile step i of this wave reads the 64 consecutive floats behind it
    auto tile = [&](const auto& w, auto ns_c, const float* src, int set, bool on) {   // src: this lane's element of tile step 0
        constexpr int ns = decltype(ns_c)::value;
        if (!on) return;
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        constexpr int G = 8;
        static_assert(ns % G == 0, "groups of 8 tile steps");
        float bc[G], bn[G];
        WN_UNROLL
        for (int e = 0; e < G; ++e) bc[e] = src[e * 64];
        WN_UNROLL
        for (int i0 = 0; i0 < ns; i0 += G) {
            if (i0 + G < ns) {
                WN_UNROLL
                for (int e = 0; e < G; ++e) bn[e] = src[(i0 + G + e) * 64];
            }
            WN_UNROLL
            for (int e = 0; e < G; e += 2) {
                acc0 = mfma16(w[i0 + e], bc[e], acc0);
                acc1 = mfma16(w[i0 + e + 1], bc[e + 1], acc1);
            }
            WN_SCHED_BARRIER();
            WN_UNROLL
            for (int e = 0; e < G; ++e) bc[e] = bn[e];
        }
        float* dst = s_red + ((set * 8 + wave) * 16 + 4 * q) * 16 + lc;
        WN_UNROLL
        for (int r = 0; r < 4; ++r) dst[r * 16] = acc0[r] + acc1[r];
    };
    auto row_sum = [&](int set, int row) -> float {   // fixed order over the 8 waves
        const float* pp = s_red + ((set * 8) * 16 + row) * 16 + col;
        float s = 0.0f;
        WN_UNROLL
        for (int w = 0; w < 8; ++w) s += pp[w * 256];
        return s;
    };
    // rows [lo, hi) of a plain [..][Bp] array (row0 = its row of k = 0) into the gate set's order (s_p).  One transfer instruction
    // moves 16 bytes per lane = 4 tile steps x 4 lane groups x 16 columns (1 KB of LDS in tile order): lane l carries columns
    // 4 (l % 4) .. + 3 of the row  k = (4 wave + qd) NSP + i4 + id,  qd = (l / 4) % 4,  id = l / 16.
    auto fetch_p = [&](wn_rsrc_t rs, int row0, int lo, int hi) {
        if (4 * wave * NSP >= hi || 4 * (wave + 1) * NSP <= lo) return;   // (wave-uniform) none of the wave's four groups reaches the range
        const int qd = (lane >> 2) & 3, id = lane >> 4, c4 = (lane & 3) * 4;
        const int k0 = (4 * wave + qd) * NSP + id;
        unsigned voff = (unsigned)(((row0 + k0) * Bp + cblk * CB + c4) * 4);
        WN_NOUNROLL
        for (int i4 = 0; i4 < NSP; i4 += 4) {
            if (k0 + i4 >= lo && k0 + i4 < hi)
                wn_buf_load_lds16_coherent(rs, reinterpret_cast<char*>(s_p + (wave * NSP + i4) * 64), (int)voff, 0u);
            voff += (unsigned)(4 * Bp * 4);
        }
    };
    // rows [0, rows) of a plain [rows][Bp] vector for the x / skip row set: into its own order (s_x), or where the gate set has them
    auto fetch_vector_x = [&](wn_rsrc_t rs, int row0, int rows) {
        if (!XCOPY) {
            fetch_p(rs, row0, 0, rows);
            return;
        }
        const int qd = (lane >> 2) & 3, id = lane >> 4, c4 = (lane & 3) * 4;
        const int k0 = (4 * wave + qd) * NSX + id;
        WN_NOUNROLL
        for (int i4 = 0; i4 < NSX; i4 += 4) {
            if (k0 + i4 < rows)
                wn_buf_load_lds16_coherent(rs, reinterpret_cast<char*>(s_x + (wave * NSX + i4) * 64), ((row0 + k0 + i4) * Bp + cblk * CB + c4) * 4, 0u);
        }
    };
    // The older taps of layer sn at step pp (rows k >= 2 R of the gate set's input) from the shared rings into their rows of s_p.
    // They were written a whole step ago (one stage after they were computed), so they do NOT wait for the flags of the previous
    // stage: with WN_DLPF_TAP_PREFETCH they are requested a stage early -- right after the tiles of stage sn - 1 have released the
    // rows (behind that stage's publish and the next stage's weight requests) --, a third (kernel_size 2) or half (kernel_size 3)
    // of a stage's input bytes whose round trip then runs beside the flag wait instead of after it.  (The slot a tap is read
    // from is overwritten at stage sn + 1 of this step, after the flags of stage sn: every unit has its copy by then.)
    auto fetch_taps = [&](int sn, int pp) {
        if (sn >= L) return;
        const int qd = (lane >> 2) & 3, id = lane >> 4, c4 = (lane & 3) * 4;
        const int k0 = (4 * wave + qd) * NSP + id;   // + i4
        const int kw_lo = 4 * wave * NSP, kw_hi = kw_lo + 4 * NSP;     // (wave-uniform) k range of the wave's groups
        if (!(kw_hi > 2 * R && kw_lo < KP)) return;
        const int d = 1 << (sn % a.depth), Dq = (K - 1) * d;
        const long qoff = dlpf_queue_off(sn, a.depth, K, R);
        // (rows of a [..][B] ring are 4 B bytes apart: 16-byte transfers from 4-byte aligned addresses; the columns of a
        // ragged last block past B read the next row's first floats, which nobody uses)
        unsigned tbase[2];   // byte offset of row c = 0 of tap jt (kernel_size <= 3: two older taps at most)
        WN_UNROLL
        for (int jt = 0; jt < 2; ++jt) {
            int slot = (pp - (K - 1 - jt) * d) % Dq;
            if (slot < 0) slot += Dq;
            tbase[jt] = (unsigned)(((qoff + (long)slot * R) * B + cblk * CB + c4) * 4);
        }
        WN_NOUNROLL
        for (int i4 = 0; i4 < NSP; i4 += 4) {
            const int kt = k0 + i4 - 2 * R;   // row of the tap part
            if (kt >= 0 && k0 + i4 < KP) {
                const int jt = kt >= R ? 1 : 0;
                wn_buf_load_lds16_coherent(rQ, reinterpret_cast<char*>(s_p + (wave * NSP + i4) * 64),
                                  (int)(tbase[jt] + (unsigned)((kt - jt * R) * B * 4)), 0u);
            }
        }
    };
    // this lane's element of tile step 0 of the two row sets
    const float* srcP = s_p + (wave * NSP) * 64 + lane;
    const float* srcX = XCOPY ? s_x + (wave * NSX) * 64 + lane : s_p + posP((4 * wave + q) * NSX) + lc;

    float wP[NSP], wX[NSX];
    auto issue_stage_weights = [&](int sn) {   // stage sn in [0, L]
        const float* img = a.wpk + ((long)sn * NU + u) * a.plan.stage_floats;
        if (sn < L) load_weights(wP, std::integral_constant<int, NSP>(), img);
        if (sn >= 1) load_weights(wX, std::integral_constant<int, NSX>(), img + 512L * NSP);
    };
    const float* pimg = a.wpost + (long)u * a.plan.post_floats;
    issue_stage_weights(0);
    if (WN_DLPF_TAP_PREFETCH) fetch_taps(0, a.p0);
    for (int p = a.p0; p < a.p1; ++p) {
        const unsigned tag0 = (unsigned)(p + 1) * (unsigned)(L + 4) + 1u;    // tag of (step p, stage s) = tag0 + s
        for (int s = 0; s <= L; ++s) {
            const bool hasP = s < L, hasX = s >= 1;
            const int d = 1 << (s % a.depth), Dq = (K - 1) * d;
            const long qoff_s = hasP ? dlpf_queue_off(s, a.depth, K, R) : 0;
            const int par = (s - 1) & 1;
            DLPF_STAMP(s, 0);
            if (s >= 1) {
                wait_flags(tag0 + (unsigned)(s - 1));
                // every unit has read the older taps of layer s-1 now: its newest value may go into the shared ring
                if (K >= 2 && live && kk >= CG && kk < 2 * CG) {
                    const int dp = 1 << ((s - 1) % a.depth), Dp = (K - 1) * dp;
                    wn_store_coherent(a.queues + (dlpf_queue_off(s - 1, a.depth, K, R) + (long)(p % Dp) * R + c0 + kk - CG) * B + b, s_xown[(kk - CG) * CB + col]);
                }
            }
            DLPF_STAMP(s, 1);
            // (1) the stage's inputs: global -> LDS.  Gate row set: [z_{s-1} | x_{s-1} | older taps of x_s] in its tile order
            if (hasP) {
                // One transfer instruction moves 16 bytes per lane = 4 tile steps x 4 lane groups x 16 columns (1 KB of LDS in tile
                // order): lane l carries columns 4 (l % 4) .. + 3 of the row  k = (4 wave + qd) NSP + i4 + id,  qd = (l / 4) % 4,
                // id = l / 16.  Two loops, one per source array (an instruction takes ONE wave-uniform resource): [z | x] rows
                // k < 2 R of the previous stage's vector, older taps k >= 2 R from the rings; a wave runs a loop only if one of its
                // groups reaches into that part.
                // rows of [z | x] that are handed over: z from stage 1 on, x from stage 2 on
                fetch_p(rZX, par * 2 * R, s >= 1 ? 0 : R, s >= 2 ? 2 * R : (s >= 1 ? R : 0));
                if (!WN_DLPF_TAP_PREFETCH) fetch_taps(s, p);
                if (s <= 1) {   // x_0 is every unit's own gather of the front-conv table (eight rows' reads in flight)
                    int tk[3];
                    WN_UNROLL
                    for (int k = 0; k < 3; ++k) tk[k] = k < K ? s_tok[k * CB + col] : -1;
                    for (int j1 = 0; j1 < R / 32; j1 += 8) {
                        float bv[8], wv[8][3];
                        WN_UNROLL
                        for (int jj = 0; jj < 8; ++jj) {
                            const int c = kk + 32 * (j1 + jj);
                            const bool ok = j1 + jj < R / 32 && live;
                            bv[jj] = ok ? a.params[a.off_causal_b + c] : 0.0f;
                            WN_UNROLL
                            for (int k = 0; k < 3; ++k)
                                wv[jj][k] = (ok && tk[k] >= 0) ? a.params[a.off_causal_w + ((long)c * a.Q + tk[k]) * K + k] : 0.0f;
                        }
                        WN_UNROLL
                        for (int jj = 0; jj < 8; ++jj) {
                            if (j1 + jj < R / 32 && live) {
                                float v = bv[jj];
                                WN_UNROLL
                                for (int k = 0; k < 3; ++k)
                                    if (tk[k] >= 0) v += wv[jj][k];   // the order of x0_of
                                s_p[posP(R + kk + 32 * (j1 + jj)) + col] = v;
                            }
                        }
                    }
                }
            }
            if (hasX && (XCOPY || !hasP)) fetch_vector_x(rZX, par * 2 * R, R);   // x / skip row set: the z part in its own order (without a
                                                                                 // copy it is where the gate set's transfers put it)
            // (2) what this thread's output reads from memory.  Output row kk: 0 .. 7 gate of channel c0 + kk, 8 .. 15 x of
            // channel c0 + kk - 8, 16 .. 16 + SU - 1 skip rows
            float e0 = 0.0f, e1 = 0.0f;
            if (live) {
                if (kk < CG) {
                    if (hasP) {
                        const int t = p > a.n_pad ? p - a.n_pad : 0;   // replicated first column inside the left padding
                        int f = t / a.Ue;
                        const float wj = a.upw[t - f * a.Ue];
                        if (f > a.F - 1) f = a.F - 1;
                        const float* Gs = a.G + ((long)b * a.F + f) * a.nG + (long)s * 2 * R;
                        e0 = wj * Gs[c0 + kk] + a.cfold[(long)s * 2 * R + c0 + kk];
                        e1 = wj * Gs[R + c0 + kk] + a.cfold[(long)s * 2 * R + R + c0 + kk];
                    }
                } else if (kk < 2 * CG) {
                    if (s < L)
                        e0 = s == 0 ? x0_of(c0 + kk - CG, col) : a.params[a.off_res_b0 + (long)(s - 1) * a.res_b_lstride + c0 + kk - CG];
                }
            }
            WN_WAIT_VMCNT(0);   // the transfers (and the stage's weights) have landed
            __syncthreads();
            DLPF_STAMP(s, 2);
            // (3) the two row sets on the matrix cores
            tile(wP, std::integral_constant<int, NSP>(), srcP, 0, hasP);
            tile(wX, std::integral_constant<int, NSX>(), srcX, 1, hasX);
            DLPF_STAMP(s, 3);
            __syncthreads();
            DLPF_STAMP(s, 4);
            // (4) sums of the 8 partial tiles and the outputs' epilogues: plain stores into the stage's vectors
            if (live) {
                if (kk < CG) {
                    if (hasP) {   // gate (wavenet.py:542-544)
                        const float sg = row_sum(0, kk), st = row_sum(0, CG + kk);
                        wn_store_coherent(zx + ((long)(s & 1) * 2 * R + c0 + kk) * Bp + b, wn_sigmoid(sg + e0) * wn_tanh(st + e1));
                    }
                } else if (kk < 2 * CG) {
                    if (s < L) {
                        const int c = kk - CG;
                        float xs;
                        if (s == 0) {   // x_0 of the unit's own channels
                            xs = e0;
                        } else {        // x_s = res_1x1(z_{s-1}) + x_{s-1}   (wavenet.py:546-548)
                            xs = row_sum(1, c) + e0 + s_xown[c * CB + col];
                            wn_store_coherent(zx + ((long)(s & 1) * 2 * R + R + c0 + c) * Bp + b, xs);
                        }
                        s_xown[c * CB + col] = xs;   // (into the shared ring at the next stage, see above)
                    }
                } else if (kk < 2 * CG + SU) {
                    if (hasX) s_sk[(kk - 2 * CG) * CB + col] += row_sum(1, CG + (kk - 2 * CG));   // skip sum (wavenet.py:545, 365)
                }
            }
            DLPF_STAMP(s, 5);
            if (s < L) {
                publish(tag0 + (unsigned)s);
                issue_stage_weights(s + 1);
                if (WN_DLPF_TAP_PREFETCH) fetch_taps(s + 1, p);
            }
            DLPF_STAMP(s, 6);
        }
        DLPF_STAMP(L + 1, 0);
        __syncthreads();   // the skip accumulators of the last stage's epilogue are complete
        // ---- post net (wavenet.py:518-523): relu(skip sum) -> conv_post_1 + relu -> conv_post_2, three more hops ----
        for (int i = tid; i < SU * nbc; i += WN_DLP_T) {
            const int r = i / nbc, c = i % nbc, row = u * SU + r;
            if (row < S) wn_store_coherent(vs + (long)row * Bp + cblk * CB + c, fmaxf(s_sk[r * CB + c] + a.bskip[row], 0.0f));
            s_sk[r * CB + c] = 0.0f;
        }
        publish(tag0 + (unsigned)L);   // (tag of stage L: its z goes nowhere, its skip rows are the vector of tag L)
        load_weights(wX, std::integral_constant<int, NSX>(), pimg);
        for (int stage = 0; stage < 2; ++stage) {
            wait_flags(tag0 + (unsigned)(L + stage));
            fetch_vector_x(wn_make_buf(stage == 0 ? vs : vo, (unsigned)((long)S * Bp * 4)), 0, S);
            float pb = 0.0f;   // the output row's bias
            {
                const int row = u * (stage == 0 ? SU : QU) + kk;
                if (live && (stage == 0 ? (kk < SU && row < S) : (kk < QU && row < Qo)))
                    pb = a.params[(stage == 0 ? a.off_post1_b : a.off_post2_b) + row];
            }
            WN_WAIT_VMCNT(0);
            __syncthreads();
            tile(wX, std::integral_constant<int, NSX>(), srcX, 0, true);
            __syncthreads();
            if (live && kk < 16) {
                if (stage == 0) {
                    const int row = u * SU + kk;
                    if (kk < SU && row < S) wn_store_coherent(vo + (long)row * Bp + b, fmaxf(row_sum(0, kk) + pb, 0.0f));
                } else {
                    const int row = u * QU + kk;
                    if (kk < QU && row < Qo) wn_store_coherent(vl + (long)row * Bp + b, row_sum(0, kk) + pb);
                }
            }
            publish(tag0 + (unsigned)(L + 1 + stage));
            if (stage == 0) load_weights(wX, std::integral_constant<int, NSX>(), pimg + 512L * NSX);
            else if (p + 1 < a.p1) issue_stage_weights(0);
        }
        // ---- token choice, by every unit for itself (wavenet.py:371-381): first-max argmax or inverse CDF on the caller's draw ----
        {
            wait_flags(tag0 + (unsigned)(L + 2));
            const wn_rsrc_t rL = wn_make_buf(vl, (unsigned)((long)Qo * Bp * 4));
            for (int r0 = 16 * wave; r0 < Qo; r0 += 128) {   // logits in natural order: row qi at s_p[qi * 16]; 16 rows per instruction
                const int row = r0 + (lane >> 2);
                if (row < Qo) wn_buf_load_lds16_coherent(rL, reinterpret_cast<char*>(s_p + r0 * CB), (row * Bp + cblk * CB + (lane & 3) * 4) * 4, 0u);
            }
            WN_WAIT_VMCNT(0);
            __syncthreads();
            const int nq = (Qo - kk + 31) / 32;   // logit rows kk + 32 j < Qo of this thread
            {
                float best = -3.0e38f;
                int bi = 0x7fffffff;
                for (int j = 0; j < nq; ++j) {
                    const int qi = kk + 32 * j;
                    const float v = live ? s_p[qi * CB + col] : -3.0e38f;
                    if (live && u == 0 && a.logits_out) a.logits_out[((long)b * a.Ttot + p) * Qo + qi] = v;
                    if (v > best) { best = v; bi = qi; }
                }
                s_red[(kk * CB + col) * 2] = best;
                reinterpret_cast<int*>(s_red)[(kk * CB + col) * 2 + 1] = bi;
            }
            __syncthreads();
            if (tid < nbc) {
                const int bb = cblk * CB + tid;
                float best = -3.0e38f;
                int bi = 0;
                for (int r = 0; r < 32; ++r) {
                    const float v = s_red[(r * CB + tid) * 2];
                    const int qi = reinterpret_cast<const int*>(s_red)[(r * CB + tid) * 2 + 1];
                    if (v > best || (v == best && qi < bi)) { best = v; bi = qi; }
                }
                int chosen = bi;
                if (a.mode == 1 && a.uniforms != nullptr) {
                    float total = 0.0f;
                    for (int qi = 0; qi < Qo; ++qi) total += expf(s_p[qi * CB + tid] - best);
                    const float target = a.uniforms[(long)bb * a.Ttot + p + 1] * total;
                    float run = 0.0f;
                    int cand = -1;
                    for (int qi = 0; qi < Qo; ++qi) {
                        run += expf(s_p[qi * CB + tid] - best);
                        if (cand < 0 && run >= target) cand = qi;
                    }
                    if (cand >= 0) chosen = cand;
                }
                const bool gen = p + 1 >= a.t_forced[bb] && p + 1 < a.t_end[bb];
                long long nxt = chosen;
                if (!gen && p + 1 < a.Ttot) {   // teacher forced / finished utterance: the token that is in the buffer
                    nxt = a.samples[(long)bb * a.Ttot + p + 1] % a.Q;
                    if (nxt < 0) nxt += a.Q;
                }
                if (gen && u == 0) a.samples[(long)bb * a.Ttot + p + 1] = chosen;
                for (int j = 0; j + 1 < K; ++j) s_tok[j * CB + tid] = s_tok[(j + 1) * CB + tid];
                s_tok[(K - 1) * CB + tid] = (int)nxt;
            }
            __syncthreads();
        }
        // (only now: the token choice staged the logits in the first rows of s_p, which a small model's tap rows overlap)
        if (WN_DLPF_TAP_PREFETCH && p + 1 < a.p1) fetch_taps(0, p + 1);
        DLPF_STAMP(L + 2, 0);
        if (s_flag[0]) break;
    }
    if (tid == 0 && s_flag[0]) wn_store_coherent_int(a.err, 1);
}

template <int NSP, int NSX>
static constexpr size_t lds_flags() {
    return ((size_t)32 * NSP * 16 + (NSP * 3 <= 160 ? 32 * NSX * 16 : 0) + 4096 + 8 * 16 + 8 * 16 + 4 * 16 + 64) * 4;
}

template <int NSP, int NSX>
static int capacity_flags() {   // as capacity_cls of wn_dlp.hip
    static int cap[WN_COOP_MAXDEV];
    static bool cap_init = false;
    constexpr size_t lds = lds_flags<NSP, NSX>();
#ifndef WN_EMU
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_dlpf<NSP, NSX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return 0;
        attr_set = true;
    }
#endif
    if (wn_coop_capacity_override() >= 0) return wn_coop_capacity_override();
    return wn_coop_capacity_cached(cap, cap_init, k_dlpf<NSP, NSX>, WN_DLP_T, lds);
}

template <int NSP, int NSX>
static int launch_flags(const WnDlpArgs& a, int nblk, wn_stream_t st) {
    if (a.plan.NU * nblk > capacity_flags<NSP, NSX>()) return 4;   // not all workgroups would be resident: no launch
    constexpr size_t lds = lds_flags<NSP, NSX>();
    WN_LAUNCH_COOP((k_dlpf<NSP, NSX>), dim3((unsigned)(a.plan.NU * nblk)), dim3(WN_DLP_T), lds, st, a);
    return 0;
}

int wn_dlpf_covers(const WnDlpPlan* plan) { return plan->ok && plan->wide && plan->RS == 16 && (plan->NSP == 48 || plan->NSP == 64) && plan->NSX == 16; }

int wn_dlpf_capacity(const WnDlpPlan* plan) {
    if (!wn_dlpf_covers(plan)) return 0;
    return plan->NSP == 48 ? capacity_flags<48, 16>() : capacity_flags<64, 16>();
}

int wn_dlpf_launch(const WnDlpArgs* ap, wn_stream_t st) {
    const WnDlpArgs& a = *ap;
    if (!wn_dlpf_covers(&a.plan) || !a.handoff || !a.flags || a.B < 1 || a.B > WN_DLPF_BMAX || a.p1 < a.p0) return 1;
    const int nblk = (a.B + WN_DLPM_CB - 1) / WN_DLPM_CB;
    if (a.Bp != nblk * WN_DLPM_CB || a.plan.NU * nblk > WN_DLPM_MAXWG) return 1;
    if (a.mode != 0 && a.mode != 1) return 2;
    if ((long)2 * 2 * a.R * a.Bp * 4 > 0x7fffffffL || a.qfloats * a.B * 4 > 0xffffffffL) return 1;
    WN_PROF("dlpf_steps", 0.0, 0.0, st);
    return a.plan.NSP == 48 ? launch_flags<48, 16>(a, nblk, st) : launch_flags<64, 16>(a, nblk, st);
}
