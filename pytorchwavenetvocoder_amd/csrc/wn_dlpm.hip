// wn_dlpm.hip -- persistent any-size decode for batches of 9 .. 48 utterances: the algorithm, the stage decomposition, the packed
// weight images, the granule hand-off and the private dilation queues of wn_dlp.hip (reference wavenet.py:355-385, 538-549,
// 518-523) with the dot products on the matrix cores.
//
// wn_dlp.hip multiplies on the fp32 VALU because a matrix tile spends its time on 16 or 32 columns whether they exist or not;
// from ~9 utterances on the columns do exist.  Here a unit owns CG = 8 residual channels (n_resch 512: 64 units) and a stage is
// two row sets of 16 rows -- [8 sigmoid | 8 tanh] over all of K and [8 x | skip rows] over the z part -- times a block of 16
// utterance columns: one v_mfma_f32_16x16x4_f32 tile per set (exact fp32 products, fp32 accumulation).  The 8 waves split K:
// wave w, lane group q = lane / 16 owns the NS consecutive k  [(4 w + q) NS, (4 w + q + 1) NS)  of a set (the SAME image the
// VALU kernel loads: wn_dlp_pack_stage with 16 rows per set), so the i-th tile step of a wave contracts the i-th k of its four
// groups; the 8 partial tiles are added through LDS in wave order.  A batch of up to 48 utterances is up to three column
// blocks, each with its OWN set of units (blocks are independent decodes that share nothing but the weight images): the grid is
// (blocks x units) workgroups, e.g. 3 x 64 = 192 CUs for n_resch 512.
// Inputs are staged in LDS as [k][17] (+ a 16-float shift per k group when NS * 17 is a multiple of 64) so that the four k rows
// a tile step reads fall into four different 16-bank groups.
#include "wn_dlp.h"

#include <type_traits>

#include "wn_prof.h"

typedef unsigned long long u64;

static __device__ __forceinline__ long dlpm_queue_off(int l, int depth, int K, int R) {
    const long cyc = l / depth, in = l % depth;
    return (long)R * (K - 1) * (cyc * ((1L << depth) - 1) + ((1L << in) - 1));
}
#define DLPM_SPIN_MAX (1 << 22)
// Timing builds (-DWN_DLP_TIMING, tools/dlp_timing.py): the stamps of wn_dlp.hip, column block 0 of unit 0, step p0 + 3:
// [stage][phase] 0 stage start, 1 inputs gathered, 2 after the barrier, 3 tiles done, 4 partial tiles in LDS (barrier),
// 5 outputs published, 6 end of the stage
#ifdef WN_DLP_TIMING
#define DLPM_STAMP(stage, ph)                                                                                        \
    do {                                                                                                             \
        if (tid == 0 && u == 0 && p == a.p0 + 3 && (stage) < 40)                                                     \
            reinterpret_cast<long long*>(a.err + 16)[(stage) * 8 + (ph)] = (long long)wall_clock64();                \
    } while (0)
#else
#define DLPM_STAMP(stage, ph)
#endif

template <int NSP, int NSX>
__global__ __launch_bounds__(WN_DLP_T, 2) void k_dlpm(WnDlpArgs a) {
    WN_DYN_SMEM(smem_raw);
    if (wn_load_coherent_int(a.err) != 0) return;   // an earlier launch on this state timed out (or this one already has): nothing to continue from
    constexpr int CG = 8, SL = 32, CB = WN_DLPM_CB, STR = 17, GJ = 16;
    constexpr int KPAD = SL * NSP;
    constexpr int PADQ = ((NSP * STR) % 64 == 0) ? 16 : 0;   // shift of k group (k / NSP) & 3
    static_assert(PADQ == 0 || NSP % (4 * NSX) == 0, "the X set's four k groups must share one shift");
    constexpr int REG0 = KPAD * STR + 64;
    float* s_in = reinterpret_cast<float*>(smem_raw);                 // inputs of the stage [k][STR], utterance column fastest
    float* s_red = s_in + REG0;                                       // partial tiles [2 sets][8 waves][16 rows][16 columns]
    float* s_xown = s_red + 4096;                                     // [8][CB] x of the unit's own channels (previous stage)
    float* s_sk = s_xown + 8 * CB;                                    // [8][CB] skip accumulators of the unit's rows
    int* s_tok = reinterpret_cast<int*>(s_sk + 8 * CB);               // [3][CB] the newest K tokens of the block's utterances
    int* s_flag = s_tok + 4 * CB;                                     // [0] a poll timed out

    const int tid = threadIdx.x, lane = tid & 63, wave = WN_UNIFORM(tid >> 6);
    const int lc = lane & 15, q = lane >> 4;     // tile column / row of this lane, its k group
    const int col = tid & 15, kk = tid >> 4;     // gather and epilogue: utterance column of the block, row index 0 .. 31
    const int R = a.R, S = a.S, L = a.L, K = a.K, B = a.B, Qo = a.Qo;
    // workgroup = (column block, unit): the blocks of 16 utterances are independent decodes that share nothing but the weights
    const int u = (int)blockIdx.x % a.plan.NU, cblk = (int)blockIdx.x / a.plan.NU;
    const int c0 = u * CG;
    const int SU = a.plan.SU, QU = a.plan.QU;
    const int nbc = (B - cblk * CB) < CB ? (B - cblk * CB) : CB;   // utterances of this block
    const int b = cblk * CB + col;
    const bool live = col < nbc;
    float* pq = a.pq + (long)blockIdx.x * a.pq_unit_stride;       // private rings [qfloats][CB]
    auto sin_off = [&](int k) -> int { return k * STR + PADQ * ((k / NSP) & 3); };
    const unsigned lo = (unsigned)(kk * B + b);                     // lane part of an address in a [row][B] array
    const unsigned lq = (unsigned)(kk * CB + col);                  // ... in the private rings
    const long st8 = (long)32 * B * 8, st4 = (long)32 * B * 4, sq4 = (long)32 * CB * 4;   // 32 rows further

    // ---- set-up: private copy of the block's columns of the dilation queues, zero staging, tokens of the context ----
    for (long i0 = 0; i0 < a.qfloats; i0 += 32 * 8) {   // 8 rows per thread in flight
        float v[8];
        WN_UNROLL
        for (int j = 0; j < 8; ++j) {
            const long i = i0 + kk + 32 * j;
            v[j] = (live && i < a.qfloats) ? a.queues[i * B + b] : 0.0f;
        }
        WN_UNROLL
        for (int j = 0; j < 8; ++j) {
            const long i = i0 + kk + 32 * j;
            if (i < a.qfloats) pq[i * CB + col] = v[j];
        }
    }
    for (int i = tid; i < REG0; i += WN_DLP_T) s_in[i] = 0.0f;
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
    // Bounded poll of up to N granules of this thread at once: whatever is not there yet is requested again TOGETHER (a unit
    // that is early finds none of its granules ready -- polled one element after the other that was one memory round trip each:
    // 21.7 us per stage in the first build, profiles/r04/NOTES.md)
    auto poll_all = [&](auto n_c, u64* gv, const char* gb, long stride, unsigned lo8, int j0, int nr, bool on, unsigned tag) {
        constexpr int N = decltype(n_c)::value;
        int spin = 0;
        for (;;) {
            WN_UNROLL
            for (int jj = 0; jj < N; ++jj)
                if (on && j0 + jj < nr && (unsigned)(gv[jj] >> 32) != tag)
                    gv[jj] = wn_granule_load(reinterpret_cast<const u64*>(gb + (j0 + jj) * stride + (size_t)lo8));
            bool all = true;
            WN_UNROLL
            for (int jj = 0; jj < N; ++jj) all = all && ((unsigned)(gv[jj] >> 32) == tag || !(on && j0 + jj < nr));
            if (all) break;
            if (++spin > DLPM_SPIN_MAX || s_flag[0]) { s_flag[0] = 1; break; }
            WN_SLEEP(1);
        }
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
    // One row set times the staged column block: ns tile steps of this wave (two accumulators, added at the end), partial tile
    // into s_red[set][wave]
    auto tile = [&](const auto& w, auto ns_c, int set, bool on) {
        constexpr int ns = decltype(ns_c)::value;
        if (!on) return;
        const float* src = s_in + sin_off((4 * wave + q) * ns) + lc;
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        // the staged values are read 8 tile steps ahead (a bounded number of registers: the weights take 64)
        constexpr int G = 8;
        static_assert(ns % G == 0, "groups of 8 tile steps");
        float bc[G], bn[G];
        WN_UNROLL
        for (int e = 0; e < G; ++e) bc[e] = src[e * STR];
        WN_UNROLL
        for (int i0 = 0; i0 < ns; i0 += G) {
            if (i0 + G < ns) {
                WN_UNROLL
                for (int e = 0; e < G; ++e) bn[e] = src[(i0 + G + e) * STR];
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
        const float* p = s_red + ((set * 8) * 16 + row) * 16 + col;
        float s = 0.0f;
        WN_UNROLL
        for (int w = 0; w < 8; ++w) s += p[w * 256];
        return s;
    };

    float wP[NSP], wX[NSX];
    auto issue_stage_weights = [&](int sn) {   // stage sn in [0, L]
        const float* img = a.wpk + ((long)sn * a.plan.NU + u) * a.plan.stage_floats;
        if (sn < L) load_weights(wP, std::integral_constant<int, NSP>(), img);
        if (sn >= 1) load_weights(wX, std::integral_constant<int, NSX>(), img + 512L * NSP);
    };
    const float* pimg = a.wpost + (long)u * a.plan.post_floats;
    const int nr = R / 32;   // rows kk + 32 j per thread and part (R % 32 == 0: a thread's rows of a part are a constant stride apart)
    issue_stage_weights(0);
    for (int p = a.p0; p < a.p1; ++p) {
        const unsigned tag0 = (unsigned)(p + 1) * (unsigned)(L + 4) + 1u;    // tag of (step p, stage s) = tag0 + s
        for (int s = 0; s <= L; ++s) {
            const bool hasP = s < L, hasX = s >= 1;
            const int d = 1 << (s % a.depth), Dq = (K - 1) * d;
            const long qoff_s = hasP ? dlpm_queue_off(s, a.depth, K, R) : 0;
            const unsigned tag = tag0 + (unsigned)(s - 1);
            DLPM_STAMP(s, 0);
            // (1) gather [z_{s-1} | x_{s-1} | older taps of x_s] into s_in.  A thread owns column `col` and the rows kk + 32 j of
            // every part: an address is (wave-uniform base of the part) + j * (uniform stride) + (one 32-bit lane offset), and
            // nothing but the poll itself is computed per element.
            // older taps of x_s: plain loads from the unit's private rings (always there).  (Moved to the end of the previous
            // stage, before or around its epilogue, they cost what they save here: 385 / 428 us per step against 373.)
            for (int jt = 0; hasP && jt < K - 1; ++jt) {
                int slot = (p - (K - 1 - jt) * d) % Dq;
                if (slot < 0) slot += Dq;
                const char* tb = reinterpret_cast<const char*>(pq + (qoff_s + (long)slot * R) * CB);
                const int kb = 2 * R + jt * R + kk;
                for (int j0 = 0; j0 < nr; j0 += GJ) {
                    float tv[GJ];
                    WN_UNROLL
                    for (int jj = 0; jj < GJ; ++jj) {
                        tv[jj] = 0.0f;
                        if (j0 + jj < nr && live) tv[jj] = *reinterpret_cast<const float*>(tb + (j0 + jj) * sq4 + (size_t)(lq * 4u));
                    }
                    WN_UNROLL
                    for (int jj = 0; jj < GJ; ++jj)
                        if (j0 + jj < nr && live) s_in[sin_off(kb + 32 * (j0 + jj)) + col] = tv[jj];
                }
            }
            // z_{s-1}: granules of the previous stage (stage 0: zeros, its weights there are zero as well)
            {
                const char* gb = reinterpret_cast<const char*>(a.gz + (long)((s - 1) & 1) * R * B);
                for (int j0 = 0; j0 < nr; j0 += GJ) {
                    u64 gv[GJ];
                    WN_UNROLL
                    for (int jj = 0; jj < GJ; ++jj) gv[jj] = s >= 1 ? 0ull : (u64)tag << 32;
                    poll_all(std::integral_constant<int, GJ>(), gv, gb, st8, lo * 8u, j0, nr, live, tag);
                    WN_UNROLL
                    for (int jj = 0; jj < GJ; ++jj)
                        if (j0 + jj < nr && live) s_in[sin_off(kk + 32 * (j0 + jj)) + col] = wn_bits_f32((unsigned)gv[jj]);
                }
            }
            // x_{s-1}: granules from stage 2 on, every unit's own gather of the front conv before; it also goes into the unit's
            // own ring of layer s-1.  (z and x polled together -- 64 registers of granules per thread -- spills: measured slower)
            {
                const char* gb = reinterpret_cast<const char*>(a.gx + (long)((s - 1) & 1) * R * B);
                const int dp = 1 << ((s >= 1 ? s - 1 : 0) % a.depth), Dp = (K - 1) * dp;
                char* rb = reinterpret_cast<char*>(pq + (dlpm_queue_off(s >= 1 ? s - 1 : 0, a.depth, K, R) + (long)(p % Dp) * R) * CB);
                for (int j0 = 0; j0 < nr; j0 += GJ) {
                    if (s <= 1) {   // x_0 = x0_of(row, col), eight rows' table reads in flight (one row after the other these two
                                    // stages took 18 - 20 us instead of 11)
                        int tk[3];
                        WN_UNROLL
                        for (int k = 0; k < 3; ++k) tk[k] = k < K ? s_tok[k * CB + col] : -1;
                        for (int j1 = 0; j1 < GJ; j1 += 8) {
                            float bv[8], wv[8][3];
                            WN_UNROLL
                            for (int jj = 0; jj < 8; ++jj) {
                                const int c = kk + 32 * (j0 + j1 + jj);
                                const bool ok = j0 + j1 + jj < nr && live;
                                bv[jj] = ok ? a.params[a.off_causal_b + c] : 0.0f;
                                WN_UNROLL
                                for (int k = 0; k < 3; ++k)
                                    wv[jj][k] = (ok && tk[k] >= 0) ? a.params[a.off_causal_w + ((long)c * a.Q + tk[k]) * K + k] : 0.0f;
                            }
                            WN_UNROLL
                            for (int jj = 0; jj < 8; ++jj) {
                                if (j0 + j1 + jj < nr && live) {
                                    float v = bv[jj];
                                    WN_UNROLL
                                    for (int k = 0; k < 3; ++k)
                                        if (tk[k] >= 0) v += wv[jj][k];   // the order of x0_of
                                    if (s == 1) *reinterpret_cast<float*>(rb + (j0 + j1 + jj) * sq4 + (size_t)(lq * 4u)) = v;
                                    s_in[sin_off(R + kk + 32 * (j0 + j1 + jj)) + col] = v;
                                }
                            }
                        }
                    } else {
                        u64 gv[GJ];
                        WN_UNROLL
                        for (int jj = 0; jj < GJ; ++jj) gv[jj] = 0ull;
                        poll_all(std::integral_constant<int, GJ>(), gv, gb, st8, lo * 8u, j0, nr, live, tag);
                        WN_UNROLL
                        for (int jj = 0; jj < GJ; ++jj) {
                            if (j0 + jj < nr && live) {
                                const float v = wn_bits_f32((unsigned)gv[jj]);
                                *reinterpret_cast<float*>(rb + (j0 + jj) * sq4 + (size_t)(lq * 4u)) = v;
                                if (hasP) s_in[sin_off(R + kk + 32 * (j0 + jj)) + col] = v;
                            }
                        }
                    }
                }
            }
            DLPM_STAMP(s, 1);
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
            __syncthreads();
            DLPM_STAMP(s, 2);
            // (3) the two row sets on the matrix cores
            tile(wP, std::integral_constant<int, NSP>(), 0, hasP);
            tile(wX, std::integral_constant<int, NSX>(), 1, hasX);
            DLPM_STAMP(s, 3);
            __syncthreads();
            DLPM_STAMP(s, 4);
            // (4) sums of the 8 partial tiles and the outputs' epilogues
            if (live) {
                if (kk < CG) {
                    if (hasP) {   // gate (wavenet.py:542-544)
                        const float sg = row_sum(0, kk), st = row_sum(0, CG + kk);
                        wn_granule_store(a.gz + ((long)(s & 1) * R + c0 + kk) * B + b, wn_sigmoid(sg + e0) * wn_tanh(st + e1),
                                         tag0 + (unsigned)s);
                    }
                } else if (kk < 2 * CG) {
                    if (s < L) {
                        const int c = kk - CG;
                        float xs;
                        if (s == 0) {   // x_0 of the unit's own channels
                            xs = e0;
                        } else {        // x_s = res_1x1(z_{s-1}) + x_{s-1}   (wavenet.py:546-548)
                            xs = row_sum(1, c) + e0 + s_xown[c * CB + col];
                            wn_granule_store(a.gx + ((long)(s & 1) * R + c0 + c) * B + b, xs, tag0 + (unsigned)s);
                        }
                        s_xown[c * CB + col] = xs;
                        if (K >= 2)   // the shared rings stay current for the next launch (and the launch path)
                            a.queues[(dlpm_queue_off(s, a.depth, K, R) + (long)(p % Dq) * R + c0 + c) * B + b] = xs;
                    }
                } else if (kk < 2 * CG + SU) {
                    if (hasX) s_sk[(kk - 2 * CG) * CB + col] += row_sum(1, CG + (kk - 2 * CG));   // skip sum (wavenet.py:545, 365)
                }
            }
            DLPM_STAMP(s, 5);
            // The next weights (next stage, or the post net's first set), AFTER the stage's outputs have left: 128 KB of
            // requests take the CU's memory pipe ~1 us to issue, and whatever is requested behind them returns behind them
            if (s < L) issue_stage_weights(s + 1);
            else load_weights(wX, std::integral_constant<int, NSX>(), pimg);
            DLPM_STAMP(s, 6);
        }
        DLPM_STAMP(L + 1, 0);
        __syncthreads();   // the skip accumulators of the last stage's epilogue are complete
        // ---- post net (wavenet.py:518-523): relu(skip sum) -> conv_post_1 + relu -> conv_post_2, three more hops ----
        for (int i = tid; i < SU * nbc; i += WN_DLP_T) {
            const int r = i / nbc, c = i % nbc, row = u * SU + r;
            if (row < S)
                wn_granule_store(a.gs + (long)row * B + cblk * CB + c, fmaxf(s_sk[r * CB + c] + a.bskip[row], 0.0f), tag0 + (unsigned)(L + 1));
            s_sk[r * CB + c] = 0.0f;
        }
        for (int stage = 0; stage < 2; ++stage) {
            const u64* src = stage == 0 ? a.gs : a.go;
            const unsigned ptag = tag0 + (unsigned)(L + 1 + stage);
            for (int j0 = 0; j0 < S / 32; j0 += 8) {
                u64 gv[8];
                WN_UNROLL
                for (int jj = 0; jj < 8; ++jj) gv[jj] = 0ull;
                poll_all(std::integral_constant<int, 8>(), gv, reinterpret_cast<const char*>(src), st8, lo * 8u, j0, S / 32, live, ptag);
                WN_UNROLL
                for (int jj = 0; jj < 8; ++jj)
                    if (live && j0 + jj < S / 32) s_in[sin_off(kk + 32 * (j0 + jj)) + col] = wn_bits_f32((unsigned)gv[jj]);
            }
            float pb = 0.0f;   // the output row's bias
            {
                const int row = u * (stage == 0 ? SU : QU) + kk;
                if (live && (stage == 0 ? (kk < SU && row < S) : (kk < QU && row < Qo)))
                    pb = a.params[(stage == 0 ? a.off_post1_b : a.off_post2_b) + row];
            }
            __syncthreads();
            tile(wX, std::integral_constant<int, NSX>(), 0, true);
            __syncthreads();
            if (live && kk < 16) {
                if (stage == 0) {
                    const int row = u * SU + kk;
                    if (kk < SU && row < S) wn_granule_store(a.go + (long)row * B + b, fmaxf(row_sum(0, kk) + pb, 0.0f), tag0 + (unsigned)(L + 2));
                } else {
                    const int row = u * QU + kk;
                    if (kk < QU && row < Qo) wn_granule_store(a.gl + (long)row * B + b, row_sum(0, kk) + pb, tag0 + (unsigned)(L + 3));
                }
            }
            if (stage == 0) load_weights(wX, std::integral_constant<int, NSX>(), pimg + 512L * NSX);
            else if (p + 1 < a.p1) issue_stage_weights(0);
        }
        // ---- token choice, by every unit for itself (wavenet.py:371-381): first-max argmax or inverse CDF on the caller's draw ----
        {
            const int nq = (Qo - kk + 31) / 32;   // logit rows kk + 32 j < Qo of this thread
            for (int j0 = 0; j0 < nq; j0 += 8) {
                u64 gv[8];
                WN_UNROLL
                for (int jj = 0; jj < 8; ++jj) gv[jj] = 0ull;
                poll_all(std::integral_constant<int, 8>(), gv, reinterpret_cast<const char*>(a.gl), st8, lo * 8u, j0, nq, live,
                         tag0 + (unsigned)(L + 3));
                WN_UNROLL
                for (int jj = 0; jj < 8; ++jj)
                    if (live && j0 + jj < nq) s_in[sin_off(kk + 32 * (j0 + jj)) + col] = wn_bits_f32((unsigned)gv[jj]);
            }
            __syncthreads();
            // first-max argmax: a thread scans its own logit rows kk + 32 j (ascending), the 32 candidates of a column are
            // compared by one thread (larger value, then smaller index); unit 0 writes the logits out, a thread its own rows
            {
                float best = -3.0e38f;
                int bi = 0x7fffffff;
                for (int j = 0; j < nq; ++j) {
                    const int qi = kk + 32 * j;
                    const float v = live ? s_in[sin_off(qi) + col] : -3.0e38f;
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
                    for (int qi = 0; qi < Qo; ++qi) total += expf(s_in[sin_off(qi) + tid] - best);
                    const float target = a.uniforms[(long)bb * a.Ttot + p + 1] * total;
                    float run = 0.0f;
                    int cand = -1;
                    for (int qi = 0; qi < Qo; ++qi) {
                        run += expf(s_in[sin_off(qi) + tid] - best);
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
        DLPM_STAMP(L + 2, 0);
        if (s_flag[0]) break;
    }
    if (tid == 0 && s_flag[0]) wn_store_coherent_int(a.err, 1);
}

template <int NSP, int NSX>
static int capacity_wide(long lds_bytes) {   // as capacity_cls of wn_dlp.hip
    static int cap[WN_COOP_MAXDEV];
    static bool cap_init = false;
#ifndef WN_EMU
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_dlpm<NSP, NSX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return 0;
        attr_set = true;
    }
#endif
    if (wn_coop_capacity_override() >= 0) return wn_coop_capacity_override();
    return wn_coop_capacity_cached(cap, cap_init, k_dlpm<NSP, NSX>, WN_DLP_T, (size_t)lds_bytes);
}

template <int NSP, int NSX>
static int launch_wide(const WnDlpArgs& a, wn_stream_t st) {
    const int nblk = (a.B + WN_DLPM_CB - 1) / WN_DLPM_CB;
    if (a.plan.NU * nblk > capacity_wide<NSP, NSX>(a.plan.lds_bytes)) return 4;   // not all workgroups would be resident: no launch
    WN_LAUNCH_COOP((k_dlpm<NSP, NSX>), dim3((unsigned)(a.plan.NU * nblk)), dim3(WN_DLP_T), (size_t)a.plan.lds_bytes, st, a);
    return 0;
}

int wn_dlpm_capacity(const WnDlpPlan* plan) {
    if (!plan->ok || !plan->wide || plan->RS != 16) return 0;
    switch (plan->cls) {
        case 2: return capacity_wide<48, 16>(plan->lds_bytes);
        case 3: return capacity_wide<64, 16>(plan->lds_bytes);
    }
    return 0;
}

int wn_dlpm_launch(const WnDlpArgs* ap, wn_stream_t st) {
    const WnDlpArgs& a = *ap;
    if (!a.plan.ok || !a.plan.wide || a.plan.RS != 16 || a.B < 1 || a.B > WN_DLPM_BMAX || a.p1 < a.p0) return 1;
    if (a.plan.NU * ((a.B + WN_DLPM_CB - 1) / WN_DLPM_CB) > WN_DLPM_MAXWG) return 1;   // all workgroups resident at once
    if (a.mode != 0 && a.mode != 1) return 2;
    WN_PROF("dlpm_steps", 0.0, 0.0, st);
    switch (a.plan.cls) {
        case 2: return launch_wide<48, 16>(a, st);
        case 3: return launch_wide<64, 16>(a, st);
    }
    return 1;
}
