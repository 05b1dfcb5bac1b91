// wn_dlp.hip -- persistent any-size decode: the reference's queue algorithm (wavenet.py:355-385 fast_generate's step loop,
// :538-549 _generate_residual_forward, :518-523 _postprocess) for models the one-workgroup kernel cannot hold, as ONE
// launch per chunk of steps.  See wn_dlp.h.
//
// Decomposition.  Unit u (one workgroup of 512 threads on its own CU) owns CG residual channels -- CG = 4 for n_resch up to
// 960 (n_resch 512: 128 units), 8 / 16 beyond: the weight stream of a stage is what ONE CU can pull (~60 GB/s measured), so
// the fewer channels a unit owns, the shorter the stage.  Folding the res 1x1 of layer s-1 into the newest tap of layer s,
//      M_s = Wd_new(s) . Wres(s-1),      c_s = cvec_s + Wd_new(s) . b_res(s-1),
// makes layer s ONE dependent stage: from (z_{s-1}, x_{s-1}) -- both complete vectors of the previous stage -- a unit computes
//      gate rows:  P = M_s z_{s-1} + Wd_new(s) x_{s-1} + sum_older-taps Wd_tap(s) x_s[t - ..] + aux + c_s,   z_s = sigmoid(P_a) tanh(P_b)
//      x rows:     x_s = Wres(s-1) z_{s-1} + b_res(s-1) + x_{s-1}
//      skip rows:  skip += Wskip(s-1) z_{s-1}                      (its share of the n_skipch rows)
// as two sets of RS = 2 CG rows.  A lane owns one row and one slice of consecutive k of a set (the weights sit in its registers,
// 16-byte non-temporal loads) and multiplies them with the staged input vectors on the fp32 VALU -- only the utterances
// that exist are computed: an f32 MFMA tile spends 64 cycles per 2 k on 32 columns, which measured 546 us per step at B = 1
// (profiles/r04/NOTES.md) --, the partial sums of a row are added through LDS by 8 lanes in a fixed butterfly order.
// The previous stage's vectors arrive as 8-byte granules {value, tag} that the consumer polls (tag = step and stage, so a
// granule is its own ready flag; measured 0.7 - 0.8 us per hop with 0 stale words, same or other XCD); the older taps come
// from the unit's PRIVATE copy of the dilation queues (every unit sees every x_s anyway and pushes it into its own rings), so
// nothing but granules crosses workgroups: no grid barrier, no fence.  The next stage's weights are requested right after a
// stage's dot products (their registers are dead): they stream under the epilogue, the hand-off and the next gather;
// everything the epilogue reads from memory is requested before them (memory returns in order).
// After the last layer: skip sum -> conv_post_1 -> conv_post_2 as three more stages, then EVERY unit picks the token itself
// from the gathered logits (argmax / inverse-CDF on the caller's uniforms: deterministic), so the next step starts without
// another hop.  Every poll is bounded; a timeout sets `err` and drains the launch.
// Measured, n_resch 512 / n_skipch 256 (profiles/r04/recipe_decode_probe.txt, dlp_timing_b1.txt): 190 us per step for one
// utterance (the 66 layer-wise launches: 620), 342 for four; a stage is 5.4 us, 3 of them the hand-off + poll.
#include "wn_dlp.h"

#include <type_traits>

#include "wn_prof.h"

typedef unsigned long long u64;

void wn_dlp_make_plan(int Q, int Qo, int R, int S, int L, int K, int wide, WnDlpPlan* p) {
    p->ok = 0;
    p->wide = wide;
    if (R < 32 || R % 32 != 0 || S % 16 != 0 || K < 2 || K > 3 || L < 1 || Q < 2) return;
    // As many units as the chip has room for: the weight stream of a stage is what one CU can pull (~60 GB/s measured), so
    // the fewer channels a unit owns the shorter the stage.  Channels per unit CG in {4, 8, 16} <-> rows per set RS = 2 CG.
    static const int cls[][3] = {{8, 24, 8}, {8, 32, 8}, {16, 48, 16}, {16, 64, 16}, {32, 96, 32}, {32, 128, 32}};   // RS, NSP, NSX
    for (int c = 0; c < 6; ++c) {
        const int RS = cls[c][0], CG = RS / 2, KQ = 64 / RS, slices = 8 * KQ;
        if (wide && (RS != 16 || S % 32 != 0)) continue;   // k_dlpm: one 16x16x4 tile per row set
        if (R % CG != 0) continue;
        const int NU = R / CG;
        if (NU > 240) continue;                    // one workgroup per CU, all resident
        const int SU = (S + NU - 1) / NU, QU = (Qo + NU - 1) / NU, KP = (K + 1) * R;
        if (CG + SU > RS || QU > RS || SU > RS) continue;
        const int kpad = slices * cls[c][1], xpad = slices * cls[c][2];
        if (KP > kpad || R > xpad || S > xpad || S > kpad || Qo > kpad) continue;
        p->cls = c; p->RS = RS; p->NSP = cls[c][1]; p->NSX = cls[c][2];
        p->NU = NU; p->SU = SU; p->QU = QU; p->KP = KP;
        p->stage_floats = 512L * (p->NSP + p->NSX);
        p->post_floats = 2L * 512 * p->NSX;
        const long region0 = (long)kpad * WN_DLP_CB + 2L * 8 * 64 * WN_DLP_CB;   // input staging [CB][kpad] + partial sums
        p->lds_bytes = (region0 + 16 * WN_DLP_BMAX + 16 * WN_DLP_BMAX + 4 * WN_DLP_BMAX + 64) * 4;
        if (wide)   // k_dlpm: inputs [kpad][17] (+ 64), partial tiles [2][8][16][16], x / skip [8][16] each, tokens, flag
            p->lds_bytes = ((long)kpad * 17 + 64 + 4096 + 8 * WN_DLPM_CB + 8 * WN_DLPM_CB + 4 * WN_DLPM_CB + 64) * 4;
        p->ok = 1;
        return;
    }
}

// ---------------------------------------------------------------------------------------------
// packing (once per decode call)
// ---------------------------------------------------------------------------------------------
__global__ void k_dlp_pack_stage(WnDlpPackArgs a) {
    const int NSP = a.plan.NSP, NSX = a.plan.NSX, R = a.R, K = a.K;
    const long per_unit = a.plan.stage_floats;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= per_unit * a.plan.NU) return;
    const int u = (int)(idx / per_unit);
    long r = idx % per_unit;
    const bool isP = r < 512L * NSP;
    if (!isP) r -= 512L * NSP;
    const int NS = isP ? NSP : NSX;
    // a lane's weights travel as 16-byte loads: [wave][t / 4][lane][t % 4]
    const int lane = (int)((r >> 2) & 63);
    const int t = (int)(((r >> 8) % (NS / 4)) * 4 + (r & 3)), w = (int)((r >> 8) / (NS / 4));
    const int RS = a.plan.RS, CG = RS / 2, KQ = 64 / RS;
    const int row = lane % RS, k = (w * KQ + lane / RS) * NS + t;   // a (wave, k part) owns NS consecutive k
    const int s = a.stage;
    float v = 0.0f;
    if (isP) {
        if (s < a.L && k < a.plan.KP) {
            const int c = u * CG + (row % CG), half = row / CG;
            const long wbase = a.lb_s + (half ? a.o_dtanh_w : a.o_dsig_w);
            if (k < R) v = (s >= 1) ? a.fold[((long)half * R + c) * R + k] : 0.0f;
            else if (k < 2 * R) v = a.params[wbase + ((long)c * R + (k - R)) * K + (K - 1)];
            else {
                const int j = (k - 2 * R) / R, i = (k - 2 * R) % R;
                v = a.params[wbase + ((long)c * R + i) * K + j];
            }
        }
    } else if (s >= 1 && k < R) {
        if (row < CG) {
            if (s < a.L) v = a.params[a.lb_prev + a.o_res_w + (long)(u * CG + row) * R + k];
        } else if (row < CG + a.plan.SU) {
            const int srow = u * a.plan.SU + (row - CG);
            if (srow < a.S) v = a.params[a.skip_prev + (long)srow * R + k];
        }
    }
    a.dst[idx] = v;
}

int wn_dlp_pack_stage(const WnDlpPackArgs* a, wn_stream_t st) {
    WN_PROF("dlp_pack", 0.0, 0.0, st);
    const long n = a->plan.stage_floats * a->plan.NU;
    WN_LAUNCH(k_dlp_pack_stage, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *a);
    return 0;
}

__global__ void k_dlp_pack_post(const float* params, long post1_w, long post2_w, int S, int Qo, WnDlpPlan plan, float* dst) {
    const int NSX = plan.NSX;
    const long per_unit = plan.post_floats;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= per_unit * plan.NU) return;
    const int u = (int)(idx / per_unit);
    long r = idx % per_unit;
    const int tile = (int)(r / (512L * NSX));
    r -= (long)tile * 512 * NSX;
    const int lane = (int)((r >> 2) & 63);
    const int t = (int)(((r >> 8) % (NSX / 4)) * 4 + (r & 3)), w = (int)((r >> 8) / (NSX / 4));
    const int RS = plan.RS, KQ = 64 / RS;
    const int row = lane % RS, k = (w * KQ + lane / RS) * NSX + t;
    float v = 0.0f;
    if (k < S) {
        if (tile == 0) {
            const int o = u * plan.SU + row;
            if (row < plan.SU && o < S) v = params[post1_w + (long)o * S + k];
        } else {
            const int o = u * plan.QU + row;
            if (row < plan.QU && o < Qo) v = params[post2_w + (long)o * S + k];
        }
    }
    dst[idx] = v;
}

int wn_dlp_pack_post(const float* params, long post1_w, long post2_w, int S, int Qo, const WnDlpPlan* plan, float* dst, wn_stream_t st) {
    WN_PROF("dlp_pack", 0.0, 0.0, st);
    const long n = plan->post_floats * plan->NU;
    WN_LAUNCH(k_dlp_pack_post, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, params, post1_w, post2_w, S, Qo, *plan, dst);
    return 0;
}

// wd_f[l][(tap*R + i)*2R + o'] (wn_api.hip pack_weights): newest tap = K-1
__global__ void k_dlp_cfold(const float* params, const float* cvec, const float* wd_f, long lb0, long lstep, long o_res_b, int L, int R,
                            int K, float* cfold) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)L * 2 * R) return;
    const int s = (int)(idx / (2 * R)), o = (int)(idx % (2 * R));
    float v = cvec[idx];
    if (s >= 1) {
        const float* wn = wd_f + (long)s * K * R * 2 * R + (long)(K - 1) * R * 2 * R;
        const float* b = params + lb0 + (long)(s - 1) * lstep + o_res_b;
        float acc = 0.0f;
        for (int j = 0; j < R; ++j) acc = fmaf(wn[(long)j * 2 * R + o], b[j], acc);
        v += acc;
    }
    cfold[idx] = v;
}

int wn_dlp_cfold(const float* params, const float* cvec, const float* wd_f, long lb0, long lstep, long o_res_b, int L, int R, int K,
                 float* cfold, wn_stream_t st) {
    WN_PROF("dlp_pack", 0.0, 0.0, st);
    const long n = (long)L * 2 * R;
    WN_LAUNCH(k_dlp_cfold, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, params, cvec, wd_f, lb0, lstep, o_res_b, L, R, K, cfold);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// the step kernel
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ long dlp_queue_off(int l, int depth, int K, int R) {
    const long cyc = l / depth, in = l % depth;
    return (long)R * (K - 1) * (cyc * ((1L << depth) - 1) + ((1L << in) - 1));
}
#define DLP_SPIN_MAX (1 << 22)
// Timing builds (-DWN_DLP_TIMING, tools/dlp_timing.py): wall-clock stamps (100 MHz) of unit 0's phases, step p0 + 3, into the
// words behind the error flag: [stage][phase] 0 stage start, 1 inputs gathered, 2 after the barrier, 3 dot products done
// (the stage's weights have landed), 4 partial sums in LDS, 5 outputs published, 6 end of the stage
#ifdef WN_DLP_TIMING
#define DLP_STAMP(stage, ph)                                                                                         \
    do {                                                                                                             \
        if (tid == 0 && u == 0 && p == a.p0 + 3 && (stage) < 40)                                                     \
            reinterpret_cast<long long*>(a.err + 16)[(stage) * 8 + (ph)] = (long long)wall_clock64();                \
    } while (0)
#else
#define DLP_STAMP(stage, ph)
#endif

template <int RS, int NSP, int NSX>
__global__ __launch_bounds__(WN_DLP_T, 2) void k_dlp(WnDlpArgs a) {
    WN_DYN_SMEM(smem_raw);
    if (wn_load_coherent_int(a.err) != 0) return;   // an earlier launch on this state timed out (or this one already has): nothing to continue from
    constexpr int CB = WN_DLP_CB, BM = WN_DLP_BMAX;
    constexpr int CG = RS / 2, KQ = 64 / RS, SL = 8 * KQ;             // channels per unit, k parts per wave, k slices per row
    constexpr int NPASS = ((3 * CG + RS / 2) * CB * 8 + WN_DLP_T - 1) / WN_DLP_T;   // passes of 512 lanes over the row sums (8 lanes each)
    constexpr int GB = RS >= 32 ? 4 : 8;                             // elements a thread requests per gather round
    constexpr int KPAD = SL * NSP;                                    // padded K of the gate rows (length of a staged input vector)
    constexpr int RED = 2 * SL * RS * CB;                             // partial sums [2 row sets][slice][row][CB]
    constexpr int REG0 = KPAD * CB;
    float* s_in = reinterpret_cast<float*>(smem_raw);                 // [CB][KPAD] inputs of the stage, k contiguous per utterance
    float* s_red = s_in + REG0;                                       // the partial sums (their own region: no barrier between a
                                                                      // stage's epilogue and the next stage's gather)
    float* s_xown = s_red + RED;                                      // [16][BM] x of the unit's own channels (previous stage)
    float* s_sk = s_xown + 16 * BM;                                   // [16][BM] skip accumulators of the unit's rows
    int* s_tok = reinterpret_cast<int*>(s_sk + 16 * BM);              // [3][BM] the newest K tokens of every utterance
    int* s_flag = s_tok + 4 * BM;                                     // [0] a poll timed out

    const int tid = threadIdx.x, lane = tid & 63, wave = WN_UNIFORM(tid >> 6);
    const int lr = lane % RS, kq = lane / RS;   // this lane's row of a set and its k part of the wave's slice
    const int u = blockIdx.x;
    const int R = a.R, S = a.S, L = a.L, K = a.K, B = a.B, Qo = a.Qo;
    const int c0 = u * CG;
    const int SU = a.plan.SU, QU = a.plan.QU, KP = a.plan.KP;
    const int ncb = (B + CB - 1) / CB;
    float* pq = a.pq + (long)u * a.pq_unit_stride;

    // ---- set-up: private copy of the dilation queues, zero the staging rows beyond K, tokens of the context ----
    {   // (16-byte copies, 8 per thread in flight: 200 MB per unit at the recipes' size and 32 utterances)
        const wn_f4* q4 = reinterpret_cast<const wn_f4*>(a.queues);
        wn_f4* p4 = reinterpret_cast<wn_f4*>(pq);
        const long n4 = a.qfloats * B / 4;   // qfloats is a multiple of R, R of 32
        for (long i = tid; i < n4; i += 8 * WN_DLP_T) {
            wn_f4 v[8];
            WN_UNROLL
            for (int j = 0; j < 8; ++j)
                if (i + j * WN_DLP_T < n4) v[j] = q4[i + j * WN_DLP_T];
            WN_UNROLL
            for (int j = 0; j < 8; ++j)
                if (i + j * WN_DLP_T < n4) p4[i + j * WN_DLP_T] = v[j];
        }
    }
    for (int i = tid; i < REG0; i += WN_DLP_T) s_in[i] = 0.0f;
    for (int i = tid; i < 16 * BM; i += WN_DLP_T) { s_xown[i] = 0.0f; s_sk[i] = 0.0f; }
    for (int i = tid; i < K * B; i += WN_DLP_T) {
        const int j = i / B, b = i % B;
        const long pos = (long)a.p0 - (K - 1 - j);
        long long tok = pos >= 0 ? a.samples[(long)b * a.Ttot + pos] % a.Q : 0;
        if (tok < 0) tok += a.Q;
        s_tok[j * BM + b] = pos >= 0 ? (int)tok : -1;
    }
    if (tid == 0) s_flag[0] = 0;
    __syncthreads();

    // x_0[c][b] of the current step: front conv as a gather of K weight columns (wavenet.py:355-356, 513-516)
    auto x0_of = [&](int c, int b) -> float {
        float v = a.params[a.off_causal_b + c];
        for (int k = 0; k < K; ++k) {
            const int tok = s_tok[k * BM + b];
            if (tok >= 0) v += a.params[a.off_causal_w + ((long)c * a.Q + tok) * K + k];
        }
        return v;
    };
    // Bounded poll of the N granules a thread has requested: whatever is not there yet is requested again TOGETHER.  (A unit
    // that is early finds none of its granules ready; polled one after the other that was one memory round trip per element:
    // 21.7 us per stage in the first build of wn_dlpm.hip, profiles/r04/NOTES.md.)
    auto poll_all = [&](auto n_c, const u64* const* gp, u64* gv, unsigned tag) {
        constexpr int N = decltype(n_c)::value;
        int spin = 0;
#ifdef WN_DLP_SEQ_POLL   // A/B: one element after the other (the round-4 first build)
        WN_UNROLL
        for (int j = 0; j < N; ++j) {
            if (gp[j] == nullptr) continue;
            while ((unsigned)(gv[j] >> 32) != tag) {
                if (++spin > DLP_SPIN_MAX || s_flag[0]) { s_flag[0] = 1; break; }
                WN_SLEEP(1);
                gv[j] = wn_granule_load(gp[j]);
            }
        }
        return;
#endif
        for (;;) {
            bool all = true;
            WN_UNROLL
            for (int j = 0; j < N; ++j) all = all && (gp[j] == nullptr || (unsigned)(gv[j] >> 32) == tag);
            if (all) break;
            if (++spin > DLP_SPIN_MAX || s_flag[0]) { s_flag[0] = 1; break; }
            WN_SLEEP(1);
            WN_UNROLL
            for (int j = 0; j < N; ++j)
                if (gp[j] != nullptr && (unsigned)(gv[j] >> 32) != tag) gv[j] = wn_granule_load(gp[j]);
        }
    };
    // gather of a complete vector (post net, logits) into s_in[uc][k]: 4 granules per thread in flight
    auto gather_vec = [&](const u64* src, int rows, int nbc, int cb, unsigned tag) {
        for (int base = tid; base < rows * nbc; base += 4 * WN_DLP_T) {
            const u64* gp[4];
            u64 gv[4];
            WN_UNROLL
            for (int j = 0; j < 4; ++j) {
                const int idx = base + j * WN_DLP_T;
                gp[j] = idx < rows * nbc ? src + (long)(idx / nbc) * B + cb * CB + idx % nbc : nullptr;
                gv[j] = 0;
                if (gp[j]) gv[j] = wn_granule_load(gp[j]);
            }
            poll_all(std::integral_constant<int, 4>(), gp, gv, tag);
            WN_UNROLL
            for (int j = 0; j < 4; ++j) {
                const int idx = base + j * WN_DLP_T;
                if (gp[j]) s_in[(idx % nbc) * KPAD + idx / nbc] = wn_bits_f32((unsigned)gv[j]);
            }
        }
    };
    // a wave's slice of a row set's weights: NS / 4 non-temporal 16-byte loads per lane (every weight is read once per step)
    auto load_weights = [&](auto& w, auto ns_c, const float* img) {
        constexpr int ns = decltype(ns_c)::value;
        const wn_f4* src = reinterpret_cast<const wn_f4*>(img) + (long)wave * (ns / 4) * 64 + lane;
        WN_UNROLL
        for (int t4 = 0; t4 < ns / 4; ++t4) {
            const wn_f4 v = wn_ld4_stream(src + (long)t4 * 64);
            w[4 * t4] = v.x; w[4 * t4 + 1] = v.y; w[4 * t4 + 2] = v.z; w[4 * t4 + 3] = v.w;
        }
    };
    // One row set: lane (row lr, k part kq) of wave w holds the weights W[lr][(KQ w + kq) NS + t], t < NS -- NS consecutive k --
    // and accumulates its slice of the dot products of row li with the staged inputs of the block's utterances on the fp32
    // VALU (one 16-byte LDS read = 4 consecutive k of one utterance; only the utterances that exist are computed); 8 KQ
    // partial sums per row go to s_red.
    auto partial_dots = [&](const auto& w, auto ns_c, bool on, int nb, float (&acc)[CB]) {
        constexpr int ns = decltype(ns_c)::value;
        WN_UNROLL
        for (int q = 0; q < CB; ++q) acc[q] = 0.0f;
        if (!on) return;
        const float* src = s_in + (wave * KQ + kq) * ns;
        WN_UNROLL
        for (int q = 0; q < CB; ++q) {
            if (q < nb) {   // block-uniform
                float a0 = 0.0f, a1 = 0.0f;
                WN_UNROLL
                for (int t4 = 0; t4 < ns / 4; ++t4) {
                    const wn_f4 v = *reinterpret_cast<const wn_f4*>(src + q * KPAD + 4 * t4);
                    a0 = fmaf(w[4 * t4], v.x, a0); a1 = fmaf(w[4 * t4 + 1], v.y, a1);
                    a0 = fmaf(w[4 * t4 + 2], v.z, a0); a1 = fmaf(w[4 * t4 + 3], v.w, a1);
                }
                acc[q] = a0 + a1;
            }
        }
    };
    auto put_partials = [&](int set, const float (&acc)[CB]) {
        wn_f4 v;
        v.x = acc[0]; v.y = acc[1]; v.z = acc[2]; v.w = acc[3];
        *reinterpret_cast<wn_f4*>(s_red + ((set * SL + wave * KQ + kq) * RS + lr) * CB) = v;
    };
    static_assert(CB == 4, "one 16-byte LDS read per k");

    // The weight registers of a stage are dead the moment its dot products are done, so the NEXT stage's weights are
    // requested right there: they stream under the partial sums, the epilogue, the hand-off of the outputs and the gather of
    // the next stage (memory returns in order, so the gather's polls simply come back behind them).  For that to cost the
    // epilogue nothing, everything the epilogue reads from memory (aux projection, constants, biases) is requested BEFORE the
    // dot products -- those loads are ahead of the weight stream in the return order.
    float wP[NSP], wX[NSX];
    auto issue_stage_weights = [&](int sn) {   // stage sn in [0, L]
        const float* img = a.wpk + ((long)sn * a.plan.NU + u) * a.plan.stage_floats;
        if (sn < L) load_weights(wP, std::integral_constant<int, NSP>(), img);
        if (sn >= 1) load_weights(wX, std::integral_constant<int, NSX>(), img + 512L * NSP);
    };
    const float* pimg = a.wpost + (long)u * a.plan.post_floats;
    issue_stage_weights(0);
    for (int p = a.p0; p < a.p1; ++p) {
        const unsigned tag0 = (unsigned)(p + 1) * (unsigned)(L + 4) + 1u;    // tag of (step p, stage s) = tag0 + s
        for (int s = 0; s <= L; ++s) {
            const bool hasP = s < L, hasX = s >= 1;
            DLP_STAMP(s, 0);
            const int d = 1 << (s % a.depth), Dq = (K - 1) * d;
            const long qoff_s = hasP ? dlp_queue_off(s, a.depth, K, R) : 0;
            for (int cb = 0; cb < ncb; ++cb) {
                // (2) gather [z_{s-1} | x_{s-1} | older taps of x_s] of the block's utterances into s_in[k][uc]: every thread
                // requests up to GB elements before it looks at the first tag (one round trip for most of the gather)
                const int krows = hasP ? KP : 2 * R;       // stage L: z for the skip rows, x only for the queue push
                const int nbc = (B - cb * CB) < CB ? (B - cb * CB) : CB;   // utterances of this block
                for (int base = tid; base < krows * nbc; base += GB * WN_DLP_T) {
                    const u64* gp[GB];
                    u64 gv[GB];
                    float fv[GB];
                    WN_UNROLL
                    for (int j = 0; j < GB; ++j) {
                        const int idx = base + j * WN_DLP_T;
                        gp[j] = nullptr;
                        fv[j] = 0.0f;
                        gv[j] = 0;
                        if (idx < krows * nbc) {
                            const int k = idx / nbc, b = cb * CB + idx % nbc;
                            if (k < R) {
                                if (s >= 1) gp[j] = a.gz + ((long)((s - 1) & 1) * R + k) * B + b;
                            } else if (k < 2 * R) {
                                if (s <= 1) fv[j] = x0_of(k - R, b);
                                else gp[j] = a.gx + ((long)((s - 1) & 1) * R + (k - R)) * B + b;
                            } else {
                                const int jt = (k - 2 * R) / R, c = (k - 2 * R) % R;
                                int slot = (p - (K - 1 - jt) * d) % Dq;
                                if (slot < 0) slot += Dq;
                                fv[j] = pq[(qoff_s + (long)slot * R + c) * B + b];
                            }
                            if (gp[j]) gv[j] = wn_granule_load(gp[j]);
                        }
                    }
                    poll_all(std::integral_constant<int, GB>(), gp, gv, tag0 + (unsigned)(s - 1));
                    WN_UNROLL
                    for (int j = 0; j < GB; ++j) {
                        const int idx = base + j * WN_DLP_T;
                        if (idx < krows * nbc) {
                            const int k = idx / nbc, uc = idx % nbc, b = cb * CB + uc;
                            float v = fv[j];
                            if (gp[j]) v = wn_bits_f32((unsigned)gv[j]);
                            if (k >= R && k < 2 * R && s >= 1) {   // x_{s-1} of this step goes into the unit's own ring of layer s-1
                                const int dp = 1 << ((s - 1) % a.depth), Dp = (K - 1) * dp;
                                pq[(dlp_queue_off(s - 1, a.depth, K, R) + (long)(p % Dp) * R + (k - R)) * B + b] = v;
                            }
                            if (hasP || k < R) s_in[uc * KPAD + k] = v;
                        }
                    }
                }
                DLP_STAMP(s, 1);
                // The stage's outputs: (3 CG + SU) x CB row sums of SL partial sums each.  Eight adjacent lanes share a row sum
                // (lane g adds the partial sums of slices [g SL/8, (g+1) SL/8), then three butterfly steps: a fixed order),
                // and lane g = 0 of the group runs the output's epilogue -- so what that lane reads from memory (aux
                // projection, constants, biases) is requested HERE, ahead of the next stage's weight stream.
                float e0[NPASS], e1[NPASS];
                WN_UNROLL
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int w8 = tid + WN_DLP_T * ps, rs = w8 >> 3;
                    e0[ps] = 0.0f;
                    e1[ps] = 0.0f;
                    if ((w8 & 7) == 0) {
                        if (rs < 2 * CG * CB) {
                            const int zi = rs >> 1, c = zi / CB, b = cb * CB + zi % CB;
                            if (hasP && (rs & 1) == 0 && b < B) {
                                const int t = p > a.n_pad ? p - a.n_pad : 0;   // replicated first column inside the left padding
                                int f = t / a.Ue;
                                const float wj = a.upw[t - f * a.Ue];
                                if (f > a.F - 1) f = a.F - 1;
                                const float* Gs = a.G + ((long)b * a.F + f) * a.nG + (long)s * 2 * R;
                                e0[ps] = wj * Gs[c0 + c] + a.cfold[(long)s * 2 * R + c0 + c];
                                e1[ps] = wj * Gs[R + c0 + c] + a.cfold[(long)s * 2 * R + R + c0 + c];
                            }
                        } else if (rs < 3 * CG * CB) {
                            const int e = rs - 2 * CG * CB, c = e / CB, b = cb * CB + e % CB;
                            if (b < B && s < L)
                                e0[ps] = s == 0 ? x0_of(c0 + c, b) : a.params[a.off_res_b0 + (long)(s - 1) * a.res_b_lstride + c0 + c];
                        }
                    }
                }
                __syncthreads();
                DLP_STAMP(s, 2);
                // (3) the two row sets: [CG sigmoid | CG tanh] rows over all of K, [CG x | skip rows] over the z part
                float accP[CB], accX[CB];
                partial_dots(wP, std::integral_constant<int, NSP>(), hasP, nbc, accP);
                partial_dots(wX, std::integral_constant<int, NSX>(), hasX, nbc, accX);
                if (cb == ncb - 1) {   // the registers are free: the next weights (next stage, or the post net's first set)
                    if (s < L) issue_stage_weights(s + 1);
                    else load_weights(wX, std::integral_constant<int, NSX>(), pimg);
                }
                DLP_STAMP(s, 3);
                put_partials(0, accP);
                put_partials(1, accX);
                __syncthreads();
                DLP_STAMP(s, 4);
                // (4) row sums and epilogues
                WN_UNROLL
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int w8 = tid + WN_DLP_T * ps, rs = w8 >> 3, g = w8 & 7;
                    int set = 1, row = 0, uc = 0, kind = 3;   // kind 0: gate row, 1: x row, 2: skip row, 3: none
                    if (rs < 2 * CG * CB) {
                        const int zi = rs >> 1;
                        kind = 0; set = 0; row = (rs & 1) * CG + zi / CB; uc = zi % CB;
                    } else if (rs < 3 * CG * CB) {
                        const int e = rs - 2 * CG * CB;
                        kind = 1; row = e / CB; uc = e % CB;
                    } else if (rs < (3 * CG + SU) * CB) {
                        const int e = rs - 3 * CG * CB;
                        kind = 2; row = CG + e / CB; uc = e % CB;
                    }
                    float sum = 0.0f;
                    if (kind != 3) {
                        const float* q = s_red + ((set * SL + g * (SL / 8)) * RS + row) * CB + uc;
                        WN_UNROLL
                        for (int i = 0; i < SL / 8; ++i) sum += q[i * RS * CB];
                    }
                    sum += __shfl_xor(sum, 1, 64);
                    sum += __shfl_xor(sum, 2, 64);
                    sum += __shfl_xor(sum, 4, 64);
                    const float other = __shfl_down(sum, 8, 64);   // a gate output's sigmoid lane takes the tanh row's sum
                    const int b = cb * CB + uc;
                    if (g == 0 && b < B) {
                        if (kind == 0 && (rs & 1) == 0 && hasP) {   // gate (wavenet.py:542-544)
                            const int c = row;
                            wn_granule_store(a.gz + ((long)(s & 1) * R + c0 + c) * B + b, wn_sigmoid(sum + e0[ps]) * wn_tanh(other + e1[ps]),
                                             tag0 + (unsigned)s);
                        } else if (kind == 1 && s < L) {
                            const int c = row;
                            float xs;
                            if (s == 0) {   // x_0 of the unit's own channels
                                xs = e0[ps];
                            } else {        // x_s = res_1x1(z_{s-1}) + x_{s-1}   (wavenet.py:546-548)
                                xs = sum + e0[ps] + s_xown[c * BM + b];
                                wn_granule_store(a.gx + ((long)(s & 1) * R + c0 + c) * B + b, xs, tag0 + (unsigned)s);
                            }
                            s_xown[c * BM + b] = xs;
                            if (K >= 2)   // the shared rings stay current for the next launch (and the launch path)
                                a.queues[(dlp_queue_off(s, a.depth, K, R) + (long)(p % Dq) * R + c0 + c) * B + b] = xs;
                        } else if (kind == 2 && hasX) {   // the unit's rows of the skip sum (wavenet.py:545, 365)
                            s_sk[(row - CG) * BM + b] += sum;
                        }
                    }
                }
                DLP_STAMP(s, 5);
                if (ncb > 1) __syncthreads();   // (several blocks per stage: the next block's gather re-uses s_in at once)
                DLP_STAMP(s, 6);
            }
        }
        DLP_STAMP(L + 1, 0);
        __syncthreads();   // the skip accumulators of the last stage's epilogue are complete
        // ---- post net (wavenet.py:518-523): relu(skip sum) -> conv_post_1 + relu -> conv_post_2, three more hops ----
        for (int i = tid; i < SU * B; i += WN_DLP_T) {
            const int r = i / B, b = i % B, row = u * SU + r;
            if (row < S) wn_granule_store(a.gs + (long)row * B + b, fmaxf(s_sk[r * BM + b] + a.bskip[row], 0.0f), tag0 + (unsigned)(L + 1));
            s_sk[r * BM + b] = 0.0f;
        }
        for (int stage = 0; stage < 2; ++stage) {
            const u64* src = stage == 0 ? a.gs : a.go;
            for (int cb = 0; cb < ncb; ++cb) {
                const int nbc = (B - cb * CB) < CB ? (B - cb * CB) : CB;
                gather_vec(src, S, nbc, cb, tag0 + (unsigned)(L + 1 + stage));
                constexpr int PPASS = (RS * CB * 8 + WN_DLP_T - 1) / WN_DLP_T;
                float pb[PPASS];   // the row's bias, ahead of the next weight stream
                WN_UNROLL
                for (int ps = 0; ps < PPASS; ++ps) {
                    const int w8 = tid + WN_DLP_T * ps, r = (w8 >> 3) / CB, row = u * (stage == 0 ? SU : QU) + r;
                    pb[ps] = 0.0f;
                    if ((w8 & 7) == 0 && (stage == 0 ? (r < SU && row < S) : (r < QU && row < Qo)))
                        pb[ps] = a.params[(stage == 0 ? a.off_post1_b : a.off_post2_b) + row];
                }
                __syncthreads();
                float acc[CB];
                partial_dots(wX, std::integral_constant<int, NSX>(), true, nbc, acc);
                if (cb == ncb - 1) {
                    if (stage == 0) load_weights(wX, std::integral_constant<int, NSX>(), pimg + 512L * NSX);
                    else if (p + 1 < a.p1) issue_stage_weights(0);
                }
                put_partials(0, acc);
                __syncthreads();
                WN_UNROLL
                for (int ps = 0; ps < PPASS; ++ps) {   // 8 lanes per row sum, as in the layer stages
                    const int w8 = tid + WN_DLP_T * ps, rs = w8 >> 3, g = w8 & 7;
                    const int r = rs / CB, uc = rs % CB, b = cb * CB + uc;
                    float sum = 0.0f;
                    if (r < RS) {
                        const float* q = s_red + ((g * (SL / 8)) * RS + r) * CB + uc;
                        WN_UNROLL
                        for (int i = 0; i < SL / 8; ++i) sum += q[i * RS * CB];
                    }
                    sum += __shfl_xor(sum, 1, 64);
                    sum += __shfl_xor(sum, 2, 64);
                    sum += __shfl_xor(sum, 4, 64);
                    if (g == 0 && r < RS && b < B) {
                        if (stage == 0) {
                            const int row = u * SU + r;
                            if (r < SU && row < S) wn_granule_store(a.go + (long)row * B + b, fmaxf(sum + pb[ps], 0.0f), tag0 + (unsigned)(L + 2));
                        } else {
                            const int row = u * QU + r;
                            if (r < QU && row < Qo) wn_granule_store(a.gl + (long)row * B + b, sum + pb[ps], tag0 + (unsigned)(L + 3));
                        }
                    }
                }
                __syncthreads();
            }
        }
        // ---- token choice, by every unit for itself (wavenet.py:371-381): first-max argmax or inverse CDF on the caller's draw ----
        for (int cb = 0; cb < ncb; ++cb) {
            const int nbc = (B - cb * CB) < CB ? (B - cb * CB) : CB;
            gather_vec(a.gl, Qo, nbc, cb, tag0 + (unsigned)(L + 3));
            __syncthreads();
            if (tid < CB && cb * CB + tid < B) {
                const int b = cb * CB + tid;
                float best = -3.0e38f;
                int bi = 0;
                for (int q = 0; q < Qo; ++q) {
                    const float v = s_in[tid * KPAD + q];
                    if (u == 0 && a.logits_out) a.logits_out[((long)b * a.Ttot + p) * Qo + q] = v;
                    if (v > best) { best = v; bi = q; }
                }
                int chosen = bi;
                if (a.mode == 1 && a.uniforms != nullptr) {
                    float total = 0.0f;
                    for (int q = 0; q < Qo; ++q) total += expf(s_in[tid * KPAD + q] - best);
                    const float target = a.uniforms[(long)b * a.Ttot + p + 1] * total;
                    float run = 0.0f;
                    int cand = -1;
                    for (int q = 0; q < Qo; ++q) {
                        run += expf(s_in[tid * KPAD + q] - best);
                        if (cand < 0 && run >= target) cand = q;
                    }
                    if (cand >= 0) chosen = cand;
                }
                const bool gen = p + 1 >= a.t_forced[b] && p + 1 < a.t_end[b];
                long long nxt = chosen;
                if (!gen && p + 1 < a.Ttot) {   // teacher forced / finished utterance: the token that is in the buffer
                    nxt = a.samples[(long)b * a.Ttot + p + 1] % a.Q;
                    if (nxt < 0) nxt += a.Q;
                }
                if (gen && u == 0) a.samples[(long)b * a.Ttot + p + 1] = chosen;
                for (int j = 0; j + 1 < K; ++j) s_tok[j * BM + b] = s_tok[(j + 1) * BM + b];
                s_tok[(K - 1) * BM + b] = (int)nxt;
            }
            __syncthreads();
        }
        DLP_STAMP(L + 2, 0);
        if (s_flag[0]) break;
    }
    if (tid == 0 && s_flag[0]) wn_store_coherent_int(a.err, 1);
}

// LDS attribute (once) and the number of workgroups of this class the device keeps resident (cached; 0: the query failed)
template <int RS, int NSP, int NSX>
static int capacity_cls(long lds_bytes) {
    static int cap[WN_COOP_MAXDEV];
    static bool cap_init = false;
#ifndef WN_EMU
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_dlp<RS, NSP, NSX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return 0;
        attr_set = true;
    }
#endif
    if (wn_coop_capacity_override() >= 0) return wn_coop_capacity_override();
    return wn_coop_capacity_cached(cap, cap_init, k_dlp<RS, NSP, NSX>, WN_DLP_T, (size_t)lds_bytes);
}

template <int RS, int NSP, int NSX>
static int launch_cls(const WnDlpArgs& a, wn_stream_t st) {
    if (a.plan.NU > capacity_cls<RS, NSP, NSX>(a.plan.lds_bytes)) return 4;   // not all workgroups would be resident: no launch
    WN_LAUNCH_COOP((k_dlp<RS, NSP, NSX>), dim3((unsigned)a.plan.NU), dim3(WN_DLP_T), (size_t)a.plan.lds_bytes, st, a);
    return 0;
}

int wn_dlp_capacity(const WnDlpPlan* plan) {
    if (!plan->ok || plan->wide) return 0;
    switch (plan->cls) {
        case 0: return capacity_cls<8, 24, 8>(plan->lds_bytes);
        case 1: return capacity_cls<8, 32, 8>(plan->lds_bytes);
        case 2: return capacity_cls<16, 48, 16>(plan->lds_bytes);
        case 3: return capacity_cls<16, 64, 16>(plan->lds_bytes);
        case 4: return capacity_cls<32, 96, 32>(plan->lds_bytes);
        case 5: return capacity_cls<32, 128, 32>(plan->lds_bytes);
    }
    return 0;
}

int wn_dlp_launch(const WnDlpArgs* ap, wn_stream_t st) {
    const WnDlpArgs& a = *ap;
    if (!a.plan.ok || a.B < 1 || a.B > WN_DLP_BMAX || a.p1 < a.p0) return 1;
    if (a.mode != 0 && a.mode != 1) return 2;
    WN_PROF("dlp_steps", 0.0, 0.0, st);
    switch (a.plan.cls) {
        case 0: return launch_cls<8, 24, 8>(a, st);
        case 1: return launch_cls<8, 32, 8>(a, st);
        case 2: return launch_cls<16, 48, 16>(a, st);
        case 3: return launch_cls<16, 64, 16>(a, st);
        case 4: return launch_cls<32, 96, 32>(a, st);
        case 5: return launch_cls<32, 128, 32>(a, st);
    }
    return 1;
}
