// wn_dlp.h -- persistent any-size decode (wn_dlp.hip): the queue algorithm of the reference (wavenet.py:355-385, 538-549) for
// models too large for the one-workgroup kernel of wn_decode.hip (the recipes' n_resch = 512), as ONE launch per chunk of
// steps instead of ~66 dependent launches per step.
//
// R / CG workgroups ("units") each own CG = 4 (8, 16 for very wide models) residual channels.  A step is a chain of L + 3 dependent stages; a stage's output
// vectors travel from workgroup to workgroup as 8-byte granules {value, tag} (one agent-scope store per lane, the consumer
// polls the granules themselves: 0.7 - 0.8 us per hop, profiles/r04/handoff_microbench.txt) -- no grid barrier, no fences.
#pragma once
#include "wn_device.h"

#define WN_DLP_T 512     // threads per workgroup (8 waves: 2 per SIMD, 256 VGPRs each for the stage's weights)
#define WN_DLP_NW 8
#define WN_DLP_CB 4      // utterance columns per block (one 16-byte LDS read per k)
#define WN_DLP_BMAX 4    // utterances per launch of the VALU kernel (k_dlp): one column block
#define WN_DLPM_CB 16    // utterance columns per block of the matrix-core kernel (k_dlpm, wn_dlpm.hip): one 16x16x4 tile
#define WN_DLPM_BMAX 48  // utterances per launch of the matrix-core kernel with the granule hand-off (wn_dlpm.hip: private queue copies per unit)
#define WN_DLPF_BMAX 64  // utterances per launch of the flag hand-off kernel (wn_dlpf.hip): up to 4 column blocks, each with its own units
#define WN_DLPM_MAXWG 256 // workgroups of one launch (all resident at once, one per CU: what the device really keeps is asked before the launch)
#ifndef WN_DLPF_ENABLE
#define WN_DLPF_ENABLE 1   // 1: the flag hand-off kernel (wn_dlpf.hip) where it covers the plan; 0: granules everywhere (A/B builds)
#endif
#ifndef WN_DLPM_BMIN
#define WN_DLPM_BMIN 5   // smallest batch that takes the matrix-core kernel with the granule hand-off (365 us per step at any batch up to 32; the VALU kernel:
                         // 366 at 4); where wn_dlpf.hip covers the plan the matrix-core kernel is used from 2 utterances on (wn_api.hip)
#endif

typedef struct WnDlpPlan {
    int ok;
    int wide;                // 1: plan of the matrix-core kernel (RS = 16, batches of 9 .. 32 utterances)
    int cls;                 // compiled class: NSP / NSX k-steps per wave of the two tiles
    int RS;                  // rows per set = 2 x channels per unit (8, 16 or 32)
    int NSP, NSX;
    int NU, SU, QU, KP;      // units, skip rows per unit, output rows per unit, K of the gate tile = (K + 1) * R
    long stage_floats;       // packed floats of one (stage, unit): 512 * (NSP + NSX)
    long post_floats;        // packed floats of one unit's two post tiles: 2 * 512 * NSX
    long lds_bytes;
} WnDlpPlan;
// ok = 0: the model is not covered (R % 32, S / Qo rows per unit, kernel_size, class sizes)
// wide = 0: the VALU kernel's plan (as many units as fit: B <= WN_DLP_BMAX); wide = 1: the matrix-core kernel's (16 rows per set)
void wn_dlp_make_plan(int Q, int Qo, int R, int S, int L, int K, int wide, WnDlpPlan* plan);

typedef struct WnDlpArgs {
    int Q, Qo, R, S, L, K, depth, nG;
    WnDlpPlan plan;
    int B;
    const float* wpk;        // [stage 0..L][unit][tile P | tile X]
    const float* wpost;      // [unit][post1 tile | post2 tile]
    const float* cfold;      // [L][2R] constant part of the gate pre-activation (incl. the folded res bias)
    const float* bskip;      // [S] sum of the skip biases
    const float* params;
    long off_causal_w, off_causal_b, off_res_b0, res_b_lstride, off_post1_b, off_post2_b;
    const float* upw;
    int Ue, F, n_pad;
    const float* G;          // (B, F, nG)
    int64_t* samples;        // (B, Ttot)
    long Ttot;
    const int* t_forced;
    const int* t_end;
    const float* uniforms;   // mode 1: (B, Ttot)
    float* logits_out;       // nullable (B, Ttot, Qo)
    int mode;                // 0 argmax, 1 sampling
    int p0, p1;
    unsigned long long* gz;  // granules [2][R][B]
    unsigned long long* gx;  // [2][R][B]
    unsigned long long* gs;  // [S][B]   relu(skip sum)
    unsigned long long* go;  // [S][B]   relu(post1)
    unsigned long long* gl;  // [Qo][B]  logits
    float* pq;               // private queue copies [unit][qfloats][B]
    long pq_unit_stride;
    float* queues;           // shared rings [qfloats][B] (prefill / the launch path's layout)
    long qfloats;
    int* err;                // set to 1 when a poll timed out
    // wn_dlpf.hip (hand-off of plain vectors + one flag per producer; plan.wide, NSP = 48): the granule regions hold plain
    // floats -- gz: [2][2R][Bp] z | x of the previous stage, gs / go / gl: [rows][Bp] --, no private queues (`queues` is read
    // by everyone, written by the owner of a channel)
    int handoff;             // 0: granules (wn_dlp.hip, wn_dlpm.hip), 1: flags (wn_dlpf.hip)
    int Bp;                  // row stride of the plain vectors: 16 * blocks
    unsigned long long* flags;   // [blocks][NU]: the tag of the latest stage the unit has published
} WnDlpArgs;

struct WnDlpPackArgs {
    int R, S, Qo, L, K;
    WnDlpPlan plan;
    int stage;               // 0 .. L
    const float* params;
    long lb_s, lb_prev;      // layer blocks of layers `stage` and `stage - 1` in the flat parameter buffer
    long o_dsig_w, o_dtanh_w, o_res_w;
    long skip_prev;          // skip_1x1.(stage-1).weight
    const float* fold;       // [2R][R] = Wd_new(stage) . Wres(stage-1), stage >= 1
    float* dst;              // wpk + stage * NU * stage_floats
};
int wn_dlp_pack_stage(const WnDlpPackArgs* a, wn_stream_t st);
int wn_dlp_pack_post(const float* params, long post1_w, long post2_w, int S, int Qo, const WnDlpPlan* plan, float* dst, wn_stream_t st);
// cfold[s][o'] = cvec[s][o'] + sum_j Wd_new(s)[o'][j] * b_res(s-1)[j]   (second term for s >= 1)
int wn_dlp_cfold(const float* params, const float* cvec, const float* wd_f, long lb0, long lstep, long o_res_b, int L, int R, int K,
                 float* cfold, wn_stream_t st);
int wn_dlp_launch(const WnDlpArgs* a, wn_stream_t st);
int wn_dlpm_launch(const WnDlpArgs* a, wn_stream_t st);   // plan.wide == 1, B <= WN_DLPM_BMAX
int wn_dlpf_launch(const WnDlpArgs* a, wn_stream_t st);   // the same with a.handoff == 1
int wn_dlpf_covers(const WnDlpPlan* plan);                 // 1: wn_dlpf.hip has a kernel for this plan
// Workgroups of the plan's kernel the current device keeps resident at once (occupancy x CUs; cached per kernel class; 0: no
// kernel for the plan / the query failed).  A launch whose grid exceeds it is refused (rc 4) instead of timing out.
int wn_dlp_capacity(const WnDlpPlan* plan);
int wn_dlpm_capacity(const WnDlpPlan* plan);
int wn_dlpf_capacity(const WnDlpPlan* plan);
