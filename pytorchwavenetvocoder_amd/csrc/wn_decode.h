// wn_decode.h -- sample-by-sample autoregressive decode (wn_decode.hip).
//
// The "Fast WaveNet" queue algorithm of the reference (wavenet.py:309-511, 538-549) as ONE
// persistent workgroup per utterance: every step streams the packed fp32 weights of the whole
// network through registers (they do not fit a CU: 4.9 MB at R=64/S=256/Q=256), keeps the
// activations of the step in LDS and the dilation queues (sum (K-1)*d*R floats = 786 KB per
// utterance at the BASELINE size) in an L2-resident ring buffer per layer.
#pragma once
#include "wn_device.h"

#define WN_DT 512  // threads per decode workgroup (8 waves, 2 per SIMD)

// How the outputs of one matrix-vector stage are spread over the 512 threads: `parts` threads
// (adjacent lanes) share one output row and split the input vector in float4 units,
// unit r of thread (o, part) covers inputs 4*(part + parts*r) .. +3.
typedef struct WnDecodePlan {
    int ok;             // 0: this configuration is not covered by the compiled unit classes
    int cls;            // compiled class index
    int UD, UR, US, UP1, UP2;  // float4 units per thread and stage (class constants)
    int R4, S4;         // input vector lengths in float4
    int lg_pd, lg_pr, lg_ps, lg_p1, lg_p2;  // log2(parts) of the dilated / res / skip / post1 / post2 stage
    long stream_f4;     // float4 units of the packed stream
    long off_cvec, off_bskip, off_wauxf, off_one;  // float offsets of the side tables behind the stream
    long total_floats;  // size of the packed decode weights
    long queue_floats;  // per utterance
    size_t lds_bytes;
} WnDecodePlan;

typedef struct WnDecodeArgs {
    int Q, A, R, S, L, K, depth;
    int Qo;               // rows of conv_post_2: Q (softmax head) or 3*n_mix (mixture-of-logistics head)
    WnDecodePlan plan;
    const float* wpack;   // packed decode weights (stream + side tables)
    const float* params;  // flat parameter buffer (biases, front conv, upsampling weights)
    long off_causal_w, off_causal_b, off_res_b0, res_b_lstride, off_post1_b, off_post2_b;
    const float* upw;     // [Ue] upsampling taps (or a vector of ones)
    int Ue;
    const float* G;       // (B, F, L*2R) aux projections at the aux rate
    long g_bstride;
    int F, n_pad;
    int64_t* samples;     // (B, Ttot) tokens; position p+1 is written by step p when p+1 >= t_forced[b]
    long s_bstride;
    const int* t_forced;  // (B) first generated position
    const int* t_end;     // (B) number of valid positions of utterance b
    int p0, p1;           // steps [p0, p1)
    float* queues;        // (B, queue_floats), zero before step 0
    long q_bstride;
    const float* uniforms;  // nullable (B, Ttot): uniform draw used for position p+1 at [p+1]; mode 2: (B, Ttot, nm+1)
    long u_bstride;
    float* logits_out;    // nullable (B, Ttot, Qo): row p = network output computed by step p
    long lo_bstride;
    int mode;             // 0 argmax, 1 sampling, 2 mixture of logistics (Qo = 3*nm, nm <= 64)
    float* wave_out;      // mode 2, nullable (B, Ttot): the drawn waveform value of position p+1
    float log_scale_min;  // mode 2: clamp of the mixture log-scales (the value wn_mol_loss was trained with)
    long w_bstride;
#ifdef WN_TIMING
    long long* dbg;
#endif
} WnDecodeArgs;

// Fills the plan for the configuration; plan->ok == 0 if no compiled class covers it.
// Q here = rows of conv_post_2 (out_channels)
void wn_decode_make_plan(int Q, int A, int R, int S, int L, int K, int depth, WnDecodePlan* plan);

struct WnDecodePackArgs {
    int Q, R, S, L, K;
    WnDecodePlan plan;
    const float* params;
    long lb0, lstep;  // layer l block = lb0 + l*lstep
    long o_dsig_w, o_dtanh_w, o_res_w;
    long skip0, ls_skip, post1_w, post2_w;
    float* stream;
};
int wn_decode_pack_stream(const WnDecodePackArgs* a, wn_stream_t st);
int wn_decode_launch(const WnDecodeArgs* a, int B, wn_stream_t st);
