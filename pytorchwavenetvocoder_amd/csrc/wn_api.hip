// wn_api.hip -- C-ABI entry points of libwavenet_hip.so (see include/wavenet_hip.h).
//
// Host-side orchestration only: parameter layout, workspace carving and the launch sequence of
// the forward / loss / backward / Adam steps of the WaveNet training path.  All arithmetic is in
// the HIP kernels of wn_gemm.hip, wn_elem.hip and wn_fused.hip.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/wavenet_hip.h"
#include "wn_decode.h"
#include "wn_dlp.h"
#include "wn_elem.h"
#include "wn_fused.h"
#include "wn_gemm.h"
#include "wn_gemm6.h"
#include "wn_prof.h"

// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// Every entry point starts from a clean slate: the error text is reset and an error some EARLIER runtime call of
// the process left behind (hipGetLastError is sticky per thread: e.g. a device probe before the framework
// initialised the runtime) is discarded, so rt_check only reports launches of this call.
#ifdef WN_EMU
static void api_enter() { g_err[0] = 0; }
#else
static void api_enter() {
    g_err[0] = 0;
    (void)hipGetLastError();
}
#endif

#ifdef WN_EMU
static int rt_check(const char*) { return 0; }
static void rt_event_record(void*, wn_stream_t) { wn_prof_mark("bucket_event"); }
#else
static int rt_check(const char* where) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(100, "HIP error after %s: %s", where, hipGetErrorString(e));
    return 0;
}
static void rt_event_record(void* ev, wn_stream_t st) {
    wn_prof_mark("bucket_event");
    (void)hipEventRecord((hipEvent_t)ev, st);
}
#endif

// ------------------------------------------------------------------------------------------
// Internal side stream of wn_backward.  The data chain of the backward pass (gate', dX: 2 dependent launches per
// layer, HBM-bound) and the weight-gradient contractions of the layers already walked (matrix-core-bound) do not
// depend on each other, so the latter are enqueued on a second stream: they fill the drain/ramp gap between two
// dependent chain launches and the partial last round of their tiles.  Fork/join with events; the caller sees
// the usual stream semantics (everything is complete in `stream` order when the call's work retires).
// One non-blocking stream + event pool per device, created on first use and kept (the only state of the library);
// a mutex makes the record/wait pairs of concurrent callers atomic.  Opt-in (WN_FLAG_BWD_OVERLAP / WN_FLAG_FWD_OVERLAP):
// see DESIGN.md 5.1 for the measurement; per-launch profiling (wn_prof_enable) keeps everything on the caller's stream.
// ------------------------------------------------------------------------------------------
#define WN_DW_FLUSH_DEFAULT 5
#ifdef WN_EMU
struct SideRt {
    wn_stream_t st;
};
struct SideLock {   // emulator: one in-order "stream", but the overlap launch sequences (chunked skip-sum) still run
    SideRt* rt;
    SideLock(bool want, wn_stream_t) : rt(nullptr) {
        static SideRt one = {nullptr};
        if (want) rt = &one;
    }
};
static int side_link(SideRt*, wn_stream_t, wn_stream_t) { return 0; }
#else
#include <mutex>
#include <vector>
struct SideRt {
    hipStream_t st = nullptr;
    std::vector<hipEvent_t> ev;
    size_t next = 0;
    std::mutex mu;
};
static SideRt* side_get(wn_stream_t caller) {
    static std::mutex g_mu;
    static SideRt* g_rt[64] = {nullptr};
    int dev = 0, sdev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    // the launches follow the current device (as everywhere in this library); a caller stream of another device
    // gets the serial mode instead of a cross-device fork
    if (caller && (hipStreamGetDevice(caller, &sdev) != hipSuccess || sdev != dev)) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_rt[dev]) {
        SideRt* r = new SideRt();
        // lowest priority: the side stream carries filler work, the caller's stream carries the dependent chain
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
        const hipError_t ce = hipStreamCreateWithPriority(&r->st, hipStreamNonBlocking, least);
        if (ce != hipSuccess) {
            (void)hipGetLastError();
            delete r;
            return nullptr;
        }
        g_rt[dev] = r;
    }
    return g_rt[dev];
}
struct SideLock {   // holds the device's side runtime for one call (nullptr = run serially on the caller's stream)
    SideRt* rt;
    SideLock(bool want, wn_stream_t caller) : rt(want ? side_get(caller) : nullptr) {
        if (rt) {
            rt->mu.lock();
            rt->next = 0;
        }
    }
    ~SideLock() {
        if (rt) rt->mu.unlock();
    }
    SideLock(const SideLock&) = delete;
    SideLock& operator=(const SideLock&) = delete;
};
// work enqueued on `to` after this call waits for everything enqueued on `from` so far
static int side_link(SideRt* rt, wn_stream_t from, wn_stream_t to) {
    if (!rt || from == to) return 0;
    if (rt->next == rt->ev.size()) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(100, "hipEventCreate failed");
        rt->ev.push_back(e);
    }
    hipEvent_t e = rt->ev[rt->next++];
    if (hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess)
        return fail(100, "stream fork/join failed: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}
#endif

#define WN_TRY(expr)                                                                     \
    do {                                                                                 \
        int _rc = (expr);                                                                \
        if (_rc != 0) return (g_err[0] ? _rc : fail(_rc, "%s failed (rc=%d)", #expr, _rc)); \
    } while (0)

extern "C" int wn_abi_version(void) { return WN_ABI_VERSION; }
extern "C" const char* wn_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------
// configuration / parameter layout
// ------------------------------------------------------------------------------------------
struct Dims {
    int Q, A, R, S, L, K, U;
    int Qo;  // output channels of the post net (Q for the softmax head)
};

static int check_cfg(const WnConfig* c, Dims* d) {
    if (!c) return fail(1, "cfg is NULL");
    if (c->n_quantize < 2 || c->n_aux < 1 || c->n_resch < 1 || c->n_skipch < 1 || c->dilation_depth < 1 ||
        c->dilation_depth > 24 || c->dilation_repeat < 1 || c->kernel_size < 1 || c->kernel_size > 8 ||
        c->upsampling_factor < 0 || c->out_channels < 0)
        return fail(1, "invalid WnConfig");
    d->Q = c->n_quantize;
    d->Qo = c->out_channels > 0 ? c->out_channels : c->n_quantize;
    d->A = c->n_aux;
    d->R = c->n_resch;
    d->S = c->n_skipch;
    d->L = c->dilation_depth * c->dilation_repeat;
    d->K = c->kernel_size;
    d->U = c->upsampling_factor;
    return 0;
}

static inline int dilation_of(const WnConfig* c, int l) { return 1 << (l % c->dilation_depth); }

struct Lay {
    long post2_w, post2_b, post1_w, post1_b;
    long skip0, ls_skip;          // skip_1x1.l : skip0 + l*ls_skip (+S*R for bias)
    long layers0, LB;             // layer l block at layers0 + (L-1-l)*LB
    long o_dsig_w, o_dsig_b, o_dtanh_w, o_dtanh_b, o_asig_w, o_asig_b, o_atanh_w, o_atanh_b, o_res_w, o_res_b;
    long causal_w, causal_b, up_w, up_b;
    long total;
};

static Lay make_lay(const Dims& d) {
    Lay y;
    long o = 0;
    y.post2_w = o; o += (long)d.Qo * d.S;
    y.post2_b = o; o += d.Qo;
    y.post1_w = o; o += (long)d.S * d.S;
    y.post1_b = o; o += d.S;
    y.skip0 = o;
    y.ls_skip = (long)d.S * d.R + d.S;
    o += y.ls_skip * d.L;
    y.layers0 = o;
    long q = 0;
    y.o_dsig_w = q; q += (long)d.R * d.R * d.K;
    y.o_dsig_b = q; q += d.R;
    y.o_dtanh_w = q; q += (long)d.R * d.R * d.K;
    y.o_dtanh_b = q; q += d.R;
    y.o_asig_w = q; q += (long)d.R * d.A;
    y.o_asig_b = q; q += d.R;
    y.o_atanh_w = q; q += (long)d.R * d.A;
    y.o_atanh_b = q; q += d.R;
    y.o_res_w = q; q += (long)d.R * d.R;
    y.o_res_b = q; q += d.R;
    y.LB = q;
    o += y.LB * d.L;
    y.causal_w = o; o += (long)d.R * d.Q * d.K;
    y.causal_b = o; o += d.R;
    if (d.U > 0) {
        y.up_w = o; o += d.U;
        y.up_b = o; o += 1;
    } else {
        y.up_w = y.up_b = -1;
    }
    y.total = o;
    return y;
}
static inline long layer_base(const Lay& y, const Dims& d, int l) { return y.layers0 + (long)(d.L - 1 - l) * y.LB; }

extern "C" int wn_num_layers(const WnConfig* cfg) {
    Dims d;
    if (check_cfg(cfg, &d)) return -1;
    return d.L;
}

extern "C" int wn_receptive_field(const WnConfig* cfg) {
    Dims d;
    if (check_cfg(cfg, &d)) return -1;
    long sum = 0;
    for (int l = 0; l < d.L; ++l) sum += dilation_of(cfg, l);
    return (int)((d.K - 1) * sum + 1);
}

extern "C" int64_t wn_param_count(const WnConfig* cfg) {
    Dims d;
    if (check_cfg(cfg, &d)) return -1;
    return make_lay(d).total;
}

extern "C" int wn_param_offset(const WnConfig* cfg, int kind, int layer, int64_t* offset, int64_t* numel) {
    Dims d;
    WN_TRY(check_cfg(cfg, &d));
    const Lay y = make_lay(d);
    long off = -1, n = 0;
    const bool per_layer = (kind >= WN_P_DSIG_W && kind <= WN_P_RES_B);
    if (per_layer && (layer < 0 || layer >= d.L)) return fail(2, "layer %d out of range", layer);
    const long lb = per_layer ? layer_base(y, d, layer) : 0;
    switch (kind) {
        case WN_P_CAUSAL_W: off = y.causal_w; n = (long)d.R * d.Q * d.K; break;
        case WN_P_CAUSAL_B: off = y.causal_b; n = d.R; break;
        case WN_P_UP_W: off = y.up_w; n = d.U; break;
        case WN_P_UP_B: off = y.up_b; n = 1; break;
        case WN_P_DSIG_W: off = lb + y.o_dsig_w; n = (long)d.R * d.R * d.K; break;
        case WN_P_DSIG_B: off = lb + y.o_dsig_b; n = d.R; break;
        case WN_P_DTANH_W: off = lb + y.o_dtanh_w; n = (long)d.R * d.R * d.K; break;
        case WN_P_DTANH_B: off = lb + y.o_dtanh_b; n = d.R; break;
        case WN_P_ASIG_W: off = lb + y.o_asig_w; n = (long)d.R * d.A; break;
        case WN_P_ASIG_B: off = lb + y.o_asig_b; n = d.R; break;
        case WN_P_ATANH_W: off = lb + y.o_atanh_w; n = (long)d.R * d.A; break;
        case WN_P_ATANH_B: off = lb + y.o_atanh_b; n = d.R; break;
        case WN_P_SKIP_W: off = y.skip0 + layer * y.ls_skip; n = (long)d.S * d.R; break;
        case WN_P_SKIP_B: off = y.skip0 + layer * y.ls_skip + (long)d.S * d.R; n = d.S; break;
        case WN_P_RES_W: off = lb + y.o_res_w; n = (long)d.R * d.R; break;
        case WN_P_RES_B: off = lb + y.o_res_b; n = d.R; break;
        case WN_P_POST1_W: off = y.post1_w; n = (long)d.S * d.S; break;
        case WN_P_POST1_B: off = y.post1_b; n = d.S; break;
        case WN_P_POST2_W: off = y.post2_w; n = (long)d.Qo * d.S; break;
        case WN_P_POST2_B: off = y.post2_b; n = d.Qo; break;
        default: return fail(2, "unknown tensor kind %d", kind);
    }
    if ((kind == WN_P_SKIP_W || kind == WN_P_SKIP_B) && (layer < 0 || layer >= d.L))
        return fail(2, "layer %d out of range", layer);
    if (off < 0) return fail(3, "tensor kind %d does not exist for this config", kind);
    if (offset) *offset = off;
    if (numel) *numel = n;
    return 0;
}

extern "C" int wn_num_buckets(const WnConfig* cfg, int lpb) {
    Dims d;
    if (check_cfg(cfg, &d)) return -1;
    if (lpb < 1) lpb = d.L;
    return 1 + (d.L + lpb - 1) / lpb + 1;
}

extern "C" int wn_bucket_range(const WnConfig* cfg, int lpb, int bucket, int64_t* lo, int64_t* hi) {
    Dims d;
    WN_TRY(check_cfg(cfg, &d));
    if (lpb < 1) lpb = d.L;
    const Lay y = make_lay(d);
    const int ngroups = (d.L + lpb - 1) / lpb;
    long a, b;
    // ONE layer bucket: the skip_1x1 parameters (laid out right in front of the layers) belong to it, not to the head -- their
    // gradients may come from the same launch as the res_1x1 gradients (k_dw_skipres8, behind the data chain), and the head
    // bucket (post-net only) is final, and its event recorded, before the chain starts either way.
    const bool skip_with_layers = ngroups == 1;
    if (bucket == 0) {
        a = 0;
        b = skip_with_layers ? y.skip0 : y.layers0;
    } else if (bucket <= ngroups) {
        const int g = bucket - 1;  // layers processed: L-1-g*lpb ... down
        const int first = g * lpb;
        int last = first + lpb;
        if (last > d.L) last = d.L;
        a = skip_with_layers ? y.skip0 : y.layers0 + (long)first * y.LB;
        b = y.layers0 + (long)last * y.LB;
    } else if (bucket == ngroups + 1) {
        a = y.causal_w;
        b = y.total;
    } else {
        return fail(2, "bucket %d out of range", bucket);
    }
    if (lo) *lo = a;
    if (hi) *hi = b;
    return 0;
}

extern "C" int wn_dead_param_range(const WnConfig* cfg, int64_t* lo, int64_t* hi) {
    Dims d;
    WN_TRY(check_cfg(cfg, &d));
    const Lay y = make_lay(d);
    const long lb = layer_base(y, d, d.L - 1);
    if (lo) *lo = lb + y.o_res_w;
    if (hi) *hi = lb + y.o_res_b + d.R;
    return 0;
}

// ------------------------------------------------------------------------------------------
// split-K plan for the weight-gradient GEMMs (contraction over time)
// ------------------------------------------------------------------------------------------
struct DwPlan {
    int ksplit, kchunk, nz;
};
// Split-K plan of the fused skip + res launch (wn_gemm6.h WnDwSkipRes): one resident round of its workgroups (512 threads, 96 KB
// of LDS: one per CU)
static DwPlan dw_skipres_plan(int S, int nl, int Kdim, int nbatch) {
    const long tiles = (long)(S / 256) * ((nl + 1) / 2) * nbatch;
    long ks = 256 / (tiles > 0 ? tiles : 1);
    const long maxks = (Kdim + 255) / 256;
    if (ks > maxks) ks = maxks;
    if (ks < 1) ks = 1;
    int kchunk = (int)((Kdim + ks - 1) / ks);
    kchunk = (kchunk + 31) / 32 * 32;
    DwPlan p;
    p.kchunk = kchunk;
    p.ksplit = (Kdim + kchunk - 1) / kchunk;
    if (p.ksplit < 1) p.ksplit = 1;
    p.nz = p.ksplit * nbatch;
    return p;
}
static DwPlan dw_plan(int M, int N, int Kdim, int nbatch) {
    const int tm = (M + (M > 64 ? 127 : 63)) / (M > 64 ? 128 : 64);
    // column tile width the kernel will use (the 256-row tiles of wn_gemm6_dw_tall always come with 128 columns)
    const int tnw = wn_gemm6_dw_tall(M, N) ? 128 : 64 * wn_gemm6_dw_tn(M, N);
    const int tn = (N + tnw - 1) / tnw;
    const long tiles = (long)tm * tn * nbatch;
    // 128 x 128 tiles (k_gemm6_dw<2,2>, 3 workgroups per CU): as many k-chunks as fit ONE resident round of 768
    // workgroups -- measured against "at least 1024" on config 2 (720 instead of 1200 workgroups for the layer-batched
    // launches, 768 instead of 1024 for the post-net ones): same time for dw_dilated / dw_skip, -15 % for dw_post1/2 and
    // for the reductions of the fewer partials.  64-wide tiles keep the old rule (dw_res: 0.54 vs 0.59 ms).
    long ks;
    if (wn_gemm6_dw_big(M, N)) {    // 256 x 256 tiles (k_gemm6_dw<4,4>, 1 workgroup per CU): one resident round of 256
        const long big = (long)(M / 256) * (N / 256) * nbatch;
        ks = 256 / big;
    } else if (wn_gemm6_dw_tall(M, N)) {   // 256 x 128 tiles (k_gemm6_dw<4,2>, 2 workgroups per CU): one resident round of 512
        const long tall = (long)(M / 256) * tn * nbatch;
        ks = 512 / tall;
    } else if (M > 64 && tnw == 192) {   // 128 x 192 tiles (k_gemm6_dw<2,3>, 2 workgroups per CU): one resident round of 512
        ks = 512 / tiles;
    } else if (M > 64 && tnw == 128) {
        ks = 768 / tiles;
    } else {
        ks = (1024 + tiles - 1) / tiles;
    }
    const long maxks = (Kdim + 255) / 256;
    if (ks > maxks) ks = maxks;
    if (ks < 1) ks = 1;
    int kchunk = (int)((Kdim + ks - 1) / ks);
    kchunk = (kchunk + 31) / 32 * 32;
    if (kchunk < 32) kchunk = 32;
    DwPlan p;
    p.kchunk = kchunk;
    p.ksplit = (Kdim + kchunk - 1) / kchunk;
    if (p.ksplit < 1) p.ksplit = 1;
    p.nz = p.ksplit * nbatch;
    return p;
}

// ------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------
struct Ws {
    // packed weights
    long wc_f, wd_f, waux_f, cvec, rowsum_aux, wres_f, wskip_f, bskip, w1_f, w2_f, wd_b, one, apk;
    // saved activations
    long X, G, Sg, Gt, Z, O1, O2;
    // scratch
    long P, dO2, dSk, dZ, dXall, dG, dw_partial, dc, tmpS, partial, rs_partial, red_scratch, loss_partial;
    long partial2, rs_partial2;   // res_1x1 partials of the fused skip + res weight-gradient launch (wn_dw_skipres_launch)
    long amax_partial, amax_partial_floats;
    long dw_ovf;   // one word (of 64): raised by an fp16-pair weight-gradient launch whose result was not finite (WN_FLAG_DW_F16PAIR)
    long dGp, qp;  // aux-gradient partials of the gate kernel (WN_FLAG_AUX_FUSED); 0 floats when the mode cannot apply
    long img_fwd16, img16_floats;  // two-piece fp16 images of the fused forward block (WN_FLAG_FUSED_F16PAIR)
    long img_taps16, img_res16, img_taps16_floats, img_res16_floats;   // ... of the backward chain (WN_FLAG_CHAIN_F16PAIR)
    long amaxP, amaxP_lfloats;     // max |dP_l| per 32-sample tile, [L][B * ceil(T / 32)]
    long img_fwd, img_taps, img_res, img_floats;  // pre-split LDS weight images of the fused split kernels (0 floats: not applicable)
    long wskipT_f, dZs;  // chain mode (wn_fused_chain_supported): skip weights as [s][l*R + i], dZs = Wskip^T dSkip (B, L*R, T)
    long dZs_floats;
    long red_scratch_floats;
    long apk_floats;
    long apk_pre[6];   // the six weight sets of a training step, split ONCE per step (pack_weights): offsets, -1 = none
    long apk_wide[5];  // wide models (n_resch % 128 == 0, any-size path): the five per-layer weight sets of a step, all layers, split
    long apk_pre16[6], apk_wide16[5];   // the same sets as two-piece fp16 images (WN_FLAG_MM_F16PAIR; same sizes and layer strides)
    long apk_wide_l[5];   // once per step as well (round 3: 149 little pack launches per step); offset of layer 0 / floats per layer
    long front_partial, front_partial_floats;
    long total;
    int F;  // frames (T/U, or T without upsampling)
};

static inline long al64(long n) { return (n + 63) / 64 * 64; }

static int make_ws(const Dims& d, int B, int T, Ws* w, bool training = true) {
    if (B < 1 || T < 1) return fail(1, "B and T must be positive");
    const int Ue = d.U > 0 ? d.U : 1;
    if (T % Ue != 0) return fail(1, "T=%d is not a multiple of upsampling_factor=%d", T, d.U);
    const int F = T / Ue;
    w->F = F;
    const long BRT = (long)B * d.R * T, BST = (long)B * d.S * T;
    long o = 0;
#define CARVE(field, n) \
    w->field = o;       \
    o += al64((long)(n));
    CARVE(wc_f, (long)d.K * d.Q * d.R);
    CARVE(wd_f, (long)d.L * d.K * d.R * 2 * d.R);
    CARVE(waux_f, (long)d.A * d.L * 2 * d.R);
    CARVE(cvec, (long)d.L * 2 * d.R);
    CARVE(rowsum_aux, (long)d.L * 2 * d.R);
    CARVE(wres_f, (long)d.L * d.R * d.R);
    CARVE(wskip_f, (long)d.L * d.R * d.S);
    CARVE(bskip, d.S);
    CARVE(w1_f, (long)d.S * d.S);
    CARVE(w2_f, (long)d.S * d.Qo);
    CARVE(wd_b, (long)d.L * d.K * 2 * d.R * d.R);
    CARVE(one, 64);
    CARVE(X, (long)d.L * BRT);
    CARVE(G, (long)B * d.L * 2 * d.R * F);
    CARVE(Sg, (long)d.L * BRT);
    CARVE(Gt, (long)d.L * BRT);
    CARVE(Z, (long)d.L * BRT);
    CARVE(O1, BST);
    CARVE(O2, BST);
    CARVE(P, (long)d.L * 2 * BRT);  // forward scratch (layered path) / dP of every layer (backward)
    CARVE(dO2, BST);
    CARVE(dSk, BST);
    CARVE(dZ, BRT);
    CARVE(dXall, (long)d.L * BRT);  // dL/dx_l of every layer
    CARVE(dG, (long)d.L * B * 2 * d.R * F);
    CARVE(dw_partial, (long)d.L * B * 2 * d.R * Ue);
    {
        const bool auxf = wn_fused_supported(d.R, d.K, d.S) && d.U >= 16 && d.U % 16 == 0;
        CARVE(dGp, auxf ? (long)d.L * B * 2 * d.R * (T / 16) : 0);
        CARVE(qp, auxf ? (long)d.L * B * T : 0);
    }
    {
        const bool img = wn_fused_supported(d.R, d.K, d.S) && wn_fused_image_floats(d.K, d.L, 0) > 0;
        w->img_floats = img ? wn_fused_image_floats(d.K, d.L, 0) : 0;
        CARVE(img_fwd, w->img_floats);
        CARVE(img_taps, img ? wn_fused_image_floats(d.K, d.L, 1) : 0);
        CARVE(img_res, img ? wn_fused_image_floats(d.K, d.L, 2) : 0);
        w->img16_floats = img ? wn_fused_image16_floats(d.K, d.L, 0) : 0;
        CARVE(img_fwd16, w->img16_floats);
        w->img_taps16_floats = img ? wn_fused_image16_floats(d.K, d.L, 1) : 0;
        CARVE(img_taps16, w->img_taps16_floats);
        w->img_res16_floats = img ? wn_fused_image16_floats(d.K, d.L, 2) : 0;
        CARVE(img_res16, w->img_res16_floats);
        w->amaxP_lfloats = img ? al64((long)B * ((T + 31) / 32)) : 0;
        CARVE(amaxP, w->amaxP_lfloats * d.L);
    }
    {
        const bool chain = wn_fused_chain_supported(d.R, d.K, d.S) && d.L > 1;
        w->dZs_floats = chain ? (long)B * d.L * d.R * T : 0;
        CARVE(wskipT_f, chain ? (long)d.S * d.L * d.R : 0);
        CARVE(dZs, w->dZs_floats);
    }
    CARVE(dc, (long)d.L * 2 * d.R);
    CARVE(tmpS, d.S > d.Qo ? d.S : d.Qo);
    // partial buffers: max over the dW GEMMs issued by wn_backward
    long pmax = 0, rmax = 0;
    {
        struct { int M, N, K; } gs[] = {
            {d.Qo, d.S, T}, {d.S, d.S, T}, {d.S, d.L * d.R, T}, {2 * d.R, d.K * d.R, T},
            {d.R, d.R, T}, {2 * d.R, d.A, F}, {2 * d.R, d.A, T}, {d.R, d.K * d.Q, T}};
        for (unsigned i = 0; i < sizeof(gs) / sizeof(gs[0]); ++i) {
            for (int nl = 1; nl <= d.L; ++nl) {  // layer-batched launches: any bucket size
                DwPlan p = dw_plan(gs[i].M, gs[i].N, gs[i].K, B * nl);
                long need = (long)p.nz * gs[i].M * gs[i].N;
                if (need > pmax) pmax = need;
                long rneed = (long)p.nz * gs[i].M;
                if (rneed > rmax) rmax = rneed;
            }
        }
    }
    long p2max = 0, r2max = 0;
    for (int nl = 1; nl <= d.L; ++nl) {   // the fused skip + res launch of any bucket size: its own plan for both outputs
        if (!wn_dw_skipres_supported(d.S, d.R, nl, nl > 1 ? nl - 1 : 1)) continue;
        const DwPlan p = dw_skipres_plan(d.S, nl, T, B);
        const long need = (long)p.nz * d.S * d.R * nl, rneed = (long)p.nz * d.S;
        if (need > pmax) pmax = need;
        if (rneed > rmax) rmax = rneed;
        if ((long)p.nz * nl * d.R * d.R > p2max) p2max = (long)p.nz * nl * d.R * d.R;
        if ((long)p.nz * nl * d.R > r2max) r2max = (long)p.nz * nl * d.R;
    }
    CARVE(partial, pmax);
    CARVE(rs_partial, rmax);
    CARVE(partial2, p2max);
    CARVE(rs_partial2, r2max);
    w->red_scratch_floats = 1 << 20;
    CARVE(red_scratch, w->red_scratch_floats);
    CARVE(loss_partial, 2 * wn_softmax_ce_nblocks(B, T) + 64);   // CE epilogue: one partial per 128-column block
    CARVE(dw_ovf, 64);   // [0] overflow word, [1] a_mul, [2] max |dlogits| of the last loss call (wn_dw_prepare, wn_elem.h)
    // block maxima of |dlogits|: one per loss block (the loss calls), or one per (row, 4096 columns) of a dlogits scan (wn_backward)
    w->amax_partial_floats = 2 * wn_softmax_ce_nblocks(B, T) + 64 + (long)B * d.Qo * ((T + 4095) / 4096);
    CARVE(amax_partial, w->amax_partial_floats);
    w->front_partial_floats = wn_front_dw_supported(d.R, d.K, d.Q) ? wn_front_dw_partial_floats(B, T, d.R, d.K, d.Q) : 0;
    CARVE(front_partial, w->front_partial_floats);
    {   // split-bf16 weights of the forward-type contractions (wn_gemm6): one buffer, re-packed before each use
        const int mk[][2] = {{d.S, d.L * d.R}, {d.S, d.S}, {d.Qo, d.S}, {d.S, d.Qo}, {2 * d.R, d.K * d.R},
                             {d.R, d.R}, {d.R, d.S}, {d.R, d.K * 2 * d.R}, {d.L * d.R, d.S}};
        long e = 0;
        for (unsigned i = 0; i < sizeof(mk) / sizeof(mk[0]); ++i) {
            const long ei = wn_gemm6_apk_elems(mk[i][0], mk[i][1]);
            if (ei > e) e = ei;
        }
        w->apk_floats = (e + 1) / 2;
        CARVE(apk, w->apk_floats);
        // ... and one buffer each for the weight sets every step uses (same order as pre_jobs())
        const int pre[6][2] = {{d.S, d.L * d.R}, {d.S, d.S}, {d.Qo, d.S}, {d.S, d.Qo}, {d.S, d.S}, {d.L * d.R, d.S}};
        for (int i = 0; i < 6; ++i) {
            w->apk_pre[i] = -1;
            if (pre[i][0] < 128 || (i == 5 && w->dZs_floats <= 0)) continue;   // (the split contraction wants M >= 128)
            w->apk_pre[i] = o;
            o += al64((wn_gemm6_apk_elems(pre[i][0], pre[i][1]) + 1) / 2);
        }
        for (int i = 0; i < 6; ++i) {
            w->apk_pre16[i] = -1;
            if (w->apk_pre[i] < 0) continue;
            w->apk_pre16[i] = o;
            o += al64((wn_gemm6_apk_elems(pre[i][0], pre[i][1]) + 1) / 2);
        }
    }
        {   // (same order as wide_jobs())
            const bool wide = training && d.R % 128 == 0 && !wn_fused_supported(d.R, d.K, d.S);
            const int mk[5][2] = {{2 * d.R, d.K * d.R}, {d.R, d.R}, {d.R, d.S}, {d.R, d.R}, {d.R, d.K * 2 * d.R}};
            for (int i = 0; i < 5; ++i) {
                w->apk_wide[i] = -1;
                w->apk_wide_l[i] = al64((wn_gemm6_apk_elems(mk[i][0], mk[i][1]) + 1) / 2);
                if (!wide) continue;
                w->apk_wide[i] = o;
                o += w->apk_wide_l[i] * d.L;
            }
            for (int i = 0; i < 5; ++i) {
                w->apk_wide16[i] = -1;
                if (!wide) continue;
                w->apk_wide16[i] = o;
                o += w->apk_wide_l[i] * d.L;
            }
        }
#undef CARVE
    w->total = o;
    return 0;
}

extern "C" size_t wn_workspace_bytes(const WnConfig* cfg, int B, int T) {
    Dims d;
    if (check_cfg(cfg, &d)) return 0;
    Ws w;
    if (make_ws(d, B, T, &w)) return 0;
    return (size_t)w.total * sizeof(float);
}

// Where the tensors a wn_forward / wn_backward pair leaves in the workspace live (parity tests compare them with the
// oracle's intermediates; the training path itself never calls this).  kind: see WN_WS_* in the header.
extern "C" int wn_workspace_region(const WnConfig* cfg, int B, int T, int kind, int64_t* offset_floats, int64_t* n_floats) {
    api_enter();
    Dims d;
    WN_TRY(check_cfg(cfg, &d));
    Ws w;
    WN_TRY(make_ws(d, B, T, &w));
    if (!offset_floats || !n_floats) return fail(1, "NULL argument");
    const long BRT = (long)B * d.R * T, BST = (long)B * d.S * T;
    long off, n;
    switch (kind) {
        case WN_WS_X: off = w.X; n = (long)d.L * BRT; break;
        case WN_WS_SIGMOID: off = w.Sg; n = (long)d.L * BRT; break;
        case WN_WS_TANH: off = w.Gt; n = (long)d.L * BRT; break;
        case WN_WS_Z: off = w.Z; n = (long)d.L * BRT; break;
        case WN_WS_RELU_SKIP: off = w.O1; n = BST; break;
        case WN_WS_RELU_POST1: off = w.O2; n = BST; break;
        case WN_WS_DSKIP: off = w.dSk; n = BST; break;
        case WN_WS_DP: off = w.P; n = (long)d.L * 2 * BRT; break;
        case WN_WS_DX: off = w.dXall; n = (long)d.L * BRT; break;
        default: return fail(1, "unknown workspace region %d", kind);
    }
    *offset_floats = off;
    *n_floats = n;
    return 0;
}

struct Ctx {
    const WnConfig* cfg;
    Dims d;
    Lay y;
    Ws w;
    int B, T;
    float* ws;
    wn_stream_t st;
    bool fused;
    bool split_bf16;  // forward-type contractions on the bf16 matrix cores (3-way split, fp32-equivalent)
    int dw_products;  // products per multiply of the weight-gradient contractions: 6, or 3 with WN_FLAG_DW_3PRODUCT
    bool chain_f16;   // WN_FLAG_CHAIN_F16PAIR: the backward chain kernel on the block-scaled fp16 pair split (k_chain64s<.., H16>)
    bool fused_f16;   // WN_FLAG_FUSED_F16PAIR: the fused 64-channel forward block on the fp16 pair split (block-scaled, k_resblock_fwd_h)
    bool mm_f16;      // WN_FLAG_MM_F16PAIR: the forward / data-gradient split contractions (k_gemm6) take the fp16 pair split as well, each
                      // followed by its conditional six-product redo
    int dw_f16_mode;  // WN_FLAG_DW_F16PAIR / WN_FLAG_MM_F16PAIR: 0 off; 1 the caller's exponent (| WN_FLAG_DW_F16_EXP_VALID); 2 max |dlogits| as the loss call of
                      // this workspace measured it (| WN_FLAG_DW_F16_AMAX_WS); 3 measured by a scan of the dlogits given to wn_backward
    float dw_f16_mul; // != 0 (WN_FLAG_DW_F16PAIR): weight gradients by the fp16 pair split; -1: the gradient operand times the power of two
                      // wn_dw_prepare leaves in the workspace (every mode: one code path);
    int* dw_ovf;      // their overflow word (workspace): a raised word makes the six-product launch behind each of them do the work
    const float* params;   // set by the training entry points: lets fw_gemm recognise the pre-split weight sets
    bool have_pre;         // apk_pre[] of this workspace is valid (regular layout, not the decode state)
};

static int make_ctx(Ctx* c, const WnConfig* cfg, int B, int T, void* ws, size_t ws_bytes, int flags, void* stream) {
    c->cfg = cfg;
    WN_TRY(check_cfg(cfg, &c->d));
    c->y = make_lay(c->d);
    WN_TRY(make_ws(c->d, B, T, &c->w));
    if (!ws) return fail(1, "workspace is NULL");
    if (ws_bytes < (size_t)c->w.total * sizeof(float))
        return fail(1, "workspace too small: %zu < %zu bytes", ws_bytes, (size_t)c->w.total * sizeof(float));
    c->B = B;
    c->T = T;
    c->ws = (float*)ws;
    c->st = (wn_stream_t)stream;
    c->fused = wn_fused_supported(c->d.R, c->d.K, c->d.S) && !(flags & WN_FLAG_NO_FUSED);
    c->split_bf16 = !(flags & WN_FLAG_EXACT_MFMA);
    c->dw_products = (flags & WN_FLAG_DW_3PRODUCT) ? 3 : 6;
    c->mm_f16 = c->split_bf16 && (flags & WN_FLAG_MM_F16PAIR);
    c->fused_f16 = c->fused && c->split_bf16 && (flags & WN_FLAG_FUSED_F16PAIR) && c->w.img16_floats > 0;
    c->chain_f16 = c->fused && c->split_bf16 && (flags & WN_FLAG_CHAIN_F16PAIR) && c->w.img_taps16_floats > 0;
    c->dw_f16_mode = !((flags & WN_FLAG_DW_F16PAIR) || c->mm_f16) ? 0 : (flags & WN_FLAG_DW_F16_EXP_VALID) ? 1 : (flags & WN_FLAG_DW_F16_AMAX_WS) ? 2 : 3;
    c->dw_f16_mul = (flags & WN_FLAG_DW_F16PAIR) ? -1.0f : 0.0f;
    c->dw_ovf = reinterpret_cast<int*>(c->ws + c->w.dw_ovf);
    c->params = nullptr;
    c->have_pre = true;
    return 0;
}

// ------------------------------------------------------------------------------------------
// weight packing (once per forward; weights change every optimizer step)
// ------------------------------------------------------------------------------------------
// Weights x activations contraction: split-bf16 matrix-core kernel when the launch has its shape
// (>= 128 output rows, k-minor operands, no shifts), the exact-f32 MFMA kernel otherwise.
struct GateEpi {   // optional gate epilogue of a split contraction (wn_gemm6.h); all NULL = plain
    int gate_R = 0;
    float *S = nullptr, *Gt = nullptr, *Z = nullptr;
    const float* G = nullptr;
    long g_bstride = 0;
    int F = 0, U = 1;
    const float *upw = nullptr, *cvec = nullptr;
    const float *bw_S = nullptr, *bw_Gt = nullptr;
    float* bw_dP = nullptr;
};
static int dzs_layers(const Dims& d) { return d.L; }   // layers bwd_dz_skip_all contracts (all: the head of the chain takes its rows)

// The weight sets of the split contractions every training step launches: (A, lda, M, K) and where their split form lives.
// They are split ONCE per step by one launch at the end of pack_weights (six dependent little launches in front of the
// contractions otherwise); wn_backward finds them in the workspace wn_forward left.
struct PreJob { const float* A; long lda; int M, K; long off, off16; };
static int pre_jobs(const Ctx& c, const float* params, PreJob (&j)[6]) {
    const Dims& d = c.d;
    const Lay& y = c.y;
    const Ws& w = c.w;
    float* ws = c.ws;
    const PreJob all[6] = {{ws + w.wskip_f, d.S, d.S, d.L * d.R, w.apk_pre[0], w.apk_pre16[0]},
                           {ws + w.w1_f, d.S, d.S, d.S, w.apk_pre[1], w.apk_pre16[1]},
                           {ws + w.w2_f, d.Qo, d.Qo, d.S, w.apk_pre[2], w.apk_pre16[2]},
                           {params ? params + y.post2_w : nullptr, d.S, d.S, d.Qo, w.apk_pre[3], w.apk_pre16[3]},
                           {params ? params + y.post1_w : nullptr, d.S, d.S, d.S, w.apk_pre[4], w.apk_pre16[4]},
                           {ws + w.wskipT_f, (long)d.L * d.R, dzs_layers(d) * d.R, d.S, w.apk_pre[5], w.apk_pre16[5]}};
    int n = 0;
    if (!c.have_pre || !c.split_bf16) return 0;
    for (int i = 0; i < 6; ++i)
        if (all[i].A && all[i].off >= 0) j[n++] = all[i];
    return n;
}
// The per-layer weight sets of a wide model's step (any-size path with the split contractions, n_resch % 128 == 0): layer l of
// set i lives at A + l * lstride (floats) and its split form at apk_wide[i] + l * apk_wide_l[i].  Set 0 is packed with the
// gate row permutation (gate_R = R: the forward gate epilogue), the others plainly.
struct WideJob { const float* A; long lstride, lda; int M, K, gate_R, nl; long off, off_l, off16; };
static int wide_jobs(const Ctx& c, const float* params, WideJob (&j)[5]) {
    const Dims& d = c.d;
    const Lay& y = c.y;
    const Ws& w = c.w;
    if (!c.have_pre || !c.split_bf16 || c.fused || w.apk_wide[0] < 0 || !params) return 0;
    const long lb0 = layer_base(y, d, 0);
    const WideJob all[5] = {
        {c.ws + w.wd_f, (long)d.K * d.R * 2 * d.R, 2 * d.R, 2 * d.R, d.K * d.R, d.R, d.L, w.apk_wide[0], w.apk_wide_l[0], w.apk_wide16[0]},   // fwd_dilated_gate
        {c.ws + w.wres_f, (long)d.R * d.R, d.R, d.R, d.R, 0, d.L, w.apk_wide[1], w.apk_wide_l[1], w.apk_wide16[1]},                          // fwd_res
        {params + y.skip0, y.ls_skip, d.R, d.R, d.S, 0, d.L, w.apk_wide[2], w.apk_wide_l[2], w.apk_wide16[2]},                               // bwd_dz_skip
        {params + lb0 + y.o_res_w, -y.LB, d.R, d.R, d.R, 0, d.L, w.apk_wide[3], w.apk_wide_l[3], w.apk_wide16[3]},                           // bwd_dz_res
        {c.ws + w.wd_b, (long)d.K * 2 * d.R * d.R, d.R, d.R, d.K * 2 * d.R, 0, d.L, w.apk_wide[4], w.apk_wide_l[4], w.apk_wide16[4]}};      // bwd_dx_dilated
    for (int i = 0; i < 5; ++i) j[i] = all[i];
    return 5;
}
static long prepacked_offset(const Ctx& c, const WnGemmArgs& g, int gate_R = 0, long* off16 = nullptr) {
    if (off16) *off16 = -1;
    if (gate_R == 0) {
        PreJob j[6];
        const int n = pre_jobs(c, c.params, j);
        for (int i = 0; i < n; ++i)
            if (j[i].A == g.A && j[i].lda == g.lda && j[i].M == g.M && j[i].K == g.K) {
                if (off16) *off16 = j[i].off16;
                return j[i].off;
            }
    }
    WideJob wj[5];
    const int nw = wide_jobs(c, c.params, wj);
    for (int i = 0; i < nw; ++i) {
        if (wj[i].lda != g.lda || wj[i].M != g.M || wj[i].K != g.K || wj[i].gate_R != gate_R || wj[i].lstride == 0) continue;
        const long diff = g.A - wj[i].A;
        if (diff % wj[i].lstride != 0) continue;
        const long l = diff / wj[i].lstride;
        if (l >= 0 && l < wj[i].nl) {
            if (off16 && wj[i].off16 >= 0) *off16 = wj[i].off16 + l * wj[i].off_l;
            return wj[i].off + l * wj[i].off_l;
        }
    }
    return -1;
}

static bool fw_gemm_split_ok(const Ctx& c, const WnGemmArgs& g) {
    return c.split_bf16 && g.M >= 128 && !g.a_kmajor && !g.b_kmajor &&
           (g.b_seg_len >= g.K || g.b_seg_len % 16 == 0) && g.ksplit == 1 && g.nlayer == 1 && !g.b_relu &&
           !g.b_index && g.a_zstride == 0 && !g.a_rowsum &&
           wn_gemm6_apk_elems(g.M, g.K) <= 2 * c.w.apk_floats && (long)g.M * g.ldc * 4 < 0x7ffffff0L;
}
struct CeEpi {   // softmax cross-entropy as the epilogue of the contraction that produces the logits (wn_gemm6.h)
    const int64_t* target;
    int t_start;
    float gs;
    float* partial;
    float* amax;
};
// n_origin: index of column 0 of this launch in the caller's full (B, C, T) tensor (a loss-window launch starts at t0): the
// alternating tile signs of k_gemm6 follow the ABSOLUTE column, so a windowed launch produces bit for bit what the full one
// produces on those columns (same ReLU masks in the training step's and the module's forward).
// grad_b: the B operand is a back-propagated gradient (fp16 pair mode: scaled by the measured 2^8 / max |dlogits|, wn_dw_prepare)
static int fw_gemm(const Ctx& c, const WnGemmArgs& g, const GateEpi* ge = nullptr, const CeEpi* ce = nullptr, int n_origin = 0,
                   bool grad_b = false) {
    const bool ok = c.split_bf16 && g.M >= 128 && !g.a_kmajor && !g.b_kmajor &&
                    (g.b_seg_len >= g.K || g.b_seg_len % 16 == 0) && g.ksplit == 1 && g.nlayer == 1 && !g.b_relu &&
                    !g.b_index && g.a_zstride == 0 && !g.a_rowsum &&
                    wn_gemm6_apk_elems(g.M, g.K) <= 2 * c.w.apk_floats && (long)g.M * g.ldc * 4 < 0x7ffffff0L;
    if (!ok) return (ge || ce) ? fail(3, "gate / loss epilogue needs the split contraction") : wn_gemm_launch(&g, c.st);
    unsigned short* apk = reinterpret_cast<unsigned short*>(c.ws + c.w.apk);
    long pre16 = -1;
    const long pre = prepacked_offset(c, g, ge ? ge->gate_R : 0, &pre16);   // (gate' epilogues use the plain packing: gate_R = 0)
    if (pre >= 0)
        apk = reinterpret_cast<unsigned short*>(c.ws + pre);   // split once per step by pack_weights
    else
        WN_TRY(wn_gemm6_pack(g.A, g.lda, g.M, g.K, apk, ge ? ge->gate_R : 0, c.st));
    WnGemm6Args a;
    wn_gemm6_no_gate(&a);
    if (ge) {
        a.gate_R = ge->gate_R; a.gate_S = ge->S; a.gate_Gt = ge->Gt; a.gate_Z = ge->Z; a.gate_G = ge->G; a.gate_gb = ge->g_bstride;
        a.gate_F = ge->F; a.gate_U = ge->U; a.gate_upw = ge->upw; a.gate_cvec = ge->cvec;
        a.gbw_S = ge->bw_S; a.gbw_Gt = ge->bw_Gt; a.gbw_dP = ge->bw_dP;
    }
    a.M = g.M; a.N = g.N; a.K = g.K;
    a.Apk = apk; a.Mpad = (g.M + WN_G6_BM - 1) / WN_G6_BM * WN_G6_BM;
    a.B = g.B; a.ldb = g.ldb; a.b_zstride = g.b_zstride; a.b_seg_len = g.b_seg_len; a.b_seg_stride = g.b_seg_stride;
    a.b_shift0 = g.b_shift0; a.b_shift_step = g.b_shift_step; a.b_clen = g.b_clen;
    a.D = g.D; a.ldd = g.ldd; a.d_zstride = g.d_zstride; a.accumulate = g.accumulate;
    a.C = g.C; a.ldc = g.ldc; a.c_zstride = g.c_zstride;
    a.bias = g.bias; a.E = g.E; a.lde = g.lde; a.e_zstride = g.e_zstride; a.relu = g.relu;
    a.nbatch = g.nbatch; a.tag = g.tag;
    a.n_phase = (n_origin / WN_G6_BN) & 1;
    if (ce) {
        a.ce_target = reinterpret_cast<const long long*>(ce->target); a.ce_tstride = g.ldc; a.ce_t_start = ce->t_start;
        a.ce_gs = ce->gs; a.ce_partial = ce->partial; a.ce_amax = ce->amax;
    }
    // WN_FLAG_MM_F16PAIR: the fp16 pair split with the conditional six-product redo behind it -- for the weight sets pack_weights
    // split both ways, and only where a redo may simply run again (no C += result, no in-place residual)
    const bool writes_acc = g.accumulate && !(ge && ge->bw_dP);
    if (c.mm_f16 && c.dw_ovf && pre >= 0 && pre16 >= 0 && !writes_acc && !(g.D && g.D == g.C)) {
        WnGemm6Args h = a;
        h.f16 = 1; h.Apk = reinterpret_cast<unsigned short*>(c.ws + pre16); h.b_mul = grad_b ? -1.0f : 16.0f; h.ovf = c.dw_ovf;   // activations times 2^4: second pieces normal down to 2^-7
        WN_TRY(wn_gemm6_launch(&h, c.st));
        a.ovf = c.dw_ovf;   // f16 = 0 with ovf: works only if the launch above raised the word
    }
    return wn_gemm6_launch(&a, c.st);
}

static int pack_weights(const Ctx& c, const float* params) {
    const Dims& d = c.d;
    const Lay& y = c.y;
    const Ws& w = c.w;
    float* ws = c.ws;
    const long lb0 = layer_base(y, d, 0);
    const long lstep = -y.LB;  // layer l block = lb0 + l*lstep
    WnCopy4 cp;
    WnCopy4Batch jobs;  // all re-layouts below (11 of them): one launch
    jobs.njobs = 0;
    // wc_f[tap][q][r] = causal_w[r][q][tap]
    cp.n0 = d.K; cp.n1 = d.Q; cp.n2 = d.R; cp.nl = 1;
    cp.d0 = (long)d.Q * d.R; cp.d1 = d.R; cp.d2 = 1; cp.dl = 0;
    cp.s0 = 1; cp.s1 = d.K; cp.s2 = (long)d.Q * d.K; cp.sl = 0;
    WN_TRY(wn_copy4_batch_add(&jobs, ws + w.wc_f, params + y.causal_w, &cp));
    // wd_f[l][(tap*R+i)*2R + o'] = W{sig,tanh}[o][i][tap] ;  wd_b[l][(tap*2R+o')*R + i] = same
    for (int half = 0; half < 2; ++half) {
        const long src = lb0 + (half ? y.o_dtanh_w : y.o_dsig_w);
        cp.n0 = d.K; cp.n1 = d.R /*i*/; cp.n2 = d.R /*o*/; cp.nl = d.L;
        cp.s0 = 1; cp.s1 = d.K; cp.s2 = (long)d.R * d.K; cp.sl = lstep;
        cp.d0 = (long)d.R * 2 * d.R; cp.d1 = 2 * d.R; cp.d2 = 1; cp.dl = (long)d.K * d.R * 2 * d.R;
        WN_TRY(wn_copy4_batch_add(&jobs, ws + w.wd_f + (long)half * d.R, params + src, &cp));
        cp.d0 = (long)2 * d.R * d.R; cp.d1 = 1; cp.d2 = d.R; cp.dl = (long)d.K * 2 * d.R * d.R;
        WN_TRY(wn_copy4_batch_add(&jobs, ws + w.wd_b + (long)half * d.R * d.R, params + src, &cp));
        // waux_f[a][l*2R + o'] = Waux{sig,tanh}_l[o][a]
        const long asrc = lb0 + (half ? y.o_atanh_w : y.o_asig_w);
        cp.n0 = 1; cp.n1 = d.A; cp.n2 = d.R; cp.nl = d.L;
        cp.s0 = 0; cp.s1 = 1; cp.s2 = d.A; cp.sl = lstep;
        cp.d0 = 0; cp.d1 = (long)d.L * 2 * d.R; cp.d2 = 1; cp.dl = 2 * d.R;
        WN_TRY(wn_copy4_batch_add(&jobs, ws + w.waux_f + (long)half * d.R, params + asrc, &cp));
    }
    // wres_f[l][i*R + o] = Wres_l[o][i]
    cp.n0 = 1; cp.n1 = d.R; cp.n2 = d.R; cp.nl = d.L;
    cp.s0 = 0; cp.s1 = 1; cp.s2 = d.R; cp.sl = lstep;
    cp.d0 = 0; cp.d1 = d.R; cp.d2 = 1; cp.dl = (long)d.R * d.R;
    WN_TRY(wn_copy4_batch_add(&jobs, ws + w.wres_f, params + lb0 + y.o_res_w, &cp));
    // wskip_f[(l*R + r)*S + s] = Wskip_l[s][r]
    cp.n0 = 1; cp.n1 = d.R; cp.n2 = d.S; cp.nl = d.L;
    cp.s0 = 0; cp.s1 = 1; cp.s2 = d.R; cp.sl = y.ls_skip;
    cp.d0 = 0; cp.d1 = d.S; cp.d2 = 1; cp.dl = (long)d.R * d.S;
    WN_TRY(wn_copy4_batch_add(&jobs, ws + w.wskip_f, params + y.skip0, &cp));
    if (w.dZs_floats > 0) {  // wskipT_f[s*(L*R) + l*R + i] = Wskip_l[s][i]: the A operand of dZs = Wskip^T dSkip (all layers)
        cp.n0 = 1; cp.n1 = d.S; cp.n2 = d.R; cp.nl = d.L;
        cp.s0 = 0; cp.s1 = d.R; cp.s2 = 1; cp.sl = y.ls_skip;
        cp.d0 = 0; cp.d1 = (long)d.L * d.R; cp.d2 = 1; cp.dl = d.R;
        WN_TRY(wn_copy4_batch_add(&jobs, ws + w.wskipT_f, params + y.skip0, &cp));
    }
    // w1_f[i*S + o] = W1[o][i] ; w2_f[i*Q + q] = W2[q][i]
    cp.n0 = 1; cp.n1 = d.S; cp.n2 = d.S; cp.nl = 1;
    cp.s0 = 0; cp.s1 = 1; cp.s2 = d.S; cp.sl = 0;
    cp.d0 = 0; cp.d1 = d.S; cp.d2 = 1; cp.dl = 0;
    WN_TRY(wn_copy4_batch_add(&jobs, ws + w.w1_f, params + y.post1_w, &cp));
    cp.n1 = d.S; cp.n2 = d.Qo; cp.s1 = 1; cp.s2 = d.S; cp.d1 = d.Qo; cp.d2 = 1;
    WN_TRY(wn_copy4_batch_add(&jobs, ws + w.w2_f, params + y.post2_w, &cp));
    WN_TRY(wn_copy4_batch(&jobs, c.st));
    if (c.fused && c.split_bf16 && w.img_floats > 0)   // LDS images of the split kernels: one launch for all layers
        WN_TRY(wn_fused_pack_images(ws + w.wd_f, ws + w.wres_f, ws + w.wd_b, params, lb0 + y.o_res_w, lstep, ws + w.img_fwd,
                                    ws + w.img_taps, ws + w.img_res, d.K, d.L, c.st));
    if (c.fused_f16 || c.chain_f16)
        WN_TRY(wn_fused_pack_images16(ws + w.wd_f, ws + w.wres_f, ws + w.wd_b, params, lb0 + y.o_res_w, lstep,
                                      c.fused_f16 ? ws + w.img_fwd16 : nullptr, c.chain_f16 ? ws + w.img_taps16 : nullptr,
                                      ws + w.img_res16, d.K, d.L, c.st));
    // cvec / rowsum_aux / bskip / one
    WnCvecArgs ca;
    ca.params = params;
    ca.off_dsig_b = lb0 + y.o_dsig_b; ca.off_dtanh_b = lb0 + y.o_dtanh_b;
    ca.off_asig_w = lb0 + y.o_asig_w; ca.off_atanh_w = lb0 + y.o_atanh_w;
    ca.off_asig_b = lb0 + y.o_asig_b; ca.off_atanh_b = lb0 + y.o_atanh_b;
    ca.ls_dil = lstep; ca.ls_aux = lstep;
    ca.off_up_b = y.up_b;
    ca.L = d.L; ca.R = d.R; ca.A = d.A;
    ca.cvec = ws + w.cvec; ca.rowsum_aux = ws + w.rowsum_aux;
    WN_TRY(wn_cvec(&ca, c.st));
    WN_TRY(wn_sum_layers(params, y.skip0 + (long)d.S * d.R, y.ls_skip, d.L, d.S, ws + w.bskip, c.st));
    WN_TRY(wn_fill(ws + w.one, 1.0f, 64, c.st));
    {   // the split form of the weight sets every step contracts with: one launch (after the re-layouts above)
        PreJob pj[6];
        const int n = pre_jobs(c, params, pj);
        WideJob wj[5];
        const int nw = wide_jobs(c, c.params, wj);   // (only the training entry points declare their params: the look-up side needs them)
        if (n + nw > 0) {
            WnGemm6PackJobs jobs;
            jobs.njobs = n + nw;
            for (int i = 0; i < n; ++i) {
                jobs.src[i] = pj[i].A; jobs.lda[i] = pj[i].lda; jobs.M[i] = pj[i].M; jobs.K[i] = pj[i].K;
                jobs.dst[i] = reinterpret_cast<unsigned short*>(ws + pj[i].off);
                wn_gemm6_pack_job_single(&jobs, i);
            }
            int nj = n + nw;
            for (int i = 0; i < nw; ++i) {   // every layer of a wide model's five per-layer sets: one launch instead of 5 L - 1
                const int q = n + i;
                jobs.src[q] = wj[i].A; jobs.lda[q] = wj[i].lda; jobs.M[q] = wj[i].M; jobs.K[q] = wj[i].K;
                jobs.dst[q] = reinterpret_cast<unsigned short*>(ws + wj[i].off);
                jobs.nl[q] = wj[i].nl; jobs.src_lstride[q] = wj[i].lstride; jobs.dst_lstride[q] = 2 * wj[i].off_l;
                jobs.gate_R[q] = wj[i].gate_R; jobs.f16[q] = 0;
            }
            if (c.mm_f16) {   // ... and as two-piece fp16 images (the bf16 ones stay: the redo launches contract with them)
                for (int q = 0; q < n + nw; ++q) {
                    const long o16 = q < n ? pj[q].off16 : wj[q - n].off16;
                    if (o16 < 0 || nj >= WN_G6_PACK_MAXJOBS) continue;
                    jobs.src[nj] = jobs.src[q]; jobs.lda[nj] = jobs.lda[q]; jobs.M[nj] = jobs.M[q]; jobs.K[nj] = jobs.K[q];
                    jobs.nl[nj] = jobs.nl[q]; jobs.src_lstride[nj] = jobs.src_lstride[q]; jobs.dst_lstride[nj] = jobs.dst_lstride[q];
                    jobs.gate_R[nj] = jobs.gate_R[q]; jobs.f16[nj] = 1;
                    jobs.dst[nj] = reinterpret_cast<unsigned short*>(ws + o16);
                    ++nj;
                }
            }
            jobs.njobs = nj;
            WN_TRY(wn_gemm6_pack_batch(&jobs, c.st));
        }
    }
    return rt_check("pack_weights");
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// front conv, aux projection and the residual stack: leaves X_l, s_l, g_l, z_l of every layer in the workspace
// skip-sum of layers [lo, hi) into O1: O1 = (lo == 0 ? b_skip : O1) + sum_l Wskip_l z_l, relu when `last`
// (wavenet.py:533,238,519).  One launch over all layers is the serial form; wn_forward's overlap mode issues it in
// chunks on the side stream while the residual stack is still running.
static int skip_sum(const Ctx& c, int lo, int hi, bool last, int t0 = 0) {   // t0: columns [t0, T) only (loss window)
    const Dims& d = c.d;
    const Ws& w = c.w;
    float* ws = c.ws;
    const long BRT = (long)c.B * d.R * c.T;
    WnGemmArgs g = wn_gemm_default();
    g.M = d.S; g.N = c.T - t0; g.K = (hi - lo) * d.R;
    g.A = ws + w.wskip_f + (long)lo * d.R * d.S; g.lda = d.S;
    g.B = ws + w.Z + (long)lo * BRT + t0; g.ldb = c.T; g.b_zstride = (long)d.R * c.T; g.b_clen = c.T - t0;
    g.b_seg_len = d.R; g.b_seg_stride = BRT;
    g.C = ws + w.O1 + t0; g.ldc = c.T; g.c_zstride = (long)d.S * c.T;
    if (lo == 0) g.bias = ws + w.bskip;
    else { g.D = ws + w.O1 + t0; g.ldd = c.T; g.d_zstride = (long)d.S * c.T; }  // in place: an element is read by the lane that writes it
    g.relu = last ? 1 : 0; g.nbatch = c.B; g.tag = "fwd_skip_sum";
    return fw_gemm(c, g, nullptr, nullptr, t0);
}

// `side` != nullptr (fused path only): the skip-sum of every `chunk` finished layers is issued on cs->st
static int forward_stack(const Ctx& c, const float* params, const int64_t* x, const float* h, SideRt* side = nullptr,
                         const Ctx* cs = nullptr, int chunk = 0, int* skip_done = nullptr) {
    const WnConfig* cfg = c.cfg;
    const int B = c.B, T = c.T;
    const Dims& d = c.d;
    const Lay& y = c.y;
    const Ws& w = c.w;
    float* ws = c.ws;
    const int F = w.F, Ue = d.U > 0 ? d.U : 1;
    const long BRT = (long)B * d.R * T;

    WN_TRY(pack_weights(c, params));
    if (c.mm_f16 && c.dw_ovf) WN_TRY(wn_fill(c.ws + c.w.dw_ovf, 0.0f, 1, c.st));   // overflow word of the fp16 pair launches of this pass
    // front: one-hot + causal conv as a gather  (wavenet.py:513-516)
    WN_TRY(wn_front_gather(x, ws + w.wc_f, params + y.causal_b, ws + w.X, B, T, d.Q, d.R, d.K, c.st));
    // frame-rate aux projection for all layers at once: G[b][l*2R+o'][f] = Waux_l . h[b][:, f]
    {
        WnGemmArgs g = wn_gemm_default();
        g.M = d.L * 2 * d.R; g.N = F; g.K = d.A;
        g.A = ws + w.waux_f; g.lda = (long)d.L * 2 * d.R;
        g.B = h; g.ldb = F; g.b_zstride = (long)d.A * F; g.b_clen = F;
        g.C = ws + w.G; g.ldc = F; g.c_zstride = (long)d.L * 2 * d.R * F;
        g.nbatch = B; g.tag = "fwd_aux_frames";
        WN_TRY(wn_gemm_launch(&g, c.st));
    }
    const float* upw = d.U > 0 ? params + y.up_w : ws + w.one;
    const long g_bstride = (long)d.L * 2 * d.R * F;
    for (int l = 0; l < d.L; ++l) {
        const int dil = dilation_of(cfg, l);
        const float* Xl = ws + w.X + (long)l * BRT;
        float* Xn = (l + 1 < d.L) ? ws + w.X + (long)(l + 1) * BRT : nullptr;
        const float* Gl = ws + w.G + (long)l * 2 * d.R * F;
        float* Sl = ws + w.Sg + (long)l * BRT;
        float* Gtl = ws + w.Gt + (long)l * BRT;
        float* Zl = ws + w.Z + (long)l * BRT;
        const long lb = layer_base(y, d, l);
        if (c.fused) {
            WN_TRY(wn_fused_resblock_fwd(ws + w.wd_f + (long)l * d.K * d.R * 2 * d.R, ws + w.wres_f + (long)l * d.R * d.R,
                                         ws + w.cvec + (long)l * 2 * d.R, params + lb + y.o_res_b, Xl, Gl, g_bstride, upw, Xn,
                                         Sl, /*tanh half: not saved, backward rebuilds it as z / s*/ nullptr, Zl, B, T,
                                         d.K, dil, Ue, F,
                                         c.fused_f16 ? 2 : (c.split_bf16 ? 1 : 0),
                                         c.fused_f16 ? ws + w.img_fwd16 + (long)l * (w.img16_floats / d.L)
                                                     : ((w.img_floats > 0) ? ws + w.img_fwd + (long)l * (w.img_floats / d.L) : nullptr), c.st));
            if (side && (l + 1) % chunk == 0 && l + 1 < d.L) {
                WN_TRY(side_link(side, c.st, cs->st));  // z of layers [*skip_done, l] is enqueued
                WN_TRY(skip_sum(*cs, *skip_done, l + 1, false));
                *skip_done = l + 1;
            }
        } else {
            // P = sum_tap W_tap . x[t-(K-1-tap)d]            (wavenet.py:527-528)
            WnGemmArgs g = wn_gemm_default();
            g.M = 2 * d.R; g.N = T; g.K = d.K * d.R;
            g.A = ws + w.wd_f + (long)l * d.K * d.R * 2 * d.R; g.lda = 2 * d.R;
            g.B = Xl; g.ldb = T; g.b_zstride = (long)d.R * T; g.b_clen = T;
            g.b_seg_len = d.R; g.b_seg_stride = 0; g.b_shift0 = (d.K - 1) * dil; g.b_shift_step = -dil;
            g.C = ws + w.P; g.ldc = T; g.c_zstride = (long)2 * d.R * T;
            g.nbatch = B; g.tag = "fwd_dilated_layered";
            if (d.R % 128 == 0 && fw_gemm_split_ok(c, g)) {
                // wide models: the gate is the epilogue of the contraction (sigmoid / tanh rows paired by the weight
                // packing), the 2R pre-activations never go to memory                  (wavenet.py:527-532)
                GateEpi ge;
                ge.gate_R = d.R; ge.S = Sl; ge.Gt = Gtl; ge.Z = Zl; ge.G = Gl; ge.g_bstride = g_bstride; ge.F = F; ge.U = Ue;
                ge.upw = upw; ge.cvec = ws + w.cvec + (long)l * 2 * d.R;
                g.tag = "fwd_dilated_gate";
                WN_TRY(fw_gemm(c, g, &ge));
            } else {
                WN_TRY(fw_gemm(c, g));
                // z = sigmoid(.)*tanh(.)                           (wavenet.py:529-532)
                WN_TRY(wn_gate_fwd(ws + w.P, Gl, g_bstride, upw, ws + w.cvec + (long)l * 2 * d.R, Sl, Gtl, Zl, B, T, d.R, Ue, F,
                                   c.st));
            }
            // x_{l+1} = res_1x1(z) + x_l                       (wavenet.py:534-535); dead for the last layer
            if (Xn) {
                WnGemmArgs r = wn_gemm_default();
                r.M = d.R; r.N = T; r.K = d.R;
                r.A = ws + w.wres_f + (long)l * d.R * d.R; r.lda = d.R;
                r.B = Zl; r.ldb = T; r.b_zstride = (long)d.R * T; r.b_clen = T;
                r.C = Xn; r.ldc = T; r.c_zstride = (long)d.R * T;
                r.bias = params + lb + y.o_res_b;
                r.D = Xl; r.ldd = T; r.d_zstride = (long)d.R * T;
                r.nbatch = B; r.tag = "fwd_res_layered";
                WN_TRY(fw_gemm(c, r));
            }
        }
    }
    return 0;
}

// wn_forward, optionally with the softmax cross-entropy as the epilogue of conv_post_2 (`ce`: the logits are not written)
static int forward_impl(const WnConfig* cfg, int B, int T, const float* params, const int64_t* x, const float* h,
                        float* logits, const CeEpi* ce_in, void* wsp, size_t ws_bytes, int flags, void* stream, const char* who) {
    Ctx c;
    WN_TRY(make_ctx(&c, cfg, B, T, wsp, ws_bytes, flags, stream));
    if (!params || !x || !h || (!logits && !ce_in)) return fail(1, "NULL argument");
    c.params = params;
    // overlap mode (opt-in, fused kernels): partial skip-sums run on the internal side stream beside the stack
    SideLock side((flags & WN_FLAG_FWD_OVERLAP) && c.fused && !wn_prof_is_on(), c.st);
    Ctx cs = c;
    int skip_done = 0;
#ifndef WN_EMU
    if (side.rt) cs.st = side.rt->st;
#endif
    const Dims& d = c.d;
    const Lay& y = c.y;
    const Ws& w = c.w;
    float* ws = c.ws;
    // Loss window (training step only: `ce_in` given, the logits themselves are not an output).  The loss of train.py:534-536
    // covers [:, receptive_field:], and everything between the skip sum and the loss is pointwise in time
    // (wavenet.py:518-523,533): the skip-sum / post-net contractions of the step run over the columns [t0, T) only, t0 = the
    // first loss position rounded down to a whole 128-column tile -- the same window wn_backward_window takes, so nothing in
    // front of it is ever read.  The residual stack itself needs every position.  dlogits[.., t < t0] is zero-filled.
    const int t0 = ce_in ? (ce_in->t_start / 128) * 128 : 0;
    const int Tw = T - t0;
    // The columns in front of the window are not written by this call.  A later backward pass whose window starts further
    // left (wn_backward = t_first 0) contracts them with dlogits == 0: any FINITE value there contributes exactly nothing,
    // uninitialised memory (NaN / Inf bit patterns) would not.  So they are zero-filled unless the caller vouches for the
    // workspace (WN_FLAG_WS_FINITE: allocated zero-filled, or written by an earlier full forward).
    if (t0 > 0 && !(flags & WN_FLAG_WS_FINITE)) {
        WN_TRY(wn_fill_cols(c.ws + c.w.O1, (long)B * c.d.S, T, t0, c.st));
        WN_TRY(wn_fill_cols(c.ws + c.w.O2, (long)B * c.d.S, T, t0, c.st));
    }
    WN_TRY(forward_stack(c, params, x, h, side.rt, &cs, (d.L + 2) / 3, &skip_done));
    WN_TRY(side_link(side.rt, cs.st, c.st));  // join: O1 holds the sum of layers [0, skip_done)
    // skip-sum over (the remaining) layers as ONE contraction with K = L*R (wavenet.py:533,238), relu fused (:519)
    WN_TRY(skip_sum(c, skip_done, d.L, true, t0));
    {   // conv_post_1 + relu  (wavenet.py:520-521)
        WnGemmArgs g = wn_gemm_default();
        g.M = d.S; g.N = Tw; g.K = d.S;
        g.A = ws + w.w1_f; g.lda = d.S;
        g.B = ws + w.O1 + t0; g.ldb = T; g.b_zstride = (long)d.S * T; g.b_clen = Tw;
        g.C = ws + w.O2 + t0; g.ldc = T; g.c_zstride = (long)d.S * T;
        g.bias = params + y.post1_b; g.relu = 1; g.nbatch = B; g.tag = "fwd_post1";
        WN_TRY(fw_gemm(c, g, nullptr, nullptr, t0));
    }
    {   // conv_post_2  (wavenet.py:522)
        WnGemmArgs g = wn_gemm_default();
        g.M = d.Qo; g.N = Tw; g.K = d.S;
        g.A = ws + w.w2_f; g.lda = d.Qo;
        g.B = ws + w.O2 + t0; g.ldb = T; g.b_zstride = (long)d.S * T; g.b_clen = Tw;
        g.C = logits ? logits + t0 : nullptr; g.ldc = T; g.c_zstride = (long)d.Qo * T;
        g.bias = params + y.post2_b; g.nbatch = B; g.tag = "fwd_post2";
        if (ce_in) {
            CeEpi ce = *ce_in;
            ce.target = ce_in->target + t0;     // column j of the window is position t0 + j (row stride T)
            ce.t_start = ce_in->t_start - t0;
            ce.partial = ws + w.loss_partial;
            ce.amax = ws + w.amax_partial;
            g.tag = "fwd_post2_ce";
            WN_TRY(fw_gemm(c, g, nullptr, &ce, t0));   // g.C = the caller's dlogits (or NULL)
            if (logits && t0 > 0) WN_TRY(wn_fill_cols(logits, (long)B * d.Qo, T, t0, c.st));
        } else {
            WN_TRY(fw_gemm(c, g, nullptr, nullptr, t0));
        }
    }
    return rt_check(who);
}

extern "C" int wn_forward(const WnConfig* cfg, int B, int T, const float* params, const int64_t* x, const float* h,
                          float* logits, void* wsp, size_t ws_bytes, int flags, void* stream) {
    api_enter();
    return forward_impl(cfg, B, T, params, x, h, logits, nullptr, wsp, ws_bytes, flags, stream, "wn_forward");
}

// 1: wn_forward_loss runs the loss as the epilogue of conv_post_2 for this model / flags (softmax head with at most 256
// classes on the split contractions); 0: it needs the logits scratch buffer and runs wn_forward + wn_softmax_ce_loss.
extern "C" int wn_forward_loss_fused(const WnConfig* cfg, int B, int T, int flags) {
    api_enter();
    Dims d;
    Ws w;
    if (check_cfg(cfg, &d) || B < 1 || T < 1 || make_ws(d, B, T, &w)) return 0;
    const bool split = !(flags & WN_FLAG_EXACT_MFMA);
    // the conditions under which fw_gemm takes the split contraction for conv_post_2, plus: every class in one 256-row block
    return (split && d.Qo == d.Q && d.Qo >= 128 && d.Qo <= WN_G6_BM &&
            wn_gemm6_apk_elems(d.Qo, d.S) <= 2 * w.apk_floats && (long)d.Qo * T * 4 < 0x7ffffff0L) ? 1 : 0;
}

extern "C" int wn_forward_loss(const WnConfig* cfg, int B, int T, const float* params, const int64_t* x, const float* h,
                               const int64_t* target, int t_start, float grad_scale, float loss_scale, float* loss,
                               float* dlogits, float* logits_scratch, void* wsp, size_t ws_bytes, int flags, void* stream) {
    api_enter();
    if (!target || !loss) return fail(1, "NULL argument");
    if (t_start < 0 || t_start >= T) return fail(1, "t_start=%d outside [0,%d)", t_start, T);
    if (!wn_forward_loss_fused(cfg, B, T, flags)) {
        if (!logits_scratch) return fail(1, "this model / flag set needs the logits scratch buffer (wn_forward_loss_fused() == 0)");
        WN_TRY(forward_impl(cfg, B, T, params, x, h, logits_scratch, nullptr, wsp, ws_bytes, flags, stream, "wn_forward_loss"));
        return wn_softmax_ce_loss(cfg, B, T, logits_scratch, target, t_start, grad_scale, loss_scale, loss, dlogits, wsp, ws_bytes,
                                  stream);
    }
    CeEpi ce;
    ce.target = target; ce.t_start = t_start; ce.gs = grad_scale / ((float)B * (float)(T - t_start)); ce.partial = nullptr; ce.amax = nullptr;
    WN_TRY(forward_impl(cfg, B, T, params, x, h, dlogits, &ce, wsp, ws_bytes, flags, stream, "wn_forward_loss"));
    Ctx c;
    WN_TRY(make_ctx(&c, cfg, B, T, wsp, ws_bytes, flags, stream));
    const int t0w = (t_start / 128) * 128;   // the column window forward_impl ran the loss epilogue over
    const int np = ((T - t0w + WN_G6_BN - 1) / WN_G6_BN) * B;
    WN_TRY(wn_sum_partials(c.ws + c.w.loss_partial, np, loss_scale / ((float)B * (float)(T - t_start)), loss,
                           dlogits ? c.ws + c.w.amax_partial : nullptr, c.ws + c.w.dw_ovf + 2, c.st));   // + max |dlogits| (WN_FLAG_DW_F16_AMAX_WS)
    return rt_check("wn_forward_loss");
}

// ------------------------------------------------------------------------------------------
// loss
// ------------------------------------------------------------------------------------------
extern "C" int wn_softmax_ce_loss(const WnConfig* cfg, int B, int T, const float* logits, const int64_t* target, int t_start,
                                  float grad_scale, float loss_scale, float* loss, float* dlogits, void* wsp, size_t ws_bytes,
                                  void* stream) {
    api_enter();
    Ctx c;
    WN_TRY(make_ctx(&c, cfg, B, T, wsp, ws_bytes, 0, stream));
    if (!logits || !target || !loss) return fail(1, "NULL argument");
    if (t_start < 0 || t_start >= T) return fail(1, "t_start=%d outside [0,%d)", t_start, T);
    int np = 0;
    const float gs = grad_scale / ((float)B * (float)(T - t_start));
    WN_TRY(wn_softmax_ce(logits, target, dlogits, c.ws + c.w.loss_partial, &np, B, T, c.d.Qo, t_start, gs,
                         dlogits ? c.ws + c.w.amax_partial : nullptr, c.st));
    WN_TRY(wn_sum_partials(c.ws + c.w.loss_partial, np, loss_scale / ((float)B * (float)(T - t_start)), loss,
                           dlogits ? c.ws + c.w.amax_partial : nullptr, c.ws + c.w.dw_ovf + 2, c.st));   // + max |dlogits| (WN_FLAG_DW_F16_AMAX_WS)
    return rt_check("wn_softmax_ce_loss");
}

// Mixture-of-logistics head (BASELINE configs[3]; absent from the reference): mean negative log-likelihood of
// the target waveform y (B,T) in [-1,1] under the 3*n_mix output channels, over positions t >= t_start,
// and its gradient in the layout wn_backward takes.
extern "C" int wn_mol_loss(const WnConfig* cfg, int B, int T, const float* out, const float* y, int t_start, float grad_scale,
                           float loss_scale, int num_classes, float log_scale_min, float* loss, float* dout, void* wsp,
                           size_t ws_bytes, void* stream) {
    api_enter();
    Ctx c;
    WN_TRY(make_ctx(&c, cfg, B, T, wsp, ws_bytes, 0, stream));
    if (!out || !y || !loss) return fail(1, "NULL argument");
    if (c.d.Qo % 3 != 0) return fail(1, "out_channels=%d is not 3 * n_mixture", c.d.Qo);
    if (t_start < 0 || t_start >= T) return fail(1, "t_start=%d outside [0,%d)", t_start, T);
    int np = 0;
    const float gs = grad_scale / ((float)B * (float)(T - t_start));
    WN_TRY(wn_mol_nll(out, y, dout, c.ws + c.w.loss_partial, &np, B, T, c.d.Qo / 3, t_start, gs, num_classes, log_scale_min, c.st));
    // (no measured maximum for this head: a backward call with WN_FLAG_DW_F16PAIR scans the gradient it is given)
    WN_TRY(wn_sum_partials(c.ws + c.w.loss_partial, np, loss_scale / ((float)B * (float)(T - t_start)), loss, nullptr, c.ws + c.w.dw_ovf + 2, c.st));
    return rt_check("wn_mol_loss");
}

// ------------------------------------------------------------------------------------------
// The rest of the C ABI, by strand (same translation unit: they use the helpers above)
// ------------------------------------------------------------------------------------------
#include "wn_api_backward.inl"
#include "wn_api_ops.inl"
#include "wn_api_decode.inl"
