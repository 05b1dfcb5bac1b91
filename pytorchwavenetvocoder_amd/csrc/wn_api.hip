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
    if (bucket == 0) {
        a = 0;
        b = y.layers0;
    } else if (bucket <= ngroups) {
        const int g = bucket - 1;  // layers processed: L-1-g*lpb ... down
        const int first = g * lpb;
        int last = first + lpb;
        if (last > d.L) last = d.L;
        a = y.layers0 + (long)first * y.LB;
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
    long dGp, qp;  // aux-gradient partials of the gate kernel (WN_FLAG_AUX_FUSED); 0 floats when the mode cannot apply
    long img_fwd, img_taps, img_res, img_floats;  // pre-split LDS weight images of the fused split kernels (0 floats: not applicable)
    long wskipT_f, dZs;  // chain mode (wn_fused_chain_supported): skip weights as [s][l*R + i], dZs = Wskip^T dSkip (B, L*R, T)
    long dZs_floats;
    long red_scratch_floats;
    long apk_floats;
    long apk_pre[6];   // the six weight sets of a training step, split ONCE per step (pack_weights): offsets, -1 = none
    long apk_wide[5];  // wide models (n_resch % 128 == 0, any-size path): the five per-layer weight sets of a step, all layers, split
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
    CARVE(partial, pmax);
    CARVE(rs_partial, rmax);
    w->red_scratch_floats = 1 << 20;
    CARVE(red_scratch, w->red_scratch_floats);
    CARVE(loss_partial, 2 * wn_softmax_ce_nblocks(B, T) + 64);   // CE epilogue: one partial per 128-column block
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
struct PreJob { const float* A; long lda; int M, K; long off; };
static int pre_jobs(const Ctx& c, const float* params, PreJob (&j)[6]) {
    const Dims& d = c.d;
    const Lay& y = c.y;
    const Ws& w = c.w;
    float* ws = c.ws;
    const PreJob all[6] = {{ws + w.wskip_f, d.S, d.S, d.L * d.R, w.apk_pre[0]},
                           {ws + w.w1_f, d.S, d.S, d.S, w.apk_pre[1]},
                           {ws + w.w2_f, d.Qo, d.Qo, d.S, w.apk_pre[2]},
                           {params ? params + y.post2_w : nullptr, d.S, d.S, d.Qo, w.apk_pre[3]},
                           {params ? params + y.post1_w : nullptr, d.S, d.S, d.S, w.apk_pre[4]},
                           {ws + w.wskipT_f, (long)d.L * d.R, dzs_layers(d) * d.R, d.S, w.apk_pre[5]}};
    int n = 0;
    if (!c.have_pre || !c.split_bf16) return 0;
    for (int i = 0; i < 6; ++i)
        if (all[i].A && all[i].off >= 0) j[n++] = all[i];
    return n;
}
// The per-layer weight sets of a wide model's step (any-size path with the split contractions, n_resch % 128 == 0): layer l of
// set i lives at A + l * lstride (floats) and its split form at apk_wide[i] + l * apk_wide_l[i].  Set 0 is packed with the
// gate row permutation (gate_R = R: the forward gate epilogue), the others plainly.
struct WideJob { const float* A; long lstride, lda; int M, K, gate_R, nl; long off, off_l; };
static int wide_jobs(const Ctx& c, const float* params, WideJob (&j)[5]) {
    const Dims& d = c.d;
    const Lay& y = c.y;
    const Ws& w = c.w;
    if (!c.have_pre || !c.split_bf16 || c.fused || w.apk_wide[0] < 0 || !params) return 0;
    const long lb0 = layer_base(y, d, 0);
    const WideJob all[5] = {
        {c.ws + w.wd_f, (long)d.K * d.R * 2 * d.R, 2 * d.R, 2 * d.R, d.K * d.R, d.R, d.L, w.apk_wide[0], w.apk_wide_l[0]},   // fwd_dilated_gate
        {c.ws + w.wres_f, (long)d.R * d.R, d.R, d.R, d.R, 0, d.L, w.apk_wide[1], w.apk_wide_l[1]},                          // fwd_res
        {params + y.skip0, y.ls_skip, d.R, d.R, d.S, 0, d.L, w.apk_wide[2], w.apk_wide_l[2]},                               // bwd_dz_skip
        {params + lb0 + y.o_res_w, -y.LB, d.R, d.R, d.R, 0, d.L, w.apk_wide[3], w.apk_wide_l[3]},                           // bwd_dz_res
        {c.ws + w.wd_b, (long)d.K * 2 * d.R * d.R, d.R, d.R, d.K * 2 * d.R, 0, d.L, w.apk_wide[4], w.apk_wide_l[4]}};      // bwd_dx_dilated
    for (int i = 0; i < 5; ++i) j[i] = all[i];
    return 5;
}
static long prepacked_offset(const Ctx& c, const WnGemmArgs& g, int gate_R = 0) {
    if (gate_R == 0) {
        PreJob j[6];
        const int n = pre_jobs(c, c.params, j);
        for (int i = 0; i < n; ++i)
            if (j[i].A == g.A && j[i].lda == g.lda && j[i].M == g.M && j[i].K == g.K) return j[i].off;
    }
    WideJob wj[5];
    const int nw = wide_jobs(c, c.params, wj);
    for (int i = 0; i < nw; ++i) {
        if (wj[i].lda != g.lda || wj[i].M != g.M || wj[i].K != g.K || wj[i].gate_R != gate_R || wj[i].lstride == 0) continue;
        const long diff = g.A - wj[i].A;
        if (diff % wj[i].lstride != 0) continue;
        const long l = diff / wj[i].lstride;
        if (l >= 0 && l < wj[i].nl) return wj[i].off + l * wj[i].off_l;
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
};
// n_origin: index of column 0 of this launch in the caller's full (B, C, T) tensor (a loss-window launch starts at t0): the
// alternating tile signs of k_gemm6 follow the ABSOLUTE column, so a windowed launch produces bit for bit what the full one
// produces on those columns (same ReLU masks in the training step's and the module's forward).
static int fw_gemm(const Ctx& c, const WnGemmArgs& g, const GateEpi* ge = nullptr, const CeEpi* ce = nullptr, int n_origin = 0) {
    const bool ok = c.split_bf16 && g.M >= 128 && !g.a_kmajor && !g.b_kmajor &&
                    (g.b_seg_len >= g.K || g.b_seg_len % 16 == 0) && g.ksplit == 1 && g.nlayer == 1 && !g.b_relu &&
                    !g.b_index && g.a_zstride == 0 && !g.a_rowsum &&
                    wn_gemm6_apk_elems(g.M, g.K) <= 2 * c.w.apk_floats && (long)g.M * g.ldc * 4 < 0x7ffffff0L;
    if (!ok) return (ge || ce) ? fail(3, "gate / loss epilogue needs the split contraction") : wn_gemm_launch(&g, c.st);
    unsigned short* apk = reinterpret_cast<unsigned short*>(c.ws + c.w.apk);
    const long pre = prepacked_offset(c, g, ge ? ge->gate_R : 0);   // (gate' epilogues use the plain packing: gate_R = 0)
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
        a.ce_gs = ce->gs; a.ce_partial = ce->partial;
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
            for (int i = 0; i < nw; ++i) {   // every layer of a wide model's five per-layer sets: one launch instead of 5 L - 1
                const int q = n + i;
                jobs.src[q] = wj[i].A; jobs.lda[q] = wj[i].lda; jobs.M[q] = wj[i].M; jobs.K[q] = wj[i].K;
                jobs.dst[q] = reinterpret_cast<unsigned short*>(ws + wj[i].off);
                jobs.nl[q] = wj[i].nl; jobs.src_lstride[q] = wj[i].lstride; jobs.dst_lstride[q] = 2 * wj[i].off_l;
                jobs.gate_R[q] = wj[i].gate_R;
            }
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
                                         c.split_bf16 ? 1 : 0,
                                         (w.img_floats > 0) ? ws + w.img_fwd + (long)l * (w.img_floats / d.L) : nullptr, c.st));
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
    ce.target = target; ce.t_start = t_start; ce.gs = grad_scale / ((float)B * (float)(T - t_start)); ce.partial = nullptr;
    WN_TRY(forward_impl(cfg, B, T, params, x, h, dlogits, &ce, wsp, ws_bytes, flags, stream, "wn_forward_loss"));
    Ctx c;
    WN_TRY(make_ctx(&c, cfg, B, T, wsp, ws_bytes, flags, stream));
    const int t0w = (t_start / 128) * 128;   // the column window forward_impl ran the loss epilogue over
    const int np = ((T - t0w + WN_G6_BN - 1) / WN_G6_BN) * B;
    WN_TRY(wn_sum_partials(c.ws + c.w.loss_partial, np, loss_scale / ((float)B * (float)(T - t_start)), loss, c.st));
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
    WN_TRY(wn_softmax_ce(logits, target, dlogits, c.ws + c.w.loss_partial, &np, B, T, c.d.Qo, t_start, gs, c.st));
    WN_TRY(wn_sum_partials(c.ws + c.w.loss_partial, np, loss_scale / ((float)B * (float)(T - t_start)), loss, c.st));
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
    WN_TRY(wn_sum_partials(c.ws + c.w.loss_partial, np, loss_scale / ((float)B * (float)(T - t_start)), loss, c.st));
    return rt_check("wn_mol_loss");
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
struct DwOut {          // destination mapping of a weight-gradient (see WnReduceArgs)
    float* out;
    int m_seg, n_seg;
    long m_seg_stride, m_stride, n_seg_stride, n_stride;
    const float* addend_m;
    const float* addend_scale_ptr;
    float* rowsum_out;  // nullable: [M] contiguous destination of sum_k A(m,k)
    long out_lstride, addend_lstride, rowsum_lstride;  // per layer of a batched launch
};

// dW[l][m][n] = sum_{b, k} A_{l,b}(m,k) * B_{l,b}(n,k)   (k = time) for nl layers in ONE launch,
// split over (layer, b, k-chunks) and reduced in a fixed order.
// fixed-order sum of `nz` partial [M][N] matrices per layer (and of the [M] row-sum partials) into their mapped destinations
static int dw_reduce(const Ctx& c, const float* partial, const float* rs_partial, int nz, int M, int N, const DwOut& o, int nl) {
    WnReduceArgs r;
    r.partial = partial; r.nz = nz; r.M = M; r.N = N;
    r.out = o.out; r.m_seg = o.m_seg; r.n_seg = o.n_seg;
    r.m_seg_stride = o.m_seg_stride; r.m_stride = o.m_stride; r.n_seg_stride = o.n_seg_stride; r.n_stride = o.n_stride;
    r.scale = 1.0f; r.accumulate = 0; r.addend_m = o.addend_m; r.addend_scale_ptr = o.addend_scale_ptr;
    r.scratch = c.ws + c.w.red_scratch; r.scratch_floats = c.w.red_scratch_floats;
    r.nl = nl; r.out_lstride = o.out_lstride; r.addend_lstride = o.addend_lstride;
    WN_TRY(wn_reduce(&r, c.st));
    if (o.rowsum_out) {
        WnReduceArgs q;
        q.partial = rs_partial; q.nz = nz; q.M = M; q.N = 1;
        q.out = o.rowsum_out; q.m_seg = 0x7fffffff; q.n_seg = 0x7fffffff;
        q.m_seg_stride = 0; q.m_stride = 1; q.n_seg_stride = 0; q.n_stride = 0;
        q.scale = 1.0f; q.accumulate = 0; q.addend_m = nullptr; q.addend_scale_ptr = nullptr;
        q.scratch = c.ws + c.w.red_scratch; q.scratch_floats = c.w.red_scratch_floats;
        q.nl = nl; q.out_lstride = o.rowsum_lstride; q.addend_lstride = 0;
        WN_TRY(wn_reduce(&q, c.st));
    }
    return 0;
}

static int dw_gemm(const Ctx& c, WnGemmArgs g, const DwOut& o, int nl = 1) {
    const DwPlan p = dw_plan(g.M, g.N, g.K, c.B * nl);
    const int nz_layer = p.ksplit * c.B;
    g.a_kmajor = 1; g.b_kmajor = 1;
    g.nlayer = nl; g.nbatch = c.B; g.ksplit = p.ksplit; g.kchunk = p.kchunk;
    g.C = c.ws + c.w.partial; g.ldc = g.N; g.c_zstride = (long)g.M * g.N;
    g.a_rowsum = o.rowsum_out ? c.ws + c.w.rs_partial : nullptr;
    if (c.split_bf16 && wn_gemm6_dw_eligible(&g))
        WN_TRY(wn_gemm6_dw_launch(&g, c.dw_products, c.st));
    else
        WN_TRY(wn_gemm_launch(&g, c.st));
    return dw_reduce(c, c.ws + c.w.partial, c.ws + c.w.rs_partial, nz_layer, g.M, g.N, o, nl);
}

static DwOut dw_out_plain(float* out, long ld, float* rowsum_out) {
    DwOut o;
    o.out = out; o.m_seg = 0x7fffffff; o.n_seg = 0x7fffffff;
    o.m_seg_stride = 0; o.m_stride = ld; o.n_seg_stride = 0; o.n_stride = 1;
    o.addend_m = nullptr; o.addend_scale_ptr = nullptr; o.rowsum_out = rowsum_out;
    o.out_lstride = 0; o.addend_lstride = 0; o.rowsum_lstride = 0;
    return o;
}

extern "C" int wn_backward(const WnConfig* cfg, int B, int T, const float* params, const int64_t* x, const float* h,
                           const float* dlogits, float* grads, void* wsp, size_t ws_bytes, void* const* events, int n_events,
                           int lpb, int flags, void* stream) {
    return wn_backward_window(cfg, B, T, params, x, h, dlogits, 0, grads, wsp, ws_bytes, events, n_events, lpb, flags, stream);
}

extern "C" int wn_backward_window(const WnConfig* cfg, int B, int T, const float* params, const int64_t* x, const float* h,
                                  const float* dlogits, int t_first, float* grads, void* wsp, size_t ws_bytes,
                                  void* const* events, int n_events, int lpb, int flags, void* stream) {
    api_enter();
    Ctx c;
    WN_TRY(make_ctx(&c, cfg, B, T, wsp, ws_bytes, flags, stream));
    if (!params || !x || !h || !dlogits || !grads) return fail(1, "NULL argument");
    c.params = params;
    if (t_first < 0 || t_first >= T) return fail(1, "t_first=%d outside [0,%d)", t_first, T);
    // WN_FLAG_REPACK: `params` changed since the forward call (or the caller cannot tell): rebuild every re-laid-out /
    // pre-split weight set of the workspace from the buffer given HERE, so that the backward contractions use one
    // consistent set of weights (the saved activations are the forward pass's own either way).
    if (flags & WN_FLAG_REPACK) WN_TRY(pack_weights(c, params));
    // Loss window.  The loss of train.py:534-536 covers [:, receptive_field:], so dlogits is exactly zero in front of it, and
    // everything between the logits and the residual stack is pointwise in time: dO2, dSkip and the skip part of every
    // layer's dZ are zero there too, and those columns contribute nothing to the post-net / skip weight gradients.  The
    // contractions of this part run over [t0, T) only (t0 = t_first rounded down to a whole 128-column tile, so that every
    // row keeps its alignment); dSkip is zero-filled in front of t0 and the chain kernel takes dZs as zero there: the chain
    // itself needs every position (dX_l[t] depends on dP_l[t + dilation]).  13 % less matrix work in these launches at the benchmark's geometry.
    const int t0 = (t_first / 128) * 128;
    const int Tw = T - t0;
    // c = the data chain on the caller's stream; cs = the weight gradients, on the side stream unless serial
    SideLock side((flags & WN_FLAG_BWD_OVERLAP) && !wn_prof_is_on(), c.st);
    Ctx cs = c;
#ifndef WN_EMU
    if (side.rt) cs.st = side.rt->st;
#endif
    const Dims& d = c.d;
    const Lay& y = c.y;
    const Ws& w = c.w;
    float* ws = c.ws;
    const int F = w.F, Ue = d.U > 0 ? d.U : 1;
    const long BRT = (long)B * d.R * T;
    if (lpb < 1) lpb = d.L;
    const int nb = wn_num_buckets(cfg, lpb);
    if (events && n_events < nb) return fail(1, "need %d bucket events, got %d", nb, n_events);
    int bucket = 0;

    // ---- post-net backward (wavenet.py:518-523 reversed) ----
    {   // dO2 = W2^T dlogits, masked by relu'(O2)
        WnGemmArgs g = wn_gemm_default();
        g.M = d.S; g.N = Tw; g.K = d.Qo;
        g.A = params + y.post2_w; g.lda = d.S;
        g.B = dlogits + t0; g.ldb = T; g.b_zstride = (long)d.Qo * T; g.b_clen = Tw;
        g.C = ws + w.dO2 + t0; g.ldc = T; g.c_zstride = (long)d.S * T;
        g.E = ws + w.O2 + t0; g.lde = T; g.e_zstride = (long)d.S * T;
        g.nbatch = B; g.tag = "bwd_post2_dx";
        WN_TRY(fw_gemm(c, g, nullptr, nullptr, t0));
    }
    {   // dSkip = W1^T dO2, masked by relu'(skip-sum)
        WnGemmArgs g = wn_gemm_default();
        g.M = d.S; g.N = Tw; g.K = d.S;
        g.A = params + y.post1_w; g.lda = d.S;
        g.B = ws + w.dO2 + t0; g.ldb = T; g.b_zstride = (long)d.S * T; g.b_clen = Tw;
        g.C = ws + w.dSk + t0; g.ldc = T; g.c_zstride = (long)d.S * T;
        g.E = ws + w.O1 + t0; g.lde = T; g.e_zstride = (long)d.S * T;
        g.nbatch = B; g.tag = "bwd_post1_dx";
        WN_TRY(fw_gemm(c, g, nullptr, nullptr, t0));
        if (t0 > 0) WN_TRY(wn_fill_cols(ws + w.dSk, (long)B * d.S, T, t0, c.st));
    }
    WN_TRY(side_link(side.rt, c.st, cs.st));  // fork: dO2, dSkip (and everything before this call) are ready
    {   // d conv_post_2.{weight,bias}
        WnGemmArgs g = wn_gemm_default();
        g.M = d.Qo; g.N = d.S; g.K = Tw;
        g.A = dlogits + t0; g.lda = T; g.a_zstride = (long)d.Qo * T;
        g.B = ws + w.O2 + t0; g.ldb = T; g.b_zstride = (long)d.S * T; g.b_clen = Tw; g.tag = "dw_post2";
        WN_TRY(dw_gemm(cs, g, dw_out_plain(grads + y.post2_w, d.S, grads + y.post2_b)));
    }
    {   // d conv_post_1.{weight,bias}
        WnGemmArgs g = wn_gemm_default();
        g.M = d.S; g.N = d.S; g.K = Tw;
        g.A = ws + w.dO2 + t0; g.lda = T; g.a_zstride = (long)d.S * T;
        g.B = ws + w.O1 + t0; g.ldb = T; g.b_zstride = (long)d.S * T; g.b_clen = Tw; g.tag = "dw_post1";
        WN_TRY(dw_gemm(cs, g, dw_out_plain(grads + y.post1_w, d.S, grads + y.post1_b)));
    }
    {   // d skip_1x1.l.weight for all layers in one contraction; bias = rowsum(dSkip) for every layer
        WnGemmArgs g = wn_gemm_default();
        g.M = d.S; g.N = d.L * d.R; g.K = Tw;
        g.A = ws + w.dSk + t0; g.lda = T; g.a_zstride = (long)d.S * T;
        g.B = ws + w.Z + t0; g.ldb = T; g.b_zstride = (long)d.R * T; g.b_clen = Tw;
        g.b_seg_len = d.R; g.b_seg_stride = BRT; g.tag = "dw_skip";
        DwOut o;
        o.out = grads + y.skip0; o.m_seg = 0x7fffffff; o.m_seg_stride = 0; o.m_stride = d.R;
        o.n_seg = d.R; o.n_seg_stride = y.ls_skip; o.n_stride = 1;
        o.addend_m = nullptr; o.addend_scale_ptr = nullptr; o.rowsum_out = ws + w.tmpS;
        o.out_lstride = 0; o.addend_lstride = 0; o.rowsum_lstride = 0;
        WN_TRY(dw_gemm(cs, g, o));
        WnCopy4 cp;
        cp.n0 = 1; cp.n1 = 1; cp.n2 = d.S; cp.nl = d.L;
        cp.s0 = 0; cp.s1 = 0; cp.s2 = 1; cp.sl = 0;
        cp.d0 = 0; cp.d1 = 0; cp.d2 = 1; cp.dl = y.ls_skip;
        WN_TRY(wn_copy4(grads + y.skip0 + (long)d.S * d.R, ws + w.tmpS, &cp, cs.st));
    }
    if (events) rt_event_record(events[bucket], cs.st);
    bucket++;

    // ---- residual stack, last layer first (wavenet.py:525-536 reversed) ----
    // The data chain (gate', dX) runs layer by layer; dP_l and dX_l of every layer are kept so that
    // the weight gradients of a whole bucket of layers are produced by ONE launch per tensor kind
    // (layer = outermost z dimension of the dW contraction), then reduced in a fixed order.
    const float* upw = d.U > 0 ? params + y.up_w : ws + w.one;
    const long g_bstride = (long)d.L * 2 * d.R * F;
    const long P_L = 2 * BRT;
    // WN_FLAG_AUX_FUSED: the gate kernel leaves the partial sums of the aux-path gradients behind, dP is not re-read
    // for them (split kernels, upsampling layer with U % 16 == 0)
    const bool aux_fused = (flags & WN_FLAG_AUX_FUSED) && c.fused && c.split_bf16 && d.U >= 16 && d.U % 16 == 0 && w.dGp != w.qp;
    // Chain mode (default for the fused split kernels, kernel_size <= 2): one launch per layer computes dX_l AND, from it,
    // dP_{l-1}; the skip part of every layer's dZ is contracted up front, dZs[b][l*R + i][t] = sum_s Wskip_l[s][i] dSkip[b][s][t]
    // (layers 0 .. L-2; the last layer's gate' takes dSkip itself, it has no dX input).  WN_FLAG_NO_CHAIN: the former pair.
    const bool chain = c.fused && c.split_bf16 && !(flags & WN_FLAG_NO_CHAIN) && w.dZs_floats > 0 &&
                       wn_fused_chain_supported(d.R, d.K, d.S);
    const long zs_bstride = (long)d.L * d.R * T;
    if (chain) {
        WnGemmArgs g = wn_gemm_default();
        g.M = dzs_layers(d) * d.R; g.N = Tw; g.K = d.S;
        g.A = ws + w.wskipT_f; g.lda = (long)d.L * d.R;
        g.B = ws + w.dSk + t0; g.ldb = T; g.b_zstride = (long)d.S * T; g.b_clen = Tw;
        g.C = ws + w.dZs + t0; g.ldc = T; g.c_zstride = zs_bstride;
        g.nbatch = B; g.tag = "bwd_dz_skip_all";
        WN_TRY(fw_gemm(c, g, nullptr, nullptr, t0));
        // dZs[.., t < t0] stays unwritten: the chain kernel takes it as zero without reading it (ChainArgs.zs_t0)
    }
    // WN_FLAG_BWD_OVERLAP_HEAD: only the post-net / skip weight gradients (matrix-bound) go to the side stream, the
    // per-layer groups (HBM-bound like the chain itself) follow the chain on the caller's stream
    const Ctx& cl = (flags & WN_FLAG_BWD_OVERLAP_HEAD) ? c : cs;
    auto flush_bucket = [&](int lo, int hi) -> int {
        const Ctx& c = cl;  // every launch of a flush is a weight gradient
        const int nl = hi - lo;
        const long lb_lo = layer_base(y, d, lo);
        float* dc = ws + w.dc + (long)lo * 2 * d.R;
        {   // d dil_{sigmoid,tanh}.l.conv.weight ; dc_l = rowsum(dP_l) -> conv + aux biases
            WnGemmArgs g = wn_gemm_default();
            g.M = 2 * d.R; g.N = d.K * d.R; g.K = T;
            g.A = ws + w.P + (long)lo * P_L; g.lda = T; g.a_zstride = (long)2 * d.R * T; g.a_lstride = P_L;
            g.B = ws + w.X + (long)lo * BRT; g.ldb = T; g.b_zstride = (long)d.R * T; g.b_lstride = BRT; g.b_clen = T;
            g.b_seg_len = d.R; g.b_seg_stride = 0; g.b_shift0 = d.K - 1; g.b_shift_step = -1;
            g.b_dil_depth = cfg->dilation_depth; g.b_layer0 = lo;
            g.tag = "dw_dilated";
            DwOut o;
            o.out = grads + lb_lo + y.o_dsig_w;
            o.m_seg = d.R; o.m_seg_stride = y.o_dtanh_w - y.o_dsig_w; o.m_stride = (long)d.R * d.K;
            o.n_seg = d.R; o.n_seg_stride = 1; o.n_stride = d.K;
            o.addend_m = nullptr; o.addend_scale_ptr = nullptr; o.rowsum_out = dc;
            o.out_lstride = -y.LB; o.addend_lstride = 0; o.rowsum_lstride = 2 * d.R;
            WN_TRY(dw_gemm(c, g, o, nl));
            WnCopy4 cp;  // biases: dil_{sig,tanh}.bias = dc ; aux_{sig,tanh}.bias = dc
            cp.n0 = 1; cp.n1 = 2; cp.n2 = d.R; cp.nl = nl;
            cp.s0 = 0; cp.s1 = d.R; cp.s2 = 1; cp.sl = 2 * d.R;
            cp.d0 = 0; cp.d1 = y.o_dtanh_b - y.o_dsig_b; cp.d2 = 1; cp.dl = -y.LB;
            WN_TRY(wn_copy4(grads + lb_lo + y.o_dsig_b, dc, &cp, c.st));
            cp.d1 = y.o_atanh_b - y.o_asig_b;
            WN_TRY(wn_copy4(grads + lb_lo + y.o_asig_b, dc, &cp, c.st));
        }
        {   // d res_1x1.l = dX_{l+1} . z_l^T ; the last layer's res_1x1 is dead -> zeros
            const int hi_res = hi < d.L ? hi : d.L - 1;
            if (hi == d.L) WN_TRY(wn_fill(grads + layer_base(y, d, d.L - 1) + y.o_res_w, 0.0f, (long)d.R * d.R + d.R, c.st));
            if (hi_res > lo) {
                WnGemmArgs g = wn_gemm_default();
                g.M = d.R; g.N = d.R; g.K = T;
                g.A = ws + w.dXall + (long)(lo + 1) * BRT; g.lda = T; g.a_zstride = (long)d.R * T; g.a_lstride = BRT;
                g.B = ws + w.Z + (long)lo * BRT; g.ldb = T; g.b_zstride = (long)d.R * T; g.b_lstride = BRT; g.b_clen = T;
                g.tag = "dw_res";
                DwOut o = dw_out_plain(grads + lb_lo + y.o_res_w, d.R, grads + lb_lo + y.o_res_b);
                o.out_lstride = -y.LB; o.rowsum_lstride = -y.LB;
                WN_TRY(dw_gemm(c, g, o, hi_res - lo));
            }
        }
        {   // d aux_1x1_{sigmoid,tanh}.l.weight
            DwOut o;
            o.out = grads + lb_lo + y.o_asig_w;
            o.m_seg = d.R; o.m_seg_stride = y.o_atanh_w - y.o_asig_w; o.m_stride = d.A;
            o.n_seg = 0x7fffffff; o.n_seg_stride = 0; o.n_stride = 1;
            o.rowsum_out = nullptr; o.out_lstride = -y.LB; o.rowsum_lstride = 0;
            WnGemmArgs g = wn_gemm_default();
            g.tag = "dw_aux";
            g.M = 2 * d.R; g.N = d.A;
            if (d.U > 0) {
                // through the upsampling layer: dG[f] = sum_j w[j] dP[fU+j]; dW = dG.h^T + b_up*dc (x) 1
                if (aux_fused)
                    WN_TRY(wn_aux_finish(ws + w.dGp + (long)lo * B * 2 * d.R * (T / 16), (long)B * 2 * d.R * (T / 16),
                                         ws + w.qp + (long)lo * B * T, (long)B * T, ws + w.dG,
                                         ws + w.dw_partial + (long)lo * B * 2 * d.R * Ue, B, T, 2 * d.R, Ue, F, nl, c.st));
                else
                    WN_TRY(wn_aux_bwd(ws + w.P + (long)lo * P_L, P_L, ws + w.G + (long)lo * 2 * d.R * F, g_bstride, upw,
                                      ws + w.dG, ws + w.dw_partial + (long)lo * B * 2 * d.R * Ue, B, T, 2 * d.R, Ue, F, nl, c.st));
                g.K = F;
                g.A = ws + w.dG; g.lda = F; g.a_zstride = (long)2 * d.R * F; g.a_lstride = (long)B * 2 * d.R * F;
                g.B = h; g.ldb = F; g.b_zstride = (long)d.A * F; g.b_lstride = 0; g.b_clen = F;
                o.addend_m = dc; o.addend_scale_ptr = params + y.up_b; o.addend_lstride = 2 * d.R;
            } else {
                g.K = T;
                g.A = ws + w.P + (long)lo * P_L; g.lda = T; g.a_zstride = (long)2 * d.R * T; g.a_lstride = P_L;
                g.B = h; g.ldb = T; g.b_zstride = (long)d.A * T; g.b_lstride = 0; g.b_clen = T;
                o.addend_m = nullptr; o.addend_scale_ptr = nullptr; o.addend_lstride = 0;
            }
            WN_TRY(dw_gemm(c, g, o, nl));
        }
        return 0;
    };

    // Weight gradients are issued for groups of walked layers: a whole bucket in serial mode (largest launches), at
    // most WN_DW_FLUSH_DEFAULT layers in overlap mode so that they start while the chain is still running; flags bits
    // 8..15 override the group size.  (The split-K plan, hence the rounding, depends on the group size.)
    int fmax = (flags >> 8) & 0xff;
    if (fmax == 0) fmax = (side.rt && !(flags & WN_FLAG_BWD_OVERLAP_HEAD)) ? WN_DW_FLUSH_DEFAULT : d.L;
    int bucket_hi = d.L;  // layers [l, bucket_hi) have been walked but not flushed yet
    for (int l = d.L - 1; l >= 0; --l) {
        const int dil = dilation_of(cfg, l);
        const long lb = layer_base(y, d, l);
        const float* Sl = ws + w.Sg + (long)l * BRT;
        const float* Gtl = ws + w.Gt + (long)l * BRT;   // any-size path only: the fused forward saves s and z = s * tanh
        const float* Zl = ws + w.Z + (long)l * BRT;   // second gate operand of the fused kernels: z = s * tanh (g = z / s)
        const int gz = 1;
        float* dP = ws + w.P + (long)l * P_L;
        const float* dXn = (l + 1 < d.L) ? ws + w.dXall + (long)(l + 1) * BRT : nullptr;  // null: dead (last layer)
        float* dXl = ws + w.dXall + (long)l * BRT;
        if (chain) {
            if (l == d.L - 1) {   // head of the chain: gate' of the last layer on its rows of dZs (no dX input)
                WN_TRY(wn_fused_bwd_chain_head(ws + w.dZs + (long)l * d.R * T, zs_bstride, Sl, Zl, gz, dP,
                                               ws + w.G + (long)l * 2 * d.R * F, g_bstride, upw, Ue, F,
                                               aux_fused ? ws + w.dGp + (long)l * B * 2 * d.R * (T / 16) : nullptr,
                                               aux_fused ? ws + w.qp + (long)l * B * T : nullptr, B, T, t0, c.st));
            }
            if (l > 0) {  // dX_l from dP_l, and gate' of layer l-1 from it
                const long lbp = layer_base(y, d, l - 1);
                WN_TRY(wn_fused_bwd_chain(ws + w.wd_b + (long)l * d.K * 2 * d.R * d.R, dP, dXn, dXl, params + lbp + y.o_res_w,
                                          ws + w.dZs + (long)(l - 1) * d.R * T, zs_bstride, ws + w.Sg + (long)(l - 1) * BRT,
                                          ws + w.Z + (long)(l - 1) * BRT, gz, ws + w.P + (long)(l - 1) * P_L,
                                          ws + w.G + (long)(l - 1) * 2 * d.R * F, g_bstride, upw, Ue, F,
                                          aux_fused ? ws + w.dGp + (long)(l - 1) * B * 2 * d.R * (T / 16) : nullptr,
                                          aux_fused ? ws + w.qp + (long)(l - 1) * B * T : nullptr, B, T, d.K, dil,
                                          (w.img_floats > 0) ? ws + w.img_taps + (long)l * (wn_fused_image_floats(d.K, d.L, 1) / d.L) : nullptr,
                                          (w.img_floats > 0) ? ws + w.img_res + (long)(l - 1) * (wn_fused_image_floats(d.K, d.L, 2) / d.L) : nullptr,
                                          t0, c.st));
            } else {      // tail: dX_0
                WN_TRY(wn_fused_bwd_dx(ws + w.wd_b, dP, dXn, dXl, B, T, d.K, dil, 1, c.st));
            }
        } else if (c.fused) {
            // dZ = Wskip^T dSk (+ Wres^T dXn) -> gate' -> dP
            if (aux_fused)
                WN_TRY(wn_fused_bwd_gate_aux(params + y.skip0 + (long)l * y.ls_skip, params + lb + y.o_res_w, ws + w.dSk, dXn,
                                             Sl, Zl, gz, dP, ws + w.G + (long)l * 2 * d.R * F, g_bstride, upw, Ue, F,
                                             ws + w.dGp + (long)l * B * 2 * d.R * (T / 16), ws + w.qp + (long)l * B * T, B, T,
                                             d.S, c.st));
            else
                WN_TRY(wn_fused_bwd_gate(params + y.skip0 + (long)l * y.ls_skip, params + lb + y.o_res_w, ws + w.dSk, dXn, Sl,
                                         Zl, gz, dP, B, T, d.S, c.split_bf16 ? 1 : 0, c.st));
            WN_TRY(wn_fused_bwd_dx(ws + w.wd_b + (long)l * d.K * 2 * d.R * d.R, dP, dXn, dXl, B, T, d.K, dil, c.split_bf16 ? 1 : 0, c.st));
        } else {
            // dZ = Wskip_l^T dSkip (+ Wres_l^T dX_{l+1}) -> gate' -> dP.  Wide models on the split kernels: gate' is the
            // epilogue of the LAST of the two contractions (dZ never leaves the chip for it).
            WnGemmArgs gs = wn_gemm_default();
            gs.M = d.R; gs.N = T; gs.K = d.S;
            gs.A = params + y.skip0 + (long)l * y.ls_skip; gs.lda = d.R;
            gs.B = ws + w.dSk; gs.ldb = T; gs.b_zstride = (long)d.S * T; gs.b_clen = T;
            gs.C = ws + w.dZ; gs.ldc = T; gs.c_zstride = (long)d.R * T;
            gs.nbatch = B; gs.tag = "bwd_dz_skip_layered";
            WnGemmArgs gr = wn_gemm_default();
            gr.M = d.R; gr.N = T; gr.K = d.R;
            gr.A = params + lb + y.o_res_w; gr.lda = d.R;
            gr.B = dXn; gr.ldb = T; gr.b_zstride = (long)d.R * T; gr.b_clen = T;
            gr.C = ws + w.dZ; gr.ldc = T; gr.c_zstride = (long)d.R * T;
            gr.accumulate = 1; gr.nbatch = B; gr.tag = "bwd_dz_res_layered";
            const bool epi = d.R % 128 == 0 && fw_gemm_split_ok(c, gs) && (!dXn || fw_gemm_split_ok(c, gr));
            GateEpi ge;
            ge.bw_S = Sl; ge.bw_Gt = Gtl; ge.bw_dP = dP;
            if (epi) {
                if (dXn) {
                    WN_TRY(fw_gemm(c, gs));
                    gr.tag = "bwd_dz_res_gate";
                    WN_TRY(fw_gemm(c, gr, &ge));
                } else {
                    gs.tag = "bwd_dz_skip_gate";
                    WN_TRY(fw_gemm(c, gs, &ge));
                }
            } else {
                WN_TRY(fw_gemm(c, gs));
                if (dXn) WN_TRY(fw_gemm(c, gr));
                WN_TRY(wn_gate_bwd(ws + w.dZ, Sl, Gtl, dP, B, T, d.R, c.st));
            }
            {   // dX_l = dX_{l+1} + sum_tap W_tap^T dP[t + (K-1-tap) d]
                WnGemmArgs g = wn_gemm_default();
                g.M = d.R; g.N = T; g.K = d.K * 2 * d.R;
                g.A = ws + w.wd_b + (long)l * d.K * 2 * d.R * d.R; g.lda = d.R;
                g.B = dP; g.ldb = T; g.b_zstride = (long)2 * d.R * T; g.b_clen = T;
                g.b_seg_len = 2 * d.R; g.b_seg_stride = 0; g.b_shift0 = -(d.K - 1) * dil; g.b_shift_step = dil;
                g.C = dXl; g.ldc = T; g.c_zstride = (long)d.R * T;
                if (dXn) { g.D = dXn; g.ldd = T; g.d_zstride = (long)d.R * T; }
                g.nbatch = B; g.tag = "bwd_dx_dilated";
                WN_TRY(fw_gemm(c, g));
            }
        }
        const int done = d.L - l;  // layers walked
        const bool bucket_end = (done % lpb == 0 || l == 0);
        if (bucket_end || bucket_hi - l >= fmax) {
            WN_TRY(side_link(side.rt, c.st, cl.st));  // dP, dX of layers [l, bucket_hi) are enqueued
            if (flags & WN_FLAG_BWD_OVERLAP_HEAD)       // the split-K partial buffers are shared with the head's launches
                WN_TRY(side_link(side.rt, cs.st, c.st));
            WN_TRY(flush_bucket(l, bucket_hi));
            bucket_hi = l;
            if (bucket_end) {
                if (events) rt_event_record(events[bucket], cl.st);
                bucket++;
            }
        }
    }
    const float* dXn = ws + w.dXall;  // dL/dx_0
    // ---- front conv: scatter over the token indices, or (large tables) the one-hot contraction ----
    if (wn_front_dw_supported(d.R, d.K, d.Q) &&
        wn_front_dw_partial_floats(B, T, d.R, d.K, d.Q) <= w.front_partial_floats) {
        WN_TRY(wn_front_dw(dXn, x, ws + w.front_partial, grads + y.causal_w, grads + y.causal_b, B, T, d.R, d.K, d.Q, cl.st));
    } else {
        WnGemmArgs g = wn_gemm_default();
        g.M = d.R; g.N = d.K * d.Q; g.K = T;
        g.A = dXn; g.lda = T; g.a_zstride = (long)d.R * T;
        g.B = ws + w.X; /* unused (b_index set) */ g.ldb = 0; g.b_zstride = 0; g.b_clen = T;
        g.b_seg_len = d.Q; g.b_shift0 = d.K - 1; g.b_shift_step = -1;
        g.b_index = x; g.b_index_zstride = T; g.b_index_mod = d.Q; g.tag = "dw_front_onehot";
        DwOut o;
        o.out = grads + y.causal_w; o.m_seg = 0x7fffffff; o.m_seg_stride = 0; o.m_stride = (long)d.Q * d.K;
        o.n_seg = d.Q; o.n_seg_stride = 1; o.n_stride = d.K;
        o.addend_m = nullptr; o.addend_scale_ptr = nullptr; o.rowsum_out = grads + y.causal_b;
        o.out_lstride = 0; o.addend_lstride = 0; o.rowsum_lstride = 0;
        WN_TRY(dw_gemm(cl, g, o));
    }
    // ---- upsampling layer parameters ----
    if (d.U > 0) {
        WnReduceArgs r;
        r.partial = ws + w.dw_partial; r.nz = d.L * B * 2 * d.R; r.M = 1; r.N = d.U;
        r.out = grads + y.up_w; r.m_seg = 0x7fffffff; r.n_seg = 0x7fffffff;
        r.m_seg_stride = 0; r.m_stride = 0; r.n_seg_stride = 0; r.n_stride = 1;
        r.scale = 1.0f; r.accumulate = 0; r.addend_m = nullptr; r.addend_scale_ptr = nullptr;
        r.scratch = ws + w.red_scratch; r.scratch_floats = w.red_scratch_floats;
        r.nl = 1; r.out_lstride = 0; r.addend_lstride = 0;
        WN_TRY(wn_reduce(&r, cl.st));
        // d b_up = sum_{l,o'} rowsum(Waux_l)[o'] * dc_l[o']
        WN_TRY(wn_dot(ws + w.rowsum_aux, ws + w.dc, (long)d.L * 2 * d.R, grads + y.up_b, 0, cl.st));
    }
    if (events) rt_event_record(events[bucket], cl.st);
    bucket++;
    WN_TRY(side_link(side.rt, cs.st, c.st));  // join: the caller's stream continues after every gradient
    return rt_check("wn_backward");
}

// ------------------------------------------------------------------------------------------
extern "C" int wn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                            float lr, float beta1, float beta2, float eps, float weight_decay, int64_t skip_lo,
                            int64_t skip_hi, void* stream) {
    api_enter();
    if (!params || !grads || !exp_avg || !exp_avg_sq || n <= 0 || step < 1) return fail(1, "bad wn_adam_step argument");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    WN_TRY(wn_adam(params, grads, exp_avg, exp_avg_sq, (long)n, (float)((double)lr / bc1), (float)sqrt(bc2), beta1, beta2, eps,
                   weight_decay, (long)skip_lo, (long)skip_hi, (wn_stream_t)stream));
    return rt_check("wn_adam_step");
}

// ------------------------------------------------------------------------------------------
// op-level entry points
// ------------------------------------------------------------------------------------------
extern "C" int wn_op_front(const float* weight, const float* bias, const int64_t* x, float* out, float* scratch, int B, int T,
                           int Q, int R, int K, void* stream) {
    api_enter();
    WnCopy4 cp;
    cp.n0 = K; cp.n1 = Q; cp.n2 = R; cp.nl = 1;
    cp.d0 = (long)Q * R; cp.d1 = R; cp.d2 = 1; cp.dl = 0;
    cp.s0 = 1; cp.s1 = K; cp.s2 = (long)Q * K; cp.sl = 0;
    WN_TRY(wn_copy4(scratch, weight, &cp, (wn_stream_t)stream));
    WN_TRY(wn_front_gather(x, scratch, bias, out, B, T, Q, R, K, (wn_stream_t)stream));
    return rt_check("wn_op_front");
}

extern "C" int wn_op_causal_conv(const float* weight, const float* bias, const float* x, float* y, float* scratch, int B, int T,
                                 int Cin, int Cout, int K, int dilation, void* stream) {
    api_enter();
    WnCopy4 cp;  // scratch[(tap*Cin + i)*Cout + o] = W[o][i][tap]
    cp.n0 = K; cp.n1 = Cin; cp.n2 = Cout; cp.nl = 1;
    cp.s0 = 1; cp.s1 = K; cp.s2 = (long)Cin * K; cp.sl = 0;
    cp.d0 = (long)Cin * Cout; cp.d1 = Cout; cp.d2 = 1; cp.dl = 0;
    WN_TRY(wn_copy4(scratch, weight, &cp, (wn_stream_t)stream));
    WnGemmArgs g = wn_gemm_default();
    g.M = Cout; g.N = T; g.K = K * Cin;
    g.A = scratch; g.lda = Cout;
    g.B = x; g.ldb = T; g.b_zstride = (long)Cin * T; g.b_clen = T;
    g.b_seg_len = Cin; g.b_seg_stride = 0; g.b_shift0 = (K - 1) * dilation; g.b_shift_step = -dilation;
    g.C = y; g.ldc = T; g.c_zstride = (long)Cout * T;
    g.bias = bias; g.nbatch = B;
    WN_TRY(wn_gemm_launch(&g, (wn_stream_t)stream));
    return rt_check("wn_op_causal_conv");
}

// UpSampling.forward (wavenet.py:141-154): y[b][c][f U + j] = x[b][c][f] w[j] + bias  (ConvTranspose2d (1,U)/(1,U), one kernel
// shared by all channels); weight [U], bias [1] or NULL.
extern "C" int wn_op_upsampling(const float* weight, const float* bias, const float* x, float* y, int B, int C, int F, int U,
                                void* stream) {
    api_enter();
    if (!weight || !x || !y || B < 1 || C < 1 || F < 1 || U < 1) return fail(1, "bad argument");
    WN_TRY(wn_decode_ctx_aux_rows(x, weight, bias, y, B, C, F, U, F * U, 0, 0, (wn_stream_t)stream));
    return rt_check("wn_op_upsampling");
}

// dst (B, C, R) = src (B, R, C) transposed: the layout change between the reference's logits (B, T, Q) (wavenet.py:522) and the
// kernels' (B, Q, T), for a gradient that arrives from an external loss (nets/wavenet.py: the autograd bridge).
extern "C" int wn_op_transpose_last2(const float* src, float* dst, int B, int R, int C, void* stream) {
    api_enter();
    if (!src || !dst || src == dst || B < 1 || R < 1 || C < 1 || (long)((R + 31) / 32) > 65535 || B > 65535) return fail(1, "bad argument");
    WN_TRY(wn_transpose_last2(src, dst, B, R, C, (wn_stream_t)stream));
    return rt_check("wn_op_transpose_last2");
}

// ------------------------------------------------------------------------------------------
// autoregressive decode (wavenet.py:309-511, 538-549)
// ------------------------------------------------------------------------------------------
static int decode_plan(const WnConfig* cfg, Dims* d, WnDecodePlan* pl) {
    WN_TRY(check_cfg(cfg, d));
    wn_decode_make_plan(d->Qo, d->A, d->R, d->S, d->L, d->K, cfg->dilation_depth, pl);
    if (d->Q > 256) pl->ok = 0;
    if (d->Qo != d->Q && (d->Qo % 3 != 0 || d->Qo / 3 > 64)) pl->ok = 0;  // mixture head: one lane per component
    if (!pl->ok)
        return fail(3, "decode kernel: configuration not covered (needs n_resch<=64, n_skipch<=256, n_quantize<=256, "
                       "out_channels<=256, kernel_size<=3); use the layer-wise path");
    return 0;
}

extern "C" int wn_decode_supported(const WnConfig* cfg) {
    Dims d;
    WnDecodePlan pl;
    const int rc = decode_plan(cfg, &d, &pl);
    api_enter();
    return rc == 0 ? 1 : 0;
}

extern "C" int64_t wn_decode_pack_floats(const WnConfig* cfg) {
    Dims d;
    WnDecodePlan pl;
    if (decode_plan(cfg, &d, &pl)) return -1;
    return pl.total_floats;
}

extern "C" int64_t wn_decode_state_floats(const WnConfig* cfg) {
    Dims d;
    WnDecodePlan pl;
    if (decode_plan(cfg, &d, &pl)) return -1;
    return pl.queue_floats > 0 ? pl.queue_floats : 4;
}

extern "C" int64_t wn_decode_stream_bytes(const WnConfig* cfg) {
    Dims d;
    WnDecodePlan pl;
    if (decode_plan(cfg, &d, &pl)) return -1;
    return pl.stream_f4 * 16;
}

extern "C" int wn_decode_pack(const WnConfig* cfg, const float* params, float* wpack, void* stream) {
    api_enter();
    Dims d;
    WnDecodePlan pl;
    WN_TRY(decode_plan(cfg, &d, &pl));
    if (!params || !wpack) return fail(1, "NULL argument");
    const Lay y = make_lay(d);
    wn_stream_t st = (wn_stream_t)stream;
    const long lb0 = layer_base(y, d, 0), lstep = -y.LB;
    WnDecodePackArgs pa;
    pa.Q = d.Qo; pa.R = d.R; pa.S = d.S; pa.L = d.L; pa.K = d.K;  // Q of the packer = rows of conv_post_2
    pa.plan = pl;
    pa.params = params;
    pa.lb0 = lb0; pa.lstep = lstep;
    pa.o_dsig_w = y.o_dsig_w; pa.o_dtanh_w = y.o_dtanh_w; pa.o_res_w = y.o_res_w;
    pa.skip0 = y.skip0; pa.ls_skip = y.ls_skip; pa.post1_w = y.post1_w; pa.post2_w = y.post2_w;
    pa.stream = wpack;
    WN_TRY(wn_decode_pack_stream(&pa, st));
    // side tables: cvec (all constant terms of the gate pre-activation), summed skip bias, the aux
    // weights as the [a][l*2R+o'] operand of the aux-rate GEMM, a vector of ones (U == 0)
    WnCvecArgs ca;
    ca.params = params;
    ca.off_dsig_b = lb0 + y.o_dsig_b; ca.off_dtanh_b = lb0 + y.o_dtanh_b;
    ca.off_asig_w = lb0 + y.o_asig_w; ca.off_atanh_w = lb0 + y.o_atanh_w;
    ca.off_asig_b = lb0 + y.o_asig_b; ca.off_atanh_b = lb0 + y.o_atanh_b;
    ca.ls_dil = lstep; ca.ls_aux = lstep;
    ca.off_up_b = y.up_b;
    ca.L = d.L; ca.R = d.R; ca.A = d.A;
    ca.cvec = wpack + pl.off_cvec;
    ca.rowsum_aux = wpack + pl.off_wauxf;  // scratch: overwritten by the aux weights below
    WN_TRY(wn_cvec(&ca, st));
    WN_TRY(wn_sum_layers(params, y.skip0 + (long)d.S * d.R, y.ls_skip, d.L, d.S, wpack + pl.off_bskip, st));
    WnCopy4 cp;
    for (int half = 0; half < 2; ++half) {
        const long asrc = lb0 + (half ? y.o_atanh_w : y.o_asig_w);
        cp.n0 = 1; cp.n1 = d.A; cp.n2 = d.R; cp.nl = d.L;
        cp.s0 = 0; cp.s1 = 1; cp.s2 = d.A; cp.sl = lstep;
        cp.d0 = 0; cp.d1 = (long)d.L * 2 * d.R; cp.d2 = 1; cp.dl = 2 * d.R;
        WN_TRY(wn_copy4(wpack + pl.off_wauxf + (long)half * d.R, params + asrc, &cp, st));
    }
    WN_TRY(wn_fill(wpack + pl.off_one, 1.0f, 64, st));
    return rt_check("wn_decode_pack");
}

extern "C" int wn_decode_aux(const WnConfig* cfg, int B, int F, const float* wpack, const float* h, float* G,
                             void* stream) {
    api_enter();
    Dims d;
    WnDecodePlan pl;
    WN_TRY(decode_plan(cfg, &d, &pl));
    if (!wpack || !h || !G || B <= 0 || F <= 0) return fail(1, "bad argument");
    const int nG = d.L * 2 * d.R;
    // G[b] (F x nG) = h[b]^T (F x A) . waux_f (A x nG)
    WnGemmArgs g = wn_gemm_default();
    g.M = F; g.N = nG; g.K = d.A;
    g.A = h; g.lda = F; g.a_zstride = (long)d.A * F;
    g.B = wpack + pl.off_wauxf; g.ldb = nG; g.b_zstride = 0; g.b_clen = nG;
    g.C = G; g.ldc = nG; g.c_zstride = (long)F * nG;
    g.nbatch = B; g.tag = "decode_aux_frames";
    WN_TRY(wn_gemm_launch(&g, (wn_stream_t)stream));
    return rt_check("wn_decode_aux");
}

extern "C" int wn_decode_steps(const WnConfig* cfg, int B, const float* params, const float* wpack, const float* G, int F,
                               int n_pad, int64_t* samples, int64_t Ttot, const int32_t* t_forced, const int32_t* t_end,
                               int p0, int p1, float* state, const float* uniforms, float* logits_out, int mode,
                               float* wave_out, float log_scale_min, void* stream) {
    api_enter();
    Dims d;
    WnDecodePlan pl;
    WN_TRY(decode_plan(cfg, &d, &pl));
    if (!params || !wpack || !G || !samples || !t_forced || !t_end || !state) return fail(1, "NULL argument");
    if (B <= 0 || F <= 0 || n_pad < 0 || p0 < 0 || p1 < p0 || Ttot <= 0 || p1 > Ttot - 1)
        return fail(1, "bad decode range: B=%d F=%d n_pad=%d steps [%d,%d) Ttot=%ld", B, F, n_pad, p0, p1, (long)Ttot);
    if (mode != 0 && mode != 1 && mode != 2) return fail(1, "mode should be 0 (argmax), 1 (sampling) or 2 (mixture of logistics)");
    if (mode != 0 && !uniforms) return fail(1, "sampling modes need the uniform draws");
    if (mode == 2 && (d.Qo % 3 != 0 || d.Qo == d.Q)) return fail(1, "mode 2 needs out_channels = 3 * n_mixture");
    if (p1 == p0) return 0;
    const Lay y = make_lay(d);
    WnDecodeArgs a;
    a.Q = d.Q; a.Qo = d.Qo; a.A = d.A; a.R = d.R; a.S = d.S; a.L = d.L; a.K = d.K; a.depth = cfg->dilation_depth;
    a.plan = pl;
    a.wpack = wpack;
    a.params = params;
    a.off_causal_w = y.causal_w; a.off_causal_b = y.causal_b;
    a.off_res_b0 = layer_base(y, d, 0) + y.o_res_b; a.res_b_lstride = -y.LB;
    a.off_post1_b = y.post1_b; a.off_post2_b = y.post2_b;
    a.upw = d.U > 0 ? params + y.up_w : wpack + pl.off_one;
    a.Ue = d.U > 0 ? d.U : 1;
    a.G = G; a.g_bstride = (long)F * d.L * 2 * d.R; a.F = F; a.n_pad = n_pad;
    a.samples = samples; a.s_bstride = Ttot;
    a.t_forced = t_forced; a.t_end = t_end;
    a.p0 = p0; a.p1 = p1;
    a.queues = state; a.q_bstride = pl.queue_floats > 0 ? pl.queue_floats : 4;
    a.uniforms = uniforms; a.u_bstride = Ttot;   // mode 2: rows of nm+1 draws, indexed (b*Ttot + p+1)*(nm+1)
    a.logits_out = logits_out; a.lo_bstride = Ttot * d.Qo;
    a.mode = mode;
    a.wave_out = wave_out; a.w_bstride = Ttot; a.log_scale_min = log_scale_min;
    WN_TRY(wn_decode_launch(&a, B, (wn_stream_t)stream));
    return rt_check("wn_decode_steps");
}

// ------------------------------------------------------------------------------------------
// any-size decode: the queue algorithm (wavenet.py:397-511) as layer-wise launches.  Utterances are the
// contiguous axis of every matrix ("time" of the contraction kernels = utterance index), so one step
// of all utterances is ~100 launches of the training kernels on [channels x B] operands: weights are
// read once per step for the whole batch.  Used when the persistent decode kernel does not cover the
// model size (e.g. the n_resch = 512 recipe default).
// ------------------------------------------------------------------------------------------
struct DlLay {
    Ws w;  // packed-weight region of a (B=1, T=Ue) training workspace
    long queues, xin, P, Sg, Gt, Zcat, gstep, skpart, O1, O2, logits, total;
    long qfloats_per_utt;
    // persistent path (wn_dlp.hip), when the plan covers the model and nb <= WN_DLP_BMAX
    WnDlpPlan dlp;
    long dlp_w, dlp_post, dlp_cfold, dlp_fold, dlp_gz, dlp_gx, dlp_gs, dlp_go, dlp_gl, dlp_flags, dlp_pq, dlp_err;
    int dlp_flags_on;        // 1: wn_dlpf.hip (plain vectors + flags) runs this model / batch
    int dlp_grid, dlp_capacity;   // workgroups of the persistent launch the plan asks for / the device keeps resident at once
};

// granules (mode bit WN_DECODE_GRANULES): the persistent launches hand their vectors over as 8-byte granules everywhere
// (wn_dlp.hip, wn_dlpm.hip) instead of plain vectors + flags where wn_dlpf.hip covers the plan -- the state layout depends on it,
// so every call of one decode passes the same bit.
static int dl_layout(const WnConfig* cfg, const Dims& d, int nb, bool granules, DlLay* y) {
    const int Ue = d.U > 0 ? d.U : 1;
    WN_TRY(make_ws(d, 1, Ue, &y->w, /*training*/ false));
    long sumd = 0;
    for (int l = 0; l < d.L; ++l) sumd += dilation_of(cfg, l);
    y->qfloats_per_utt = (long)(d.K - 1) * sumd * d.R;
    long o = y->w.total;
#define DCARVE(field, n) \
    y->field = o;        \
    o += al64((long)(n));
    DCARVE(queues, y->qfloats_per_utt * nb + 64);
    DCARVE(xin, (long)d.L * d.K * d.R * nb);
    DCARVE(P, (long)2 * d.R * nb);
    DCARVE(Sg, (long)d.R * nb);
    DCARVE(Gt, (long)d.R * nb);
    DCARVE(Zcat, (long)d.L * d.R * nb);
    DCARVE(gstep, (long)d.L * 2 * d.R * nb);
    DCARVE(skpart, (long)d.L * d.S * nb);
    DCARVE(O1, (long)d.S * nb);
    DCARVE(O2, (long)d.S * nb);
    DCARVE(logits, (long)d.Qo * nb);
    // up to WN_DLP_BMAX utterances: the VALU kernel (wn_dlp.hip); up to WN_DLPM_BMAX: the matrix-core kernel (wn_dlpm.hip)
    // (the flag hand-off kernel wn_dlpf.hip costs the same ~250 us per step for 2 .. 16 utterances at n_resch 512, the VALU kernel
    // 252 / 321 / 366 for 2 / 3 / 4: from 2 utterances on where wn_dlpf.hip covers the matrix-core plan, from WN_DLPM_BMIN otherwise)
    int wide = nb >= WN_DLPM_BMIN ? 1 : 0;
    if (!wide && nb >= 2 && WN_DLPF_ENABLE && !granules) {
        WnDlpPlan pw;
        wn_dlp_make_plan(d.Q, d.Qo, d.R, d.S, d.L, d.K, 1, &pw);
        if (wn_dlpf_covers(&pw)) wide = 1;
    }
    wn_dlp_make_plan(d.Q, d.Qo, d.R, d.S, d.L, d.K, wide, &y->dlp);
    {
        const bool flags_ok = WN_DLPF_ENABLE && !granules && wn_dlpf_covers(&y->dlp);
        if (nb > (y->dlp.wide ? (flags_ok ? WN_DLPF_BMAX : WN_DLPM_BMAX) : WN_DLP_BMAX)) y->dlp.ok = 0;
    }
    const int dlp_blocks = y->dlp.wide ? (nb + WN_DLPM_CB - 1) / WN_DLPM_CB : 1;   // k_dlpm: a set of units per block of 16 utterances
    if (y->dlp.ok && y->dlp.NU * dlp_blocks > WN_DLPM_MAXWG) y->dlp.ok = 0;
    y->dlp_flags_on = (y->dlp.ok && WN_DLPF_ENABLE && !granules && wn_dlpf_covers(&y->dlp)) ? 1 : 0;
    y->dlp_grid = y->dlp_capacity = 0;
    if (y->dlp.ok) {
        // every workgroup of the launch waits for the others: all of them must be resident at once.  Asked of the device (occupancy
        // of the chosen kernel x CUs) here, where the path is chosen -- a grid that does not fit (a partitioned GPU, fewer CUs)
        // decodes by layer-wise launches instead of running its polls into their time-outs.
        y->dlp_grid = y->dlp.NU * dlp_blocks;
        y->dlp_capacity = y->dlp_flags_on ? wn_dlpf_capacity(&y->dlp) : (y->dlp.wide ? wn_dlpm_capacity(&y->dlp) : wn_dlp_capacity(&y->dlp));
        if (y->dlp_grid > y->dlp_capacity) { y->dlp.ok = 0; y->dlp_flags_on = 0; }
    }
    if (y->dlp.ok) {
        const WnDlpPlan& pl = y->dlp;
        DCARVE(dlp_w, (long)(d.L + 1) * pl.NU * pl.stage_floats);
        DCARVE(dlp_post, (long)pl.NU * pl.post_floats);
        DCARVE(dlp_cfold, (long)d.L * 2 * d.R);
        DCARVE(dlp_fold, (long)2 * d.R * d.R);
        // hand-off regions: 8-byte granules (two floats each) [rows][nb], or -- wn_dlpf.hip: plain vectors + one flag per unit and
        // block -- floats [rows][Bp], Bp = 16 * blocks (which fit the same regions carved with Bp columns)
        const long Bp = pl.wide ? (long)dlp_blocks * WN_DLPM_CB : nb;
        DCARVE(dlp_gz, 2L * 2 * d.R * Bp);
        DCARVE(dlp_gx, 2L * 2 * d.R * Bp);
        DCARVE(dlp_gs, 2L * d.S * Bp);
        DCARVE(dlp_go, 2L * d.S * Bp);
        DCARVE(dlp_gl, 2L * d.Qo * Bp);
        DCARVE(dlp_flags, 2L * pl.NU * dlp_blocks);
        // private copies of the dilation queues (not with the flag hand-off: its shared rings are read by every unit)
        DCARVE(dlp_pq, y->dlp_flags_on ? 64 : (pl.wide ? (long)pl.NU * dlp_blocks * y->qfloats_per_utt * WN_DLPM_CB : (long)pl.NU * y->qfloats_per_utt * nb));
        DCARVE(dlp_err, 1024);   // error word (+ the stamps of a timing build)
    }
#undef DCARVE
    y->total = o;
    return 0;
}

static void dl_ctx(Ctx* c, const WnConfig* cfg, const Dims& d, const DlLay& y, int nb, float* state, void* stream) {
    c->cfg = cfg;
    c->d = d;
    c->y = make_lay(d);
    c->w = y.w;
    c->B = 1;
    c->T = nb;
    c->ws = state;
    c->st = (wn_stream_t)stream;
    c->fused = false;
    // exact f32 MFMA here: with a handful of utterance columns the contractions are weight-streaming bound and
    // the split path would re-split (or stream 1.5x the bytes of) the weights on every step
    c->split_bf16 = false;
    c->dw_products = 6;
    c->params = nullptr;
    c->have_pre = false;
}

extern "C" int64_t wn_decode_layered_state_floats(const WnConfig* cfg, int B, int mode) {
    Dims d;
    if (check_cfg(cfg, &d) || B < 1) return -1;
    DlLay y;
    if (dl_layout(cfg, d, B, (mode & WN_DECODE_GRANULES) != 0, &y)) return -1;
    return y.total;
}

// Float offset inside `state` of the error word of the persistent path (an int: non-zero after a launch whose workgroups
// timed out waiting for each other), or -1 when wn_decode_layered_steps runs as layer-wise launches for this model / B.
extern "C" int64_t wn_decode_layered_error_offset(const WnConfig* cfg, int B, int mode) {
    Dims d;
    if (check_cfg(cfg, &d) || B < 1) return -1;
    DlLay y;
    if (dl_layout(cfg, d, B, (mode & WN_DECODE_GRANULES) != 0, &y) || !y.dlp.ok) return -1;
    return y.dlp_err;
}

// What the persistent launch of (cfg, B, mode) needs and what the current device offers: *workgroups = its grid (0: no plan
// covers this model / B), *capacity = workgroups of that kernel resident at once (occupancy x CUs).  Returns 1 when the
// persistent path will be used (grid <= capacity), 0 when wn_decode_layered_steps decodes by layer-wise launches, < 0 on a bad
// argument.
extern "C" int wn_decode_layered_residency(const WnConfig* cfg, int B, int mode, int* workgroups, int* capacity) {
    api_enter();
    Dims d;
    if (check_cfg(cfg, &d) || B < 1) return -1;
    DlLay y;
    if (dl_layout(cfg, d, B, (mode & WN_DECODE_GRANULES) != 0, &y)) return -1;
    if (workgroups) *workgroups = y.dlp_grid;
    if (capacity) *capacity = y.dlp_capacity;
    return y.dlp.ok ? 1 : 0;
}

// Packs the weights into `state` (which must be zero-filled first: the queues start from zero history) and
// computes the aux projections G (B, F, L*2R) of all layers at the aux rate.  params == NULL: `state` already holds the
// weights packed by an earlier call (same cfg, B and parameters) -- only the projection of this window of h is computed
// (windowed decoding without an upsampling layer calls this once per chunk of steps).
extern "C" int wn_decode_layered_prepare(const WnConfig* cfg, int B, int F, const float* params, const float* h, float* G,
                                         float* state, int64_t state_floats, int mode, void* stream) {
    api_enter();
    Dims d;
    WN_TRY(check_cfg(cfg, &d));
    if (!h || !G || !state || B < 1 || F < 1) return fail(1, "bad argument");
    DlLay y;
    WN_TRY(dl_layout(cfg, d, B, (mode & WN_DECODE_GRANULES) != 0, &y));
    if (state_floats < y.total) return fail(1, "decode state too small: %ld < %ld floats", (long)state_floats, y.total);
    Ctx c;
    dl_ctx(&c, cfg, d, y, B, state, stream);
    if (params) WN_TRY(pack_weights(c, params));
    if (params && y.dlp.ok) {   // persistent path: per-stage weight images with the res 1x1 folded into the next layer's newest tap
        const Lay& lay = c.y;
        const long lb0 = layer_base(lay, d, 0), lstep = -lay.LB;
        for (int s = 0; s <= d.L; ++s) {
            if (s >= 1 && s < d.L) {   // fold[o'][i] = sum_j Wd_new(s)[o'][j] Wres(s-1)[j][i]
                WnGemmArgs f = wn_gemm_default();
                f.M = 2 * d.R; f.N = d.R; f.K = d.R;
                f.A = state + y.w.wd_f + (long)s * d.K * d.R * 2 * d.R + (long)(d.K - 1) * d.R * 2 * d.R; f.lda = 2 * d.R;
                f.B = params + layer_base(lay, d, s - 1) + lay.o_res_w; f.ldb = d.R; f.b_clen = d.R;
                f.C = state + y.dlp_fold; f.ldc = d.R;
                f.nbatch = 1; f.tag = "dlp_fold";
                WN_TRY(wn_gemm_launch(&f, c.st));
            }
            WnDlpPackArgs pa;
            pa.R = d.R; pa.S = d.S; pa.Qo = d.Qo; pa.L = d.L; pa.K = d.K; pa.plan = y.dlp; pa.stage = s;
            pa.params = params;
            pa.lb_s = s < d.L ? layer_base(lay, d, s) : 0;
            pa.lb_prev = s >= 1 ? layer_base(lay, d, s - 1) : 0;
            pa.o_dsig_w = lay.o_dsig_w; pa.o_dtanh_w = lay.o_dtanh_w; pa.o_res_w = lay.o_res_w;
            pa.skip_prev = s >= 1 ? lay.skip0 + (long)(s - 1) * lay.ls_skip : 0;
            pa.fold = state + y.dlp_fold;
            pa.dst = state + y.dlp_w + (long)s * y.dlp.NU * y.dlp.stage_floats;
            WN_TRY(wn_dlp_pack_stage(&pa, c.st));
        }
        WN_TRY(wn_dlp_pack_post(params, lay.post1_w, lay.post2_w, d.S, d.Qo, &y.dlp, state + y.dlp_post, c.st));
        WN_TRY(wn_dlp_cfold(params, state + y.w.cvec, state + y.w.wd_f, lb0, lstep, lay.o_res_b, d.L, d.R, d.K, state + y.dlp_cfold,
                            c.st));
    }
    const int nG = d.L * 2 * d.R;
    WnGemmArgs g = wn_gemm_default();  // G[b] (F x nG) = h[b]^T (F x A) . waux_f (A x nG)
    g.M = F; g.N = nG; g.K = d.A;
    g.A = h; g.lda = F; g.a_zstride = (long)d.A * F;
    g.B = state + y.w.waux_f; g.ldb = nG; g.b_zstride = 0; g.b_clen = nG;
    g.C = G; g.ldc = nG; g.c_zstride = (long)F * nG;
    g.nbatch = B; g.tag = "decode_aux_frames";
    WN_TRY(wn_gemm_launch(&g, c.st));
    return rt_check("wn_decode_layered_prepare");
}

extern "C" int wn_decode_layered_steps(const WnConfig* cfg, int B, const float* params, const float* G, int F, int n_pad,
                                       int64_t* samples, int64_t Ttot, const int32_t* t_forced, const int32_t* t_end, int p0,
                                       int p1, float* state, int64_t state_floats, const float* uniforms, float* logits_out,
                                       int mode, float* wave_out, float log_scale_min, void* stream) {
    api_enter();
    Dims d;
    WN_TRY(check_cfg(cfg, &d));
    if (!params || !G || !samples || !t_forced || !t_end || !state) return fail(1, "NULL argument");
    if (B <= 0 || F <= 0 || n_pad < 0 || p0 < 0 || p1 < p0 || Ttot <= 0 || p1 > Ttot - 1)
        return fail(1, "bad decode range: B=%d F=%d n_pad=%d steps [%d,%d) Ttot=%ld", B, F, n_pad, p0, p1, (long)Ttot);
    const bool by_launches = (mode & WN_DECODE_BY_LAUNCHES) != 0, granules = (mode & WN_DECODE_GRANULES) != 0;
    mode &= ~(WN_DECODE_BY_LAUNCHES | WN_DECODE_GRANULES);
    if (mode != 0 && mode != 1 && mode != 2) return fail(1, "mode should be 0 (argmax), 1 (sampling) or 2 (mixture of logistics)");
    if (mode != 0 && !uniforms) return fail(1, "sampling modes need the uniform draws");
    if (mode == 2 && (d.Qo % 3 != 0 || d.Qo == d.Q)) return fail(1, "mode 2 needs out_channels = 3 * n_mixture");
    DlLay y;
    WN_TRY(dl_layout(cfg, d, B, granules, &y));
    if (state_floats < y.total) return fail(1, "decode state too small: %ld < %ld floats", (long)state_floats, y.total);
    Ctx c;
    dl_ctx(&c, cfg, d, y, B, state, stream);
    const Lay& lay = c.y;
    const Ws& w = y.w;
    float* ws = state;
    const int nb = B;
    if (y.dlp.ok && mode != 2 && !by_launches) {
        // persistent path: ONE launch for the whole range of steps (wn_dlp.hip); the softmax head's two modes
        if (p1 == p0) return 0;
        WnDlpArgs a;
        a.Q = d.Q; a.Qo = d.Qo; a.R = d.R; a.S = d.S; a.L = d.L; a.K = d.K; a.depth = cfg->dilation_depth; a.nG = d.L * 2 * d.R;
        a.plan = y.dlp; a.B = nb;
        a.wpk = ws + y.dlp_w; a.wpost = ws + y.dlp_post; a.cfold = ws + y.dlp_cfold; a.bskip = ws + w.bskip;
        a.params = params; a.off_causal_w = lay.causal_w; a.off_causal_b = lay.causal_b;
        a.off_res_b0 = layer_base(lay, d, 0) + lay.o_res_b; a.res_b_lstride = -lay.LB;
        a.off_post1_b = lay.post1_b; a.off_post2_b = lay.post2_b;
        a.upw = d.U > 0 ? params + lay.up_w : ws + w.one; a.Ue = d.U > 0 ? d.U : 1; a.F = F; a.n_pad = n_pad;
        a.G = G; a.samples = samples; a.Ttot = Ttot; a.t_forced = t_forced; a.t_end = t_end; a.uniforms = uniforms;
        a.logits_out = logits_out; a.mode = mode; a.p0 = p0; a.p1 = p1;
        a.gz = reinterpret_cast<unsigned long long*>(ws + y.dlp_gz); a.gx = reinterpret_cast<unsigned long long*>(ws + y.dlp_gx);
        a.gs = reinterpret_cast<unsigned long long*>(ws + y.dlp_gs); a.go = reinterpret_cast<unsigned long long*>(ws + y.dlp_go);
        a.gl = reinterpret_cast<unsigned long long*>(ws + y.dlp_gl);
        a.pq = ws + y.dlp_pq; a.pq_unit_stride = y.qfloats_per_utt * (y.dlp.wide ? WN_DLPM_CB : nb);
        a.queues = ws + y.queues; a.qfloats = y.qfloats_per_utt;
        a.err = reinterpret_cast<int*>(ws + y.dlp_err);
        a.handoff = y.dlp_flags_on; a.Bp = y.dlp.wide ? ((nb + WN_DLPM_CB - 1) / WN_DLPM_CB) * WN_DLPM_CB : nb;
        a.flags = reinterpret_cast<unsigned long long*>(ws + y.dlp_flags);
        const int rc = y.dlp_flags_on ? wn_dlpf_launch(&a, c.st) : (y.dlp.wide ? wn_dlpm_launch(&a, c.st) : wn_dlp_launch(&a, c.st));
        if (rc == 4)
            return fail(4, "the persistent decode launch needs %d workgroups resident at once, the device keeps %d: "
                           "mode | WN_DECODE_BY_LAUNCHES decodes by layer-wise launches", y.dlp_grid, y.dlp_capacity);
        if (rc != 0) return fail(3, "wn_dlp%s_launch failed (rc=%d)", y.dlp_flags_on ? "f" : (y.dlp.wide ? "m" : ""), rc);
        return rt_check("wn_decode_layered_steps");
    }
    WnDlArgs a;
    a.nb = nb; a.L = d.L; a.K = d.K; a.R = d.R; a.Q = d.Q; a.depth = cfg->dilation_depth; a.nG = d.L * 2 * d.R;
    a.n_pad = n_pad; a.Ue = d.U > 0 ? d.U : 1; a.F = F;
    a.params = params; a.off_causal_w = lay.causal_w; a.off_causal_b = lay.causal_b;
    a.upw = d.U > 0 ? params + lay.up_w : ws + w.one;
    a.G = G; a.samples = samples; a.Ttot = Ttot;
    a.queues = ws + y.queues; a.xin = ws + y.xin; a.gstep = ws + y.gstep;
    const long RB = (long)d.R * nb;
    const bool gate_fused = d.R % 16 == 0;
    for (int p = p0; p < p1; ++p) {
        a.p = p;
        WN_TRY(wn_dl_inputs(&a, c.st));
        for (int l = 0; l < d.L; ++l) {
            const long lb = layer_base(lay, d, l);
            float* xin_l = ws + y.xin + (long)l * d.K * RB;
            float* z_l = ws + y.Zcat + (long)l * RB;
            {   // both rows of the gate: taps [history | newest] x packed dilated weights  (wavenet.py:540-541)
                WnDlMmArgs g;
                g.M = 2 * d.R; g.K = d.K * d.R; g.nb = nb;
                g.A = ws + w.wd_f + (long)l * d.K * d.R * 2 * d.R; g.lda = 2 * d.R; g.a_zstride = 0;
                g.B = xin_l; g.ldb = nb; g.b_zstride = 0;
                g.C = ws + y.P; g.ldc = nb; g.c_zstride = 0;
                g.bias = nullptr; g.D = nullptr; g.ldd = 0; g.relu = 0; g.nz = 1; g.tag = "dl_dilated";
                g.gate_R = 0; g.gate_g = nullptr; g.gate_c = nullptr;
                if (gate_fused) {  // z = sigmoid(.)*tanh(.) in the epilogue (wavenet.py:542-544): one launch less per layer
                    g.gate_R = d.R; g.gate_g = ws + y.gstep + (long)l * 2 * RB; g.gate_c = ws + w.cvec + (long)l * 2 * d.R;
                    g.C = z_l;
                }
                WN_TRY(wn_dl_mm(&g, c.st));
            }
            if (!gate_fused)
                WN_TRY(wn_gate_fwd(ws + y.P, ws + y.gstep + (long)l * 2 * RB, 0, ws + w.one, ws + w.cvec + (long)l * 2 * d.R,
                                   ws + y.Sg, ws + y.Gt, z_l, 1, nb, d.R, 1, nb, c.st));
            if (l + 1 < d.L) {  // next layer input = res_1x1(z) + x  (wavenet.py:546-548)
                WnDlMmArgs r;
                r.M = d.R; r.K = d.R; r.nb = nb;
                r.A = ws + w.wres_f + (long)l * d.R * d.R; r.lda = d.R; r.a_zstride = 0;
                r.B = z_l; r.ldb = nb; r.b_zstride = 0;
                r.C = xin_l + (long)d.K * RB + (long)(d.K - 1) * RB; r.ldc = nb; r.c_zstride = 0;
                r.bias = params + lb + lay.o_res_b;
                r.D = xin_l + (long)(d.K - 1) * RB; r.ldd = nb;
                r.relu = 0; r.nz = 1; r.tag = "dl_res";
                r.gate_R = 0; r.gate_g = nullptr; r.gate_c = nullptr;
                WN_TRY(wn_dl_mm(&r, c.st));
            }
        }
        {   // skip-sum over all layers + relu (wavenet.py:545,365-366): one launch over the layers, then a fixed-order sum
            WnDlMmArgs g;
            g.M = d.S; g.K = d.R; g.nb = nb;
            g.A = ws + w.wskip_f; g.lda = d.S; g.a_zstride = (long)d.R * d.S;
            g.B = ws + y.Zcat; g.ldb = nb; g.b_zstride = RB;
            g.C = ws + y.skpart; g.ldc = nb; g.c_zstride = (long)d.S * nb;
            g.bias = nullptr; g.D = nullptr; g.ldd = 0; g.relu = 0; g.nz = d.L; g.tag = "dl_skip";
            g.gate_R = 0; g.gate_g = nullptr; g.gate_c = nullptr;
            WN_TRY(wn_dl_mm(&g, c.st));
            WN_TRY(wn_dl_sum(ws + y.skpart, d.L, (long)d.S * nb, d.S, nb, ws + w.bskip, 1, ws + y.O1, c.st));
        }
        {
            WnDlMmArgs g;
            g.M = d.S; g.K = d.S; g.nb = nb;
            g.A = ws + w.w1_f; g.lda = d.S; g.a_zstride = 0;
            g.B = ws + y.O1; g.ldb = nb; g.b_zstride = 0;
            g.C = ws + y.O2; g.ldc = nb; g.c_zstride = 0;
            g.bias = params + lay.post1_b; g.D = nullptr; g.ldd = 0; g.relu = 1; g.nz = 1; g.tag = "dl_post1";
            g.gate_R = 0; g.gate_g = nullptr; g.gate_c = nullptr;
            WN_TRY(wn_dl_mm(&g, c.st));
        }
        {
            WnDlMmArgs g;
            g.M = d.Qo; g.K = d.S; g.nb = nb;
            g.A = ws + w.w2_f; g.lda = d.Qo; g.a_zstride = 0;
            g.B = ws + y.O2; g.ldb = nb; g.b_zstride = 0;
            g.C = ws + y.logits; g.ldc = nb; g.c_zstride = 0;
            g.bias = params + lay.post2_b; g.D = nullptr; g.ldd = 0; g.relu = 0; g.nz = 1; g.tag = "dl_post2";
            g.gate_R = 0; g.gate_g = nullptr; g.gate_c = nullptr;
            WN_TRY(wn_dl_mm(&g, c.st));
        }
        if (mode == 2)
            WN_TRY(wn_dl_select_mol(ws + y.logits, d.Qo / 3, nb, d.Q, samples, wave_out, Ttot, t_forced, t_end, p, uniforms,
                                    logits_out, log_scale_min, c.st));
        else
            WN_TRY(wn_dl_select(ws + y.logits, d.Qo, nb, samples, Ttot, t_forced, t_end, p, uniforms, logits_out, mode, c.st));
        WN_TRY(wn_dl_push(&a, c.st));
    }
    return rt_check("wn_decode_layered_steps");
}

// ---- parallel context walk (reference wavenet.py:338-349: the "prepare buffer" pass is a full forward) --------
// The context of a generation call (left padding + given samples, >= receptive field positions) is known up
// front, so its dilation queues need no sample-by-sample walk: the training forward's residual stack computes
// the layer inputs of all positions at once and the newest (K-1)*d_l of every layer are copied into the queues.
// The stack runs with the aux features at SAMPLE rate (configuration with upsampling_factor = 0, whose flat
// parameter layout is a prefix of the model's: the upsampling layer's parameters are the last entries), because
// the left padding replicates the first UPSAMPLED column (wavenet.py:336), which no frame-rate input can express.
static WnConfig ctx_cfg(const WnConfig* cfg) {
    WnConfig c0 = *cfg;
    c0.upsampling_factor = 0;
    return c0;
}

extern "C" int wn_decode_ctx_aux(const WnConfig* cfg, int B, int F, int Tctx, int n_pad, int pos0, const float* params,
                                 const float* h, float* h_ctx, void* stream) {
    api_enter();
    Dims d;
    WN_TRY(check_cfg(cfg, &d));
    if (!params || !h || !h_ctx || B < 1 || F < 1 || Tctx < 1 || n_pad < 0 || pos0 < 0) return fail(1, "bad argument");
    const Lay y = make_lay(d);
    const float* upw = d.U > 0 ? params + y.up_w : nullptr;
    const float* upb = d.U > 0 ? params + y.up_b : nullptr;
    WN_TRY(wn_decode_ctx_aux_rows(h, upw, upb, h_ctx, B, d.A, F, d.U, Tctx, n_pad, pos0, (wn_stream_t)stream));
    return rt_check("wn_decode_ctx_aux");
}

extern "C" size_t wn_decode_prefill_workspace_bytes(const WnConfig* cfg, int B, int Tctx) {
    if (!cfg) return 0;
    const WnConfig c0 = ctx_cfg(cfg);
    return wn_workspace_bytes(&c0, B, Tctx);
}

extern "C" int wn_decode_prefill(const WnConfig* cfg, int B, int Tctx, int pos0, const float* params, const int64_t* x_ctx,
                                 const float* h_ctx, void* wsp, size_t ws_bytes, float* state, int64_t state_floats, int state_B,
                                 int state_b0, int layered, int flags, void* stream) {
    api_enter();
    if (!cfg) return fail(1, "config is NULL");
    const WnConfig c0 = ctx_cfg(cfg);
    Ctx c;
    WN_TRY(make_ctx(&c, &c0, B, Tctx, wsp, ws_bytes, flags, stream));
    if (!params || !x_ctx || !h_ctx || !state) return fail(1, "NULL argument");
    if (pos0 < 0) return fail(1, "pos0=%d", pos0);
    if (state_b0 < 0 || state_b0 + B > state_B) return fail(1, "utterances [%d, %d) outside a state of %d", state_b0, state_b0 + B, state_B);
    const Dims& d = c.d;
    if (Tctx < wn_receptive_field(cfg)) return fail(1, "context of %d positions is shorter than the receptive field", Tctx);
    if (pos0 > 0 && Tctx < wn_receptive_field(cfg) + cfg->kernel_size - 1)
        return fail(1, "a context tail needs receptive field + kernel_size - 1 = %d positions, got %d",
                    wn_receptive_field(cfg) + cfg->kernel_size - 1, Tctx);
    float* qdst;
    long elem_stride, utt_stride;
    if (layered) {
        Dims dm;
        WN_TRY(check_cfg(cfg, &dm));
        DlLay y;
        WN_TRY(dl_layout(cfg, dm, state_B, (layered & WN_DECODE_GRANULES) != 0, &y));
        if (state_floats < y.total) return fail(1, "decode state too small: %ld < %ld floats", (long)state_floats, y.total);
        qdst = state + y.queues + state_b0; elem_stride = state_B; utt_stride = 1;
    } else {
        Dims dm;
        WnDecodePlan pl;
        WN_TRY(decode_plan(cfg, &dm, &pl));
        const long per = pl.queue_floats > 0 ? pl.queue_floats : 4;
        if (state_floats < per * state_B)
            return fail(1, "decode state too small: %ld < %ld floats", (long)state_floats, per * state_B);
        qdst = state + per * state_b0; elem_stride = 1; utt_stride = per;
    }
    WN_TRY(forward_stack(c, params, x_ctx, h_ctx));
    // decoding resumes at the last context position pos0 + Tctx-1 (its logits choose the first new sample)
    WN_TRY(wn_decode_fill_queues(c.ws + c.w.X, qdst, d.L, B, d.R, Tctx, d.K, cfg->dilation_depth, Tctx - 1, pos0, elem_stride,
                                 utt_stride, c.st));
    return rt_check("wn_decode_prefill");
}

extern "C" int wn_op_gemm(const struct WnGemmArgs* args, void* stream) {
    api_enter();
    WN_TRY(wn_gemm_launch(args, (wn_stream_t)stream));
    return rt_check("wn_op_gemm");
}
