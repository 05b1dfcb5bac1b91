// wn_api_decode.inl: autoregressive decode entry points (wavenet.py:309-511, 538-549): the one-workgroup kernel, the any-size path (persistent launches / layer-wise launches), the parallel context pass -- part of the ONE translation unit wn_api.hip (included at its end: the entry points share its file-local
// helpers -- error text, parameter layout, workspace carving, launch contexts).  Not compiled on its own.
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
    c->dw_f16_mul = 0.0f;
    c->dw_f16_mode = 0;
    c->mm_f16 = false;
    c->fused_f16 = false;
    c->chain_f16 = false;
    c->dw_ovf = nullptr;
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

