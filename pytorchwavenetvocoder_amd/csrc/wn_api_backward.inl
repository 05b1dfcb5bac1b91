// wn_api_backward.inl: wn_backward / wn_backward_window (autograd backward, train.py:538), wn_adam_step (train.py:457-460,539) -- part of the ONE translation unit wn_api.hip (included at its end: the entry points share its file-local
// helpers -- error text, parameter layout, workspace carving, launch contexts).  Not compiled on its own.
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
    if (c.split_bf16 && wn_gemm6_dw_eligible(&g)) {
        if (c.dw_f16_mul != 0.0f) {
            // fp16 pair split; the six-product launch behind it returns at once unless a gradient left fp16's range
            WN_TRY(wn_gemm6_dw_launch(&g, 3, c.dw_f16_mul, c.dw_ovf, c.st));
            WN_TRY(wn_gemm6_dw_launch(&g, 6, 0.0f, c.dw_ovf, c.st));
        } else {
            WN_TRY(wn_gemm6_dw_launch(&g, c.dw_products, 0.0f, nullptr, c.st));
        }
    } else
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
    if (c.dw_f16_mode) {   // before the side stream forks: overflow word := 0, a_mul := the scale of this call's gradient (wn_elem.h)
        const int tw0 = (t_first / 128) * 128;
        float* words = c.ws + c.w.dw_ovf;
        if (c.dw_f16_mode == 1) {
            WN_TRY(wn_dw_prepare(words, ldexpf(1.0f, ((flags >> WN_FLAG_DW_F16_EXP_SHIFT) & 63) + WN_DW_F16_HEADROOM), nullptr, 0, WN_DW_F16_HEADROOM, c.st));
        } else if (c.dw_f16_mode == 2) {
            WN_TRY(wn_dw_prepare(words, 0.0f, nullptr, 0, WN_DW_F16_HEADROOM, c.st));
        } else {   // nobody vouches for the size of this gradient: one pass over it (the loss window's columns)
            const long rows = (long)B * c.d.Qo;
            const int nchunk = (T - tw0 + 4095) / 4096;
            if (rows * nchunk > c.w.amax_partial_floats) return fail(1, "dlogits scan: partial buffer too small");
            WN_TRY(wn_absmax_rows(dlogits, rows, T, tw0, T - tw0, c.ws + c.w.amax_partial, c.st));
            WN_TRY(wn_dw_prepare(words, 0.0f, c.ws + c.w.amax_partial, (int)(rows * nchunk), WN_DW_F16_HEADROOM, c.st));
        }
    }
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
        WN_TRY(fw_gemm(c, g, nullptr, nullptr, t0, true));
    }
    {   // dSkip = W1^T dO2, masked by relu'(skip-sum)
        WnGemmArgs g = wn_gemm_default();
        g.M = d.S; g.N = Tw; g.K = d.S;
        g.A = params + y.post1_w; g.lda = d.S;
        g.B = ws + w.dO2 + t0; g.ldb = T; g.b_zstride = (long)d.S * T; g.b_clen = Tw;
        g.C = ws + w.dSk + t0; g.ldc = T; g.c_zstride = (long)d.S * T;
        g.E = ws + w.O1 + t0; g.lde = T; g.e_zstride = (long)d.S * T;
        g.nbatch = B; g.tag = "bwd_post1_dx";
        WN_TRY(fw_gemm(c, g, nullptr, nullptr, t0, true));
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
    // Fused skip + res weight gradients (k_dw_skipres: z of every layer read once instead of twice): with the fp16 pair split on,
    // the skip gradients wait for the data chain and are produced per bucket together with the res_1x1 gradients.
    // Only with ONE layer bucket, where the skip weights belong to that bucket (wn_bucket_range); with several, they are part of the
    // head bucket, whose event would then be the last one recorded and hold back the exchange of every layer bucket behind it.
    const bool skipres = c.split_bf16 && c.dw_f16_mul != 0.0f && d.L > 1 && lpb >= d.L && wn_dw_skipres_supported(d.S, d.R, d.L, d.L - 1);
    if (!skipres) {   // d skip_1x1.l.weight for all layers in one contraction; bias = rowsum(dSkip) for every layer
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
    if (events) rt_event_record(events[bucket], cs.st);   // head bucket: the post-net (+ every skip_1x1 with several layer buckets)
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
        WN_TRY(fw_gemm(c, g, nullptr, nullptr, t0, true));
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
        const int hi_res = hi < d.L ? hi : d.L - 1;
        const bool fuse = skipres && hi_res > lo;   // (a bucket holding only the last layer has no res_1x1 gradient)
        if (fuse) {
            // skip_1x1 and res_1x1 of layers [lo, hi) against one read of z; the partial sums land where the two separate launches
            // put them, so their conditional six-product launches (same split-K plan) and their reductions follow unchanged
            if (hi == d.L) WN_TRY(wn_fill(grads + layer_base(y, d, d.L - 1) + y.o_res_w, 0.0f, (long)d.R * d.R + d.R, c.st));
            const DwPlan p = dw_skipres_plan(d.S, nl, T, B);
            const int n_res = hi_res - lo;
            WnDwSkipRes a;
            a.S = d.S; a.nl = nl; a.n_res = n_res; a.K = T; a.nbatch = B; a.ksplit = p.ksplit; a.kchunk = p.kchunk;
            a.dS = ws + w.dSk; a.ds_ld = T; a.ds_zstride = (long)d.S * T;
            a.Z = ws + w.Z + (long)lo * BRT; a.z_ld = T; a.z_zstride = (long)d.R * T; a.z_lstride = BRT;
            a.dX = ws + w.dXall + (long)(lo + 1) * BRT; a.dx_ld = T; a.dx_zstride = (long)d.R * T; a.dx_lstride = BRT;
            a.Cskip = ws + w.partial; a.Cres = ws + w.partial2;
            a.rs_skip = hi == d.L ? ws + w.rs_partial : nullptr;   // rowsum(dSkip): once per step, by the first bucket
            a.rs_res = ws + w.rs_partial2;
            WN_TRY(wn_dw_skipres_launch(&a, c.dw_f16_mul, c.dw_ovf, c.st));
            // the redo launches (no work unless the word is up) and the reductions
            WnGemmArgs gs = wn_gemm_default();
            gs.M = d.S; gs.N = nl * d.R; gs.K = T;
            gs.A = a.dS; gs.lda = T; gs.a_zstride = a.ds_zstride;
            gs.B = a.Z; gs.ldb = T; gs.b_zstride = a.z_zstride; gs.b_clen = T;
            gs.b_seg_len = d.R; gs.b_seg_stride = BRT; gs.tag = "dw_skip";
            gs.a_kmajor = 1; gs.b_kmajor = 1; gs.nlayer = 1; gs.nbatch = B; gs.ksplit = p.ksplit; gs.kchunk = p.kchunk;
            gs.C = a.Cskip; gs.ldc = gs.N; gs.c_zstride = (long)gs.M * gs.N;
            gs.a_rowsum = a.rs_skip;
            WN_TRY(wn_gemm6_dw_launch(&gs, 6, 0.0f, c.dw_ovf, c.st));
            WnGemmArgs gr = wn_gemm_default();
            gr.M = d.R; gr.N = d.R; gr.K = T;
            gr.A = a.dX; gr.lda = T; gr.a_zstride = a.dx_zstride; gr.a_lstride = BRT;
            gr.B = a.Z; gr.ldb = T; gr.b_zstride = a.z_zstride; gr.b_lstride = BRT; gr.b_clen = T; gr.tag = "dw_res";
            gr.a_kmajor = 1; gr.b_kmajor = 1; gr.nlayer = n_res; gr.nbatch = B; gr.ksplit = p.ksplit; gr.kchunk = p.kchunk;
            gr.C = a.Cres; gr.ldc = gr.N; gr.c_zstride = (long)gr.M * gr.N;
            gr.a_rowsum = a.rs_res;
            WN_TRY(wn_gemm6_dw_launch(&gr, 6, 0.0f, c.dw_ovf, c.st));
            DwOut o;
            o.out = grads + y.skip0 + (long)lo * y.ls_skip; o.m_seg = 0x7fffffff; o.m_seg_stride = 0; o.m_stride = d.R;
            o.n_seg = d.R; o.n_seg_stride = y.ls_skip; o.n_stride = 1;
            o.addend_m = nullptr; o.addend_scale_ptr = nullptr; o.rowsum_out = a.rs_skip ? ws + w.tmpS : nullptr;
            o.out_lstride = 0; o.addend_lstride = 0; o.rowsum_lstride = 0;
            WN_TRY(dw_reduce(c, a.Cskip, a.rs_skip, p.nz, d.S, nl * d.R, o, 1));
            if (a.rs_skip) {   // skip_1x1.l.bias = rowsum(dSkip) for every layer of the stack
                WnCopy4 cp;
                cp.n0 = 1; cp.n1 = 1; cp.n2 = d.S; cp.nl = d.L;
                cp.s0 = 0; cp.s1 = 0; cp.s2 = 1; cp.sl = 0;
                cp.d0 = 0; cp.d1 = 0; cp.d2 = 1; cp.dl = y.ls_skip;
                WN_TRY(wn_copy4(grads + y.skip0 + (long)d.S * d.R, ws + w.tmpS, &cp, c.st));
            }
            DwOut r = dw_out_plain(grads + lb_lo + y.o_res_w, d.R, grads + lb_lo + y.o_res_b);
            r.out_lstride = -y.LB; r.rowsum_lstride = -y.LB;
            WN_TRY(dw_reduce(c, a.Cres, a.rs_res, p.nz, d.R, d.R, r, n_res));
        } else {   // d res_1x1.l = dX_{l+1} . z_l^T ; the last layer's res_1x1 is dead -> zeros
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
                                               aux_fused ? ws + w.qp + (long)l * B * T : nullptr, B, T, t0,
                                               c.chain_f16 ? ws + w.amaxP + (long)l * w.amaxP_lfloats : nullptr, c.st));
            }
            if (l > 0) {  // dX_l from dP_l, and gate' of layer l-1 from it
                const long lbp = layer_base(y, d, l - 1);
                WN_TRY(wn_fused_bwd_chain(ws + w.wd_b + (long)l * d.K * 2 * d.R * d.R, dP, dXn, dXl, params + lbp + y.o_res_w,
                                          ws + w.dZs + (long)(l - 1) * d.R * T, zs_bstride, ws + w.Sg + (long)(l - 1) * BRT,
                                          ws + w.Z + (long)(l - 1) * BRT, gz, ws + w.P + (long)(l - 1) * P_L,
                                          ws + w.G + (long)(l - 1) * 2 * d.R * F, g_bstride, upw, Ue, F,
                                          aux_fused ? ws + w.dGp + (long)(l - 1) * B * 2 * d.R * (T / 16) : nullptr,
                                          aux_fused ? ws + w.qp + (long)(l - 1) * B * T : nullptr, B, T, d.K, dil,
                                          c.chain_f16 ? ws + w.img_taps16 + (long)l * (w.img_taps16_floats / d.L)
                                                      : ((w.img_floats > 0) ? ws + w.img_taps + (long)l * (wn_fused_image_floats(d.K, d.L, 1) / d.L) : nullptr),
                                          c.chain_f16 ? ws + w.img_res16 + (long)(l - 1) * (w.img_res16_floats / d.L)
                                                      : ((w.img_floats > 0) ? ws + w.img_res + (long)(l - 1) * (wn_fused_image_floats(d.K, d.L, 2) / d.L) : nullptr),
                                          t0, c.chain_f16 ? ws + w.amaxP + (long)l * w.amaxP_lfloats : nullptr,
                                          c.chain_f16 ? ws + w.amaxP + (long)(l - 1) * w.amaxP_lfloats : nullptr, c.st));
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
                    WN_TRY(fw_gemm(c, gs, nullptr, nullptr, 0, true));
                    gr.tag = "bwd_dz_res_gate";
                    WN_TRY(fw_gemm(c, gr, &ge, nullptr, 0, true));
                } else {
                    gs.tag = "bwd_dz_skip_gate";
                    WN_TRY(fw_gemm(c, gs, &ge, nullptr, 0, true));
                }
            } else {
                WN_TRY(fw_gemm(c, gs, nullptr, nullptr, 0, true));
                if (dXn) WN_TRY(fw_gemm(c, gr, nullptr, nullptr, 0, true));
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
                WN_TRY(fw_gemm(c, g, nullptr, nullptr, 0, true));
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

