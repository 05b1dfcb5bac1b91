// wn_api_ops.inl: op-level entry points (stand-alone layer modules, parity tests) -- part of the ONE translation unit wn_api.hip (included at its end: the entry points share its file-local
// helpers -- error text, parameter layout, workspace carving, launch contexts).  Not compiled on its own.
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

// Backward of CausalConv1d (autograd of wavenet.py:95-121) as three contractions of the same exact-f32 kernel family:
//   dx[b,i,t]  = sum_k sum_o W[o,i,k] dy[b,o,t + (K-1-k)d]            (taps transposed: dy read AHEAD, zero beyond T)
//   dW[o,i,k]  = sum_b sum_t dy[b,o,t] x[b,i,t - (K-1-k)d]            (k = time; per-(b, k-chunk) partials, fixed-order sum)
//   db[o]      = sum_b sum_t dy[b,o,t]                                 (row sums of the same launch)
// Any of dx / dw / db may be NULL.  scratch: wn_op_causal_conv_backward_scratch_floats() floats.
static void ccb_plan(int B, int T, int* ksplit, int* kchunk) {
    int ks = (T + 4095) / 4096;
    if (ks * B > 512) ks = 512 / B;
    if (ks < 1) ks = 1;
    int kc = ((T + ks - 1) / ks + 31) / 32 * 32;
    if (kc < 32) kc = 32;
    *kchunk = kc;
    *ksplit = (T + kc - 1) / kc;
}

extern "C" long wn_op_causal_conv_backward_scratch_floats(int B, int T, int Cin, int Cout, int K) {
    if (B < 1 || T < 1 || Cin < 1 || Cout < 1 || K < 1) return 0;
    int ks, kc;
    ccb_plan(B, T, &ks, &kc);
    const long nz = (long)B * ks;
    return (long)K * Cin * Cout + nz * Cout * K * Cin + nz * Cout + 64;
}

extern "C" int wn_op_causal_conv_backward(const float* weight, const float* x, const float* dy, float* dx, float* dw, float* db,
                                          float* scratch, int B, int T, int Cin, int Cout, int K, int dilation, void* stream) {
    api_enter();
    if (!weight || !x || !dy || !scratch || B < 1 || T < 1 || Cin < 1 || Cout < 1 || K < 1 || dilation < 1) return fail(1, "bad argument");
    wn_stream_t st = (wn_stream_t)stream;
    if (dx) {
        WnCopy4 cp;  // scratch[(tap*Cout + o)*Cin + i] = W[o][i][tap]
        cp.n0 = K; cp.n1 = Cout; cp.n2 = Cin; cp.nl = 1;
        cp.s0 = 1; cp.s1 = (long)Cin * K; cp.s2 = K; cp.sl = 0;
        cp.d0 = (long)Cout * Cin; cp.d1 = Cin; cp.d2 = 1; cp.dl = 0;
        WN_TRY(wn_copy4(scratch, weight, &cp, st));
        WnGemmArgs g = wn_gemm_default();
        g.M = Cin; g.N = T; g.K = K * Cout;
        g.A = scratch; g.lda = Cin;
        g.B = dy; g.ldb = T; g.b_zstride = (long)Cout * T; g.b_clen = T;
        g.b_seg_len = Cout; g.b_seg_stride = 0; g.b_shift0 = -(K - 1) * dilation; g.b_shift_step = dilation;
        g.C = dx; g.ldc = T; g.c_zstride = (long)Cin * T;
        g.nbatch = B; g.tag = "op_causal_conv_dx";
        WN_TRY(wn_gemm_launch(&g, st));
    }
    if (dw || db) {
        int ks, kc;
        ccb_plan(B, T, &ks, &kc);
        const int nz = B * ks;
        const int N = K * Cin;
        float* partial = scratch + (long)K * Cin * Cout;
        float* rs_partial = partial + (long)nz * Cout * N;
        WnGemmArgs g = wn_gemm_default();
        g.M = Cout; g.N = N; g.K = T;
        g.A = dy; g.lda = T; g.a_zstride = (long)Cout * T; g.a_kmajor = 1;
        g.B = x; g.ldb = T; g.b_zstride = (long)Cin * T; g.b_kmajor = 1; g.b_clen = T;
        g.b_seg_len = Cin; g.b_seg_stride = 0; g.b_shift0 = (K - 1) * dilation; g.b_shift_step = -dilation;
        g.C = partial; g.ldc = N; g.c_zstride = (long)Cout * N;
        g.nbatch = B; g.ksplit = ks; g.kchunk = kc;
        g.a_rowsum = db ? rs_partial : nullptr;
        g.tag = "op_causal_conv_dw";
        WN_TRY(wn_gemm_launch(&g, st));
        if (dw) {   // dW[o][i][tap] = sum_z partial[z][o][tap*Cin + i]
            WnReduceArgs r;
            r.partial = partial; r.nz = nz; r.M = Cout; r.N = N;
            r.out = dw; r.m_seg = 0x7fffffff; r.n_seg = Cin;
            r.m_seg_stride = 0; r.m_stride = (long)Cin * K; r.n_seg_stride = 1; r.n_stride = K;
            r.scale = 1.0f; r.accumulate = 0; r.addend_m = nullptr; r.addend_scale_ptr = nullptr;
            r.scratch = nullptr; r.scratch_floats = 0; r.nl = 1; r.out_lstride = 0; r.addend_lstride = 0;
            WN_TRY(wn_reduce(&r, st));
        }
        if (db) {
            WnReduceArgs q;
            q.partial = rs_partial; q.nz = nz; q.M = Cout; q.N = 1;
            q.out = db; q.m_seg = 0x7fffffff; q.n_seg = 0x7fffffff;
            q.m_seg_stride = 0; q.m_stride = 1; q.n_seg_stride = 0; q.n_stride = 0;
            q.scale = 1.0f; q.accumulate = 0; q.addend_m = nullptr; q.addend_scale_ptr = nullptr;
            q.scratch = nullptr; q.scratch_floats = 0; q.nl = 1; q.out_lstride = 0; q.addend_lstride = 0;
            WN_TRY(wn_reduce(&q, st));
        }
    }
    return rt_check("wn_op_causal_conv_backward");
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

extern "C" int wn_op_gemm(const struct WnGemmArgs* args, void* stream) {
    api_enter();
    WN_TRY(wn_gemm_launch(args, (wn_stream_t)stream));
    return rt_check("wn_op_gemm");
}
