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
