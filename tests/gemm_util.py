# -*- coding: utf-8 -*-
"""Op-level helper: call wn_op_gemm through the C-ABI (ctypes mirror of WnGemmArgs)."""
import ctypes

import torch

from pytorchwavenetvocoder_amd import _lib


def run_gemm(lib, device, **kw):
    g = _lib.WnGemmArgs.default()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            setattr(g, k, v.data_ptr())
        else:
            setattr(g, k, v)
    st = None
    if device != "cpu":
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.check(lib.wn_op_gemm(ctypes.byref(g), st), "wn_op_gemm")
    if device != "cpu":
        torch.cuda.synchronize()


def check_dw_type(lib, device, M, N, T, B, ksplit, seg_len=None, shift0=0, shift_step=0, onehot_Q=None, seed=0):
    """C[z] = sum_k A_b[m][k] * Bop_b[n][k]  (k = time) against a torch reference."""
    gen = torch.Generator().manual_seed(seed)
    A = torch.randn(B, M, T, generator=gen)
    kchunk = ((T + ksplit - 1) // ksplit + 31) // 32 * 32
    ksplit = (T + kchunk - 1) // kchunk
    nz = B * ksplit
    C = torch.zeros(nz, M, N)
    rows = torch.zeros(nz, M)
    if onehot_Q is None:
        seg_len = seg_len or 0x7fffffff
        nrows = N if seg_len > N else seg_len
        X = torch.randn(B, nrows, T, generator=gen)
        ref_B = torch.zeros(B, N, T)
        for n in range(N):
            seg, rr = (n // seg_len, n % seg_len) if seg_len <= N else (0, n)
            sh = shift0 + seg * shift_step
            for b in range(B):
                src = X[b, rr]
                if sh >= 0:
                    ref_B[b, n, sh:] = src[:T - sh] if sh < T else 0
                else:
                    ref_B[b, n, :T + sh] = src[-sh:]
        kw = dict(B=X.to(device), ldb=T, b_zstride=nrows * T, b_seg_len=seg_len, b_seg_stride=0)
    else:
        Q = onehot_Q
        idx = torch.randint(0, Q, (B, T), generator=gen)
        ref_B = torch.zeros(B, N, T)
        for n in range(N):
            seg, q = n // Q, n % Q
            sh = shift0 + seg * shift_step
            for b in range(B):
                hit = (idx[b] == q).float()
                if sh >= 0:
                    ref_B[b, n, sh:] = hit[:T - sh]
                else:
                    ref_B[b, n, :T + sh] = hit[-sh:]
        idx_d = idx.to(device)
        kw = dict(B=A.to(device), ldb=0, b_zstride=0, b_seg_len=Q, b_index=idx_d, b_index_zstride=T, b_index_mod=Q)
    Cd, rd, Ad = C.to(device), rows.to(device), A.to(device)
    run_gemm(lib, device, M=M, N=N, K=T, A=Ad, lda=T, a_zstride=M * T, a_kmajor=1, b_kmajor=1,
             b_shift0=shift0, b_shift_step=shift_step, b_clen=T, C=Cd, ldc=N, c_zstride=M * N,
             nbatch=B, ksplit=ksplit, kchunk=kchunk, a_rowsum=rd, **kw)
    got = Cd.cpu().sum(0)
    ref = torch.einsum("bmk,bnk->mn", A.double(), ref_B.double()).float()
    err = float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
    rerr = float((rd.cpu().sum(0) - A.sum(dim=(0, 2))).abs().max())
    return err, rerr
