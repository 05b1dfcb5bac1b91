# -*- coding: utf-8 -*-
"""Op-level C-ABI checks shared by the CPU (kernel emulator) and GPU test files: wn_op_front and wn_op_causal_conv against
the oracle's restatement of OneHot + CausalConv1d (reference wavenet.py:78-92, 95-121, 513-516)."""
import ctypes

import numpy as np
import torch

from oracle import wavenet_oracle as O

# the last five have B * T >= 16384: the LDS-table variant of the gather (k_front_gather_lds), incl. a ragged last chunk,
# tables that do not fit one CU's LDS and are cut into groups of output rows (Q = 256, R = 64, K = 3: two groups of 32;
# R = 128: two of 64; R = 96: three of 32) and negative / out-of-range indices
FRONT_CASES = [(256, 64, 2, 2, 1000), (256, 64, 3, 1, 333), (37, 12, 2, 3, 77), (256, 512, 2, 1, 257),
               (256, 64, 2, 3, 5501), (64, 32, 3, 2, 8200), (256, 64, 3, 2, 8200), (256, 128, 2, 2, 8200), (256, 96, 2, 2, 8200)]
CONV_CASES = [(64, 64, 2, 1, 2, 500), (64, 128, 2, 16, 1, 300), (64, 64, 3, 4, 2, 257), (12, 20, 3, 7, 3, 91),
              (64, 64, 2, 512, 1, 300), (256, 64, 2, 1, 1, 200)]


def _stream(device):
    device = torch.device(device)
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream) if device.type == "cuda" else None


def check_op_front(lib, device, Q, R, K, B, T):
    rs = np.random.RandomState(Q + R + K)
    w = torch.from_numpy(rs.standard_normal((R, Q, K)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(R).astype(np.float32))
    x = torch.from_numpy(rs.randint(-Q, 3 * Q, (B, T)))       # values outside [0, Q): taken modulo Q (wavenet.py:88)
    ref = O.causal_conv1d(O.onehot(x, Q, torch.float32).transpose(1, 2), w, b, 1)
    out = torch.empty((B, R, T), dtype=torch.float32, device=device)
    scratch = torch.empty(K * Q * R, dtype=torch.float32, device=device)
    wd, bd, xd = w.to(device), b.to(device), x.to(device)
    rc = lib.wn_op_front(wd.data_ptr(), bd.data_ptr(), xd.data_ptr(), out.data_ptr(), scratch.data_ptr(), B, T, Q, R, K,
                         _stream(device))
    lib.check(rc, "wn_op_front")
    assert float((out.cpu() - ref).abs().max()) <= 1e-6   # a gather + K-1 adds


def check_op_causal_conv(lib, device, Cin, Cout, K, d, B, T):
    rs = np.random.RandomState(Cin + Cout + K + d)
    w = torch.from_numpy((rs.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(Cout).astype(np.float32))
    x = torch.from_numpy(rs.standard_normal((B, Cin, T)).astype(np.float32))
    ref = O.causal_conv1d(x, w, b, d)
    assert tuple(ref.shape) == (B, Cout, T)
    y = torch.empty((B, Cout, T), dtype=torch.float32, device=device)
    scratch = torch.empty(w.numel(), dtype=torch.float32, device=device)
    wd, bd, xd = w.to(device), b.to(device), x.to(device)
    rc = lib.wn_op_causal_conv(wd.data_ptr(), bd.data_ptr(), xd.data_ptr(), y.data_ptr(), scratch.data_ptr(), B, T, Cin, Cout,
                               K, d, _stream(device))
    lib.check(rc, "wn_op_causal_conv")
    assert float((y.cpu() - ref).abs().max()) <= 1e-5
    return w, b, x, ref


def check_op_causal_conv_backward(lib, device, Cin, Cout, K, d, B, T):
    """wn_op_causal_conv_backward against autograd of the oracle's restatement of CausalConv1d (reference wavenet.py:95-121:
    nn.Conv1d with padding (K-1)d, last (K-1)d outputs dropped): dx, dW, db, each also with the other outputs NULL."""
    rs = np.random.RandomState(Cin + Cout + K + d + 1)
    w = torch.from_numpy((rs.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)).requires_grad_(True)
    b = torch.from_numpy(rs.standard_normal(Cout).astype(np.float32)).requires_grad_(True)
    x = torch.from_numpy(rs.standard_normal((B, Cin, T)).astype(np.float32)).requires_grad_(True)
    dy = torch.from_numpy(rs.standard_normal((B, Cout, T)).astype(np.float32))
    O.causal_conv1d(x, w, b, d).backward(dy)
    n = lib.wn_op_causal_conv_backward_scratch_floats(B, T, Cin, Cout, K)
    assert n > 0
    scratch = torch.empty(n, dtype=torch.float32, device=device)
    wd, xd, dyd = w.detach().to(device), x.detach().to(device), dy.to(device)
    dx = torch.full((B, Cin, T), float("nan"), dtype=torch.float32, device=device)
    dw = torch.full((Cout, Cin, K), float("nan"), dtype=torch.float32, device=device)
    db = torch.full((Cout,), float("nan"), dtype=torch.float32, device=device)
    rc = lib.wn_op_causal_conv_backward(wd.data_ptr(), xd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                        scratch.data_ptr(), B, T, Cin, Cout, K, d, _stream(device))
    lib.check(rc, "wn_op_causal_conv_backward")
    for got, ref, what in ((dx, x.grad, "dx"), (dw, w.grad, "dw"), (db, b.grad, "db")):
        err = float((got.cpu() - ref).abs().max())
        assert err <= 1e-5 * max(1.0, float(ref.abs().max())), (what, err)
    # dx alone / dw alone (the NULL outputs are skipped)
    dx2 = torch.empty_like(dx)
    rc = lib.wn_op_causal_conv_backward(wd.data_ptr(), xd.data_ptr(), dyd.data_ptr(), dx2.data_ptr(), None, None,
                                        scratch.data_ptr(), B, T, Cin, Cout, K, d, _stream(device))
    lib.check(rc, "wn_op_causal_conv_backward")
    assert torch.equal(dx2, dx)
    dw2 = torch.empty_like(dw)
    rc = lib.wn_op_causal_conv_backward(wd.data_ptr(), xd.data_ptr(), dyd.data_ptr(), None, dw2.data_ptr(), None,
                                        scratch.data_ptr(), B, T, Cin, Cout, K, d, _stream(device))
    lib.check(rc, "wn_op_causal_conv_backward")
    assert torch.equal(dw2, dw)
