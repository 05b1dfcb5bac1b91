# -*- coding: utf-8 -*-
"""Op-level entry points of the C ABI and the stand-alone layer modules on the GPU against the oracle's restatement of
the reference layers: wn_op_front (OneHot + front CausalConv1d as a gather, wavenet.py:78-92,513-516), wn_op_causal_conv
/ ``CausalConv1d.forward`` (wavenet.py:95-121) and ``UpSampling.forward`` (wavenet.py:124-154, golden vector)."""
import os

import numpy as np
import pytest
import torch

from tests import ops_common as OC

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _lib():
    from pytorchwavenetvocoder_amd import _lib as L
    lib = L.load_library()
    assert not lib.is_emulator
    return lib


@pytest.mark.parametrize("Q,R,K,B,T", OC.FRONT_CASES)
def test_op_front_is_onehot_plus_causal_conv(Q, R, K, B, T):
    OC.check_op_front(_lib(), DEV, Q, R, K, B, T)


@pytest.mark.parametrize("Cin,Cout,K,d,B,T", OC.CONV_CASES)
def test_causal_conv_op_and_module(Cin, Cout, K, d, B, T):
    """y[t] = b + sum_k W[:,:,k] x[t - (K-1-k) d], zero history -- also with a dilation larger than the sequence."""
    from pytorchwavenetvocoder_amd.nets import CausalConv1d
    w, b, x, ref = OC.check_op_causal_conv(_lib(), DEV, Cin, Cout, K, d, B, T)
    m = CausalConv1d(Cin, Cout, K, d)
    with torch.no_grad():
        m.conv.weight.copy_(w)
        m.conv.bias.copy_(b)
    m.to(DEV)
    with torch.no_grad():
        ym = m(x.to(DEV))
    assert ym.grad_fn is None
    assert tuple(ym.shape) == (B, Cout, T) and float((ym.cpu() - ref).abs().max()) <= 1e-5
    with pytest.raises(Exception):
        m(x)        # CPU tensor: no fallback
    # the op-level backward, then the module under autograd: the reference's module is an ordinary differentiable nn.Module
    # (wavenet.py:95-121) -- x, weight and bias gradients against torch's own Conv1d + slice on the CPU
    OC.check_op_causal_conv_backward(_lib(), DEV, Cin, Cout, K, d, B, T)
    ref_conv = torch.nn.Conv1d(Cin, Cout, K, padding=(K - 1) * d, dilation=d)
    with torch.no_grad():
        ref_conv.weight.copy_(w)
        ref_conv.bias.copy_(b)
    xc = x.clone().requires_grad_(True)
    yr = ref_conv(xc)
    yr = yr[:, :, :-(K - 1) * d] if (K - 1) * d > 0 else yr       # wavenet.py:118-121
    gy = torch.randn(B, Cout, T)
    yr.backward(gy)
    xg = x.to(DEV).requires_grad_(True)
    yg = m(xg)
    assert yg.grad_fn is not None and float((yg.detach().cpu() - yr.detach()).abs().max()) <= 1e-5
    yg.backward(gy.to(DEV))
    for got, want, what in ((xg.grad, xc.grad, "dx"), (m.conv.weight.grad, ref_conv.weight.grad, "dw"), (m.conv.bias.grad, ref_conv.bias.grad, "db")):
        assert float((got.cpu() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), what


def test_upsampling_module_on_the_gpu_vs_reference_vector():
    from pytorchwavenetvocoder_amd.nets import UpSampling, initialize
    z = np.load(os.path.join(GOLDEN, "upsampling.npz"))
    up = UpSampling(z["w"].shape[-1])
    with torch.no_grad():
        up.conv.weight.copy_(torch.from_numpy(z["w"]))
        up.conv.bias.copy_(torch.from_numpy(z["b"]))
    up.to(DEV)
    y = up(torch.from_numpy(z["h"]).to(DEV))
    assert float((y.cpu() - torch.from_numpy(z["y"])).abs().max()) <= 1e-6
    # reference test/test_upsampling.py:13-20
    aux = torch.randn(1, 28, 1000)
    conv = UpSampling(10)
    conv.apply(initialize)
    conv.to(DEV)
    out = conv(aux.to(DEV)).cpu()
    assert out.shape[-1] == aux.shape[-1] * 10
    assert torch.equal(out, aux.repeat_interleave(10, dim=2))
    # with a gradient asked for the module stays differentiable (forward still the HIP op): x, weight and bias gradients
    # against torch's own ConvTranspose2d on the CPU (what the reference's module is, wavenet.py:141-154)
    up2 = UpSampling(8)
    ref = torch.nn.ConvTranspose2d(1, 1, kernel_size=(1, 8), stride=(1, 8))
    up2.conv.load_state_dict(ref.state_dict())
    up2.to(DEV)
    xc = torch.randn(2, 5, 7, requires_grad=True)
    xg = xc.detach().to(DEV).requires_grad_(True)
    gy = torch.randn(2, 5, 56)
    yr = ref(xc.unsqueeze(1)).squeeze(1)
    yr.backward(gy)
    yg = up2(xg)
    assert yg.grad_fn is not None and float((yg.detach().cpu() - yr.detach()).abs().max()) <= 1e-6
    yg.backward(gy.to(DEV))
    assert float((xg.grad.cpu() - xc.grad).abs().max()) <= 1e-5
    assert float((up2.conv.weight.grad.cpu() - ref.weight.grad).abs().max()) <= 1e-4
    assert float((up2.conv.bias.grad.cpu() - ref.bias.grad).abs().max()) <= 1e-4
    with torch.no_grad():
        assert up2(xg).grad_fn is None


def test_transpose_op_and_the_autograd_bridge_with_an_external_loss():
    """wn_op_transpose_last2 ((B, R, C) -> (B, C, R): bit-exact, ragged tiles) and what it is for: ``model(x, h)`` under an
    EXTERNAL loss (the reference's own training step, train.py:534-538 -- nn.CrossEntropyLoss on the (B, T, Q) logits) hands a
    (B, T, Q) gradient back through autograd; the bridge moves it into the kernels' (B, Q, T) layout with this op."""
    lib = _lib()
    import ctypes
    for B, R, C in ((2, 37, 70), (1, 64, 32), (3, 5, 257)):
        x = torch.randn(B, R, C, device=DEV)
        y = torch.empty(B, C, R, device=DEV)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib.check(lib.wn_op_transpose_last2(x.data_ptr(), y.data_ptr(), B, R, C, st), "wn_op_transpose_last2")
        assert torch.equal(y, x.transpose(1, 2).contiguous())
    from oracle import wavenet_oracle as O
    from pytorchwavenetvocoder_amd.nets import WaveNet
    cfg_t = (32, 4, 32, 32, 3, 1, 2, 4)
    cfg = O.OracleConfig(*cfg_t)
    params = O.random_params(cfg, 7, scale=0.2)
    xb, hb, tb = O.synthetic_batch(cfg, 2, 48, 8)
    model = WaveNet(*cfg_t)
    model.load_state_dict(params)
    model.to(DEV)
    out = model(xb.to(DEV), hb.to(DEV))                       # (B, T, Q)
    loss = torch.nn.functional.cross_entropy(out[:, cfg.receptive_field:].reshape(-1, cfg.n_quantize),
                                             tb[:, cfg.receptive_field:].reshape(-1).to(DEV))
    loss.backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = torch.nn.functional.cross_entropy(O.forward(cfg, leaves, xb, hb)[:, cfg.receptive_field:].reshape(-1, cfg.n_quantize),
                                            tb[:, cfg.receptive_field:].reshape(-1))
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5
    for k, p in model.named_parameters():
        g = leaves[k].grad
        if p.grad is None:
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        assert float((p.grad.cpu() - g).abs().max()) <= 1e-4 * max(float(g.abs().max()), 1e-12) + 1e-7, k
