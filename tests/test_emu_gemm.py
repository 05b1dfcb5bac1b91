# -*- coding: utf-8 -*-
"""Op-level checks of the generic f32-MFMA contraction (csrc/wn_gemm.hip) through wn_op_gemm, kernel
sources under the host emulator: dW-type (k = time) with split-K, segments + shifts (16-byte and
scalar staging paths), one-hot operand, row sums."""
import pytest

from tests.emu_util import emu_library
from tests.gemm_util import check_dw_type

pytestmark = pytest.mark.emu


@pytest.mark.parametrize("kw", [
    dict(M=64, N=64, T=1000, B=2, ksplit=4),                                        # aligned -> 16-byte loads
    dict(M=128, N=128, T=1000, B=1, ksplit=3, seg_len=64, shift0=4, shift_step=-4),  # two taps, shift 4 / 0
    dict(M=128, N=128, T=600, B=1, ksplit=2, seg_len=64, shift0=1, shift_step=-1),   # shift 1 -> scalar path
    dict(M=70, N=50, T=333, B=2, ksplit=2),                                          # ragged everything
    dict(M=64, N=512, T=1500, B=1, ksplit=6, onehot_Q=256, shift0=1, shift_step=-1),
    dict(M=64, N=74, T=700, B=2, ksplit=3, onehot_Q=37, shift0=1, shift_step=-1),
])
def test_dw_type(kw):
    err, rerr = check_dw_type(emu_library(), "cpu", **kw)
    assert err <= 2e-6, err
    assert rerr <= 1e-4, rerr
