# -*- coding: utf-8 -*-
"""Mixture-of-logistics head: kernel vs this repo's CPU restatement (the reference has no such head)."""
import pytest

from tests import mol_common as MC
from tests.emu_util import emu_library


@pytest.mark.emu
def test_mol_loss_op_emulator():
    MC.check_mol_loss_op(emu_library(), "cpu")


@pytest.mark.emu
def test_mol_training_step_emulator():
    MC.check_mol_training_step(emu_library(), "cpu")


@pytest.mark.emu
def test_mol_generation_emulator():
    MC.check_mol_generation(emu_library(), "cpu")


def _gpu_lib():
    from pytorchwavenetvocoder_amd import _lib
    return _lib.load_library()


@pytest.mark.gpu
def test_mol_loss_op_gpu():
    MC.check_mol_loss_op(_gpu_lib(), "cuda:0")


@pytest.mark.gpu
def test_mol_training_step_gpu():
    MC.check_mol_training_step(_gpu_lib(), "cuda:0")


@pytest.mark.gpu
def test_mol_generation_gpu():
    MC.check_mol_generation(_gpu_lib(), "cuda:0")


@pytest.mark.emu
def test_log_scale_min_is_part_of_the_model_configuration():
    """The clamp of the mixture log-scales is a constructor argument: train.py stores it in model.conf, decode.py rebuilds
    the model with it (round-2 advisor finding: a model trained with a non-default clamp sampled with -7 after a reload);
    state_dict keeps the reference's parameter keys."""
    import argparse
    from pytorchwavenetvocoder_amd.bin import decode as D, train as T
    from pytorchwavenetvocoder_amd.nets import WaveNet
    a = T.get_parser().parse_args(["--waveforms", "w", "--feats", "f", "--stats", "s", "--expdir", "e",
                                   "--n_mixture", "2", "--log_scale_min", "-5.5"])
    assert a.log_scale_min == -5.5
    conf = argparse.Namespace(n_quantize=16, n_aux=3, n_resch=8, n_skipch=8, dilation_depth=2, dilation_repeat=1,
                              kernel_size=2, upsampling_factor=0, use_upsampling_layer=False, n_mixture=2,
                              log_scale_min=-5.5)
    import inspect
    assert "log_scale_min" in inspect.signature(WaveNet.__init__).parameters
    assert "log_scale_min=getattr(config" in inspect.getsource(D.build_model)
    m = WaveNet(16, 3, 8, 8, 2, 1, 2, 0, n_mixture=2, _library=emu_library(), log_scale_min=conf.log_scale_min)
    assert m.log_scale_min == -5.5 and not any("log_scale" in k for k in m.state_dict())


def test_mol_vs_restatement_harness_emulator():
    """The harness of the stated-size GPU test (tests/test_gpu_fullsize.py::test_config4_mol_head_stated_size...) on a toy
    geometry with kernel_size 3 and an upsampling layer, on the emulator."""
    from tests import parity_common as PC
    r = PC.run_mol_vs_restatement((32, 4, 8, 12, 3, 2, 3, 4), 4, 2, 64, 31, emu_library(), "cpu", scale=0.3, threads=4)
    print(r)
    assert r["grad_key"] is not None
