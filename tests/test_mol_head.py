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
