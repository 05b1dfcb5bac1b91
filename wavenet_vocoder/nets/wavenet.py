# -*- coding: utf-8 -*-
"""``wavenet_vocoder.nets.wavenet`` -> pytorchwavenetvocoder_amd.nets.wavenet (the HIP-backed WaveNet)."""
from pytorchwavenetvocoder_amd.nets.wavenet import *  # noqa: F401,F403
from pytorchwavenetvocoder_amd.nets.wavenet import (CausalConv1d, OneHot, UpSampling, WaveNet, decode_mu_law,  # noqa: F401
                                                    encode_mu_law, initialize)
