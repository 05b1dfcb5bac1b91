# -*- coding: utf-8 -*-
"""MI355X-native WaveNet-vocoder training path (drop-in for wavenet_vocoder.nets + bin/train.py).

    from pytorchwavenetvocoder_amd.nets import WaveNet, initialize, encode_mu_law
"""
__version__ = "0.1.0"
