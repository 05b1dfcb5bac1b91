# -*- coding: utf-8 -*-
"""Import-path alias: ``wavenet_vocoder`` == the MI355X-native package ``pytorchwavenetvocoder_amd``.

The reference's own callers use this name -- ``from wavenet_vocoder.nets import WaveNet`` (reference
wavenet_vocoder/bin/train.py:25-27, decode.py:22-27, test/test_wavenet.py:11-13) and the recipes put
``$PRJ_ROOT/wavenet_vocoder/bin`` and ``$PRJ_ROOT/wavenet_vocoder/utils`` on PATH (egs/*/path.sh:5) to find
``train.py`` / ``decode.py`` / ``run.pl`` / ``parse_options.sh`` -- so a checkout of this repository can stand
where the reference's checkout stood without an import swap.  Nothing is implemented here: every module
re-exports the product package, whose compute path is the gfx950 library behind include/wavenet_hip.h.
"""
