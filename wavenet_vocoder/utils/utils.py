# -*- coding: utf-8 -*-
"""``wavenet_vocoder.utils.utils`` -> pytorchwavenetvocoder_amd.utils.utils."""
from pytorchwavenetvocoder_amd.utils.utils import *  # noqa: F401,F403
from pytorchwavenetvocoder_amd.utils.utils import __all__  # noqa: F401
