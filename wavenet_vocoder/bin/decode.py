#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""``decode.py`` as the recipes call it (found on PATH through egs/*/path.sh:5): the command line of the reference's
wavenet_vocoder/bin/decode.py, served by pytorchwavenetvocoder_amd.bin.decode on the MI355X HIP path."""
import os
import sys

# run as a script from anywhere: the repository root (two levels up) carries both packages
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.realpath(__file__)))))

from pytorchwavenetvocoder_amd.bin.decode import *  # noqa: E402,F401,F403
from pytorchwavenetvocoder_amd.bin.decode import main  # noqa: E402

if __name__ == "__main__":
    main()
