#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""TEST INFRASTRUCTURE.  Puts the reference's own model file where the GPU box can find it.

    python oracle/build_ref.py            # /root/reference/wavenet_vocoder/nets/wavenet.py -> oracle/_ref/wavenet.py

The reference is pure Python, so "building" it is a file copy: ``wavenet_vocoder/nets/wavenet.py`` imports nothing but
logging / sys / time / numpy / torch (wavenet.py:6-14) and is loadable as a single module.  ``oracle/_ref/`` is listed in
.gitignore -- the reference's source never enters this repository's history -- but not in .gpurunignore, so the copy travels to
the GPU box like the built .so files, where /root/reference does not exist.  ``bench.py``'s ``cpu_baseline`` leg then times THE
REFERENCE ITSELF (``kind: "reference"``) through ``oracle/ref_step.py``; without the copy it times the restatement
(``kind: "port"``).  Called by ``__graft_entry__.build()`` when /root/reference is present.
"""
import hashlib
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/wavenet_vocoder/nets/wavenet.py"
DST_DIR = os.path.join(HERE, "_ref")
DST = os.path.join(DST_DIR, "wavenet.py")


def build_ref(verbose=True):
    """Copy the reference model file into oracle/_ref/ (no-op without /root/reference).  Returns the path or None."""
    if not os.path.exists(SRC):
        if verbose:
            print("oracle/_ref: %s not present (GPU box: the copy made in the build container is used)" % SRC)
        return DST if os.path.exists(DST) else None
    os.makedirs(DST_DIR, exist_ok=True)
    shutil.copyfile(SRC, DST)
    with open(DST, "rb") as f:
        digest = hashlib.sha256(f.read()).hexdigest()
    with open(os.path.join(DST_DIR, "SOURCE.txt"), "w") as f:
        f.write("%s\nsha256 %s\n" % (SRC, digest))
    if verbose:
        print("oracle/_ref/wavenet.py <- %s (sha256 %s)" % (SRC, digest[:16]))
    return DST


if __name__ == "__main__":
    sys.exit(0 if build_ref() else 1)
