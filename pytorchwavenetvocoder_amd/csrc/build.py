#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Build libwavenet_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python pytorchwavenetvocoder_amd/csrc/build.py [--force]

The library is built IN-TREE next to the sources (it is git-ignored but travels to the GPU box).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["wn_gemm.hip", "wn_gemm6.hip", "wn_elem.hip", "wn_fused.hip", "wn_decode.hip", "wn_dlp.hip", "wn_dlpm.hip", "wn_dlpf.hip", "wn_prof.hip", "wn_api.hip"]
HEADERS = ["wn_api_backward.inl", "wn_api_ops.inl", "wn_api_decode.inl", "wn_device.h", "wn_gemm.h", "wn_gemm6.h", "wn_elem.h", "wn_fused.h", "wn_decode.h", "wn_dlp.h", "wn_prof.h", "../../include/wavenet_hip.h", "../../include/wavenet_hip_gemm.h"]
LIB = os.path.join(HERE, "libwavenet_hip.so")
STAMP = os.path.join(HERE, ".libwavenet_hip.stamp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc otherwise packs adjacent scalar f32 adds / multiplies into v_pk_* instructions, which issue slower
# than their scalar halves beside MFMAs (MI355X_MICROARCH.md; 1 778 of them in wn_gemm6, 1 485 in wn_fused): same box 10.08 ->
# 9.98 ms per step alone, 9.90 -> 9.85 on top of the k_gemm6 interleave (profiles/r04/ab_gemm6_fine_noslp.txt)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize"]


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS + ["build.py"]:
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def up_to_date():
    """True when the .so next to the sources was built from exactly these sources and flags."""
    try:
        return os.path.exists(LIB) and open(STAMP).read().strip() == _digest()
    except OSError:
        return False


def build(force=False, verbose=True):
    dig = _digest()
    if not force and up_to_date():
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError("hipcc not found at %s; cannot build the gfx950 library" % HIPCC)
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, s.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(HERE, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, cwd=HERE)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % s)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=HERE)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
