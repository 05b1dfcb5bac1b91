#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""TEST INFRASTRUCTURE: compile the HIP kernel sources for the HOST (g++ -DWN_EMU, hip_emu.h).

Gives tests/ a way to execute the very same kernel source (index arithmetic, LDS layouts, MFMA
lane maps) without a GPU.  Never used by the product package.
"""
import fcntl
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "pytorchwavenetvocoder_amd", "csrc")
OUT = os.path.join(HERE, "_build")
SOURCES = ["wn_gemm.hip", "wn_gemm6.hip", "wn_elem.hip", "wn_fused.hip", "wn_decode.hip", "wn_dlp.hip", "wn_dlpm.hip", "wn_dlpf.hip", "wn_prof.hip", "wn_api.hip"]
LIB = os.path.join(OUT, "libwavenet_emu.so")
FLAGS = ["-O2", "-std=c++17", "-fPIC", "-DWN_EMU", "-Wno-psabi", "-mfma", "-ffp-contract=off", "-I", HERE, "-I", CSRC, "-x", "c++"]
# A/B build macros of the kernel sources (e.g. WN_EMU_EXTRA_FLAGS="-DWN_G6_FINE") get their own build directory
_EXTRA = os.environ.get("WN_EMU_EXTRA_FLAGS", "").split()
if _EXTRA:
    FLAGS = FLAGS[:4] + _EXTRA + FLAGS[4:]
    OUT = os.path.join(HERE, "_build", "variant_" + hashlib.sha256(" ".join(_EXTRA).encode()).hexdigest()[:10])
    LIB = os.path.join(OUT, "libwavenet_emu.so")


def _digest():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".inl"))]
    files += [os.path.join(HERE, "hip_emu.h"), os.path.join(ROOT, "include", "wavenet_hip.h"),
              os.path.join(ROOT, "include", "wavenet_hip_gemm.h"), __file__]
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "lock"), "w") as lk:   # pytest-xdist workers would otherwise build into the same objects
        fcntl.flock(lk, fcntl.LOCK_EX)
        return _build_locked(force)


def _build_locked(force):
    stamp = os.path.join(OUT, "stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    objs, procs = [], []
    for s in SOURCES:
        o = os.path.join(OUT, s.replace(".hip", ".o"))
        procs.append((s, subprocess.Popen(["g++"] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o])))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("g++ (emu) failed on %s" % s)
    subprocess.check_call(["g++", "-shared", "-fPIC", "-o", LIB] + objs)
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
