#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""HBM traffic per kernel launch from two rocprofv3 counter passes (tools/pmc_traffic.sh):

    python tools/pmc_traffic.py <dir of the --pmc FETCH_SIZE pass> <dir of the --pmc WRITE_SIZE pass> <steps> [bench log] [dir of
        the --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass] > pmc_traffic.json

MFMA utilisation of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): the busy counter is in cycles summed
over the chip's 1024 SIMDs (32 per v_mfma_f32_32x32x16_bf16: the chain kernel's 5 760 tiles x 240 MFMAs x 32 = 44 236 800 is
reproduced exactly), GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, counter notes).

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB; on gfx950 FETCH_SIZE tallies the 128-byte fabric
requests at 64 bytes (MI355X_MICROARCH.md, section HBM) -- calibrated on this code's own dword buffer accesses: the
fused forward kernel writes exactly 4 x B*R*T*4 = 188.7 MB (WRITE_SIZE 187.2 MB) and the fused gate kernel must read
329.7 MB (2 * FETCH_SIZE = 329.6 MB).  `steps` = number of training steps the profiled command ran (warm-up + timed +
profile steps), used for the whole-step total."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarize  # noqa: E402

# bench.py tag -> kernel symbol(s) that implement it (the first one present in the trace is reported)
# (prefixes: the first kernel of the trace that starts with one of them)
TAGS = {"fused_bwd_gate": ["void k_conv64s<2>", "void k_conv64s<0>"],
        "fused_resblock_fwd": ["void k_resblock_fwd_h<2", "void k_resblock_fwd_h<3", "void k_resblock_fwd_h<1",
                               "void k_resblock_fwd_s<2", "void k_resblock_fwd_s<3", "void k_resblock_fwd_s<1"],
        "fused_bwd_dx": ["void k_conv64s<1>"],
        "fused_bwd_chain": ["void k_chain64s<1, 2, false", "void k_chain64s<0, 2, false", "void k_chain64s<1, 3, false", "void k_chain64s<1, 1, false",
                            "void k_chain64s<0, 1, false", "void k_chain64s<1, 2>", "void k_chain64s<0, 2>"]}


def find_kernel(prefixes, *tables):
    for pre in prefixes:
        for tab in tables:
            for k in tab:
                if k.startswith(pre):
                    return k
    return None


def engine_flags_of(log_path):
    """The launch mode of the profiled run: bench.py prints its JSON line into the log; `config.engine_flags`."""
    try:
        for line in open(log_path):
            line = line.strip()
            if line.startswith("{") and '"metric"' in line:
                return int(json.loads(line)["config"]["engine_flags"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    fetch, write, steps = summarize([sys.argv[1]]), summarize([sys.argv[2]]), float(sys.argv[3])
    per, total, setup = {}, 0.0, 0.0
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, {}).get("FETCH_SIZE", 0.0)
        w = write.get(k, {}).get("WRITE_SIZE", 0.0)
        n = max(fetch.get(k, {}).get("launches", 0), write.get(k, {}).get("launches", 0))
        b = (2.0 * f + w) * 1024.0
        per[k[:80]] = b
        # torch's fill kernels are SET-UP (the zero-filled workspace / gradient / moment buffers, allocated once per run: the
        # engine zero-fills its 11.3 GB workspace since round 4), not traffic of a training step
        if k.startswith("void at::native::") and "FillFunctor" in k:
            setup += b * n
        else:
            total += b * n
    out = {}
    for tag, names in TAGS.items():
        name = find_kernel(names, fetch, write)
        if name is not None:
            out[tag] = {"kernel": name, "fetch_size_kb": fetch.get(name, {}).get("FETCH_SIZE"),
                        "write_size_kb": write.get(name, {}).get("WRITE_SIZE"),
                        "hbm_bytes_per_launch": (2.0 * fetch.get(name, {}).get("FETCH_SIZE", 0.0) +
                                                 write.get(name, {}).get("WRITE_SIZE", 0.0)) * 1024.0}
    if len(sys.argv) > 5:
        mf = summarize([sys.argv[5]])
        util = {}
        for k, v in mf.items():
            act = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            if act > 0:
                util[k[:80]] = {"mfma_util": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (act * 1024.0), "active_cycles": act,
                                "mfma_busy_cycles": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), "launches": v.get("launches", 0)}
        out["_mfma_util_per_kernel"] = util
        for tag, names in TAGS.items():
            name = find_kernel(names, util)
            if name is not None and tag in out:
                out[tag]["mfma_util"] = util[name]["mfma_util"]
    out["_engine_flags"] = engine_flags_of(sys.argv[4]) if len(sys.argv) > 4 else None
    # which build this was taken on: the digest of the kernel sources + flags the library was built from (the same one the
    # loader checks, pytorchwavenetvocoder_amd/csrc/build.py), and the commit the tree was at (tools/BUILD_COMMIT, written
    # before the snapshot is sent to the GPU box, which has no .git)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        sys.path.insert(0, root)
        from pytorchwavenetvocoder_amd.csrc import build as _b
        out["_source_digest"] = _b._digest()
    except Exception as e:  # noqa: BLE001
        out["_source_digest"] = "unavailable: %r" % (e,)
    try:
        out["_commit"] = open(os.path.join(root, "tools", "BUILD_COMMIT")).read().strip()
    except OSError:
        out["_commit"] = None
    out["_step_total_bytes"] = total / steps
    out["_setup_fill_bytes_excluded"] = setup
    out["_per_kernel_bytes_per_launch"] = per
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_traffic.sh); bytes = "
                    "(2*FETCH_SIZE + WRITE_SIZE)*1024, see tools/pmc_traffic.py; the step total includes the model "
                    "set-up kernels of the run divided by the number of steps")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
