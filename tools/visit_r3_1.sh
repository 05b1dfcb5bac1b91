#!/bin/bash
# round-3 visit 1: baseline bench on this box + 1-wave-per-SIMD / co-residency probes of the fused kernels
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
WN_AB_VARIANTS="WN_X=1 WN_LIB_PATH=$ROOT/tools/exp/libwn_ft256.so WN_LIB_PATH=$ROOT/tools/exp/libwn_ft256lb.so" \
WN_ABK_KERNELS="fused_bwd_chain fused_resblock_fwd fused_bwd_gate fused_bwd_dx dw_dilated dw_res dw_skip" bash tools/gpu_visit.sh abk
WN_AB_VARIANTS="WN_X=1 WN_ENGINE_FLAGS=1316 WN_ENGINE_FLAGS=292 WN_LIB_PATH=$ROOT/tools/exp/libwn_ft256lb.so WN_LIB_PATH=$ROOT/tools/exp/libwn_ft256lb.so,WN_ENGINE_FLAGS=1316 WN_LIB_PATH=$ROOT/tools/exp/libwn_ft256lb.so,WN_ENGINE_FLAGS=292 WN_LIB_PATH=$ROOT/tools/exp/libwn_ft256lb.so,WN_ENGINE_FLAGS=7716" bash tools/gpu_visit.sh lpb
