#!/bin/bash
# round-3 visit 14: forward block with weight fragments requested one unit ahead (asm-pinned ds_reads) vs the committed build
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"
E=$ROOT/tools/exp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_aux_modes.py -x -q -m gpu > $OUT/pytest_gpu_quick.txt 2>&1; echo "pytest quick rc=$?"; tail -2 $OUT/pytest_gpu_quick.txt
WN_AB_VARIANTS="WN_X=1 WN_LIB_PATH=$E/libwn_base.so" WN_ABK_KERNELS="fused_resblock_fwd fused_bwd_chain" bash tools/gpu_visit.sh abk
