#!/bin/bash
# round-3 visit 6: reversed tile walks / non-temporal read-once loads (variant builds), configs[3] after the dW / front fixes
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"
E=$ROOT/tools/exp
WN_LIB_PATH=$E/libwn_revnt.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_gpu_revnt.txt 2>&1; echo "pytest (rev+nt variant) rc=$?"; tail -2 $OUT/pytest_gpu_revnt.txt
WN_AB_VARIANTS="WN_X=1 WN_LIB_PATH=$E/libwn_rev.so WN_LIB_PATH=$E/libwn_nt.so WN_LIB_PATH=$E/libwn_revnt.so" \
WN_ABK_KERNELS="fused_bwd_chain fused_resblock_fwd fused_bwd_gate fused_bwd_dx dw_dilated dw_res" bash tools/gpu_visit.sh abk
bash tools/gpu_visit.sh recipesize
