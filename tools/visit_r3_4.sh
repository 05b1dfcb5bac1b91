#!/bin/bash
# round-3 visit 4: forward loss window + chain accumulating from zero (A/B vs weight-gradient waves), role-only timing builds
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_aux_modes.py -x -q -m gpu > $OUT/pytest_gpu_quick.txt 2>&1; echo "pytest quick rc=$?"; tail -3 $OUT/pytest_gpu_quick.txt
E=$ROOT/tools/exp
WN_AB_VARIANTS="WN_ENGINE_FLAGS=160 WN_X=1 WN_LIB_PATH=$E/libwn_dwonly.so WN_LIB_PATH=$E/libwn_chainonly.so WN_LIB_PATH=$E/libwn_dwprio0.so WN_LIB_PATH=$E/libwn_dwprio3.so" \
WN_ABK_KERNELS="fused_bwd_chain fused_bwd_chain_dw fused_resblock_fwd fwd_skip_sum fwd_post1 fwd_post2_ce dw_dilated dw_res" bash tools/gpu_visit.sh abk
timeout 600 python tools/grad_gap_probe.py > $OUT/grad_gap_probe.txt 2>&1; echo "gap probe rc=$?"; grep -v "worst per" $OUT/grad_gap_probe.txt | tail -7
