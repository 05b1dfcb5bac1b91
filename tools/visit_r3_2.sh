#!/bin/bash
# round-3 visit 2: chain launches with weight-gradient waves (default) vs WN_FLAG_NO_CHAIN_DW (160 = AUX_FUSED | NO_CHAIN_DW)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_aux_modes.py -x -q -m gpu > $OUT/pytest_gpu_quick.txt 2>&1; echo "pytest quick rc=$?"; tail -3 $OUT/pytest_gpu_quick.txt
WN_AB_VARIANTS="WN_X=1 WN_ENGINE_FLAGS=160" \
WN_ABK_KERNELS="fused_bwd_chain fused_bwd_chain_dw fused_resblock_fwd dw_dilated dw_res dw_skip reduce" bash tools/gpu_visit.sh abk
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "cfg2_full_size_vs_oracle or config4_stated or recipe_size_model_at" > $OUT/pytest_gpu_fullsize.txt 2>&1; echo "pytest fullsize rc=$?"; tail -3 $OUT/pytest_gpu_fullsize.txt; grep -h "vs oracle" $OUT/pytest_gpu_fullsize.txt
timeout 600 python tools/grad_gap_probe.py > $OUT/grad_gap_probe.txt 2>&1; echo "gap probe rc=$?"; cat $OUT/grad_gap_probe.txt | tail -12
