#!/bin/bash
# round-3 visit 7: non-temporal read-once loads (now default) vs off, and further candidates
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"
E=$ROOT/tools/exp
WN_AB_VARIANTS="WN_X=1 WN_LIB_PATH=$E/libwn_ntbwdst.so WN_LIB_PATH=$E/libwn_ntfwdst.so WN_LIB_PATH=$E/libwn_ntfwdst2.so WN_LIB_PATH=$E/libwn_ntallst.so" \
WN_ABK_KERNELS="fused_bwd_chain fused_resblock_fwd dw_dilated dw_res dw_skip fwd_skip_sum bwd_dz_skip_all fwd_post1 bwd_post1_dx bwd_post2_dx dw_post1" bash tools/gpu_visit.sh abk
