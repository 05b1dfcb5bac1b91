#!/bin/bash
# round-3 visit 3: pipelined weight-gradient waves A/B; bucket split cost; chain-vs-pair bias probe
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_gpu_quick.txt 2>&1; echo "pytest quick rc=$?"; tail -3 $OUT/pytest_gpu_quick.txt
WN_AB_VARIANTS="WN_X=1 WN_ENGINE_FLAGS=160" \
WN_ABK_KERNELS="fused_bwd_chain fused_bwd_chain_dw fused_resblock_fwd dw_dilated dw_res dw_skip reduce_partials" bash tools/gpu_visit.sh abk
for lpb in 0 15 10; do
  for fl in 32 160; do
    WN_ENGINE_FLAGS=$fl timeout 200 python bench.py --repeats 3 --no-cpu-baseline --no-decode --profile-steps 0 --layers-per-bucket $lpb > $OUT/bench_ab.json 2>> $OUT/bench.err
    python -c "import json; d=json.load(open('$OUT/bench_ab.json')); print('flags $fl lpb $lpb: ms/step median %.3f min %.3f' % (d['ms_per_step'], d['ms_per_step_min']))"
  done
done | tee $OUT/lpb_probe.txt
timeout 600 python tools/chain_pair_diff.py > $OUT/chain_pair_diff.txt 2>&1; echo "diff probe rc=$?"; tail -36 $OUT/chain_pair_diff.txt
