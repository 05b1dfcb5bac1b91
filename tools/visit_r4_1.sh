#!/bin/bash
# Round-4 visit 1: hand-off micro-benchmark, A/B of the k_gemm6 fine interleave / no-SLP builds (headline + recipe size),
# wide-path drift probe (default build and alternating-sign tiles), configs[3] geometry PMC pass, the new stated-size tests.
#   gpurun --timeout 1500 -- 'bash tools/visit_r4_1.sh [parts]'     parts: micro ab recipe drift gap pmc3 tests
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"
PARTS="${*:-micro ab recipe drift gap pmc3 tests}"
date +%s > $OUT/t0
has() { echo " $PARTS " | grep -q " $1 "; }
el() { echo "[elapsed $(( $(date +%s) - $(cat $OUT/t0) )) s] $*"; }
if has micro; then
  timeout 120 tools/exp/handoff > $OUT/handoff.txt 2>&1; el "handoff rc=$?"; cat $OUT/handoff.txt
fi
if has ab; then
  WN_AB_VARIANTS="WN_X=1 WN_LIB_PATH=tools/exp/libwn_fine.so WN_LIB_PATH=tools/exp/libwn_noslp.so WN_LIB_PATH=tools/exp/libwn_finenoslp.so" \
  WN_ABK_KERNELS="fwd_skip_sum bwd_dz_skip_all dw_skip bwd_post2_dx bwd_post1_dx fwd_post1 fwd_post2_ce dw_dilated dw_res fused_bwd_chain fused_resblock_fwd" \
    bash tools/gpu_visit.sh abk > $OUT/abk_visit.txt 2>&1; el "abk done"; cat $OUT/abk_probe.txt
fi
if has recipe; then
  for v in base fine noslp finenoslp; do
    if [ $v = base ]; then L=""; else L="tools/exp/libwn_$v.so"; fi
    WN_LIB_PATH=$L timeout 200 python tools/recipe_bench.py --steps 3 > $OUT/recipe_$v.json 2> $OUT/recipe_$v.err
    WN_LIB_PATH=$L timeout 200 python tools/recipe_bench.py --resch 64 --kernel-size 3 --upsampling 256 --T 26112 --batch 8 --steps 10 > $OUT/config4_$v.json 2> $OUT/config4_$v.err
    python - <<P
import json
for name in ("recipe_$v", "config4_$v"):
    try:
        d = json.load(open("$OUT/%s.json" % name))
    except Exception as e:
        print(name, "FAILED", e); continue
    ks = d["kernels"]
    print("%-18s %8.3f ms/step | " % (name, d["ms_per_step"]) + " ".join("%s %.2f" % (k, v["ms"]) for k, v in list(ks.items())[:9]))
P
  done | tee $OUT/recipe_ab.txt
  el "recipe A/B done"
fi
if has drift; then
  timeout 400 python tools/wide_drift_probe.py > $OUT/wide_drift_base.txt 2>&1; el "drift base rc=$?"; tail -12 $OUT/wide_drift_base.txt
  WN_LIB_PATH=tools/exp/libwn_flip.so timeout 400 python tools/wide_drift_probe.py > $OUT/wide_drift_flip.txt 2>&1; el "drift flip rc=$?"; tail -12 $OUT/wide_drift_flip.txt
fi
if has gap; then
  timeout 300 python tools/grad_gap_probe.py 2 6656 3 256 6 > $OUT/grad_gap_config4_b2.txt 2>&1; el "gap rc=$?"; cat $OUT/grad_gap_config4_b2.txt
fi
if has pmc3; then
  WN_PMC_NAME=config4 WN_PMC_STEPS=4 WN_PMC_CMD="python tools/recipe_bench.py --resch 64 --kernel-size 3 --upsampling 256 --T 26112 --batch 8 --steps 1" \
    bash tools/pmc_traffic.sh > $OUT/pmc_config4.txt 2>&1; el "pmc3 done"; tail -25 $OUT/pmc_config4.txt
  find $OUT/pmc_config4 -name "*.db" -delete 2>/dev/null
fi
if has tests; then
  timeout 900 python -m pytest tests -q -m gpu -s -k "mol_head_stated or timed_size or test_gpu_parity or test_gpu_ops or rccl or api" > $OUT/pytest_gpu_sel.txt 2>&1; el "pytest(selected) rc=$?"; tail -5 $OUT/pytest_gpu_sel.txt
  grep -h "vs oracle\|vs own\|err \|STATED\|TIMED" $OUT/pytest_gpu_sel.txt | head -20
fi
lscpu | grep -E "Model name|^CPU\(s\)|Socket" > $OUT/host.txt
el "end"
