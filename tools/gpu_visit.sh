#!/bin/bash
# One GPU visit: parity tests, smoke, bench (A/B of the bucket structure), rocprofv3 kernel stats of the same command, PMC
# passes, the recipe's stage 4/5, the 2-rank control flow on one GPU.  Each part has its own timeout and log under gpurun_out/.
#   gpurun --timeout 1500 -- 'bash tools/gpu_visit.sh [parts...]'      parts: tests smoke bench lpb abk rocprof pmc recipe tworank recipesize
#   round 4 added: micro (tools/microbench/handoff.hip, built into tools/exp/ beforehand), drift (tools/wide_drift_probe.py),
#   gap (tools/grad_gap_probe.py on the configs[3] reduced case), pmc3 (counter passes of the configs[3] geometry), seltests
#   (pytest -m gpu -k "$WN_TEST_K")
#   round 5 added: pins (tests/test_gpu_decode_pins.py + ops + co-residency, verbose), soak (tools/microbench/handoff_soak.hip),
#   pmcrecipe (counter passes of the recipe-size training step), k3probe (recipe-size decode speed, kernel_size 3 and 2),
#   dw3 (weight-gradient variants, same-box A/B), rocprofextra (rocprofv3 kernel stats of the recipe-size / configs[3] workloads)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"
PARTS="${*:-tests smoke bench lpb rocprof pmc recipe tworank recipesize}"
date +%s > $OUT/t0
has() { echo " $PARTS " | grep -q " $1 "; }
if has tests; then
  timeout 1200 python -m pytest tests -q -m gpu -s > $OUT/pytest_gpu_full.txt 2>&1; echo "pytest(full) rc=$?"; tail -4 $OUT/pytest_gpu_full.txt
  grep -h "vs oracle\|err \|FULL SIZE" $OUT/pytest_gpu_full.txt | head -20
fi
if has smoke; then
  timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"
fi
if has pmc; then
  # BEFORE the bench part: bench.py quotes the counter file of the newest profiles/rNN (PMC_FILES) and checks its source digest;
  # within a visit the fresh summary is copied there on the box, so that the bench line of the same visit says traffic_same_build
  # (the copy on the box is lost with it: copy gpurun_out/pmc_traffic.json into profiles/ and commit it as well)
  bash tools/pmc_traffic.sh
  NEWEST=$(ls -d profiles/r[0-9][0-9] 2>/dev/null | sort | tail -1)
  [ -n "$NEWEST" ] && [ -f $OUT/pmc_traffic.json ] && cp $OUT/pmc_traffic.json $NEWEST/pmc_traffic.json
fi
if has bench; then
  timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
fi
if has lpb; then
  # same-box A/B, interleaved, two rounds.  A variant is a comma-separated list of environment settings, e.g.
  #   WN_AB_VARIANTS="WN_X=1 WN_ENGINE_FLAGS=96 WN_LIB_PATH=tools/exp/libwn_x.so"   (launch-mode flags / variant builds: DESIGN.md 5.2)
  for rep in 1 2; do
    for cfg in ${WN_AB_VARIANTS:-WN_X=1 WN_ENGINE_FLAGS=96}; do
      env $(echo $cfg | tr ',' ' ') timeout 200 python bench.py --repeats 3 --no-cpu-baseline --no-decode --no-extras --profile-steps 0 > $OUT/bench_ab.json 2>> $OUT/bench.err
      python - <<P
import json
d = json.load(open("$OUT/bench_ab.json"))
print("%-40s ms/step median %.3f min %.3f max %.3f" % ("$cfg", d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"]))
P
    done
  done | tee $OUT/ab_probe.txt
fi
if has abk; then
  # like lpb, with the per-kernel HIP-event table of each variant (WN_AB_VARIANTS; WN_ABK_KERNELS = tags to print)
  for rep in 1 2; do
    for cfg in ${WN_AB_VARIANTS:-WN_X=1}; do
      env $(echo $cfg | tr ',' ' ') timeout 200 python bench.py --repeats 3 --no-cpu-baseline --no-decode --no-extras > $OUT/bench_abk.json 2>> $OUT/bench.err
      python - <<P
import json
d = json.load(open("$OUT/bench_abk.json"))
ks = d.get("kernels", {})
sel = "${WN_ABK_KERNELS:-bwd_dz_skip_all fwd_skip_sum dw_skip_res dw_skip dw_res dw_dilated bwd_post2_dx bwd_post1_dx fwd_post1 fwd_post2 dw_post1 dw_post2 fill_cols softmax_ce fused_bwd_chain fused_resblock_fwd}".split()
print("%-32s ms/step median %.3f min %.3f max %.3f | " % ("$cfg", d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"]) +
      " ".join("%s %.3f" % (k, ks[k]["ms_per_step"]) for k in sel if k in ks))
P
    done
  done | tee $OUT/abk_probe.txt
fi
if has rocprof; then
  rm -rf $OUT/prof_stats
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python $ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-decode --no-extras > $OUT/bench_rocprof.json 2> $OUT/rocprof.err); echo "rocprof rc=$?"
  find $OUT/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/rocprofv3_kernel_stats.csv
  python tools/kernel_gaps.py $OUT/prof_stats > $OUT/kernel_gaps.txt 2>&1; cat $OUT/kernel_gaps.txt
  find $OUT/prof_stats -name "*kernel_trace.csv" -delete
  head -12 $OUT/rocprofv3_kernel_stats.csv
fi
if has rocprofextra; then
  # rocprofv3 kernel stats of the two other training workloads (recipe-size model, configs[3] geometry)
  for w in recipe_size config4; do
    rm -rf $OUT/prof_$w
    if [ $w = recipe_size ]; then A="--resch 512 --batch 4 --steps 2"; else A="--resch 64 --kernel-size 3 --upsampling 256 --T 26112 --batch 8 --steps 3"; fi
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -- python $ROOT/tools/recipe_bench.py $A > $OUT/${w}_under_rocprofv3.json 2> $OUT/rocprof_$w.err); echo "rocprof $w rc=$?"
    find $OUT/prof_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${w}_rocprofv3_kernel_stats.csv
    find $OUT/prof_$w -name "*kernel_trace.csv" -delete
    head -8 $OUT/${w}_rocprofv3_kernel_stats.csv
  done
fi
if has recipesize; then
  # BASELINE configs[3] geometry (kernel_size 3, upsampling 256, T = 26112), softmax head
  timeout 300 python tools/recipe_bench.py --resch 64 --kernel-size 3 --upsampling 256 --T 26112 --batch 8 --steps 10 > $OUT/config4_bench.json 2> $OUT/config4_bench.err; echo "configs[3] bench rc=$?"
  python - <<P
import json
d = json.load(open("$OUT/config4_bench.json"))
print({k: v for k, v in d.items() if k != "kernels"})
for k, v in list(d["kernels"].items())[:12]:
    print("%-24s %3d launches %8.3f ms  tflops %s  GB/s %s" % (k, v["launches"], v["ms"], v["tflops"] and round(v["tflops"], 1), v["GBps"] and round(v["GBps"])))
P
  timeout 300 python tools/recipe_bench.py > $OUT/recipe_size_bench.json 2>> $OUT/recipe_size_bench.err; echo "recipe-size bench rc=$?"
  python - <<P
import json
d = json.load(open("$OUT/recipe_size_bench.json"))
print({k: v for k, v in d.items() if k != "kernels"})
for k, v in list(d["kernels"].items())[:14]:
    print("%-24s %3d launches %8.3f ms  tflops %s  GB/s %s" % (k, v["launches"], v["ms"], v["tflops"] and round(v["tflops"], 1), v["GBps"] and round(v["GBps"])))
P
fi
if has micro; then
  [ -x tools/exp/handoff ] || hipcc --offload-arch=gfx950 -O3 -o tools/exp/handoff tools/microbench/handoff.hip
  timeout 120 tools/exp/handoff > $OUT/handoff.txt 2>&1; echo "handoff rc=$?"; cat $OUT/handoff.txt
fi
if has drift; then
  timeout 400 python tools/wide_drift_probe.py $WN_DRIFT_ARGS > $OUT/wide_drift.txt 2>&1; echo "drift rc=$?"; tail -8 $OUT/wide_drift.txt
fi
if has gap; then
  timeout 300 python tools/grad_gap_probe.py 2 6656 3 256 6 > $OUT/grad_gap_config4_b2.txt 2>&1; echo "gap rc=$?"; cat $OUT/grad_gap_config4_b2.txt
fi
if has pmc3; then
  WN_PMC_NAME=config4 WN_PMC_STEPS=4 WN_PMC_CMD="python tools/recipe_bench.py --resch 64 --kernel-size 3 --upsampling 256 --T 26112 --batch 8 --steps 1" \
    bash tools/pmc_traffic.sh > $OUT/pmc_config4.txt 2>&1; tail -25 $OUT/pmc_config4.txt
fi
if has seltests; then
  timeout 1200 python -m pytest tests -q -m gpu -s -k "$WN_TEST_K" > $OUT/pytest_gpu_sel.txt 2>&1; echo "pytest(selected) rc=$?"; tail -5 $OUT/pytest_gpu_sel.txt
  grep -h "vs oracle\|vs own\|err \|STATED\|TIMED\|FULL SIZE" $OUT/pytest_gpu_sel.txt | head -20
fi
if has pins; then
  # round 5: the persistent-decode pins + what else changed (stand-alone modules, co-residency gate), verbose
  timeout 900 python -m pytest tests/test_gpu_decode_pins.py tests/test_gpu_ops.py tests/test_gpu_rccl_coresidency.py -q -m gpu -s -x > $OUT/pytest_gpu_pins.txt 2>&1; echo "pytest(pins) rc=$?"; tail -5 $OUT/pytest_gpu_pins.txt
  grep -h "LONG HORIZON\|WALKED\|SOAK\|k_dlpf<\|residency:\|backward pass median" $OUT/pytest_gpu_pins.txt | head -40
fi
if has soak; then
  [ -x tools/exp/handoff_soak ] || hipcc --offload-arch=gfx950 -O3 -o tools/exp/handoff_soak tools/microbench/handoff_soak.hip
  timeout 200 tools/exp/handoff_soak ${WN_SOAK_STAGES:-1000000} 1 > $OUT/handoff_soak.txt 2>&1; echo "soak rc=$?"; cat $OUT/handoff_soak.txt
  timeout 100 tools/exp/handoff_soak 200000 0 >> $OUT/handoff_soak.txt 2>&1; echo "soak(idle) rc=$?"; tail -2 $OUT/handoff_soak.txt
fi
if has pmcrecipe; then
  # counter passes of the recipe-size training step (n_resch 512, B = 4 x T = 23040: 2 warm-up + 1 timed + 1 event-logged step)
  WN_PMC_NAME=recipe WN_PMC_STEPS=4 WN_PMC_CMD="python tools/recipe_bench.py --resch 512 --batch 4 --steps 1" \
    bash tools/pmc_traffic.sh > $OUT/pmc_recipe.txt 2>&1; tail -25 $OUT/pmc_recipe.txt
fi
if has k3probe; then
  timeout 300 python tools/recipe_decode_probe.py --kernel-size 3 --steps 300 --batches 1,2,16,32,48,64 > $OUT/recipe_decode_probe_k3.txt 2>&1; echo "k3 decode probe rc=$?"; cat $OUT/recipe_decode_probe_k3.txt
  timeout 300 python tools/recipe_decode_probe.py --kernel-size 2 --steps 300 --batches 1,2,16,32,48,64,96 > $OUT/recipe_decode_probe_k2.txt 2>&1; echo "k2 decode probe rc=$?"; cat $OUT/recipe_decode_probe_k2.txt
fi
if has dw3; then
  # weight-gradient variants at the recipe size and the configs[3] geometry: same-box A/B, interleaved.  A variant = comma-separated
  # environment settings (WN_ENGINE_FLAGS: 32 = default, 262176 = + WN_FLAG_DW_3PRODUCT; WN_LIB_PATH = a variant build)
  for rep in 1 2; do
    for cfg in ${WN_DW3_VARIANTS:-WN_ENGINE_FLAGS=32 WN_ENGINE_FLAGS=262176}; do
      env $(echo $cfg | tr ',' ' ') timeout 300 python tools/recipe_bench.py --resch 512 --batch 4 --steps 3 > $OUT/dw3_recipe.json 2>> $OUT/dw3.err
      env $(echo $cfg | tr ',' ' ') timeout 300 python tools/recipe_bench.py --resch 64 --kernel-size 3 --upsampling 256 --T 26112 --batch 8 --steps 10 > $OUT/dw3_config4.json 2>> $OUT/dw3.err
      python - <<P
import json
for name in ("recipe", "config4"):
    d = json.load(open("$OUT/dw3_%s.json" % name))
    ks = d["kernels"]
    print("%-64s %-8s %8.3f ms/step | " % ("$cfg", name, d["ms_per_step"]) + " ".join("%s %.3f" % (k, v["ms"]) for k, v in ks.items() if k.startswith("dw_")))
P
    done
  done | tee $OUT/dw3_probe.txt
fi
if has recipe; then
  timeout 300 bash tools/recipe_stage45.sh run > $OUT/recipe_stage45.txt 2>&1; echo "recipe rc=$?"; tail -3 $OUT/recipe_stage45.txt
fi
if has tworank; then
  WN_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 2 --no-cpu-baseline --no-decode --no-extras > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "2-rank gloo rc=$?"; cut -c1-300 $OUT/bench_2rank_gloo.json
fi
if has eightrank; then
  # world 8 on ONE GPU (functional: the ranks share the device and rendezvous over gloo): bench.py's N = 8 control flow -- eight
  # processes, three gradient buckets each, the exposed-exchange measurement, the max-over-ranks timing -- and train.py --n_gpus 8
  WN_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --steps 3 --warmup 1 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-decode --no-extras > $OUT/bench_8rank_gloo.json 2> $OUT/bench_8rank_gloo.err; echo "8-rank gloo rc=$?"; cut -c1-300 $OUT/bench_8rank_gloo.json
fi
lscpu | grep -E "Model name|^CPU\(s\)|Socket" > $OUT/host.txt
free -g | head -2 >> $OUT/host.txt
echo "elapsed $(( $(date +%s) - $(cat $OUT/t0) )) s"
