#!/bin/bash
# First GPU visit of the next round: everything needed to make WN_FLAG_AUX_FUSED the default with evidence.
#   1. the whole `-m gpu` suite with the flag as the engine default (WN_ENGINE_FLAGS=32)
#   2. bench.py --aux-fused (self-check at full size, then the timed steps) and the default beside it
#   3. rocprofv3 kernel stats + the two PMC passes with the flag (k_conv64s<2> is then the dominant kernel)
# Afterwards: engine.DEFAULT_FLAGS = _lib.FLAG_AUX_FUSED, copy gpurun_out/{bench_aux.json, prof_aux, pmc_traffic.json} into
# profiles/, add "void k_conv64s<2>(ConvArgs)" to tools/pmc_traffic.py TAGS["fused_bwd_gate"].
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"; rm -rf $OUT/prof_aux $OUT/pmc_fetch $OUT/pmc_write
WN_ENGINE_FLAGS=32 timeout 400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_aux.txt 2>&1; echo "pytest(aux default) rc=$?"; tail -3 $OUT/pytest_gpu_aux.txt
timeout 200 python bench.py --aux-fused > $OUT/bench_aux.json 2> $OUT/bench_aux.err; echo "bench --aux-fused rc=$?"; cut -c1-260 $OUT/bench_aux.json
python - <<'P'
import json
d = json.loads(open("gpurun_out/bench_aux.json").readline())
print("aux_gradient:", d["config"]["aux_gradient"])
print("ms/step", d["ms_per_step"], "gate us/launch", 1e3 * d["kernels"]["fused_bwd_gate"]["ms_per_step"] / 30)
P
timeout 200 python bench.py --no-cpu-baseline --no-decode 2>/dev/null | cut -c1-230
(cd /tmp && WN_ENGINE_FLAGS=32 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_aux -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode > $OUT/bench_aux_rocprof.json 2> $OUT/rocprof_aux.err); echo "rocprof rc=$?"
WN_ENGINE_FLAGS=32 bash tools/pmc_traffic.sh
