#!/bin/bash
# One GPU-box visit: A/B of the stream-overlap modes, then tests / bench / rocprofv3 under the fastest mode.
# Everything lands in gpurun_out/.   usage: gpurun --timeout 780 -- 'bash tools/gpu_round.sh'
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"
date +%s > $OUT/t0
WN_SIDE_PRIORITY=low timeout 240 python tools/overlap_probe.py > $OUT/overlap_probe_low.txt 2>&1; echo "probe low rc=$?"
WN_SIDE_PRIORITY=normal timeout 120 python tools/overlap_probe.py > $OUT/overlap_probe_normal.txt 2>&1; echo "probe normal rc=$?"
tail -12 $OUT/overlap_probe_low.txt; tail -11 $OUT/overlap_probe_normal.txt
python - > $OUT/best_env.sh <<'P'
import json, os
best = None
for pr in ("low", "normal"):
    try:
        d = json.load(open("gpurun_out/overlap_probe_%s.json" % pr))
    except (OSError, ValueError):
        continue
    if d.get("bitwise_equal") and (best is None or d["fastest_ms"] < best[0] - 0.02):
        best = (d["fastest_ms"], d["fastest_flags"], pr)
if best is None:
    print("export WN_ENGINE_FLAGS=4")       # serial
else:
    print("export WN_ENGINE_FLAGS=%d WN_SIDE_PRIORITY=%s  # %.3f ms/step" % (best[1], best[2], best[0]))
P
cat $OUT/best_env.sh; source $OUT/best_env.sh
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"
timeout 240 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overlap or engine_vs_golden or bucketed or full_size" > $OUT/pytest_overlap.txt 2>&1; echo "pytest(overlap) rc=$?"; tail -3 $OUT/pytest_overlap.txt
timeout 240 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode > $OUT/bench_rocprof.json 2> $OUT/rocprof.err); echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats*" | head -3
echo "elapsed $(( $(date +%s) - $(cat $OUT/t0) )) s"
timeout ${FULL_TEST_TIMEOUT:-300} python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_full.txt 2>&1; echo "pytest(full) rc=$?"; tail -3 $OUT/pytest_gpu_full.txt
echo "elapsed $(( $(date +%s) - $(cat $OUT/t0) )) s"
