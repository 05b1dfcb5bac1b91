#!/bin/bash
# Final visit of a round: full GPU tests, smoke, bench (+ the same under rocprofv3), 2-rank control flow over gloo on one GPU.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"; rm -rf $OUT/prof_final
date +%s > $OUT/t0
timeout 400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_full.txt 2>&1; echo "pytest(full) rc=$?"; tail -3 $OUT/pytest_gpu_full.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"
timeout 240 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_final -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode > $OUT/bench_rocprof.json 2> $OUT/rocprof.err); echo "rocprof rc=$?"
WN_BENCH_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-decode > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "2-rank gloo rc=$?"; cut -c1-300 $OUT/bench_2rank_gloo.json
lscpu | grep -E "Model name|^CPU\(s\)|Socket" > $OUT/host.txt
echo "elapsed $(( $(date +%s) - $(cat $OUT/t0) )) s"
