#!/bin/bash
# L2 reuse of the shifted taps: per-launch times of the fused kernels with and without the tap-interleaved chunk order of the dX kernel
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
for cfg in "0" "1" "0" "1"; do
  set -- $cfg
  WN_DX_INTERLEAVE=$1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('dx_interleave $1  ms/step %.3f | fwd %.1f gate %.1f dx %.1f us per launch' % (d['ms_per_step'], 1e3*k['fused_resblock_fwd']['ms_per_step']/30, 1e3*k['fused_bwd_gate']['ms_per_step']/30, 1e3*k['fused_bwd_dx']['ms_per_step']/30))
"
done | tee gpurun_out/tap_probe.txt
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size or fused_equals_layered or engine_vs_golden" 2>&1 | tail -2
