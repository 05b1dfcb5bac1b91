#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
for rep in 1 2; do for bal in 1 0; do
  WN_CHAIN_BALANCE=$bal python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-decode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('balance $bal  ms/step %.3f | fwd %.1f gate %.1f dx %.1f us per launch' % (d['ms_per_step'], 1e3*k['fused_resblock_fwd']['ms_per_step']/30, 1e3*k['fused_bwd_gate']['ms_per_step']/30, 1e3*k['fused_bwd_dx']['ms_per_step']/30))
"
done; done | tee gpurun_out/balance_probe.txt
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size or fused_equals_layered or overlap" 2>&1 | tail -2
