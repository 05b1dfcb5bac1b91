#!/bin/bash
# how many CUs do the persistent fused kernels need?  per-kernel HIP-event times of bench.py for WN_CHAIN_BLOCKS = 256 .. 128
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
for nb in 256 224 192 160 128; do
  WN_CHAIN_BLOCKS=$nb python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('blocks %3d  ms/step %.3f | fwd %.1f gate %.1f dx %.1f us per launch' % ($nb, d['ms_per_step'], 1e3*k['fused_resblock_fwd']['ms_per_step']/30, 1e3*k['fused_bwd_gate']['ms_per_step']/30, 1e3*k['fused_bwd_dx']['ms_per_step']/30))
"
done | tee gpurun_out/chain_blocks_probe.txt
