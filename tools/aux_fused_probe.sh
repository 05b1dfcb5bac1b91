#!/bin/bash
# WN_FLAG_AUX_FUSED (32) against the default: step time, per-kernel times, and the GPU parity test of the mode
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
for fl in 0 32 0 32; do
  WN_ENGINE_FLAGS=$fl python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
g=lambda n: k.get(n,{}).get('ms_per_step',0.0)
print('flags %2d  ms/step %.3f | gate %.1f us per launch | aux_bwd %.3f aux_finish %.3f dw_dilated %.3f ms' % ($fl, d['ms_per_step'], 1e3*g('fused_bwd_gate')/30, g('aux_bwd'), g('aux_finish'), g('dw_dilated')))
"
done | tee gpurun_out/aux_fused_probe.txt
timeout 100 python -m pytest tests/test_zz_aux_fused_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee -a gpurun_out/aux_fused_probe.txt
