#!/bin/bash
# ab_bench.sh variant [variant ...] -- bench.py (short) with the product library ("base") or tools/exp/libwn_<variant>.so
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for v in "$@"; do
  if [ "$v" = base ]; then unset WN_LIB_PATH; else export WN_LIB_PATH="$ROOT/tools/exp/libwn_$v.so"; fi
  python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-decode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('%-8s ms/step %.2f | fwd %.1f gate %.1f dx %.1f us | dw_dil %.2f dw_skip %.2f skip %.2f dw_res %.2f ms' % ('$v', d['ms_per_step'], 1e3*k['fused_resblock_fwd']['ms_per_step']/30, 1e3*k['fused_bwd_gate']['ms_per_step']/30, 1e3*k['fused_bwd_dx']['ms_per_step']/30, k['dw_dilated']['ms_per_step'], k['dw_skip']['ms_per_step'], k['fwd_skip_sum']['ms_per_step'], k['dw_res']['ms_per_step']))
"
done
