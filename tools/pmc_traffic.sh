#!/bin/bash
# Two separate counter passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2) over 3 training steps; no tracing
# domains next to --pmc.   gpurun --timeout 400 -- 'bash tools/pmc_traffic.sh'
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out"; mkdir -p $OUT; rm -rf $OUT/pmc_fetch $OUT/pmc_write; export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-decode"
timeout 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
cd $ROOT
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write 3 > $OUT/pmc_traffic.json; echo "summary rc=$?"
python - <<'P'
import json
d = json.load(open("gpurun_out/pmc_traffic.json"))
for k in ("fused_bwd_gate", "fused_resblock_fwd", "fused_bwd_dx"):
    print(k, d.get(k))
print("step total GB", d["_step_total_bytes"] / 1e9)
P
du -sh $OUT/pmc_fetch $OUT/pmc_write
