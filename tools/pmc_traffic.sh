#!/bin/bash
# Counter passes over 3 training steps of bench.py in the engine's default launch mode (one rocprofv3 run per counter set:
# FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2; the SQ set has its own pass; no tracing domains next to --pmc):
#   gpurun --timeout 600 -- 'bash tools/pmc_traffic.sh'
# -> gpurun_out/pmc/{fetch,write,mfma}/  raw counter CSVs,  gpurun_out/pmc_traffic.json  (read by bench.py from profiles/),
#    gpurun_out/pmc_mfma.json (per-kernel MFMA-busy / SQ-busy / GUI-active means).  Copy what is to be judged into profiles/.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out"; mkdir -p $OUT; rm -rf $OUT/pmc; mkdir -p $OUT/pmc; export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-decode --no-extras $WN_PMC_BENCH_ARGS"
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/fetch -- $CMD > $OUT/pmc/fetch.log 2>&1; echo "fetch rc=$?"
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/write -- $CMD > $OUT/pmc/write.log 2>&1; echo "write rc=$?"
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc/mfma -- $CMD > $OUT/pmc/mfma.log 2>&1; echo "mfma rc=$?"
cd $ROOT
python tools/pmc_traffic.py $OUT/pmc/fetch $OUT/pmc/write 3 $OUT/pmc/fetch.log $OUT/pmc/mfma > $OUT/pmc_traffic.json; echo "summary rc=$?"
python tools/pmc_summary.py $OUT/pmc/mfma > $OUT/pmc_mfma.json; echo "mfma summary rc=$?"
python - <<'P'
import json
d = json.load(open("gpurun_out/pmc_traffic.json"))
for k in ("fused_bwd_gate", "fused_resblock_fwd", "fused_bwd_dx", "fused_bwd_chain"):
    print(k, d.get(k))
print("engine flags", d.get("_engine_flags"), "step total GB", d["_step_total_bytes"] / 1e9)
m = json.load(open("gpurun_out/pmc_mfma.json"))
for k, v in sorted(m.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0) * kv[1].get("launches", 0))[:12]:
    b = v.get("SQ_BUSY_CYCLES", 0)
    print("%-70s mfma_busy/sq_busy %.3f  launches %d" % (k[:70], v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / b if b else 0, v.get("launches", 0)))
P
du -sh $OUT/pmc
