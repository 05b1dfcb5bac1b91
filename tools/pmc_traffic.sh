#!/bin/bash
# Counter passes over 3 training steps of bench.py in the engine's default launch mode (one rocprofv3 run per counter set:
# FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2; the SQ set has its own pass; no tracing domains next to --pmc):
#   gpurun --timeout 600 -- 'bash tools/pmc_traffic.sh'
# -> gpurun_out/pmc/{fetch,write,mfma}/  raw counter CSVs,  gpurun_out/pmc_traffic.json  (read by bench.py from profiles/),
#    gpurun_out/pmc_mfma.json (per-kernel MFMA-busy / SQ-busy / GUI-active means).  Copy what is to be judged into profiles/.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
#   WN_PMC_CMD="python tools/recipe_bench.py --resch 64 --kernel-size 3 ..." WN_PMC_STEPS=4 WN_PMC_NAME=config4  another workload
#   (then the summaries are gpurun_out/pmc_traffic_config4.json / pmc_mfma_config4.json and the raw CSVs gpurun_out/pmc_config4/)
OUT="$ROOT/gpurun_out"; mkdir -p $OUT; export TMPDIR=/tmp
SUF="${WN_PMC_NAME:+_$WN_PMC_NAME}"
PMC="$OUT/pmc$SUF"; rm -rf $PMC; mkdir -p $PMC
cd /tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-decode --no-extras $WN_PMC_BENCH_ARGS"
STEPS=3
if [ -n "$WN_PMC_CMD" ]; then CMD="$(echo "$WN_PMC_CMD" | sed "s# tools/# $ROOT/tools/#g")"; STEPS="${WN_PMC_STEPS:-3}"; fi
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $PMC/fetch -- $CMD > $PMC/fetch.log 2>&1; echo "fetch rc=$?"
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $PMC/write -- $CMD > $PMC/write.log 2>&1; echo "write rc=$?"
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $PMC/mfma -- $CMD > $PMC/mfma.log 2>&1; echo "mfma rc=$?"
cd $ROOT
python tools/pmc_traffic.py $PMC/fetch $PMC/write $STEPS $PMC/fetch.log $PMC/mfma > $OUT/pmc_traffic$SUF.json; echo "summary rc=$?"
python tools/pmc_summary.py $PMC/mfma > $OUT/pmc_mfma$SUF.json; echo "mfma summary rc=$?"
SUF="$SUF" python - <<'P'
import json, os
suf = os.environ.get("SUF", "")
d = json.load(open("gpurun_out/pmc_traffic%s.json" % suf))
for k in ("fused_bwd_gate", "fused_resblock_fwd", "fused_bwd_dx", "fused_bwd_chain"):
    print(k, d.get(k))
print("engine flags", d.get("_engine_flags"), "step total GB", d["_step_total_bytes"] / 1e9)
m = json.load(open("gpurun_out/pmc_mfma%s.json" % suf))
for k, v in sorted(m.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0) * kv[1].get("launches", 0))[:12]:
    b = v.get("SQ_BUSY_CYCLES", 0)
    print("%-70s mfma_busy/sq_busy %.3f  launches %d" % (k[:70], v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / b if b else 0, v.get("launches", 0)))
P
du -sh $PMC
