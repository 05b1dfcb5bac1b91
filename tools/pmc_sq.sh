#!/bin/bash
# One rocprofv3 --pmc pass with the SQ wait / issue counters over 3 training steps (where do the waves of each kernel spend
# their cycles: parked on s_waitcnt / barriers, stalled at issue, or issuing).   gpurun -- 'bash tools/pmc_sq.sh'
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out"; mkdir -p $OUT; rm -rf $OUT/pmc_sq; export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-decode"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
cd $ROOT
python tools/pmc_summary.py $OUT/pmc_sq > $OUT/pmc_sq.json
python - <<'P'
import json
m = json.load(open("gpurun_out/pmc_sq.json"))
for k, v in sorted(m.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0) * kv[1].get("launches", 0))[:10]:
    w = v.get("SQ_WAVE_CYCLES", 1) or 1
    print("%-44s waves-cycles %.3g  wait_any %.2f  wait_inst %.2f (lds %.2f)  active %.2f (valu %.2f lds %.2f)  lds_conflict/wave-cyc %.3f" % (
        k[:44], w, v.get("SQ_WAIT_ANY", 0) / w, v.get("SQ_WAIT_INST_ANY", 0) / w, v.get("SQ_WAIT_INST_LDS", 0) / w,
        v.get("SQ_ACTIVE_INST_ANY", 0) / w, v.get("SQ_ACTIVE_INST_VALU", 0) / w, v.get("SQ_ACTIVE_INST_LDS", 0) / w,
        v.get("SQ_LDS_BANK_CONFLICT", 0) / w))
P
