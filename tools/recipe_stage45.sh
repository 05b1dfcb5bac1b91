#!/bin/bash
# Runs the REFERENCE's recipe script (egs/arctic/sd/run.sh, stages 4 = training and 5 = decoding, unmodified) against this
# repository's wavenet_vocoder/{bin,utils} -- the drop-in claim of north_star, executed instead of argued.
#
#   tools/recipe_stage45.sh prepare [/root/reference]   (build container: the reference checkout is readable here)
#       stages the recipe directory as a user would have it -- egs/arctic/sd/{run.sh,cmd.sh,path.sh,conf} copied
#       byte-for-byte from the reference into the git-ignored egs/ (never committed; sha256 recorded) -- plus what the
#       recipe expects of the user's machine: tools/venv/bin/activate (path.sh sources it; a stub that only adds a `bc`
#       stand-in to PATH and this checkout to PYTHONPATH) and a synthetic one-utterance corpus in the layout stages
#       0-3 would have left (data/tr_slt/{wav_hpf.scp,feats.scp,stats.h5}, data/ev_slt/feats.scp).
#   tools/recipe_stage45.sh run                          (GPU box: through gpurun; egs/ and tools/venv/ travel with the snapshot)
#       cd egs/arctic/sd && ./run.sh --stage 45 ... with a small model / few iterations; logs -> gpurun_out/recipe_stage45/
#   tools/recipe_stage45.sh clean
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
RECIPE="${ROOT}/egs/arctic/sd"
case "${1:-}" in
prepare)
    REF="${2:-/root/reference}"
    [ -r "${REF}/egs/arctic/sd/run.sh" ] || { echo "no reference recipe under ${REF}"; exit 1; }
    rm -rf "${ROOT}/egs" "${ROOT}/tools/venv"
    mkdir -p "${RECIPE}" "${ROOT}/tools/venv/bin"
    cp -r "${REF}/egs/arctic/sd/run.sh" "${REF}/egs/arctic/sd/cmd.sh" "${REF}/egs/arctic/sd/path.sh" "${REF}/egs/arctic/sd/conf" "${RECIPE}/"
    chmod -R u+w "${RECIPE}"
    (cd "${RECIPE}" && sha256sum run.sh cmd.sh path.sh > SHA256.reference)
    (cd "${REF}/egs/arctic/sd" && sha256sum run.sh cmd.sh path.sh) | diff - "${RECIPE}/SHA256.reference"
    # the user's environment: a "venv" whose activate only extends PATH / PYTHONPATH, and a bc stand-in (the image has none;
    # run.sh uses it once: upsampling_factor=$(echo "${shiftms} * ${fs} / 1000" | bc))
    cat > "${ROOT}/tools/venv/bin/activate" <<'EOS'
# stand-in for the recipe's virtualenv (egs/*/path.sh sources $PRJ_ROOT/tools/venv/bin/activate)
_venv_bin="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
export PATH="${_venv_bin}:${PATH}"
export PYTHONPATH="$(cd "${_venv_bin}/../../.." && pwd)${PYTHONPATH:+:${PYTHONPATH}}"
EOS
    cat > "${ROOT}/tools/venv/bin/bc" <<'EOS'
#!/usr/bin/env python
"""bc stand-in: integer arithmetic expressions on stdin, one result per line (truncating division like bc's scale=0)."""
import re
import sys
for line in sys.stdin:
    line = line.strip()
    if not line:
        continue
    if not re.fullmatch(r"[0-9+\-*/() ]+", line):
        sys.exit("bc stand-in: unsupported expression %r" % line)
    print(int(eval(line.replace("/", "//"))))
EOS
    chmod +x "${ROOT}/tools/venv/bin/bc"
    python "${ROOT}/tools/make_synth_corpus.py" "${RECIPE}"
    echo "staged ${RECIPE}"
    ;;
run)
    OUT="${ROOT}/gpurun_out/recipe_stage45"
    mkdir -p "${OUT}"
    cd "${RECIPE}"
    sha256sum -c SHA256.reference | tee "${OUT}/recipe_files_sha256_check.txt"
    set +e
    # feature_type world / 28 aux dims as the recipe fixes them; everything else are the recipe's own options
    ./run.sh --stage 45 --use_noise_shaping false --n_resch 64 --n_skipch 64 --dilation_depth 6 --dilation_repeat 2 \
        --iters 30 --batch_length 4000 --batch_size 2 --checkpoint_interval 15 --decode_batch_size 2 --tag synth \
        > "${OUT}/run_sh_stdout.txt" 2>&1
    rc=$?
    set -e
    echo "run.sh exit status ${rc}" | tee -a "${OUT}/run_sh_stdout.txt"
    cp -r exp/tr_arctic_synth/log "${OUT}/train_log" 2>/dev/null || true
    cp -r exp/tr_arctic_synth/wav/log "${OUT}/decode_log" 2>/dev/null || true
    ls -la exp/tr_arctic_synth exp/tr_arctic_synth/wav > "${OUT}/expdir_listing.txt" 2>&1 || true
    exit ${rc}
    ;;
clean)
    rm -rf "${ROOT}/egs" "${ROOT}/tools/venv"
    ;;
*)
    echo "usage: $0 prepare [reference-root] | run | clean"; exit 1 ;;
esac
