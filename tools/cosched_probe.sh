#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
for cfg in "256 0" "256 64" "240 16" "192 64" "240 32" "192 48"; do
  set -- $cfg
  if [ "$2" = 0 ]; then unset WN_SIDE_CUS; else export WN_SIDE_CUS=$2; fi
  WN_CHAIN_BLOCKS=$1 timeout 120 python tools/cosched_probe.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/cosched_probe.txt
