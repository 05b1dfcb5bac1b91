#!/bin/bash
# build_variant.sh NAME "EXTRA_FLAGS" [source.hip ...] -- experimental gfx950 build of the library with
# extra compile flags for the listed sources (default: wn_fused.hip) into tools/exp/libwn_NAME.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/pytorchwavenetvocoder_amd/csrc"
NAME="$1"; FLAGS="$2"; shift 2
SRCS="${@:-wn_fused.hip}"
mkdir -p "$ROOT/tools/exp"
OBJS=""
for f in wn_gemm wn_gemm6 wn_elem wn_fused wn_decode wn_dlp wn_dlpm wn_dlpf wn_prof wn_api; do
  if echo " $SRCS " | grep -q " $f.hip "; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c "$CSRC/$f.hip" -o "$ROOT/tools/exp/$f.$NAME.o"
    OBJS="$OBJS $ROOT/tools/exp/$f.$NAME.o"
  else
    OBJS="$OBJS $CSRC/$f.o"
  fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/exp/libwn_$NAME.so" $OBJS
echo "$ROOT/tools/exp/libwn_$NAME.so"
