#!/bin/bash
# Round-4 A/B builds (tools/exp/libwn_<name>.so, loaded through WN_LIB_PATH):
#   fine       k_gemm6: one piece of a k-step's loads / operand split after every PAIR of MFMAs (-DWN_G6_FINE)
#   noslp      every source with -fno-slp-vectorize (no v_pk_* VALU beside the MFMAs: guide, "packed f32 VALU is an anti-lever")
#   finenoslp  both
#   flip       k_gemm6: alternating-sign column tiles for the backward-dX type launches (-DWN_G6_FLIP; tools/wide_drift_probe.py)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/pytorchwavenetvocoder_amd/csrc"
EXP="$ROOT/tools/exp"; mkdir -p "$EXP"
ALL="wn_gemm wn_gemm6 wn_elem wn_fused wn_decode wn_prof wn_api"
CC="hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC"
python "$CSRC/build.py" > /dev/null
( for f in $ALL; do $CC -fno-slp-vectorize -c "$CSRC/$f.hip" -o "$EXP/$f.noslp.o" & done; wait )
$CC -DWN_G6_FINE -c "$CSRC/wn_gemm6.hip" -o "$EXP/wn_gemm6.fine.o" &
$CC -DWN_G6_FINE -fno-slp-vectorize -c "$CSRC/wn_gemm6.hip" -o "$EXP/wn_gemm6.finenoslp.o" &
$CC -DWN_G6_FLIP -c "$CSRC/wn_gemm6.hip" -o "$EXP/wn_gemm6.flip.o" &
wait
link() { name=$1; shift; hipcc --offload-arch=gfx950 -shared -fPIC -o "$EXP/libwn_$name.so" "$@"; echo "$EXP/libwn_$name.so"; }
base_objs() { for f in $ALL; do [ "$f" = "$1" ] || echo "$CSRC/$f.o"; done; }
noslp_objs() { for f in $ALL; do [ "$f" = "$1" ] || echo "$EXP/$f.noslp.o"; done; }
link fine $(base_objs wn_gemm6) "$EXP/wn_gemm6.fine.o"
link noslp $(noslp_objs none)
link finenoslp $(noslp_objs wn_gemm6) "$EXP/wn_gemm6.finenoslp.o"
link flip $(base_objs wn_gemm6) "$EXP/wn_gemm6.flip.o"
