#!/bin/bash
# Build experiment variants of libvbx_hip.so side by side (vbx_amd/csrc/libvbx_hip_<tag>.so); select one at run time
# with VBX_AMD_LIB=<path>.  usage: tools/build_variants.sh tag1:"-DX=1 -DY=2" tag2:"..."
cd "$(dirname "$0")/../vbx_amd/csrc" || exit 1
for spec in "$@"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -mllvm -slp-vectorize-hor=false $flags -o libvbx_hip_$tag.so vbx_capi.hip && echo "built $tag ($flags)" ) &
done
wait
