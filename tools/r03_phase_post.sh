#!/bin/bash
cd vbx_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -DVBX_PHASE_CLOCKS -o /tmp/libvbx_hip_clk.so vbx_capi.hip && cd ../..
export VBX_AMD_LIB=/tmp/libvbx_hip_clk.so VBX_AMD_NO_REBUILD=1
echo "=== Sp=32 (64 x 10000 x 30)"; python tools/phase_timeline.py 64 10000 30 2>&1 | grep -v "chunk_loglik wave" | tail -12
echo "=== Sp=32, one recording"; python tools/phase_timeline.py 1 10000 30 2>&1 | grep -v "chunk_loglik wave" | tail -8
