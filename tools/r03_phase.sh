#!/bin/bash
cd vbx_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -DVBX_PHASE_CLOCKS -o /tmp/libvbx_hip_clk.so vbx_capi.hip && cd ../..
export VBX_AMD_LIB=/tmp/libvbx_hip_clk.so VBX_AMD_NO_REBUILD=1
echo "=== Sp=64 (5 x T=200000, S=50)"; python tools/phase_timeline.py 5 200000 50 2>&1 | grep -E "chunk_loglik wave|median|second|kernel span|first 1024|the rest" | head -30
echo "=== Sp=32 (64 x 10000 x 30)"; python tools/phase_timeline.py 64 10000 30 2>&1 | grep -E "chunk_loglik wave|median|second|kernel span" | head -20
