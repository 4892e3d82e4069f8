#!/bin/bash
# A/B of the one-row-ahead fetch in the operator recursion (VBX_OP_PREFETCH) on the bench workload, C3 and C5, both precisions
build() { (cd vbx_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed $2 -o $1 vbx_capi.hip); }
build /tmp/libvbx_pf0.so -DVBX_OP_PREFETCH=0 &
build /tmp/libvbx_pf1.so -DVBX_OP_PREFETCH=1 &
wait
export VBX_AMD_NO_REBUILD=1
for lib in pf0 pf1 pf0 pf1; do
  export VBX_AMD_LIB=/tmp/libvbx_$lib.so
  python tools/kbench.py --tag ${lib}_b64_fp32
  python tools/kbench.py --precision fp64 --tag ${lib}_b64_fp64
  python tools/kbench.py --batch 1 --T 50000 --tag ${lib}_C3
  python tools/kbench.py --batch 1 --tag ${lib}_single
  python tools/kbench.py --sweep shared --T 200000 --S 50 --iters 10 --tag ${lib}_C5
done
unset VBX_AMD_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
