#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
echo "--- wave priority of the latency kernels (walk, group products, fin): 0 / 1 / 3, two rounds"
for rep in 1 2; do for v in prio0 prio3 prio1; do
  export VBX_AMD_LIB=$PWD/vbx_amd/csrc/libvbx_hip_$v.so
  python tools/kbench.py --precision fp32-split --iters 150 --tag split_$v | cut -c1-25,150-420
done; done
for v in prio0 prio3; do
  export VBX_AMD_LIB=$PWD/vbx_amd/csrc/libvbx_hip_$v.so
  python tools/kbench.py --precision fp64 --iters 60 --tag f64_$v | cut -c1-25,150-420
  python tools/kbench.py --precision fp32 --iters 100 --tag exact_$v | cut -c1-25,150-420
  python tools/kbench.py --sweep shared --T 200000 --S 50 --precision fp32-split --iters 8 --tag c5_$v | cut -c1-25,150-460
  python tools/kbench.py --precision fp32-split --batch 8 --iters 100 --tag split8_$v | cut -c1-25,150-420
  python tools/kbench.py --precision fp32-split --batch 1 --iters 100 --tag split1_$v | cut -c1-25,150-420
done
