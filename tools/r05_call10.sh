#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
echo "--- 64-frame tiles (libvbx_hip_t64.so) against 128 (default): small batches"
for lib in default t64; do
  if [ $lib = default ]; then unset VBX_AMD_LIB; else export VBX_AMD_LIB=$PWD/vbx_amd/csrc/libvbx_hip_$lib.so; fi
  for p in fp32 fp64; do
    python tools/kbench.py --precision $p --batch 1 --iters 100 --tag ${lib}_${p}_b1 | cut -c1-330
    python tools/kbench.py --precision $p --batch 8 --iters 100 --tag ${lib}_${p}_b8 | cut -c1-330
    python tools/kbench.py --precision $p --batch 1 --T 10000 --S 10 --iters 100 --tag ${lib}_${p}_c2 | cut -c1-330
    python tools/kbench.py --precision $p --batch 1 --T 50000 --S 30 --iters 60 --tag ${lib}_${p}_c3 | cut -c1-330
    python tools/kbench.py --precision $p --batch 64 --iters 40 --tag ${lib}_${p}_b64 | cut -c1-330
  done
done
export VBX_AMD_LIB=$PWD/vbx_amd/csrc/libvbx_hip_t64.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_drop_in.py tests/test_gpu_ahc.py -q -x -k "not split" 2>&1 | tail -15
