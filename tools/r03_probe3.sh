#!/bin/bash
export VBX_AMD_NO_REBUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/valu_probe.hip && /tmp/valu_probe > $out/r03_valu_probe.txt 2>&1
cat $out/r03_valu_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_ahc.py -x -q -m gpu -k "sweep or shared or score_matrix or c5" 2>&1 | tail -15
