#!/bin/bash
export VBX_AMD_NO_REBUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k torchrun 2>&1 | tail -60 > $out/r03_mr.log; tail -60 $out/r03_mr.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_driver.py tests/test_gpu_drop_in.py -x -q -m gpu -k "shapes_sweep or more_speakers or 256 or driver_reproduces or minimal_caller" 2>&1 | tail -25
