#!/bin/bash
export VBX_AMD_NO_REBUILD=1
python tools/r03_s257.py 2>&1 | tail -30
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_driver.py tests/test_gpu_drop_in.py -x -q -m gpu 2>&1 | tail -8
