#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
python tools/bench_call.py batch
python tools/bench_call.py
timeout 600 python -m pytest tests/test_gpu_multirank.py -q 2>&1 | tail -3
