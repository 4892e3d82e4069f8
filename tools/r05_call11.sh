#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "more_than_1024 or more_speakers or shapes_sweep or wide" 2>&1 | tail -25
