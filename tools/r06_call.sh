export VBX_AMD_NO_REBUILD=1
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_trajectory.py -q -s 2>&1 | grep -E "max over|passed|failed" | cut -c1-250
python -m pytest tests/test_gpu_split.py tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
