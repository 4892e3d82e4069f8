export VBX_AMD_NO_REBUILD=1
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_trajectory.py -q -k "not c5" -s 2>&1 | grep -E "max over|passed|failed|continued|Error" | cut -c1-250
python -m pytest tests/test_gpu_parity.py -x -q -k "resident_setter or more_than_1024" 2>&1 | tail -3
