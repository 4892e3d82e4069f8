export VBX_AMD_NO_REBUILD=1
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "enqueued_uploads or python_batch_api or sharing_across or stream_groups or resident_setter or sweep or slot_set_again or early_stop" 2>&1 | tail -3
python -m pytest tests/test_gpu_split.py tests/test_gpu_drop_in.py tests/test_driver.py -x -q -m gpu 2>&1 | tail -3
python tools/call_breakdown.py fp32-split 40 2>&1 | sed -n 2,4p
python tools/call_breakdown.py fp64 40 2>&1 | sed -n 2,3p
python tools/bench_call.py batch 2>&1 | tail -6
