export VBX_AMD_NO_REBUILD=1
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_trajectory.py -x -q -k "not c5" -s 2>&1 | tail -30 > gpurun_out/r06_call1_traj.log
python -m pytest tests/test_gpu_parity.py -x -q -k "resident_setter or more_than_1024" -s 2>&1 | tail -30 > gpurun_out/r06_call1_parity.log
cd /tmp && export TMPDIR=/tmp
for nb in 1 8; do
  timeout 300 rocprofv3 --kernel-trace -d /tmp/gap$nb -o trace -- python $GRAFT_REPO_ROOT/tools/profile_target.py --batch $nb --precision fp32-split --iters 300 > $GRAFT_REPO_ROOT/gpurun_out/r06_gap_b$nb.log 2>&1
  python $GRAFT_REPO_ROOT/tools/launch_gaps.py /tmp/gap$nb/trace_results.db $GRAFT_REPO_ROOT/gpurun_out/r06_launch_gaps_b$nb.txt
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/r06_call1_traj.log gpurun_out/r06_call1_parity.log
