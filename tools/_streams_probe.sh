cd $GRAFT_REPO_ROOT
HEAD_LIB=$GRAFT_REPO_ROOT/build/libvbx_head.so
{
for shape in "1 10000 30" "1 20000 30" "1 30000 30" "1 38000 30" "1 50000 30" "1 70000 30" "1 100000 30" "1 120000 30" "1 150000 30" "1 200000 30" "1 50000 10" "1 100000 10" "2 30000 30" "4 20000 30"; do
  set -- $shape
  echo "--- $shape head / new"
  VBX_AMD_LIB=$HEAD_LIB timeout 120 python tools/ab_quick.py --batch $1 --T $2 --S $3 --iters 400 --reps 4 2>&1 | tail -1
  timeout 120 python tools/ab_quick.py --batch $1 --T $2 --S $3 --iters 400 --reps 4 2>&1 | tail -1
done
for shape in "1 30000 30" "1 100000 30"; do
  set -- $shape
  echo "--- fp64 $shape head / new"
  VBX_AMD_LIB=$HEAD_LIB timeout 120 python tools/ab_quick.py --batch $1 --T $2 --S $3 --iters 300 --reps 4 --precision fp64 2>&1 | tail -1
  timeout 120 python tools/ab_quick.py --batch $1 --T $2 --S $3 --iters 300 --reps 4 --precision fp64 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory.py -x -q -m gpu -k "walk or long or trajectory and c3 or level" 2>&1 | tail -3
} > gpurun_out/walk_probe2.txt 2>&1
cat gpurun_out/walk_probe2.txt
