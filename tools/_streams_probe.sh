cd $GRAFT_REPO_ROOT
{
for shape in "2 10000" "3 10000" "4 10000" "5 10000" "6 10000" "8 10000" "10 10000" "12 10000" "16 10000" "20 10000" "8 2000" "2 50000" "1 50000"; do
  set -- $shape
  timeout 120 python tools/ab_quick.py --batch $1 --T $2 --iters 400 --reps 4 2>&1 | tail -1
done
for p in fp32 fp64; do for n in 4 8 16; do timeout 120 python tools/ab_quick.py --batch $n --iters 300 --reps 3 --precision $p 2>&1 | tail -1; done; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream or group or shared or batch" 2>&1 | tail -3
} > gpurun_out/streams_probe3.txt 2>&1
cat gpurun_out/streams_probe3.txt
