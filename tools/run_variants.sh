python -m pytest tests/test_gpu_parity.py -q -x -k "two_level or long_recordings" 2>&1 | tail -3
python -m pytest tests/test_gpu_configs.py -q -x 2>&1 | tail -3
for g in 0 16 24 32; do
VBX_AMD_SCAN_GROUP=$g python tools/kbench.py --tag T200k-g$g --batch 1 --T 200000 --S 50 --iters 20 2>&1 | tail -1
done
for g in 0 8 12 16; do
VBX_AMD_SCAN_GROUP=$g python tools/kbench.py --tag T50k-g$g --batch 1 --T 50000 --S 30 --iters 20 2>&1 | tail -1
done
for g in 0 4 6 8; do
VBX_AMD_SCAN_GROUP=$g python tools/kbench.py --tag T10k-g$g --batch 1 --iters 50 2>&1 | tail -1
done
VBX_AMD_SCAN_GROUP=0 python tools/kbench.py --tag T200k-f64 --batch 1 --T 200000 --S 50 --iters 20 --precision fp64 2>&1 | tail -1
