export VBX_AMD_NO_REBUILD=1
cd $GRAFT_REPO_ROOT
for nb in 1 8; do python tools/ab_quick.py --batch $nb --iters 300; done
python tools/ab_quick.py --batch 8 --precision fp32 --iters 300
python tools/ab_quick.py --batch 8 --precision fp64 --iters 300
python tools/ab_quick.py --batch 32 --precision fp64 --iters 300
python tools/ab_quick.py
python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py -x -q 2>&1 | tail -2
