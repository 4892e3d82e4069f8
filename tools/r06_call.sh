export VBX_AMD_NO_REBUILD=1
cd $GRAFT_REPO_ROOT
for f in 0 1; do
 for nb in 1 8 16; do VBX_AMD_FOLD_WALK=$f python tools/ab_quick.py --batch $nb --iters 300 | sed "s/^/fold=$f /"; done
 VBX_AMD_FOLD_WALK=$f python tools/ab_quick.py --batch 1 --S 10 --iters 300 | sed "s/^/fold=$f /"
 VBX_AMD_FOLD_WALK=$f python tools/ab_quick.py --batch 1 --T 50000 --iters 200 | sed "s/^/fold=$f /"
 VBX_AMD_FOLD_WALK=$f python tools/ab_quick.py --batch 8 --precision fp64 --iters 300 | sed "s/^/fold=$f /"
done
VBX_AMD_FOLD_WALK=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
