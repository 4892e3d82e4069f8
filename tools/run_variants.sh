python tools/kbench.py --tag T50k --batch 1 --T 50000 --iters 40 2>&1 | tail -1
python tools/kbench.py --tag T200k --batch 1 --T 200000 --S 50 --iters 20 2>&1 | tail -1
python tools/kbench.py --tag T200k-f64 --batch 1 --T 200000 --S 50 --iters 20 --precision fp64 2>&1 | tail -1
python tools/kbench.py --tag b8 --batch 8 --iters 60 2>&1 | tail -1
