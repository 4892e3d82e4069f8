python tools/kbench.py --tag main 2>&1 | tail -1
python tools/kbench.py --tag main-f64 --precision fp64 2>&1 | tail -1
python tools/kbench.py --tag single --batch 1 2>&1 | tail -1
