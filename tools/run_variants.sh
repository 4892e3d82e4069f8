python tools/kbench.py --tag s128 --batch 1 --S 128 2>&1 | tail -1
python tools/kbench.py --tag s200 --batch 1 --S 200 2>&1 | tail -1
python tools/kbench.py --tag s200f64 --batch 1 --S 200 --precision fp64 2>&1 | tail -1
