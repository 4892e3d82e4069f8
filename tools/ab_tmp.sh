export VBX_AMD_NO_REBUILD=1
python bench.py --steps 20000 --warmup 5 --cpu-iters 0 --no-single --batch 32 2>/dev/null | tail -1 > gpurun_out/p1.json &
python bench.py --steps 20000 --warmup 5 --cpu-iters 0 --no-single --batch 32 2>/dev/null | tail -1 > gpurun_out/p2.json &
wait
python -c "
import json
a=json.load(open('gpurun_out/p1.json')); b=json.load(open('gpurun_out/p2.json'))
print('two concurrent processes, batch 32 each:', a['value_without_kernel_events'], b['value_without_kernel_events'], 'sum', a['value_without_kernel_events']+b['value_without_kernel_events'])"
python bench.py --steps 20000 --warmup 5 --cpu-iters 0 --no-single --batch 64 2>/dev/null | tail -1 > gpurun_out/p1.json
python -c "
import json
a=json.load(open('gpurun_out/p1.json')); print('one process, batch 64:', a['value_without_kernel_events'])"
