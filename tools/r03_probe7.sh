#!/bin/bash
export VBX_AMD_NO_REBUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p /tmp/mr && python - <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import test_driver as td, json
paths = td._write_inputs('/tmp/mr')
json.dump(td._argv(paths, ['--timing']), open('/tmp/mr/argv.json','w'))
PY
ARGS=$(python -c "import json; print(' '.join(json.load(open('/tmp/mr/argv.json'))))")
echo "--- single process"; PYTHONFAULTHANDLER=1 timeout 300 python -m vbx_amd.vbhmm $ARGS > $out/r03_mr_single.out 2> $out/r03_mr_single.err; echo rc=$?; tail -5 $out/r03_mr_single.err
echo "--- torchrun 2"; VBX_AMD_DEVICE=0 VBX_AMD_DIST_BACKEND=gloo PYTHONFAULTHANDLER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29511 -m vbx_amd.vbhmm $ARGS > $out/r03_mr_tr.out 2> $out/r03_mr_tr.err; echo rc=$?
grep -v "^\[Gloo\]" $out/r03_mr_tr.err | head -80
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_driver.py tests/test_gpu_drop_in.py -x -q -m gpu -k "shapes_sweep or more_speakers or 256 or driver_reproduces or minimal_caller" 2>&1 | tail -25
