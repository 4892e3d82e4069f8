for s in 0 1 0 1; do
echo "stages=$s"
VBX_AMD_LINKAGE_STAGES=$s python tools/bench_driver.py --recordings 2 --xvectors 20000 --cpu-recordings 0 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(' 2x20k', round(d['seconds'],3), d['stages_s'])"
VBX_AMD_LINKAGE_STAGES=$s python tools/bench_driver.py --recordings 1 --xvectors 20000 --cpu-recordings 0 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(' 1x20k', round(d['seconds'],3), d['stages_s'])"
done
