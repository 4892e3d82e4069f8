#!/bin/bash
# The closing run of a round in ONE gpurun call:  tools/closing_run.sh <round tag, e.g. r04>
#   the -m gpu suite, smoke(), the round's profiles (tools/profiles.sh: they are stamped with the hash of the kernel sources
#   and copied into profiles/ on the box), THEN the two bench lines, which quote the profiles they find.
# What comes back: gpurun_out/profiles_<r>/* (copy into profiles/), gpurun_out/<r>_config_parity.json, gpurun_out/<r>_bench_*.json
r=${1:-r05}
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
out=gpurun_out
( time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 ) 2>&1 | tail -8
cp $out/config_parity.json $out/${r}_config_parity.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
( time bash tools/profiles.sh $r > $out/${r}_profiles.log 2>&1 ) 2>&1 | grep real
# (stdout of bench.py = the compact line the driver parses; the full record goes to --full-out; stderr repeats it)
( time python bench.py --full-out $out/${r}_bench_default.json > $out/${r}_bench_default_compact.json 2> /dev/null ) 2>&1 | grep real
( time python bench.py --steps 20 --warmup 5 --full-out $out/${r}_bench_driver_args.json > $out/${r}_bench_driver_args_compact.json 2> /dev/null ) 2>&1 | grep real
wc -c $out/${r}_bench_driver_args_compact.json
python tools/bench_summary.py $out/${r}_bench_default.json $out/${r}_bench_driver_args.json | cut -c1-200
du -sh $out | tail -1
