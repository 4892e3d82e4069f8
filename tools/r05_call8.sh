#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out/timeline
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for p in fp32-split; do for st in 3 2; do
  timeout 600 rocprofv3 --kernel-trace -d $out/t_${p}_$st -o t -- python $GRAFT_REPO_ROOT/tools/profile_target.py --precision $p --streams $st --iters 120 > $out/t_${p}_$st.log 2>&1
  grep "ms per iteration" $out/t_${p}_$st.log
  python $GRAFT_REPO_ROOT/tools/timeline_overlap.py $out/t_${p}_$st/t_results.db
done; done
timeout 600 rocprofv3 --kernel-trace -d $out/t_c5 -o t -- python $GRAFT_REPO_ROOT/tools/kbench.py --sweep shared --T 200000 --S 50 --precision fp32-split --iters 8 > $out/t_c5.log 2>&1
python $GRAFT_REPO_ROOT/tools/timeline_overlap.py $out/t_c5/t_results.db 0.6
rm -rf $out/t_*/
