for v in drain0 drain1 drain0 drain1; do
VBX_AMD_LIB=vbx_amd/csrc/libvbx_hip_$v.so python tools/kbench.py --tag $v --iters 60 2>&1 | tail -1
done
for v in clk1; do
echo "== $v"
VBX_AMD_LIB=vbx_amd/csrc/libvbx_hip_$v.so python tools/phase_timeline.py 2>&1 | grep -v "^blk\|chunk_loglik" | head -12
done
