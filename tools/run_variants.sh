for v in t0 t1 t0 t1 t0 t1; do
VBX_AMD_LIB=vbx_amd/csrc/libvbx_hip_$v.so python tools/kbench.py --tag $v --iters 150 2>&1 | tail -1
done
