python -m pytest tests -q -x -m gpu 2>&1 | tail -3
for v in p0 p1 p0 p1; do
VBX_AMD_LIB=vbx_amd/csrc/libvbx_hip_$v.so python tools/kbench.py --tag $v --iters 150 2>&1 | tail -1
done
for v in p0 p1; do
VBX_AMD_LIB=vbx_amd/csrc/libvbx_hip_$v.so python tools/kbench.py --tag $v-b1 --batch 1 --iters 100 2>&1 | tail -1
VBX_AMD_LIB=vbx_amd/csrc/libvbx_hip_$v.so python tools/kbench.py --tag $v-f64 --precision fp64 --iters 50 2>&1 | tail -1
done
