# Scratch script of the last measurement run on the GPU box (gpurun -- 'bash tools/run_variants.sh > gpurun_out/x.log').
# Typical use: build A/B libraries with tools/build_variants.sh tagA:"-DX=1" tagB:"-DX=0", then alternate them in ONE run
# (boxes of the pool differ by ~4 %, runs on the same box by < 1 %):
for v in tagA tagB tagA tagB; do
VBX_AMD_LIB=vbx_amd/csrc/libvbx_hip_$v.so python tools/kbench.py --tag $v --iters 150 2>&1 | tail -1
done
