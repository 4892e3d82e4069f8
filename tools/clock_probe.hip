// clock_probe.hip -- what shader clock does a latency-bound single-wave kernel actually get?
// Build: hipcc --offload-arch=gfx950 -O3 -o clock_probe tools/clock_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float* out, long long* ticks, int iters) {
    float x = threadIdx.x * 1e-9f + 1.0f;
    const long long c0 = clock64();
    const long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 0.999999f, 1e-7f);   // dependent chain
    const long long c1 = clock64();
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = c1 - c0; ticks[1] = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <int MODE> __global__ void reduce_chain(float* out, long long* ticks, int iters) {
    float x = threadIdx.x * 1e-3f + 1.0f;
    const long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        float v = x;
        if (MODE == 0) {            // 4 DPP + bpermute(16) : current allreduce_sum<32>
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
            v += __shfl_xor(v, 16, 64);
        } else if (MODE == 1) {     // 4 DPP + permlane16_swap
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
            auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
            v = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
        } else {                    // LDS-free reference: plain dependent fma (cost floor)
            v = v * 1.0001f;
        }
        x = v * 0.03125f + 0.5f;
    }
    const long long c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = c1 - c0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
    float* out; long long* ticks; long long h[2];
    hipMalloc(&out, 1 << 20); hipMalloc(&ticks, 16);
    const int iters = 200000;
    for (int blocks : {1, 256, 2048}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a); chain<<<blocks, 64>>>(out, ticks, iters); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
            printf("fma chain  blocks=%4d  shader_cycles/iter=%.2f  wall100MHz_ticks=%lld -> shader clock %.0f MHz; event %.3f ms -> %.1f ns/iter\n",
                   blocks, (double)h[0] / iters, h[1], (double)h[0] / (h[1] / 100.0), ms, ms * 1e6 / iters);
        }
    }
    const int it2 = 20000;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        if (mode == 0) reduce_chain<0><<<1, 64>>>(out, ticks, it2);
        if (mode == 1) reduce_chain<1><<<1, 64>>>(out, ticks, it2);
        if (mode == 2) reduce_chain<2><<<1, 64>>>(out, ticks, it2);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
        float ho[64]; hipMemcpy(ho, out, 256, hipMemcpyDeviceToHost);
        printf("reduce mode %d: %.1f shader cycles/iter, %.1f ns/iter, out[0]=%g out[40]=%g\n", mode, (double)h[0] / it2, ms * 1e6 / it2, ho[0], ho[40]);
    }
    return 0;
}
