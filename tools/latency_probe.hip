// latency_probe.hip -- single-wave latencies on gfx950 that the scan kernels are bound by.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int MODE> __global__ void probe(float* out, long long* ticks, int iters) {
    __shared__ float lds[4096];
    float x = threadIdx.x * 1e-3f + 1.0f, y = x + 0.25f;
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (float)((i * 7 + 3) & 1023);
    __syncthreads();
    int idx = threadIdx.x;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP64(x = __builtin_fmaf(x, 0.999f, 0.001f);) }                       // dependent fma
        if (MODE == 1) { REP64(x = __builtin_fmaf(x, 0.999f, 0.001f); y = __builtin_fmaf(y, 0.998f, 0.002f);) }  // 2 chains
        if (MODE == 2) { REP64(x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false)); x *= 0.5f;) } // dpp add + mul
        if (MODE == 3) { REP64({ auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false); x = (__builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1])) * 0.5f; }) }
        if (MODE == 4) { REP64(x = __builtin_amdgcn_rcpf(x) + 0.5f;) }                           // rcp + add
        if (MODE == 5) { REP64(idx = (int)lds[idx & 4095];) idx &= 4095; x = idx; }               // dependent LDS read (+cvt)
        if (MODE == 6) { REP64(lds[threadIdx.x] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); x = lds[(threadIdx.x + 1) & 63] * 0.999f;) } // LDS write->read round trip
        if (MODE == 7) { REP64(x = (float)__builtin_amdgcn_frexp_expf(x) * 0.01f + 1.5f;) }      // frexp_exp + cvt + fma
        if (MODE == 8) { REP64(x = __builtin_amdgcn_ldexpf(x, -1) + 1.0f;) }
    }
    const long long c1 = clock64();
    if (threadIdx.x == 0) ticks[0] = c1 - c0;
    out[threadIdx.x] = x + y + idx;
}
int main() {
    float* out; long long* ticks; long long h;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&ticks, 16);
    const int iters = 2000;
    const char* names[] = {"dependent v_fma_f32", "two interleaved fma chains (per pair)", "dpp add + mul", "permlane16_swap + add + mul",
                           "v_rcp_f32 + add", "dependent ds_read_b32 (+cvt)", "ds_write -> ds_read round trip (+mul)", "frexp_exp + cvt + fma", "ldexp + add"};
    for (int mode = 0; mode < 9; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: probe<0><<<1, 64>>>(out, ticks, iters); break; case 1: probe<1><<<1, 64>>>(out, ticks, iters); break;
                case 2: probe<2><<<1, 64>>>(out, ticks, iters); break; case 3: probe<3><<<1, 64>>>(out, ticks, iters); break;
                case 4: probe<4><<<1, 64>>>(out, ticks, iters); break; case 5: probe<5><<<1, 64>>>(out, ticks, iters); break;
                case 6: probe<6><<<1, 64>>>(out, ticks, iters); break; case 7: probe<7><<<1, 64>>>(out, ticks, iters); break;
                case 8: probe<8><<<1, 64>>>(out, ticks, iters); break;
            }
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        printf("%-42s %7.1f cycles per unit\n", names[mode], (double)h / (iters * 64.0));
    }
    return 0;
}
