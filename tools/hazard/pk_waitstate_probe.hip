// pk_waitstate_probe.hip -- two DEPENDENT packed-f32 vector instructions: which separators are a wait state?
//
// DESIGN section 6.  The compiler keeps one wait state between a v_pk_*_f32 and a vector instruction that reads its
// result: an "s_nop 0" where the two would be adjacent.  In the code that failed in chunk_post the ONLY thing between two
// dependent v_pk_fma_f32 was an "s_waitcnt lgkmcnt(0)" -- counted as the wait state by the compiler, but a wait that
// is already satisfied when the wavefront gets there (because it was held up before) may not cost the hardware a cycle.
//     v_pk_fma_f32 v[6:7], v[52:53], v[8:9], v[6:7]
//     s_waitcnt    lgkmcnt(0)
//     v_pk_fma_f32 v[2:3], v[2:3], v[14:15], v[6:7]
// Victim wavefronts run producer / separator / consumer with nothing outstanding on any counter; the other wavefronts
// of the workgroup keep the SIMDs busy with matrix instructions and 16-byte LDS reads, as chunk_post's do.
//
// OUTCOME (profiles/r04_hazard/r04_pk_waitstate_probe.txt): 0 wrong in every configuration, adjacent dependent packed FMAs included -- not the cause.
//
//   hipcc --offload-arch=gfx950 -O3 -o pk_waitstate_probe tools/hazard/pk_waitstate_probe.hip && ./pk_waitstate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define SETUP "v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\tv_mov_b32 v14, 1.0\n\tv_mov_b32 v15, 1.0\n\tv_mov_b32 v20, 0\n\ts_nop 7\n\t"
#define PRODUCER "v_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[10:11]\n\t"          /* a + b */
#define CONSUMER "v_pk_fma_f32 v[16:17], v[12:13], v[14:15], v[10:11]\n\ts_nop 1\n\t" /* (a + b) + b */
#define FINISH "v_mov_b32 %0, v16\n\tv_mov_b32 %1, v17\n\t"
#define RUN(SEP)                                                                                                        \
    asm volatile(SETUP PRODUCER SEP CONSUMER FINISH                                                                     \
                 : "=v"(r0), "=v"(r1)                                                                                   \
                 : "v"(a0), "v"(a1), "v"(b0), "v"(b1)                                                                   \
                 : "memory", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20")

template <int SEP>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int iters, int loaders) {
    __shared__ __attribute__((aligned(16))) float4 lds[2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = tid; q < 2048; q += 512) lds[q] = float4{(float)q, 1.f, 2.f, 3.f};
    __syncthreads();
    if (wave >= 8 - loaders) {                                   // 16-byte LDS reads feeding f16 matrix instructions
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const float4 va = lds[(lane + 64 * u + 17 * it) & 2047], vb = lds[(lane + 64 * u + 64 + 17 * it) & 2047];
                const h8 a = __builtin_bit_cast(h8, va), b = __builtin_bit_cast(h8, vb);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, acc0, 0, 0, 0);
            }
        }
        if (acc0[0] + acc1[1] == -1.f) out[0] = 1;               // (keeps the loop)
        return;
    }
    unsigned long long bad = 0, stale = 0, n = 0;
    for (int it = 0; it < iters; ++it) {
        const float a0 = (float)((it & 1023) * 64 + lane), a1 = a0 + 0.5f, b0 = 1e6f + (float)lane, b1 = 2e6f + (float)lane;
        float r0, r1;
        if (SEP == 0) RUN("");                                            // adjacent: no wait state at all
        if (SEP == 1) RUN("s_nop 0\n\t");                                 // what the compiler inserts
        if (SEP == 2) RUN("s_waitcnt lgkmcnt(0)\n\t");                    // satisfied on arrival
        if (SEP == 3) RUN("s_waitcnt vmcnt(0)\n\t");
        if (SEP == 4) RUN("v_mov_b32 v20, v14\n\t");                      // an independent vector instruction
        if (SEP == 5) RUN("s_mov_b32 m0, m0\n\t");                        // an independent scalar instruction
        const float w0 = (a0 + b0) + b0, w1 = (a1 + b1) + b1;
        if (r0 != w0 || r1 != w1) {
            ++bad;
            if ((r0 == a0 + b0 || r0 == w0) && (r1 == a1 + b1 || r1 == w1)) ++stale;     // the consumer saw the register as it was before the producer
        }
        ++n;
    }
    atomicAdd(&out[1], bad);
    atomicAdd(&out[2], stale);
    atomicAdd(&out[3], n);
}

template <int SEP> void run(int loaders) {
    static const char* sep[] = {"nothing (adjacent)", "s_nop 0", "s_waitcnt lgkmcnt(0)", "s_waitcnt vmcnt(0)", "an independent v_mov_b32", "an independent s_mov_b32"};
    unsigned long long* d;
    hipMalloc(&d, 32);
    hipMemset(d, 0, 32);
    hipLaunchKernelGGL((probe<SEP>), dim3(256 * 4), dim3(512), 0, 0, d, 8000, loaders);
    unsigned long long h[4];
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("v_pk_fma_f32 -> %-26s -> dependent v_pk_fma_f32 | %d MFMA+LDS wavefronts of 8: %llu wrong of %llu (%llu = computed from the value before the producer)\n",
           sep[SEP], loaders, h[1], h[3], h[2]);
    fflush(stdout);
    hipFree(d);
}

int main() {
    for (int loaders : {0, 4, 6}) {
        run<0>(loaders);
        run<1>(loaders);
        run<2>(loaders);
        run<3>(loaders);
        run<4>(loaders);
        run<5>(loaders);
    }
    return 0;
}
