// valu_probe.hip -- VALU issue cost on gfx950 of the instructions the operator recursion is made of, per SIMD, with
// 1 / 2 / 4 waves resident on every SIMD (one workgroup of 256 / 512 / 1024 threads per CU, 256 workgroups):
//   v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32, v_add_f32 with a DPP operand, v_fma_f64, and one frame of the
//   operator recursion written with packed and with plain instructions (16 states per lane).
// Every variant runs N independent dependency chains per wave (N = 8) so that a single wave is not latency-bound.
// Prints SIMD cycles per wave-instruction (clock64 around the loop, x waves per SIMD / instructions).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_probe tools/valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2_t __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(1024) void probe(float* out, long long* ticks, int iters) {
    const float s = 1.0f + 1e-7f * threadIdx.x;
    float a[16];
    float2_t p[8];
    double d[8];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = s + k;
#pragma unroll
    for (int k = 0; k < 8; ++k) { p[k] = float2_t{s + k, s - k}; d[k] = s + k; }
    const float2_t c2 = float2_t{0.999f, 1.001f}, m2 = float2_t{0.9999f, 1.0001f};
    __syncthreads();
    const long long c0 = clock64();
    // (inline assembly: left to itself the compiler packs adjacent plain FMAs into v_pk_fma_f32 again)
#define FMA32(x, y, z) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define MUL32(x, y) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define ADD32(x, y) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define PKFMA(x, y, z) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define PKFMA3(x, y, z) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z))
#define PKMUL(x, y) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define PKADD(x, y) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define DPPADD(x) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x))
#define FMA64(x, y, z) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
    const float cs = 0.999f, ms = 0.001f;
    const double cd = 0.999, md = 0.001;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {                     // 16 independent v_fma_f32
#pragma unroll
            for (int k = 0; k < 16; ++k) FMA32(a[k], cs, ms);
        } else if (MODE == 1) {              // 8 independent v_pk_fma_f32 (16 FMAs)
#pragma unroll
            for (int k = 0; k < 8; ++k) PKFMA(p[k], c2, m2);
        } else if (MODE == 2) {              // 8 v_pk_mul_f32
#pragma unroll
            for (int k = 0; k < 8; ++k) PKMUL(p[k], c2);
        } else if (MODE == 3) {              // 8 v_pk_add_f32
#pragma unroll
            for (int k = 0; k < 8; ++k) PKADD(p[k], m2);
        } else if (MODE == 4) {              // 16 v_add_f32 with a DPP operand (quad_perm)
#pragma unroll
            for (int k = 0; k < 16; ++k) DPPADD(a[k]);
        } else if (MODE == 5) {              // 8 v_fma_f64
#pragma unroll
            for (int k = 0; k < 8; ++k) FMA64(d[k], cd, md);
        } else if (MODE == 6) {              // one operator frame, packed: x = b (c sig + x) on 8 pairs, pairwise sum tree
            float2_t v[8];
            float2_t sig2 = float2_t{a[0], a[0]};
#pragma unroll
            for (int k = 0; k < 8; ++k) { PKFMA3(p[k], c2, sig2); PKMUL(p[k], m2); v[k] = p[k]; }
#pragma unroll
            for (int w = 4; w >= 1; w >>= 1)
#pragma unroll
                for (int k = 0; k < w; ++k) PKADD(v[k], v[k + w]);
            float lo = v[0].x, hi = v[0].y;
            ADD32(lo, hi);
            DPPADD(lo);
            MUL32(lo, ms);
            a[0] = lo;
        } else if (MODE == 7) {              // the same frame with plain instructions: 16 fma, 16 mul, 15 add + 3
            float sig = a[15];
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(cs), "v"(sig)); MUL32(a[k], cs); v[k] = a[k]; }
#pragma unroll
            for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                for (int k = 0; k < w; ++k) ADD32(v[k], v[k + w]);
            float lo = v[0];
            DPPADD(lo);
            MUL32(lo, ms);
            a[15] = lo;
        }
    }
    const long long c1 = clock64();
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += a[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += p[k].x + p[k].y + (float)d[k];
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = c1 - c0;
    if (acc == 12345.678f) out[0] = acc;
}

// Do f32 MFMAs and packed-f32 VALU work of DIFFERENT waves overlap on one SIMD?  Even waves of a workgroup issue
// v_mfma_f32_16x16x4_f32 on 8 independent accumulators, odd waves v_pk_fma_f32 on 8 independent pairs; each role stamps
// its own loop.  ROLE 0: all waves MFMA, 1: all waves VALU, 2: mixed (half and half on every SIMD).
typedef float float4_t __attribute__((ext_vector_type(4)));
template <int ROLE> __global__ __launch_bounds__(1024) void probe_mix(float* out, long long* ticks, int iters) {
    const int wave = threadIdx.x >> 6;
    // waves go to SIMDs round-robin: waves w and w + 4 share a SIMD; mixed = waves 0-3 MFMA, 4-7 VALU, ...
    const bool mfma = ROLE == 0 || (ROLE == 2 && ((wave >> 2) & 1) == 0);
    const float s = 1.0f + 1e-7f * threadIdx.x;
    float4_t acc[8];
    float2_t p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc[k] = float4_t{s, s, s, s}; p[k] = float2_t{s + k, s - k}; }
    const float2_t c2 = float2_t{0.999f, 1.001f}, m2 = float2_t{0.9999f, 1.0001f};
    __syncthreads();
    const long long c0 = clock64();
    if (mfma) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(s, 0.5f, acc[k], 0, 0, 0);
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)            // 32 packed FMAs per iteration: about the issue time of 8 MFMAs
#pragma unroll
                for (int k = 0; k < 8; ++k) PKFMA(p[k], c2, m2);
        }
    }
    const long long c1 = clock64();
    float accs = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) accs += acc[k].x + acc[k].y + acc[k].z + acc[k].w + p[k].x + p[k].y;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) ticks[threadIdx.x ? 1 : 0] = c1 - c0;
    if (accs == 12345.678f) out[0] = accs;
}

int main() {
    float* out; long long* ticks; long long h;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&ticks, 16);
    const int iters = 4000;
    const char* names[] = {"v_fma_f32 x16", "v_pk_fma_f32 x8", "v_pk_mul_f32 x8", "v_pk_add_f32 x8", "v_add_f32 dpp x16", "v_fma_f64 x8",
                           "operator frame, packed (8 pk_fma + 8 pk_mul + 7 pk_add + 3)", "operator frame, plain (16 fma + 16 mul + 15 add + 3)"};
    const int ninst[] = {16, 8, 8, 8, 16, 8, 26, 50};
    for (int mode = 0; mode < 8; ++mode) {
        printf("%-62s", names[mode]);
        for (int wps = 1; wps <= 4; wps *= 2) {                 // waves per SIMD
            const int threads = 256 * wps;
            for (int rep = 0; rep < 2; ++rep) {
                switch (mode) {
                    case 0: probe<0><<<256, threads>>>(out, ticks, iters); break; case 1: probe<1><<<256, threads>>>(out, ticks, iters); break;
                    case 2: probe<2><<<256, threads>>>(out, ticks, iters); break; case 3: probe<3><<<256, threads>>>(out, ticks, iters); break;
                    case 4: probe<4><<<256, threads>>>(out, ticks, iters); break; case 5: probe<5><<<256, threads>>>(out, ticks, iters); break;
                    case 6: probe<6><<<256, threads>>>(out, ticks, iters); break; case 7: probe<7><<<256, threads>>>(out, ticks, iters); break;
                }
                (void)hipDeviceSynchronize();
            }
            (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
            // cycles the SIMD spends per wave-instruction: elapsed / (instructions per wave x waves on the SIMD)
            printf("  %d w/SIMD: %6.2f cyc/inst (%7.1f cyc/iter/wave)", wps, (double)h / ((double)iters * ninst[mode] * wps), (double)h / iters);
        }
        printf("\n");
    }
    printf("\nf32 MFMA 16x16x4 (8 per iteration) and v_pk_fma_f32 (32 per iteration) of different waves on one SIMD, 1024-thread workgroups "
           "(4 waves per SIMD):\n");
    long long h2[2];
    const char* rn[] = {"all 16 waves MFMA", "all 16 waves packed VALU", "waves 0-3, 8-11 MFMA / waves 4-7, 12-15 VALU"};
    for (int role = 0; role < 3; ++role) {
        for (int rep = 0; rep < 2; ++rep) {
            if (role == 0) probe_mix<0><<<256, 1024>>>(out, ticks, 2000);
            if (role == 1) probe_mix<1><<<256, 1024>>>(out, ticks, 2000);
            if (role == 2) probe_mix<2><<<256, 1024>>>(out, ticks, 2000);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h2, ticks, 16, hipMemcpyDeviceToHost);
        printf("  %-48s wave 0: %8.1f ticks per iteration   wave 4: %8.1f ticks per iteration\n", rn[role], (double)h2[0] / 2000, (double)h2[1] / 2000);
    }
    return 0;
}
