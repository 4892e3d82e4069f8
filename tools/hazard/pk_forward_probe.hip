// pk_forward_probe.hip -- does an LDS-queue or memory instruction that reads a VGPR in the instruction slot right after a
// PACKED-f32 vector instruction (v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32) wrote it see the new value?
//
// DESIGN section 6.  For a vector consumer the compiler puts one wait state between the two (the "s_nop 0" after every
// v_pk_*_f32 whose result the next vector instruction reads); for ds_* / global_* consumers it does not, and the sequence
// that failed in chunk_post was
//     v_pk_add_f32    v[2:3], v[2:3], v[4:5]
//     ds_bpermute_b32 v4, v6, v2              <- reads v2 in the next slot
//     ds_bpermute_b32 v5, v6, v3
// Victim wavefronts run that pair with 0 or 1 wait states in between, for three producers (packed add, packed fma, two
// plain v_add_f32 as the control) and three consumers (ds_bpermute_b32, ds_write_b64 + read back, global_store_dwordx2 +
// read back); the other wavefronts of the workgroup keep the SIMDs and the LDS busy as chunk_post's do.
//
// OUTCOME (profiles/r04_hazard/r04_pk_forward_probe.txt): 0 wrong in every configuration -- not the cause.
//
//   hipcc --offload-arch=gfx950 -O3 -o pk_forward_probe tools/hazard/pk_forward_probe.hip && ./pk_forward_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define PROD_PKADD "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n\t"
#define PROD_PKFMA "v_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[10:11]\n\t"
#define PROD_PLAIN "v_add_f32 v10, v10, v12\n\tv_add_f32 v11, v11, v13\n\t"
#define CONS_BPERM "ds_bpermute_b32 v16, %6, v10\n\tds_bpermute_b32 v17, %6, v11\n\ts_waitcnt lgkmcnt(0)\n\t"
#define CONS_DSWR "ds_write_b64 %7, v[10:11]\n\tds_read_b64 v[16:17], %7\n\ts_waitcnt lgkmcnt(0)\n\t"
#define CONS_GLOB "global_store_dwordx2 %8, v[10:11], off\n\ts_waitcnt vmcnt(0)\n\tglobal_load_dwordx2 v[16:17], %8, off sc0 sc1\n\ts_waitcnt vmcnt(0)\n\t"
#define SETUP "v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\tv_mov_b32 v14, 1.0\n\tv_mov_b32 v15, 1.0\n\ts_nop 7\n\t"
#define FINISH "v_mov_b32 %0, v16\n\tv_mov_b32 %1, v17\n\t"
#define RUN(PROD, GAPS, CONS)                                                                                           \
    asm volatile(SETUP PROD GAPS CONS FINISH                                                                            \
                 : "=v"(r0), "=v"(r1)                                                                                   \
                 : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(addr), "v"(lds_off), "v"(gptr)                               \
                 : "memory", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17")

template <int PROD, int CONS, int GAP>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, float2* scratch, int iters, int loaders) {
    __shared__ __attribute__((aligned(16))) float4 lds[2048];
    __shared__ float2 slot[512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = tid; q < 2048; q += 512) lds[q] = float4{(float)q, 1.f, 2.f, 3.f};
    __syncthreads();
    if (wave >= 8 - loaders) {                                   // 16-byte LDS reads feeding f16 matrix instructions
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        for (int it = 0; it < iters * 3; ++it) {
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
    const int src = CONS == 0 ? (lane ^ 16) : lane;              // whose values this lane must get back
    const int addr = src << 2;
    const unsigned lds_off = (unsigned)(size_t)(&slot[tid]);     // (LDS aperture offset = low 32 bits of the generic pointer's offset)
    float2* gptr = scratch + (size_t)blockIdx.x * 512 + tid;
    for (int it = 0; it < iters; ++it) {
        const float a0 = (float)((it & 1023) * 64 + lane), a1 = a0 + 0.5f, b0 = 1e6f + (float)lane, b1 = 2e6f + (float)lane;
        float r0, r1;
        if (PROD == 0 && GAP == 0) { if (CONS == 0) RUN(PROD_PKADD, "", CONS_BPERM); else if (CONS == 1) RUN(PROD_PKADD, "", CONS_DSWR); else RUN(PROD_PKADD, "", CONS_GLOB); }
        if (PROD == 0 && GAP == 1) { if (CONS == 0) RUN(PROD_PKADD, "s_nop 0\n\t", CONS_BPERM); else if (CONS == 1) RUN(PROD_PKADD, "s_nop 0\n\t", CONS_DSWR); else RUN(PROD_PKADD, "s_nop 0\n\t", CONS_GLOB); }
        if (PROD == 1 && GAP == 0) { if (CONS == 0) RUN(PROD_PKFMA, "", CONS_BPERM); else if (CONS == 1) RUN(PROD_PKFMA, "", CONS_DSWR); else RUN(PROD_PKFMA, "", CONS_GLOB); }
        if (PROD == 1 && GAP == 1) { if (CONS == 0) RUN(PROD_PKFMA, "s_nop 0\n\t", CONS_BPERM); else if (CONS == 1) RUN(PROD_PKFMA, "s_nop 0\n\t", CONS_DSWR); else RUN(PROD_PKFMA, "s_nop 0\n\t", CONS_GLOB); }
        if (PROD == 2) { if (CONS == 0) RUN(PROD_PLAIN, "", CONS_BPERM); else if (CONS == 1) RUN(PROD_PLAIN, "", CONS_DSWR); else RUN(PROD_PLAIN, "", CONS_GLOB); }
        // what the lane whose values come back held: old (a) and new (a + b, or b * 1 + a)
        const float sa0 = (float)((it & 1023) * 64 + src), sa1 = sa0 + 0.5f, sb0 = 1e6f + (float)src, sb1 = 2e6f + (float)src;
        const float w0 = sa0 + sb0, w1 = sa1 + sb1;
        if (r0 != w0 || r1 != w1) {
            ++bad;
            if ((r0 == sa0 || r0 == w0) && (r1 == sa1 || r1 == w1)) ++stale;     // the register as it was BEFORE the packed instruction
        }
        ++n;
    }
    atomicAdd(&out[1], bad);
    atomicAdd(&out[2], stale);
    atomicAdd(&out[3], n);
}

template <int PROD, int CONS, int GAP> void run(int loaders) {
    static const char* prod[] = {"v_pk_add_f32", "v_pk_fma_f32", "2 x v_add_f32 (control)"};
    static const char* cons[] = {"ds_bpermute_b32", "ds_write_b64", "global_store_dwordx2"};
    unsigned long long* d;
    float2* scratch;
    const int blocks = 256 * 4;
    hipMalloc(&d, 32);
    hipMalloc(&scratch, sizeof(float2) * 512 * blocks);
    hipMemset(d, 0, 32);
    hipLaunchKernelGGL((probe<PROD, CONS, GAP>), dim3(blocks), dim3(512), 0, 0, d, scratch, CONS == 2 ? 2000 : 8000, loaders);
    unsigned long long h[4];
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("%-24s -> %d wait state(s) -> %-21s | %d MFMA+LDS wavefronts of 8: %llu wrong of %llu (%llu of them = the value before the write)\n",
           prod[PROD], GAP, cons[CONS], loaders, h[1], h[3], h[2]);
    fflush(stdout);
    hipFree(d);
    hipFree(scratch);
}

template <int CONS> void all(int loaders) {
    run<2, CONS, 0>(loaders);
    run<0, CONS, 0>(loaders);
    run<0, CONS, 1>(loaders);
    run<1, CONS, 0>(loaders);
    run<1, CONS, 1>(loaders);
}

int main() {
    for (int loaders : {0, 4, 6}) {
        all<0>(loaders);
        all<1>(loaders);
        all<2>(loaders);
    }
    return 0;
}
