// pk_opsel_probe.hip -- packed-f32 vector instructions whose LOW half takes an operand from the HIGH register of a pair
// (op_sel bit set) while other wavefronts of the CU run matrix instructions: DESIGN section 6.
//
// tools/hazard/cut_sequence_probe.hip reproduced chunk_post's wrong sums outside the library and lds_return_probe.hip reduced
// them to ONE instruction, on registers that had been valid for a long time:
//     v_pk_fma_f32 v[6:7], v[54:55], v[10:11], v[6:7] op_sel:[0,1,0]        low half wrong in lanes 48-63, ~3e-5 of the time
// This probe runs the operand-select forms one by one, says what the wrong value was, and varies what the other
// wavefronts of the workgroup do: LDS reads, matrix instructions of every operand width (f16 / bf16 with K = 32 -- the
// 128-bit operands new on gfx950 --, f32, f64, the K = 16 f16 one, fp8), both, or nothing.
//
//   hipcc --offload-arch=gfx950 -O3 -o pk_opsel_probe tools/hazard/pk_opsel_probe.hip && ./pk_opsel_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ float val(unsigned it, unsigned a, unsigned b) {        // [0.5, 1)
    unsigned h = (it * 2654435761u) ^ (a * 40503u + 0x9e3779b9u) ^ (b * 2246822519u);
    h ^= h >> 15; h *= 2654435761u; h ^= h >> 13;
    return __builtin_bit_cast(float, 0x3f000000u | (h & 0x7fffffu));
}

// a = v[54:55], b = v[10:11], c = v[6:7]; result in v[6:7] (v[16:17] for the two-operand forms)
#define SETUP "v_mov_b32 v54, %2\n\tv_mov_b32 v55, %3\n\tv_mov_b32 v10, %4\n\tv_mov_b32 v11, %5\n\tv_mov_b32 v6, %6\n\tv_mov_b32 v7, %7\n\ts_nop 7\n\t"
#define HI_6 "7"
#define HI_16 "17"
#define HI(R) HI_##R
#define RUN(INSN, RES)                                                                                                  \
    asm volatile(SETUP INSN "\n\ts_nop 1\n\tv_mov_b32 %0, v" #RES "\n\tv_mov_b32 %1, v" HI(RES) "\n\t"                     \
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)                            \
                 : "memory", "v6", "v7", "v10", "v11", "v16", "v17", "v54", "v55")

template <int FORM, int LOADER>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int iters, int loaders) {
    __shared__ __attribute__((aligned(16))) float4 lds[2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = tid; q < 2048; q += 512) lds[q] = float4{(float)q, 1.f, 2.f, 3.f};
    __syncthreads();
    if (wave >= 8 - loaders) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        for (int it = 0; it < iters * 3; ++it) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const float4 va = LOADER == 2 ? float4{1.f, 2.f, 3.f, (float)(it + u)} : lds[(lane + 64 * u + 17 * it) & 2047];
                const float4 vb = LOADER == 2 ? float4{1.f, 2.f, (float)(it - u), 3.f} : lds[(lane + 64 * u + 64 + 17 * it) & 2047];
                const h8 a = __builtin_bit_cast(h8, va), b = __builtin_bit_cast(h8, vb);
                if (LOADER == 3) {                               // the exact-f32 matrix instruction of the fp32 path
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va.x, vb.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vb.y, va.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va.z, va.w, acc0, 0, 0, 0);
                } else if (LOADER == 4) {                        // ... and the f64 one of the fp64 path
                    typedef double d4 __attribute__((ext_vector_type(4)));
                    d4 c = {acc0[0], acc0[1], acc1[0], acc1[1]};
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)va.x, (double)vb.x, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)vb.y, (double)va.y, c, 0, 0, 0);
                    acc0[0] = (float)c[0]; acc0[1] = (float)c[1]; acc1[0] = (float)c[2]; acc1[1] = (float)c[3];
                } else if (LOADER == 5) {                        // bf16, same shape as the f16 one
                    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
                    const b8 ba = __builtin_bit_cast(b8, va), bb = __builtin_bit_cast(b8, vb);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb, ba, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, ba, acc0, 0, 0, 0);
                } else if (LOADER == 6) {                        // the f16 instruction of the older parts (K = 16)
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    const h4 ha = {a[0], a[1], a[2], a[3]}, hb = {b[0], b[1], b[2], b[3]};
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, hb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x16f16(hb, ha, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, ha, acc0, 0, 0, 0);
                } else if (LOADER == 7) {                        // fp8 (K = 32)
                    const long la = __builtin_bit_cast(long, va.x * 1.0 + va.y), lb = __builtin_bit_cast(long, vb.x * 1.0 + vb.y);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(lb, la, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, la, acc0, 0, 0, 0);
                } else if (LOADER != 1) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, acc0, 0, 0, 0);
                } else {
                    acc0[0] += va.x + vb.y;
                    acc1[1] += va.z + vb.w;
                }
            }
        }
        if (acc0[0] + acc1[1] == -1.f) out[0] = 1;               // (keeps the loop)
        return;
    }
    for (int it0 = 0; it0 < iters; ++it0) {
        const unsigned it = (unsigned)it0 * 1024u + blockIdx.x;
        const float a0 = val(it, lane, 11), a1 = val(it, lane, 12), b0 = val(it, lane, 15), b1 = val(it, lane, 16), c0 = val(it, lane, 13), c1 = val(it, lane, 14);
        float r0, r1, e0, e1, alt0, alt1;                        // e: what the ISA says; alt: the same with the select bits ignored
        if (FORM == 0) { RUN("v_pk_fma_f32 v[6:7], v[54:55], v[10:11], v[6:7]", 6); e0 = fmaf(a0, b0, c0); e1 = fmaf(a1, b1, c1); alt0 = e0; alt1 = e1; }
        if (FORM == 1) { RUN("v_pk_fma_f32 v[6:7], v[54:55], v[10:11], v[6:7] op_sel:[0,1,0]", 6); e0 = fmaf(a0, b1, c0); e1 = fmaf(a1, b1, c1); alt0 = fmaf(a0, b0, c0); alt1 = e1; }
        if (FORM == 2) { RUN("v_pk_fma_f32 v[6:7], v[54:55], v[10:11], v[6:7] op_sel:[1,0,0]", 6); e0 = fmaf(a1, b0, c0); e1 = fmaf(a1, b1, c1); alt0 = fmaf(a0, b0, c0); alt1 = e1; }
        if (FORM == 3) { RUN("v_pk_fma_f32 v[6:7], v[54:55], v[10:11], v[6:7] op_sel:[0,0,1] op_sel_hi:[1,1,0]", 6); e0 = fmaf(a0, b0, c1); e1 = fmaf(a1, b1, c0); alt0 = fmaf(a0, b0, c0); alt1 = fmaf(a1, b1, c1); }
        if (FORM == 4) { RUN("v_pk_fma_f32 v[6:7], v[54:55], v[10:11], v[6:7] op_sel_hi:[1,0,1]", 6); e0 = fmaf(a0, b0, c0); e1 = fmaf(a1, b0, c1); alt0 = e0; alt1 = fmaf(a1, b1, c1); }
        if (FORM == 5) { RUN("v_pk_mul_f32 v[16:17], v[54:55], v[10:11] op_sel:[0,1]", 16); e0 = a0 * b1; e1 = a1 * b1; alt0 = a0 * b0; alt1 = e1; }
        if (FORM == 6) { RUN("v_pk_add_f32 v[16:17], v[54:55], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]", 16); e0 = a0 + b1; e1 = a1 + b0; alt0 = a0 + b0; alt1 = a1 + b1; }
        if (FORM == 7) { RUN("v_pk_mul_f32 v[16:17], v[54:55], v[10:11] op_sel_hi:[1,0]", 16); e0 = a0 * b0; e1 = a1 * b0; alt0 = e0; alt1 = a1 * b1; }
        if (r0 != e0) {
            atomicAdd(&out[1], 1ull);
            atomicAdd(&out[8 + (lane >> 4)], 1ull);
            if (r0 == alt0) atomicAdd(&out[4], 1ull);            // as if the select bit had not been there
            if (out[6] == 0 && atomicAdd(&out[6], 1ull) == 0) {
                float* rec = reinterpret_cast<float*>(out + 16);
                rec[0] = r0; rec[1] = e0; rec[2] = alt0; rec[3] = a0; rec[4] = a1; rec[5] = b0; rec[6] = b1; rec[7] = c0; rec[8] = c1;
            }
        }
        if (r1 != e1) {
            atomicAdd(&out[2], 1ull);
            atomicAdd(&out[12 + (lane >> 4)], 1ull);
            if (r1 == alt1) atomicAdd(&out[5], 1ull);
        }
        atomicAdd(&out[3], (unsigned long long)(lane == 0 ? 64 : 0));
    }
}

template <int FORM, int LOADER> void run(int loaders, int iters) {
    static const char* form[] = {"v_pk_fma_f32 (no operand select: control)", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0]",
                                 "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 op_sel_hi:[1,0]"};
    static const char* loader[] = {"LDS reads + f16 MFMA", "LDS reads only", "f16 MFMA only", "LDS + f32 MFMA 16x16x4", "LDS + f64 MFMA 16x16x4", "LDS + bf16 MFMA 16x16x32", "LDS + f16 MFMA 16x16x16", "LDS + fp8 MFMA 16x16x32"};
    unsigned long long* d;
    hipMalloc(&d, 256);
    unsigned long long h[32], tot[32] = {0};
    float rec[9] = {0};
    for (int l = 0; l < 4; ++l) {
        hipMemset(d, 0, 256);
        hipLaunchKernelGGL((probe<FORM, LOADER>), dim3(256 * 4), dim3(512), 0, 0, d, iters, loaders);
        hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        for (int q = 0; q < 16; ++q) tot[q] += h[q];
        if (h[6] && rec[1] == 0) memcpy(rec, h + 16, sizeof(rec));
    }
    printf("%-46s | %d of 8 wavefronts: %-24s | low half wrong %llu (rows %llu %llu %llu %llu; = select ignored: %llu), high half wrong %llu (rows %llu %llu %llu %llu; = select ignored: %llu) of %llu\n",
           form[FORM], loaders, loader[LOADER], tot[1], tot[8], tot[9], tot[10], tot[11], tot[4], tot[2], tot[12], tot[13], tot[14], tot[15], tot[5], tot[3]);
    if (rec[1] != 0)
        printf("      e.g. low half got %.9g, want %.9g (select ignored: %.9g); a = (%.9g, %.9g) b = (%.9g, %.9g) c = (%.9g, %.9g)\n", rec[0], rec[1], rec[2], rec[3], rec[4], rec[5],
               rec[6], rec[7], rec[8]);
    fflush(stdout);
    hipFree(d);
}

int main() {
    run<1, 0>(0, 300);
    run<0, 0>(4, 300);
    run<1, 0>(4, 300);
    run<1, 1>(4, 300);
    run<1, 2>(4, 300);
    run<1, 0>(1, 300);
    run<1, 0>(2, 300);
    run<1, 0>(6, 300);
    run<1, 3>(4, 300);
    run<1, 4>(4, 300);
    run<6, 3>(4, 300);
    run<6, 4>(4, 300);
    run<1, 5>(4, 300);
    run<1, 6>(4, 300);
    run<1, 7>(4, 300);
    run<2, 0>(4, 300);
    run<3, 0>(4, 300);
    run<4, 0>(4, 300);
    run<5, 0>(4, 300);
    run<6, 0>(4, 300);
    run<7, 0>(4, 300);
    return 0;
}
