// lds_return_probe.hip -- narrowing down what tools/cut_sequence_probe.hip reproduces: in the failing
// stream the LOW half of the two packed FMAs that take their multiplier from the HIGH register of a pair
// (op_sel:[0,1,0]; the pair = dwords 0-1 of a ds_read_b128 the wave has just waited for) loses its product in lanes 48-63.
//
//   mode 0: is it the load?   ds_write_b64 (16 lanes) ; ds_read_b128 x 2 ; s_waitcnt lgkmcnt(1) ; v_mov of dwords 0..3 ;
//           s_waitcnt lgkmcnt(0) ; v_mov of dwords 4..7 -- destination registers pre-filled with a sentinel
//   mode 1: is it the instruction?   v_pk_fma_f32 d, a, b, c op_sel:[0,1,0] on registers that have been valid for long
//   mode 2: both: the loads as in mode 0, consumed by  v_pk_fma (plain) ; s_nop 0 ; v_pk_fma op_sel:[0,1,0]
//
// OUTCOME (profiles/r04_hazard/r04_lds_return_probe.txt): the loads alone never fail; the op_sel:[0,1,0] FMA alone does -- tools/hazard/pk_opsel_probe.hip takes it from there.
//
//   hipcc --offload-arch=gfx950 -O3 -o lds_return_probe tools/hazard/lds_return_probe.hip && ./lds_return_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float val(unsigned it, unsigned a, unsigned b) {        // [0.5, 1)
    unsigned h = (it * 2654435761u) ^ (a * 40503u + 0x9e3779b9u) ^ (b * 2246822519u);
    h ^= h >> 15; h *= 2654435761u; h ^= h >> 13;
    return __builtin_bit_cast(float, 0x3f000000u | (h & 0x7fffffu));
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int iters, int loaders) {
    __shared__ __attribute__((aligned(16))) float4 lds[2048];
    __shared__ __attribute__((aligned(16))) float mvw[8][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = tid; q < 2048; q += 512) lds[q] = float4{(float)q, 1.f, 2.f, 3.f};
    __syncthreads();
    if (wave >= 8 - loaders) {                                   // 16-byte LDS reads feeding f16 matrix instructions
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        for (int it = 0; it < iters * 12; ++it) {
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
    const int i16 = lane & 15, g4 = lane >> 4;
    const unsigned base = (unsigned)(size_t)(&mvw[wave][0]);
    const unsigned so_addr = base + i16 * 8, g_addr = base + g4 * 32;
    unsigned long long bad = 0, n = 0;
    for (int it0 = 0; it0 < iters; ++it0) {
        const unsigned it = (unsigned)it0 * 1024u + blockIdx.x;
        const float b0 = val(it, 1000 + 2 * i16, 7), b1 = val(it, 1001 + 2 * i16, 7);
        float w[8];
        for (int ii = 0; ii < 8; ++ii) w[ii] = val(it, 1000 + g4 * 8 + ii, 7);    // what this lane's two reads must return
        if (MODE == 0) {
            float r[8];
            asm volatile(
                "v_mov_b32 v28, %8\n\tv_mov_b32 v29, %9\n\tv_mov_b32 v90, %10\n\tv_mov_b32 v14, %11\n\tv_mov_b32 v1, %12\n\t"
                "v_mov_b32 v10, -1\n\tv_mov_b32 v11, -1\n\tv_mov_b32 v12, -1\n\tv_mov_b32 v13, -1\n\t"
                "v_mov_b32 v15, -1\n\tv_mov_b32 v16, -1\n\tv_mov_b32 v17, -1\n\ts_nop 4\n\t"
                "v_cmp_gt_u32_e32 vcc, 16, v1\n\t"
                "s_and_saveexec_b64 s[8:9], vcc\n\t"
                "ds_write_b64 v90, v[28:29]\n\t"
                "s_or_b64 exec, exec, s[8:9]\n\t"
                "ds_read_b128 v[10:13], v14\n\t"
                "ds_read_b128 v[14:17], v14 offset:16\n\t"
                "s_waitcnt lgkmcnt(1)\n\t"
                "v_mov_b32 %0, v10\n\tv_mov_b32 %1, v11\n\tv_mov_b32 %2, v12\n\tv_mov_b32 %3, v13\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_mov_b32 %4, v14\n\tv_mov_b32 %5, v15\n\tv_mov_b32 %6, v16\n\tv_mov_b32 %7, v17\n\t"
                : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                : "v"(b0), "v"(b1), "v"(so_addr), "v"(g_addr), "v"(lane)
                : "memory", "vcc", "s8", "s9", "v1", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v28", "v29", "v90");
            for (int ii = 0; ii < 8; ++ii)
                if (r[ii] != w[ii]) {
                    ++bad;
                    atomicAdd(&out[8 + ii], 1ull);                                   // which dword
                    atomicAdd(&out[16 + g4], 1ull);                                  // which row of lanes
                    if (__builtin_bit_cast(unsigned, r[ii]) == 0xffffffffu) atomicAdd(&out[4], 1ull);     // the sentinel
                }
        } else {
            const float o0 = val(it, lane, 11), o1 = val(it, lane, 12), c0 = val(it, lane, 13), c1 = val(it, lane, 14);
            float r0, r1;
            if (MODE == 1) {
                asm volatile(
                    "v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v54, %4\n\tv_mov_b32 v55, %5\n\tv_mov_b32 v6, %6\n\tv_mov_b32 v7, %7\n\ts_nop 7\n\t"
                    "v_pk_fma_f32 v[6:7], v[54:55], v[10:11], v[6:7] op_sel:[0,1,0]\n\ts_nop 1\n\t"
                    "v_mov_b32 %0, v6\n\tv_mov_b32 %1, v7\n\t"
                    : "=v"(r0), "=v"(r1)
                    : "v"(w[0]), "v"(w[1]), "v"(o0), "v"(o1), "v"(c0), "v"(c1)
                    : "memory", "v6", "v7", "v10", "v11", "v54", "v55");
            } else {
                asm volatile(
                    "v_mov_b32 v28, %2\n\tv_mov_b32 v29, %3\n\tv_mov_b32 v90, %4\n\tv_mov_b32 v14, %5\n\tv_mov_b32 v1, %6\n\t"
                    "v_mov_b32 v54, %7\n\tv_mov_b32 v55, %8\n\tv_mov_b32 v6, %9\n\tv_mov_b32 v7, %10\n\t"
                    "v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\ts_nop 4\n\t"
                    "v_cmp_gt_u32_e32 vcc, 16, v1\n\t"
                    "s_and_saveexec_b64 s[8:9], vcc\n\t"
                    "ds_write_b64 v90, v[28:29]\n\t"
                    "s_or_b64 exec, exec, s[8:9]\n\t"
                    "ds_read_b128 v[10:13], v14\n\t"
                    "ds_read_b128 v[14:17], v14 offset:16\n\t"
                    "s_waitcnt lgkmcnt(1)\n\t"
                    "v_pk_fma_f32 v[6:7], v[6:7], v[10:11], 0 op_sel_hi:[1,0,0]\n\t"
                    "s_nop 0\n\t"
                    "v_pk_fma_f32 v[6:7], v[54:55], v[10:11], v[6:7] op_sel:[0,1,0]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\ts_nop 1\n\t"
                    "v_mov_b32 %0, v6\n\tv_mov_b32 %1, v7\n\t"
                    : "=v"(r0), "=v"(r1)
                    : "v"(b0), "v"(b1), "v"(so_addr), "v"(g_addr), "v"(lane), "v"(o0), "v"(o1), "v"(c0), "v"(c1)
                    : "memory", "vcc", "s8", "s9", "v1", "v6", "v7", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v28", "v29", "v54", "v55", "v90");
            }
            const float e0 = MODE == 1 ? __builtin_fmaf(o0, w[1], c0) : __builtin_fmaf(o0, w[1], c0 * w[0]);
            const float e1 = MODE == 1 ? __builtin_fmaf(o1, w[1], c1) : __builtin_fmaf(o1, w[1], c1 * w[0]);
            if (r0 != e0 || r1 != e1) {
                ++bad;
                atomicAdd(&out[8 + (r0 != e0 ? 0 : 1)], 1ull);                       // low or high half
                atomicAdd(&out[16 + g4], 1ull);
            }
        }
        ++n;
    }
    atomicAdd(&out[1], bad);
    atomicAdd(&out[3], n);
}

template <int MODE> void run(int loaders, int iters) {
    static const char* mode[] = {"loads alone (sentinel in the destination registers)", "v_pk_fma_f32 op_sel:[0,1,0] alone", "loads consumed by the two packed FMAs"};
    unsigned long long* d;
    hipMalloc(&d, 256);
    unsigned long long h[32], tot[32] = {0};
    for (int l = 0; l < 6; ++l) {
        hipMemset(d, 0, 256);
        hipLaunchKernelGGL((probe<MODE>), dim3(256 * 4), dim3(512), 0, 0, d, iters, loaders);
        hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        for (int q = 0; q < 32; ++q) tot[q] += h[q];
    }
    printf("%-54s | %d of 8 wavefronts do LDS reads + MFMA: %llu wrong of %llu", mode[MODE], loaders, tot[1], tot[3] * (MODE == 0 ? 8 : 1));
    if (tot[1]) {
        if (MODE == 0) {
            printf("; by dword of the two reads:");
            for (int q = 0; q < 8; ++q) printf(" %llu", tot[8 + q]);
            printf("; sentinel seen %llu", tot[4]);
        } else {
            printf("; low half %llu, high half %llu", tot[8], tot[9]);
        }
        printf("; by row of 16 lanes: %llu %llu %llu %llu", tot[16], tot[17], tot[18], tot[19]);
    }
    printf("\n");
    fflush(stdout);
    hipFree(d);
}

int main() {
    for (int loaders : {0, 4}) {
        run<0>(loaders, 400);
        run<1>(loaders, 400);
        run<2>(loaders, 400);
    }
    return 0;
}
