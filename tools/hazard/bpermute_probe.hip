// bpermute_probe.hip -- does ds_bpermute_b32 read its ADDRESS register after the next instruction has overwritten it?
//
// in chunk_post the compiler emitted
//     ds_bpermute_b32 v5, v6, v3
//     v_or_b32_e32    v6, 1, v26            <- the permute's address register, re-used at once
// and the product that used it came out wrong in ~3 of 1400 workgroups, only while other workgroups kept the CU's LDS queue
// full of 16-byte reads.  This probe isolates the pattern: "victim" wavefronts issue the permute and overwrite its address
// register 0, 1, 2, 4 or 8 instruction slots later (s_nop in between), "load" wavefronts of the same workgroups hammer the
// LDS with ds_read_b128, and every wrong lane value is counted.
//
// OUTCOME (profiles/r04_hazard/r04_bpermute_probe.txt): 0 wrong lane values in every configuration -- the permute was not the
// cause; see tools/hazard/pk_opsel_probe.hip and DESIGN section 6.
//
//   hipcc --offload-arch=gfx950 -O3 -o bpermute_probe tools/hazard/bpermute_probe.hip && ./bpermute_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int GAP, bool MFMA>
__global__ __launch_bounds__(512) void probe(unsigned long long* wrong, unsigned long long* tried, int iters, int loaders) {
    __shared__ __attribute__((aligned(16))) float4 lds[2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = tid; q < 2048; q += 512) lds[q] = float4{(float)q, 1.f, 2.f, 3.f};
    __syncthreads();
    if (wave >= 8 - loaders) {                                   // load wavefronts: back-to-back 16-byte LDS reads feeding
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));  // f16 matrix instructions, as chunk_post's accumulation does
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const float4 va = lds[(lane + 64 * u + 17 * it) & 2047], vb = lds[(lane + 64 * u + 64 + 17 * it) & 2047];
                const h8 a = __builtin_bit_cast(h8, va), b = __builtin_bit_cast(h8, vb);
                if (MFMA) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, acc0, 0, 0, 0);
                } else {
                    acc0[0] += va.x + vb.y;
                    acc1[1] += va.z + vb.w;
                }
            }
        }
        if (acc0[0] + acc1[1] == -1.f) wrong[0] = 1;             // (keeps the loop)
        return;
    }
    unsigned long long bad = 0, n = 0;
    const int want_lane = lane ^ 16;
    for (int it = 0; it < iters; ++it) {
        const int value = (it << 8) | lane;                      // what every lane offers
        int addr = want_lane << 2, got, junk = 0x7ffffffc - (lane << 2);          // junk: another lane's address (masked to 0..252 by the hardware: & 0xfc)
        if (GAP == 0)
            asm volatile("ds_bpermute_b32 %0, %1, %2\n\tv_mov_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(got), "+v"(addr) : "v"(value), "v"(junk) : "memory");
        else if (GAP == 1)
            asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_nop 0\n\tv_mov_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(got), "+v"(addr) : "v"(value), "v"(junk) : "memory");
        else if (GAP == 2)
            asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_nop 1\n\tv_mov_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(got), "+v"(addr) : "v"(value), "v"(junk) : "memory");
        else if (GAP == 4)
            asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_nop 3\n\tv_mov_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(got), "+v"(addr) : "v"(value), "v"(junk) : "memory");
        else if (GAP == 8)
            asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_nop 7\n\tv_mov_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(got), "+v"(addr) : "v"(value), "v"(junk) : "memory");
        else                                                      // GAP < 0: the address register is left alone (control)
            asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(got), "+v"(addr) : "v"(value) : "memory");
        bad += got != ((it << 8) | want_lane);
        ++n;
    }
    atomicAdd(wrong + 1, bad);
    atomicAdd(tried, n);
}

template <int GAP, bool MFMA> void run(int loaders, const char* what) {
    unsigned long long *d_wrong, *d_tried, h[3] = {0, 0, 0};
    hipMalloc(&d_wrong, 16);
    hipMalloc(&d_tried, 8);
    hipMemset(d_wrong, 0, 16);
    hipMemset(d_tried, 0, 8);
    hipLaunchKernelGGL((probe<GAP, MFMA>), dim3(256 * 8), dim3(512), 0, 0, d_wrong, d_tried, 2000, loaders);
    hipDeviceSynchronize();
    hipMemcpy(h, d_wrong, 16, hipMemcpyDeviceToHost);
    hipMemcpy(h + 2, d_tried, 8, hipMemcpyDeviceToHost);
    printf("%-58s load waves per workgroup %d (%s): %llu wrong lane values of %llu\n", what, loaders, MFMA ? "LDS reads + f16 MFMA" : "LDS reads", h[1], h[2]);
    hipFree(d_wrong);
    hipFree(d_tried);
}

int main() {
    for (int loaders : {0, 4, 6}) {
        run<-1, false>(loaders, "address register left alone (control)");
        run<0, false>(loaders, "address register overwritten by the NEXT instruction");
        run<-1, true>(loaders, "address register left alone (control)");
        run<0, true>(loaders, "address register overwritten by the NEXT instruction");
        run<1, true>(loaders, "... one wait state later (s_nop 0)");
        run<2, true>(loaders, "... two wait states later (s_nop 1)");
        run<4, true>(loaders, "... four wait states later (s_nop 3)");
        run<8, true>(loaders, "... eight wait states later (s_nop 7)");
    }
    return 0;
}
