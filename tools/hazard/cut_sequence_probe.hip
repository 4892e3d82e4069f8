// cut_sequence_probe.hip -- the instruction stream that failed in chunk_post, outside of chunk_post.
//
// The -DVBX_CUT_VIA_BPERMUTE build of the library (the product at the cut of the backward wave with __shfl_xor, as round 3
// compiled it) gives wrong sums in a few of 1400 workgroups per launch once the chip is full.  Isolated patterns that
// were suspected first -- the permute whose address register is overwritten in the next slot, a packed instruction
// feeding an LDS instruction in the next slot, an s_waitcnt as the only separator of two dependent packed FMAs
// (bpermute_probe / pk_forward_probe / pk_waitstate_probe) -- never failed.  This probe runs the WHOLE stream of the failing
// build, register for register, on fresh data every iteration, checks it against the same arithmetic in C++ and, for the
// wavefronts that come out wrong, works out which product is missing from the sum: always one of the two packed FMAs
// with op_sel:[0,1,0], in lanes 48-63, in the low half.  tools/hazard/pk_opsel_probe.hip then needs that instruction only.
//
//   hipcc --offload-arch=gfx950 -O3 -o cut_sequence_probe tools/hazard/cut_sequence_probe.hip && ./cut_sequence_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__device__ __forceinline__ float val(unsigned it, unsigned a, unsigned b) {        // [0.5, 1)
    unsigned h = (it * 2654435761u) ^ (a * 40503u + 0x9e3779b9u) ^ (b * 2246822519u);
    h ^= h >> 15; h *= 2654435761u; h ^= h >> 13;
    return __builtin_bit_cast(float, 0x3f000000u | (h & 0x7fffffu));
}
__device__ __forceinline__ float bnd(unsigned it, int j) { return val(it, 1000 + j, 7); }             // mv_w[j], j = 0..31
__device__ __forceinline__ float opv(unsigned it, int lane, int r, int ii) { return val(it, lane * 16 + r * 8 + ii, 3); }
__device__ __forceinline__ float tot_of(unsigned it, int lane, int r, int terms = 8) {   // what the packed FMAs compute for a lane
    const int g4 = lane >> 4;
    float t = 0.f;
    for (int ii = 0; ii < terms; ++ii) t = __builtin_fmaf(opv(it, lane, r, ii), bnd(it, g4 * 8 + ii), t);
    return t;
}

#define STREAM(P1, P2, P3, P4, P5, P6) \
        asm volatile( \
            "v_mov_b32 v28, %2\n\tv_mov_b32 v29, %3\n\t" \
            "v_mov_b32 v6, %4\n\tv_mov_b32 v7, %5\n\tv_mov_b32 v54, %6\n\tv_mov_b32 v55, %7\n\t" \
            "v_mov_b32 v8, %8\n\tv_mov_b32 v9, %9\n\tv_mov_b32 v52, %10\n\tv_mov_b32 v53, %11\n\t" \
            "v_mov_b32 v2, %12\n\tv_mov_b32 v3, %13\n\tv_mov_b32 v50, %14\n\tv_mov_b32 v51, %15\n\t" \
            "v_mov_b32 v4, %16\n\tv_mov_b32 v5, %17\n\tv_mov_b32 v48, %18\n\tv_mov_b32 v49, %19\n\t" \
            "v_mov_b32 v90, %20\n\tv_mov_b32 v14, %21\n\tv_mov_b32 v1, %22\n\t" \
            "v_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\tv_mov_b32 v26, 0\n\t" \
            "s_nop 4\n\t" \
            "v_cmp_gt_u32_e32 vcc, 16, v1\n\t" \
            "s_and_saveexec_b64 s[8:9], vcc\n\t" \
            "ds_write_b64 v90, v[28:29]\n\t" \
            "s_or_b64 exec, exec, s[8:9]\n\t" \
            "v_mbcnt_lo_u32_b32 v10, -1, 0\n\t" \
            "v_mbcnt_hi_u32_b32 v20, -1, v10\n\t" \
            "v_and_b32_e32 v10, 64, v20\n\t" \
            "v_add_u32_e32 v21, 64, v10\n\t" \
            "ds_read_b128 v[10:13], v14\n\t" \
            "v_xor_b32_e32 v15, 16, v20\n\t" \
            "v_cmp_lt_i32_e32 vcc, v15, v21\n\t" \
            "s_mov_b32 s8, 0xff800000\n\t" \
            "s_brev_b32 s28, 15\n\t" \
            "v_cndmask_b32_e32 v15, v20, v15, vcc\n\t" \
            "v_lshlrev_b32_e32 v22, 2, v15\n\t" \
            "ds_read_b128 v[14:17], v14 offset:16\n\t" \
            "s_waitcnt lgkmcnt(1)\n\t" \
            "v_pk_fma_f32 v[6:7], v[6:7], v[10:11], 0 op_sel_hi:[1,0,0]\n\t" \
            "s_nop 0\n\t" \
            "v_pk_fma_f32 v[6:7], v[54:55], v[10:11], v[6:7] op_sel:[0,1,0]\n\t" \
            "s_nop 0\n\t" \
            "v_pk_fma_f32 v[6:7], v[8:9], v[12:13], v[6:7] op_sel_hi:[1,0,1]\n\t" P1 \
            "v_mov_b32_e32 v8, v13\n\t" \
            "v_pk_fma_f32 v[6:7], v[52:53], v[8:9], v[6:7] op_sel_hi:[1,0,1]\n\t" \
            "s_waitcnt lgkmcnt(0)\n\t" P2 \
            "v_pk_fma_f32 v[2:3], v[2:3], v[14:15], v[6:7] op_sel_hi:[1,0,1]\n\t" P3 \
            "v_xor_b32_e32 v6, 32, v20\n\t" \
            "v_pk_fma_f32 v[2:3], v[50:51], v[14:15], v[2:3] op_sel:[0,1,0]\n\t" \
            "v_cmp_lt_i32_e32 vcc, v6, v21\n\t" \
            "v_pk_fma_f32 v[2:3], v[4:5], v[16:17], v[2:3] op_sel_hi:[1,0,1]\n\t" P4 \
            "v_mov_b32_e32 v4, v17\n\t" \
            "v_pk_fma_f32 v[2:3], v[48:49], v[4:5], v[2:3] op_sel_hi:[1,0,1]\n\t" P5 \
            "ds_bpermute_b32 v4, v22, v2\n\t" \
            "ds_bpermute_b32 v5, v22, v3\n\t" \
            "v_cndmask_b32_e32 v6, v20, v6, vcc\n\t" \
            "v_lshlrev_b32_e32 v6, 2, v6\n\t" \
            "v_cmp_lt_i32_e32 vcc, s8, v47\n\t" \
            "v_cmp_lt_i32_e64 s[8:9], s8, v46\n\t" \
            "s_waitcnt lgkmcnt(0)\n\t" \
            "v_pk_add_f32 v[2:3], v[2:3], v[4:5]\n\t" P6 \
            "ds_bpermute_b32 v4, v6, v2\n\t" \
            "ds_bpermute_b32 v5, v6, v3\n\t" \
            "v_bfrev_b32_e32 v7, 15\n\t" \
            "v_or_b32_e32 v6, 1, v26\n\t" \
            "s_waitcnt lgkmcnt(0)\n\t" \
            "v_pk_add_f32 v[2:3], v[2:3], v[4:5]\n\t" \
            "s_nop 1\n\t" \
            "v_mov_b32 %0, v2\n\tv_mov_b32 %1, v3\n\t" \
            : "=v"(r0), "=v"(r1) \
            : "v"(b0), "v"(b1), "v"(o[0][0]), "v"(o[1][0]), "v"(o[0][1]), "v"(o[1][1]), "v"(o[0][2]), "v"(o[1][2]), "v"(o[0][3]), "v"(o[1][3]), \
              "v"(o[0][4]), "v"(o[1][4]), "v"(o[0][5]), "v"(o[1][5]), "v"(o[0][6]), "v"(o[1][6]), "v"(o[0][7]), "v"(o[1][7]), \
              "v"(so_addr), "v"(g_addr), "v"(lane) \
            : "memory", "vcc", "s8", "s9", "s28", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", \
              "v16", "v17", "v20", "v21", "v22", "v26", "v28", "v29", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v90");

template <int VARIANT, int LOADER>
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
        for (int it = 0; it < iters * 24; ++it) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const float4 va = LOADER == 2 ? float4{1.f, 2.f, 3.f, (float)(it + u)} : lds[(lane + 64 * u + 17 * it) & 2047];
                const float4 vb = LOADER == 2 ? float4{1.f, 2.f, (float)(it - u), 3.f} : lds[(lane + 64 * u + 64 + 17 * it) & 2047];
                const h8 a = __builtin_bit_cast(h8, va), b = __builtin_bit_cast(h8, vb);
                if (LOADER != 1) {
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
    unsigned long long bad = 0, n = 0;
    const int i16 = lane & 15, g4 = lane >> 4;
    const unsigned base = (unsigned)(size_t)(&mvw[wave][0]);
    const unsigned so_addr = base + i16 * 8, g_addr = base + g4 * 32;
    for (int it0 = 0; it0 < iters; ++it0) {
        const unsigned it = (unsigned)it0 * 1024u + blockIdx.x;
        const float b0 = bnd(it, 2 * i16), b1 = bnd(it, 2 * i16 + 1);
        float o[2][8];
        for (int r = 0; r < 2; ++r)
            for (int ii = 0; ii < 8; ++ii) o[r][ii] = opv(it, lane, r, ii);
        float r0, r1;
        if (VARIANT == 0) STREAM("", "", "", "", "", "")
        if (VARIANT == 1) STREAM("", "s_nop 0\n\t", "", "", "", "")                      // a real wait state behind the s_waitcnt
        if (VARIANT == 2) STREAM("s_nop 0\n\t", "", "", "s_nop 0\n\t", "", "")             // before the v_mov that overwrites a source of the packed FMA in front of it
        if (VARIANT == 3) STREAM("", "", "s_nop 0\n\t", "", "", "")                      // before the v_xor that overwrites v6, src2 of the packed FMA in front of it
        if (VARIANT == 4) STREAM("", "", "", "", "s_nop 1\n\t", "s_nop 1\n\t")             // between the packed producers and the permutes
        if (VARIANT == 5) STREAM("s_nop 1\n\t", "s_nop 1\n\t", "s_nop 1\n\t", "s_nop 1\n\t", "s_nop 1\n\t", "s_nop 1\n\t")
        // the same arithmetic, lane by lane
        float w[2];
        for (int r = 0; r < 2; ++r) {
            const float s_a = tot_of(it, lane, r) + tot_of(it, lane ^ 16, r);
            const float s_b = tot_of(it, lane ^ 32, r) + tot_of(it, lane ^ 48, r);
            w[r] = s_a + s_b;
        }
        if (r0 != w[0] || r1 != w[1]) {
            ++bad;
            if (lane == 0) {                                     // keep what is needed to see WHICH value was off
                const unsigned long long k = atomicAdd(&out[2], 1ull);
                if (k < 4) {
                    float* rec = reinterpret_cast<float*>(out + 8) + k * 72;
                    rec[0] = r0; rec[1] = w[0]; rec[2] = r1; rec[3] = w[1];
                    for (int q = 0; q < 4; ++q)
                        for (int ii = 0; ii < 8; ++ii) {
                            rec[8 + q * 8 + ii] = opv(it, lane ^ (16 * q), 0, ii);       // the operator values of lanes 0 / 16 / 32 / 48, state 0
                            rec[40 + q * 8 + ii] = bnd(it, q * 8 + ii);                  // the vector as the four rows read it
                        }
                }
            }
        }
        ++n;
    }
    atomicAdd(&out[1], bad);
    atomicAdd(&out[3], n);
}

template <int VARIANT, int LOADER> void run(int loaders, int iters) {
    static const char* variant[] = {"as it failed", "+ s_nop 0 behind the s_waitcnt between two dependent packed FMAs", "+ s_nop 0 before the v_mov that overwrites a source of the packed FMA in front of it",
                                    "+ s_nop 0 before the v_xor that overwrites src2 of the packed FMA in front of it", "+ s_nop 1 between the packed producers and the permutes", "+ s_nop 1 in all six places"};
    static const char* loader[] = {"16-byte LDS reads + f16 MFMA", "16-byte LDS reads only", "f16 MFMA only"};
    unsigned long long* d;
    const size_t bytes = 64 + 4 * 72 * sizeof(float);
    hipMalloc(&d, bytes);
    unsigned long long wrong = 0, total = 0, launches_with = 0;
    unsigned long long h[8 + 4 * 36];
    const int launches = 6;
    for (int l = 0; l < launches; ++l) {
        hipMemset(d, 0, bytes);
        hipLaunchKernelGGL((probe<VARIANT, LOADER>), dim3(256 * 4), dim3(512), 0, 0, d, iters, loaders);
        hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
        wrong += h[1];
        total += h[3];
        launches_with += h[1] != 0;
        if (h[1] && launches_with == 1) {
            const float* rec = reinterpret_cast<const float*>(h + 8);
            for (unsigned long long k = 0; k < (h[2] < 4 ? h[2] : 4); ++k, rec += 72) {
                // rec: got / want (state 0), got / want (state 1), then the operator values of lanes 0 / 16 / 32 / 48 and the
                // vector as rows 0..3 read it: which single product, left out, explains the sum lane 0 got?
                const float* o = rec + 8;
                const float* w = rec + 40;
                int hit_q = -1, hit_ii = -1;
                for (int q = 0; q < 4 && hit_q < 0; ++q)
                    for (int skip = 0; skip < 8 && hit_q < 0; ++skip) {
                        float t[4];
                        for (int r = 0; r < 4; ++r) {
                            t[r] = 0.f;
                            for (int ii = 0; ii < 8; ++ii)
                                if (!(r == q && ii == skip)) t[r] = fmaf(o[r * 8 + ii], w[r * 8 + ii], t[r]);
                        }
                        if ((t[0] + t[1]) + (t[2] + t[3]) == rec[0]) { hit_q = q; hit_ii = skip; }
                    }
                printf("    a wrong wavefront, lane 0: state 0 got %.9g, want %.9g (state 1 got %.9g, want %.9g): ", rec[0], rec[1], rec[2], rec[3]);
                if (hit_q >= 0)
                    printf("exactly the sum without the product of term %d in lanes %d-%d (%s)\n", hit_ii, 16 * hit_q, 16 * hit_q + 15,
                           hit_ii == 1 || hit_ii == 5 ? "one of the two v_pk_fma_f32 ... op_sel:[0,1,0]" : "NOT one of the op_sel:[0,1,0] instructions");
                else
                    printf("not explained by one missing product\n");
            }
        }
    }
    printf("%-88s | %d of 8 wavefronts do %-28s: %llu wrong lane results of %llu in %d launches (%llu of them with wrong results)\n", variant[VARIANT], loaders, loader[LOADER], wrong,
           total, launches, launches_with);
    fflush(stdout);
    hipFree(d);
}

int main() {
    run<0, 0>(0, 100);
    run<0, 0>(4, 200);
    run<0, 0>(6, 200);
    run<0, 1>(4, 200);
    run<0, 2>(4, 200);
    run<5, 0>(4, 200);
    return 0;
}
