// vbx_split.hpp -- the two GEMM-shaped contractions of an iteration (rho alpha^T, VBx.py:97; gamma^T rho, VBx.py:96) on
// the f16 matrix cores with error-compensated operands ("split" mode of the fp32 path, VBX_OPT_GEMM).
//
// v_mfma_f32_16x16x4_f32 is exact f32 but runs at the f32 VECTOR rate (64 FLOP / clk / SIMD: 32 cycles per instruction,
// 1/16 of the f16 / bf16 matrix rate) and does not overlap with vector instructions on a SIMD (NOTES.md round 3,
// tools/valu_probe.hip): 27-30 % of the SIMD cycles of both per-chunk kernels.  Here every f32 operand x is carried as
// two f16 values,
//     x 2^e = hi + lo,   hi = f16(x 2^e),   lo = f16(x 2^e - hi)
// (22 significant bits; e an exact power-of-two scale chosen per recording / per speaker so that the largest magnitude
// sits in [2^13, 2^14): hi never overflows and lo stays a normal f16 number for everything within 2^-17 of the largest),
// and a product is three v_mfma_f32_16x16x32_f16 with f32 accumulation,
//     a b ~ hi_a lo_b + lo_a hi_b + hi_a hi_b      (the dropped lo lo term is 2^-22 of the product)
// i.e. 3 x 16 cycles per K = 32 against 8 x 32: a fifth of the matrix cycles, and they run on the matrix cores proper.
// The operands that do not change over the iterations are split ONCE: rho lives in HBM a second and a third time as f16
// pairs in MFMA fragment order -- 4 bytes per element like the f32 copy, so the kernels' traffic is unchanged:
//     rho_a  A operand of rho alpha^T (rows = frames, k = feature dims)      chunk_loglik
//     rho_b  B operand of gamma^T rho (k = frames, columns = feature dims)   chunk_post
// alpha is split by fin_kernel when it writes the model (per-speaker scale), gamma by chunk_post where it is computed
// (scale 2^14: gamma <= 1).
//
// Fragment order (one v_mfma_f32_16x16x32_f16 operand = 8 halfs per lane = one 16-byte load; lane = 16 g + i):
//   A: row i, k-slots (g, e), e = 0..7        B: column i, k-slots (g, e)
// Which k a slot (kk, g, e) stands for is free as long as A and B agree (the sum over k is order-free):
//   rho_a / alpha : feature dim  d = 32 kk + 8 g + e
//   rho_b / gamma : frame        f = 16 e + 4 kk + g      (the eight frames one thread of chunk_post's posterior pass owns
//                                                          are its own operand: no transposition, one 16-byte LDS store)
#pragma once
#include "vbx_scan.hpp"

namespace vbx {

__device__ __forceinline__ f4 mfma_h(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// (small terms first; VBX_SPLIT_LOLO=1 adds the lo lo term, 2^-22 of a product at most: A/B builds)
#ifndef VBX_SPLIT_LOLO
#define VBX_SPLIT_LOLO 0
#endif
// rho alpha^T: the B operand (alpha) carries a third term -- alpha multiplies every frame, its representation error is
// systematic (fin_kernel) -- while the A operand's error is independent from frame to frame and averages out
__device__ __forceinline__ f4 mfma_split3(h8 ah, h8 al, h8 bh, h8 bl, h8 bl2, f4 c) {
    c = mfma_h(ah, bl2, c);
    c = mfma_h(ah, bl, c);
    c = mfma_h(al, bh, c);
    return mfma_h(ah, bh, c);
}
__device__ __forceinline__ f4 mfma_split(h8 ah, h8 al, h8 bh, h8 bl, f4 c) {
#if VBX_SPLIT_LOLO
    c = mfma_h(al, bl, c);
#endif
    c = mfma_h(ah, bl, c);
    c = mfma_h(al, bh, c);
    return mfma_h(ah, bh, c);
}

// largest |rho| of a recording -> amax[0], and the smallest over its frames of the frame's largest |rho| -> amax[1] (bits of
// a non-negative float order like integers; all-zero frames do not count); grid = tiles of the recording.  The pair is the
// dynamic range the ONE power-of-two scale of a recording has to cover: a frame whose largest element sits more than
// kSplitRangeBits below the recording's largest would carry its lo halves as f16 subnormals (fewer than the 22 bits the
// mode promises), so such a batch multiplies exactly instead (prepare_split, vbx_host_batch.hpp).  NaN / Inf: fmaxf drops a NaN, so
// the scale comes from the finite elements and the NaN itself reaches the matrix cores as an f16 NaN -- the result is NaN
// in both modes.
constexpr int kSplitRangeBits = 10;
__global__ __launch_bounds__(256) void rho_absmax_kernel(const float* __restrict__ rho, int T, int Dp, int* __restrict__ amax) {
    const int rows = min(kTileFrames, T - (int)blockIdx.x * kTileFrames);
    const float* __restrict__ src = rho + (long long)blockIdx.x * kTileFrames * Dp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float m = 0.0f, lo = __builtin_huge_valf();
    for (int r = wave; r < rows; r += 4) {             // a wave per frame
        float rm = 0.0f;
        for (int d = lane; d < Dp; d += 64) rm = fmaxf(rm, fabsf(src[(long long)r * Dp + d]));
        rm = allreduce_max<64>(rm);
        m = fmaxf(m, rm);
        if (rm > 0.0f) lo = fminf(lo, rm);
    }
    if (lane == 0) {
        if (m > 0.0f) atomicMax(amax, __builtin_bit_cast(int, m));
        if (lo < __builtin_huge_valf()) atomicMin(amax + 1, __builtin_bit_cast(int, lo));
    }
}

// rho (f32, [T][Dp]) of one recording -> its tiles of rho_a and rho_b; grid = tiles of the recording, block = 256.
// A tile of either is kTileFrames x Dp x 2 halfs; frames past the end of the recording are zero.
//   rho_a tile: [m-tile 0..7][kk 0..Dp/32-1][hi | lo][lane][8]   frame 16 mt + i, dim 32 kk + 8 g + e
//   rho_b tile: [slab 0..Dp/32-1][h 0..1][kk 0..3][hi | lo][lane][8]   dim 32 slab + 2 i + h, frame 16 e + 4 kk + g
// (the column relabelling of rho_b -- dims 2 i and 2 i + 1 in the two N-tiles of a slab -- gives chunk_post 8-byte stores)
__global__ __launch_bounds__(256) void rho_split_kernel(const float* __restrict__ rho, int T, int Dp, const int* __restrict__ amax,
                                                        int* __restrict__ rho_e, _Float16* __restrict__ rho_a,
                                                        _Float16* __restrict__ rho_b) {
    const int e2 = split_exponent(__builtin_bit_cast(float, *amax));
    if (blockIdx.x == 0 && threadIdx.x == 0) *rho_e = e2;
    const int t0 = blockIdx.x * kTileFrames, KK = Dp >> 5;
    const long long tile_halfs = (long long)kTileFrames * Dp * 2;
    _Float16* __restrict__ ta = rho_a + blockIdx.x * tile_halfs;
    _Float16* __restrict__ tb = rho_b + blockIdx.x * tile_halfs;
    const int nitems = 8 * KK * 64;                    // (m-tile, kk, lane) resp. (slab, h, kk, lane): the same count
    for (int it = threadIdx.x; it < nitems; it += 256) {
        const int lane = it & 63, i = lane & 15, g = lane >> 4;
        {
            const int kk = (it >> 6) % KK, mt = (it >> 6) / KK;
            const int f = t0 + 16 * mt + i;
            h8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = f < T ? scale2(rho[(long long)f * Dp + 32 * kk + 8 * g + e], e2) : 0.0f;
                _Float16 a, b;
                split_f16(v, a, b);
                hi[e] = a;
                lo[e] = b;
            }
            h8* dst = reinterpret_cast<h8*>(ta) + ((long long)(mt * KK + kk) * 2) * 64 + lane;
            dst[0] = hi;
            dst[64] = lo;
        }
        {
            const int kk = (it >> 6) & 3, sh = (it >> 8);          // sh = 2 slab + h
            const int d = 32 * (sh >> 1) + 2 * i + (sh & 1);
            h8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int f = t0 + 16 * e + 4 * kk + g;
                const float v = f < T ? scale2(rho[(long long)f * Dp + d], e2) : 0.0f;
                _Float16 a, b;
                split_f16(v, a, b);
                hi[e] = a;
                lo[e] = b;
            }
            h8* dst = reinterpret_cast<h8*>(tb) + ((long long)(sh * 4 + kk) * 2) * 64 + lane;
            dst[0] = hi;
            dst[64] = lo;
        }
    }
}

}  // namespace vbx
