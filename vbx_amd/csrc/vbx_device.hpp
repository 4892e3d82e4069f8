// vbx_device.hpp -- wavefront-level building blocks for gfx950 (CDNA4, wave64).
//
// Nothing here is portable HIP on purpose: 64-lane wavefronts, DPP row operations and the
// f32/f64 16x16x4 MFMA fragment layouts of gfx950 are hard-wired.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

// Build with -DVBX_PHASE_CLOCKS to make the two per-chunk kernels stamp the shader clock at every phase boundary
// (the per-phase cycle counts quoted in DESIGN.md come from such builds; tools/phase_timeline.py).
#ifdef VBX_PHASE_CLOCKS
#define VBX_CLOCKS_DECL() long long clk[8]; int nclk = 0
#define VBX_STAMP() do { if (nclk < 8) clk[nclk++] = clock64(); } while (0)
#else
#define VBX_CLOCKS_DECL()
#define VBX_STAMP()
#endif

namespace vbx {

constexpr int kWave = 64;
constexpr int kTileFrames = 128;  // frames handled by one workgroup tile

// ---------------------------------------------------------------------------------------
// small vector types
// ---------------------------------------------------------------------------------------
template <typename R> struct Vec;
template <> struct Vec<float> {
    using v2 = float __attribute__((ext_vector_type(2)));
    using v4 = float __attribute__((ext_vector_type(4)));
};
template <> struct Vec<double> {
    using v2 = double __attribute__((ext_vector_type(2)));
    using v4 = double __attribute__((ext_vector_type(4)));
};

// N adjacent values of one lane (N = 1, 2 or 4) moved as ONE LDS / global access of 4N or 8N bytes
template <typename R, int N> struct Pack;
template <typename R> struct Pack<R, 1> { using type = R; };
template <typename R> struct Pack<R, 2> { using type = typename Vec<R>::v2; };
template <typename R> struct Pack<R, 4> { using type = typename Vec<R>::v4; };
template <int N, typename R> __device__ __forceinline__ void load_pack(R (&dst)[N], const R* src) {
    using P = typename Pack<R, N>::type;
    const P v = *reinterpret_cast<const P*>(src);
    if constexpr (N == 1) dst[0] = v;
    else {
#pragma unroll
        for (int r = 0; r < N; ++r) dst[r] = v[r];
    }
}
template <int N, typename R> __device__ __forceinline__ void store_pack(R* dst, const R (&src)[N]) {
    using P = typename Pack<R, N>::type;
    P v;
    if constexpr (N == 1) v = src[0];
    else {
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] = src[r];
    }
    *reinterpret_cast<P*>(dst) = v;
}

// ---------------------------------------------------------------------------------------
// MFMA 16x16x4, f32 and f64.  One A value and one B value per lane:
//   A[i = lane & 15][k = lane >> 4],  B[k = lane >> 4][j = lane & 15]
// C/D: col = lane & 15; row = 4*(lane>>4) + reg (f32)  |  (lane>>4) + 4*reg (f64).
// The f32 form is an exact k-ordered fmaf chain (157 TF peak); f64 runs at 78.6 TF.
// ---------------------------------------------------------------------------------------
template <typename R> struct Mfma16;
template <> struct Mfma16<float> {
    using acc_t = Vec<float>::v4;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int reg) { return ((lane >> 4) << 2) + reg; }
};
template <> struct Mfma16<double> {
    using acc_t = Vec<double>::v4;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + (reg << 2); }
};

// ---------------------------------------------------------------------------------------
// DPP lane exchange (no LDS traffic).  All 64 lanes must be active at the call site.
//   0xB1 quad_perm(1,0,3,2) = xor 1      0x4E quad_perm(2,3,0,1) = xor 2
//   0x141 row_half_mirror (i -> 7-i)     0x140 row_mirror (i -> 15-i)
// After the two quad steps every lane of a quad holds the quad sum, so the mirrors act as
// xor 4 and xor 8 inside a butterfly.
// ---------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffLL), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, 0xF, 0xF, false);
    const long long out = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(double, out);
}

// Cross-row exchange on the VALU (gfx950: v_permlane16_swap / v_permlane32_swap), ~3x cheaper
// than a ds_bpermute round trip (measured: 4 DPP + swap = 56 cycles vs 92 with ds_bpermute).
//   permlane16_swap(v, v) -> { [r0 r0 r2 r2], [r1 r1 r3 r3] }   (rows of 16 lanes)
//   permlane32_swap(v, v) -> { [lo lo], [hi hi] }               (halves of 32 lanes)
// Both results combined with a commutative op give the xor-16 / xor-32 butterfly stage.
struct SwapPair { unsigned a, b; };
__device__ __forceinline__ SwapPair swap16(unsigned v) {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return SwapPair{r[0], r[1]};
}
__device__ __forceinline__ SwapPair swap32(unsigned v) {
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return SwapPair{r[0], r[1]};
}
template <int STAGE> __device__ __forceinline__ void cross_rows(float v, float& x, float& y) {
    const SwapPair p = STAGE == 16 ? swap16(__builtin_bit_cast(unsigned, v)) : swap32(__builtin_bit_cast(unsigned, v));
    x = __builtin_bit_cast(float, p.a);
    y = __builtin_bit_cast(float, p.b);
}
template <int STAGE> __device__ __forceinline__ void cross_rows(int v, int& x, int& y) {
    const SwapPair p = STAGE == 16 ? swap16((unsigned)v) : swap32((unsigned)v);
    x = (int)p.a;
    y = (int)p.b;
}
template <int STAGE> __device__ __forceinline__ void cross_rows(double v, double& x, double& y) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
    const SwapPair lo = STAGE == 16 ? swap16((unsigned)bits) : swap32((unsigned)bits);
    const SwapPair hi = STAGE == 16 ? swap16((unsigned)(bits >> 32)) : swap32((unsigned)(bits >> 32));
    x = __builtin_bit_cast(double, ((unsigned long long)hi.a << 32) | lo.a);
    y = __builtin_bit_cast(double, ((unsigned long long)hi.b << 32) | lo.b);
}
template <int CTRL> __device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}

// v + (the value lane ^ STAGE holds), STAGE = 16 or 32, on the VALU (v_permlane16/32_swap) instead of __shfl_xor's
// ds_bpermute_b32 round trip through the LDS queue (56 against 92 cycles for a 64-lane butterfly, see above).
// (Round 4 took the permute for the cause of chunk_post's wrong sums for a while; it is not -- DESIGN section 6: the
//  cause is a packed-f32 operand form the compiler happened to emit next to it, which vbx_amd/build.py now rejects.)
// A value in a register of its own.  A multiplier that is the ODD element of a pair loaded from LDS or memory invites the
// compiler to broadcast it with "v_pk_fma_f32 ... op_sel:[0,1,0]" -- the packed form that misreads src1 on gfx950 beside
// the K = 32 f16 matrix instructions (vbx_amd/build.py refuses a library that holds it).  Passing the
// multiplier through here makes it an opaque register; the broadcast then is the harmless op_sel_hi form or a plain v_fma.
template <typename R> __device__ __forceinline__ R lone_register(R v) {
    asm("" : "+v"(v));
    return v;
}

// (-DVBX_XOR_VIA_BPERMUTE: the round-3 code generation again, for tools/hazard/bpermute_compare.py)
template <int STAGE, typename T> __device__ __forceinline__ T add_xor(T v) {
#ifdef VBX_XOR_VIA_BPERMUTE
    return v + __shfl_xor(v, STAGE, 64);
#else
    T a, b;
    cross_rows<STAGE>(v, a, b);
    return a + b;
#endif
}
template <int STAGE> __device__ __forceinline__ int max_xor(int v) {
#ifdef VBX_XOR_VIA_BPERMUTE
    const int o = __shfl_xor(v, STAGE, 64);
    return v > o ? v : o;
#else
    int a, b;
    cross_rows<STAGE>(v, a, b);
    return a > b ? a : b;
#endif
}

// value held by one lane, as a wave-uniform scalar (v_readlane_b32 -> SGPR)
__device__ __forceinline__ float read_lane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double read_lane(double v, int lane) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// (one v_max_* each -- a compare + select would also keep the DPP move of a butterfly stage from folding into it)
__device__ __forceinline__ float vmax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double vmax(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ int vmax(int a, int b) { return a > b ? a : b; }

// Butterfly all-reduce over aligned groups of W lanes (W = 16, 32 or 64): every lane of a
// group ends up with the group's result (identical bits in every lane: each stage combines a
// symmetric pair).
template <int W, typename R> __device__ __forceinline__ R allreduce_sum(R v) {
    static_assert(W == 16 || W == 32 || W == 64, "group width");
#ifdef VBX_NO_DPP
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    if (W >= 32) v += __shfl_xor(v, 16, 64);
    if (W >= 64) v += __shfl_xor(v, 32, 64);
#else
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    if (W >= 32) { R a, b; cross_rows<16>(v, a, b); v = a + b; }
    if (W >= 64) { R a, b; cross_rows<32>(v, a, b); v = a + b; }
#endif
    return v;
}

template <int W, typename R> __device__ __forceinline__ R allreduce_max_impl(R v);

// f32: the butterfly runs on an order-preserving integer image of the value (sign-magnitude -> two's complement: one
// v_max_i32_dpp per stage; on floats every stage is a DPP move + a canonicalising max of the moved operand + the max)
template <int W, typename R> __device__ __forceinline__ R allreduce_max(R v) {
    if constexpr (std::is_same<R, float>::value) {
        int k = __builtin_bit_cast(int, v);
        k ^= (k >> 31) & 0x7fffffff;
        k = allreduce_max_impl<W, int>(k);
        k ^= (k >> 31) & 0x7fffffff;
        return __builtin_bit_cast(float, k);
    } else {
        return allreduce_max_impl<W, R>(v);
    }
}

template <int W, typename R> __device__ __forceinline__ R allreduce_max_impl(R v) {
    static_assert(W == 16 || W == 32 || W == 64, "group width");
#ifdef VBX_NO_DPP
    v = vmax(v, __shfl_xor(v, 1, 64));
    v = vmax(v, __shfl_xor(v, 2, 64));
    v = vmax(v, __shfl_xor(v, 4, 64));
    v = vmax(v, __shfl_xor(v, 8, 64));
    if (W >= 32) v = vmax(v, __shfl_xor(v, 16, 64));
    if (W >= 64) v = vmax(v, __shfl_xor(v, 32, 64));
#else
    v = vmax(v, dpp_mov<0xB1>(v));
    v = vmax(v, dpp_mov<0x4E>(v));
    v = vmax(v, dpp_mov<0x141>(v));
    v = vmax(v, dpp_mov<0x140>(v));
    if (W >= 32) { R a, b; cross_rows<16>(v, a, b); v = vmax(a, b); }
    if (W >= 64) { R a, b; cross_rows<32>(v, a, b); v = vmax(a, b); }
#endif
    return v;
}

// ---------------------------------------------------------------------------------------
// scalar math per arithmetic type
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // 1 ulp
__device__ __forceinline__ double fast_rcp(double x) { return 1.0 / x; }
// exp of a non-positive argument: v_exp_f32 on x*log2(e).  The multiplication costs |x| * 2^-24 of relative
// accuracy, i.e. 1e-6 for likelihoods 1e-7 times the row maximum -- far inside the 1e-4 budget of the f32 path.
__device__ __forceinline__ float exp_r(float x) { return __expf(x); }
__device__ __forceinline__ double exp_r(double x) { return exp(x); }
template <typename R> __device__ __forceinline__ R neg_inf() { return -(R)INFINITY; }

// Copy `count` vectors from global memory to LDS with N independent loads in flight per thread.
// A plain `dst[q] = src[q]` loop waits for every load before issuing the next one: at ~0.6 us per
// round trip (data written by the previous kernel usually sits behind another XCD's L2) that
// serialisation was the largest single cost of the first scan kernels.
template <int N, typename V>
__device__ __forceinline__ void stage_to_lds(V* dst, const V* __restrict__ src, int count, int tid, int nthreads) {
    for (int base = 0; base < count; base += N * nthreads) {
        V tmp[N];
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const int idx = base + u * nthreads + tid;
            if (idx < count) tmp[u] = src[idx];
        }
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const int idx = base + u * nthreads + tid;
            if (idx < count) dst[idx] = tmp[u];
        }
    }
}

// Running product with an integer exponent on the side: prod * 2^expo, renormalised now and
// then so that thousands of per-frame scales can be multiplied without a log per frame.
struct ScaledProduct {
    double mant = 1.0;
    long long expo = 0;
    __device__ __forceinline__ void mul(double s) { mant *= s; }
    __device__ __forceinline__ void renorm() {
        int e;
        mant = frexp(mant, &e);
        expo += e;
    }
    __device__ __forceinline__ double log_value() const {
        return log(mant) + (double)expo * 0.69314718055994530942;
    }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release / acquire over ALL address
// spaces: with a global store behind it the compiler must emit s_waitcnt vmcnt(0) before the s_barrier -- and vmcnt counts
// loads too, so every global load a wave has in flight for a LATER phase is waited for at the barrier (chunk_post: the rho
// fragments requested ahead of the accumulation; chunk_loglik: the stores of b draining into HBM while the operators are
// built from the LDS copy).  Where the phases on either side of a barrier exchange data through LDS only, this is the one to use.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Block-wide sum of one double per thread through LDS (blockDim.x multiple of 64, <= 1024).
__device__ __forceinline__ double block_sum(double v, double* lds /* >= 16 doubles */) {
    v = allreduce_sum<64>(v);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[wave] = v;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < nw; ++w) tot += lds[w];
    return tot;
}

// ---------------------------------------------------------------------------------------
// f16 operand pairs of the split GEMMs (vbx_split.hpp)
// ---------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
using f4 = Vec<float>::v4;

constexpr int kSplitTop = 14;          // scaled operands: largest magnitude in [2^13, 2^14)
constexpr int kSplitMaxDp = 1024;      // fin_kernel keeps a speaker's alpha row in LDS while it looks for the scale

// exponent e such that amax 2^e lies in [2^13, 2^14); 0 for amax = 0 (or not finite)
__device__ __forceinline__ int split_exponent(float amax) {
    if (!(amax > 0.0f) || !(amax < INFINITY)) return 0;
    return max(-100, min(100, kSplitTop - __builtin_amdgcn_frexp_expf(amax)));
}

__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// offsets in halfs inside one recording's alpha fragments [Sp / 16][Dp / 32][hi | lo | lo2][64 lanes][8]
__device__ __forceinline__ long long alpha_frag_offset(int s, int d, int hl, int Dp) {
    const int n = s >> 4, j = s & 15, kk = d >> 5, g = (d & 31) >> 3, e = d & 7;
    return ((((long long)n * (Dp >> 5) + kk) * 3 + hl) * 64 + 16 * g + j) * 8 + e;
}

}  // namespace vbx
