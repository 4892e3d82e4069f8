// vbx_fb_dense.hpp -- forward_backward (VBx.py:146-175) for an ARBITRARY S x S transition matrix.
//
// VBx() itself only ever builds tr = I*loopProb + (1-loopProb)*pi (VBx.py:98), which the other kernels exploit (one
// reduction per frame instead of a mat-vec); the reference's helper, however, is a general HMM forward-backward, and
// this kernel is its device counterpart: scaled linear domain, one workgroup per direction,
//     forward :  a_t = b_t * (A^T ahat_{t-1}),        s_t = sum a_t,    ahat_t = a_t / s_t     (VBx.py:167-168)
//     backward:  beta_t = A (b_{t+1} * bhat_{t+1}),   q_t = sum beta_t, bhat_t = beta_t / q_t  (VBx.py:170-171)
// with A = tr + 1e-8 (VBx.py:158) and b_t = exp(lls_t - max lls_t).  Both are "out[o] = sum_k v[k] M[k][o]" with
// M = A (forward) or A^T (backward); thread (o, h) keeps its NI = S / HL entries of M in registers for the whole walk,
// the partial sums of the HL column blocks meet in LDS.  The host turns (ahat, s, bhat, q, max) into lfw / lbw / tll /
// the posteriors in float64.  One dependent mat-vec per frame: ~1 us per frame, any S <= 256 -- a compatibility path,
// not a hot one (the reference needs 80 us per frame at S = 30).
#pragma once
#include "vbx_device.hpp"

namespace vbx {

template <int SP> struct FbDenseCfg {
    static constexpr int HL = SP < 1024 / SP ? SP : 1024 / SP;     // column blocks per output
    static constexpr int NI = SP / HL;                              // entries of M per thread
    static constexpr int kThreads = SP * HL;
};

// M0: A padded to [SP][SP] row-major (forward: M[k][o] = A[k][o]); M1: A^T.  bmat [T][SP], v0: ip + 1e-8 (forward).
// out [T][SP] (ahat | bhat), scale [T] (s | q; q[T-1] = 1).
template <typename R, int SP>
__global__ __launch_bounds__(FbDenseCfg<SP>::kThreads) void fb_dense_kernel(const R* __restrict__ M0, const R* __restrict__ M1,
                                                                            const R* __restrict__ bmat, const R* __restrict__ v0,
                                                                            R* __restrict__ ahat, R* __restrict__ bhat,
                                                                            R* __restrict__ fscale, R* __restrict__ bscale, int T, int S) {
    using Cfg = FbDenseCfg<SP>;
    constexpr int HL = Cfg::HL, NI = Cfg::NI, NW = (Cfg::kThreads + 63) / 64;
    __shared__ R vec[SP];
    __shared__ R part[Cfg::kThreads];
    __shared__ R wsum[NW];
    const int dir = blockIdx.x;
    const int tid = threadIdx.x, o = tid % SP, h = tid / SP;
    const R* __restrict__ M = dir == 0 ? M0 : M1;
    R m[NI];
#pragma unroll
    for (int ii = 0; ii < NI; ++ii) m[ii] = M[(long long)(h * NI + ii) * SP + o];
    R* __restrict__ out = dir == 0 ? ahat : bhat;
    R* __restrict__ scale = dir == 0 ? fscale : bscale;

    // sum over the first SP threads' values (others pass 0); every thread gets it
    auto block_total = [&](R v) {
        v = allreduce_sum<64>(v);
        if ((tid & 63) == 0) wsum[tid >> 6] = v;
        __syncthreads();
        R tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += wsum[w];
        return tot;
    };
    auto matvec = [&]() {                                   // vec -> sum_k vec[k] M[k][o], valid in threads < SP
        R acc = 0;
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) acc += vec[h * NI + ii] * m[ii];
        part[tid] = acc;
        __syncthreads();
        R tot = 0;
        if (tid < SP) {
#pragma unroll
            for (int q = 0; q < HL; ++q) tot += part[q * SP + o];
        }
        return tot;
    };

    if (dir == 0) {
        for (int t = 0; t < T; ++t) {
            const R b = (tid < SP) ? bmat[(long long)t * SP + o] : (R)0;
            R a;
            if (t == 0) {
                a = (tid < SP && o < S) ? b * v0[o] : (R)0;                       // VBx.py:163
                __syncthreads();
            } else {
                a = matvec() * b;
                if (!(tid < SP && o < S)) a = 0;
            }
            const R s = block_total(a);
            const R ah = a * fast_rcp(s);
            if (tid < SP) {
                vec[o] = ah;
                out[(long long)t * SP + o] = ah;
                if (o == 0) scale[t] = s;
            }
            __syncthreads();
        }
    } else {
        R bh = (tid < SP && o < S) ? (R)1 : (R)0;
        if (tid < SP) {
            out[(long long)(T - 1) * SP + o] = bh;
            if (o == 0) scale[T - 1] = 1;
        }
        for (int t = T - 2; t >= 0; --t) {
            const R b = (tid < SP) ? bmat[(long long)(t + 1) * SP + o] : (R)0;
            if (tid < SP) vec[o] = b * bh;
            __syncthreads();
            R beta = matvec();
            if (!(tid < SP && o < S)) beta = 0;
            const R q = block_total(beta);
            bh = beta * fast_rcp(q);
            if (tid < SP) {
                out[(long long)t * SP + o] = bh;
                if (o == 0) scale[t] = q;
            }
            __syncthreads();
        }
    }
}

}  // namespace vbx
