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
// not a hot one (the reference needs 80 us per frame at S = 30).  More states: fb_dense_big_kernel below.
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

// More than 256 states: the thread's column of M no longer fits in registers, so M stays in HBM / L2 (S^2 entries, read
// once per frame: coalesced along o) and a thread owns the outputs o = tid, tid + 1024, ...; the vector lives in LDS.
// Any S (Sp = S rounded up to 64) up to kFbDenseBigMax states; ~S^2 / 1024 loads per thread and frame -- tens of
// microseconds per frame at S = 1024: a compatibility path for the reference's general helper (VBx.py:146-175).
constexpr int kFbDenseBigMax = 4096;
template <typename R>
__global__ __launch_bounds__(1024) void fb_dense_big_kernel(const R* __restrict__ M0, const R* __restrict__ M1,
                                                            const R* __restrict__ bmat, const R* __restrict__ v0,
                                                            R* __restrict__ ahat, R* __restrict__ bhat,
                                                            R* __restrict__ fscale, R* __restrict__ bscale, int T, int S, int Sp) {
    constexpr int NO = kFbDenseBigMax / 1024;              // outputs per thread
    __shared__ R vec[kFbDenseBigMax];
    __shared__ R wsum[16];
    const int dir = blockIdx.x, tid = threadIdx.x;
    const R* __restrict__ M = dir == 0 ? M0 : M1;
    R* __restrict__ out = dir == 0 ? ahat : bhat;
    R* __restrict__ scale = dir == 0 ? fscale : bscale;
    auto block_total = [&](R v) {
        v = allreduce_sum<64>(v);
        __syncthreads();                                    // (wsum of the previous round has been read)
        if ((tid & 63) == 0) wsum[tid >> 6] = v;
        __syncthreads();
        R tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += wsum[w];
        return tot;
    };
    auto matvec = [&](R (&acc)[NO]) {                       // vec -> sum_k vec[k] M[k][o] for this thread's outputs
#pragma unroll
        for (int u = 0; u < NO; ++u) acc[u] = 0;
        for (int k0 = 0; k0 < S; k0 += 8) {
            R mv[8][NO];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int u = 0; u < NO; ++u) {
                    const int o = tid + 1024 * u, k = min(k0 + kk, S - 1);
                    mv[kk][u] = o < Sp ? M[(long long)k * Sp + o] : (R)0;
                }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int u = 0; u < NO; ++u) acc[u] += (k0 + kk < S ? vec[k0 + kk] : (R)0) * mv[kk][u];
        }
    };
    R cur[NO];
    if (dir == 0) {
        for (int t = 0; t < T; ++t) {
            R a[NO], part = 0;
            if (t > 0) matvec(a);
#pragma unroll
            for (int u = 0; u < NO; ++u) {
                const int o = tid + 1024 * u;
                const R b = o < S ? bmat[(long long)t * Sp + o] : (R)0;
                a[u] = o < S ? (t == 0 ? b * v0[o] : a[u] * b) : (R)0;        // VBx.py:163,167
                part += a[u];
            }
            const R s = block_total(part);                  // (its barriers: every thread is done reading vec)
            const R is = fast_rcp(s);
#pragma unroll
            for (int u = 0; u < NO; ++u) {
                const int o = tid + 1024 * u;
                if (o < Sp) {
                    vec[o] = a[u] * is;
                    out[(long long)t * Sp + o] = a[u] * is;
                }
            }
            if (tid == 0) scale[t] = s;
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int u = 0; u < NO; ++u) {
            const int o = tid + 1024 * u;
            cur[u] = o < S ? (R)1 : (R)0;
            if (o < Sp) out[(long long)(T - 1) * Sp + o] = cur[u];
        }
        if (tid == 0) scale[T - 1] = 1;
        for (int t = T - 2; t >= 0; --t) {
#pragma unroll
            for (int u = 0; u < NO; ++u) {
                const int o = tid + 1024 * u;
                if (o < Sp) vec[o] = o < S ? bmat[(long long)(t + 1) * Sp + o] * cur[u] : (R)0;
            }
            __syncthreads();
            R beta[NO], part = 0;
            matvec(beta);
#pragma unroll
            for (int u = 0; u < NO; ++u) {
                if (tid + 1024 * u >= S) beta[u] = 0;
                part += beta[u];
            }
            const R q = block_total(part);
            const R iq = fast_rcp(q);
#pragma unroll
            for (int u = 0; u < NO; ++u) {
                const int o = tid + 1024 * u;
                cur[u] = beta[u] * iq;
                if (o < Sp) out[(long long)t * Sp + o] = cur[u];
            }
            if (tid == 0) scale[t] = q;
            __syncthreads();
        }
    }
}

}  // namespace vbx
