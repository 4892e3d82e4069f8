// vbx_big.hpp -- more than 1024 speakers (HMM states) per recording: a compatibility path.
//
// The reference takes any number of states (VBx.py:76-85: `pi` of any length; vbhmm.py:151 sets S to the number of AHC
// clusters).  Up to 1024 the sequential walk keeps a recording's state vector in the registers of ONE wavefront
// (fb_seq_kernel: 16 states per lane) and the per-recording reductions give every speaker a thread (fin_kernel, post_kernel<SP>).
// Beyond that the same three steps run on a whole workgroup per recording and direction, with loops over blocks of states:
//
//   fb_big_kernel        forward / backward walk (VBx.py:146-175, in the linear domain as fb_seq_kernel): 1024 threads, a thread
//                        holds the states tid + 1024 r, one workgroup-wide sum per frame (two barriers): ~1.5 us per frame
//   post_big_kernel      posteriors and the "entered" statistic of a tile (VBx.py:101-103, 174): row sums first (a wavefront
//                        per frame), then a thread per state over the tile's frames
//   iter_fin_big_kernel  the iteration-finishing role of fin_kernel (ELBO, pi update, history, stop test: VBx.py:100-105,
//                        122-125) with loops over the speakers; the M-step role of fin_kernel itself is per (recording, speaker)
//                        and takes any count as it is, as do loglik_kernel + rownorm_kernel and mstep_acc_kernel.
//
// O(T S) work per iteration like every other path (the reference: O(T S^2)); it is about taking such a recording at all,
// not about speed.  Padded width Sp = the power of two >= S, up to kBigMaxStates.
#pragma once
#include "vbx_kernels.hpp"

namespace vbx {

constexpr int kBigMaxStates = 16384;               // NR = Sp / 1024 = 2, 4, 8 or 16 states per thread of the walk

// sum over the 1024 threads of a workgroup, the same value in every thread (two barriers; `red` = 16 doubles of LDS)
template <typename R> __device__ __forceinline__ R big_block_sum(R v, double* red) {
    v = allreduce_sum<64>(v);
    __syncthreads();                                   // (the previous round's readers are done with `red`)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = (double)v;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w];
    return (R)tot;
}

// grid = (n_rec, 2), block = 1024: blockIdx.y = 0 walks forward, 1 backward; NR = Sp / 1024
template <typename R, int NR>
__global__ __launch_bounds__(1024) void fb_big_kernel(BatchView<R> bt) {
    constexpr int kBigRegs = NR;
    __shared__ double red[16];
    const int rec = blockIdx.x;
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int Sp = bt.Sp, T = rd.T, tid = threadIdx.x;
    const R lp = (R)rd.lp;
    const R* __restrict__ B = bt.bmat + rd.row0 * Sp;
    R c[kBigRegs], v[kBigRegs], bcur[kBigRegs], bnext[kBigRegs];
#pragma unroll
    for (int r = 0; r < kBigRegs; ++r) {
        const int j = tid + 1024 * r;
        const bool live = r < NR && j < rd.S;
        c[r] = live ? (R)((1.0 - rd.lp) * bt.pi[(long long)rec * Sp + j] + 1e-8) : (R)0;
        v[r] = 0;
        bcur[r] = bnext[r] = 0;
    }
    auto load_row = [&](R (&dst)[kBigRegs], int t) {
#pragma unroll
        for (int r = 0; r < kBigRegs; ++r)
            dst[r] = (r < NR && t >= 0 && t < T) ? B[(long long)t * Sp + tid + 1024 * r] : (R)0;
    };
    if (blockIdx.y == 0) {
        // ------------------------------------------------------------------ forward   (VBx.py:163, 167-168)
        R* __restrict__ A = bt.ahat + rd.row0 * Sp;
        R p0[kBigRegs];
#pragma unroll
        for (int r = 0; r < kBigRegs; ++r) {
            const int j = tid + 1024 * r;
            p0[r] = (r < NR && j < rd.S) ? (R)(bt.ip[(long long)rec * Sp + j] + 1e-8) : (R)0;
        }
        ScaledProduct sp;
        load_row(bcur, 0);
        for (int t = 0; t < T; ++t) {
            load_row(bnext, t + 1);                    // in flight across the sum
            R a[kBigRegs], part = 0;
#pragma unroll
            for (int r = 0; r < kBigRegs; ++r) {
                a[r] = (t == 0) ? bcur[r] * p0[r] : bcur[r] * (lp * v[r] + c[r]);
                part += a[r];
            }
            const R s = big_block_sum(part, red);
            const R inv = (R)1 / s;
#pragma unroll
            for (int r = 0; r < kBigRegs; ++r) {
                v[r] = a[r] * inv;
                if (r < NR) A[(long long)t * Sp + tid + 1024 * r] = v[r];
                bcur[r] = bnext[r];
            }
            sp.mul((double)s);
            if ((t & 15) == 15) sp.renorm();
            if (bt.fw_scale && tid == 0) bt.fw_scale[rd.row0 + t] = s;
        }
        double msum = 0.0;
        for (int t = tid; t < T; t += 1024) msum += (double)bt.mrow[rd.row0 + t];
        msum = big_block_sum(msum, red);
        sp.renorm();
        if (tid == 0) bt.state[rec].tll = sp.log_value() + msum;
    } else {
        // ------------------------------------------------------------------ backward  (VBx.py:165, 170-171)
        R* __restrict__ Bh = bt.bhat + rd.row0 * Sp;
#pragma unroll
        for (int r = 0; r < kBigRegs; ++r) {
            v[r] = 1;
            if (r < NR) Bh[(long long)(T - 1) * Sp + tid + 1024 * r] = v[r];
        }
        load_row(bcur, T - 1);                         // row t + 1 serves frame t
        for (int t = T - 2; t >= 0; --t) {
            load_row(bnext, t);
            R e[kBigRegs], part = 0;
#pragma unroll
            for (int r = 0; r < kBigRegs; ++r) {
                e[r] = bcur[r] * v[r];
                part += c[r] * e[r];
            }
            const R q = big_block_sum(part, red);
            const R sc = lp / q;
            if (bt.bw_scale && tid == 0) bt.bw_scale[rd.row0 + t] = q;
#pragma unroll
            for (int r = 0; r < kBigRegs; ++r) {
                v[r] = sc * e[r] + (R)1;
                if (r < NR) Bh[(long long)t * Sp + tid + 1024 * r] = v[r];
                bcur[r] = bnext[r];
            }
        }
    }
}

// grid = ntiles_total, block = 256
template <typename R>
__global__ __launch_bounds__(256) void post_big_kernel(BatchView<R> bt) {
    __shared__ R inv_l[kTileFrames];
    const int tile = blockIdx.x;
    const int rec = bt.tile_rec[tile];
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int Sp = bt.Sp, t0 = bt.tile_t0[tile], tend = min(t0 + kTileFrames, rd.T);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const R lp = (R)rd.lp;
    const R* __restrict__ A = bt.ahat + rd.row0 * Sp;
    const R* __restrict__ Bh = bt.bhat + rd.row0 * Sp;
    R* __restrict__ G = bt.gamma + rd.row0 * Sp;
    // 1 / sum_j ahat bhat of every frame of the tile: a wavefront per frame
    for (int f = t0 + wave; f < tend; f += 4) {
        R sum = 0;
        for (int j = lane; j < Sp; j += 64) sum += A[(long long)f * Sp + j] * Bh[(long long)f * Sp + j];
        sum = allreduce_sum<64>(sum);
        if (lane == 0) inv_l[f - t0] = (R)1 / sum;
    }
    __syncthreads();
    // a thread per state over the frames of the tile (adjacent threads = adjacent states: coalesced rows)
    for (int j = threadIdx.x; j < Sp; j += 256) {
        const R cj = (j < rd.S) ? (R)((1.0 - rd.lp) * bt.pi[(long long)rec * Sp + j] + 1e-8) : (R)1;
        double ent = 0.0;
        R aprev = t0 >= 1 ? A[(long long)(t0 - 1) * Sp + j] : (R)0;
        for (int f = t0; f < tend; ++f) {
            const R a = A[(long long)f * Sp + j];
            const R gm = a * Bh[(long long)f * Sp + j] * inv_l[f - t0];
            G[(long long)f * Sp + j] = gm;
            if (f >= 1 && j < rd.S) ent += (double)(gm / (lp * aprev + cj));
            aprev = a;
        }
        bt.epart[(long long)tile * Sp + j] = ent;
    }
}

// grid = n_rec, block = 1024: what fin_kernel does under blockIdx.y == Sp (mode & 2), for Sp > 1024.  Launched on its own,
// BEFORE fin_kernel in mode 1 (the host swaps the state buffers in between, as after any launch with a finishing role).
template <typename R, int NR>
__global__ __launch_bounds__(1024) void iter_fin_big_kernel(BatchView<R> bt) {
    constexpr int kBigRegs = NR;
    __shared__ double lds[16];
    __shared__ int done_sh;
    const int rec = blockIdx.x, Sp = bt.Sp, tid = threadIdx.x;
    RecState st = bt.state[rec];
    const RecDesc rd = bt.recs[rec];
    const int u0 = rd.tile0, nu = rd.ntiles;
    if (st.done) {                                         // frozen: the state just moves to the other buffer
        if (tid == 0) bt.state_out[rec] = st;
        return;
    }
    double pn[kBigRegs], pj[kBigRegs], part_pn = 0.0, part_em = 0.0;
#pragma unroll
    for (int r = 0; r < kBigRegs; ++r) {
        pn[r] = pj[r] = 0.0;
        const int j = tid + 1024 * r;
        if (r < NR && j < rd.S) {
            double ent = 0.0;
            for (int tl = 0; tl < nu; ++tl) ent += bt.epart[(long long)(u0 + tl) * Sp + j];
            pj[r] = bt.pi[(long long)rec * Sp + j];
            const double g0 = bt.gamma0 ? (double)bt.gamma0[(long long)rec * Sp + j] : (double)bt.gamma[rd.row0 * Sp + j];
            pn[r] = g0 + (1.0 - rd.lp) * pj[r] * ent;                                    // VBx.py:101-103
            part_pn += pn[r];
            part_em += bt.emodel[(long long)(st.n_iters & 1) * bt.vec_stride + (long long)rec * Sp + j];
        }
    }
    const double tot = block_sum(part_pn, lds);
    const double emt = block_sum(part_em, lds);
#pragma unroll
    for (int r = 0; r < kBigRegs; ++r) {
        const int j = tid + 1024 * r;
        if (r < NR) {
            bt.pi_prev[(long long)rec * Sp + j] = pj[r];
            bt.pi[(long long)rec * Sp + j] = pn[r] / tot;                                // VBx.py:104
        }
    }
    if (tid == 0) {
        const double tll = st.tll;                         // (the sequential walk has left it in the state)
        const double elbo = tll + rd.Fa * rd.gsum + 0.5 * rd.Fb * emt;                   // VBx.py:100
        const int it = st.n_iters;
        if (it < bt.max_iters) bt.Li[(long long)rec * bt.max_iters + it] = elbo;         // VBx.py:105
        if (it > 0 && elbo - st.elbo_prev < bt.epsilon) {                                // VBx.py:122-125
            st.done = 1;
            if (elbo - st.elbo_prev < 0) st.warned = 1;
        }
        st.elbo_prev = elbo;
        st.n_iters = it + 1;
        bt.state_out[rec] = st;
        done_sh = st.done;
    }
    __syncthreads();
    if (done_sh)
        for (int tl = tid; tl < rd.ntiles; tl += 1024) bt.tile_done[rd.tile0 + tl] = 1;
}

}  // namespace vbx
