// vbx_chunk_loglik.hpp -- chunk_loglik: per-frame speaker log-likelihoods and the chunk's transfer operator in one
// pass over rho (the other per-chunk kernel, chunk_post, is in vbx_chunk_post.hpp).
#pragma once
#include <type_traits>
#include "vbx_scan.hpp"

namespace vbx {

// =======================================================================================
// chunk_loglik_kernel (one workgroup = one chunk of kTileFrames frames):
//   phase 1  per-frame speaker log-likelihoods on MFMA 16x16x4 (loglik_kernel):          VBx.py:97
//            l = Fa (rho alpha^T + bias),  m_t = max_s l,  b = exp(l - m_t)  -> bmat, mrow (HBM) and LDS
//   phase 2  the chunk's forward transfer operator (scan1_kernel), straight from LDS:   VBx.py:167-171
//            x <- b_t (lp x + c sum(x))   for every operator column, t = t0 .. t0+len-1
// Phase 2 differs from scan1_kernel in instruction count only: columns are rescaled (exact powers of two)
// every four frames instead of every frame -- a frame shrinks a column sum by at least min c = 1e-8, so
// four frames stay inside the f32 range.
// =======================================================================================
constexpr int kAlphaSlice = 128;
template <typename R, int SP> struct ChunkLoglikCfg {
    static constexpr int kBytes = (kTileFrames > kAlphaSlice + 4 ? kTileFrames : kAlphaSlice + 4) * SP * (int)sizeof(R) + 1024;
    static constexpr bool kFits = kBytes <= 160 * 1024;
};

template <typename R, int SP>
__global__ __launch_bounds__(256, (SP * (int)sizeof(R) <= 128 ? 8 : SP * (int)sizeof(R) <= 256 ? 4 : 2)) void chunk_loglik_kernel(BatchView<R> bt) {
    using M = Mfma16<R>;
    using acc_t = typename M::acc_t;
    using R4 = typename Vec<R>::v4;
    constexpr int NT = SP / 16;
    // operator build: PH lanes share a column and hold NR states each.  Fewer lanes per column = fewer issue
    // slots per frame (the column sum needs log2(PH) DPP stages with their wait states): 72 / 46 / 39 slots per
    // chunk-frame for PH = 8 / 4 / 2 at SP = 32, but also fewer wavefronts to hide each other's latency.
    // Measured on 64 recordings of T = 10 000: 171 / 163 / 176 us per launch, so PH = 4 (two wavefronts build
    // the operator, the other two retire after phase 1).  With the packed two-operation frame of phase 2,
    // PH = 8 and PH = 4 measure the same (346 vs 344-349 us per iteration).
    constexpr int kOperatorLanes = 4;
    constexpr int kLanesWanted = SP / 4 < kOperatorLanes ? SP / 4 : kOperatorLanes;     // a lane keeps >= 4 states
    constexpr int PH = (SP * kLanesWanted <= 256) ? kLanesWanted : 256 / SP, NR = SP / PH;
    constexpr int AST = kAlphaSlice + 4;               // padded row of the alpha slice: conflict-free fragment reads
    // one LDS region, two lives: the alpha slice during the MFMA pass, then b of the chunk
    constexpr int kLds = kTileFrames * SP > SP * AST ? kTileFrames * SP : SP * AST;
    __shared__ __attribute__((aligned(16))) R lds[kLds];
    R* const al = lds;                                 // alpha[:, k0 : k0 + kAlphaSlice], rows padded to AST
    R* const btile = lds;                              // b[kTileFrames][SP]
    const int tile = blockIdx.x;
    const int rec = bt.tile_rec[tile];
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int Dp = bt.Dp;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int t0 = bt.tile_t0[tile];
    const int len = min(kTileFrames, rd.T - t0);
    const R lp = (R)rd.lp;

    VBX_CLOCKS_DECL();
    VBX_STAMP();
    // ---- phase 1: wave w owns frames [32w, 32w+32) of the chunk = 2 M-tiles ----------------------
    {
        const int f0 = t0 + 32 * wave;
        const R* __restrict__ rho = bt.rho + rd.row0 * Dp;
        const R* __restrict__ alpha = bt.alpha + (long long)rec * SP * Dp;
        acc_t acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = acc_t{0, 0, 0, 0};
        // rows past the end of the recording are clamped (their results are never stored)
        const int rowA0 = min(f0 + i, rd.T - 1), rowA1 = min(f0 + 16 + i, rd.T - 1);
        constexpr int QB = 2;                          // K blocks of 16 whose rho fragments are loaded together
        // The speaker means (B operand) are staged in LDS once per workgroup, kAlphaSlice feature dims at a
        // time: every wave needs all of alpha, and fetching it per wave from L2 cost as much as streaming rho
        // (a CU sustains ~10 B/clk of global loads whether they hit L2 or HBM).
#pragma unroll 1
        for (int k0 = 0; k0 < Dp; k0 += kAlphaSlice) {
            const int kw = min(kAlphaSlice, Dp - k0);          // multiple of 32
            if (k0 > 0) __syncthreads();
            {
                const int vpr = kw / 4;                        // 16-byte vectors per speaker row
                for (int idx = tid; idx < SP * vpr; idx += 256) {
                    const int row = idx / vpr, c4 = idx - row * vpr;
                    *reinterpret_cast<R4*>(al + row * AST + 4 * c4) =
                        *reinterpret_cast<const R4*>(alpha + (long long)row * Dp + k0 + 4 * c4);
                }
            }
            const int nq = kw / 16;
            R4 a0[QB], a1[QB];
            auto load_a = [&](int q0) {
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    const int kk = k0 + 16 * min(q0 + u, nq - 1) + 4 * g;
                    a0[u] = *reinterpret_cast<const R4*>(rho + (long long)rowA0 * Dp + kk);
                    a1[u] = *reinterpret_cast<const R4*>(rho + (long long)rowA1 * Dp + kk);
                }
            };
            load_a(0);
            __syncthreads();
#pragma unroll 1
            for (int q0 = 0; q0 < nq; q0 += QB) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    R4 bfr[QB];
#pragma unroll
                    for (int u = 0; u < QB; ++u)
                        bfr[u] = *reinterpret_cast<const R4*>(al + (16 * n + i) * AST + 16 * min(q0 + u, nq - 1) + 4 * g);
#pragma unroll
                    for (int u = 0; u < QB; ++u) {
                        if (q0 + u < nq) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                acc[0][n] = M::mma(a0[u][r], bfr[u][r], acc[0][n]);
                                acc[1][n] = M::mma(a1[u][r], bfr[u][r], acc[1][n]);
                            }
                        }
                    }
                }
                if (q0 + QB < nq) load_a(q0 + QB);
            }
        }
        VBX_STAMP();
        __syncthreads();                               // every wave is done with the alpha slice: b may overwrite it
        const R Fa = (R)rd.Fa;
        R biasv[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) biasv[n] = bt.bias[(long long)rec * SP + 16 * n + i];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int fl = 32 * wave + 16 * m + M::row(lane, r);     // frame within the chunk
                R v[NT];
                R mx = neg_inf<R>();
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int s = 16 * n + i;
                    v[n] = (s < rd.S) ? Fa * (acc[m][n][r] + biasv[n]) : neg_inf<R>();
                    mx = vmax(mx, v[n]);
                }
                mx = allreduce_max<16>(mx);
                const bool ok = fl < len;
                const long long cell = (rd.row0 + t0 + fl) * SP;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const R b = exp_r(v[n] - mx);
                    btile[fl * SP + 16 * n + i] = b;
                    if (ok) bt.bmat[cell + 16 * n + i] = b;
                }
                if (ok && i == 0) bt.mrow[rd.row0 + t0 + fl] = mx;
            }
        }
    }
    VBX_STAMP();
    __syncthreads();
    VBX_STAMP();

    // ---- phase 2: transfer operators -----------------------------------------------------------------
    // bt.spt == 2: one operator per half tile (frames [0, 64) and [64, len)), built side by side by two
    // groups of NOPT threads when the workgroup is wide enough, else one after the other.
    {
        constexpr int NOPT = SP * PH;                      // threads that build one operator
        constexpr int PAR = 256 / NOPT >= 2 ? 2 : 1;       // operators built side by side
        const int nhalf = bt.spt == 2 ? (len > kScanHalf ? 2 : 1) : 1;
        const int grp = tid / NOPT, lt = tid % NOPT;
        for (int h0 = 0; h0 < nhalf; h0 += PAR) {
            const int half = h0 + grp;
            if (grp < PAR && half < nhalf) {
                const int lo = bt.spt == 2 ? half * kScanHalf : 0;
                const int hi = bt.spt == 2 ? min(len, lo + kScanHalf) : len;
                const int col = lt / PH, part = lt % PH, j0 = part * NR;
                // With lp > 0 the recursion runs on z_f = x_f / lp^(transitions so far):
                //     x <- b (lp x + c sum(x))     becomes     z <- b (z + (c / lp) sum(z)),
                // one FMA and one product per state instead of three operations; lp^(transitions) goes into the
                // column's mantissa and exponent at the end.  lp == 0 (or subnormally small) keeps the plain form.
                const bool scaled = rd.lp >= 0x1p-20;
                R x[NR], c[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    x[r] = (j0 + r == col) ? (R)1 : (R)0;
                    const double cj = (1.0 - rd.lp) * bt.pi[(long long)rec * SP + j0 + r] + 1e-8;
                    c[r] = (j0 + r < rd.S) ? (R)(scaled ? cj / rd.lp : cj) : (R)0;
                }
                int expo = 0, step = lo;
                if (t0 + lo == 0) {          // frame 0 of the recording: x <- b_0 * x (VBx.py:163, no transition)
#pragma unroll
                    for (int r = 0; r < NR; ++r) x[r] *= btile[j0 + r];
                    step = 1;
                }
                const int transitions = hi - step;
                auto colsum = [&]() {        // pairwise: packed adds
                    R v[NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) v[r] = x[r];
#pragma unroll
                    for (int w = NR / 2; w >= 1; w >>= 1)
#pragma unroll
                        for (int r = 0; r < w; ++r) v[r] += v[r + w];
                    return column_sum<PH>(v[0]);
                };
                auto recursion = [&](auto scaled_tag) {
                    // written on pairs of states: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do two per issue slot
                    using R2 = typename Vec<R>::v2;
                    constexpr int NP = NR / 2;
                    static_assert(NR % 4 == 0, "operator lanes hold a multiple of four states");
                    R2 x2[NP], c2[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        x2[p] = R2{x[2 * p], x[2 * p + 1]};
                        c2[p] = R2{c[2 * p], c[2 * p + 1]};
                    }
                    const R2 lp2 = R2{lp, lp};
                    auto colsum2 = [&]() {
                        R2 v[NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p) v[p] = x2[p];
#pragma unroll
                        for (int w = NP / 2; w >= 1; w >>= 1)
#pragma unroll
                            for (int p = 0; p < w; ++p) v[p] += v[p + w];
                        return column_sum<PH>(v[0].x + v[0].y);
                    };
                    auto frame = [&](int f, R sig) {
                        const R2 sig2 = R2{sig, sig};
                        const R* row = btile + f * SP + j0;
#pragma unroll
                        for (int q = 0; q < NR / 4; ++q) {
                            const R4 b4 = *reinterpret_cast<const R4*>(row + 4 * q);
                            const R2 b0 = R2{b4.x, b4.y}, b1 = R2{b4.z, b4.w};
                            if (decltype(scaled_tag)::value) {
                                x2[2 * q] = b0 * (c2[2 * q] * sig2 + x2[2 * q]);
                                x2[2 * q + 1] = b1 * (c2[2 * q + 1] * sig2 + x2[2 * q + 1]);
                            } else {
                                x2[2 * q] = b0 * (lp2 * x2[2 * q] + c2[2 * q] * sig2);
                                x2[2 * q + 1] = b1 * (lp2 * x2[2 * q + 1] + c2[2 * q + 1] * sig2);
                            }
                        }
                    };
                    auto renorm = [&]() {            // column sum back to [0.5, 1): one exact product per pair
                        R sig = colsum2();
                        const int e = rescale_exponent(sig);
                        expo += e;
                        const R sc = scale2((R)1, -e);
                        const R2 sc2 = R2{sc, sc};
#pragma unroll
                        for (int p = 0; p < NP; ++p) x2[p] *= sc2;
                        return sig * sc;
                    };
                    for (; step + 4 <= hi; step += 4) {
                        frame(step, renorm());
#pragma unroll
                        for (int k = 1; k < 4; ++k) frame(step + k, colsum2());
                    }
                    for (; step < hi; ++step) frame(step, renorm());
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        x[2 * p] = x2[p].x;
                        x[2 * p + 1] = x2[p].y;
                    }
                };
                if (scaled) {
                    recursion(std::true_type{});
                    const double l2 = (double)transitions * log2(rd.lp), fl = floor(l2);
                    const R mant = (R)exp2(l2 - fl);             // lp^transitions = mant * 2^fl, mant in [1, 2)
                    expo += (int)fl;
#pragma unroll
                    for (int r = 0; r < NR; ++r) x[r] *= mant;
                } else {
                    recursion(std::false_type{});
                }
                {   // final power-of-two normalisation: column sums end in [0.5, 1)
                    const R sig = colsum();
                    const int e = rescale_exponent(sig);
                    expo += e;
#pragma unroll
                    for (int r = 0; r < NR; ++r) x[r] = scale2(x[r], -e);
                    // an all-zero column (b = 0 for its state at frame 0, or a padded state) must never win the
                    // exponent maximum in scan2
                    if (!(sig > (R)0)) expo = -(1 << 24);
                }
                const long long chunk = (long long)tile * bt.spt + half;
                R* __restrict__ dst = bt.op + (chunk * SP + col) * SP + j0;
#pragma unroll
                for (int r = 0; r < NR; ++r) dst[r] = x[r];
                if (part == 0) bt.opexp[chunk * SP + col] = expo;
            }
        }
    }
    VBX_STAMP();
#ifdef VBX_PHASE_CLOCKS
    if ((blockIdx.x % 1000) == 1 && (lane == 0) && bt.state[rec].n_iters == 3)
        printf("chunk_loglik wave %d: mfma %lld  epilogue %lld  wait %lld  operator %lld cycles\n", wave,
               clk[1] - clk[0], clk[2] - clk[1], clk[3] - clk[2], clk[4] - clk[3]);
#endif
}

}  // namespace vbx
