// vbx_chunk_loglik.hpp -- chunk_loglik: per-frame speaker log-likelihoods and the chunk's transfer operator in one
// pass over rho (the other per-chunk kernel, chunk_post, is in vbx_chunk_post.hpp).
#pragma once
#include "vbx_operator.hpp"
#include "vbx_split.hpp"

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

#ifndef VBX_SPLIT_LOGLIK_WAVES
#define VBX_SPLIT_LOGLIK_WAVES 6
#endif
// SPLIT (fp32 only): the product on v_mfma_f32_16x16x32_f16 with f16 operand pairs (vbx_split.hpp) -- rho from its
// fragment-ordered copy rho_a, alpha from the fragments fin_kernel wrote.
// LAT (round 6): the instance for batches that do not fill the chip.  There a launch's duration is one workgroup's life plus
// what its bytes take at (bytes in flight) / (memory latency) -- and a workgroup that fetches its rho slab one K-block ahead has
// a quarter of it in flight: 632 workgroups x 16 KB / 2 us = 5 TB/s whatever the memory system could deliver.  LAT requests the
// whole slab before the first product (64 registers per lane: fine at three or four workgroups per CU) -- for Dp <= 128.
template <typename R, int SP, bool SPLIT = false, bool LAT = false>
__global__ __launch_bounds__(256, (LAT ? (SP * (int)sizeof(R) <= 128 ? 4 : 2)
                                       : SP * (int)sizeof(R) <= 128 ? (SPLIT && SP == 32 ? VBX_SPLIT_LOGLIK_WAVES : 8) : SP * (int)sizeof(R) <= 256 ? 4 : 2)) void chunk_loglik_kernel(BatchView<R> bt) {
    static_assert(!SPLIT || sizeof(R) == 4, "the split GEMM is a mode of the fp32 path");
    using M = Mfma16<R>;
    using acc_t = typename M::acc_t;
    using R4 = typename Vec<R>::v4;
    constexpr int NT = SP / 16;
    // operator build: PH lanes share a column and hold NR states each; a thread can build NC columns side by side
    // (independent recursions on the same rows of b: vbx_operator.hpp).  Fewer lanes per column = fewer issue slots per
    // frame (the column sum needs log2(PH) DPP stages with their wait states) but fewer wavefronts to hide each other's
    // latency.  One column per thread: PH = 8 / 4 / 2 at SP = 32 measured 171 / 163 / 176 us per launch of 64 recordings
    // (round 1), so PH = 4.  Round 3 tried NC = 2 columns per thread on PH = 8 lanes (same thread count, half the LDS reads,
    // two independent chains per wavefront) against the phase's 800 cycles per frame and wave at SP = 64: SLOWER everywhere --
    // chunk_loglik 137 -> 147 us (SP = 32, f32), 250 -> 274 (f64), 823 -> 854 (SP = 64, the C5 sweep): the third DPP stage
    // and the second set of sums cost more issue slots than the second chain hides (A/B builds, -DVBX_OP_COLS=2).
#ifndef VBX_OP_COLS
#define VBX_OP_COLS 1
#endif
    constexpr int NC = (SP >= 32) ? VBX_OP_COLS : 1;
    constexpr int kOperatorLanes = 4 * NC;
    constexpr int kLanesWanted = SP / 4 < kOperatorLanes ? SP / 4 : kOperatorLanes;     // a lane keeps >= 4 states
    constexpr int PH = (SP * kLanesWanted / NC <= 256) ? kLanesWanted : 256 * NC / SP, NR = SP / PH;
    constexpr int AST = kAlphaSlice + 4;               // padded row of the alpha slice: conflict-free fragment reads
    // one LDS region, two lives: the alpha slice during the MFMA pass, then b of the chunk
    constexpr int kLds = kTileFrames * SP > SP * AST ? kTileFrames * SP : SP * AST;
    __shared__ __attribute__((aligned(16))) R lds[kLds];
    R* const al = lds;                                 // alpha[:, k0 : k0 + kAlphaSlice], rows padded to AST
    R* const btile = lds;                              // b[kTileFrames][SP]
    const int tile = tile_of_block(bt, blockIdx.x);
    if (tile < 0) return;
    const int rec = bt.tile_rec[tile];
    const RecState st = bt.state[rec];
    if (st.done) return;
    const int par = st.n_iters & 1;                    // the model of the iteration in flight (fin_kernel)
    const RecDesc rd = bt.recs[rec];
    const int Dp = bt.Dp;
    // (the wave index as a scalar: everything derived from it -- frame ranges, loop bounds, roles -- stays in SGPRs)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int t0 = bt.tile_t0[tile];
    const int len = min(kTileFrames, rd.T - t0);

    VBX_CLOCKS_DECL();
    VBX_STAMP();
    // ---- phase 1: wave w owns frames [32w, 32w+32) of the chunk = 2 M-tiles ----------------------
    {
        const int f0 = t0 + 32 * wave;
        const R* __restrict__ rho = bt.rho + rd.rho_row0 * Dp;
        const R* __restrict__ alpha = bt.alpha + (long long)par * bt.model_stride + (long long)rec * SP * Dp;
        acc_t acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = acc_t{0, 0, 0, 0};
        R cscale[NT];                                  // what turns an accumulator into rho . alpha (1, or 2^-(e_rho + e_alpha))
#pragma unroll
        for (int n = 0; n < NT; ++n) cscale[n] = (R)1;
        if constexpr (SPLIT) {
            // ---- f16 pairs: per K-block of 32 dims a wave loads its four A fragments (2 M-tiles x {hi, lo}: one
            // contiguous KB per load instruction) and multiplies them with the NT x {hi, lo} B fragments of the block,
            // staged in LDS in fragment order (lane-linear 16-byte reads); three MFMAs per product (vbx_split.hpp)
            const int KK = Dp >> 5;
            const int rtile = tile - rd.tile0 + rd.rho_tile0;
            const h8* __restrict__ ra = reinterpret_cast<const h8*>(bt.rho_a) + (long long)rtile * (kTileFrames * Dp / 4)
                                        + (long long)(2 * wave) * KK * 128 + lane;
            const h8* __restrict__ af = reinterpret_cast<const h8*>(bt.alpha_frag)
                                        + ((long long)par * bt.n_rec + rec) * (SP * Dp * 3 / 8);
            h8* const afl = reinterpret_cast<h8*>(lds);    // [n][kk of the slice: 2][hi | lo | lo2][lane]
            const int e_rho = bt.rho_e[rd.rho_rec];
#pragma unroll
            for (int n = 0; n < NT; ++n)
                cscale[n] = scale2((R)1, -(e_rho + bt.alpha_e[(long long)par * bt.vec_stride + (long long)rec * SP + 16 * n + i]));
            constexpr int NBUF = LAT ? 4 : 2;             // LAT: the whole slab (KK <= 4: the host launches LAT only then)
            h8 a[NBUF][2][2];                              // [buffer][M-tile][hi | lo]
            auto load_a = [&](int buf, int kk) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    a[buf][m][0] = ra[((long long)m * KK + kk) * 128];
                    a[buf][m][1] = ra[((long long)m * KK + kk) * 128 + 64];
                }
            };
            load_a(0, 0);
            if constexpr (LAT) {
#pragma unroll
                for (int kk = 1; kk < 4; ++kk)
                    if (kk < KK) load_a(kk, kk);
            }
            // (three terms of alpha per K-block: two K-blocks per staging round keep the slice inside the region b will take)
            constexpr int KS = 2;
            // one staging round: the alpha terms of K-blocks [kk0, kk0 + KS) into LDS, then their products.  `kslot` = the A
            // buffer of K-block kk0 (a compile-time constant in the LAT instance, whose rounds are unrolled: a register array
            // indexed at run time would live in scratch memory)
            auto round = [&](int kk0, auto slot0) {
                constexpr int kSlot0 = decltype(slot0)::value;
                const int nk = min(KS, KK - kk0);
                if (kk0 > 0) __syncthreads();
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    for (int q = tid; q < nk * 192; q += 256) afl[n * (KS * 192) + q] = af[((long long)n * KK + kk0) * 192 + q];
                __syncthreads();
#pragma unroll
                for (int kl = 0; kl < KS; ++kl) {
                    if (kl < nk) {
                        const int kk = kk0 + kl;
                        if (!LAT && kk + 1 < KK) load_a((kl + 1) & 1, kk + 1);      // (kk0 is even: buffer = kl & 1)
                        constexpr int kBufBase = LAT ? kSlot0 : 0;
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const h8* bf = afl + (n * KS + kl) * 192 + lane;
                            const h8 bh = bf[0], bl = bf[64], bl2 = bf[128];
#pragma unroll
                            for (int m = 0; m < 2; ++m)
                                acc[m][n] = mfma_split3(a[kBufBase + kl][m][0], a[kBufBase + kl][m][1], bh, bl, bl2, acc[m][n]);
                        }
                    }
                }
            };
            if constexpr (LAT) {
                round(0, std::integral_constant<int, 0>{});
                if (KK > KS) round(KS, std::integral_constant<int, KS>{});
            } else {
#pragma unroll 1
                for (int kk0 = 0; kk0 < KK; kk0 += KS) round(kk0, std::integral_constant<int, 0>{});
            }
        } else {
        // rows past the end of the recording are clamped (their results are never stored)
        const int rowA0 = min(f0 + i, rd.T - 1), rowA1 = min(f0 + 16 + i, rd.T - 1);
        constexpr int QB = LAT ? 4 : 2;   // K blocks of 16 whose rho fragments are loaded together (LAT: half a slice)
        // The speaker means (B operand) are staged in LDS once per workgroup, kAlphaSlice feature dims at a
        // time: every wave needs all of alpha, and fetching it per wave from L2 cost as much as streaming rho
        // (a CU sustains ~10 B/clk of global loads whether they hit L2 or HBM).
#pragma unroll 1
        for (int k0 = 0; k0 < Dp; k0 += kAlphaSlice) {
            const int kw = min(kAlphaSlice, Dp - k0);          // multiple of 32
            if (k0 > 0) __syncthreads();
            {
                // eight threads per speaker row, 32 rows per pass (kw is a multiple of 32: no division by a run-time
                // vector count in the index arithmetic)
                const int c8 = tid & 7;
#pragma unroll
                for (int row = tid >> 3; row < SP; row += 32) {
                    for (int c4 = c8; 4 * c4 < kw; c4 += 8)
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
        }
        VBX_STAMP();
        __syncthreads();                               // every wave is done with the alpha slice: b may overwrite it
        const R Fa = (R)rd.Fa;
        R biasv[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) biasv[n] = bt.bias[(long long)par * bt.vec_stride + (long long)rec * SP + 16 * n + i];
        R falo[NT];                                    // Fa * (what the bias lost when it was rounded to R): BatchView::bias_lo
#pragma unroll
        for (int n = 0; n < NT; ++n) falo[n] = Fa * bt.bias_lo[(long long)par * bt.vec_stride + (long long)rec * SP + 16 * n + i];
        // (a full chunk -- all but the last one of a recording -- stores without per-row conditions: each of them is a
        //  branch around the store)
        auto epilogue = [&](auto full_tag) {
            constexpr bool kFull = decltype(full_tag)::value;
            R* __restrict__ bout = bt.bmat + (rd.row0 + t0) * SP + i;
            R* __restrict__ mout = bt.mrow + rd.row0 + t0;
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
                        v[n] = (s < rd.S) ? Fa * (SPLIT ? acc[m][n][r] * cscale[n] + biasv[n] : acc[m][n][r] + biasv[n]) + falo[n] : neg_inf<R>();
                        mx = vmax(mx, v[n]);
                    }
                    mx = allreduce_max<16>(mx);
                    const bool ok = kFull || fl < len;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const R b = exp_r(v[n] - mx);
                        btile[fl * SP + 16 * n + i] = b;
                        if (ok) bout[fl * SP + 16 * n] = b;
                    }
                    if (ok && i == 0) mout[fl] = mx;
                }
            }
        };
        if (len == kTileFrames) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    }
    VBX_STAMP();
    lds_barrier();                 // (phase 2 reads the LDS copy of b; the stores of b and of the row maxima drain meanwhile)
    VBX_STAMP();

    // ---- phase 2: transfer operators, one column per group of PH lanes (vbx_operator.hpp) ------------------------------
    // The tile is cut at frame kTileFrames / 2: P1 (frames before the cut) and P2 (frames from the cut on) are built
    // side by side by two sets of waves where 2 SP PH <= 256 (one after the other at SP = 64), so the dependent chain
    // of a column is 64 frames long, and handed to chunk_post (`oph`), which re-runs the two halves of the tile at
    // the same time: it gets the vectors at the cut from one product with P1 / P2^T each.  The boundary walk
    // (scan2) keeps working on whole tiles: P = P2 P1 is composed here on v_mfma 16x16x4, exactly as
    // scan_compose_kernel multiplies chunk operators (weights 2^E shifted by the largest exponent on a column's
    // support, the column's scale kept as an integer exponent).
    {
        constexpr int NOPT = SP * PH / NC;                 // threads that build one operator
        constexpr int CS = SP / NC;                        // a thread's columns: colbase, colbase + CS, ...
        constexpr int H = kTileFrames / 2;
        constexpr bool kSideBySide = 2 * NOPT <= 256;
        constexpr int kNoMass = -(1 << 24), kNever = -(1 << 28);
        __shared__ int eF[SP], eW[SP];
        const bool split = bt.oph != nullptr;              // (uniform) off: one operator over the whole tile, as scan1 builds it
        const bool two = split && len > H;                 // (uniform) the tile has a second half
        const int my_half = !kSideBySide ? 0 : (NOPT % 64 == 0) ? wave / (NOPT / 64) : tid / NOPT;   // (scalar: loop bounds)
        const int otid = kSideBySide ? tid % NOPT : tid;
        const int colbase = otid / PH, part = otid % PH, j0 = part * NR;
        const bool builder = kSideBySide ? tid < 2 * NOPT : tid < NOPT;
        R x[2][NC][NR];                                    // [0]: my operator (side by side) or P1; [1]: P2 (SP = 64)
        int expo[2][NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) expo[0][k] = expo[1][k] = kNoMass;
        const R* c_rec = bt.cop + (long long)rec * SP;
        const LpPow* lppow = bt.lppow + (long long)rec * (kTileFrames + 1);
        auto build = [&](int half, R (&xo)[NC][NR], int (&eo)[NC]) {
            const int lo = half * H, hi = (half == 0 && split) ? min(len, H) : len;
            operator_columns<R, SP, PH, NC>(btile, lo, hi, t0 + lo == 0, colbase, part, rd.lp, c_rec, lppow, xo, eo);
            if (bt.oph) {
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int col = colbase + k * CS;
                    R* __restrict__ dst = bt.oph + (((long long)tile * 2 + half) * SP + col) * SP + j0;
#pragma unroll
                    for (int r = 0; r < NR; ++r) dst[r] = xo[k][r];
                    if (part == 0) bt.ophexp[((long long)tile * 2 + half) * SP + col] = eo[k];
                }
            }
        };
        if (builder) {
            if (kSideBySide) {
                if (my_half == 0 || two) build(my_half, x[0], expo[0]);
            } else {
                build(0, x[0], expo[0]);
                if (two) build(1, x[1], expo[1]);
            }
        }
        VBX_STAMP();                                       // (the operators are built and stored)
        if (!two) {                                        // P = P1
            if (builder && my_half == 0) {
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int col = colbase + k * CS;
                    R* __restrict__ dst = bt.op + ((long long)tile * SP + col) * SP + j0;
#pragma unroll
                    for (int r = 0; r < NR; ++r) dst[r] = x[0][k][r];
                    if (part == 0) bt.opexp[(long long)tile * SP + col] = expo[0][k];
                }
            }
        } else {
            // P = P2 P1.  F = P2 as built (column j contiguous) and W[j][i] = P1[j, i] 2^(E2_j - top_i) go through the
            // LDS region b has left; wave w < SP/16 multiplies the columns [16w, 16w + 16).
            using M = Mfma16<R>;
            R* const Fl = lds;
            R* const Wt = lds + SP * SP;
            lds_barrier();                               // every column is built: b is dead
            const bool holds_p2 = builder && (kSideBySide ? my_half == 1 : true);
            const bool holds_p1 = builder && my_half == 0;
            if (holds_p2) {
                const R (&x2)[NC][NR] = x[kSideBySide ? 0 : 1];
#pragma unroll
                for (int k = 0; k < NC; ++k) {
#pragma unroll
                    for (int r = 0; r < NR; ++r) Fl[(colbase + k * CS) * SP + j0 + r] = x2[k][r];
                    if (part == 0) eF[colbase + k * CS] = expo[kSideBySide ? 0 : 1][k];
                }
            }
            lds_barrier();
            if (holds_p1) {
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int col = colbase + k * CS;
                    int tj[NR], top = kNever;
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int e = eF[j0 + r];
                        tj[r] = (x[0][k][r] > (R)0 && e > kNoMass / 2) ? e + exponent_of(x[0][k][r]) : kNever;
                        top = max(top, tj[r]);
                    }
                    if (PH >= 2) top = max(top, dpp_mov<0xB1>(top));
                    if (PH >= 4) top = max(top, dpp_mov<0x4E>(top));
                    if (PH >= 8) top = max(top, dpp_mov<0x141>(top));
                    if (PH >= 16) top = max(top, dpp_mov<0x140>(top));
                    const bool alive = top > -(1 << 27) && expo[0][k] > kNoMass / 2;
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        Wt[(j0 + r) * SP + col] = (alive && tj[r] > -(1 << 27)) ? scale2(x[0][k][r], eF[j0 + r] - top) : (R)0;
                    if (part == 0) eW[col] = alive ? expo[0][k] + top : kNoMass;
                }
            }
            lds_barrier();
            if (wave < NT) {
                using acc_t = typename M::acc_t;
                const int ci = 16 * wave + i;              // my column of P (i = lane & 15, g = lane >> 4)
                acc_t acc[NT];
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) acc[mt] = acc_t{0, 0, 0, 0};
#pragma unroll 4
                for (int kk = 0; kk < SP / 4; ++kk) {
                    const R bv = Wt[(4 * kk + g) * SP + ci];
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt) acc[mt] = M::mma(Fl[(4 * kk + g) * SP + 16 * mt + i], bv, acc[mt]);
                }
                R sig = 0;
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) sig += (acc[mt][0] + acc[mt][1]) + (acc[mt][2] + acc[mt][3]);
                sig = add_xor<16>(sig);
                sig = add_xor<32>(sig);
                const int ew = eW[ci];
                const bool ok = ew > kNoMass / 2 && sig > (R)0;
                const int e = ok ? rescale_exponent(sig) : 0;
                R* __restrict__ dst = bt.op + ((long long)tile * SP + ci) * SP;
#pragma unroll
                for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[16 * mt + M::row(lane, r)] = ok ? scale2(acc[mt][r], -e) : (R)0;
                if (g == 0) bt.opexp[(long long)tile * SP + ci] = ok ? ew + e : kNoMass;
            }
        }
    }
    VBX_STAMP();
#ifdef VBX_PHASE_CLOCKS
    if ((tile % 1000) == 1 && (lane == 0) && st.n_iters == 3)
        printf("chunk_loglik wave %d: mfma %lld  epilogue %lld  wait %lld  operators built %lld  composed / stored %lld cycles\n", wave,
               clk[1] - clk[0], clk[2] - clk[1], clk[3] - clk[2], clk[4] - clk[3], clk[5] - clk[4]);
#endif
}

}  // namespace vbx
