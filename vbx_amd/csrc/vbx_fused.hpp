// vbx_fused.hpp -- per-chunk kernels that keep a chunk's lattice in LDS.
//
// chunk_post_kernel (one workgroup = one chunk of kTileFrames frames):
//     re-run of the chunk from its boundary vectors (scan3 of vbx_scan.hpp), posteriors and prior
//     statistics (VBx.py:101-103, 173-174) and the NEXT iteration's M-step accumulation gamma^T rho
//     (VBx.py:96) -- the forward / backward vectors never leave the CU.
//
#pragma once
#include <type_traits>
#include "vbx_scan.hpp"

// Build with -DVBX_PHASE_CLOCKS to make one workgroup in a thousand print the shader-clock cycles it spent
// in every phase of the two kernels below (the numbers quoted in DESIGN.md section 10 come from this).
#ifdef VBX_PHASE_CLOCKS
#define VBX_CLOCKS_DECL() long long clk[8]; int nclk = 0
#define VBX_STAMP() clk[nclk++] = clock64()
#else
#define VBX_CLOCKS_DECL()
#define VBX_STAMP()
#endif

namespace vbx {

// Does the chunk_post kernel's LDS footprint fit the CU?  (3 lattices of kTileFrames x SP)
template <typename R, int SP> struct ChunkPostCfg {
    static constexpr int kLattice = kTileFrames * SP;
    static constexpr int kBytes = 3 * kLattice * (int)sizeof(R) + 8192;
    static constexpr bool kFits = kBytes <= 160 * 1024;
};

template <typename R, int SP>
__global__ __launch_bounds__(256, (SP * (int)sizeof(R) <= 128 ? 3 : 1)) void chunk_post_kernel(BatchView<R> bt) {
    using M = Mfma16<R>;
    using acc_t = typename M::acc_t;
    using R2 = typename Vec<R>::v2;
    using R4 = typename Vec<R>::v4;
    constexpr int NREG = SP / 16;                      // states per lane in the re-run
    constexpr int NT = SP / 16;                        // M-tiles (speakers) of the accumulation
    constexpr int LAT = kTileFrames * SP;
    constexpr int KS = kTileFrames / 4;                // MFMA k-steps per chunk
    __shared__ __attribute__((aligned(16))) R btile[LAT];
    __shared__ __attribute__((aligned(16))) R af[LAT];
    __shared__ __attribute__((aligned(16))) R bf[LAT];
    __shared__ R sfl[kTileFrames];                     // sig_t = sum(a_t) of the row stored in af
    __shared__ R qfl[kTileFrames];                     // q_t: every element of the row stored in bf is >= q_t > 0
    __shared__ R tl_sig[2][2];                         // per forward task: scale at its end / on entry
    __shared__ int tl_expo[2];
    __shared__ __attribute__((aligned(16))) R c_l[SP];
    __shared__ __attribute__((aligned(16))) R aprev0[SP];
    __shared__ double ent_w[4][SP];
    __shared__ double red[16];

    const int tile = blockIdx.x;
    const int rec = bt.tile_rec[tile];
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int Dp = bt.Dp;
    const int t0 = bt.tile_t0[tile];
    const int len = min(kTileFrames, rd.T - t0);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i16 = lane & 15, g4 = lane >> 4;
    const R lp = (R)rd.lp;
    const R* __restrict__ rho = bt.rho + rd.row0 * Dp;

    VBX_CLOCKS_DECL();
    VBX_STAMP();
    // ---- phase 0: the chunk's shifted likelihoods go to LDS --------------------------------------
    stage_to_lds<(kTileFrames * SP / 4 + 255) / 256>(reinterpret_cast<R4*>(btile),
                                                     reinterpret_cast<const R4*>(bt.bmat + (rd.row0 + t0) * SP),
                                                     len * SP / 4, tid, 256);
    if (tid < SP)
        c_l[tid] = (tid < rd.S) ? (R)((1.0 - rd.lp) * bt.pi[(long long)rec * SP + tid] + 1e-8) : (R)0;
    const bool chunk0 = (t0 == 0);
    const int r0 = chunk0 ? 1 : 0;                     // first row the forward groups consume
    __syncthreads();
    VBX_STAMP();

    // rho fragments of d-slab `wave` (B operand of the accumulation at the end), requested after the wave's
    // re-run task so that the 2*KS registers are not live across its loop.
    R2 bv[KS];
    auto load_slab = [&](int slab) {
        // rows past the end of the recording are read as they lie (the next recording, or the zero rows behind
        // the last one): their gamma is zero, so the value does not matter, unconditional loads all stay in
        // flight together, and every address is one uniform base + a per-lane offset
        const R* __restrict__ src = rho + (long long)t0 * Dp + 32 * slab;
        if (slab * 32 < Dp) {
#pragma unroll
            for (int u = 0; u < KS; ++u) bv[u] = *reinterpret_cast<const R2*>(src + 4 * u * Dp + g4 * Dp + 2 * i16);
        }
    };

    // ---- re-run from the boundary vectors (VBx.py:167-171 in the linear domain) -----------------------
    // One task = (scan chunk, direction) = one wavefront: with bt.spt == 2 the tile has two scan chunks of
    // kScanHalf frames, so waves 0..3 = (first half, forward) (first half, backward) (second half, forward)
    // (second half, backward); with bt.spt == 1 waves 0-1 cover the whole tile and waves 2-3 only prefetch.
    // A lone wavefront on a dependent instruction stream pays ~8-10 cycles per instruction (measured: 16 issue
    // slots per frame = 160 cycles), so the loops are written for instruction count.  They carry UNNORMALISED
    // vectors (no reciprocal on the chain),
    //     forward :  a_t = b_t (lp a_{t-1} + c s_{t-1}),   s_t = sum a_t              (af, sfl)
    //     backward:  x_{t-1} = lp b_t x_t + q_t,           q_t = sum c b_t x_t        (bf, qfl)
    // rescaled by an exact power of two every four frames (worst case a frame shrinks the scale by
    // min c = 1e-8).  A lane holds NREG adjacent states, 16 lanes a vector (the four 16-lane rows of the
    // wave do the same work and store the same values); rows of b are fetched four frames ahead.
    R c[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) c[r] = c_l[i16 * NREG + r];
    auto load_rows = [&](R (&dst)[4][NREG], int f, int dir) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < NREG; ++r) dst[k][r] = btile[(f + dir * k) * SP + i16 * NREG + r];
    };
    const int spt = bt.spt;
    const int half = spt == 2 ? wave >> 1 : 0;
    const int lo = spt == 2 ? half * kScanHalf : 0, hi = spt == 2 ? min(len, lo + kScanHalf) : len;
    const bool has_task = wave < 2 * spt && lo < hi;
    const long long chunk = (long long)tile * spt + half;
    if (has_task && (wave & 1) == 0) {
        R a[NREG];
        const R* __restrict__ bnd = bt.fbound + chunk * SP + i16 * NREG;
        const bool first = t0 + lo == 0;                     // frame 0 of the recording: a_0 = b_0 (ip + 1e-8), VBx.py:163
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            a[r] = bnd[r];
            if (lo == 0 && !first) aprev0[i16 * NREG + r] = a[r];      // a[t0-1] (any scale) for the statistics of frame t0
            if (first) a[r] *= btile[i16 * NREG + r];
        }
        R sig = a[0];
#pragma unroll
        for (int r = 1; r < NREG; ++r) sig += a[r];
        sig = allreduce_sum<16>(sig);
        if (first) {
#pragma unroll
            for (int r = 0; r < NREG; ++r) af[i16 * NREG + r] = a[r];
            sfl[0] = sig;
        }
        int expo = 0;                                        // power-of-two bookkeeping of the total log-likelihood
        const R sig_in = sig;
        auto renorm = [&]() {                                // (rows already stored keep their own scale: a row
            const int e = rescale_exponent(sig);             //  of af is only ever used together with its sfl)
            expo += e;
            sig = scale2(sig, -e);
#pragma unroll
            for (int r = 0; r < NREG; ++r) a[r] = scale2(a[r], -e);
        };
        auto step = [&](const R (&b)[NREG], int f) {
#pragma unroll
            for (int r = 0; r < NREG; ++r) a[r] = b[r] * (lp * a[r] + c[r] * sig);
            R sm = a[0];
#pragma unroll
            for (int r = 1; r < NREG; ++r) sm += a[r];
            sig = allreduce_sum<16>(sm);
#pragma unroll
            for (int r = 0; r < NREG; ++r) af[f * SP + i16 * NREG + r] = a[r];
            sfl[f] = sig;
        };
        int f = first ? lo + 1 : lo;
        R cur[4][NREG], nxt[4][NREG];
        if (f + 4 <= hi) load_rows(cur, f, 1);
        for (; f + 4 <= hi; f += 4) {
            if (f + 8 <= hi) load_rows(nxt, f + 4, 1);
            renorm();
#pragma unroll
            for (int k = 0; k < 4; ++k) step(cur[k], f + k);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < NREG; ++r) cur[k][r] = nxt[k][r];
        }
        renorm();
        for (; f < hi; ++f) {                                // up to three left-over frames
            R b[NREG];
#pragma unroll
            for (int r = 0; r < NREG; ++r) b[r] = btile[f * SP + i16 * NREG + r];
            step(b, f);
        }
        if (lane == 0) {                                     // log of the product of this scan chunk's forward scales
            tl_sig[half][0] = sig;
            tl_sig[half][1] = first ? (R)1 : sig_in;
            tl_expo[half] = expo;
        }
    } else if (has_task) {
        R x[NREG];
        const R* __restrict__ bnd = bt.gbound + chunk * SP + i16 * NREG;     // backward vector at frame hi-1
        R part = 0;
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            x[r] = bnd[r];
            part += x[r];
        }
        part = allreduce_sum<16>(part);
        {
            const int e = rescale_exponent(part);
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                x[r] = scale2(x[r], -e);
                bf[(hi - 1) * SP + i16 * NREG + r] = x[r];
            }
            qfl[hi - 1] = scale2(part, -e) * (R)(1.0 / SP);  // a positive scale of the row, like q below
        }
        R q = 1;
        auto step = [&](const R (&b)[NREG], int f) {         // consumes row f, produces x_{f-1}
            R u[NREG];
#pragma unroll
            for (int r = 0; r < NREG; ++r) u[r] = b[r] * x[r];
            R qs = c[0] * u[0];
#pragma unroll
            for (int r = 1; r < NREG; ++r) qs += c[r] * u[r];
            q = allreduce_sum<16>(qs);
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                x[r] = lp * u[r] + q;
                bf[(f - 1) * SP + i16 * NREG + r] = x[r];
            }
            qfl[f - 1] = q;
        };
        int f = hi - 1;
        R cur[4][NREG], nxt[4][NREG];
        if (f - 4 >= lo) load_rows(cur, f, -1);
        for (; f - 4 >= lo; f -= 4) {                        // consumes rows f .. f-3 (all > lo)
            if (f - 8 >= lo) load_rows(nxt, f - 4, -1);
#pragma unroll
            for (int k = 0; k < 4; ++k) step(cur[k], f - k);
            const int e = rescale_exponent(q);               // (rows already stored keep their own scale)
#pragma unroll
            for (int r = 0; r < NREG; ++r) x[r] = scale2(x[r], -e);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < NREG; ++r) cur[k][r] = nxt[k][r];
        }
        for (; f >= lo + 1; --f) {                           // up to three left-over frames
            R b[NREG];
#pragma unroll
            for (int r = 0; r < NREG; ++r) b[r] = btile[f * SP + i16 * NREG + r];
            step(b, f);
        }
    }
    load_slab(wave);
    VBX_STAMP();
    __syncthreads();
    VBX_STAMP();

    // ---- posteriors and the "entered" statistic                               (VBx.py:101-103,174) --
    //   gamma_t = a_t x_t / sum;   entered_j += gamma_t[j] s_{t-1} / (lp a_{t-1}[j] + c_j s_{t-1}),  t >= 1
    // rows are brought to scale 1 first (a/s sums to 1, x/q >= 1 elementwise): no product can underflow.
    // Same lane layout as the re-run: 16 lanes x NREG states per frame, four frames per wavefront pass.
    {
        R* __restrict__ G = bt.gamma + (rd.row0 + t0) * SP;
        R ent[NREG];
#pragma unroll
        for (int r = 0; r < NREG; ++r) ent[r] = 0;
#pragma unroll
        for (int it = 0; it < kTileFrames / 16; ++it) {
            const int f = 16 * it + 4 * wave + g4;
            const bool ok = f < len;
            const int fr = ok ? f : 0;
            const R isig = fast_rcp(sfl[fr]), iq = fast_rcp(qfl[fr]);   // (applied one after the other: their product may overflow)
            const R sp = fr > 0 ? sfl[fr - 1] : tl_sig[0][1];
            R a[NREG], x[NREG], ap[NREG], g[NREG];
            load_pack<NREG>(a, af + fr * SP + i16 * NREG);
            load_pack<NREG>(x, bf + fr * SP + i16 * NREG);
            load_pack<NREG>(ap, (fr > 0 ? af + (fr - 1) * SP : aprev0) + i16 * NREG);
#pragma unroll
            for (int r = 0; r < NREG; ++r) g[r] = (a[r] * isig) * (x[r] * iq);
            R sum = g[0];
#pragma unroll
            for (int r = 1; r < NREG; ++r) sum += g[r];
            sum = allreduce_sum<16>(sum);
            const R inv = ok ? fast_rcp(sum) : (R)0;
            const bool stat = ok && t0 + f >= 1;               // frame 0 of the recording has no "entered" term
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                g[r] *= inv;
                const R term = g[r] * sp * fast_rcp(lp * ap[r] + c[r] * sp);
                ent[r] += stat ? term : (R)0;                  // (select, not multiply: ap is undefined for frame 0)
            }
            store_pack<NREG>(bf + f * SP + i16 * NREG, g);     // A operand of the accumulation below (0 past the end)
            if (ok) store_pack<NREG>(G + f * SP + i16 * NREG, g);
        }
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            double e = (double)ent[r];                     // <= 8 terms per lane in working precision
            e += __shfl_xor(e, 16, 64);
            e += __shfl_xor(e, 32, 64);
            if (g4 == 0) ent_w[wave][i16 * NREG + r] = e;
        }
        // this chunk's share of the total log-likelihood (VBx.py:173): log of the product of its forward
        // scales = log(s_end 2^expo / s_in), plus the row maxima taken out of the likelihoods
        double mpartial = 0.0;
        if (tid < len) mpartial = (double)bt.mrow[rd.row0 + t0 + tid];
        if (tid >= 128 && tid < 130 && (tid == 128 || (spt == 2 && len > kScanHalf))) {
            const int h = tid - 128;                         // one term per forward task
            mpartial += log((double)tl_sig[h][0]) - log((double)tl_sig[h][1]) + (double)tl_expo[h] * 0.69314718055994530942;
        }
        mpartial = block_sum(mpartial, red);               // (contains the barrier ent_w needs)
        if (tid < SP) {
            const double e = (ent_w[0][tid] + ent_w[1][tid]) + (ent_w[2][tid] + ent_w[3][tid]);
            bt.epart[(long long)tile * SP + tid] = tid < rd.S ? e : 0.0;
        }
        if (tid == 0) bt.tllpart[tile] = mpartial;
    }
    __syncthreads();
    VBX_STAMP();

    // ---- next M-step: C[s][d] = sum_t gamma[t][s] rho[t][d] on MFMA 16x16x4        (VBx.py:96) --
    // M index i of tile mu <-> speaker NT*i + mu (one vector LDS read feeds every tile);
    // N index j of half h <-> feature 32*slab + 2j + h (one 8/16-byte global load feeds both).
    for (int slab = wave; slab * 32 < Dp; slab += 4) {
        acc_t acc[NT][2];
        R nsum[NT];
#pragma unroll
        for (int mu = 0; mu < NT; ++mu) {
            acc[mu][0] = acc_t{0, 0, 0, 0};
            acc[mu][1] = acc_t{0, 0, 0, 0};
            nsum[mu] = 0;
        }
        const bool first = slab == wave;
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const int f = 4 * u + g4;
            R2 b2 = bv[u];
            if (!first) b2 = *reinterpret_cast<const R2*>(rho + (long long)t0 * Dp + 32 * slab + 4 * u * Dp + g4 * Dp + 2 * i16);
            R av[NT];
#pragma unroll
            for (int mu = 0; mu < NT; ++mu) av[mu] = bf[f * SP + NT * i16 + mu];
#pragma unroll
            for (int mu = 0; mu < NT; ++mu) {
                nsum[mu] += av[mu];
                acc[mu][0] = M::mma(av[mu], b2.x, acc[mu][0]);
                acc[mu][1] = M::mma(av[mu], b2.y, acc[mu][1]);
            }
        }
        R* __restrict__ part = bt.mpart + (long long)tile * SP * Dp;
#pragma unroll
        for (int mu = 0; mu < NT; ++mu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = NT * M::row(lane, r) + mu;
                *reinterpret_cast<R2*>(part + (long long)s * Dp + 32 * slab + 2 * i16) = R2{acc[mu][0][r], acc[mu][1][r]};
            }
        }
        if (slab == 0) {
#pragma unroll
            for (int mu = 0; mu < NT; ++mu) {
                R v = nsum[mu];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (g4 == 0) bt.npart[(long long)tile * SP + NT * i16 + mu] = v;
            }
        }
    }
    VBX_STAMP();
#ifdef VBX_PHASE_CLOCKS
    if ((blockIdx.x % 1000) == 1 && (lane == 0) && bt.state[rec].n_iters == 3)
        printf("chunk_post wave %d: stage %lld  rerun %lld  wait %lld  post %lld  mfma %lld cycles\n", wave,
               clk[1] - clk[0], clk[2] - clk[1], clk[3] - clk[2], clk[4] - clk[3], clk[5] - clk[4]);
#endif
}

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
