// vbx_chunk_post.hpp -- chunk_post: re-run of a chunk from its boundary vectors, posteriors, prior statistics,
// log-likelihood share and the NEXT iteration's M-step accumulation gamma^T rho   (VBx.py:96,101-103,167-174).
//
// One workgroup = one chunk (tile of kTileFrames frames):
//
//   stage    b of the tile -> LDS region `bl`
//   re-run   the tile is cut at frame H = kTileFrames / 2 and each half [lo, hi) is walked by two waves, one
//            forward from a_(lo-1), one backward from x_(hi-1): unnormalised vectors rescaled by powers of two every
//            four frames.  The outer vectors are the boundary vectors of the walk (scan2); the ones at the cut are one
//            product each with the half-tile operators chunk_loglik has left in `oph`:
//                a_(H-1) = P1 a_in        x_(H-1) = P2^T x_(len-1)
//            (64 dependent steps per chain instead of 128; a tile of at most H frames, or a batch without `oph`, is one
//            half walked by waves 0 and 1; a full tile that is not the first of its recording packs the two forward
//            chains into wave 0 and the two backward chains into wave 1, rows 0-1 / 2-3 of the wave: PACKED below).
//            The two chains of a half meet in its middle m: each stores only what the OTHER has not produced yet in
//            region `r1`, and everything after the crossing overwrites rows of b that both have consumed:
//                rows [lo, m) : a_f -> r1[f]  (forward, before the barrier)    x_f -> bl[f]  (backward, after it)
//                rows [m, hi) : x_f -> r1[f]  (backward, before the barrier)   a_f -> bl[f]  (forward, after it)
//   post     gamma ~ a x, "entered" statistic, log-likelihood share; gamma -> bl (A operand of the accumulation)
//   MFMA     C[s][d] = sum_t gamma[t][s] rho[t][d] on v_mfma 16x16x4, rho fetched a quarter of the chunk ahead.
//
// What leaves the workgroup: the partial sums of the tile (gamma^T rho, sum gamma, "entered", tll share) -- and NOT
// gamma: nothing in the iteration loop reads it (iter_fin needs row 0 only: gamma0).  The responsibilities a caller
// asks for are written by the REPLAY instance of the same kernel after the last iteration (re-run + posteriors only,
// from the b, boundary vectors and priors of each recording's last iteration, which stay in HBM untouched once it
// has stopped).  Measured on 64 recordings of T = 10 000, S = 30: 174 -> 165 us per launch, 77 MB less written.
//
// Tried and dropped (round 2): one workgroup walking a RUN of several tiles with the accumulators kept in registers
// across them (one partial per run instead of per tile) and b of the next tile prefetched into `r1` during the
// accumulation.  As a loop the compiler keeps per-lane addresses of the whole body live across it (100-130 spilled
// registers at the 128-register budget of four workgroups per CU; laundering the lane index per tile and unrolling
// the loop completely brings that to ~25); measured 182 us (two tiles per run), 226 us (three), 277-333 us (five:
// one balanced round of persistent workgroups also puts the phases of the whole chip in lock-step) against 165 us.
//
// (What worked in the end is the half-tile scheme above: two operators per tile from chunk_loglik, 8 KB per tile.)
// Also tried and dropped (round 2): the tile cut into four SUB-CHUNKS of 32 frames that re-run side by side in the four
// 16-lane rows of the two re-run waves (32 dependent steps instead of 128; elimination runs of the re-run cut to 64 /
// 32 / 4 frames: 134 / 119 / 107 us).  The vectors at the three inner edges need the sub-chunks' S x S transfer
// operators; built here from the b tile (vbx_operator.hpp, all four waves) and pushed through by three mat-vecs per
// direction the kernel took 190 us: the operator build is S^2 FMAs per frame of VALU work -- the same 40-45 us it
// costs chunk_loglik, chip-wide VALU throughput, not latency -- and the edge chain + two more barriers eat what is
// left of the gain.  Handing the operators over from chunk_loglik instead would add 16 KB per tile to both kernels'
// traffic (+20 %) for an estimated 135 us.
//
// LDS: 2 regions of kTileFrames x SP + ~3 KB: 35 KB at SP = 32 (f32) -> four workgroups per CU.
// Instrumentation build: -DVBX_PHASE_CLOCKS (per-workgroup phase stamps, tools/phase_timeline.py).
#pragma once
#include <type_traits>
#include "vbx_scan.hpp"
#include "vbx_split.hpp"

// 0: the four chains of a split tile on four waves (A/B builds)
#ifndef VBX_POST_PACKED
#define VBX_POST_PACKED 1
#endif

namespace vbx {

#ifdef VBX_PHASE_CLOCKS
constexpr int kClockTiles = 8192;
__device__ long long g_phase_clocks[kClockTiles * 16];
#endif

template <typename R, int SP> struct ChunkPostCfg {
    static constexpr int kBytes = 2 * kTileFrames * SP * (int)sizeof(R) + 6144;
    static constexpr bool kFits = kBytes <= 160 * 1024;
    static constexpr int kPerCU = kBytes <= 32 * 1024 ? 5 : kBytes <= 40 * 1024 ? 4 : kBytes <= 53 * 1024 ? 3
                                  : kBytes <= 80 * 1024 ? 2 : 1;
};

// REPLAY: write gamma only (see above).  SPLIT (fp32 only): gamma^T rho on v_mfma_f32_16x16x32_f16 with f16 operand pairs
// (vbx_split.hpp) -- rho from its fragment-ordered copy rho_b, gamma split where it is computed.
// FOLD (round 6, small batches): the last level of the boundary walk -- the steps INSIDE a group of chunks, a launch of its own
// otherwise (scan2_kernel level 3) -- runs here: the workgroup stages the operators of its group that lie before and behind its
// chunk (at most kTileFrames / SP of them: the region r1 is free until the re-run starts), wave 0 walks forward from the group's
// left edge, wave 1 backward from its right edge, with the very arithmetic of scan2 (walk_step).  One dependent launch fewer
// per iteration where an iteration IS its launches (one recording of T = 10 000: 51.7 us in six launches).
// The FOLD instances are the small-batch instances altogether: in split mode they also request the wave's whole rho slab of the
// accumulation in one go, when the re-run is over and before the posterior pass (its registers exist anyway: occupancy stays at
// four; round 5 requested one k-step there and the other three after the pass): with a few hundred workgroups on the chip a
// launch streams at (bytes in flight) / (latency).  Measured (8 recordings): 66.2 -> 64.1 us per iteration.  NOT in exact f32
// (232 registers: two workgroups per CU, i.e. 512 slots for the 632 workgroups of eight recordings: 70.9 -> 79.9 us) nor in
// fp64.  Since the end of round 6 every split instance with SP >= 32 does the same (kAllAhead below).
template <typename R, int SP, bool REPLAY, bool SPLIT = false, bool FOLD = false>
__global__ __launch_bounds__(256, (ChunkPostCfg<R, SP>::kPerCU)) void chunk_post_kernel(BatchView<R> bt) {
    static_assert(!SPLIT || (sizeof(R) == 4 && !REPLAY), "the split GEMM is a mode of the fp32 iteration");
    using M = Mfma16<R>;
    using acc_t = typename M::acc_t;
    using R2 = typename Vec<R>::v2;
    using R4 = typename Vec<R>::v4;
    constexpr int NREG = SP / 16;                      // states per lane in the re-run
    constexpr int NT = SP / 16;                        // M-tiles (speakers) of the accumulation
    constexpr int KS = kTileFrames / 4;                // MFMA k-steps per chunk
    constexpr int LAT = kTileFrames * SP;
    constexpr int NST = (LAT / 4 + 255) / 256;         // 16/32-byte vectors of a b tile per thread
    __shared__ __attribute__((aligned(16))) R region[2][LAT];
    __shared__ R sfl[kTileFrames];                     // s_f = sum(a_f) of the stored forward row
    __shared__ R qfl[kTileFrames];                     // q_f: every element of the stored backward row is >= q_f > 0
    __shared__ R tl_sig[2][2];                         // per forward chain: its final and its initial vector sum
    __shared__ int tl_expo[2];
    __shared__ __attribute__((aligned(16))) R mv_w[2][SP];   // operands of the two products at the cut
    __shared__ __attribute__((aligned(16))) R c_l[SP];
    __shared__ __attribute__((aligned(16))) R aprev0[SP];
    __shared__ double ent_w[4][SP];
    __shared__ double red[16];
    __shared__ R nsum_w[SPLIT ? 4 : 1][SP];            // SPLIT: sum of gamma per wave and speaker (taken from registers)

    // (without a table the tiles are taken from the last one down: chunk_loglik has just written b and the half-tile operators
    //  front to back, so the ones written last -- still in the memory-side cache -- are read first; VBX_POST_REVERSE=0: A/B)
#ifndef VBX_POST_REVERSE
#define VBX_POST_REVERSE 1
#endif
    // (round 6: only when the launch's working set is beyond the L2s -- 80 KB per tile against 8 x 4 MB.  A smaller batch keeps the
    //  tile -> XCD map of chunk_loglik, block b on XCD b % 8, so that b and the half-tile operators are found in the L2 that wrote
    //  them: one recording 46.9 -> 45.8 us per iteration, fp64 69.1 -> 67.1, two recordings 49.5 -> 48.2, eight: no difference)
    const int tile = (VBX_POST_REVERSE && !REPLAY && !bt.tile_order && bt.ntiles_total > 512) ? bt.ntiles_total - 1 - (int)blockIdx.x
                                                                                              : tile_of_block(bt, blockIdx.x);
    if (tile < 0 || (!REPLAY && bt.tile_done[tile])) return;
    VBX_CLOCKS_DECL();
    // (the wave index as a scalar: roles, frame ranges and loop bounds of the re-run stay in SGPRs)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i16 = lane & 15, g4 = lane >> 4;
    const int so = i16 * NREG;                         // first state of this lane
    const int Dp = bt.Dp;
    {
        // one scalar load gives every address of the first round of vector loads (tile table of vbx_host_batch.hpp)
        const int4 td = bt.tile_desc[tile];            // {recording, t0, frames, first row}
        const int rec = td.x, t0 = td.y, len = td.z;
        const long long trow = td.w;
        constexpr int H = kTileFrames / 2;
        constexpr int kNoMass = -(1 << 24), kNever = -(1 << 28);
        // roles: wave 0 forward / wave 3 backward over the first half, wave 2 forward / wave 1 backward over the second
        const bool split = bt.oph != nullptr && len > H;   // (uniform)
        const int half = (split && (wave == 1 || wave == 2)) ? 1 : 0;
        const bool is_fwd = wave == 0 || (split && wave == 2), is_bwd = wave == 1 || (split && wave == 3);
        const int lo = half * H, hi = split && half == 0 ? H : len;
        const int m0 = split ? H / 2 : len / 2, m1 = H + (len - H) / 2;      // where the chains of a half cross
        const int m = half ? m1 : m0;
        auto low_part = [&](int f) { return split && f >= H ? f < m1 : f < m0; };   // a_f in r1 and x_f in bl?
        // PACKED (a full tile that is not the first of its recording, re-run as halves): the two forward chains share
        // wave 0 and the two backward chains wave 1 -- rows 0-1 of the wave walk the first half, rows 2-3 the second,
        // same instruction stream, LDS rows offset by H frames per lane -- while waves 2 and 3 only compute the vectors at
        // the cut.  Half as many chain waves compete for a SIMD and the re-run issues half the vector instructions.
        const bool chunk0 = (t0 == 0);
        const bool packed = VBX_POST_PACKED && split && len == kTileFrames && !chunk0;     // (uniform)
        const int hl = packed ? (g4 >> 1) : 0;             // the half this lane's row walks in a packed wave
        const int hoff = hl * H, soh = so + hoff * SP;     // ... as a frame offset / an offset into a lattice region
        const int clo = packed ? 0 : lo, chi = packed ? H : hi, cm = packed ? H / 2 : m;   // chain range and crossing
        const bool run_fwd = packed ? wave == 0 : is_fwd, run_bwd = packed ? wave == 1 : is_bwd;
        const double lp_d = bt.recs[rec].lp;
        const int n_spk = bt.recs[rec].S;
        const R lp = (R)lp_d;
        const R* __restrict__ rho = bt.rho + bt.recs[rec].rho_row0 * Dp;
        R* const bl = region[0];                       // b, then a (rows >= m) / x (rows < m), then gamma
        R* const r1 = region[1];                       // a_f below the crossing of its half, x_f from it on
        VBX_STAMP();
        // everything a re-run wave needs from HBM is requested before the b tile, so that it is there when the tile is:
        // the boundary vector and, for the waves that start at the cut, their quarter of the operator (the four 16-lane
        // rows of a wave split the sum of the product)
        constexpr int QS = SP / 4;                         // columns (forward) / rows (backward) per 16-lane row
        R bnd_v[NREG], opf[QS][NREG], opb[NREG][QS];
        int ope[NREG];
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            bnd_v[r] = 0;
            ope[r] = 0;
        }
        // FOLD: the operators between the group's edges and this chunk, requested now, stored to r1 behind the b tile's loads
        constexpr int kFoldOps = kTileFrames / SP;         // operators r1 can hold
        constexpr int VPO = SP * SP / 4, NVF = FOLD ? SP / 8 : 1;
        __shared__ int fexp[FOLD ? kTileFrames : 1];
        __shared__ R fvec[FOLD ? 2 : 1][SP];
        __shared__ R fwl[FOLD ? 2 : 1][SP];
        R4 ftmp[NVF];
        int fe = 0, nf = 0, nb = 0, ga = 0, gb = 0;
        if constexpr (FOLD) {
            const int G = bt.sgroup, cb0 = bt.recs[rec].tile0, K = bt.recs[rec].ntiles;
            const int jpos = (tile - cb0) % G;
            ga = tile - jpos;
            gb = min(ga + G, cb0 + K);
            nf = jpos;                                     // forward steps: operators ga .. tile - 1
            nb = gb - 1 - tile;                            // backward steps: operators gb - 1 .. tile + 1
            const int nops = nf + nb;                      // <= G - 1 <= kFoldOps (the host folds only then)
            auto op_of = [&](int slot) { return slot < nf ? ga + slot : gb - 1 - (slot - nf); };
#pragma unroll
            for (int u = 0; u < NVF; ++u) {
                const int v = u * 256 + tid, slot = v / VPO, e = v % VPO;
                ftmp[u] = reinterpret_cast<const R4*>(bt.op + (long long)op_of(slot < nops ? slot : 0) * SP * SP)[e];
            }
            if (tid < kTileFrames) {
                const int slot = tid / SP;
                fe = bt.opexp[(long long)op_of(slot < nops ? slot : 0) * SP + tid % SP];
            }
        } else if (is_fwd || is_bwd) {
            const R* __restrict__ bnd = (is_fwd ? bt.fbound : bt.gbound) + (long long)tile * SP + so;
            load_pack<NREG>(bnd_v, bnd);
        }
        if (split && wave >= 2) {
            const long long o = (long long)tile * 2 + (wave == 2 ? 0 : 1);      // P1 for wave 2, P2 for wave 3
            const R* __restrict__ op = bt.oph + o * SP * SP;
            if (wave == 2) {
#pragma unroll
                for (int ii = 0; ii < QS; ++ii) load_pack<NREG>(opf[ii], op + (g4 * QS + ii) * SP + so);
            } else {
#pragma unroll
                for (int r = 0; r < NREG; ++r)
#pragma unroll
                    for (int q4 = 0; q4 < QS / 4; ++q4) {
                        R t4[4];
                        load_pack<4>(t4, op + (so + r) * SP + g4 * QS + 4 * q4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) opb[r][4 * q4 + e] = t4[e];
                    }
            }
#pragma unroll
            for (int r = 0; r < NREG; ++r) ope[r] = bt.ophexp[o * SP + so + r];
        }
        stage_to_lds<NST>(reinterpret_cast<R4*>(bl), reinterpret_cast<const R4*>(bt.bmat + trow * SP), len * SP / 4, tid, 256);
        if (tid < SP) {
            const double pj = (REPLAY ? bt.pi_prev : bt.pi)[(long long)rec * SP + tid];
            c_l[tid] = (tid < n_spk) ? (R)((1.0 - lp_d) * pj + 1e-8) : (R)0;
        }
        if constexpr (FOLD) {
#pragma unroll
            for (int u = 0; u < NVF; ++u) {
                const int v = u * 256 + tid;
                if (v / VPO < nf + nb) reinterpret_cast<R4*>(r1)[v] = ftmp[u];
            }
            if (tid < kTileFrames) fexp[tid] = fe;
        }
        __syncthreads();
        if constexpr (FOLD) {
            constexpr int HL = 64 / SP, NI = SP / HL;
            const int j = lane / HL, h = lane % HL;
            if (wave == 0) {                               // forward from the vector entering the group's first chunk
                R y = bt.fbound[(long long)ga * SP + j];
                for (int q = 0; q < nf; ++q) {
                    R opv[NI];
                    int ej;
                    walk_fetch<R, SP, 0>(opv, ej, r1 + q * SP * SP, fexp + q * SP, j, h);
                    y = walk_step<R, SP, 0>(y, opv, ej, fwl[0], j, h, n_spk);
                }
                if (h == 0) fvec[0][j] = y;
            } else if (wave == 1) {                        // backward from the vector at the end of the group's last chunk
                R y = bt.gbound[(long long)(gb - 1) * SP + j];
                for (int q = 0; q < nb; ++q) {
                    R opv[NI];
                    int ej;
                    walk_fetch<R, SP, 1>(opv, ej, r1 + (nf + q) * SP * SP, fexp + (nf + q) * SP, j, h);
                    y = walk_step<R, SP, 1>(y, opv, ej, fwl[1], j, h, n_spk);
                }
                if (h == 0) fvec[1][j] = y;
            }
            __syncthreads();                               // (also: r1 is free again before a chain writes to it)
            if (is_fwd || is_bwd) load_pack<NREG>(bnd_v, fvec[is_fwd ? 0 : 1] + so);
        }
        VBX_STAMP();

        // ---- re-run: wave 0 forward, wave 1 backward (VBx.py:167-171 in the linear domain) ----------------------
        // A lone wavefront on a dependent instruction stream pays ~8-10 cycles per instruction, so the loops are
        // written for instruction count.  They carry UNNORMALISED vectors (no reciprocal on the chain),
        //     forward :  a_t = b_t (lp a_{t-1} + c s_{t-1}),   s_t = sum a_t
        //     backward:  x_{t-1} = lp b_t x_t + q_t,           q_t = sum c b_t x_t
        // rescaled by an exact power of two every four frames (worst case a frame shrinks the scale by min c = 1e-8).
        // A lane holds NREG adjacent states, 16 lanes a vector; rows of b are fetched four frames ahead.
        R c[NREG];
#pragma unroll
        for (int r = 0; r < NREG; ++r) c[r] = c_l[so + r];
        auto load_rows = [&](R (&dst)[4][NREG], int f, int dir) {
#pragma unroll
            for (int k = 0; k < 4; ++k) load_pack<NREG>(dst[k], bl + (f + dir * k) * SP + soh);
        };
        // forward state
        R a[NREG], sig = 1, sig_in = 1;
        int expo = 0, ff = clo;
        auto f_renorm = [&]() {
            const int e = rescale_exponent(sig);
            expo += e;
            sig = scale2(sig, -e);
#pragma unroll
            for (int r = 0; r < NREG; ++r) a[r] = scale2(a[r], -e);
        };
        // (a chain stores into one region per phase -- below the crossing a goes to r1 and x over b, from it on the other
        //  way round -- so the region is a pointer set once per phase, not a select per frame)
        R* fdst = r1;
        R* xdst = r1;
        auto f_store = [&](int f) { store_pack<NREG>(fdst + f * SP + soh, a); sfl[f + hoff] = sig; };
        auto f_step = [&](const R (&b)[NREG], int f) {
#pragma unroll
            for (int r = 0; r < NREG; ++r) a[r] = b[r] * (lp * a[r] + c[r] * sig);
            R sm = a[0];
#pragma unroll
            for (int r = 1; r < NREG; ++r) sm += a[r];
            sig = allreduce_sum<16>(sm);
            f_store(f);
        };
        auto f_run = [&](int end) {                              // frames ff .. end-1
            R cu[4][NREG], nx[4][NREG];
            if (ff + 4 <= end) load_rows(cu, ff, 1);
            for (; ff + 4 <= end; ff += 4) {
                if (ff + 8 <= end) load_rows(nx, ff + 4, 1);
                f_renorm();
#pragma unroll
                for (int k = 0; k < 4; ++k) f_step(cu[k], ff + k);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int r = 0; r < NREG; ++r) cu[k][r] = nx[k][r];
            }
            f_renorm();
            for (; ff < end; ++ff) {
                R b[NREG];
                load_pack<NREG>(b, bl + ff * SP + soh);
                f_step(b, ff);
            }
        };
        // backward state: x = x_{fb} (unnormalised), produced by consuming rows > fb
        R x[NREG], q = 1;
        int fb = chi - 1;
        auto b_store = [&](int f) { store_pack<NREG>(xdst + f * SP + soh, x); qfl[f + hoff] = q; };
        auto b_step = [&](const R (&b)[NREG], bool store) {      // consumes row fb, produces x_{fb-1}
            R u[NREG];
#pragma unroll
            for (int r = 0; r < NREG; ++r) u[r] = b[r] * x[r];
            R qs = c[0] * u[0];
#pragma unroll
            for (int r = 1; r < NREG; ++r) qs += c[r] * u[r];
            q = allreduce_sum<16>(qs);
#pragma unroll
            for (int r = 0; r < NREG; ++r) x[r] = lp * u[r] + q;
            --fb;
            if (store) b_store(fb);
        };
        auto b_renorm = [&]() {
            const int e = rescale_exponent(q);
            q = scale2(q, -e);                                   // (x_{m-1} and its q wait in registers for the barrier)
#pragma unroll
            for (int r = 0; r < NREG; ++r) x[r] = scale2(x[r], -e);
        };

        if (is_fwd) {
            if (half == 0) {
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    a[r] = bnd_v[r];
                    if (!chunk0) aprev0[so + r] = a[r];          // a[t0-1] (any scale) for the statistics of frame t0
                    if (chunk0) a[r] *= bl[so + r];              // frame 0: a_0 = b_0 (ip + 1e-8), VBx.py:163
                }
            } else {
                // a_(H-1) = P1 a_in = sum_i (y_i 2^(E_i)) col_i, weights shifted by the largest exponent on the support of
                // y (as scan2 pushes a boundary vector through a chunk operator); row g4 of the wave sums its QS columns
                int tj[NREG], top = kNever;
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    tj[r] = (bnd_v[r] > (R)0 && ope[r] > kNoMass / 2) ? ope[r] + exponent_of(bnd_v[r]) : kNever;
                    top = max(top, tj[r]);
                }
                top = allreduce_max<16>(top);
                R w[NREG];
#pragma unroll
                for (int r = 0; r < NREG; ++r) w[r] = tj[r] > -(1 << 27) ? scale2(bnd_v[r], ope[r] - top) : (R)0;
                if (g4 == 0) store_pack<NREG>(mv_w[0] + so, w);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int r = 0; r < NREG; ++r) a[r] = 0;
#pragma unroll
                for (int ii = 0; ii < QS; ++ii) {
                    const R wi = mv_w[0][g4 * QS + ii];
#pragma unroll
                    for (int r = 0; r < NREG; ++r) a[r] += wi * opf[ii][r];
                }
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    a[r] = add_xor<16>(a[r]);
                    a[r] = add_xor<32>(a[r]);
                    if (so + r >= n_spk) a[r] = 0;               // padded speakers carry no mass
                }
            }
            if (packed && wave == 2) {                           // the vector at the cut goes to wave 0 through LDS
                __builtin_amdgcn_wave_barrier();                 // (every lane has read the weights in mv_w[0])
                if (g4 == 0) store_pack<NREG>(mv_w[0] + so, a);
            }
        } else if (is_bwd) {
            if (half == 1 || !split) {
#pragma unroll
                for (int r = 0; r < NREG; ++r) x[r] = bnd_v[r];
            } else {
                // x_(H-1) = P2^T x_(len-1):  (F^T g)_j = 2^(E_j) <col_j, g>, outputs rescaled by the largest exponent
                if (g4 == 0) store_pack<NREG>(mv_w[1] + so, bnd_v);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                int tj[NREG], top = kNever;
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    R tot = 0;
#pragma unroll
                    for (int ii = 0; ii < QS; ++ii) tot += opb[r][ii] * mv_w[1][g4 * QS + ii];
#ifdef VBX_CUT_VIA_BPERMUTE
                    // (A/B build, tools/hazard/bpermute_compare.py: with __shfl_xor here the compiler vectorises the product
                    //  over r into v_pk_fma_f32 ... op_sel:[0,1,0] -- the form of DESIGN section 6 -- and the build fails as round 3's did)
                    tot += __shfl_xor(tot, 16, 64);
                    tot += __shfl_xor(tot, 32, 64);
#else
                    tot = add_xor<16>(tot);
                    tot = add_xor<32>(tot);
#endif
                    x[r] = tot;
                    tj[r] = (tot > (R)0 && ope[r] > kNoMass / 2) ? ope[r] + exponent_of(tot) : kNever;
                    top = max(top, tj[r]);
                }
                top = allreduce_max<16>(top);
#pragma unroll
                for (int r = 0; r < NREG; ++r)
                    x[r] = (tj[r] > -(1 << 27) && so + r < n_spk) ? scale2(x[r], ope[r] - top) : (R)0;
            }
            if (packed && wave == 3) {                           // the vector at the cut goes to wave 1 through LDS
                __builtin_amdgcn_wave_barrier();
                if (g4 == 0) store_pack<NREG>(mv_w[1] + so, x);
            }
        }
        if (packed) {
            __syncthreads();
            if (wave == 0 && hl == 1) load_pack<NREG>(a, mv_w[0] + so);      // rows 2-3: the second half starts from a_(H-1)
            if (wave == 1 && hl == 0) load_pack<NREG>(x, mv_w[1] + so);      // rows 0-1: the first half starts from x_(H-1)
        }
        if (run_fwd) {
            sig = a[0];
#pragma unroll
            for (int r = 1; r < NREG; ++r) sig += a[r];
            sig = allreduce_sum<16>(sig);
            sig_in = sig;
            if (chunk0 && half == 0) {
                fdst = 0 < cm ? r1 : bl;                         // (row 0 of bl if len == 1: b_0 is not needed again)
                f_store(0);
                ff = 1;
            }
            fdst = r1;
            f_run(max(cm, ff));                                  // rows < m -> r1
        } else if (run_bwd) {
            R part = 0;
#pragma unroll
            for (int r = 0; r < NREG; ++r) part += x[r];
            part = allreduce_sum<16>(part);
            const int e = rescale_exponent(part);
#pragma unroll
            for (int r = 0; r < NREG; ++r) x[r] = scale2(x[r], -e);
            q = scale2(part, -e) * (R)(1.0 / SP);                // a positive scale of the row, like q of the steps
            b_store(chi - 1);                                    // -> r1 (hi-1 >= m always)
            // consume rows hi-1 .. max(m, lo+1); the outputs with index >= m go to r1, the last one (x_{m-1})
            // stays in registers until the barrier: its slot in bl still holds b_{m-1}, which the forward wave
            // may not have consumed yet
            const int stop = max(cm, clo + 1);
            R cu[4][NREG], nx[4][NREG];
            if (fb - 3 >= stop) load_rows(cu, fb, -1);
            while (fb - 3 >= stop) {                             // a block of four rows fb .. fb-3, all >= stop
                const bool last_block = fb - 4 < stop;           // its last output is x_{stop-1}
                if (fb - 7 >= stop) load_rows(nx, fb - 4, -1);
#pragma unroll
                for (int k = 0; k < 4; ++k) b_step(cu[k], !(last_block && k == 3) || stop - 1 >= cm);
                b_renorm();
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int r = 0; r < NREG; ++r) cu[k][r] = nx[k][r];
            }
            while (fb >= stop) {
                R b[NREG];
                load_pack<NREG>(b, bl + fb * SP + soh);
                b_step(b, fb - 1 >= cm);
            }
        }
        VBX_STAMP();
        __syncthreads();                                         // crossing: rows >= m of each half are consumed by its
                                                                 // backward wave, rows < m by its forward wave
        if (run_fwd) {
            fdst = bl;
            f_run(chi);                                          // rows >= m -> bl (over b_f, after reading it)
            const int th = packed ? hl : half;                   // (one lane per chain: lane 0, and lane 32 of a packed wave)
            if (i16 == 0 && (packed ? (g4 & 1) == 0 : g4 == 0)) {
                tl_sig[th][0] = sig;
                tl_sig[th][1] = (chunk0 && half == 0) ? (R)1 : sig_in;
                tl_expo[th] = expo;
            }
        } else if (run_bwd && cm > clo) {
            // x = x_{m-1} is in registers, rows m-1 .. lo+1 remain.  Every output x_{f-1} lands on b_{f-1}, the row
            // the NEXT step consumes, so rows are always in registers before their slot is written: up to three
            // leading single rows and the first block of four are fetched before the first store.
            const int n = fb - clo, tail = n & 3;                // fb == m - 1
            R lead[3][NREG], cu[4][NREG], nx[4][NREG];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < tail) load_pack<NREG>(lead[k], bl + (fb - k) * SP + soh);
            if (n - tail >= 4) load_rows(cu, fb - tail, -1);
            xdst = bl;
            b_store(fb);                                         // x_{m-1} -> bl[m-1]
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < tail) b_step(lead[k], true);
            if (tail) b_renorm();
            while (fb - clo >= 4) {                               // blocks of four rows fb .. fb-3 (fb - lo is a multiple of 4 here)
                if (fb - clo - 4 >= 4) load_rows(nx, fb - 4, -1);
#pragma unroll
                for (int k = 0; k < 4; ++k) b_step(cu[k], true);
                b_renorm();
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int r = 0; r < NREG; ++r) cu[k][r] = nx[k][r];
            }
        }
        // rho fragments of the first quarter of the accumulation (d-slab `wave`): in flight during the posterior phase
        constexpr int QK = KS / 4;                               // k-steps per quarter
        R2 bq[2][QK];
        // (rows past the end of the recording belong to the next recording or to the zero padding behind the last
        //  one: finite values that meet gamma = 0, so the addresses need no clamp -- uniform base + per-lane offset)
        const int lane_off = g4 * Dp + 2 * i16;
        auto load_quarter = [&](R2 (&dst)[QK], int slab, int qi) {
            const R* __restrict__ src = rho + (long long)t0 * Dp + 32 * slab;
#pragma unroll
            for (int u = 0; u < QK; ++u) dst[u] = *reinterpret_cast<const R2*>(src + 4 * (qi * QK + u) * Dp + lane_off);
        };
        // SPLIT: the B operand comes from rho_b, [slab][h][kk][hi | lo][lane] in 16-byte fragments (vbx_split.hpp): per
        // k-step of 32 frames a wave fetches the four fragments of its slab, one contiguous KB per load instruction
        h8 bs[4][2][2];                                          // [k-step][h][hi | lo]: a wave's whole slab (64 registers)
        const h8* __restrict__ rb = nullptr;
        if constexpr (SPLIT) {
            const RecDesc& rdd = bt.recs[rec];
            rb = reinterpret_cast<const h8*>(bt.rho_b) + (long long)(tile - rdd.tile0 + rdd.rho_tile0) * (kTileFrames * Dp / 4) + lane;
        }
        auto load_kstep = [&](int buf, int slab, int kk) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bs[buf][h][0] = rb[((long long)(2 * slab + h) * 4 + kk) * 128];
                bs[buf][h][1] = rb[((long long)(2 * slab + h) * 4 + kk) * 128 + 64];
            }
        };
        // Split mode: the wave's whole slab of the accumulation is requested NOW, between the re-run and the posterior pass (its
        // registers exist anyway; at SP = 16 the instance would spill).  Round 5 requested k-step 0 here and k-steps 1-3 after the
        // pass; with the streams of a big batch side by side the earlier request measured 0.2315 / 0.2319 / 0.2340 -> 0.2249 / 0.2258 / 0.2275 ms per step (three
        // A/B pairs in one call; on ONE stream 0.2603 -> 0.2629: the loads of a lone launch queue up behind each other), the C5
        // sweep 1.421 -> 1.408; small batches: see FOLD above
#ifndef VBX_POST_ALLAHEAD
#define VBX_POST_ALLAHEAD 1
#endif
        constexpr bool kAllAhead = (FOLD || (VBX_POST_ALLAHEAD && SP >= 32)) && SPLIT && !REPLAY;
        R2 bq_all[kAllAhead && !SPLIT ? 4 : 1][QK];
        if constexpr (SPLIT) {
            if (wave * 32 < Dp) {
                load_kstep(0, wave, 0);
                if constexpr (kAllAhead) {
#pragma unroll
                    for (int kk = 1; kk < 4; ++kk) load_kstep(kk, wave, kk);
                }
            }
        } else if constexpr (kAllAhead) {
            if (wave * 32 < Dp) {
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) load_quarter(bq_all[qi], wave, qi);
            }
        } else {
            if (!REPLAY && wave * 32 < Dp) load_quarter(bq[0], wave, 0);
        }
        VBX_STAMP();
        __syncthreads();
        VBX_STAMP();

        // ---- posteriors and the "entered" statistic                               (VBx.py:101-103,174) --
        //   gamma_t = a_t x_t / sum;   entered_j += gamma_t[j] s_{t-1} / (lp a_{t-1}[j] + c_j s_{t-1}),  t >= 1
        // rows are brought to scale 1 first (a/s sums to 1, x/q >= 1 elementwise): no product can underflow.
        // pass 1 reads (a, x, a of the previous frame) into registers, pass 2 writes gamma over bl: a row of bl may
        // hold the a or x another frame's pass 1 still needs.
        {
            constexpr int NIT = kTileFrames / 16;
            R gam[NIT][NREG], ent[NREG];
#pragma unroll
            for (int r = 0; r < NREG; ++r) ent[r] = 0;
            // kFast: a full tile re-run as two halves (all but the last tile of a recording).  Which region holds a_f / x_f
            // is then a compile-time property of the unrolled iteration (bit 5 of f = 16 it + 4 wave + g4), every frame
            // exists, and the address / select arithmetic of the general form -- half of its instructions -- goes away.
            const int f16 = 4 * wave + g4;
            auto pass1 = [&](auto fast_tag) {
                constexpr bool kFast = decltype(fast_tag)::value;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int f = 16 * it + f16;
                    const bool ok = kFast || f < len;
                    const int fr = ok ? f : 0;
                    const R isig = fast_rcp(sfl[fr]), iq = fast_rcp(qfl[fr]);   // (applied one after the other: their product may overflow)
                    R sp;
                    if (kFast && it > 0) sp = sfl[fr - 1];
                    else sp = fr > 0 ? sfl[fr - 1] : tl_sig[0][1];
                    R av[NREG], xv[NREG], ap[NREG];
                    bool lowf, lowp;
                    if constexpr (kFast) {
                        lowf = ((16 * it) & (kTileFrames / 4)) == 0;
                        lowp = f16 > 0 ? lowf : ((16 * it - 16) & (kTileFrames / 4)) == 0;
                    } else {
                        lowf = low_part(fr);
                        lowp = low_part(fr - 1);
                    }
                    load_pack<NREG>(av, (lowf ? r1 : bl) + fr * SP + so);
                    load_pack<NREG>(xv, (lowf ? bl : r1) + fr * SP + so);
                    if (!REPLAY) {
                        const R* app = (lowp ? r1 : bl) + (fr - 1) * SP;
                        if (!kFast || it == 0) app = fr == 0 ? aprev0 : app;
                        load_pack<NREG>(ap, app + so);
                    }
#pragma unroll
                    for (int r = 0; r < NREG; ++r) gam[it][r] = (av[r] * isig) * (xv[r] * iq);
                    R sum = gam[it][0];
#pragma unroll
                    for (int r = 1; r < NREG; ++r) sum += gam[it][r];
                    sum = allreduce_sum<16>(sum);
                    const R inv = ok ? fast_rcp(sum) : (R)0;
                    const bool stat = ok && t0 + f >= 1;           // frame 0 of the recording has no "entered" term
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        gam[it][r] *= inv;
                        if (!REPLAY) {
                            const R term = gam[it][r] * sp * fast_rcp(lp * ap[r] + c[r] * sp);
                            ent[r] += stat ? term : (R)0;          // (select, not multiply: ap is undefined for frame 0)
                        }
                    }
                }
            };
            if (split && len == kTileFrames) pass1(std::true_type{});
            else pass1(std::false_type{});
            if (REPLAY) {
                // the responsibilities themselves: [T][SP] in HBM (VBx.py:99,126)
                R* __restrict__ G = bt.gamma + trow * SP;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int f = 16 * it + 4 * wave + g4;
                    if (f < len) store_pack<NREG>(G + f * SP + so, gam[it]);
                }
                return;
            }
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                double e = (double)ent[r];                     // <= 8 terms per lane in working precision
                e = add_xor<16>(e);
                e = add_xor<32>(e);
                if (g4 == 0) ent_w[wave][so + r] = e;
            }
            double mpartial = 0.0;                             // this chunk's share of the total log-likelihood (VBx.py:173)
            if (tid < len) mpartial = (double)bt.mrow[trow + tid];
            if (tid == 128 || (tid == 129 && split)) {         // one term per forward chain
                const int h = tid - 128;
                // (f32: the sums carry 24 bits, v_log_f32 gives the difference to ~1e-7 absolute -- 1e-12 of an ELBO)
                if (sizeof(R) == 4)
                    mpartial += (double)(__logf((float)tl_sig[h][0]) - __logf((float)tl_sig[h][1]));
                else
                    mpartial += log((double)tl_sig[h][0]) - log((double)tl_sig[h][1]);
                mpartial += (double)tl_expo[h] * 0.69314718055994530942;
            }
            mpartial = block_sum(mpartial, red);               // (its barriers also end pass 1)
            if (tid < SP) {
                const double e = (ent_w[0][tid] + ent_w[1][tid]) + (ent_w[2][tid] + ent_w[3][tid]);
                bt.epart[(long long)tile * SP + tid] = tid < n_spk ? e : 0.0;
            }
            if (tid == 0) bt.tllpart[tile] = mpartial;
            if constexpr (SPLIT) {
                // The eight frames of this thread, f = 16 it + 4 wave + g4, are the k-slots (kk = wave, g = g4, e = it) of
                // rho_b's order, and its NREG states the rows i16 of the NREG M-tiles: the thread's gamma 2^14 IS lane
                // (16 g4 + i16)'s A fragment of k-step `wave`, one 16-byte store per M-tile and half.
                static_assert(NIT == 8, "one A fragment = eight frames");
                h8* const gfr = reinterpret_cast<h8*>(bl);         // [M-tile][k-step][hi | lo][lane]
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    h8 hi, lo;
                    R nloc = 0;
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        _Float16 a, b;
                        split_f16(gam[it][r] * (R)(1 << kSplitTop), a, b);
                        hi[it] = a;
                        lo[it] = b;
                        nloc += gam[it][r];
                    }
                    gfr[(r * 4 + wave) * 128 + lane] = hi;
                    gfr[(r * 4 + wave) * 128 + 64 + lane] = lo;
                    nloc = add_xor<16>(nloc);
                    nloc = add_xor<32>(nloc);
                    if (g4 == 0) nsum_w[wave][so + r] = nloc;
                }
            } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int f = 16 * it + 4 * wave + g4;
                store_pack<NREG>(bl + f * SP + so, gam[it]);       // A operand of the accumulation below (0 past the end)
            }
            }
            // iter_fin needs the responsibilities of frame 0 (VBx.py:102)
            if (chunk0 && wave == 0 && g4 == 0) store_pack<NREG>(bt.gamma0 + (long long)rec * SP + so, gam[0]);
            if constexpr (SPLIT) {
                // the other three k-steps of the wave's slab go into flight together, now that the registers of the posterior
                // pass are free: ONE exposed round trip in the accumulation instead of three (phase stamps, 64 recordings:
                // the accumulation phase was 11.6 k of a workgroup's 41 k cycles, all of it waiting for these loads one
                // k-step at a time)
                if (!kAllAhead && wave * 32 < Dp) {
#pragma unroll
                    for (int kk = 1; kk < 4; ++kk) load_kstep(kk, wave, kk);
                }
            }
        }
        lds_barrier();                 // (gamma and the sums meet in LDS; the partial sums stored above and the loads in flight need no wait)
        VBX_STAMP();

        // ---- next M-step: C[s][d] = sum_t gamma[t][s] rho[t][d] on MFMA 16x16x4        (VBx.py:96) --
        // M index i of tile mu <-> speaker NT*i + mu (one vector LDS read feeds every tile);
        // N index j of half h <-> feature 32*slab + 2j + h (one 8/16-byte global load feeds both).
        if constexpr (SPLIT) {
            if (tid < SP) bt.npart[(long long)tile * SP + tid] = (nsum_w[0][tid] + nsum_w[1][tid]) + (nsum_w[2][tid] + nsum_w[3][tid]);
            const h8* const gfr = reinterpret_cast<const h8*>(bl);
            const R descale = scale2((R)1, -(kSplitTop + bt.rho_e[bt.recs[rec].rho_rec]));
            for (int slab = wave; slab * 32 < Dp; slab += 4) {
                acc_t acc[NT][2];
#pragma unroll
                for (int mu = 0; mu < NT; ++mu) {
                    acc[mu][0] = acc_t{0, 0, 0, 0};
                    acc[mu][1] = acc_t{0, 0, 0, 0};
                }
                if (slab != wave) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) load_kstep(kk, slab, kk);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int mu = 0; mu < NT; ++mu) {
                        const h8 ah = gfr[(mu * 4 + kk) * 128 + lane], al = gfr[(mu * 4 + kk) * 128 + 64 + lane];
#pragma unroll
                        for (int h = 0; h < 2; ++h) acc[mu][h] = mfma_split(ah, al, bs[kk][h][0], bs[kk][h][1], acc[mu][h]);
                    }
                }
                R* __restrict__ part = bt.mpart + (long long)tile * SP * Dp;
#pragma unroll
                for (int mu = 0; mu < NT; ++mu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int s = NT * M::row(lane, r) + mu;
                        if (s < n_spk)         // (rows of padded speakers are zero: neither stored nor read back, fin_kernel)
                            *reinterpret_cast<R2*>(part + (long long)s * Dp + 32 * slab + 2 * i16) =
                                R2{acc[mu][0][r] * descale, acc[mu][1][r] * descale};
                    }
                }
            }
        } else
        for (int slab = wave; slab * 32 < Dp; slab += 4) {
            acc_t acc[NT][2];
            R nsum[NT];
#pragma unroll
            for (int mu = 0; mu < NT; ++mu) {
                acc[mu][0] = acc_t{0, 0, 0, 0};
                acc[mu][1] = acc_t{0, 0, 0, 0};
                nsum[mu] = 0;
            }
            if (slab != wave || kAllAhead) { if (!(kAllAhead && slab == wave)) load_quarter(bq[0], slab, 0); }
            auto quarter = [&](const R2 (&bfr)[QK], int qi) {
#pragma unroll
                for (int u = 0; u < QK; ++u) {
                    const int f = 4 * (qi * QK + u) + g4;
                    R av[NT];
                    load_pack<NT>(av, bl + f * SP + NT * i16);
#pragma unroll
                    for (int mu = 0; mu < NT; ++mu) {
                        if (slab == 0) nsum[mu] += av[mu];      // (sum of gamma: stored by slab 0 only)
                        acc[mu][0] = M::mma(av[mu], bfr[u].x, acc[mu][0]);
                        acc[mu][1] = M::mma(av[mu], bfr[u].y, acc[mu][1]);
                    }
                }
            };
            if (kAllAhead && slab == wave) {                     // the small-batch instances: the slab has been in flight since the start
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) quarter(bq_all[qi], qi);
            } else {
#pragma unroll 1
                for (int pair = 0; pair < 2; ++pair) {           // (not unrolled: bounds how many LDS reads are hoisted)
                    load_quarter(bq[1], slab, 2 * pair + 1);     // next quarter in flight
                    quarter(bq[0], 2 * pair);
                    if (pair == 0) load_quarter(bq[0], slab, 2);
                    quarter(bq[1], 2 * pair + 1);
                }
            }
            R* __restrict__ part = bt.mpart + (long long)tile * SP * Dp;
#pragma unroll
            for (int mu = 0; mu < NT; ++mu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int s = NT * M::row(lane, r) + mu;
                    if (s < n_spk)             // (rows of padded speakers are zero: neither stored nor read back, fin_kernel)
                        *reinterpret_cast<R2*>(part + (long long)s * Dp + 32 * slab + 2 * i16) = R2{acc[mu][0][r], acc[mu][1][r]};
                }
            }
            if (slab == 0) {
#pragma unroll
                for (int mu = 0; mu < NT; ++mu) {
                    R v = nsum[mu];
                    v = add_xor<16>(v);
                    v = add_xor<32>(v);
                    if (g4 == 0) bt.npart[(long long)tile * SP + NT * i16 + mu] = v;
                }
            }
        }
        VBX_STAMP();
#ifdef VBX_PHASE_CLOCKS
        // plain stores into a module-scope table (printf would fence the whole XCD): read with vbx_debug_clocks()
        // (one stream group only: tile numbers index the table)
        if (lane == 0 && (wave == 0 || wave == 2) && bt.state[rec].n_iters == 3 && tile < kClockTiles) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            long long* dstc = g_phase_clocks + ((long long)tile * 2 + (wave >> 1)) * 8;
            for (int k = 0; k < 7; ++k) dstc[k] = clk[k];
            dstc[7] = hw;
        }
#endif
    }
}

}  // namespace vbx
