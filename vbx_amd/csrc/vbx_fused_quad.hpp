// vbx_fused_quad.hpp -- chunk_post for four tiles per workgroup.
//
// What the phase clocks of chunk_post_mid_kernel showed (tools/phase_timeline.py): a workgroup lives 65 k cycles,
// half of them in the re-run, where two of its four waves walk 64 + 64 dependent frames with 16 of their 64
// lanes busy, and the other half in three memory round trips (stage b, gamma/statistics, rho for gamma^T rho)
// that nothing in the workgroup overlaps.  This kernel changes the shape of the workgroup instead of the code
// of the recursion:
//
//   * one wavefront re-runs FOUR tiles, one per 16-lane row (the recursion of a tile needs 16 lanes at SP <= 32):
//     the same ~16 issue slots per frame now advance four tiles.  Wave 0 = forward, wave 1 = backward,
//     meet-in-the-middle layout of vbx_fused_mid.hpp with the midpoint fixed at frame 64;
//   * eight waves, one workgroup per CU, 256 registers per wave: every wave requests the rho fragments of its
//     share of the accumulation (one tile, every second k-step, all 128 features = 128 registers) behind the b
//     tiles, i.e. the 64 KB of rho per tile cross the memory system WHILE the re-run runs, and gamma goes from
//     the registers it is computed in straight into the MFMA as the A operand;
//   * the tile table tile_desc = {recording, t0, frames, first row} and the per-tile convergence flag tile_done
//     make every address of the first round of loads depend on one scalar load only.
//
// Short tiles (the tail of a recording) are padded with zero rows of b: the forward rows stay zero, the backward
// recursion of such a row is (re)started at its last frame by a rarely taken variant of the four-frame block.
// Results are those of chunk_post_kernel up to the order of the gamma^T rho sum over the frames of a tile (even and odd
// k-steps are accumulated by two waves and added at the end): tests/test_gpu_parity.py::test_chunk_post_variants_agree.
//
// Status: correct and tested, 171-175 us per launch of 64 recordings against 162-164 us of chunk_post_mid_kernel (the
// default), for the reasons in DESIGN.md section 10; kept behind VBX_OPT_POST_KERNEL = 2 as the base of a pipelined
// version.
#pragma once
#include <type_traits>

#include "vbx_fused_mid.hpp"

namespace vbx {

constexpr int kQuadTiles = 4;          // tiles per workgroup = 16-lane rows of a wavefront

template <typename R, int SP> struct ChunkPostQuadCfg {
    static constexpr int kSkew = SP;   // elements: shifts the tiles of a quad onto different LDS banks
    static constexpr int kBytes =
        kQuadTiles * ((2 * kTileFrames * SP + 3 * kSkew) * (int)sizeof(R) + 2 * kTileFrames * (int)sizeof(R) + 128 +
                      2 * SP * (int)sizeof(R)) + 8 * SP * 8 + 256;
    static constexpr bool kFits = sizeof(R) == 4 && SP <= 32 && kBytes <= 160 * 1024;
    static constexpr int kMaxDp = 128;  // two feature slabs of 64 live in registers
};

template <typename R, int SP>
__global__ __launch_bounds__(512) void chunk_post_quad_kernel(BatchView<R> bt) {
    using M = Mfma16<R>;
    using acc_t = typename M::acc_t;
    using R2 = typename Vec<R>::v2;
    using R4 = typename Vec<R>::v4;
    constexpr int TL = kQuadTiles;
    constexpr int NREG = SP / 16;                      // states per lane in the re-run
    constexpr int NT = SP / 16;                        // M-tiles (speakers) of the accumulation
    constexpr int HALF = kTileFrames / 2;
    constexpr int KS = kTileFrames / 4;                // MFMA k-steps per tile
    constexpr int NBLK = kTileFrames / 4;              // four-frame blocks of the re-run
    constexpr int SKEW = ChunkPostQuadCfg<R, SP>::kSkew;
    constexpr int TS = kTileFrames * SP + SKEW;        // tile stride of bl
    constexpr int HS = HALF * SP + SKEW;               // tile stride of afh / bfh
    __shared__ __attribute__((aligned(16))) R bl[TL * TS];     // b, then a (rows >= 64) / x (rows < 64), then gamma
    __shared__ __attribute__((aligned(16))) R afh[TL * HS];    // a_f,  f < 64
    __shared__ __attribute__((aligned(16))) R bfh[TL * HS];    // x_f,  f >= 64  (row f - 64)
    __shared__ R sfl[TL][kTileFrames];                 // s_f = sum(a_f) of the stored forward row
    __shared__ R qfl[TL][kTileFrames];                 // q_f: every element of the stored backward row is >= q_f > 0
    __shared__ int efl[TL][NBLK];                      // power-of-two exponent taken out of the forward rows of a block
    __shared__ R sgin[TL];                             // sum of the forward vector entering the tile (1 for frame 0)
    __shared__ R lp_l[TL];
    __shared__ __attribute__((aligned(16))) R c_l[TL][SP];
    __shared__ __attribute__((aligned(16))) R aprev0[TL][SP];
    __shared__ double ent_w[2 * TL][SP];
    __shared__ double mred[2 * TL];

    const int quad = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int so = i16 * NREG;                         // first state of this lane
    const int Dp = bt.Dp;

    // ---- round trip 1: the four tile descriptors and their convergence flags (scalar) -------------------------
    int t_rec[TL], t_t0[TL], t_len[TL], t_row[TL];
    int any = 0;
#pragma unroll
    for (int s = 0; s < TL; ++s) {
        const int4 d = bt.tile_desc[quad * TL + s];
        const int dn = bt.tile_done[quad * TL + s];
        t_rec[s] = d.x;
        t_t0[s] = d.y;
        t_len[s] = dn ? 0 : d.z;
        t_row[s] = d.w;
        any |= t_len[s];
    }
    if (!any) return;
    VBX_CLOCKS_DECL();
    VBX_STAMP();
    auto pick = [&](const int (&v)[TL], int s) { return s == 0 ? v[0] : s == 1 ? v[1] : s == 2 ? v[2] : v[3]; };

    // ---- round trip 2: everything else, critical loads first ---------------------------------------------------
    // b tiles (rows past the end of a tile are replaced by zeros below; the reads stay inside bmat + its padding)
    constexpr int NST = kTileFrames * SP / 4 / 512;    // R4 per thread and tile
    R4 stg[TL][NST];
#pragma unroll
    for (int s = 0; s < TL; ++s) {
        if (t_len[s] > 0) {
            const R4* __restrict__ src = reinterpret_cast<const R4*>(bt.bmat + (long long)t_row[s] * SP);
#pragma unroll
            for (int u = 0; u < NST; ++u) stg[s][u] = src[u * 512 + tid];
        }
    }
    // boundary vectors of the re-run (lane row g4 = tile slot)
    const int my_len = pick(t_len, g4), my_t0 = pick(t_t0, g4);
    R bnd[NREG];
    {
        const int tile = min(quad * TL + g4, bt.ntiles_total - 1);
        const R* __restrict__ src = (wave == 0 ? bt.fbound : bt.gbound) + (long long)tile * SP + so;
        if (wave < 2) load_pack<NREG>(bnd, src);
    }
    // transition parameters: thread (slot, state) for tid < TL * SP
    double pi_v = 0.0, lp_v = 0.0;
    int s_v = 0;
    if (tid < TL * SP) {
        const int rec = pick(t_rec, tid / SP);
        pi_v = bt.pi[(long long)rec * SP + tid % SP];
        lp_v = bt.recs[rec].lp;
        s_v = bt.recs[rec].S;
    }
    // this wave's share of the accumulation: tile slot wave / 2, feature slabs wave % 2 and wave % 2 + 2
    const int ps = wave >> 1, sub = wave & 1;
    const int p_len = pick(t_len, ps), p_row = pick(t_row, ps), p_t0 = pick(t_t0, ps), p_rec = pick(t_rec, ps);
    const R* __restrict__ rsrc = bt.rho + (long long)p_row * Dp + g4 * Dp + 4 * i16;
    constexpr int NIT = kTileFrames / 8;               // k-steps (groups of four frames) per wave: u = 2 it + sub
    constexpr int NSL = 2;                             // feature slabs of 64 held in registers (Dp <= 128)
    R4 rq[NSL][NIT];                                   // 16 bytes per lane and load: 32 loads per wave
    auto load_rho = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) rq[0][it] = *reinterpret_cast<const R4*>(rsrc + 4 * (2 * it + sub) * Dp);
        // (a last slab of 32 features: lanes 8..15 read the head of the next row, their columns are never stored)
        if (Dp > 64) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) rq[1][it] = *reinterpret_cast<const R4*>(rsrc + 64 + 4 * (2 * it + sub) * Dp);
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) rq[1][it] = R4{0, 0, 0, 0};
        }
    };
    double mval = 0.0;                                 // log-likelihood shift of frame 64 * sub + lane of the tile
    if (64 * sub + lane < p_len) mval = (double)bt.mrow[(long long)p_row + 64 * sub + lane];

    // ---- LDS: b tiles (zero rows behind a short tile), c ------------------------------------------------------
#pragma unroll
    for (int s = 0; s < TL; ++s) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int idx = u * 512 + tid;
            R4 v = R4{0, 0, 0, 0};
            if (t_len[s] > 0) v = stg[s][u];
            if (idx * 4 >= t_len[s] * SP) v = R4{0, 0, 0, 0};
            reinterpret_cast<R4*>(bl + s * TS)[idx] = v;
        }
    }
    if (tid < TL * SP) {
        c_l[tid / SP][tid % SP] = (tid % SP < s_v) ? (R)((1.0 - lp_v) * pi_v + 1e-8) : (R)0;
        if (tid % SP == 0) lp_l[tid / SP] = (R)lp_v;
    }
    __syncthreads();
    VBX_STAMP();
#ifdef VBX_PHASE_CLOCKS
    const long long wall0 = wall_clock64();
#endif
    // rho: requested behind the b tiles (every CU of the chip is in this phase at the same time: requested
    // together, the 64 KB of rho per tile delay the 16 KB of b the re-run waits for), in flight during the re-run
    // The two re-run waves go first, while the CU's memory queue is empty: a request issued behind the 192 of the
    // other six waves stalls the issuing wave until those have drained (measured: 13 k cycles).
    if (wave >= 2) __builtin_amdgcn_s_sleep(32);
    if (p_len > 0) load_rho();

    // ---- re-run: wave 0 forward, wave 1 backward, lane row = tile ---------------------------------------------
    R* const bls = bl + g4 * TS + so;
    R c[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) c[r] = c_l[g4][so + r];
    const R lp = lp_l[g4], p_lp = lp_l[ps];
    auto load_rows = [&](R (&dst)[4][NREG], const R* base, int dir) {
#pragma unroll
        for (int k = 0; k < 4; ++k) load_pack<NREG>(dst[k], base + dir * k * SP);
    };
    auto copy_rows = [&](R (&dst)[4][NREG], const R (&src)[4][NREG]) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < NREG; ++r) dst[k][r] = src[k][r];
    };
    R cur[4][NREG], nxt[4][NREG];
    // forward state
    R a[NREG], sig = 1;
    int expo = 0;
    auto f_renorm = [&]() {
        const int e = rescale_exponent(sig);
        expo += e;
        sig = scale2(sig, -e);
#pragma unroll
        for (int r = 0; r < NREG; ++r) a[r] = scale2(a[r], -e);
    };
    auto f_finish = [&](R* a_dst, int f) {
        R sm = a[0];
#pragma unroll
        for (int r = 1; r < NREG; ++r) sm += a[r];
        sig = allreduce_sum<16>(sm);
        store_pack<NREG>(a_dst, a);
        sfl[g4][f] = sig;
    };
    auto f_step = [&](const R (&b)[NREG], R* a_dst, int f) {
#pragma unroll
        for (int r = 0; r < NREG; ++r) a[r] = b[r] * (lp * a[r] + c[r] * sig);
        f_finish(a_dst, f);
    };
    // backward state: x = x_f (unnormalised), q = its floor
    R x[NREG], q = 1, xinit[NREG], qinit = 1;
    unsigned startmask = 0;                            // blocks in which some row of a SHORT tile starts
    auto b_step = [&](auto slow, const R (&b)[NREG], R* x_dst, int p) {     // consumes row p + 1, produces x_p
        R u[NREG];
#pragma unroll
        for (int r = 0; r < NREG; ++r) u[r] = b[r] * x[r];
        R qs = c[0] * u[0];
#pragma unroll
        for (int r = 1; r < NREG; ++r) qs += c[r] * u[r];
        q = allreduce_sum<16>(qs);
#pragma unroll
        for (int r = 0; r < NREG; ++r) x[r] = lp * u[r] + q;
        if (decltype(slow)::value) {
            const bool st = (p == my_len - 1);
            q = st ? qinit : q;
#pragma unroll
            for (int r = 0; r < NREG; ++r) x[r] = st ? xinit[r] : x[r];
        }
        if (x_dst) {
            store_pack<NREG>(x_dst, x);
            qfl[g4][p] = q;
        }
    };
    auto b_renorm = [&]() {
        const int e = rescale_exponent(q);
        q = scale2(q, -e);
#pragma unroll
        for (int r = 0; r < NREG; ++r) x[r] = scale2(x[r], -e);
    };
    // one block of the backward recursion: consumes rows 4 blk + 3 .. 4 blk (cur), produces x_{4 blk + 2} .. x_{4 blk - 1}
    auto b_block = [&](int blk, R* dst_base, int dst_stride, bool keep_last) {
        // dst_base points at the slot of x_{4 blk + 2}; the slots of the following outputs are dst_stride apart
        if (__builtin_expect((startmask >> blk) & 1u, 0)) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                b_step(std::true_type{}, cur[k], (keep_last && k == 3) ? nullptr : dst_base + k * dst_stride, 4 * blk + 2 - k);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                b_step(std::false_type{}, cur[k], (keep_last && k == 3) ? nullptr : dst_base + k * dst_stride, 4 * blk + 2 - k);
        }
        b_renorm();
    };

    if (wave == 0) {
        const bool chunk0 = (my_t0 == 0);
        R* const afs = afh + g4 * HS + so;
        load_rows(cur, bls, 1);
        load_rows(nxt, bls + 4 * SP, 1);
#pragma unroll
        for (int r = 0; r < NREG; ++r) a[r] = my_len > 0 ? bnd[r] : (R)0;
        R sm = a[0];
#pragma unroll
        for (int r = 1; r < NREG; ++r) sm += a[r];
        sig = allreduce_sum<16>(sm);
        sgin[g4] = chunk0 ? (R)1 : sig;
        store_pack<NREG>(&aprev0[g4][so], a);                 // a[t0-1] (any scale) for the statistics of frame t0
        if (!chunk0) f_renorm();
        efl[g4][0] = expo;
        // frame 0 of the recording: a_0 = b_0 (ip + 1e-8), VBx.py:163; otherwise an ordinary step
#pragma unroll
        for (int r = 0; r < NREG; ++r) a[r] = cur[0][r] * (chunk0 ? a[r] : lp * a[r] + c[r] * sig);
        f_finish(afs, 0);
#pragma unroll
        for (int k = 1; k < 4; ++k) f_step(cur[k], afs + k * SP, k);
        copy_rows(cur, nxt);
#pragma unroll 1
        for (int blk = 1; blk < NBLK / 2; ++blk) {           // frames 4 .. 63 -> afh; the last prefetch is block 16 of bl
            load_rows(nxt, bls + (4 * blk + 4) * SP, 1);
            f_renorm();
            efl[g4][blk] = expo;
#pragma unroll
            for (int k = 0; k < 4; ++k) f_step(cur[k], afs + (4 * blk + k) * SP, 4 * blk + k);
            copy_rows(cur, nxt);
        }
    } else if (wave == 1) {
        R* const bfs = bfh + g4 * HS + so;
        R part = bnd[0];
#pragma unroll
        for (int r = 1; r < NREG; ++r) part += bnd[r];
        part = allreduce_sum<16>(part);
        const int e = rescale_exponent(part);
#pragma unroll
        for (int r = 0; r < NREG; ++r) xinit[r] = my_len > 0 ? scale2(bnd[r], -e) : (R)0;
        qinit = my_len > 0 ? scale2(part, -e) * (R)(1.0 / SP) : (R)1;   // a positive scale of the row, like q of the steps
#pragma unroll
        for (int s = 0; s < TL; ++s)
            if (t_len[s] > 0 && t_len[s] < kTileFrames) startmask |= 1u << (t_len[s] >> 2);
        const bool full = (my_len == kTileFrames);
#pragma unroll
        for (int r = 0; r < NREG; ++r) x[r] = full ? xinit[r] : (R)0;
        q = full ? qinit : (R)1;
        load_rows(cur, bls + (kTileFrames - 1) * SP, -1);
        store_pack<NREG>(bfs + (HALF - 1) * SP, x);           // x_127
        qfl[g4][kTileFrames - 1] = q;
#pragma unroll 1
        for (int blk = NBLK - 1; blk >= NBLK / 2; --blk) {    // rows 127 .. 64; x_126 .. x_64 -> bfh, x_63 stays in registers
            load_rows(nxt, bls + (4 * blk - 1) * SP, -1);     // (the last prefetch is rows 63 .. 60)
            b_block(blk, bfs + (4 * blk + 2 - HALF) * SP, -SP, blk == NBLK / 2);
            copy_rows(cur, nxt);
        }
    }
    VBX_STAMP();
    __syncthreads();                                          // midpoint: rows >= 64 of b are consumed by the backward
                                                              // wave, rows < 64 by the forward wave
    if (wave == 0) {
#pragma unroll 1
        for (int blk = NBLK / 2; blk < NBLK; ++blk) {         // frames 64 .. 127: a_f over b_f, after reading it
            if (blk + 1 < NBLK) load_rows(nxt, bls + (4 * blk + 4) * SP, 1);
            f_renorm();
            efl[g4][blk] = expo;
#pragma unroll
            for (int k = 0; k < 4; ++k) f_step(cur[k], bls + (4 * blk + k) * SP, 4 * blk + k);
            copy_rows(cur, nxt);
        }
    } else if (wave == 1) {
        // every output x_p lands on b_p, the row the NEXT step consumes: rows are in registers (cur, fetched before
        // the barrier, and nxt, fetched at the top of a block) before their slot is written
        store_pack<NREG>(bls + (HALF - 1) * SP, x);           // x_63
        qfl[g4][HALF - 1] = q;
#pragma unroll 1
        for (int blk = NBLK / 2 - 1; blk >= 1; --blk) {       // rows 63 .. 4
            load_rows(nxt, bls + (4 * blk - 1) * SP, -1);
            b_block(blk, bls + (4 * blk + 2) * SP, -SP, false);
            copy_rows(cur, nxt);
        }
        // rows 3, 2, 1 -> x_2, x_1, x_0 (row 0 has no predecessor inside the tile)
        if (startmask & 1u) {
#pragma unroll
            for (int k = 0; k < 3; ++k) b_step(std::true_type{}, cur[k], bls + (2 - k) * SP, 2 - k);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) b_step(std::false_type{}, cur[k], bls + (2 - k) * SP, 2 - k);
        }
    }
    VBX_STAMP();
#ifdef VBX_PHASE_CLOCKS
    const long long wall1 = wall_clock64();
#endif
    __syncthreads();
    VBX_STAMP();

    // ---- posteriors, the "entered" statistic and the next M-step          (VBx.py:96,101-103,174) --
    // wave = (tile slot ps, parity sub of the k-steps): lane (g4, i16) computes gamma of frame 4 u + g4, states
    // NT i16 + mu -- exactly the A operand of v_mfma_16x16x4 for k-step u, so gamma goes from the registers it
    // is computed in into C[s][d] += gamma[t][s] rho[t][d] without passing through LDS.  The two waves of a tile
    // add their halves of the k-steps at the end.  M index i of tile mu <-> speaker NT*i + mu; N index j of
    // quarter h <-> feature 64*slab + 4j + h.
    acc_t acc[NSL][NT][4];
    R nsum[NT];
#pragma unroll
    for (int mu = 0; mu < NT; ++mu) {
        nsum[mu] = 0;
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int h = 0; h < 4; ++h) acc[sl][mu][h] = acc_t{0, 0, 0, 0};
    }
    {
        const R* const bl_s = bl + ps * TS + so;
        const R* const af_s = afh + ps * HS + so;
        const R* const bf_s = bfh + ps * HS + so;
        R* __restrict__ G = bt.gamma + (long long)p_row * SP + so;
        R pc[NREG], ent[NREG];
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            pc[r] = c_l[ps][so + r];
            ent[r] = 0;
        }
        if (p_len > 0)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int f = 8 * it + 4 * sub + g4;              // it < 8 <=> f < 64
            const bool ok = f < p_len;
            const R isig = fast_rcp(sfl[ps][f]), iq = fast_rcp(qfl[ps][f]);   // (applied one after the other: their product may overflow)
            const R sp = f > 0 ? sfl[ps][max(f - 1, 0)] : sgin[ps];
            R av[NREG], xv[NREG], ap[NREG];
            load_pack<NREG>(av, (it < NIT / 2 ? af_s : bl_s) + f * SP);
            load_pack<NREG>(xv, it < NIT / 2 ? bl_s + f * SP : bf_s + (f - HALF) * SP);
            const R* app = it < NIT / 2 ? af_s + (f - 1) * SP : bl_s + (f - 1) * SP;
            if (it == NIT / 2 && f == HALF) app = af_s + (HALF - 1) * SP;
            if (it == 0 && f == 0) app = &aprev0[ps][so];
            load_pack<NREG>(ap, app);
            R g[NREG];
#pragma unroll
            for (int r = 0; r < NREG; ++r) g[r] = (av[r] * isig) * (xv[r] * iq);
            R sum = g[0];
#pragma unroll
            for (int r = 1; r < NREG; ++r) sum += g[r];
            sum = allreduce_sum<16>(sum);
            const R inv = fast_rcp(sum);
            const bool stat = ok && p_t0 + f >= 1;            // frame 0 of the recording has no "entered" term
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                g[r] = ok ? g[r] * inv : (R)0;                // (select: rows behind a short tile hold 0 * inf)
                const R term = g[r] * sp * fast_rcp(p_lp * ap[r] + pc[r] * sp);
                ent[r] += stat ? term : (R)0;
            }
            // (no branch inside the loop: one scheduling region, so that the MFMAs of one k-step can run under the
            //  vector work of the next; frames behind the end of the tile store into the dump)
            store_pack<NREG>(ok ? G + (long long)f * SP : bt.dump + so, g);
#pragma unroll
            for (int mu = 0; mu < NT; ++mu) {
                nsum[mu] += g[mu];
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
                    for (int h = 0; h < 4; ++h) acc[sl][mu][h] = M::mma(g[mu], rq[sl][it][h], acc[sl][mu][h]);
            }
        }
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            double e = (double)ent[r];                        // <= 16 terms per lane in working precision
            e += __shfl_xor(e, 16, 64);
            e += __shfl_xor(e, 32, 64);
            if (g4 == 0) ent_w[wave][so + r] = e;
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) mval += __shfl_xor(mval, sh, 64);
        if (lane == 0) mred[wave] = mval;
    }
    __syncthreads();                                          // the lattices are dead: bl becomes the exchange buffer
    VBX_STAMP();
    if (tid < TL * SP) {
        const int s = tid / SP, j = tid % SP;
        if (pick(t_len, s) > 0)
            bt.epart[(long long)(quad * TL + s) * SP + j] = j < s_v ? ent_w[2 * s][j] + ent_w[2 * s + 1][j] : 0.0;
    }
    if (tid >= 512 - TL) {                                    // this tile's share of the total log-likelihood (VBx.py:173)
        const int s = tid - (512 - TL), ln = pick(t_len, s);
        if (ln > 0)
            bt.tllpart[quad * TL + s] = (mred[2 * s] + mred[2 * s + 1]) + (log((double)sfl[s][ln - 1]) - log((double)sgin[s]) +
                                                                          (double)efl[s][(ln - 1) >> 2] * 0.69314718055994530942);
    }
    // wave (ps, sub) finishes slab sub and hands the other one to its partner
    constexpr int XW = NT * 4 * 4 * 64;                       // elements a wave hands over
    static_assert(8 * XW <= TL * TS && 8 * SP <= TL * HS, "exchange buffers");
    R* const xch = bl;
    R* const nsx = afh;
    {
        R* dst = xch + wave * XW + lane;
#pragma unroll
        for (int mu = 0; mu < NT; ++mu)
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[((mu * 4 + h) * 4 + r) * 64] = sub ? acc[0][mu][h][r] : acc[1][mu][h][r];
#pragma unroll
        for (int mu = 0; mu < NT; ++mu) {
            R v = nsum[mu];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g4 == 0) nsx[wave * SP + NT * i16 + mu] = v;
        }
    }
    __syncthreads();
    if (p_len > 0) {
        const R* src = xch + (wave ^ 1) * XW + lane;
        R* __restrict__ part = bt.mpart + (long long)(quad * TL + ps) * SP * Dp;
        if (64 * sub + 4 * i16 < Dp) {
#pragma unroll
            for (int mu = 0; mu < NT; ++mu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    R4 o;
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const R mine = sub ? acc[1][mu][h][r] : acc[0][mu][h][r];
                        const R other = src[((mu * 4 + h) * 4 + r) * 64];
                        o[h] = sub ? other + mine : mine + other;       // (even k-steps + odd k-steps, whichever wave adds them)
                    }
                    const int s = NT * M::row(lane, r) + mu;
                    *reinterpret_cast<R4*>(part + (long long)s * Dp + 64 * sub + 4 * i16) = o;
                }
            }
        }
        if (sub == 0 && lane < SP) bt.npart[(long long)(quad * TL + ps) * SP + lane] = nsx[wave * SP + lane] + nsx[(wave + 1) * SP + lane];
    }
    VBX_STAMP();
#ifdef VBX_PHASE_CLOCKS
    if (lane == 0 && wave < 2 && bt.state[p_rec].n_iters == 3 && blockIdx.x < kClockTiles) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        long long* dst = g_phase_clocks + ((long long)blockIdx.x * 2 + wave) * 8;
        for (int k = 0; k < 7; ++k) dst[k] = clk[k];
        dst[7] = wall1 - wall0;      // 100 MHz ticks across the re-run: clk[3] - clk[1] over this = shader clocks per 10 ns
        (void)hw;
    }
#endif
}

}  // namespace vbx
