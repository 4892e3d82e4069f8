// vbx_fused_mid.hpp -- chunk_post with a meet-in-the-middle lattice layout.
//
// Same work as chunk_post_kernel (vbx_fused.hpp): re-run of a chunk from its boundary vectors, posteriors,
// "entered" statistic, log-likelihood share and the next M-step accumulation.  chunk_post_kernel keeps three
// lattices of kTileFrames x SP in LDS (b, forward, backward) = 48 KB at SP = 32 -> three workgroups per CU,
// and the kernel is bound by (latency of the re-run) / (workgroups in flight).  Here each direction stores
// only the half of its lattice that the OTHER direction has not produced yet, and everything after the
// midpoint overwrites rows of b that both directions have already consumed:
//
//     rows [0, mid)   : a_f -> afh[f]   (forward, before the midpoint barrier)
//                       x_f -> bl[f]    (backward, after the barrier; b_f has been consumed by both)
//     rows [mid, len) : x_f -> bfh[f-mid] (backward, before the barrier)
//                       a_f -> bl[f]    (forward, after the barrier)
//     finally         : gamma_f -> bl[f]
//
// 16 + 8 + 8 KB at SP = 32 -> five workgroups per CU by LDS.  The rho fragments of the accumulation are no
// longer held in registers across the re-run (64 registers); they are fetched in quarters, one quarter ahead.
// Instrumentation builds: -DVBX_PHASE_CLOCKS (per-workgroup phase stamps, tools/phase_timeline.py) and
// -DVBX_EXPERIMENT_SKIP_RERUN (elimination run of DESIGN section 10: the kernel without its re-run loops; the
// results are garbage, only the timing means something).
#pragma once
#include "vbx_fused.hpp"

namespace vbx {

#ifdef VBX_PHASE_CLOCKS
constexpr int kClockTiles = 8192;
__device__ long long g_phase_clocks[kClockTiles * 16];
#endif

template <typename R, int SP> struct ChunkPostMidCfg {
    static constexpr int kBytes = 2 * kTileFrames * SP * (int)sizeof(R) + 6144;
    static constexpr bool kFits = kBytes <= 160 * 1024;
    static constexpr int kPerCU = kBytes <= 32 * 1024 ? 5 : kBytes <= 40 * 1024 ? 4 : kBytes <= 53 * 1024 ? 3
                                  : kBytes <= 80 * 1024 ? 2 : 1;
};

template <typename R, int SP>
__global__ __launch_bounds__(256, (ChunkPostMidCfg<R, SP>::kPerCU)) void chunk_post_mid_kernel(BatchView<R> bt) {
    using M = Mfma16<R>;
    using acc_t = typename M::acc_t;
    using R2 = typename Vec<R>::v2;
    using R4 = typename Vec<R>::v4;
    constexpr int NREG = SP / 16;                      // states per lane in the re-run
    constexpr int NT = SP / 16;                        // M-tiles (speakers) of the accumulation
    constexpr int HALF = kTileFrames / 2;
    constexpr int KS = kTileFrames / 4;                // MFMA k-steps per chunk
    __shared__ __attribute__((aligned(16))) R bl[kTileFrames * SP];    // b, then a (rows >= mid) / x (rows < mid), then gamma
    __shared__ __attribute__((aligned(16))) R afh[HALF * SP];          // a_f,  f < mid
    __shared__ __attribute__((aligned(16))) R bfh[HALF * SP];          // x_f,  f >= mid  (row f - mid)
    __shared__ R sfl[kTileFrames];                     // s_f = sum(a_f) of the stored forward row
    __shared__ R qfl[kTileFrames];                     // q_f: every element of the stored backward row is >= q_f > 0
    __shared__ R tl_sig[2];
    __shared__ int tl_expo;
    __shared__ __attribute__((aligned(16))) R c_l[SP];
    __shared__ __attribute__((aligned(16))) R aprev0[SP];
    __shared__ double ent_w[4][SP];
    __shared__ double red[16];

    const int tile = blockIdx.x;
    // one scalar load gives every address of the first round of vector loads (tile table of vbx_capi.hip)
    const int4 td = bt.tile_desc[tile];
    if (bt.tile_done[tile]) return;
    VBX_CLOCKS_DECL();
    VBX_STAMP();
    const int rec = td.x, t0 = td.y, len = td.z;
    const long long trow = td.w;                       // first frame row of the tile
    const int Dp = bt.Dp;
    const int mid = len / 2;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i16 = lane & 15, g4 = lane >> 4;
    const int so = i16 * NREG;                         // first state of this lane
    const double lp_d = bt.recs[rec].lp;               // (scalar loads: in flight next to the b tile)
    const int n_spk = bt.recs[rec].S;
    const R lp = (R)lp_d;
    const R* __restrict__ rho = bt.rho + (trow - t0) * Dp;

    stage_to_lds<(kTileFrames * SP / 4 + 255) / 256>(reinterpret_cast<R4*>(bl),
                                                     reinterpret_cast<const R4*>(bt.bmat + trow * SP),
                                                     len * SP / 4, tid, 256);
    if (tid < SP)
        c_l[tid] = (tid < n_spk) ? (R)((1.0 - lp_d) * bt.pi[(long long)rec * SP + tid] + 1e-8) : (R)0;
    const bool chunk0 = (t0 == 0);
    __syncthreads();
    VBX_STAMP();

    // ---- re-run: wave 0 forward, wave 1 backward; same unnormalised recursions as chunk_post_kernel -----------
    R c[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) c[r] = c_l[so + r];
    auto load_rows = [&](R (&dst)[4][NREG], int f, int dir) {
#pragma unroll
        for (int k = 0; k < 4; ++k) load_pack<NREG>(dst[k], bl + (f + dir * k) * SP + so);
    };
    // forward state
    R a[NREG], sig = 1, sig_in = 1;
    int expo = 0, ff = 0;
    auto f_renorm = [&]() {
        const int e = rescale_exponent(sig);
        expo += e;
        sig = scale2(sig, -e);
#pragma unroll
        for (int r = 0; r < NREG; ++r) a[r] = scale2(a[r], -e);
    };
    auto f_store = [&](int f) { store_pack<NREG>((f < mid ? afh + f * SP : bl + f * SP) + so, a); sfl[f] = sig; };
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
        R cur[4][NREG], nxt[4][NREG];
        if (ff + 4 <= end) load_rows(cur, ff, 1);
        for (; ff + 4 <= end; ff += 4) {
            if (ff + 8 <= end) load_rows(nxt, ff + 4, 1);
            f_renorm();
#pragma unroll
            for (int k = 0; k < 4; ++k) f_step(cur[k], ff + k);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < NREG; ++r) cur[k][r] = nxt[k][r];
        }
        f_renorm();
        for (; ff < end; ++ff) {
            R b[NREG];
            load_pack<NREG>(b, bl + ff * SP + so);
            f_step(b, ff);
        }
    };
    // backward state: x = x_{fb} (unnormalised), produced by consuming rows > fb
    R x[NREG], q = 1;
    int fb = len - 1;
    auto b_store = [&](int f) { store_pack<NREG>((f < mid ? bl + f * SP : bfh + (f - mid) * SP) + so, x); qfl[f] = q; };
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
        q = scale2(q, -e);                                   // (x_{mid-1} and its q wait in registers for the barrier)
#pragma unroll
        for (int r = 0; r < NREG; ++r) x[r] = scale2(x[r], -e);
    };

    if (wave == 0) {
        const R* __restrict__ bnd = bt.fbound + (long long)tile * SP + so;
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            a[r] = bnd[r];
            if (!chunk0) aprev0[so + r] = a[r];              // a[t0-1] (any scale) for the statistics of frame t0
            if (chunk0) a[r] *= bl[so + r];                  // frame 0: a_0 = b_0 (ip + 1e-8), VBx.py:163
        }
        sig = a[0];
#pragma unroll
        for (int r = 1; r < NREG; ++r) sig += a[r];
        sig = allreduce_sum<16>(sig);
        sig_in = sig;
        if (chunk0) {
            f_store(0);                                      // (row 0 of bl if len == 1: b_0 is not needed again)
            ff = 1;
        }
#ifndef VBX_EXPERIMENT_SKIP_RERUN
        f_run(max(mid, ff));                                 // rows < mid -> afh
#endif
    } else if (wave == 1) {
        const R* __restrict__ bnd = bt.gbound + (long long)tile * SP + so;
        R part = 0;
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            x[r] = bnd[r];
            part += x[r];
        }
        part = allreduce_sum<16>(part);
        const int e = rescale_exponent(part);
#pragma unroll
        for (int r = 0; r < NREG; ++r) x[r] = scale2(x[r], -e);
        q = scale2(part, -e) * (R)(1.0 / SP);                // a positive scale of the row, like q of the steps
        b_store(len - 1);                                    // -> bfh (len-1 >= mid always)
        // consume rows len-1 .. max(mid, 1); the outputs with index >= mid go to bfh, the last one (x_{mid-1})
        // stays in registers until the barrier: its slot in bl still holds b_{mid-1}, which the forward wave
        // may not have consumed yet
        const int stop = max(mid, 1);
        R cur[4][NREG], nxt[4][NREG];
        if (fb - 3 >= stop) load_rows(cur, fb, -1);
#ifdef VBX_EXPERIMENT_SKIP_RERUN
        fb = 0;
#endif
        while (fb - 3 >= stop) {                             // a block of four rows fb .. fb-3, all >= stop
            const bool last_block = fb - 4 < stop;           // its last output is x_{stop-1}
            if (fb - 7 >= stop) load_rows(nxt, fb - 4, -1);
#pragma unroll
            for (int k = 0; k < 4; ++k) b_step(cur[k], !(last_block && k == 3) || stop - 1 >= mid);
            b_renorm();
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < NREG; ++r) cur[k][r] = nxt[k][r];
        }
        while (fb >= stop) {
            R b[NREG];
            load_pack<NREG>(b, bl + fb * SP + so);
            b_step(b, fb - 1 >= mid);
        }
    }
    VBX_STAMP();
    __syncthreads();                                         // midpoint: rows >= mid of b are consumed by the backward
                                                             // wave, rows < mid by the forward wave
    if (wave == 0) {
#ifndef VBX_EXPERIMENT_SKIP_RERUN
        f_run(len);                                          // rows >= mid -> bl (over b_f, after reading it)
#endif
        if (lane == 0) {
            tl_sig[0] = sig;
            tl_sig[1] = chunk0 ? (R)1 : sig_in;
            tl_expo = expo;
        }
    } else if (wave == 1 && mid >= 1) {
        // x = x_{mid-1} is in registers, rows mid-1 .. 1 remain.  Every output x_{f-1} lands on b_{f-1}, the row
        // the NEXT step consumes, so rows are always in registers before their slot is written: up to three
        // leading single rows and the first block of four are fetched before the first store.
        const int n = fb, tail = n & 3;                      // fb == mid - 1
        R lead[3][NREG], cur[4][NREG], nxt[4][NREG];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < tail) load_pack<NREG>(lead[k], bl + (fb - k) * SP + so);
        if (n - tail >= 4) load_rows(cur, fb - tail, -1);
        b_store(fb);                                         // x_{mid-1} -> bl[mid-1]
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < tail) b_step(lead[k], true);
        if (tail) b_renorm();
        while (fb >= 4) {                                    // blocks of four rows fb .. fb-3 (fb is a multiple of 4 here)
            if (fb - 4 >= 4) load_rows(nxt, fb - 4, -1);
#pragma unroll
            for (int k = 0; k < 4; ++k) b_step(cur[k], true);
            b_renorm();
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < NREG; ++r) cur[k][r] = nxt[k][r];
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
    if (wave * 32 < Dp) load_quarter(bq[0], wave, 0);
    VBX_STAMP();
    __syncthreads();
    VBX_STAMP();

    // ---- posteriors and the "entered" statistic                               (VBx.py:101-103,174) --
    // pass 1 reads (a, x, a of the previous frame) into registers, pass 2 writes gamma over bl: a row of bl may
    // hold the a or x another frame's pass 1 still needs.
    {
        constexpr int NIT = kTileFrames / 16;
        R* __restrict__ G = bt.gamma + trow * SP;
        R gam[NIT][NREG], ent[NREG];
#pragma unroll
        for (int r = 0; r < NREG; ++r) ent[r] = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int f = 16 * it + 4 * wave + g4;
            const bool ok = f < len;
            const int fr = ok ? f : 0;
            const R isig = fast_rcp(sfl[fr]), iq = fast_rcp(qfl[fr]);   // (applied one after the other: their product may overflow)
            const R sp = fr > 0 ? sfl[fr - 1] : tl_sig[1];
            R av[NREG], xv[NREG], ap[NREG];
            load_pack<NREG>(av, (fr < mid ? afh + fr * SP : bl + fr * SP) + so);
            load_pack<NREG>(xv, (fr < mid ? bl + fr * SP : bfh + (fr - mid) * SP) + so);
            load_pack<NREG>(ap, (fr == 0 ? aprev0 : fr - 1 < mid ? afh + (fr - 1) * SP : bl + (fr - 1) * SP) + so);
#pragma unroll
            for (int r = 0; r < NREG; ++r) gam[it][r] = (av[r] * isig) * (xv[r] * iq);
            R sum = gam[it][0];
#pragma unroll
            for (int r = 1; r < NREG; ++r) sum += gam[it][r];
            sum = allreduce_sum<16>(sum);
            const R inv = ok ? fast_rcp(sum) : (R)0;
            const bool stat = ok && t0 + f >= 1;               // frame 0 of the recording has no "entered" term
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                gam[it][r] *= inv;
                const R term = gam[it][r] * sp * fast_rcp(lp * ap[r] + c[r] * sp);
                ent[r] += stat ? term : (R)0;                  // (select, not multiply: ap is undefined for frame 0)
            }
        }
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            double e = (double)ent[r];                     // <= 8 terms per lane in working precision
            e += __shfl_xor(e, 16, 64);
            e += __shfl_xor(e, 32, 64);
            if (g4 == 0) ent_w[wave][so + r] = e;
        }
        double mpartial = 0.0;                             // this chunk's share of the total log-likelihood (VBx.py:173)
        if (tid < len) mpartial = (double)bt.mrow[trow + tid];
        if (tid == 128)
            mpartial += log((double)tl_sig[0]) - log((double)tl_sig[1]) + (double)tl_expo * 0.69314718055994530942;
        mpartial = block_sum(mpartial, red);               // (its barriers also end pass 1)
        if (tid < SP) {
            const double e = (ent_w[0][tid] + ent_w[1][tid]) + (ent_w[2][tid] + ent_w[3][tid]);
            bt.epart[(long long)tile * SP + tid] = tid < n_spk ? e : 0.0;
        }
        if (tid == 0) bt.tllpart[tile] = mpartial;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int f = 16 * it + 4 * wave + g4;
            store_pack<NREG>(bl + f * SP + so, gam[it]);       // A operand of the accumulation below (0 past the end)
            if (f < len) store_pack<NREG>(G + f * SP + so, gam[it]);
        }
    }
    __syncthreads();
    VBX_STAMP();

    // ---- next M-step: C[s][d] = sum_t gamma[t][s] rho[t][d] on MFMA 16x16x4        (VBx.py:96) --
    // M index i of tile mu <-> speaker NT*i + mu; N index j of half h <-> feature 32*slab + 2j + h.
    for (int slab = wave; slab * 32 < Dp; slab += 4) {
        acc_t acc[NT][2];
        R nsum[NT];
#pragma unroll
        for (int mu = 0; mu < NT; ++mu) {
            acc[mu][0] = acc_t{0, 0, 0, 0};
            acc[mu][1] = acc_t{0, 0, 0, 0};
            nsum[mu] = 0;
        }
        if (slab != wave) load_quarter(bq[0], slab, 0);
        auto quarter = [&](const R2 (&bfr)[QK], int qi) {
#pragma unroll
            for (int u = 0; u < QK; ++u) {
                const int f = 4 * (qi * QK + u) + g4;
                R av[NT];
                load_pack<NT>(av, bl + f * SP + NT * i16);
#pragma unroll
                for (int mu = 0; mu < NT; ++mu) {
                    nsum[mu] += av[mu];
                    acc[mu][0] = M::mma(av[mu], bfr[u].x, acc[mu][0]);
                    acc[mu][1] = M::mma(av[mu], bfr[u].y, acc[mu][1]);
                }
            }
        };
#pragma unroll 1
        for (int pair = 0; pair < 2; ++pair) {               // (not unrolled: bounds how many LDS reads are hoisted)
            load_quarter(bq[1], slab, 2 * pair + 1);         // next quarter in flight
            quarter(bq[0], 2 * pair);
            if (pair == 0) load_quarter(bq[0], slab, 2);
            quarter(bq[1], 2 * pair + 1);
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
    // plain stores into a module-scope table (printf would fence the whole XCD): read with vbx_debug_clocks()
    if (lane == 0 && (wave == 0 || wave == 2) && bt.state[rec].n_iters == 3 && blockIdx.x < kClockTiles) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        long long* dst = g_phase_clocks + ((long long)blockIdx.x * 2 + (wave >> 1)) * 8;
        for (int k = 0; k < 7; ++k) dst[k] = clk[k];
        dst[7] = hw;
    }
#endif
}

}  // namespace vbx
