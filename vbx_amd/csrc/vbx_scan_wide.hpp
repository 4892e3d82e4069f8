// vbx_scan_wide.hpp -- the exact chunked scan of vbx_scan.hpp for 64 < S <= 256 states.
//
// Same three steps (chunk operators, boundary walk, re-run from the boundaries; VBx.py:146-175 in a different
// association order), other shapes: an S x S operator is 64-256 KB, so
//   scan1_wide  one workgroup builds 16 columns of a chunk's operator (16 lanes per column, S/16 states per lane);
//               b of the chunk goes through LDS in pieces of at most 32 KB
//   scan2_wide  one workgroup of 4-8 wavefronts per (recording, direction) walks the chain; every mat-vec reads the
//               operator straight from L2 / HBM with 16-byte loads requested ahead of its step; wave w owns a block of
//               rows of it, partial sums meet in LDS (forward) / in a butterfly (backward)
//   scan3_wide  one wavefront per (chunk, direction) re-runs the chunk with lane = state (S/64 states per lane),
//               rows of b prefetched from L2 eight frames ahead, and writes ahat / bhat / the forward scales exactly
//               like scan3 -- post_kernel and the accumulation take it from there.
// A recording of T = 10 000 with S = 200 takes ~0.3 ms per iteration this way; the sequential walk it replaces
// (fb_seq_kernel: one wavefront over all T frames) 2 ms.
#pragma once
#include "vbx_scan.hpp"
#include "vbx_operator.hpp"

namespace vbx {

template <typename R, int SP> struct ScanWideCfg {
    static constexpr int PH = 16;                                      // lanes per operator column
    static constexpr int NR = SP / PH;                                 // states per lane
    static constexpr int CB = 16;                                      // columns per workgroup
    static constexpr int kPieceBytes = 32 * 1024;                      // b of a chunk goes through LDS in pieces of this size
    static constexpr int kPiece = kPieceBytes / (SP * (int)sizeof(R)) < kTileFrames ? kPieceBytes / (SP * (int)sizeof(R)) : kTileFrames;
};

template <typename R, int SP>
__global__ __launch_bounds__(256) void scan1_wide_kernel(BatchView<R> bt) {
    using Cfg = ScanWideCfg<R, SP>;
    using R4 = typename Vec<R>::v4;
    constexpr int PH = Cfg::PH, NR = Cfg::NR, CB = Cfg::CB, PIECE = Cfg::kPiece;
    constexpr int NV = (PIECE * SP / 4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) R bp[PIECE * SP];
    const int tile = blockIdx.x;
    const int rec = bt.tile_rec[tile];
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int t0 = bt.tile_t0[tile];
    const int len = min(kTileFrames, rd.T - t0);
    const int tid = threadIdx.x;
    const int col = blockIdx.y * CB + tid / PH, j0 = (tid % PH) * NR;
    const R lp = (R)rd.lp;
    R x[NR], c[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = j0 + r;
        x[r] = (j == col) ? (R)1 : (R)0;
        c[r] = (j < rd.S) ? (R)((1.0 - rd.lp) * bt.pi[(long long)rec * SP + j] + 1e-8) : (R)0;
    }
    auto colsum = [&]() {
        R v = x[0];
#pragma unroll
        for (int r = 1; r < NR; ++r) v += x[r];
        return column_sum<PH>(v);
    };
    int expo = 0;
    for (int p0 = 0; p0 < len; p0 += PIECE) {
        const int n = min(PIECE, len - p0);
        __syncthreads();
        stage_to_lds<NV>(reinterpret_cast<R4*>(bp), reinterpret_cast<const R4*>(bt.bmat + (rd.row0 + t0 + p0) * SP), n * SP / 4, tid, 256);
        __syncthreads();
        for (int s = 0; s < n; ++s) {
            const R* row = bp + s * SP + j0;
            if (t0 + p0 + s == 0) {          // frame 0 of the recording: x <- b_0 * x (VBx.py:163, no transition)
#pragma unroll
                for (int r = 0; r < NR; ++r) x[r] *= row[r];
                continue;
            }
            const R sig = colsum();
            const int e = rescale_exponent(sig);
            expo += e;
            const R sc = scale2((R)1, -e);
            const R lps = lp * sc, sgs = sig * sc;
#pragma unroll
            for (int q = 0; q < NR / 4; ++q) {
                const R4 b4 = *reinterpret_cast<const R4*>(row + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) x[4 * q + r] = b4[r] * (lps * x[4 * q + r] + c[4 * q + r] * sgs);
            }
        }
    }
    {   // final power-of-two normalisation: column sums end in [0.5, 1)
        const R sig = colsum();
        const int e = rescale_exponent(sig);
        expo += e;
#pragma unroll
        for (int r = 0; r < NR; ++r) x[r] = scale2(x[r], -e);
        if (!(sig > (R)0)) expo = -(1 << 24);     // an all-zero column must never win the exponent maximum of the walk
    }
    R* __restrict__ dst = bt.op + ((long long)tile * SP + col) * SP + j0;
#pragma unroll
    for (int r = 0; r < NR; ++r) dst[r] = x[r];
    if ((tid % PH) == 0) bt.opexp[(long long)tile * SP + col] = expo;
}

// scan1_wide with the recursion of the fused path (round 6): b of the WHOLE chunk in LDS (64 KB at S = 128 in f32: it fits since
// gfx950), 64 columns per workgroup of 1024 threads, and the column recursion of vbx_operator.hpp -- the form scaled by lp^t
// (one FMA and one product per state), written on pairs of states (v_pk_*_f32), rescaled every fourth frame -- instead of
// three operations per state, scalar, and a rescale per frame: 48 -> about 22 vector instructions per wave and frame.  The c
// of the scaled form is computed here from the priors exactly as fin_kernel computes it for chunk_loglik (f64, rounded once),
// so the kernel also serves forward_backward() calls that never run an M-step.  S > 128 keeps scan1_wide_kernel.
template <typename R, int SP> struct Scan1WideLdsCfg {
    static constexpr int COLS = 64;                                // columns per workgroup (16 lanes each)
    static constexpr int kBytes = (kTileFrames * SP + SP) * (int)sizeof(R);
    // (S = 256: sixteen states per lane spill at the 128 registers of a 1024-thread block; fp64 has no packed instructions to
    //  gain from and pays for one workgroup per CU: S = 128, T = 10 000 / 50 000 0.247 / 0.704 -> 0.251 / 0.736 ms per iteration;
    //  f32: 0.161 / 0.381 -> 0.146 / 0.327, eight recordings of S = 100: 0.405 -> 0.311)
    static constexpr bool kFits = kBytes <= 144 * 1024 && SP == 128 && sizeof(R) == 4;
};

template <typename R, int SP>
__global__ __launch_bounds__(1024) void scan1_wide_lds_kernel(BatchView<R> bt) {
    using R4 = typename Vec<R>::v4;
    constexpr int PH = 16, NR = SP / PH;
    constexpr int NV = (kTileFrames * SP / 4 + 1023) / 1024;
    __shared__ __attribute__((aligned(16))) R btile[kTileFrames * SP];
    __shared__ __attribute__((aligned(16))) R cl[SP];
    const int tile = blockIdx.x;
    const int rec = bt.tile_rec[tile];
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int t0 = bt.tile_t0[tile];
    const int len = min(kTileFrames, rd.T - t0);
    const int tid = threadIdx.x;
    stage_to_lds<NV>(reinterpret_cast<R4*>(btile), reinterpret_cast<const R4*>(bt.bmat + (rd.row0 + t0) * SP), len * SP / 4, tid, 1024);
    if (tid < SP) {
        const double cj = tid < rd.S ? (1.0 - rd.lp) * bt.pi[(long long)rec * SP + tid] + 1e-8 : 0.0;
        cl[tid] = (R)(rd.lp >= 0x1p-20 ? cj / rd.lp : cj);
    }
    __syncthreads();
    const int col = blockIdx.y * Scan1WideLdsCfg<R, SP>::COLS + tid / PH, part = tid % PH;
    R x[1][NR];
    int expo[1];
    operator_columns<R, SP, PH, 1>(btile, 0, len, t0 == 0, col, part, rd.lp, cl, bt.lppow + (long long)rec * (kTileFrames + 1), x, expo);
    R* __restrict__ dst = bt.op + ((long long)tile * SP + col) * SP + part * NR;
#pragma unroll
    for (int q = 0; q < NR / 4; ++q) *reinterpret_cast<R4*>(dst + 4 * q) = R4{x[0][4 * q], x[0][4 * q + 1], x[0][4 * q + 2], x[0][4 * q + 3]};
    if (part == 0) bt.opexp[(long long)tile * SP + col] = expo[0];
}

// Two values (a, b) in every lane -> ONE per lane, summed over the lane pair (l, l ^ STAGE), STAGE = 16 or 32: lanes with
// the STAGE bit clear get a(l) + a(l ^ STAGE), lanes with it set get b(l) + b(l ^ STAGE).  One v_permlane*_swap moves both
// operands (it exchanges the upper lanes of its first register with the lower lanes of its second), so the cross-row stages
// of a reduction of N values over a wavefront cost N / 2 + N / 4 swaps instead of 2 N.
template <int STAGE> __device__ __forceinline__ float pair_sum_xor(float a, float b) {
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    const auto r = STAGE == 16 ? __builtin_amdgcn_permlane16_swap(ua, ub, false, false) : __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
template <int STAGE> __device__ __forceinline__ double pair_sum_xor(double a, double b) {
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    const auto lo = STAGE == 16 ? __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false)
                                : __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
    const auto hi = STAGE == 16 ? __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false)
                                : __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    const double x = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi[0] << 32) | (unsigned)lo[0]);
    const double y = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi[1] << 32) | (unsigned)lo[1]);
    return x + y;
}

// block-wide maximum of one int per thread of the first SP threads (SP a multiple of 64); every thread gets it
// (an LDS barrier: the operator loads of later chain steps stay in flight across it)
template <int SP> __device__ __forceinline__ int wide_block_max(int v, int* wmax, int tid) {
    if (tid < SP) {                                    // (wave-uniform: SP is a multiple of 64)
        const int m = allreduce_max<64>(v);
        if ((tid & 63) == 0) wmax[tid >> 6] = m;
    }
    lds_barrier();
    int top = wmax[0];
#pragma unroll
    for (int w = 1; w < SP / 64; ++w) top = max(top, wmax[w]);
    return top;
}

// The boundary walk of one recording in one direction: K - 1 dependent chain steps, each a mat-vec with a chunk operator
// that lives in HBM (or another XCD's L2), by ONE workgroup of NW wavefronts.  A step is a chain of short phases that
// meet at four barriers -- largest exponent, weights, partial products, their sum -- plus the operator's way from memory:
//  * the operator is read with 16-byte loads, 64 * VEC consecutive elements per wave instruction, requested D chain
//    steps ahead (where a thread's share fits the registers D + 1 times), and the barriers of a step are LDS barriers:
//    __syncthreads() waits for every load in flight.  Round 3 read 4 / 8 bytes per lane one step ahead behind
//    __syncthreads(): 272 wave loads per operator at S = 128, 1.55 us per step in fp32 and 2.95 in fp64 -- a recording
//    of 79 chunks 125 / 240 us per launch, half of its iteration.  Now 0.95 / 1.6 us per step (74 / 128 us), S = 200 in
//    fp64 (whose walk spilled 21 registers) 1.49 -> 0.92 ms per iteration;
//  * NW = 16: with 8 wavefronts the fp32 walk at S = 128 takes 92 us, with 4 it takes 117 (tools/cu_stream_probe.hip: one
//    CU pulls 108 GB/s with 16 wavefronts loading, 87 with 4 at sixteen loads in flight each, 45 at four).
// Layout: wave w owns rows [w RPW, (w + 1) RPW) of the operator (row = the index the mat-vec sums over in the forward
// direction, the output index in the backward one); a wave instruction covers RPI whole rows (lanes l, l + LPR, ... hold
// the same columns) or, when a row is longer than 64 * VEC elements, one NG-th of a row.
template <typename R, int SP, int NW> struct WalkWideCfg {
    static constexpr int VEC = 16 / (int)sizeof(R);            // elements per lane and load
    static constexpr int EPI = 64 * VEC;                       // elements per wave instruction
    static constexpr bool kRows = EPI >= SP;
    static constexpr int RPI = kRows ? EPI / SP : 1;           // rows per instruction
    static constexpr int NG = kRows ? 1 : SP / EPI;            // instructions per row
    static constexpr int LPR = kRows ? SP / VEC : 64;          // lanes per row
    static constexpr int RPW = SP / NW;                        // rows per wave
    static constexpr int IPW = RPW * NG / RPI;                 // instructions per wave and operator
    static constexpr int kDwords = IPW * 4;                    // registers one operator share takes
    static constexpr int D = kDwords <= 16 ? 3 : (kDwords <= 32 ? 1 : 0);     // chain steps the loads run ahead (two at 32 registers spill)
    static constexpr int CHI = D > 0 ? IPW : 8;                // instructions in registers at a time when nothing runs ahead
    static_assert(RPI == 1 || RPI == 2, "rows per instruction");
    static_assert(NW * 64 >= SP, "the first SP threads carry the vector");
};

// `level` as in scan2_kernel (vbx_scan.hpp): 0 the flat chain over a recording's chunk operators (grid.x = recording), 2 over
// its GROUP operators (compose_wide_kernel below; boundaries at the group edges), 3 inside one group from the boundary level 2
// left at its edge (grid.x = group).  Chain step n uses operator op0 + n os of `ops` and writes boundary b0 + (n + 1) bs.
template <typename R, int SP, int NW>
__global__ __launch_bounds__(NW * 64) void scan2_wide_kernel(BatchView<R> bt, int level) {
    using Cfg = WalkWideCfg<R, SP, NW>;
    constexpr int VEC = Cfg::VEC, EPI = Cfg::EPI, RPI = Cfg::RPI, NG = Cfg::NG, LPR = Cfg::LPR, RPW = Cfg::RPW, IPW = Cfg::IPW;
    constexpr int D = Cfg::D, CHI = Cfg::CHI;
    constexpr bool kRows = Cfg::kRows;
    typedef R RV __attribute__((ext_vector_type(VEC)));
    __shared__ __attribute__((aligned(16))) R vec[SP];         // the vector being pushed through the chain (weights, forward)
    __shared__ __attribute__((aligned(16))) R part[NW * SP];   // forward: per wave and column; backward: per row
    __shared__ int wmax[SP / 64];
    const int rec = level == 3 ? bt.sup_rec[blockIdx.x] : blockIdx.x, dir = blockIdx.y;
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int K = rd.ntiles, G = bt.sgroup;
    const long long cb0 = rd.tile0;
    const R* __restrict__ ops = bt.op;
    const int* __restrict__ oexp = bt.opexp;
    int nops = K - 1, os = dir == 0 ? 1 : -1, bs = os;
    long long op0 = dir == 0 ? cb0 : cb0 + K - 1, b0 = op0, binit = op0;
    if (level == 2) {
        const int ns = (K + G - 1) / G;
        ops = bt.sop;
        oexp = bt.sopexp;
        nops = ns - 1;
        op0 = dir == 0 ? rd.sup0 : rd.sup0 + ns - 1;
        bs = dir == 0 ? G : -G;
        b0 = dir == 0 ? cb0 : cb0 + (long long)ns * G - 1;
    } else if (level == 3) {
        const long long a = cb0 + (long long)bt.sup_idx[blockIdx.x] * G, b = min(a + G, cb0 + K);
        nops = (int)(b - 1 - a);
        op0 = b0 = binit = dir == 0 ? a : b - 1;
        if (nops <= 0) return;
    }
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int j = tid % SP;
    constexpr int kNone = -(1 << 28);
    const int sub = kRows ? lane / LPR : 0;                    // which of the RPI rows of an instruction this lane reads
    const int col0 = kRows ? (lane % LPR) * VEC : lane * VEC;  // its first column (within group g: + g * EPI)
    // chain step n uses operator op0 + n os; n is clamped so that the requests running ahead of the last step stay inside the
    // chain (nops >= 1 wherever a step runs)
    auto fetch = [&](int n, RV (&dst)[CHI], int& e, int i0) {
        const int nn = max(min(n, nops - 1), 0);
        const long long k = op0 + (long long)nn * os;
        const RV* __restrict__ src = reinterpret_cast<const RV*>(ops + k * SP * SP) + (long long)(wave * IPW + i0) * 64 + lane;
#pragma unroll
        for (int q = 0; q < CHI; ++q) dst[q] = src[q * 64];
        if (tid < SP) e = oexp[k * SP + j];
    };
    RV ovr[D + 1][CHI];
    int ejr[D + 1];
#pragma unroll
    for (int d = 0; d <= D; ++d) ejr[d] = 0;

    if (dir == 0) {
        R y = 0;
        if (tid < SP) {
            if (level == 3) {
                y = bt.fbound[binit * SP + j];
            } else {
                y = (j < rd.S) ? (R)(bt.ip[(long long)rec * SP + j] + 1e-8) : (R)0;
                bt.fbound[binit * SP + j] = y;
            }
        }
        auto step = [&](int n, RV (&ov)[CHI], int ej) {
            if constexpr (D == 0) fetch(n, ov, ej, 0);
            // weights y_i 2^{E_i}, shifted by the largest on the support of y
            int tj = kNone;
            if (tid < SP && y > (R)0) tj = ej + exponent_of(y);
            const int top = wide_block_max<SP>(tj, wmax, tid);
            if (tid < SP) vec[j] = (y > (R)0) ? scale2(y, ej - top) : (R)0;
            lds_barrier();
            R acc[NG][VEC];
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[g][k] = 0;
#pragma unroll
            for (int i0 = 0; i0 < IPW; i0 += CHI) {
                if (i0 > 0) fetch(n, ov, ej, i0);
#pragma unroll
                for (int q = 0; q < CHI; ++q) {
                    const int row = kRows ? (wave * IPW + i0 + q) * RPI + sub : wave * RPW + (i0 + q) / NG;
                    const R w = lone_register(vec[row]);       // (not the odd half of a loaded pair: vbx_device.hpp)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[(i0 + q) % NG][k] += w * ov[q][k];
                }
            }
            if (RPI == 2) {                                    // lanes l and l + 32 hold the same columns
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[0][k] = add_xor<32>(acc[0][k]);
            }
            if (lane < LPR) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    RV v;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) v[k] = acc[g][k];
                    *reinterpret_cast<RV*>(part + wave * SP + g * EPI + col0) = v;
                }
            }
            lds_barrier();
            if (tid < SP) {
                R tot = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) tot += part[w * SP + j];
                y = (j < rd.S) ? tot : (R)0;                   // padded speakers carry no mass
                bt.fbound[(b0 + (long long)(n + 1) * bs) * SP + j] = y;
            }
            lds_barrier();
        };
        if constexpr (D > 0) {
            if (nops > 0) {
#pragma unroll
                for (int d = 0; d < D; ++d) fetch(d, ovr[d], ejr[d], 0);
            }
            for (int n = 0; n < nops; n += D + 1) {
#pragma unroll
                for (int u = 0; u <= D; ++u) {
                    if (n + u < nops) {                        // (uniform)
                        fetch(n + u + D, ovr[(u + D) % (D + 1)], ejr[(u + D) % (D + 1)], 0);
                        step(n + u, ovr[u], ejr[u]);
                    }
                }
            }
        } else {
            for (int n = 0; n < nops; ++n) step(n, ovr[0], 0);
        }
    } else {
        R g = 0;
        if (tid < SP) {
            if (level == 3) {
                g = bt.gbound[binit * SP + j];
            } else {
                g = (j < rd.S) ? (R)1 : (R)0;
                bt.gbound[binit * SP + j] = g;
            }
            vec[j] = g;
        }
        lds_barrier();
        // (F^T g)_c = 2^{E_c} <row c of the stored operator, g>: the lanes of a row multiply their VEC elements with
        // their VEC elements of g and the products meet in a butterfly over the row's lanes
        auto step = [&](int n, RV (&ov)[CHI], int ecur) {
            if constexpr (D == 0) fetch(n, ov, ecur, 0);
            RV gl[NG];
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) gl[gq] = *reinterpret_cast<const RV*>(vec + gq * EPI + col0);
            // The NV row products of a chunk of instructions are reduced over the W lanes of a row TOGETHER: the cross-row
            // stages (xor 32, xor 16) halve the number of values a lane carries (pair_sum_xor), the four stages inside a
            // row of 16 lanes run on what is left -- NV = 8 over 64 lanes: 4 + 2 pair exchanges and two DPP butterflies
            // instead of eight six-stage butterflies (in fp64, where an add is half rate and a move two, those were most
            // of the step).  Lane l ends up with the totals of the values q(l) below and the first lane of its row stores them.
            constexpr int NV = kRows ? CHI : CHI / NG;             // row products per chunk of instructions
            constexpr int W = (kRows && LPR < 64) ? 32 : 64;       // lanes a row is spread over
            constexpr bool H32 = W == 64 && NV >= 2;
            constexpr int C1 = H32 ? NV / 2 : NV;
            constexpr bool H16 = C1 >= 2;
            constexpr int C2 = H16 ? C1 / 2 : C1;
            static_assert((NV & (NV - 1)) == 0, "a power of two of row products per chunk");
            const int hb = (lane >> 5) & 1, rb = (lane >> 4) & 1;
#pragma unroll
            for (int i0 = 0; i0 < IPW; i0 += CHI) {
                if (i0 > 0) fetch(n, ov, ecur, i0);
                R pr[NV];
#pragma unroll
                for (int q = 0; q < CHI; ++q) {
                    R p = 0;
#pragma unroll
                    for (int kk = 0; kk < VEC; ++kk) p += ov[q][kk] * gl[(i0 + q) % NG][kk];
                    if constexpr (kRows) pr[q] = p;
                    else pr[q / NG] = (q % NG == 0) ? p : pr[q / NG] + p;
                }
                if constexpr (W == 64) {
                    if constexpr (H32) {
#pragma unroll
                        for (int i = 0; i < NV / 2; ++i) pr[i] = pair_sum_xor<32>(pr[2 * i], pr[2 * i + 1]);
                    } else {
                        pr[0] = add_xor<32>(pr[0]);
                    }
                }
                if constexpr (H16) {
#pragma unroll
                    for (int i = 0; i < C1 / 2; ++i) pr[i] = pair_sum_xor<16>(pr[2 * i], pr[2 * i + 1]);
                } else {
                    pr[0] = add_xor<16>(pr[0]);
                }
#pragma unroll
                for (int i = 0; i < C2; ++i) pr[i] = allreduce_sum<16>(pr[i]);
                // lanes that differ in a bit whose stage did not halve hold the same totals: one of them stores
                const bool writer = (lane & 15) == 0 && (H16 || rb == 0) && (W == 32 || H32 || hb == 0);
                if (writer) {
#pragma unroll
                    for (int i = 0; i < C2; ++i) {
                        const int q16 = H16 ? 2 * i + rb : i;                      // the slot before the xor-16 stage
                        const int q = H32 ? 2 * q16 + hb : q16;                     // ... and before the xor-32 stage
                        const int row = kRows ? (wave * IPW + i0 + q) * RPI + sub : wave * RPW + i0 / NG + q;
                        part[row] = pr[i];
                    }
                }
            }
            lds_barrier();
            R tot = 0;
            int tj = kNone;
            if (tid < SP) {
                tot = part[j];
                if (tot > (R)0) tj = ecur + exponent_of(tot);
            }
            const int top = wide_block_max<SP>(tj, wmax, tid);
            if (tid < SP) {
                g = (tot > (R)0 && j < rd.S) ? scale2(tot, ecur - top) : (R)0;
                vec[j] = g;
                bt.gbound[(b0 + (long long)(n + 1) * bs) * SP + j] = g;
            }
            lds_barrier();
        };
        if constexpr (D > 0) {
            if (nops > 0) {
#pragma unroll
                for (int d = 0; d < D; ++d) fetch(d, ovr[d], ejr[d], 0);
            }
            for (int n = 0; n < nops; n += D + 1) {
#pragma unroll
                for (int u = 0; u <= D; ++u) {
                    if (n + u < nops) {                        // (uniform)
                        fetch(n + u + D, ovr[(u + D) % (D + 1)], ejr[(u + D) % (D + 1)], 0);
                        step(n + u, ovr[u], ejr[u]);
                    }
                }
            }
        } else {
            for (int n = 0; n < nops; ++n) step(n, ovr[0], 0);
        }
    }
}

// compose_wide: the operator of a group of bt.sgroup consecutive chunks, P = F_(b-1) ... F_a -- what scan_compose_kernel is to
// the fused path (same arithmetic per product: weights 2^E shifted by the largest exponent on a column's support, the column's
// scale kept as an integer exponent; oracle/chunked_scan.py::compose), for operators that do not fit the LDS twice (round 6:
// until then the wide scan walked the flat chain of K - 1 mat-vecs of 0.95 / 1.6 us each, 74 / 128 us of an iteration of
// T = 10 000 at S = 128 and five times that at T = 50 000).  A column of P depends on that column only, so the grid is
// (group, block of 16 columns): the four waves of a workgroup share the block's columns, wave w keeps the row tiles w, w + 4, ...
// of them in the accumulator layout through the whole chain (a product is SP / 4 x SP / 64 MFMAs per wave: the f32 16x16x4
// instruction runs at the vector rate, so the rows are what gets spread), the scaled columns W go through LDS (SP x 16), F in
// slabs of KH of its columns, fetched into registers a slab ahead.  Slab rows are padded by 16 elements: the four 16-lane rows
// of a fragment read hit different banks.
template <typename R, int SP> struct ComposeWideCfg {
    static constexpr int CW = 16;                                  // columns of P per workgroup
    static constexpr int KH = 32768 / (SP * (int)sizeof(R));       // columns of F per slab (32 KB + padding)
    static constexpr int FLD = SP + 16;
    static constexpr int VPT = KH * SP * (int)sizeof(R) / 16 / 256;    // 16-byte vectors of a slab per thread
};

template <typename R, int SP>
__global__ __launch_bounds__(256) void compose_wide_kernel(BatchView<R> bt) {
    using Cfg = ComposeWideCfg<R, SP>;
    using M = Mfma16<R>;
    using acc_t = typename M::acc_t;
    typedef R RV __attribute__((ext_vector_type(16 / sizeof(R))));
    constexpr int NT = SP / 16, MT = NT / 4, KH = Cfg::KH, FLD = Cfg::FLD, VPT = Cfg::VPT, VEC = 16 / (int)sizeof(R), NSLAB = SP / KH;
    constexpr int kNoMass = -(1 << 24), kNever = -(1 << 28);
    static_assert(NT % 4 == 0 && VPT >= 1 && (KH * SP / VEC) % 256 == 0, "compose_wide shapes");
    __shared__ __attribute__((aligned(16))) R Fs[KH * FLD];
    __shared__ __attribute__((aligned(16))) R Wt[SP * 16];
    __shared__ int eF[SP];
    __shared__ int topw[4][16];
    __shared__ R sigw[4][16];
    const int sup = blockIdx.x;
    const int rec = bt.sup_rec[sup];
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int G = bt.sgroup;
    const long long cb0 = rd.tile0;
    const long long a = cb0 + (long long)bt.sup_idx[sup] * G, b = min(a + G, cb0 + rd.ntiles);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, g = lane >> 4;
    const int i = Cfg::CW * blockIdx.y + l16;          // my column of P
    R pv[MT][4];                                       // rows 16 (wave + 4 q) + M::row(lane, r)
    int eP = bt.opexp[a * SP + i];
#pragma unroll
    for (int q = 0; q < MT; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[q][r] = bt.op[(a * SP + i) * SP + 16 * (wave + 4 * q) + M::row(lane, r)];
    // slab s of operator k: columns [s KH, (s + 1) KH) of F_k = KH SP contiguous elements; thread t moves the vectors t, t + 256, ...
    RV fr[VPT];
    auto fetch = [&](long long k, int sl) {
        const RV* __restrict__ src = reinterpret_cast<const RV*>(bt.op + k * SP * SP + (long long)sl * KH * SP);
#pragma unroll
        for (int u = 0; u < VPT; ++u) fr[u] = src[u * 256 + tid];
    };
    auto stash = [&]() {                               // registers -> Fs (column kc of the slab at kc FLD)
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int v = u * 256 + tid, kc = v / (SP / VEC), m = (v % (SP / VEC)) * VEC;
            *reinterpret_cast<RV*>(Fs + kc * FLD + m) = fr[u];
        }
    };
    if (a + 1 < b) fetch(a + 1, 0);
    for (long long k = a + 1; k < b; ++k) {
        for (int t = tid; t < SP; t += 256) eF[t] = bt.opexp[k * SP + t];
        __syncthreads();
        // weights of my column, shifted by the largest exponent on its support (over the rows of all four waves)
        int tj[MT][4], top = kNever;
#pragma unroll
        for (int q = 0; q < MT; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = eF[16 * (wave + 4 * q) + M::row(lane, r)];
                tj[q][r] = (pv[q][r] > (R)0 && e > kNoMass / 2) ? e + exponent_of(pv[q][r]) : kNever;
                top = max(top, tj[q][r]);
            }
        top = max_xor<16>(top);
        top = max_xor<32>(top);
        if (g == 0) topw[wave][l16] = top;
        __syncthreads();
        top = max(max(topw[0][l16], topw[1][l16]), max(topw[2][l16], topw[3][l16]));
        const bool alive = top > -(1 << 27) && eP > kNoMass / 2;
#pragma unroll
        for (int q = 0; q < MT; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (wave + 4 * q) + M::row(lane, r);
                Wt[row * 16 + l16] = (alive && tj[q][r] > -(1 << 27)) ? scale2(pv[q][r], eF[row] - top) : (R)0;
            }
        acc_t acc[MT];
#pragma unroll
        for (int q = 0; q < MT; ++q) acc[q] = acc_t{0, 0, 0, 0};
#pragma unroll 1
        for (int sl = 0; sl < NSLAB; ++sl) {
            stash();
            __syncthreads();                           // the slab (and, first pass, W) is in LDS
            if (sl + 1 < NSLAB) fetch(k, sl + 1);      // the next slab -- or the next operator's first one -- during the MFMAs
            else if (k + 1 < b) fetch(k + 1, 0);
#pragma unroll 4
            for (int kk = 0; kk < KH / 4; ++kk) {
                const R bv = Wt[(sl * KH + 4 * kk + g) * 16 + l16];
#pragma unroll
                for (int q = 0; q < MT; ++q) acc[q] = M::mma(Fs[(4 * kk + g) * FLD + 16 * (wave + 4 * q) + l16], bv, acc[q]);
            }
            __syncthreads();                           // before the slab is overwritten
        }
        R sig = 0;
#pragma unroll
        for (int q = 0; q < MT; ++q) sig += (acc[q][0] + acc[q][1]) + (acc[q][2] + acc[q][3]);
        sig = add_xor<16>(sig);
        sig = add_xor<32>(sig);
        if (g == 0) sigw[wave][l16] = sig;
        __syncthreads();
        sig = (sigw[0][l16] + sigw[1][l16]) + (sigw[2][l16] + sigw[3][l16]);
        if (alive && sig > (R)0) {
            const int e = rescale_exponent(sig);
#pragma unroll
            for (int q = 0; q < MT; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[q][r] = scale2(acc[q][r], -e);
            eP += top + e;
        } else {
#pragma unroll
            for (int q = 0; q < MT; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[q][r] = 0;
            eP = kNoMass;
        }
        // (eF, W, topw and sigw are rewritten only behind the next product's first barrier resp. after its second one)
    }
    R* __restrict__ dst = bt.sop + ((long long)sup * SP + i) * SP;
#pragma unroll
    for (int q = 0; q < MT; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[16 * (wave + 4 * q) + M::row(lane, r)] = pv[q][r];
    if (wave == 0 && g == 0) bt.sopexp[(long long)sup * SP + i] = eP;
}

// Re-run of one chunk in one direction by one wavefront, lane = state (NREG = SP / 64 states per lane); the
// normalised recursions of fb_seq_kernel, started from the chunk's boundary vector.  grid = (ntiles_total, 2).
template <typename R, int SP>
__global__ __launch_bounds__(64) void scan3_wide_kernel(BatchView<R> bt) {
    constexpr int NREG = SP / 64, U = 8;
    const int tile = blockIdx.x, dir = blockIdx.y;
    const int rec = bt.tile_rec[tile];
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int t0 = bt.tile_t0[tile];
    const int len = min(kTileFrames, rd.T - t0);
    const int lane = threadIdx.x;
    const R lp = (R)rd.lp;
    const R* __restrict__ B = bt.bmat + (rd.row0 + t0) * SP;
    R c[NREG], x[NREG];
    R ssum = 0;
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        const int j = lane + 64 * r;
        c[r] = (j < rd.S) ? (R)((1.0 - rd.lp) * bt.pi[(long long)rec * SP + j] + 1e-8) : (R)0;
        x[r] = (dir == 0 ? bt.fbound : bt.gbound)[(long long)tile * SP + j];
        ssum += x[r];
    }
    ssum = allreduce_sum<64>(ssum);
    auto load_rows = [&](R (&dst)[U][NREG], int ffirst, int step) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int f = min(max(ffirst + step * u, 0), len - 1);
#pragma unroll
            for (int r = 0; r < NREG; ++r) dst[u][r] = B[(long long)f * SP + lane + 64 * r];
        }
    };
    if (dir == 0) {
        R* __restrict__ A = bt.ahat + (rd.row0 + t0) * SP;
        R* __restrict__ sfw = bt.sfw + rd.row0 + t0;
        const bool first = (t0 == 0);                  // frame 0 of the recording: a_0 = b_0 (ip + 1e-8), no transition
        if (!first) {
            const R inv = fast_rcp(ssum);              // ahat of the frame before the chunk
#pragma unroll
            for (int r = 0; r < NREG; ++r) x[r] *= inv;
        }
        R b0[U][NREG], b1[U][NREG];
        auto run = [&](const R (&blk)[U][NREG], int ffirst) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int f = ffirst + u;
                if (f >= len) break;                   // wave-uniform
                R a[NREG];
                R s = 0;
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    a[r] = (first && f == 0) ? blk[u][r] * x[r] : blk[u][r] * (lp * x[r] + c[r]);
                    s += a[r];
                }
                s = allreduce_sum<64>(s);
                const R inv = fast_rcp(s);
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    x[r] = a[r] * inv;
                    A[(long long)f * SP + lane + 64 * r] = x[r];
                }
                if (lane == 0) sfw[f] = s;
            }
        };
        load_rows(b0, 0, 1);
        for (int fb = 0; fb < len; fb += 2 * U) {
            load_rows(b1, fb + U, 1);
            run(b0, fb);
            load_rows(b0, fb + 2 * U, 1);
            run(b1, fb + U);
        }
    } else {
        R* __restrict__ Bh = bt.bhat + (rd.row0 + t0) * SP;
        {
            const R scl = fast_rcp(ssum) * (R)SP;      // mean 1, like the ones the sequential walk starts from
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                x[r] *= scl;
                Bh[(long long)(len - 1) * SP + lane + 64 * r] = x[r];
            }
        }
        R b0[U][NREG], b1[U][NREG];
        // block rows are b[f+1] for f = ffirst, ffirst-1, ...: consuming row f+1 gives bhat_f
        auto run = [&](const R (&blk)[U][NREG], int ffirst) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int f = ffirst - u;
                if (f < 0) break;                      // wave-uniform
                R e[NREG];
                R q = 0;
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    e[r] = blk[u][r] * x[r];
                    q += c[r] * e[r];
                }
                q = allreduce_sum<64>(q);
                const R sc = lp * fast_rcp(q);
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    x[r] = sc * e[r] + (R)1;
                    Bh[(long long)f * SP + lane + 64 * r] = x[r];
                }
            }
        };
        load_rows(b0, len - 1, -1);                    // rows len-1, len-2, ... serve f = len-2, len-3, ...
        for (int fb = len - 2; fb >= 0; fb -= 2 * U) {
            load_rows(b1, fb + 1 - U, -1);
            run(b0, fb);
            load_rows(b0, fb + 1 - 2 * U, -1);
            run(b1, fb - U);
        }
    }
}

}  // namespace vbx
