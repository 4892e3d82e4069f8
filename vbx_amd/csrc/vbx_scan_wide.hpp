// vbx_scan_wide.hpp -- the exact chunked scan of vbx_scan.hpp for 64 < S <= 256 states.
//
// Same three steps (chunk operators, boundary walk, re-run from the boundaries; VBx.py:146-175 in a different
// association order), other shapes: an S x S operator is 64-256 KB, so
//   scan1_wide  one workgroup builds 16 columns of a chunk's operator (16 lanes per column, S/16 states per lane);
//               b of the chunk goes through LDS in pieces of at most 32 KB
//   scan2_wide  one workgroup of 1024 threads per (recording, direction) walks the chain; every mat-vec reads the
//               operator straight from L2 / HBM (coalesced), partial sums meet in LDS.  Forward: thread = (row,
//               column block); backward ((F^T g)_j = <column j, g>): a wavefront per output, lanes over the rows
//   scan3_wide  one wavefront per (chunk, direction) re-runs the chunk with lane = state (S/64 states per lane),
//               rows of b prefetched from L2 eight frames ahead, and writes ahat / bhat / the forward scales exactly
//               like scan3 -- post_kernel and the accumulation take it from there.
// A recording of T = 10 000 with S = 200 takes ~0.3 ms per iteration this way; the sequential walk it replaces
// (fb_seq_kernel: one wavefront over all T frames) 2 ms.
#pragma once
#include "vbx_scan.hpp"

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

// block-wide maximum of one int per thread of the first SP threads (SP a multiple of 64); every thread gets it
template <int SP> __device__ __forceinline__ int wide_block_max(int v, int* wmax, int tid) {
    const int m = allreduce_max<64>(v);
    if ((tid & 63) == 0 && tid < SP) wmax[tid >> 6] = m;
    __syncthreads();
    int top = wmax[0];
#pragma unroll
    for (int w = 1; w < SP / 64; ++w) top = max(top, wmax[w]);
    return top;
}

template <typename R, int SP>
__global__ __launch_bounds__(1024) void scan2_wide_kernel(BatchView<R> bt) {
    constexpr int HL = 1024 / SP, NI = SP / HL;        // forward: column blocks per row, columns per block
    constexpr int NL = SP / 64;                        // backward: rows per lane
    __shared__ R vec[SP];                              // the vector being pushed through the chain (weights, forward)
    __shared__ R part[1024];
    __shared__ int wmax[SP / 64];
    const int rec = blockIdx.x, dir = blockIdx.y;
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int K = rd.ntiles;
    const long long cb0 = rd.tile0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = tid % SP, h = tid / SP;
    constexpr int kNone = -(1 << 28);
    if (dir == 0) {
        R y = 0;
        if (tid < SP) {
            y = (j < rd.S) ? (R)(bt.ip[(long long)rec * SP + j] + 1e-8) : (R)0;
            bt.fbound[cb0 * SP + j] = y;
        }
        // The operator entries of a thread (and the exponent of its column) are requested at the top of a chain step
        // for the NEXT step when they fit the registers twice (S <= 128), else for this step before the weights are
        // known, as many at a time as the registers hold: a dependent global load costs ~1 us, as much as the rest of
        // the step.
        constexpr bool kAhead = NI <= 32;
        constexpr int CH = kAhead ? NI : (sizeof(R) == 8 ? 32 : 64);     // operator entries in registers at a time
        R ov[CH], ovn[kAhead ? NI : 1];
        int ej = 0, ejn = 0;
        if (kAhead && K > 1) {
            const R* __restrict__ op = bt.op + cb0 * SP * SP;
#pragma unroll
            for (int ii = 0; ii < CH; ++ii) ov[ii] = op[(long long)(h * NI + ii) * SP + j];
            ej = bt.opexp[cb0 * SP + j];
        }
        for (int n = 0; n + 1 < K; ++n) {
            const R* __restrict__ opc = bt.op + (cb0 + n) * SP * SP;
            if constexpr (kAhead) {
                const long long k = cb0 + min(n + 1, K - 2);
                const R* __restrict__ op = bt.op + k * SP * SP;
#pragma unroll
                for (int ii = 0; ii < NI; ++ii) ovn[ii] = op[(long long)(h * NI + ii) * SP + j];
                ejn = bt.opexp[k * SP + j];
            } else {
#pragma unroll
                for (int ii = 0; ii < CH; ++ii) ov[ii] = opc[(long long)(h * NI + ii) * SP + j];
                ej = bt.opexp[(cb0 + n) * SP + j];
            }
            // weights y_i 2^{E_i}, shifted by the largest on the support of y
            int tj = kNone;
            if (tid < SP && y > (R)0) tj = ej + exponent_of(y);
            const int top = wide_block_max<SP>(tj, wmax, tid);
            if (tid < SP) vec[j] = (y > (R)0) ? scale2(y, ej - top) : (R)0;
            __syncthreads();
            R acc = 0;
#pragma unroll
            for (int c0 = 0; c0 < NI; c0 += CH) {
                if (c0 > 0) {
#pragma unroll
                    for (int ii = 0; ii < CH; ++ii) ov[ii] = opc[(long long)(h * NI + c0 + ii) * SP + j];
                }
#pragma unroll
                for (int ii = 0; ii < CH; ++ii) acc += vec[h * NI + c0 + ii] * ov[ii];
            }
            part[tid] = acc;
            if constexpr (kAhead) {
#pragma unroll
                for (int ii = 0; ii < NI; ++ii) ov[ii] = ovn[ii];
                ej = ejn;
            }
            __syncthreads();
            if (tid < SP) {
                R tot = 0;
#pragma unroll
                for (int q = 0; q < HL; ++q) tot += part[q * SP + j];
                y = (j < rd.S) ? tot : (R)0;           // padded speakers carry no mass
                bt.fbound[(cb0 + n + 1) * SP + j] = y;
            }
            __syncthreads();
        }
    } else {
        R g = 0;
        if (tid < SP) {
            g = (j < rd.S) ? (R)1 : (R)0;
            bt.gbound[(cb0 + K - 1) * SP + j] = g;
            vec[j] = g;
        }
        __syncthreads();
        // (F^T g)_c = 2^{E_c} <column c, g>: a wavefront per output column, columns wave, wave + 16, ...; the columns
        // of the next operator are requested right after the dot products of this one when the registers allow it
        // (S <= 128), else at the top of their own step
        constexpr int NC = SP / 16;
        constexpr bool kAhead = NC * NL <= 16;
        R ov[NC][NL];
        int ej = 0;
        auto fetch = [&](int n) {
            const long long k = cb0 + K - 1 - min(n, K - 2);
            const R* __restrict__ op = bt.op + k * SP * SP;
#pragma unroll
            for (int q = 0; q < NC; ++q)
#pragma unroll
                for (int r = 0; r < NL; ++r) ov[q][r] = op[(long long)(wave + 16 * q) * SP + lane + 64 * r];
            ej = bt.opexp[k * SP + j];
        };
        if (kAhead && K > 1) fetch(0);
        for (int n = 0; n + 1 < K; ++n) {
            const long long k = cb0 + K - 1 - n;
            if (!kAhead) fetch(n);
            R gl[NL];
#pragma unroll
            for (int r = 0; r < NL; ++r) gl[r] = vec[lane + 64 * r];
            R dots[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                R acc = 0;
#pragma unroll
                for (int r = 0; r < NL; ++r) acc += ov[q][r] * gl[r];
                dots[q] = acc;
            }
            const int ecur = ej;
            if (kAhead) fetch(n + 1);
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const R acc = allreduce_sum<64>(dots[q]);
                if (lane == 0) part[wave + 16 * q] = acc;
            }
            __syncthreads();
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
                bt.gbound[(k - 1) * SP + j] = g;
            }
            __syncthreads();
        }
    }
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
