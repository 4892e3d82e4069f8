// vbx_scan.hpp -- exact chunked parallel scan for the forward-backward recursion.
//
// reference: VBx/VBx.py:146-175 (forward_backward).  With c_j = (1-lp) pi_j + 1e-8 the matrix
// (tr + 1e-8) of VBx.py:158 is  lp*I + 1 c^T,  so one frame acts on the forward vector as
//     a_t     = M_t a_{t-1},        M_t  = diag(b_t) (lp*I + c 1^T)               (VBx.py:167-168)
// and on the backward vector as
//     be_{t-1} = Mt_t be_t,         Mt_t = (lp*I + 1 c^T) diag(b_t)               (VBx.py:170-171)
// Both are linear, so T frames are cut into chunks (one chunk = one tile of kTileFrames frames)
// and the 2T-step dependency chain becomes
//   scan1  per chunk, all chunks in parallel: the S x S transfer operator
//             F_k = M_{t1-1} ... M_{t0};   the backward operator is its transpose for free:
//             Mt_t^T = M_t  =>  B_k = Mt_{t0} ... Mt_{t1-1} = F_k^T
//          one operator column per lane (group), the column lives in registers, every sum over
//          states is in-lane; columns are rescaled by exact powers of two (exponent kept aside).
//   scan2  per recording: K-1 sequential mat-vecs give the vectors at every chunk boundary.
//   scan3  per chunk, all chunks in parallel: re-run the chunk from its true boundary vectors,
//          lane = speaker, one cross-lane reduction per frame; writes ahat, bhat and the chunk's
//          share of the total log-likelihood.
// The result is the same recursion in a different association order -- no approximation.
#pragma once
#include "vbx_kernels.hpp"

namespace vbx {

__device__ __forceinline__ int exponent_of(float v) { return __builtin_amdgcn_frexp_expf(v); }
__device__ __forceinline__ int exponent_of(double v) { return __builtin_amdgcn_frexp_exp(v); }
// Exponent used to rescale an operator column whose sum is v.  Clamped so that 2^-e stays finite
// when v is subnormal (b can be a subnormal number; the next frame finishes the renormalisation).
__device__ __forceinline__ int rescale_exponent(float v) {
    return v > 0.0f ? max(-126, min(126, __builtin_amdgcn_frexp_expf(v))) : 0;
}
__device__ __forceinline__ int rescale_exponent(double v) {
    return v > 0.0 ? max(-1022, min(1022, __builtin_amdgcn_frexp_exp(v))) : 0;
}
__device__ __forceinline__ float scale2(float v, int e) { return __builtin_amdgcn_ldexpf(v, e); }
__device__ __forceinline__ double scale2(double v, int e) { return __builtin_amdgcn_ldexp(v, e); }

// sum over the PH adjacent lanes that share one operator column (PH = 1, 2, 4, 8 or 16)
template <int PH, typename R> __device__ __forceinline__ R column_sum(R v) {
    if (PH >= 2) v += dpp_mov<0xB1>(v);
    if (PH >= 4) v += dpp_mov<0x4E>(v);
    if (PH >= 8) v += dpp_mov<0x141>(v);
    if (PH >= 16) v += dpp_mov<0x140>(v);
    return v;
}

// =======================================================================================
// scan1: forward transfer operator of one chunk (the backward one is its transpose).
// grid = ntiles_total, block = SP*SP/4.  lane = (column, part): PH = SP/4 adjacent lanes share a
// column and hold four states each, so a frame costs ~25 instructions per wave and many light
// waves share a SIMD (the first version kept 16 states per lane in one wave per chunk and was
// bound by its own instruction count).
//   x <- b_t * (lp*x + c*sum(x)),  t = t0 .. t0+len-1; frame 0 of a recording only applies b_0
//   (the initial vector ip + 1e-8 of VBx.py:163 is fed in by scan2).
// =======================================================================================
template <typename R, int SP>
__global__ __launch_bounds__(SP * SP / 4) void scan1_kernel(BatchView<R> bt) {
    constexpr int NR = 4, PH = SP / 4, NTHR = SP * PH;
    using R4 = typename Vec<R>::v4;
    __shared__ __attribute__((aligned(16))) R btile[kTileFrames * SP];
    const int tile = blockIdx.x;
    const int rec = bt.tile_rec[tile];
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int t0 = bt.tile_t0[tile];
    const int len = min(kTileFrames, rd.T - t0);
    const int tid = threadIdx.x;
    stage_to_lds<(kTileFrames * SP / 4 + NTHR - 1) / NTHR>(
        reinterpret_cast<R4*>(btile), reinterpret_cast<const R4*>(bt.bmat + (rd.row0 + t0) * SP), len * SP / 4, tid, NTHR);
    __syncthreads();
    const int col = tid / PH, j0 = (tid % PH) * NR;
    const R lp = (R)rd.lp;
    R x[NR], c[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = j0 + r;
        x[r] = (j == col) ? (R)1 : (R)0;
        c[r] = (j < rd.S) ? (R)((1.0 - rd.lp) * bt.pi[(long long)rec * SP + j] + 1e-8) : (R)0;
    }
    int expo = 0, first = 0;
    if (t0 == 0) {                       // frame 0: x <- b_0 * x
        const R4 b = *reinterpret_cast<const R4*>(btile + j0);
#pragma unroll
        for (int r = 0; r < NR; ++r) x[r] *= b[r];
        first = 1;
    }
#pragma unroll 2
    for (int step = first; step < len; ++step) {
        const R4 b = *reinterpret_cast<const R4*>(btile + step * SP + j0);
        const R sig = column_sum<PH>((x[0] + x[1]) + (x[2] + x[3]));
        const int e = rescale_exponent(sig);
        expo += e;
        const R sc = scale2((R)1, -e);
        const R lps = lp * sc, sgs = sig * sc;
#pragma unroll
        for (int r = 0; r < NR; ++r) x[r] = b[r] * (lps * x[r] + c[r] * sgs);
    }
    {   // final power-of-two normalisation: column sums end in [0.5, 1)
        const R sig = column_sum<PH>((x[0] + x[1]) + (x[2] + x[3]));
        const int e = rescale_exponent(sig);
        expo += e;
#pragma unroll
        for (int r = 0; r < NR; ++r) x[r] = scale2(x[r], -e);
        // an all-zero column (b = 0 for its state at frame 0, or a padded state) must never win
        // the exponent maximum in scan2
        if (!(sig > (R)0)) expo = -(1 << 24);
    }
    R4 xv = R4{x[0], x[1], x[2], x[3]};
    *reinterpret_cast<R4*>(bt.op + ((long long)tile * SP + col) * SP + j0) = xv;
    if ((tid % PH) == 0) bt.opexp[(long long)tile * SP + col] = expo;
}

// =======================================================================================
// scan2: chunk-boundary vectors of one recording and one direction.
// grid = (n_rec, 2), block = 256: wave 0 walks the chain of K-1 mat-vecs, waves 1-3 stream the
// operators (4 KB each at SP = 32) from L2 into a double-buffered LDS ring RB operators ahead --
// a single wave waiting on its own global loads spent 5 us per chunk in the first version.
//   y' = sum_i (y_i 2^{E_i}) col_i,  weights shifted by the largest exponent on the support of y
//   so that nothing that matters can underflow.
// lane = (row j, part h): HL = 64/SP lanes share an output row and split the columns.
// =======================================================================================
template <typename R, int SP> struct Scan2Cfg {
    static constexpr int kOpElems = SP * SP;
    static constexpr int kOpBytes = kOpElems * (int)sizeof(R);
    static constexpr int kBudget = 57344;
    static constexpr int RB = (kOpBytes * 16 <= kBudget) ? 8 : (kOpBytes * 8 <= kBudget) ? 4
                              : (kOpBytes * 4 <= kBudget) ? 2 : 1;
    static constexpr int NBUF = (2 * RB * kOpBytes <= kBudget) ? 2 : 1;
};

// One step of the boundary walk on ONE wavefront, shared by scan2_kernel and by the folded walk of chunk_post_kernel (which must
// give the very same bits): y <- F y (dir 0: weights y_i 2^(E_i) shifted by the largest exponent on the support of y) or
// y <- F^T y (dir 1: outputs rescaled by the largest exponent).  lane = (row j, part h), HL = 64 / SP lanes share a row and
// split the columns; `opl` the operator in LDS (column-major as stored), `ex` its SP column exponents, `wl` SP scratch words
// of the wave.  NI operator entries per lane are fetched by the caller (scan2 fetches them a step ahead).
template <typename R, int SP, int dir>
__device__ __forceinline__ void walk_fetch(R (&opv)[SP / (64 / SP)], int& ej, const R* opl, const int* ex, int j, int h) {
    constexpr int HL = 64 / SP, NI = SP / HL;
#pragma unroll
    for (int ii = 0; ii < NI; ++ii)                   // fwd: row j of columns h*NI..; bwd: column j, rows h*NI..
        opv[ii] = dir == 0 ? opl[(h * NI + ii) * SP + j] : opl[j * SP + h * NI + ii];
    ej = ex[j];
}
template <typename R, int SP, int dir>
__device__ __forceinline__ R walk_step(R y, const R (&opv)[SP / (64 / SP)], int ej, R* wl, int j, int h, int S) {
    constexpr int HL = 64 / SP, NI = SP / HL;
    R w = y;
    if (dir == 0) {                       // weights y_i 2^{E_i}, shifted by the largest on the support
        const bool pos = y > (R)0;
        const int tj = pos ? ej + exponent_of(y) : -(1 << 28);
        const int top = allreduce_max<64>(tj);
        w = pos ? scale2(y, ej - top) : (R)0;
    }
    if (h == 0) wl[j] = w;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    R acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int ii = 0; ii < NI; ++ii) acc[ii & 3] += wl[h * NI + ii] * opv[ii];
    R tot = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    tot = column_sum<HL>(tot);
    if (dir == 1) {                       // (F^T g)_j = 2^{E_j} <col_j, g>: rescale the outputs
        const bool pos = tot > (R)0;
        const int tj = pos ? ej + exponent_of(tot) : -(1 << 28);
        const int top = allreduce_max<64>(tj);
        tot = pos ? scale2(tot, ej - top) : (R)0;
    }
    __builtin_amdgcn_wave_barrier();
    return (j < S) ? tot : (R)0;          // padded speakers carry no mass in either direction
}

// The walk is used at several levels (`level` argument of scan2_kernel):
//   0  flat: every chunk operator of a recording, from the initial vector          (grid.x = recording)
//   2  the group operators of scan_compose_kernel: boundaries at the group edges     (grid.x = recording)
//   3  inside one group, from the boundary left at its edge                          (grid.x = group)
//   4  the level-2 group operators (groups of G2 groups): boundaries at their edges  (grid.x = recording)
//   5  inside one level-2 group over its group operators, from the boundary level 4 left at its edge (grid.x = level-2 group)
// A chain of K-1 dependent mat-vecs costs 0.3-0.9 us each; with groups of G chunks the critical path is G products +
// K/G + G mat-vecs (levels 2, 3); for very long recordings a third level (4, 5, 3) makes it G + G2 products and
// K/(G G2) + G2 + G mat-vecs: T = 200 000 (K = 1563): 17 products + 104 mat-vecs -> 14 products + 40 mat-vecs.
template <typename R, int SP, int dir>
__device__ __forceinline__ void scan2_body(const BatchView<R>& bt, int level, R* ring, int* exps, R* wl) {
    using Cfg = Scan2Cfg<R, SP>;
    using R4 = typename Vec<R>::v4;
    constexpr int HL = 64 / SP, NI = SP / HL, RB = Cfg::RB, NBUF = Cfg::NBUF, OPSZ = Cfg::kOpElems;
    const int rec = level == 3 ? bt.sup_rec[blockIdx.x] : level == 5 ? bt.sup2_rec[blockIdx.x] : blockIdx.x;
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int K = chunk_count(rd, bt.spt), G = bt.sgroup;      // chunks of this recording, first one cb0
    const long long cb0 = (long long)rd.tile0 * bt.spt;
    // chain step n uses operator op0 + n*os and writes the boundary b0 + (n+1)*bs; the chain starts from
    // bound[binit] (level 3) or from the initial vector, which it also stores at bound[binit]
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
    } else if (level == 4) {                    // over the level-2 group operators of the recording
        const int G2 = bt.sgroup2, n1 = (K + G - 1) / G, n2 = (n1 + G2 - 1) / G2;
        ops = bt.sop2;
        oexp = bt.sopexp2;
        nops = n2 - 1;
        op0 = dir == 0 ? rd.sup20 : rd.sup20 + n2 - 1;
        bs = dir == 0 ? G * G2 : -G * G2;
        b0 = dir == 0 ? cb0 : cb0 + (long long)n2 * G * G2 - 1;
    } else if (level == 5) {                    // inside one level-2 group: over its (level-1) group operators
        const int G2 = bt.sgroup2, n1 = (K + G - 1) / G;
        const int a = bt.sup2_idx[blockIdx.x] * G2, b = min(a + G2, n1);      // level-1 groups [a, b) of this recording
        ops = bt.sop;
        oexp = bt.sopexp;
        nops = b - 1 - a;
        if (nops <= 0) return;
        op0 = dir == 0 ? rd.sup0 + a : rd.sup0 + b - 1;
        bs = dir == 0 ? G : -G;
        // forward: from the vector entering the first chunk of the level-2 group; backward: from the vector at the last
        // chunk it really has (the recording's last group may be short), written to the regular positions below it
        binit = dir == 0 ? cb0 + (long long)a * G : min(cb0 + (long long)b * G, cb0 + K) - 1;
        b0 = dir == 0 ? binit : cb0 + (long long)b * G - 1;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane / HL, h = lane % HL;
    R* __restrict__ bound = dir == 0 ? bt.fbound : bt.gbound;

    // copies the operators of round r (chain steps r*RB ...) into ring buffer `buf`; every load of
    // the round is issued before the first LDS store
    auto load_round = [&](int r, int buf, int tid, int nthreads) {
        constexpr int PER = (OPSZ / 4 + 191) / 192;       // vectors per thread per operator (192 loaders)
        R4 tmp[RB][PER];
        int etmp[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int n = r * RB + q;
            const long long base = (op0 + (n < nops ? (long long)n * os : 0)) * SP;
            const R4* __restrict__ src = reinterpret_cast<const R4*>(ops + base * SP);
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int e = u * nthreads + tid;
                if (n < nops && e < OPSZ / 4) tmp[q][u] = src[e];
            }
            etmp[q] = (n < nops && tid < SP) ? oexp[base + tid] : 0;
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int n = r * RB + q;
            R4* dst = reinterpret_cast<R4*>(ring + (long long)(buf * RB + q) * OPSZ);
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int e = u * nthreads + tid;
                if (n < nops && e < OPSZ / 4) dst[e] = tmp[q][u];
            }
            if (n < nops && tid < SP) exps[(buf * RB + q) * SP + tid] = etmp[q];
        }
    };

    R y = 0;
    if (wave == 0) {
        if (level == 3 || level == 5) {
            y = bound[binit * SP + j];
        } else {
            if (dir == 0) y = (j < rd.S) ? (R)(bt.ip[(long long)rec * SP + j] + 1e-8) : (R)0;
            else y = (j < rd.S) ? (R)1 : (R)0;
            if (h == 0) bound[binit * SP + j] = y;
        }
    }
    auto compute_round = [&](int r, int buf) {
        R opv[2][NI];
        int ejv[2];
        auto fetch = [&](int q, int slot) {       // LDS -> registers, issued one chain step ahead
            walk_fetch<R, SP, dir>(opv[slot], ejv[slot], ring + (long long)(buf * RB + q) * OPSZ, exps + (buf * RB + q) * SP, j, h);
        };
        fetch(0, 0);
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int n = r * RB + q;
            if (n >= nops) break;
            if (q + 1 < RB) fetch(q + 1, (q + 1) & 1);
            y = walk_step<R, SP, dir>(y, opv[q & 1], ejv[q & 1], wl, j, h, rd.S);
            if (h == 0) bound[(b0 + (long long)(n + 1) * bs) * SP + j] = y;
        }
    };

    const int rounds = (nops + RB - 1) / RB;
    if (NBUF == 2) {
        load_round(0, 0, threadIdx.x, 256);
        __syncthreads();
        for (int r = 0; r < rounds; ++r) {
            if (wave == 0) compute_round(r, r & 1);
            else if (r + 1 < rounds) load_round(r + 1, (r + 1) & 1, threadIdx.x - 64, 192);
            __syncthreads();
        }
    } else {
        for (int r = 0; r < rounds; ++r) {
            load_round(r, 0, threadIdx.x, 256);
            __syncthreads();
            if (wave == 0) compute_round(r, 0);
            __syncthreads();
        }
    }
}

template <typename R, int SP>
__global__ __launch_bounds__(256) void scan2_kernel(BatchView<R> bt, int level) {
    using Cfg = Scan2Cfg<R, SP>;
    __shared__ __attribute__((aligned(16))) R ring[Cfg::NBUF * Cfg::RB * Cfg::kOpElems];
    __shared__ int exps[Cfg::NBUF * Cfg::RB * SP];
    __shared__ __attribute__((aligned(16))) R wl[SP];
    if (blockIdx.y == 0) scan2_body<R, SP, 0>(bt, level, ring, exps, wl);     // direction is a compile-time constant:
    else scan2_body<R, SP, 1>(bt, level, ring, exps, wl);                     // the two chains read the operator differently
}

// max / sum over the W adjacent lanes of a group (W = 4, 8 or 16)
template <int W> __device__ __forceinline__ int group_max(int v) {
    v = max(v, dpp_mov<0xB1>(v));
    v = max(v, dpp_mov<0x4E>(v));
    if (W >= 8) v = max(v, dpp_mov<0x141>(v));
    if (W >= 16) v = max(v, dpp_mov<0x140>(v));
    return v;
}

// =======================================================================================
// scan_compose: the operator of a group of `sgroup` consecutive chunks, P = F_(b-1) ... F_(a+1) F_a.
// grid = nsup_total, block = 256.  One product pushes every column of P through the next chunk operator exactly like
// scan2 pushes a boundary vector (weights 2^E shifted by the largest exponent on the column's support) but keeps the
// column's scale as an integer exponent:
//     P'[:, i] = 2^(e_i + top_i) sum_j (P[j, i] 2^(E_j - top_i)) F[:, j],   renormalised to a sum in [0.5, 1),
// i.e. P' = F W with W[j][i] = P[j, i] 2^(E_j - top_i): an S x S x S product on v_mfma 16x16x4.  Wave w < SP/16 owns the
// columns [16w, 16w + 16) of P in the accumulator layout (a column is spread over the four 16-lane rows of the wave);
// F (as fetched: column j contiguous) and W (row j contiguous) live in LDS and both fragments are conflict-free reads.
// Round 2: the products were S^2 LDS reads per thread on the VALU before (7 us per product at SP = 64, the largest
// part of the boundary walk of a long recording).  Model: oracle/chunked_scan.py::compose.
// =======================================================================================
// `level` 1: groups of chunk operators (op -> sop); 2: groups of group operators (sop -> sop2, three-level walk).
template <typename R, int SP>
__global__ __launch_bounds__(256) void scan_compose_kernel(BatchView<R> bt, int level) {
    using M = Mfma16<R>;
    using acc_t = typename M::acc_t;
    using R4 = typename Vec<R>::v4;
    constexpr int NT = SP / 16;                        // 16 x 16 tiles per dimension (SP = 16 / 32 / 64)
    constexpr int VPT = SP * SP / 4 / 256 > 0 ? SP * SP / 4 / 256 : 1;   // 16-byte vectors of an operator per thread
    constexpr int kNoMass = -(1 << 24), kNever = -(1 << 28);
    __shared__ __attribute__((aligned(16))) R Fl[SP * SP];
    __shared__ __attribute__((aligned(16))) R Wt[SP * SP];
    __shared__ int eF[SP];
    const int sup = blockIdx.x;
    const int rec = level == 2 ? bt.sup2_rec[sup] : bt.sup_rec[sup];
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int G = bt.sgroup;
    const long long cb0 = (long long)rd.tile0 * bt.spt;
    // the operators [a, b) multiplied here, in the array they come from, and where the product goes
    long long a, b;
    const R* __restrict__ in_op = bt.op;
    const int* __restrict__ in_exp = bt.opexp;
    R* __restrict__ out_op = bt.sop;
    int* __restrict__ out_exp = bt.sopexp;
    if (level == 2) {
        const int n1 = (chunk_count(rd, bt.spt) + G - 1) / G;
        a = rd.sup0 + (long long)bt.sup2_idx[sup] * bt.sgroup2;
        b = min(a + bt.sgroup2, (long long)rd.sup0 + n1);
        in_op = bt.sop;
        in_exp = bt.sopexp;
        out_op = bt.sop2;
        out_exp = bt.sopexp2;
    } else {
        a = cb0 + (long long)bt.sup_idx[sup] * G;
        b = min(a + G, cb0 + chunk_count(rd, bt.spt));
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, g = lane >> 4;
    const bool mul = wave < NT;                        // the waves that hold columns of P
    const int i = (mul ? 16 * wave : 0) + l16;         // my column
    R pv[NT][4];
    int eP = kNoMass;
    if (mul) {
        eP = in_exp[(long long)a * SP + i];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[mt][r] = in_op[((long long)a * SP + i) * SP + 16 * mt + M::row(lane, r)];
    }
    R4 fr[VPT];
    int efr = 0;
    auto fetch = [&](long long k) {                        // chunk operator k: global -> registers
        const R4* __restrict__ src = reinterpret_cast<const R4*>(in_op + (long long)k * SP * SP);
#pragma unroll
        for (int u = 0; u < VPT; ++u)
            if (u * 256 + tid < SP * SP / 4) fr[u] = src[u * 256 + tid];
        if (tid < SP) efr = in_exp[(long long)k * SP + tid];
    };
    if (a + 1 < b) fetch(a + 1);
    for (long long k = a + 1; k < b; ++k) {
#pragma unroll
        for (int u = 0; u < VPT; ++u)
            if (u * 256 + tid < SP * SP / 4) reinterpret_cast<R4*>(Fl)[u * 256 + tid] = fr[u];
        if (tid < SP) eF[tid] = efr;
        __syncthreads();
        if (k + 1 < b) fetch(k + 1);                       // next operator in flight during this product
        int top = kNever;
        bool alive = false;
        if (mul) {
            // weights of my column, shifted by the largest exponent on its support
            int tj[NT][4];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = eF[16 * mt + M::row(lane, r)];
                    tj[mt][r] = (pv[mt][r] > (R)0 && e > kNoMass / 2) ? e + exponent_of(pv[mt][r]) : kNever;
                    top = max(top, tj[mt][r]);
                }
            top = max_xor<16>(top);
            top = max_xor<32>(top);
            alive = top > -(1 << 27) && eP > kNoMass / 2;
#pragma unroll
            for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * mt + M::row(lane, r);
                    Wt[row * SP + i] = (alive && tj[mt][r] > -(1 << 27)) ? scale2(pv[mt][r], eF[row] - top) : (R)0;
                }
        }
        __syncthreads();
        if (mul) {
            acc_t acc[NT];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) acc[mt] = acc_t{0, 0, 0, 0};
#pragma unroll 4
            for (int kk = 0; kk < SP / 4; ++kk) {
                const R bv = Wt[(4 * kk + g) * SP + i];
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) acc[mt] = M::mma(Fl[(4 * kk + g) * SP + 16 * mt + l16], bv, acc[mt]);
            }
            R sig = 0;
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) sig += (acc[mt][0] + acc[mt][1]) + (acc[mt][2] + acc[mt][3]);
            sig = add_xor<16>(sig);
            sig = add_xor<32>(sig);
            if (alive && sig > (R)0) {
                const int e = rescale_exponent(sig);
#pragma unroll
                for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[mt][r] = scale2(acc[mt][r], -e);
                eP += top + e;
            } else {
#pragma unroll
                for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[mt][r] = 0;
                eP = kNoMass;
            }
        }
        __syncthreads();                                   // Fl / Wt are rewritten by the next product
    }
    if (mul) {
        R* __restrict__ dst = out_op + ((long long)sup * SP + i) * SP;
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[16 * mt + M::row(lane, r)] = pv[mt][r];
        if (g == 0) out_exp[(long long)sup * SP + i] = eP;
    }
}

// =======================================================================================
// scan3: re-run chunks from their boundary vectors.  One wavefront = four tasks of 16 lanes:
// (chunk A, fwd) (chunk A, bwd) (chunk B, fwd) (chunk B, bwd); a lane holds NREG = SP/16 adjacent
// speakers, so the only cross-lane operation per frame is ONE 16-lane DPP all-reduce shared by
// the four tasks (measured: a DPP stage costs ~20 cycles of a lone wave, a VALU op 4.4):
//   fwd: u = b_t (lp*ahat + c)      r = sum u        ahat' = u / r        (s_t = r -> sfw)
//   bwd: u = b_t * bhat             r = sum c*u      bhat' = lp*u/r + 1
// grid = ceil(ntiles_total / 2), block = 64.  The rows of b of both chunks are staged in LDS with
// every load in flight at once; the frame loop is unrolled in blocks of 16 and has no branches.
// =======================================================================================
template <typename R, int SP>
__global__ __launch_bounds__(64) void scan3_kernel(BatchView<R> bt) {
    constexpr int NREG = SP / 16;
    constexpr int UB = 16;
    using R4 = typename Vec<R>::v4;
    __shared__ __attribute__((aligned(16))) R btile[2 * kTileFrames * SP];
    const int lane = threadIdx.x, task = lane >> 4, i = lane & 15;
    const int cs = task >> 1;
    const bool fwd = (task & 1) == 0;
    const int tileA = 2 * blockIdx.x;
    const bool hasB = tileA + 1 < bt.ntiles_total;
    const int tile = (cs == 1 && hasB) ? tileA + 1 : tileA;
    const bool valid = cs == 0 || hasB;
    const int rec = bt.tile_rec[tile];
    const RecDesc rd = bt.recs[rec];
    const int t0 = bt.tile_t0[tile];
    const int len = min(kTileFrames, rd.T - t0);
    // both chunks are contiguous in the frame-major arrays: stage them with one pass
    const int recA = bt.tile_rec[tileA];
    const RecDesc rdA = bt.recs[recA];
    const int t0A = bt.tile_t0[tileA];
    const int lenA = min(kTileFrames, rdA.T - t0A);
    int lenB = 0;
    if (hasB) {
        const int recB = bt.tile_rec[tileA + 1];
        lenB = min(kTileFrames, bt.recs[recB].T - bt.tile_t0[tileA + 1]);
        if (bt.state[recA].done && bt.state[recB].done) return;
    } else if (bt.state[recA].done) {
        return;
    }
    stage_to_lds<(2 * kTileFrames * SP / 4 + 63) / 64>(
        reinterpret_cast<R4*>(btile), reinterpret_cast<const R4*>(bt.bmat + (rdA.row0 + t0A) * SP),
        (lenA + lenB) * SP / 4, lane, 64);
    const R* bl = btile + (cs == 1 ? lenA * SP : 0);        // this task's rows in LDS
    const int lenmax = max(lenA, lenB);

    const bool chunk0 = (t0 == 0);
    const R lp = (R)rd.lp;
    R c[NREG], x[NREG];
    const R* __restrict__ bnd = (fwd ? bt.fbound : bt.gbound) + (long long)tile * SP + i * NREG;
    R ssum = 0;
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        const int j = i * NREG + r;
        c[r] = (j < rd.S) ? (R)((1.0 - rd.lp) * bt.pi[(long long)rec * SP + j] + 1e-8) : (R)0;
        x[r] = bnd[r];
        ssum += x[r];
    }
    ssum = allreduce_sum<16>(ssum);
    R* __restrict__ out = (fwd ? bt.ahat : bt.bhat) + rd.row0 * SP + i * NREG;
    R* __restrict__ dump = bt.dump + lane * NREG;             // absorbs the stores of idle steps
    {
        const R scl = (fwd && chunk0) ? (R)1 : fast_rcp(ssum) * (fwd ? (R)1 : (R)SP);
#pragma unroll
        for (int r = 0; r < NREG; ++r) x[r] *= scl;
        if (!fwd) {
            R* dst = valid ? out + (long long)(t0 + len - 1) * SP : dump;
#pragma unroll
            for (int r = 0; r < NREG; ++r) dst[r] = x[r];
        }
    }
    __syncthreads();
    // per-lane constants of the unified step  u = b*(k1*x + k0);  r = sum(w*u);  x = u*(m1/r) + m0
    R k0[NREG], w[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        k0[r] = fwd ? c[r] : (R)0;
        w[r] = fwd ? (R)1 : c[r];
    }
    const R k1 = fwd ? lp : (R)1, m1 = fwd ? (R)1 : lp, m0 = fwd ? (R)0 : (R)1;
    const bool plain_first = fwd && chunk0;                  // frame 0 of the recording: no transition
    R* __restrict__ sfw = bt.sfw + rd.row0 + t0;
#pragma unroll 1
    for (int blk = 0; blk < lenmax; blk += UB) {
        R bv[UB][NREG];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int st = blk + u;
            int lrow = fwd ? st : len - 1 - st;
            lrow = min(max(lrow, 0), kTileFrames - 1);
#pragma unroll
            for (int r = 0; r < NREG; ++r) bv[u][r] = bl[lrow * SP + i * NREG + r];
        }
        R keep = 0;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int st = blk + u;
            const bool special = (u == 0) && (blk == 0) && plain_first;
            R uu[NREG];
            R part = 0;
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const R pre = special ? x[r] : k1 * x[r] + k0[r];
                uu[r] = bv[u][r] * pre;
                part += w[r] * uu[r];
            }
            const R rs = allreduce_sum<16>(part);
            const R sc = m1 * fast_rcp(rs);
#pragma unroll
            for (int r = 0; r < NREG; ++r) x[r] = uu[r] * sc + m0;
            const int orow = fwd ? t0 + st : t0 + len - 2 - st;
            const bool ok = valid && st < len && orow >= t0;
            R* dst = ok ? out + (long long)orow * SP : dump;
#pragma unroll
            for (int r = 0; r < NREG; ++r) dst[r] = x[r];
            keep = (i == u) ? rs : keep;
        }
        // forward scales of these 16 frames, one per lane (log s_t is summed by post_kernel)
        if (fwd && valid && blk + i < len) sfw[blk + i] = keep;
    }
}

}  // namespace vbx
