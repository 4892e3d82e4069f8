// vbx_kernels.hpp -- HIP kernels of one VB iteration (reference: VBx/VBx.py:87-125).
//
// HBM layout of a batch (all recordings share the padded widths Sp = 16*NT and Dp):
//   rho    [sum_T][Dp]  R    x-vectors scaled by sqrt(Phi)           (VBx.py:89)
//   gamma  [sum_T][Sp]  R    responsibilities (rows sum to 1)        (VBx.py:99)
//   bmat   [sum_T][Sp]  R    exp(log_p - rowmax): shifted likelihoods
//   mrow   [sum_T]      R    rowmax of log_p without the G term
//   ahat   [sum_T][Sp]  R    normalised forward vectors
//   bhat   [sum_T][Sp]  R    scaled backward vectors
//   alpha, invL [n_rec][Sp][Dp] R   speaker posteriors                (VBx.py:95-96)
//   pi     [n_rec][Sp]  f64  speaker priors                          (VBx.py:104)
// Padded speakers (s >= S) carry gamma = b = 0 and never receive mass; padded feature
// dims (d >= D) carry rho = 0 and Phi = 0, which makes invL = 1, alpha = 0 and every
// contribution to the bias and the ELBO vanish.
#pragma once
#include "vbx_device.hpp"

namespace vbx {

struct RecDesc {
    long long row0;   // first frame row of this recording in the frame-major arrays
    long long rho_row0;   // first row of its rho: row0, or the rows of the recording it shares its x-vectors with (an Fa / Fb
                          // sweep over one recording keeps ONE rho: vbx_batch_set_recording_shared)
    int T, S;         // frames, speakers (unpadded)
    int tile0;        // first workgroup tile
    int ntiles;       // tiles of kTileFrames frames
    int has_model;    // alpha/invL supplied by the caller: skip the first M-step (VBx.py:94)
    int sup0;         // first group operator of this recording (two- / three-level boundary walk)
    int sup20;        // first level-2 group operator of this recording (three-level walk)
    int rho_tile0;    // first tile of its rho in the fragment-ordered f16 copies (split GEMMs): tile0, or the owner's
    int rho_rec;      // the recording that owns that rho (index of its scale exponent, BatchView::rho_e)
    double lp, Fa, Fb;
    double gsum;      // sum_t G_t (VBx.py:87)
};

struct RecState {
    double tll;        // total forward log-likelihood without the G term (VBx.py:173)
    double elbo_prev;
    int n_iters;
    int done;
    int warned;
    int pad_;
};

struct LpPow { double mant; int fl; int pad_; };   // lp^n = mant * 2^fl, mant in [1, 2)  (host table, per recording and n)

template <typename R> struct BatchView {
    int n_rec, Sp, Dp, D, max_iters;
    int ntiles_total;
    const RecDesc* recs;
    RecState* state;       // [n_rec] the latest state (what every kernel but fin_kernel reads)
    RecState* state_out;   // fin_kernel only: where the iteration-finishing role writes the next state.  The state is
                           // double-buffered so that the two roles of one launch see ONE snapshot (`state`) whatever the
                           // order their workgroups run in; the host swaps the buffers after such a launch
    long long model_stride;   // elements between the two copies of alpha / invL (M-step k writes copy k & 1: a launch that
    int vec_stride;           // finishes iteration k-1 and starts iteration k must not clobber the model of a recording
                              // that turns out to have converged at k-1); vec_stride: the same for bias / emodel
    const int* tile_rec;   // [ntiles_total]
    const int* tile_t0;    // [ntiles_total]
    const int4* tile_desc; // [ntiles_total rounded up to 4] {recording, t0, frames, first row}; frames = 0 behind the last tile
    int* tile_done;        // [same] 1 once the tile's recording has converged (written by iter_fin)
    const int* tile_order; // null, or [blocks of the per-chunk kernels] workgroup -> tile (-1: none): recordings that share a
                           // rho are dealt so that the chunks reading one rho tile run side by side on one XCD (one L2)
    const double* phi;     // [n_rec][Dp]
    R* rho;
    R* gamma;
    R* bmat;
    R* mrow;
    R* ahat;
    R* bhat;
    R* alpha;              // [n_rec][Sp][Dp]
    R* invL;
    R* bias;               // [n_rec][Sp]   -0.5 * sum_d (invL + alpha^2) Phi
    R* bias_lo;            // [n_rec][Sp]   what rounding that f64 sum to R lost (fp32 paths; zero in fp64).  Round 6: a speaker's
                           // bias enters EVERY frame's log-likelihood, so its rounding error (2^-24 |bias|) is not noise that
                           // averages out over T but a systematic shift of the speaker's mass -- 20 x every other f32 rounding
                           // of an iteration together at the iterations where the EM map amplifies (tests/test_gpu_trajectory.py)
    double* emodel;        // [n_rec][Sp]   sum_d (log invL - invL - alpha^2 + 1)
    double* pi;            // [n_rec][Sp]
    const double* ip;      // [n_rec][Sp]   initial-state probabilities; aliases pi in VBx() (VBx.py:99)
    R* fw_scale;           // [sum_T] or null: s_t = sum(a_t)   (only the step-level API asks for these)
    R* bw_scale;           // [sum_T] or null: q_t = sum_j c_j e_{t+1}[j]
    R* mpart;              // [ntiles_total][Sp][Dp]   gamma^T rho per tile
    R* npart;              // [ntiles_total][Sp]       sum_t gamma per tile
    double* epart;         // [ntiles_total][Sp]       "entered" statistic per tile (VBx.py:101-103)
    R* gamma0;             // [n_rec][Sp] or null: responsibilities of frame 0 (fused path: gamma itself is written once,
                           // after the last iteration, vbx_chunk_post.hpp)
    double* pi_prev;       // [n_rec][Sp] the priors the last forward-backward pass ran with (gamma replay)
    double* Li;            // [n_rec][max_iters]
    double epsilon;
    // chunked scan (VBX_FB_CHUNKED): one chunk = one tile of kTileFrames frames
    R* op;                 // [ntiles_total][Sp][Sp]  forward transfer-operator columns (backward = transpose)
    int* opexp;            // [ntiles_total][Sp]      power-of-two exponent of every column
    R* oph;                // [ntiles_total][2][Sp][Sp] or null: the operators of the two halves of a tile (fused path:
                           //                     chunk_loglik -> chunk_post, which re-runs the halves side by side)
    int* ophexp;           // [ntiles_total][2][Sp]
    R* cop;                // [n_rec][Sp]  the operator recursion's c: ((1-lp) pi + 1e-8) / lp (plain for lp < 2^-20), written
                           //              by mstep_fin from the priors every iteration (f64 arithmetic, rounded once)
    const LpPow* lppow;    // [n_rec][kTileFrames + 1]  lp^n as mantissa and exponent (host: log2 / exp2 in f64)
    R* fbound;             // [ntiles_total][Sp]  forward vector entering the chunk (ahat[t0-1], any scale)
    R* gbound;             // [ntiles_total][Sp]  backward vector at the chunk's last frame (any scale)
    double* tllpart;       // [ntiles_total] or null: sum over the chunk of log s_t + m_t
    R* sfw;                // [sum_T] forward scales s_t = sum(a_t) written by scan3
    R* dump;               // [256] scratch that absorbs the stores of idle scan steps
    // two-level boundary walk (long recordings): operators of groups of `sgroup` consecutive chunks
    R* sop;                // [nsup_total][Sp][Sp]
    int* sopexp;           // [nsup_total][Sp]
    const int* sup_rec;    // [nsup_total] recording of a group
    const int* sup_idx;    // [nsup_total] index of the group within its recording
    int sgroup, nsup_total;
    // third level (very long recordings): operators of groups of `sgroup2` consecutive level-1 groups
    R* sop2;               // [nsup2_total][Sp][Sp]
    int* sopexp2;          // [nsup2_total][Sp]
    const int* sup2_rec;   // [nsup2_total] recording of a level-2 group
    const int* sup2_idx;   // [nsup2_total] index of the level-2 group within its recording
    int sgroup2, nsup2_total;
    // scan chunks per tile: 1 = one transfer operator / boundary pair per tile of kTileFrames frames;
    // 2 = per half tile (kScanHalf frames), chunk index 2*tile + half -- the fused kernels use it to re-run the
    // two halves of a tile on separate waves (half the dependent chain).  op / opexp / fbound / gbound always
    // have room for two chunks per tile.
    int spt;
    // split GEMMs (vbx_split.hpp; fp32 batches with VBX_OPT_GEMM = split): rho as f16 pairs in MFMA fragment order, the
    // model's alpha likewise (fin_kernel), and their power-of-two scales.  All null when the mode is off.
    const _Float16* rho_a;     // [tiles][kTileFrames x Dp x 2]  A operand of rho alpha^T   (chunk_loglik)
    const _Float16* rho_b;     // [tiles][kTileFrames x Dp x 2]  B operand of gamma^T rho   (chunk_post)
    const int* rho_e;          // [n_rec]  the copies hold rho 2^rho_e (indexed by RecDesc::rho_rec)
    _Float16* alpha_frag;      // [2][n_rec][Sp x Dp x 3]  alpha 2^alpha_e of the two model copies as THREE f16 terms, B operand of rho alpha^T
    int* alpha_e;              // [2][n_rec][Sp]
};

// the tile a workgroup of a per-chunk kernel works on (-1: none)
template <typename R> __device__ __forceinline__ int tile_of_block(const BatchView<R>& bt, int block) {
    return bt.tile_order ? bt.tile_order[block] : block;
}

constexpr int kScanHalf = kTileFrames / 2;
__device__ __forceinline__ int chunk_count(const RecDesc& rd, int spt) {
    return spt == 2 ? (rd.T + kScanHalf - 1) / kScanHalf : rd.ntiles;
}

// =======================================================================================
// prep: rho = X * sqrt(Phi), G_t = -0.5 (|x_t|^2 + D log 2pi)              VBx.py:87-89
// grid = tiles of the recording, block = 256.  One wavefront per frame row.
// =======================================================================================
template <typename R, typename XT>
__global__ __launch_bounds__(256) void prep_kernel(const XT* __restrict__ X, const double* __restrict__ sqrt_phi,
                                                   R* __restrict__ rho, double* __restrict__ gtile,
                                                   int T, int D, int Dp) {
    __shared__ double lds[16];
    // sqrt(Phi) may live in pinned HOST memory (the batch's argument block: no copy, no staging -- set_recording_impl): every
    // workgroup fetches it once, over the link, instead of once per element
    constexpr int kStage = 512;
    __shared__ double sphi[kStage];
    for (int d = threadIdx.x; d < min(Dp, kStage); d += 256) sphi[d] = sqrt_phi[d];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t0 = blockIdx.x * kTileFrames;
    double gacc = 0.0;
    for (int f = wave; f < kTileFrames; f += 4) {
        const int t = t0 + f;
        const bool ok = t < T;
        double ss = 0.0;
        for (int d = lane; d < Dp; d += 64) {
            const double x = (ok && d < D) ? (double)X[(long long)t * D + d] : 0.0;
            ss += x * x;
            if (ok) rho[(long long)t * Dp + d] = (R)(x * (d < kStage ? sphi[d] : sqrt_phi[d]));
        }
        ss = allreduce_sum<64>(ss);
        if (ok) gacc += -0.5 * (ss + (double)D * 1.8378770664093454836);   // log(2 pi)
    }
    const double tot = block_sum(lane == 0 ? gacc : 0.0, lds);
    if (threadIdx.x == 0) gtile[blockIdx.x] = tot;
}

// =======================================================================================
// M-step accumulation: C[s][d] = sum_t gamma[t][s] rho[t][d] for one tile of frames and one
// 32-column slab of d, on MFMA 16x16x4 (A = gamma^T fragment, B = rho fragment).      VBx.py:96
// grid = (ntiles_total, Dp/32, speaker blocks of 16 NT), block = 64 (one wavefront); more than 256 speakers: NT = 16 and
// several speaker blocks.
//   k-step = 4 frames: lane group g = lane>>4 supplies frame k0+g.
//   B fragment: lane (j,g) loads rho[k0+g][d0+2j .. d0+2j+1] as one 8/16-byte load; the two
//   values feed two N-tiles whose column j means d = d0 + 2j + {0,1} (free relabelling of N).
// =======================================================================================
template <typename R, int NT>
__global__ __launch_bounds__(64) void mstep_acc_kernel(BatchView<R> bt) {
    using M = Mfma16<R>;
    using acc_t = typename M::acc_t;
    using R2 = typename Vec<R>::v2;
    const int tile = blockIdx.x;
    const int rec = bt.tile_rec[tile];
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int Sp = bt.Sp, Dp = bt.Dp;
    const int t0 = bt.tile_t0[tile];
    const int tend = min(t0 + kTileFrames, rd.T);
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const int d0 = blockIdx.y * 32;
    const int s0 = blockIdx.z * 16 * NT;              // first speaker of this block
    const R* __restrict__ gam = bt.gamma + rd.row0 * Sp + s0;
    const R* __restrict__ rho = bt.rho + rd.rho_row0 * Dp;

    acc_t acc[NT][2];
    R nsum[NT];
#pragma unroll
    for (int mu = 0; mu < NT; ++mu) {
        acc[mu][0] = acc_t{0, 0, 0, 0};
        acc[mu][1] = acc_t{0, 0, 0, 0};
        nsum[mu] = 0;
    }
    // Issue every load of a half tile (16 k-steps = 64 frames) before the first MFMA: a wave that
    // consumes each fragment right after loading it pays one L2/fabric round trip per k-step.
    constexpr int kWords = NT * (int)(sizeof(R) / 4);                 // registers per k-step of gamma fragments
    constexpr int KB = kWords >= 32 ? 2 : kWords >= 16 ? 4 : kWords >= 8 ? 8 : 16;
#pragma unroll 1
    for (int kb0 = t0; kb0 < tend; kb0 += 4 * KB) {
        R2 bv[KB];
        R av[KB][NT];
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int t = kb0 + 4 * u + g;
            const bool ok = t < tend;
            bv[u] = R2{0, 0};
            if (ok) bv[u] = *reinterpret_cast<const R2*>(rho + (long long)t * Dp + d0 + 2 * i);
#pragma unroll
            for (int mu = 0; mu < NT; ++mu) av[u][mu] = ok ? gam[(long long)t * Sp + 16 * mu + i] : (R)0;
        }
#pragma unroll
        for (int u = 0; u < KB; ++u) {
#pragma unroll
            for (int mu = 0; mu < NT; ++mu) {
                nsum[mu] += av[u][mu];
                acc[mu][0] = M::mma(av[u][mu], bv[u].x, acc[mu][0]);
                acc[mu][1] = M::mma(av[u][mu], bv[u].y, acc[mu][1]);
            }
        }
    }
    R* __restrict__ part = bt.mpart + ((long long)tile * Sp + s0) * Dp;
#pragma unroll
    for (int mu = 0; mu < NT; ++mu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 16 * mu + M::row(lane, r);
            R2 v = R2{acc[mu][0][r], acc[mu][1][r]};
            *reinterpret_cast<R2*>(part + (long long)s * Dp + d0 + 2 * i) = v;
        }
    }
    if (blockIdx.y == 0) {
#pragma unroll
        for (int mu = 0; mu < NT; ++mu) {
            R v = nsum[mu];
            v = add_xor<16>(v);
            v = add_xor<32>(v);
            if (g == 0) bt.npart[(long long)tile * Sp + s0 + 16 * mu + i] = v;
        }
    }
}

// =======================================================================================
// fin_kernel: everything per recording that sits between two passes over the frames, in ONE launch.
//   role ITER  (blockIdx.y == Sp)  finishes iteration j = state.n_iters: ELBO (VBx.py:100), pi update (VBx.py:101-104),
//              history (VBx.py:105), convergence test (VBx.py:122-125), the c of the operator recursion from the new pi
//   role MSTEP (blockIdx.y = s)    starts iteration k: N_s, invL, alpha (VBx.py:95-96), the per-speaker bias of VBx.py:97
//              and the speaker's share of the ELBO model term (VBx.py:100); tile partials combined in f64
// mode 1: MSTEP only (first launch of a run: nothing to finish, k = n_iters); 2: ITER only (end of a run); 3: both
// (k = n_iters + 1).  Round 2 had two kernels, iter_fin at the end of an iteration and mstep_fin at the start of the
// next: two dependent launches of a few workgroups each, ~14 us of an iteration of ONE recording (55 us) and nothing the
// GPU could overlap with anything.  In one launch neither role may depend on what the other writes:
//   * both read the state snapshot `state`; ITER writes `state_out` (the host swaps the two afterwards);
//   * MSTEP of iteration k writes copy k & 1 of alpha / invL / bias / emodel, ITER reads emodel of iteration j from copy
//     j & 1, and whoever reads the model of the iteration in flight takes copy n_iters & 1 of the NEW state.  A recording
//     ITER finds converged keeps its model: the M-step this launch did for it went to the other copy;
//   * the recursion's c comes from the new pi: ITER writes it (MSTEP only in mode 1, from the pi it finds).
// grid = (n_rec, Sp + 1), block = 256, or 1024 for long recordings.
// =======================================================================================
template <typename R>
__global__ __launch_bounds__(1024) void fin_kernel(BatchView<R> bt, int mode) {
    __shared__ double lds[16];
    __shared__ double sh[1024];
    __shared__ int done_sh;
    __shared__ double arow[kSplitMaxDp];                    // split GEMMs: the speaker's alpha row, in full precision
    __shared__ float amax_sh[16];
    const int rec = blockIdx.x, Sp = bt.Sp, Dp = bt.Dp;
    const RecState st_in = bt.state[rec];
    const RecDesc rd = bt.recs[rec];
    const int u0 = rd.tile0, nu = rd.ntiles;
    if ((int)blockIdx.y == Sp) {
        // ---------------------------------------------------------------- ITER: finish iteration j = st.n_iters
        if (!(mode & 2)) return;
        RecState st = st_in;
        if (st.done) {                                     // frozen: the state just moves to the other buffer
            if (threadIdx.x == 0) bt.state_out[rec] = st;
            return;
        }
        const int j = threadIdx.x;
        // "entered" statistic: thread (slot, state) sums tiles slot, slot+nslot, ... ; Sp divides the block
        const int nslot = blockDim.x / Sp, slot = threadIdx.x / Sp, sj = threadIdx.x % Sp;
        double part = 0.0;
        for (int tl = slot; tl < nu; tl += 16 * nslot) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                v[u] = (tl + u * nslot < nu) ? bt.epart[(long long)(u0 + tl + u * nslot) * Sp + sj] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) part += v[u];
        }
        sh[threadIdx.x] = part;
        double tpart = 0.0;
        if (bt.tllpart)
            for (int tl = threadIdx.x; tl < nu; tl += blockDim.x) tpart += bt.tllpart[u0 + tl];
        __syncthreads();
        double pn = 0.0, em = 0.0, pj = 0.0;
        if (j < rd.S) {
            double ent = 0.0;
            for (int q = 0; q < nslot; ++q) ent += sh[q * Sp + j];
            pj = bt.pi[(long long)rec * Sp + j];
            const double g0 = bt.gamma0 ? (double)bt.gamma0[(long long)rec * Sp + j] : (double)bt.gamma[rd.row0 * Sp + j];
            pn = g0 + (1.0 - rd.lp) * pj * ent;
            em = bt.emodel[(long long)(st.n_iters & 1) * bt.vec_stride + (long long)rec * Sp + j];
        }
        const double tot = block_sum(pn, lds);
        const double emt = block_sum(em, lds);
        const double tll = bt.tllpart ? block_sum(tpart, lds) : st.tll;
        if (j < Sp) {
            const double pnew = pn / tot;
            bt.pi_prev[(long long)rec * Sp + j] = pj;       // what this iteration's forward-backward pass ran with
            bt.pi[(long long)rec * Sp + j] = pnew;
            if (bt.cop) {                                   // the c of the operator recursion (vbx_operator.hpp) for the next pass
                const double cj = j < rd.S ? (1.0 - rd.lp) * pnew + 1e-8 : 0.0;
                bt.cop[(long long)rec * Sp + j] = (R)(rd.lp >= 0x1p-20 ? cj / rd.lp : cj);
            }
        }
        if (j == 0) {
            const double elbo = tll + rd.Fa * rd.gsum + 0.5 * rd.Fb * emt;
            const int it = st.n_iters;
            if (it < bt.max_iters) bt.Li[(long long)rec * bt.max_iters + it] = elbo;
            if (it > 0 && elbo - st.elbo_prev < bt.epsilon) {
                st.done = 1;
                if (elbo - st.elbo_prev < 0) st.warned = 1;
            }
            st.tll = tll;
            st.elbo_prev = elbo;
            st.n_iters = it + 1;
            bt.state_out[rec] = st;
            done_sh = st.done;
        }
        __syncthreads();
        if (done_sh)
            for (int tl = threadIdx.x; tl < rd.ntiles; tl += blockDim.x) bt.tile_done[rd.tile0 + tl] = 1;
        return;
    }
    // -------------------------------------------------------------------- MSTEP: speaker s, iteration k
    if (!(mode & 1) || st_in.done) return;
    const int s = blockIdx.y;
    const int k = st_in.n_iters + ((mode & 2) ? 1 : 0);
    const bool given = (k == 0 && rd.has_model);
    if (mode == 1 && threadIdx.x == 0 && bt.cop) {
        const double cj = s < rd.S ? (1.0 - rd.lp) * bt.pi[(long long)rec * Sp + s] + 1e-8 : 0.0;
        bt.cop[(long long)rec * Sp + s] = (R)(rd.lp >= 0x1p-20 ? cj / rd.lp : cj);
    }
    const double fafb = rd.Fa / rd.Fb;
    double N = 0.0;
    // (a padded speaker, s >= S, has no mass: its partial sums are zero and are neither stored by chunk_post nor read here --
    //  at S = 50 in 64 columns that is 22 % of the largest stream of a long recording's iteration)
    const bool has_mass = s < rd.S;
    if (!given && has_mass) {
        double part = 0.0;
        for (int tl = threadIdx.x; tl < nu; tl += blockDim.x)
            part += (double)bt.npart[(long long)(u0 + tl) * Sp + s];
        N = block_sum(part, lds);
    }
    const long long sd = (long long)(k & 1) * bt.model_stride + ((long long)rec * Sp + s) * Dp;
    double bsum = 0.0, esum = 0.0;
    float amax = 0.0f;                                      // (split GEMMs: the largest |alpha| of this speaker)
    // thread = (feature d, slice of the tile range: 2 slices, 8 in a block of 1024 threads -- a recording of T = 200 000
    // has 1563 partials per speaker, twenty dependent rounds of loads on two slices); loads are clamped instead of
    // predicated so that all of a round are in flight per thread
    const int nsl = blockDim.x >> 7, slice = threadIdx.x >> 7, dl = threadIdx.x & 127;
    for (int d0 = 0; d0 < Dp; d0 += 128) {
        const int d = d0 + dl;
        const bool dok = d < Dp;
        double C = 0.0;
        if (!given && has_mass) {
            const int nt = nu, last = nt - 1;
            const R* __restrict__ mp = bt.mpart + ((long long)u0 * Sp + s) * Dp + (dok ? d : 0);
            const long long stride = (long long)Sp * Dp;
            const int per = (nt + nsl - 1) / nsl, lo = min(nt, slice * per), hi = min(nt, lo + per);
            constexpr int LB = sizeof(R) == 8 ? 20 : 40;       // loads in flight per thread: T = 10 000 in one round trip
            for (int tl = lo; tl < hi; tl += LB) {
                R v[LB];
#pragma unroll
                for (int u = 0; u < LB; ++u) v[u] = mp[(long long)min(tl + u, last) * stride];
#pragma unroll
                for (int u = 0; u < LB; ++u) C += (tl + u < hi) ? (double)v[u] : 0.0;
            }
        }
        sh[threadIdx.x] = C;
        __syncthreads();
        if (slice == 0 && dok) {
            const double phi = bt.phi[(long long)rec * Dp + d];
            double il, al, il_full = 0.0;
            if (given) {
                il = (double)bt.invL[sd + d];
                al = (double)bt.alpha[sd + d];
            } else {
                for (int q = 1; q < nsl; ++q) C += sh[threadIdx.x + 128 * q];
                il = 1.0 / (1.0 + fafb * N * phi);
                il_full = il;
                al = fafb * il * C;
                const R ilr = (R)il, alr = (R)al;      // the values every later kernel sees
                bt.invL[sd + d] = ilr;
                bt.alpha[sd + d] = alr;
                // alpha: what the product multiplies with (the model array's rounding is part of the iteration).  invL enters the
                // iteration through the bias only, so the bias takes it in full precision: its f32 rounding is a per-speaker
                // constant in every frame's log-likelihood -- 1.3e-6 of gamma after three iterations at T = 200 000 (round 6)
                al = (double)alr;
            }
            if (bt.alpha_frag) {
                // split GEMMs (round 6): the product runs on THREE f16 terms of the alpha computed here in f64 -- 33 bits -- not
                // on the f32 value of the model arrays, and the bias and the model term of the ELBO (second loop below) are taken
                // from the very value those terms add up to.  A speaker's alpha multiplies every frame of the recording, so its
                // representation error is not noise: at T = 200 000 two 11-bit terms of the f32 value were 1.9e-4 of gamma at the
                // iterations where the EM map amplifies, the f32 value itself 2.4e-5 (tests/test_gpu_trajectory.py)
                arow[d] = given ? al : fafb * il_full * C;
                amax = fmaxf(amax, fabsf((float)arow[d]));
            } else {
                bsum += (il + al * al) * phi;
                if (s < rd.S && d < bt.D) esum += log(il) - il - al * al + 1.0;
            }
        }
        __syncthreads();
    }
    if (bt.alpha_frag) {
        // alpha of this speaker as f16 pairs in the fragment order of chunk_loglik's B operand (vbx_split.hpp), scaled by
        // the power of two that puts its largest magnitude into [2^13, 2^14)
        amax = allreduce_max<64>(amax);
        if ((threadIdx.x & 63) == 0) amax_sh[threadIdx.x >> 6] = amax;
        __syncthreads();
        float m = 0.0f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, amax_sh[w]);
        const int e2 = split_exponent(m);
        _Float16* __restrict__ fr = bt.alpha_frag + ((long long)(k & 1) * bt.n_rec + rec) * Sp * Dp * 3;
        for (int d = threadIdx.x; d < Dp; d += blockDim.x) {
            const double x = __builtin_amdgcn_ldexp(arow[d], e2);
            const _Float16 hi = (_Float16)(float)x;
            const double r1 = x - (double)(float)hi;
            const _Float16 lo = (_Float16)(float)r1;
            const double r2 = r1 - (double)(float)lo;
            const _Float16 lo2 = (_Float16)(float)r2;
            fr[alpha_frag_offset(s, d, 0, Dp)] = hi;
            fr[alpha_frag_offset(s, d, 1, Dp)] = lo;
            fr[alpha_frag_offset(s, d, 2, Dp)] = lo2;
            // the alpha the product really uses, and with it the bias (VBx.py:97) and the model term of the ELBO (VBx.py:100)
            const double al = __builtin_amdgcn_ldexp(((double)(float)hi + (double)(float)lo) + (double)(float)lo2, -e2);
            const double phi = bt.phi[(long long)rec * Dp + d];
            const double il = given ? (double)bt.invL[sd + d] : 1.0 / (1.0 + fafb * N * phi);
            bsum += (il + al * al) * phi;
            if (s < rd.S && d < bt.D) esum += log(il) - il - al * al + 1.0;
        }
        if (threadIdx.x == 0) bt.alpha_e[(long long)(k & 1) * bt.vec_stride + (long long)rec * Sp + s] = e2;
    }
    bsum = block_sum(bsum, lds);
    esum = block_sum(esum, lds);
    if (threadIdx.x == 0) {
        const long long vi = (long long)(k & 1) * bt.vec_stride + (long long)rec * Sp + s;
        const double bv = -0.5 * bsum;
        const R bhi = (R)bv;
        bt.bias[vi] = bhi;
        bt.bias_lo[vi] = (R)(bv - (double)bhi);
        bt.emodel[vi] = esum;
    }
}

// =======================================================================================
// Per-frame speaker log-likelihoods on MFMA 16x16x4:                         VBx.py:97
//   l[t][s] = Fa * (rho_t . alpha_s + bias_s)      (the per-frame G term is left out: it
//   cancels in gamma and pi and enters the ELBO once as Fa * sum_t G_t)
// epilogue: m_t = max_s l, b = exp(l - m_t) -> bmat, mrow.
// grid = (ntiles_total, speaker blocks of 16 NT), block = 256: wave w owns frames [t0+32w, t0+32w+32) = 2 M-tiles.  With
// more than one speaker block (S > 256) the row maximum is not known here: the kernel leaves l itself in bmat and
// rownorm_kernel turns it into m_t and b.
//   K is relabelled so that lane group g supplies k = 16q + 4g + r for MFMA r of block q:
//   one 16-byte load per lane feeds four MFMAs, for A (rho rows) and B (alpha rows) alike.
// =======================================================================================
template <typename R, int NT>
__global__ __launch_bounds__(256) void loglik_kernel(BatchView<R> bt, R* __restrict__ lraw) {
    using M = Mfma16<R>;
    using acc_t = typename M::acc_t;
    using R4 = typename Vec<R>::v4;
    const int tile = blockIdx.x;
    const int rec = bt.tile_rec[tile];
    const RecState st = bt.state[rec];
    if (st.done) return;
    const int par = st.n_iters & 1;                    // the model of the iteration in flight (fin_kernel)
    const RecDesc rd = bt.recs[rec];
    const int Sp = bt.Sp, Dp = bt.Dp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int f0 = bt.tile_t0[tile] + 32 * wave;
    const int s0 = blockIdx.y * 16 * NT;              // first speaker of this block
    const bool tiled = gridDim.y > 1;
    const R* __restrict__ rho = bt.rho + rd.rho_row0 * Dp;
    const R* __restrict__ alpha = bt.alpha + (long long)par * bt.model_stride + ((long long)rec * Sp + s0) * Dp;

    acc_t acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = acc_t{0, 0, 0, 0};
    const int rowA0 = f0 + i, rowA1 = f0 + 16 + i;
    const bool ok0 = rowA0 < rd.T, ok1 = rowA1 < rd.T;
    constexpr int QB = sizeof(R) == 8 ? (NT >= 8 ? 2 : 4) : 8;   // K blocks of 16 loaded together
    constexpr int NB = NT < 2 ? NT : 2;     // speaker tiles whose alpha fragments are in flight together
    const int nq = Dp / 16;
#pragma unroll 1
    for (int q0 = 0; q0 < nq; q0 += QB) {
        R4 a0[QB], a1[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int kk = 16 * (q0 + u) + 4 * g;
            a0[u] = R4{0, 0, 0, 0};
            a1[u] = R4{0, 0, 0, 0};
            if (q0 + u < nq) {
                if (ok0) a0[u] = *reinterpret_cast<const R4*>(rho + (long long)rowA0 * Dp + kk);
                if (ok1) a1[u] = *reinterpret_cast<const R4*>(rho + (long long)rowA1 * Dp + kk);
            }
        }
#pragma unroll
        for (int n0 = 0; n0 < NT; n0 += NB) {
            R4 bfr[NB][QB];
#pragma unroll
            for (int nn = 0; nn < NB; ++nn)
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    const int kk = 16 * (q0 + u) + 4 * g;
                    bfr[nn][u] = R4{0, 0, 0, 0};
                    if (q0 + u < nq)
                        bfr[nn][u] = *reinterpret_cast<const R4*>(alpha + (long long)(16 * (n0 + nn) + i) * Dp + kk);
                }
#pragma unroll
            for (int nn = 0; nn < NB; ++nn)
#pragma unroll
                for (int u = 0; u < QB; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[0][n0 + nn] = M::mma(a0[u][r], bfr[nn][u][r], acc[0][n0 + nn]);
                        acc[1][n0 + nn] = M::mma(a1[u][r], bfr[nn][u][r], acc[1][n0 + nn]);
                    }
        }
    }
    const R Fa = (R)rd.Fa;
    R biasv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) biasv[n] = bt.bias[(long long)par * bt.vec_stride + (long long)rec * Sp + s0 + 16 * n + i];
    R falo[NT];                                        // Fa * (the part of the bias its R rounding lost): BatchView::bias_lo
#pragma unroll
    for (int n = 0; n < NT; ++n) falo[n] = Fa * bt.bias_lo[(long long)par * bt.vec_stride + (long long)rec * Sp + s0 + 16 * n + i];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int frame = f0 + 16 * m + M::row(lane, r);
            R v[NT];
            R mx = neg_inf<R>();
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int s = s0 + 16 * n + i;
                v[n] = (s < rd.S) ? Fa * (acc[m][n][r] + biasv[n]) + falo[n] : neg_inf<R>();
                mx = vmax(mx, v[n]);
            }
            mx = allreduce_max<16>(mx);
            if (frame < rd.T) {
                const long long cell = (rd.row0 + frame) * Sp + s0;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    bt.bmat[cell + 16 * n + i] = tiled ? v[n] : exp_r(v[n] - mx);
                    if (lraw) lraw[cell + 16 * n + i] = v[n];
                }
                if (!tiled && i == 0) bt.mrow[rd.row0 + frame] = mx;
            }
        }
    }
}

// More than 256 speakers: m_t = max_s l, b = exp(l - m_t) over the rows loglik_kernel left in bmat (VBx.py:97 epilogue).
// grid = ntiles_total, block = 256: one wavefront per frame row.
template <typename R>
__global__ __launch_bounds__(256) void rownorm_kernel(BatchView<R> bt) {
    const int tile = blockIdx.x;
    const int rec = bt.tile_rec[tile];
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int Sp = bt.Sp, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t0 = bt.tile_t0[tile], tend = min(t0 + kTileFrames, rd.T);
    for (int f = t0 + wave; f < tend; f += 4) {
        R* __restrict__ row = bt.bmat + (rd.row0 + f) * Sp;
        R mx = neg_inf<R>();
        for (int s = lane; s < Sp; s += 64) mx = vmax(mx, row[s]);
        mx = allreduce_max<64>(mx);
        for (int s = lane; s < Sp; s += 64) row[s] = exp_r(row[s] - mx);
        if (lane == 0) bt.mrow[rd.row0 + f] = mx;
    }
}

// =======================================================================================
// Sequential forward-backward (VBX_FB_SEQUENTIAL): wave 0 walks forward, wave 1 backward,
// lane = speaker (NREG registers per lane when Sp > 64).                VBx.py:146-175
// With c_j = (1-lp) pi_j + 1e-8 the matrix (tr + 1e-8) of VBx.py:158 is lp*I + 1 c^T, so
//   a_t  = b_t * (lp * ahat_{t-1} + c)            ahat = a / sum(a)          (VBx.py:167-168)
//   be_t = (lp * e + q) / q,  e = b_{t+1} * be_{t+1},  q = sum_j c_j e_j     (VBx.py:170-171)
// One cross-lane reduction per step.  tll = sum_t log sum(a_t) + sum_t m_t (VBx.py:173).
// grid = n_rec, block = 128.
// =======================================================================================
template <typename R, int NREG, int U>
struct RowBlock {
    R v[U][NREG];
};

template <typename R, int NREG>
__global__ __launch_bounds__(128) void fb_seq_kernel(BatchView<R> bt) {
    constexpr int U = NREG <= 4 ? 8 : NREG <= 8 ? 4 : 2;      // rows in flight per block: 2 x U x NREG registers
    const int rec = blockIdx.x;
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int Sp = bt.Sp, T = rd.T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const R lp = (R)rd.lp;
    const R* __restrict__ B = bt.bmat + rd.row0 * Sp;
    R c[NREG], p0[NREG];
    bool act[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        const int j = lane + 64 * r;
        act[r] = j < Sp;
        const double pj = (j < rd.S) ? bt.pi[(long long)rec * Sp + j] : 0.0;
        c[r] = (j < rd.S) ? (R)((1.0 - rd.lp) * pj + 1e-8) : (R)0;
        p0[r] = (j < rd.S) ? (R)(bt.ip[(long long)rec * Sp + j] + 1e-8) : (R)0;
    }
    auto load_rows = [&](RowBlock<R, NREG, U>& blk, int tfirst, int dir) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tfirst + dir * u;
#pragma unroll
            for (int r = 0; r < NREG; ++r)
                blk.v[u][r] = (t >= 0 && t < T && act[r]) ? B[(long long)t * Sp + lane + 64 * r] : (R)0;
        }
    };

    if (wave == 0) {
        // ------------------------------------------------------------------ forward
        R* __restrict__ A = bt.ahat + rd.row0 * Sp;
        R ah[NREG];
#pragma unroll
        for (int r = 0; r < NREG; ++r) ah[r] = 0;
        ScaledProduct sp;
        auto run = [&](const RowBlock<R, NREG, U>& blk, int tfirst) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = tfirst + u;
                if (t >= T) break;        // wave-uniform
                R a[NREG];
                R s = 0;
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    a[r] = (t == 0) ? blk.v[u][r] * p0[r] : blk.v[u][r] * (lp * ah[r] + c[r]);
                    s += a[r];
                }
                s = allreduce_sum<64>(s);
                const R inv = fast_rcp(s);
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    ah[r] = a[r] * inv;
                    if (act[r]) A[(long long)t * Sp + lane + 64 * r] = ah[r];
                }
                sp.mul((double)s);
                if ((t & 15) == 15) sp.renorm();
                if (bt.fw_scale && lane == 0) bt.fw_scale[rd.row0 + t] = s;
            }
        };
        RowBlock<R, NREG, U> b0, b1;
        load_rows(b0, 0, 1);
        for (int tb = 0; tb < T; tb += 2 * U) {
            load_rows(b1, tb + U, 1);
            run(b0, tb);
            load_rows(b0, tb + 2 * U, 1);
            run(b1, tb + U);
        }
        double msum = 0.0;
        for (int t = lane; t < T; t += 64) msum += (double)bt.mrow[rd.row0 + t];
        msum = allreduce_sum<64>(msum);
        sp.renorm();
        if (lane == 0) bt.state[rec].tll = sp.log_value() + msum;
    } else {
        // ------------------------------------------------------------------ backward
        R* __restrict__ Bh = bt.bhat + rd.row0 * Sp;
        R bh[NREG];
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            bh[r] = 1;
            if (act[r]) Bh[(long long)(T - 1) * Sp + lane + 64 * r] = bh[r];
        }
        // block rows are b[t+1] for t = tfirst, tfirst-1, ...
        auto run = [&](const RowBlock<R, NREG, U>& blk, int tfirst) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = tfirst - u;
                if (t < 0) break;         // wave-uniform
                R e[NREG];
                R q = 0;
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    e[r] = blk.v[u][r] * bh[r];
                    q += c[r] * e[r];
                }
                q = allreduce_sum<64>(q);
                const R sc = lp * fast_rcp(q);
                if (bt.bw_scale && lane == 0) bt.bw_scale[rd.row0 + t] = q;
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    bh[r] = sc * e[r] + (R)1;
                    if (act[r]) Bh[(long long)t * Sp + lane + 64 * r] = bh[r];
                }
            }
        };
        RowBlock<R, NREG, U> b0, b1;
        load_rows(b0, T - 1, -1);            // rows T-1, T-2, ... serve t = T-2, T-3, ...
        for (int tb = T - 2; tb >= 0; tb -= 2 * U) {
            load_rows(b1, tb + 1 - U, -1);
            run(b0, tb);
            load_rows(b0, tb + 1 - 2 * U, -1);
            run(b1, tb - U);
        }
    }
}

// =======================================================================================
// Posteriors and prior statistics from ahat/bhat:                 VBx.py:101-103, 174
//   gamma_t = ahat_t * bhat_t / sum(ahat_t * bhat_t)
//   entered_j += gamma_t[j] / (lp * ahat_{t-1}[j] + c_j)   for t >= 1   
// grid = ntiles_total, block = 256.  A frame occupies W = min(Sp, 64) adjacent lanes.
// =======================================================================================
template <typename R, int SP>
__global__ __launch_bounds__(256) void post_kernel(BatchView<R> bt) {
    constexpr int W = SP < 64 ? SP : 64;
    constexpr int FPW = 64 / W;        // frames per wavefront pass
    constexpr int NREG = SP / W;
    __shared__ double ent_lds[SP];
    const int tile = blockIdx.x;
    const int rec = bt.tile_rec[tile];
    if (bt.state[rec].done) return;
    const RecDesc rd = bt.recs[rec];
    const int t0 = bt.tile_t0[tile];
    const int tend = min(t0 + kTileFrames, rd.T);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = lane / W, jl = lane % W;
    const R lp = (R)rd.lp;
    const R* __restrict__ A = bt.ahat + rd.row0 * SP;
    const R* __restrict__ Bh = bt.bhat + rd.row0 * SP;
    R* __restrict__ G = bt.gamma + rd.row0 * SP;
    for (int j = threadIdx.x; j < SP; j += 256) ent_lds[j] = 0.0;
    __syncthreads();
    R c[NREG];
    double ent[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        const int j = jl + W * r;
        c[r] = (j < rd.S) ? (R)((1.0 - rd.lp) * bt.pi[(long long)rec * SP + j] + 1e-8) : (R)1;
        ent[r] = 0.0;
    }
    constexpr int kPasses = kTileFrames / (4 * FPW);                 // passes of this wavefront over the tile
    constexpr int PB = (kPasses * NREG <= 32) ? kPasses : 32 / NREG;   // passes whose loads are in flight together
    static_assert(kPasses % PB == 0, "pass batching");
#pragma unroll 1
    for (int p0 = 0; p0 < kPasses; p0 += PB) {
        R av[PB][NREG], bv[PB][NREG], apv[PB][NREG];
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int f = t0 + wave * FPW + (p0 + p) * 4 * FPW + sub;
            const bool ok = f < tend;
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const int j = jl + W * r;
                av[p][r] = ok ? A[(long long)f * SP + j] : (R)0;
                bv[p][r] = ok ? Bh[(long long)f * SP + j] : (R)0;
                apv[p][r] = (ok && f >= 1) ? A[(long long)(f - 1) * SP + j] : (R)0;
            }
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int f = t0 + wave * FPW + (p0 + p) * 4 * FPW + sub;
            const bool ok = f < tend;
            R g[NREG];
            R sum = 0;
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                g[r] = av[p][r] * bv[p][r];
                sum += g[r];
            }
            sum = allreduce_sum<W>(sum);
            const R inv = ok ? (R)1 / sum : (R)0;
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const int j = jl + W * r;
                const R gm = g[r] * inv;
                if (ok) G[(long long)f * SP + j] = gm;
                if (ok && f >= 1 && j < rd.S) ent[r] += (double)(gm / (lp * apv[p][r] + c[r]));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NREG; ++r) atomicAdd(&ent_lds[jl + W * r], ent[r]);
    __syncthreads();
    for (int j = threadIdx.x; j < SP; j += 256) bt.epart[(long long)tile * SP + j] = ent_lds[j];
    if (bt.tllpart) {      // chunked scan: this tile's share of the total log-likelihood (VBx.py:173)
        __shared__ double tl_lds[16];
        double part = 0.0;
        for (int f = t0 + threadIdx.x; f < tend; f += 256)
            part += log((double)bt.sfw[rd.row0 + f]) + (double)bt.mrow[rd.row0 + f];
        part = block_sum(part, tl_lds);
        if (threadIdx.x == 0) bt.tllpart[tile] = part;
    }
}

}  // namespace vbx
