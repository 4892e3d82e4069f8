// vbx_ahc.hpp -- score stage of the AHC initialisation that runs right before VBx() in vbhmm.py:135-138
// (SURVEY.md section 8f, rank 1): the T x T cosine-similarity matrix and the two-Gaussian calibration of
// its T*T entries.  Everything is float64, as in the reference (diarization_lib.py).
//
//   cos_similarity  (diarization_lib.py:190-213)   xn = x / (|x| + 1e-32);  C = xn xn^T   on v_mfma_f64_16x16x4
//   twoGMMcalib_lin (diarization_lib.py:13-31)     20 EM passes over the scores resident in HBM: one
//                                                  streaming kernel per pass (6 sums), parameters stay on the device
#pragma once
#include "vbx_device.hpp"

namespace vbx {

// rows of x scaled to unit length, feature dim padded with zeros to Dp (multiple of 16)
__global__ __launch_bounds__(256) void cos_norm_kernel(const double* __restrict__ x, double* __restrict__ xn,
                                                        long long T, int D, int Dp) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + wave;
    if (t >= T) return;
    double ss = 0.0;
    for (int d = lane; d < D; d += 64) {
        const double v = x[t * D + d];
        ss += v * v;
    }
    ss = allreduce_sum<64>(ss);
    const double inv = 1.0 / (sqrt(ss) + 1.0e-32);          // diarization_lib.py:201
    for (int d = lane; d < Dp; d += 64) xn[t * Dp + d] = d < D ? x[t * D + d] * inv : 0.0;
}

// C[i][j] = <xn_i, xn_j>.  grid = (ceil(T/64), ceil(T/64)), block = 256: wave w owns the 32 x 32 sub-tile
// (w>>1, w&1) = 2 x 2 MFMA tiles.  K is relabelled so that lane group g supplies k = 16q + 4g + r for MFMA r
// of block q: one 32-byte load per lane and operand feeds four MFMAs.  xn (T x Dp doubles) lives in L2.
__global__ __launch_bounds__(256) void cos_gemm_kernel(const double* __restrict__ xn, double* __restrict__ C,
                                                        long long T, int Dp) {
    using M = Mfma16<double>;
    using acc_t = M::acc_t;
    using D4 = Vec<double>::v4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const long long r0 = (long long)blockIdx.y * 64 + 32 * (wave >> 1), c0 = (long long)blockIdx.x * 64 + 32 * (wave & 1);
    acc_t acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = acc_t{0, 0, 0, 0};
    const double* __restrict__ pa[2];
    const double* __restrict__ pb[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        pa[m] = xn + min(r0 + 16 * m + i, T - 1) * Dp + 4 * g;     // rows past the end are clamped, never stored
        pb[m] = xn + min(c0 + 16 * m + i, T - 1) * Dp + 4 * g;
    }
    for (int q = 0; q < Dp; q += 16) {
        D4 a[2], b[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            a[m] = *reinterpret_cast<const D4*>(pa[m] + q);
            b[m] = *reinterpret_cast<const D4*>(pb[m] + q);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = M::mma(a[m][r], b[n][r], acc[m][n]);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long row = r0 + 16 * m + M::row(lane, r), col = c0 + 16 * n + i;
                if (row < T && col < T) C[row * T + col] = acc[m][n][r];
            }
}

// condensed (upper triangle, row by row: scipy.spatial.distance.squareform's vector form) copy of scale * S: what the
// clustering that follows the calibration consumes (vbhmm.py:139 squareform(-scr_mx)).  grid = T - 1 rows.
__global__ __launch_bounds__(256) void condense_kernel(const double* __restrict__ s, double* __restrict__ out, long long T,
                                                        double scale) {
    const long long i = blockIdx.x;
    const double* __restrict__ row = s + i * T;
    double* __restrict__ dst = out + i * (2 * T - i - 1) / 2 - (i + 1);      // dst[j] = element (i, j), j > i
    for (long long j = i + 1 + threadIdx.x; j < T; j += 256) dst[j] = scale * row[j];
}

// ---- two-Gaussian calibration ---------------------------------------------------------------------
// parameter block on the device: [0,1] weights  [2,3] means  [4] var  [5..9] the same before the last update
constexpr int kGmmPartials = 2048;         // workgroups of a streaming pass (fixed: deterministic summation order)

// pass 0: sum s;  pass 1: sum (s - mean)^2          (np.mean, np.std / np.var: diarization_lib.py:20-21)
template <int PASS>
__global__ __launch_bounds__(256) void gmm_moment_kernel(const double* __restrict__ s, long long n, const double* par,
                                                          double* __restrict__ part) {
    __shared__ double lds[16];
    const double mean = PASS == 1 ? par[10] : 0.0;
    double acc = 0.0;
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256) {
        const double v = s[k] - mean;
        acc += PASS == 1 ? v * v : v;
    }
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void gmm_init_kernel(const double* __restrict__ part, int npart, long long n, double* par,
                                                        int pass) {
    __shared__ double lds[16];
    double acc = 0.0;
    for (int k = threadIdx.x; k < npart; k += 256) acc += part[k];
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) {
        if (pass == 0) {
            par[10] = acc / (double)n;                      // mean
        } else {
            const double var = acc / (double)n, sd = sqrt(var);
            par[0] = par[1] = 0.5;                          // diarization_lib.py:19-21
            par[2] = par[10] - sd;
            par[3] = par[10] + sd;
            par[4] = var;
        }
    }
}

// one EM pass: responsibilities of the two Gaussians and the six sums of diarization_lib.py:24-29
__global__ __launch_bounds__(256) void gmm_pass_kernel(const double* __restrict__ s, long long n, const double* __restrict__ par,
                                                        double* __restrict__ part) {
    __shared__ double lds[16];
    const double lw0 = log(par[0]), lw1 = log(par[1]), m0 = par[2], m1 = par[3], var = par[4];
    const double hl = 0.5 * log(var), hv = 0.5 / var;
    double c0 = 0, c1 = 0, s0 = 0, s1 = 0, q0 = 0, q1 = 0;
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256) {
        const double v = s[k];
        const double l0 = lw0 - hl - (v - m0) * (v - m0) * hv, l1 = lw1 - hl - (v - m1) * (v - m1) * hv;
        // scipy.special.softmax over the two classes: exp(l - max) / sum.  The larger class is exp(0) = 1 exactly,
        // so one exponential gives the same bits as two.
        const double d = l1 - l0, em = exp(-fabs(d)), inv = 1.0 / (1.0 + em);
        const double g0 = d > 0.0 ? em * inv : inv, g1 = d > 0.0 ? inv : em * inv;
        c0 += g0;
        c1 += g1;
        s0 += v * g0;
        s1 += v * g1;
        q0 += v * v * g0;
        q1 += v * v * g1;
    }
    double* dst = part + (long long)blockIdx.x * 6;
    double v6[6] = {c0, c1, s0, s1, q0, q1};
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const double tot = block_sum(v6[e], lds);
        if (threadIdx.x == 0) dst[e] = tot;
    }
}

// M-step of the calibration (diarization_lib.py:26-29); keeps the parameters the pass started with
__global__ __launch_bounds__(256) void gmm_update_kernel(const double* __restrict__ part, int npart, double* par) {
    __shared__ double lds[16];
    double tot[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        double acc = 0.0;
        for (int k = threadIdx.x; k < npart; k += 256) acc += part[(long long)k * 6 + e];
        tot[e] = block_sum(acc, lds);
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int e = 0; e < 5; ++e) par[5 + e] = par[e];
        const double c0 = tot[0], c1 = tot[1];
        const double w0 = c0 / (c0 + c1), w1 = c1 / (c0 + c1);
        const double m0 = tot[2] / c0, m1 = tot[3] / c1;
        par[0] = w0;
        par[1] = w1;
        par[2] = m0;
        par[3] = m1;
        par[4] = (tot[4] / c0 - m0 * m0) * w0 + (tot[5] / c1 - m1 * m1) * w1;
    }
}

// calibrated log-odds: lls of the LAST pass (parameters par[5..9]), columns picked by the FINAL means
__global__ __launch_bounds__(256) void gmm_llr_kernel(const double* __restrict__ s, long long n, const double* __restrict__ par,
                                                       double* __restrict__ llr) {
    const double lw0 = log(par[5]), lw1 = log(par[6]), m0 = par[7], m1 = par[8], var = par[9];
    const double hl = 0.5 * log(var), hv = 0.5 / var;
    const bool hi1 = par[3] > par[2], lo1 = par[3] < par[2];      // means.argmax() / means.argmin() (first extremum wins)
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256) {
        const double v = s[k];
        const double l0 = lw0 - hl - (v - m0) * (v - m0) * hv, l1 = lw1 - hl - (v - m1) * (v - m1) * hv;
        llr[k] = (hi1 ? l1 : l0) - (lo1 ? l1 : l0);
    }
}

// ---- average linkage on the score matrix where it lies ------------------------------------------------------------
// vbhmm.py:139-141: fastcluster.linkage(squareform(-scr_mx), method='average').  The nearest-neighbour chain of
// vbx_linkage.hpp (SciPy's nn_chain, same operations in the same order -> the same merges to the last bit) walked by ONE
// persistent workgroup of 1024 threads on the T x T matrix in HBM: a chain step is a row scan (arg-min with SciPy's
// tie rules: the previous element of the chain, then the lowest index), a merge rewrites one row and one column.  No
// launch per step, nothing crosses PCIe but the T - 1 merges.  Several recordings = several workgroups on several
// streams.

// D = -S (distances), +inf on the diagonal; in place.
__global__ __launch_bounds__(256) void linkage_prepare_kernel(double* __restrict__ D, long long T) {
    const long long i = blockIdx.x;
    double* __restrict__ row = D + i * T;
    for (long long j = threadIdx.x; j < T; j += 256) row[j] = j == i ? (double)INFINITY : -row[j];
}

struct ChainMergeDev { int a, b; double d; };          // same layout as vbx::ChainMerge of the host code

// (value, index) arg-min over the 64 lanes of a wavefront, the lowest index winning ties; every lane gets the result.
// DPP row operations and permlane swaps (cheaper than __shfl_xor's ds_bpermute_b32 round trip: vbx_device.hpp, add_xor).
__device__ __forceinline__ void argmin_pair(double& m, int& mi, double ov, int oi) {
    if (ov < m || (ov == m && oi < mi)) { m = ov; mi = oi; }
}
__device__ __forceinline__ void argmin_allreduce64(double& m, int& mi) {
    argmin_pair(m, mi, dpp_mov<0xB1>(m), dpp_mov<0xB1>(mi));
    argmin_pair(m, mi, dpp_mov<0x4E>(m), dpp_mov<0x4E>(mi));
    argmin_pair(m, mi, dpp_mov<0x141>(m), dpp_mov<0x141>(mi));
    argmin_pair(m, mi, dpp_mov<0x140>(m), dpp_mov<0x140>(mi));
    {
        double ma, mb;
        int ia, ib;
        cross_rows<16>(m, ma, mb);
        cross_rows<16>(mi, ia, ib);
        m = ma; mi = ia;
        argmin_pair(m, mi, mb, ib);
        cross_rows<32>(m, ma, mb);
        cross_rows<32>(mi, ia, ib);
        m = ma; mi = ia;
        argmin_pair(m, mi, mb, ib);
    }
}

// (n_a d_a + n_b d_b) / (n_a + n_b) with every operation rounded on its own, as SciPy and the host code compute it
// (hipcc contracts a * b + c into a fused multiply-add by default, also through __dmul_rn / __dadd_rn)
__device__ __forceinline__ double average_update(double fa, double da, double fb, double db, double fs) {
#pragma clang fp contract(off)
    const double pa = fa * da;
    const double pb = fb * db;
    const double sum = pa + pb;
    return sum / fs;
}

// Staged (round 2): the chain is walked in stages of n/4 merges; between two stages the live rows and columns are
// copied into a matrix of the new dimension by a grid of workgroups (linkage_compact_* below), so a row scan reads
// what is alive and little else -- beyond its four dependent round trips a merge costs one CU's bandwidth on the ~110
// bytes per entry it touches, which shows once the matrix no longer fits the 256 MB Infinity Cache (T > ~5600).  Indices
// inside a stage are positions in the compacted matrix (`orig` maps them back when the stage ends; compaction keeps the
// order, so SciPy's tie rules -- previous chain element, then the lowest index -- pick the same clusters); the chain
// itself, the cluster sizes and (chain_length, first_live) live in global memory across stages.
// Tried on top and dropped: the top of the chain mirrored in LDS with sizes and distances (a pop without global loads,
// sizes carried through the arg-min reduction) -- 1 us per merge SLOWER (14.0 -> 15.0 us at T = 3000): the extra
// shuffle stage of every scan costs more than the three loads of a merge, which overlap with the row update.
__global__ __launch_bounds__(1024) void nn_chain_kernel(double* __restrict__ D, int n, int* __restrict__ size,
                                                         int* __restrict__ chain, const int* __restrict__ orig,
                                                         int* __restrict__ state, ChainMergeDev* __restrict__ merges,
                                                         int k_begin, int k_end) {
    __shared__ double wmin[16];
    __shared__ int widx[16];
    __shared__ int s_x, s_pred, s_merge, s_a, s_b, s_na, s_nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double inf = (double)INFINITY;
    // thread 0 keeps the chain: its length, the two top elements and their distance stay in registers (a push knows
    // them: d(y, x) = d(x, y), the matrix is kept symmetric), the rest lives in `chain`; after a merge the new top pair
    // is read back -- a chain step itself costs the row fetch and two barriers, no other dependent memory access
    int chain_length = 0, first_live = 0, top = 0, pred = -1;
    double top_pred_dist = inf;
    if (tid == 0) {                                      // where the previous stage (or the initialisation) left the chain
        chain_length = state[0];
        first_live = state[1];
        if (chain_length >= 1) {
            top = chain[chain_length - 1];
            pred = chain_length > 1 ? chain[chain_length - 2] : -1;
            top_pred_dist = pred >= 0 ? D[(long long)top * n + pred] : inf;
        }
    }
    __syncthreads();
    for (int k = k_begin; k < k_end; ++k) {
        if (tid == 0) {
            if (chain_length == 0) {
                while (size[first_live] == 0) ++first_live;  // the lowest live index starts a chain
                chain[0] = first_live;
                chain_length = 1;
                top = first_live;
                pred = -1;
                top_pred_dist = inf;
            }
            s_x = top;
        }
        while (true) {                                   // go down the chain
            __syncthreads();
            const int x = s_x;
            const double* __restrict__ row = D + (long long)x * n;
            // first index of the minimum over the live clusters (`dist < current_min` of a scan in index order)
            // (every load of a round is issued before the first compare: a dependent load per entry made a step
            //  cost as many L2 round trips as a thread has entries -- 10 us at T = 10 000)
            double m = inf;
            int mi = 0x7fffffff;
            constexpr int U = 8;
            for (int i0 = tid; i0 < n; i0 += U * 1024) {
                double v[U];
                int live[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = min(i0 + u * 1024, n - 1);
                    v[u] = row[i];
                    live[u] = size[i];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * 1024;
                    const double w = (i < n && live[u] > 0) ? v[u] : inf;     // (the diagonal holds +inf)
                    if (w < m) { m = w; mi = i; }
                }
            }
            argmin_allreduce64(m, mi);
            if (lane == 0) { wmin[wave] = m; widx[wave] = mi; }
            __syncthreads();
            if (tid == 0) {
#pragma unroll
                for (int w = 1; w < 16; ++w)
                    if (wmin[w] < m || (wmin[w] == m && widx[w] < mi)) { m = wmin[w]; mi = widx[w]; }
                double cur = top_pred_dist;                          // the previous element wins ties
                int y = pred;
                if (m < cur) { cur = m; y = mi; }
                if (chain_length > 1 && y == pred) {                 // x and y are reciprocal nearest neighbours
                    chain_length -= 2;
                    const int a = x < y ? x : y, b = x < y ? y : x;
                    s_a = a; s_b = b; s_na = size[a]; s_nb = size[b];
                    merges[k] = ChainMergeDev{a, b, cur};            // (positions: translated when the stage ends)
                    s_merge = 1;
                } else {
                    chain[chain_length++] = y;
                    pred = x;
                    top = y;
                    top_pred_dist = cur;
                    s_x = y;
                    s_merge = 0;
                }
            }
            __syncthreads();
            if (s_merge) break;
        }
        // a is dropped, b becomes the merged cluster: d(i, a u b) = (n_a d(i,a) + n_b d(i,b)) / (n_a + n_b), every
        // operation rounded on its own like the host code (no fused multiply-add)
        const int a = s_a, b = s_b;
        const double fa = (double)s_na, fb = (double)s_nb, fs = (double)(s_na + s_nb);
        const double* __restrict__ ra = D + (long long)a * n;
        double* __restrict__ rb = D + (long long)b * n;
        {
            constexpr int U = 8;
            for (int i0 = tid; i0 < n; i0 += U * 1024) {
                double va[U], vb[U];
                int live[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = min(i0 + u * 1024, n - 1);
                    va[u] = ra[i];
                    vb[u] = rb[i];
                    live[u] = size[i];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * 1024;
                    if (i >= n || i == a || i == b || live[u] == 0) continue;
                    const double v = average_update(fa, va[u], fb, vb[u], fs);
                    rb[i] = v;
                    D[(long long)i * n + b] = v;
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            size[a] = 0;
            size[b] = s_na + s_nb;
            if (chain_length >= 1) {                     // the pair on top of what is left of the chain
                top = chain[chain_length - 1];
                pred = chain_length > 1 ? chain[chain_length - 2] : -1;
                top_pred_dist = pred >= 0 ? D[(long long)top * n + pred] : inf;
            }
        }
    }
    __syncthreads();
    for (int k = k_begin + tid; k < k_end; k += 1024) {  // positions in this stage's matrix -> the clusters they stand for
        ChainMergeDev mg = merges[k];
        mg.a = orig[mg.a];
        mg.b = orig[mg.b];
        merges[k] = mg;
    }
    if (tid == 0) {
        state[0] = chain_length;
        state[1] = first_live;
    }
}

// ---- between two stages: the live clusters move up, in order ----------------------------------------------------------
// One workgroup: position of every live cluster among the live ones (block-wide prefix sum), the lists that follow the
// clusters (size, original id) and the chain re-indexed; `old_of_new` tells the matrix copy where a row / column comes from.
__global__ __launch_bounds__(1024) void linkage_compact_index_kernel(int n, const int* __restrict__ size, const int* __restrict__ orig,
                                                                      int* __restrict__ chain, int* __restrict__ state,
                                                                      int* __restrict__ size2, int* __restrict__ orig2,
                                                                      int* __restrict__ old_of_new, int* __restrict__ newidx) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int live = (i < n && size[i] > 0) ? 1 : 0;
        // inclusive prefix count inside the wave from the ballot (no shuffles: vbx_device.hpp, add_xor)
        const int incl = __popcll(__ballot(live) & ((2ull << lane) - 1ull));
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        const int pos = before + incl - live;
        if (i < n) newidx[i] = live ? pos : -1;
        if (live) {
            old_of_new[pos] = i;
            size2[pos] = size[i];
            orig2[pos] = orig[i];
        }
        __syncthreads();
        if (tid == 1023) carry = before + incl;
        __syncthreads();
    }
    const int len = state[0];
    for (int j = tid; j < len; j += 1024) chain[j] = newidx[chain[j]];      // (chain elements are alive)
    if (tid == 0) state[1] = 0;                                              // every position is alive again
}

// grid = new dimension (one row each): D2[r][c] = D1[old(r)][old(c)]
__global__ __launch_bounds__(256) void linkage_compact_matrix_kernel(const double* __restrict__ D1, int n1, double* __restrict__ D2,
                                                                      int n2, const int* __restrict__ old_of_new) {
    const int r = blockIdx.x;
    const double* __restrict__ src = D1 + (long long)old_of_new[r] * n1;
    double* __restrict__ dst = D2 + (long long)r * n2;
    for (int c = threadIdx.x; c < n2; c += 256) dst[c] = src[old_of_new[c]];
}

// ---- rounds of reciprocal nearest neighbours: the whole chip on one recording ---------------------------------------------
// Average linkage is reducible: two clusters that are each other's nearest neighbour stay so whatever else is merged, so
// ALL reciprocal pairs of the current matrix can be merged at once (the nearest-neighbour chain finds the same pairs one
// after the other).  A round =
//   rnn_rowmin   one workgroup per live row: nearest live neighbour, the lowest index among equals
//   rnn_pairs    one workgroup: the reciprocal pairs (a < b, nn[a] == b, nn[b] == a) in index order -> the merge list;
//                with lowest-index ties the globally smallest pair is always reciprocal, so every round merges something
//   rnn_rows     one workgroup per pair: row b <- (n_a row a + n_b row b) / (n_a + n_b)   (b keeps the merged cluster)
//   rnn_cols     one workgroup per live row x, merged rows included: D[x][b] <- (n_a D[x][a] + n_b D[x][b]) / (n_a + n_b) for
//                every pair -- the same operations on the same numbers as row b got, so row b and column b agree bit for
//                bit wherever x is not itself a merged row; between two merged rows the two orders of association differ
//                in the last bits, and
//   rnn_canon    makes the row of the smaller index the one value of both entries (the matrix stays exactly symmetric:
//                the nearest-neighbour relation the next round reads must be the same from both sides)
//   rnn_sizes    a dies, b takes both sizes.
// Same update formula, operation for operation, as the chain (average_update); only the ORDER in which a cluster's merges
// meet differs from the chain's, i.e. heights agree to a few ulp and the tree is the same wherever distances are distinct
// (tests/test_driver.py).  One recording of 10 000 x-vectors: a few dozen rounds instead of 30 000 dependent row scans.
struct RnnPair { int a, b, na, nb; };

__global__ __launch_bounds__(256) void rnn_rowmin_kernel(const double* __restrict__ D, int n, const int* __restrict__ size,
                                                          int* __restrict__ nn, double* __restrict__ nnd) {
    __shared__ double wm[4];
    __shared__ int wi[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    if (size[i] == 0) {
        if (tid == 0) nn[i] = -1;
        return;
    }
    const double inf = (double)INFINITY;
    const double* __restrict__ row = D + (long long)i * n;
    double m = inf;
    int mi = 0x7fffffff;
    constexpr int U = 8;
    for (int j0 = tid; j0 < n; j0 += U * 256) {
        double v[U];
        int live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = min(j0 + u * 256, n - 1);
            v[u] = row[j];
            live[u] = size[j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * 256;
            const double w = (j < n && live[u] > 0) ? v[u] : inf;          // (the diagonal holds +inf)
            if (w < m) { m = w; mi = j; }                                    // (ascending j per thread: the lowest index stays)
        }
    }
    argmin_allreduce64(m, mi);
    if ((tid & 63) == 0) { wm[tid >> 6] = m; wi[tid >> 6] = mi; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) argmin_pair(m, mi, wm[w], wi[w]);
        nn[i] = mi < n ? mi : -1;
        nnd[i] = m;
    }
}

// state[2] = merges so far, state[3] = pairs of this round; role[x] = 0 untouched, 1 dies (a), 2 merged row (b)
__global__ __launch_bounds__(1024) void rnn_pairs_kernel(int n, const int* __restrict__ size, const int* __restrict__ orig,
                                                         const int* __restrict__ nn, const double* __restrict__ nnd,
                                                         RnnPair* __restrict__ pairs, int* __restrict__ role,
                                                         ChainMergeDev* __restrict__ merges, int* __restrict__ state) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = state[2];
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        int j = -1;
        if (i < n && size[i] > 0) {
            j = nn[i];
            if (!(j > i && nn[j] == i)) j = -1;              // (the pair is recorded by its smaller index)
        }
        const int hit = j >= 0 ? 1 : 0;
        // inclusive prefix count inside the wave from the ballot (no shuffles: vbx_device.hpp, add_xor)
        const int incl = __popcll(__ballot(hit) & ((2ull << lane) - 1ull));
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (i < n) role[i] = 0;
        __syncthreads();                                     // (roles are cleared before any pair of this block sets one)
        if (hit) {
            const int p = before + incl - 1;
            pairs[p] = RnnPair{i, j, size[i], size[j]};
            merges[k0 + p] = ChainMergeDev{orig[i], orig[j], nnd[i]};
        }
        __syncthreads();
        if (tid == 1023) carry = before + incl;
        __syncthreads();
    }
    __syncthreads();
    const int np = carry;
    for (int p = tid; p < np; p += 1024) {                   // (after every role has been cleared)
        role[pairs[p].a] = 1;
        role[pairs[p].b] = 2;
    }
    if (tid == 0) {
        state[3] = np;
        state[2] = k0 + np;
    }
}

// grid = pairs of the round
__global__ __launch_bounds__(256) void rnn_rows_kernel(double* __restrict__ D, int n, const int* __restrict__ size,
                                                        const RnnPair* __restrict__ pairs) {
    const RnnPair pr = pairs[blockIdx.x];
    const double fa = (double)pr.na, fb = (double)pr.nb, fs = (double)(pr.na + pr.nb);
    const double* __restrict__ ra = D + (long long)pr.a * n;
    double* __restrict__ rb = D + (long long)pr.b * n;
    for (int k = threadIdx.x; k < n; k += 256) {
        if (k == pr.a || k == pr.b || size[k] == 0) continue;
        rb[k] = average_update(fa, ra[k], fb, rb[k], fs);
    }
}

// grid = rows of the matrix (dead and dying rows leave at once)
__global__ __launch_bounds__(256) void rnn_cols_kernel(double* __restrict__ D, int n, const int* __restrict__ size,
                                                        const int* __restrict__ role, const RnnPair* __restrict__ pairs,
                                                        const int* __restrict__ state) {
    const int x = blockIdx.x;
    if (size[x] == 0 || role[x] == 1) return;
    const int np = state[3];
    double* __restrict__ row = D + (long long)x * n;
    for (int p = threadIdx.x; p < np; p += 256) {
        const RnnPair pr = pairs[p];
        if (pr.b == x) continue;                             // (its own pair: the diagonal)
        row[pr.b] = average_update((double)pr.na, row[pr.a], (double)pr.nb, row[pr.b], (double)(pr.na + pr.nb));
    }
}

// grid = pairs: the entries between two merged rows take the value the row of the smaller index holds
__global__ __launch_bounds__(256) void rnn_canon_kernel(double* __restrict__ D, int n, const RnnPair* __restrict__ pairs,
                                                         const int* __restrict__ state) {
    const int np = state[3];
    const int b = pairs[blockIdx.x].b;
    for (int q = threadIdx.x; q < np; q += 256) {
        const int b2 = pairs[q].b;
        if (b2 > b) D[(long long)b2 * n + b] = D[(long long)b * n + b2];
    }
}

__global__ __launch_bounds__(256) void rnn_sizes_kernel(int* __restrict__ size, const RnnPair* __restrict__ pairs,
                                                         const int* __restrict__ state) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= state[3]) return;
    const RnnPair pr = pairs[p];
    size[pr.a] = 0;
    size[pr.b] = pr.na + pr.nb;
}

}  // namespace vbx
