// vbx_frontend.hpp -- what the reference driver computes either side of VBx(), for ALL x-vectors of an archive at once
// (everything float64 like the reference):
//
//   vbhmm.py:125-129   xproj = l2_norm( l2_norm(x - mean1) lda - mean2 )         xv_center_norm, xv_gemm, xv_norm
//   vbhmm.py:153       fea   = (xproj - plda_mu) plda_tr^T [:, :lda_dim]          xv_gemm (the mean goes through the
//                                                                                 product: - plda_mu plda_tr^T)
//   vbhmm.py:150-152   qinit = softmax(init_smoothing * onehot(AHC labels))       qinit_kernel, straight into gamma
//   vbhmm.py:160-162   first / second speaker = argsort(-q, axis=1)[:, 0 / 1]     top2_kernel
//
// The products are small (n x 256 x 128 and n x 128 x 128 for n ~ 1e5 x-vectors of an archive: 9 GFLOP) and run on
// v_mfma_f64_16x16x4; their point is that nothing but the raw x-vectors goes up and nothing but labels comes down.
#pragma once
#include "vbx_device.hpp"

namespace vbx {

// y[t][0..Kp) = (x[t] - mean) / |x[t] - mean|, zero-padded to Kp; x has row stride ldx.  One wavefront per row.
// mean == nullptr: no centring; in place (y == x, ldx == Kp) is allowed: a row is read completely before it is written.
template <typename XT>
__global__ __launch_bounds__(256) void xv_center_norm_kernel(const XT* x, const double* __restrict__ mean,
                                                              double* y, long long n, int D, int ldx, int Kp) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + wave;
    if (t >= n) return;
    double ss = 0.0;
    for (int d = lane; d < D; d += 64) {
        const double v = (double)x[t * ldx + d] - (mean ? mean[d] : 0.0);
        ss += v * v;
    }
    ss = allreduce_sum<64>(ss);
    const double nrm = sqrt(ss);                               // diarization_lib.py:185: a / np.linalg.norm(a, axis=1)
    for (int d0 = 0; d0 < Kp; d0 += 64) {
        const int d = d0 + lane;
        const double v = d < D ? (double)x[t * ldx + d] - (mean ? mean[d] : 0.0) : 0.0;
        if (d < Kp) y[t * Kp + d] = v / nrm;
    }
}

// C[t][c] = sum_k A[t][k] B[k][c] - sub[c] for c < N (stored with row stride ldc); A [n][Kp], B [Kp][Np], Kp % 4 == 0,
// Np % 16 == 0.  grid = ceil(n / 64), block = 256: wave w owns rows 16w .. 16w+15 of the block and every column tile.
__global__ __launch_bounds__(256) void xv_gemm_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                       const double* __restrict__ sub, double* __restrict__ C,
                                                       long long n, int Kp, int Np, int N, int ldc) {
    using M = Mfma16<double>;
    using acc_t = M::acc_t;
    constexpr int NTB = 8;                                     // column tiles per pass (64 accumulator registers)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    const long long r0 = (long long)blockIdx.x * 64 + 16 * wave;
    if (r0 >= n) return;
    const double* __restrict__ pa = A + min(r0 + i, n - 1) * Kp + kq;          // rows past the end are clamped, never stored
    for (int nt0 = 0; nt0 * 16 < Np; nt0 += NTB) {
        acc_t acc[NTB];
#pragma unroll
        for (int u = 0; u < NTB; ++u) acc[u] = acc_t{0, 0, 0, 0};
        for (int k0 = 0; k0 < Kp; k0 += 4) {
            const double a = pa[k0];
            const double* __restrict__ pb = B + (long long)(k0 + kq) * Np + i;
#pragma unroll
            for (int u = 0; u < NTB; ++u) {
                const int c0 = 16 * (nt0 + u);
                if (c0 < Np) acc[u] = M::mma(a, pb[c0], acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < NTB; ++u) {
            const int c = 16 * (nt0 + u) + i;
            if (c < N) {
                const double s = sub ? sub[c] : 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long row = r0 + M::row(lane, r);
                    if (row < n) C[row * ldc + c] = acc[u][r] - s;
                }
            }
        }
    }
}

// gamma[t][s] = hi if s == labels[t] else lo (s < S), 0 for the padded speakers: softmax(smoothing * onehot) has two
// distinct values.  vbhmm.py:150-152.
template <typename R>
__global__ __launch_bounds__(256) void qinit_kernel(const int* __restrict__ labels, R* __restrict__ gamma, long long T,
                                                     int S, int Sp, double hi, double lo) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * Sp) return;
    const long long t = idx / Sp;
    const int s = (int)(idx - t * Sp);
    gamma[idx] = s >= S ? (R)0 : (s == labels[t] ? (R)hi : (R)lo);
}

// out[t][s] = (double)gamma[t][s], s < S: the responsibilities a caller asks for (VBx.py:126: gamma[T][S] float64) unpadded and
// widened on the device, so that ONE copy lands them in the caller's array (vbx_batch_get_result).
template <typename R>
__global__ __launch_bounds__(256) void unpack_gamma_kernel(const R* __restrict__ gamma, double* __restrict__ out, long long T,
                                                            int S, int Sp) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * S) return;
    const long long t = idx / S;
    out[idx] = (double)gamma[t * Sp + (idx - t * S)];
}

// gamma[t][s] = (R)src[t][s] for s < S, 0 for the padded speakers: the caller's initial responsibilities (VBx.py:79-85) padded
// and converted on the device from the rows as they were uploaded (round 6: the host loop that did this cost a third of an upload)
template <typename R, typename SRC>
__global__ __launch_bounds__(256) void pad_gamma_kernel(const SRC* __restrict__ src, R* __restrict__ gamma, long long T, int S, int Sp) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * Sp) return;
    const long long t = idx / Sp;
    const int s = (int)(idx - t * Sp);
    gamma[idx] = s < S ? (R)src[t * S + s] : (R)0;
}

// The small per-recording arguments of the recordings set since the last synchronize, from the batch's pinned host block
// (per recording {Phi[Dp], sqrt Phi[Dp], pi0[Sp], flags[2]}) to where the kernels read them: one launch instead of two copies
// per recording.  flags[0] != 0: Phi is in the slot; flags[1] != 0: pi0 is.
__global__ __launch_bounds__(256) void scatter_args_kernel(const double* __restrict__ h_args, int per, int Dp, int Sp,
                                                           double* __restrict__ phi, double* __restrict__ pi) {
    const int rec = blockIdx.x;
    const double* a = h_args + (long long)rec * per;
    const bool has_phi = a[2 * Dp + Sp] != 0.0, has_pi = a[2 * Dp + Sp + 1] != 0.0;
    if (has_phi)
        for (int d = threadIdx.x; d < Dp; d += 256) phi[(long long)rec * Dp + d] = a[d];
    if (has_pi)
        for (int s = threadIdx.x; s < Sp; s += 256) pi[(long long)rec * Sp + s] = a[2 * Dp + s];
}

// All speaker models of a batch (or the ones flagged) unpadded and widened into ONE block: out[rec][s][d] at rec * S_max * D
template <typename R>
__global__ __launch_bounds__(256) void unpack_models_kernel(const R* __restrict__ alpha, const R* __restrict__ invL, const int* __restrict__ copy_of,
                                                            long long model_stride, double* __restrict__ out_alpha, double* __restrict__ out_invL,
                                                            int n_rec, int Sp, int D, int Dp) {
    const int rec = blockIdx.x;
    const long long base = (long long)copy_of[rec] * model_stride + (long long)rec * Sp * Dp;
    for (int idx = threadIdx.x; idx < Sp * D; idx += 256) {
        const int s = idx / D, d = idx - s * D;
        out_alpha[(long long)rec * Sp * D + idx] = (double)alpha[base + (long long)s * Dp + d];
        out_invL[(long long)rec * Sp * D + idx] = (double)invL[base + (long long)s * Dp + d];
    }
}

// out[s][d] = (double)model[s][d], s < S, d < D: alpha / invL as a caller gets them (VBx.py:126 return_model)
template <typename R>
__global__ __launch_bounds__(256) void unpack_model_kernel(const R* __restrict__ model, double* __restrict__ out, int S, int D, int Dp) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * D) return;
    const int s = idx / D;
    out[idx] = (double)model[(long long)s * Dp + (idx - s * D)];
}

// Largest and second largest responsibility of every frame; ties go to the lower index (what a stable argsort of -q
// gives; numpy's default argsort leaves the order of ties unspecified).  second = -1 when S == 1.
template <typename R>
__global__ __launch_bounds__(256) void top2_kernel(const R* __restrict__ gamma, int* __restrict__ first,
                                                    int* __restrict__ second, long long T, int S, int Sp) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const R* __restrict__ row = gamma + t * Sp;
    int i1 = 0, i2 = -1;
    R v1 = row[0], v2 = 0;
    for (int s = 1; s < S; ++s) {
        const R v = row[s];
        if (v > v1) {
            i2 = i1; v2 = v1;
            i1 = s; v1 = v;
        } else if (i2 < 0 || v > v2) {
            i2 = s; v2 = v;
        }
    }
    first[t] = i1;
    second[t] = i2;
}

}  // namespace vbx
