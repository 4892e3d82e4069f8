// vbx_operator.hpp -- one column of the forward transfer operator of a run of frames (VBx.py:167-171 in the linear
// domain), shared by chunk_loglik (the chunk operators of the boundary walk) and chunk_post (the operators of the four
// sub-chunks its re-run is cut into).
//
//   x <- b_f (lp x + c sum(x))   for f = lo .. hi-1, started from the unit vector e_col
//
// PH adjacent lanes share a column and hold NR = SP / PH states each (a multiple of four).  With lp > 0 the recursion
// runs on z_f = x_f / lp^(transitions so far):  z <- b (z + (c / lp) sum(z)), one FMA and one product per state instead of
// three operations; lp^(transitions) goes into the column's mantissa and exponent at the end (lp == 0, or subnormally
// small, keeps the plain form).  The loop is written on pairs of states so that every product, FMA and partial sum
// is a v_pk_*_f32; columns are rescaled by exact powers of two every four frames (a frame shrinks a column sum by at
// least min c = 1e-8, so four frames stay inside the f32 range) and the exponent is kept aside.
#pragma once
#include <type_traits>
#include "vbx_scan.hpp"

namespace vbx {

// btile: b of the tile in LDS, [frames][SP].  first_plain: frame lo is frame 0 of the recording (x <- b_0 x, VBx.py:163:
// no transition).  c_rec: the recursion's c of the recording, [SP] (c / lp in the scaled form: BatchView::cop, written by
// fin_kernel); lppow: lp^n as (mantissa, exponent) for n = 0 .. kTileFrames (BatchView::lppow, host table) -- a dozen f64
// divisions, a log2 and an exp2 per wave and operator used to be a fifth of chunk_loglik's vector instructions.
// A thread builds NC columns at once -- colbase, colbase + SP / NC, ... -- on the same NR states: the rows of b and c are
// fetched once for all of them and the NC recursions are independent instruction streams.  NC = 1 is what runs: two
// columns per thread (on twice the lanes per column, the same thread count) measured 4-7 % slower on every shape
// (vbx_chunk_loglik.hpp; NOTES.md, round 3).
// -> x[k][NR] (column sums in [0.5, 1)), expo[k] (the column is x * 2^expo; -(1 << 24) for an all-zero column, which must
// never win an exponent maximum).
template <typename R, int SP, int PH, int NC>
__device__ __forceinline__ void operator_columns(const R* btile, int lo, int hi, bool first_plain, int colbase, int part,
                                                 double lp_d, const R* __restrict__ c_rec, const LpPow* __restrict__ lppow,
                                                 R (&x)[NC][SP / PH], int (&expo)[NC]) {
    using R2 = typename Vec<R>::v2;
    using R4 = typename Vec<R>::v4;
    constexpr int NR = SP / PH, NP = NR / 2, CS = SP / NC;
    static_assert(NR % 4 == 0, "operator lanes hold a multiple of four states");
    const int j0 = part * NR;
    const R lp = (R)lp_d;
    const bool scaled = lp_d >= 0x1p-20;
    R c[NR];
#pragma unroll
    for (int q = 0; q < NR / 4; ++q) {
        const R4 c4 = *reinterpret_cast<const R4*>(c_rec + j0 + 4 * q);
        c[4 * q] = c4.x; c[4 * q + 1] = c4.y; c[4 * q + 2] = c4.z; c[4 * q + 3] = c4.w;
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
#pragma unroll
        for (int r = 0; r < NR; ++r) x[k][r] = (j0 + r == colbase + k * CS) ? (R)1 : (R)0;
        expo[k] = 0;
    }
    int step = lo;
    if (first_plain) {
#pragma unroll
        for (int k = 0; k < NC; ++k)
#pragma unroll
            for (int r = 0; r < NR; ++r) x[k][r] *= btile[lo * SP + j0 + r];
        step = lo + 1;
    }
    const int transitions = hi - step;
    auto recursion = [&](auto scaled_tag) {
        R2 x2[NC][NP], c2[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            c2[p] = R2{c[2 * p], c[2 * p + 1]};
#pragma unroll
            for (int k = 0; k < NC; ++k) x2[k][p] = R2{x[k][2 * p], x[k][2 * p + 1]};
        }
        const R2 lp2 = R2{lp, lp};
        auto colsum2 = [&](int k) {              // pairwise: packed adds
            R2 v[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) v[p] = x2[k][p];
#pragma unroll
            for (int w = NP / 2; w >= 1; w >>= 1)
#pragma unroll
                for (int p = 0; p < w; ++p) v[p] += v[p + w];
            return column_sum<PH>(v[0].x + v[0].y);
        };
        auto frame = [&](int f, const R (&sig)[NC]) {
            const R* row = btile + f * SP + j0;
#pragma unroll
            for (int q = 0; q < NR / 4; ++q) {
                const R4 b4 = *reinterpret_cast<const R4*>(row + 4 * q);
                const R2 b0 = R2{b4.x, b4.y}, b1 = R2{b4.z, b4.w};
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const R2 sig2 = R2{sig[k], sig[k]};
                    if (decltype(scaled_tag)::value) {
                        x2[k][2 * q] = b0 * (c2[2 * q] * sig2 + x2[k][2 * q]);
                        x2[k][2 * q + 1] = b1 * (c2[2 * q + 1] * sig2 + x2[k][2 * q + 1]);
                    } else {
                        x2[k][2 * q] = b0 * (lp2 * x2[k][2 * q] + c2[2 * q] * sig2);
                        x2[k][2 * q + 1] = b1 * (lp2 * x2[k][2 * q + 1] + c2[2 * q + 1] * sig2);
                    }
                }
            }
        };
        auto renorm = [&](int k) {               // column sum back to [0.5, 1): one exact product per pair
            R sig = colsum2(k);
            const int e = rescale_exponent(sig);
            expo[k] += e;
            const R sc = scale2((R)1, -e);
            const R2 sc2 = R2{sc, sc};
#pragma unroll
            for (int p = 0; p < NP; ++p) x2[k][p] *= sc2;
            return sig * sc;
        };
        R sig[NC];
        for (; step + 4 <= hi; step += 4) {
#pragma unroll
            for (int k = 0; k < NC; ++k) sig[k] = renorm(k);
            frame(step, sig);
#pragma unroll
            for (int u = 1; u < 4; ++u) {
#pragma unroll
                for (int k = 0; k < NC; ++k) sig[k] = colsum2(k);
                frame(step + u, sig);
            }
        }
        for (; step < hi; ++step) {
#pragma unroll
            for (int k = 0; k < NC; ++k) sig[k] = renorm(k);
            frame(step, sig);
        }
#pragma unroll
        for (int k = 0; k < NC; ++k)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                x[k][2 * p] = x2[k][p].x;
                x[k][2 * p + 1] = x2[k][p].y;
            }
    };
    R mant = (R)1;
    int fl = 0;
    if (scaled) {
        recursion(std::true_type{});
        const LpPow pw = lppow[transitions];                 // lp^transitions = mant * 2^fl, mant in [1, 2)
        mant = (R)pw.mant;
        fl = pw.fl;
    } else {
        recursion(std::false_type{});
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {   // lp^transitions, then the final power-of-two normalisation: column sums end in [0.5, 1)
        expo[k] += fl;
        R v = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            x[k][r] *= mant;
            v += x[k][r];
        }
        const R sig = column_sum<PH>(v);
        const int e = rescale_exponent(sig);
        expo[k] += e;
#pragma unroll
        for (int r = 0; r < NR; ++r) x[k][r] = scale2(x[k][r], -e);
        if (!(sig > (R)0)) expo[k] = -(1 << 24);
    }
}

}  // namespace vbx
