// vbx_host_steps.hpp -- C ABI: step-level entry points (forward_backward, mstep, loglik) the parity tests call
// (one translation unit with vbx_capi.hip, which includes the parts in order; not a stand-alone header)
#pragma once
extern "C" {

// ---------------------------------------------------------------------------------------
// step-level entry points (parity tests)
// ---------------------------------------------------------------------------------------
extern "C++" {
namespace {
template <typename R>
int fb_step_impl(vbx_batch* b, int64_t T, int32_t S, const double* lls, double* gamma, double* tll, double* entered,
                 double* lfw, double* lbw) {
    vbx_ctx* ctx = b->ctx;
    const int Sp = b->Sp;
    std::vector<R> bm((size_t)T * Sp, (R)0), mr((size_t)T);
    for (int64_t t = 0; t < T; ++t) {
        double m = -INFINITY;
        for (int s = 0; s < S; ++s) m = std::max(m, lls[(size_t)t * S + s]);
        const R mq = (R)m;                      // the device keeps the row max in working precision
        mr[(size_t)t] = mq;
        for (int s = 0; s < S; ++s) bm[(size_t)t * Sp + s] = (R)std::exp(lls[(size_t)t * S + s] - (double)mq);
    }
    HIPCHK(ctx, hipMemcpy(b->d_bmat, bm.data(), sizeof(R) * bm.size(), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(b->d_mrow, mr.data(), sizeof(R) * mr.size(), hipMemcpyHostToDevice));
    const bool want_logs = lfw || lbw;
    if (want_logs) {
        int rc2 = dmalloc_bytes(ctx, &b->d_fw_scale, (size_t)T * sizeof(R));
        if (rc2 == VBX_OK) rc2 = dmalloc_bytes(ctx, &b->d_bw_scale, (size_t)T * sizeof(R));
        if (rc2 != VBX_OK) return rc2;
    }
    b->fuse = 0;                                  // stand-alone scan kernels: one operator per tile
    int rc = choose_fb_algo(b, want_logs);
    if (rc != VBX_OK) return rc;
    rc = upload_recs(b);
    if (rc != VBX_OK) return rc;
    launch_fb<R>(b, 0.0);
    launch_post<R>(b, 0.0);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    RecState st;
    HIPCHK(ctx, hipMemcpy(&st, b->d_state + (size_t)b->state_cur * b->n_rec, sizeof st, hipMemcpyDeviceToHost));
    if (b->use_chunked) {
        std::vector<double> tp((size_t)b->ntiles_total);
        HIPCHK(ctx, hipMemcpy(tp.data(), b->d_tllpart, sizeof(double) * tp.size(), hipMemcpyDeviceToHost));
        st.tll = 0.0;
        for (double v : tp) st.tll += v;
    }
    if (tll) *tll = st.tll;
    if (gamma) {
        b->mirrors_valid = false;
        rc = leaf_get_result(b, 0, gamma, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr);
        if (rc != VBX_OK) return rc;
    }
    if (entered) {
        std::vector<double> ep((size_t)b->ntiles_total * Sp);
        HIPCHK(ctx, hipMemcpy(ep.data(), b->d_epart, sizeof(double) * ep.size(), hipMemcpyDeviceToHost));
        for (int s = 0; s < S; ++s) {
            double acc = 0.0;
            for (int tl = 0; tl < b->ntiles_total; ++tl) acc += ep[(size_t)tl * Sp + s];
            entered[s] = acc;
        }
    }
    if (want_logs) {
        // lfw[t] = log ahat[t] + sum_{u<=t} (log s_u + m_u);  lbw[t] = log bhat[t] + sum_{u>t} (log q_{u-1} + m_u)
        std::vector<R> ah((size_t)T * Sp), bh((size_t)T * Sp), fs((size_t)T), bs((size_t)T);
        HIPCHK(ctx, hipMemcpy(ah.data(), b->d_ahat, sizeof(R) * ah.size(), hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(bh.data(), b->d_bhat, sizeof(R) * bh.size(), hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(fs.data(), b->d_fw_scale, sizeof(R) * fs.size(), hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(bs.data(), b->d_bw_scale, sizeof(R) * bs.size(), hipMemcpyDeviceToHost));
        if (lfw) {
            double cum = 0.0;
            for (int64_t t = 0; t < T; ++t) {
                cum += std::log((double)fs[(size_t)t]) + (double)mr[(size_t)t];
                for (int s = 0; s < S; ++s) lfw[(size_t)t * S + s] = std::log((double)ah[(size_t)t * Sp + s]) + cum;
            }
        }
        if (lbw) {
            double cum = 0.0;
            for (int64_t t = T - 1; t >= 0; --t) {
                if (t < T - 1) cum += std::log((double)bs[(size_t)t]) + (double)mr[(size_t)t + 1];
                for (int s = 0; s < S; ++s) lbw[(size_t)t * S + s] = std::log((double)bh[(size_t)t * Sp + s]) + cum;
            }
        }
    }
    return VBX_OK;
}
}  // namespace
}  // extern "C++"

int vbx_forward_backward(vbx_ctx* ctx, int64_t T, int32_t S, const double* lls, const double* pi, const double* ip,
                         double loopProb, int precision, int fb_algo, double* gamma, double* tll, double* entered,
                         double* lfw, double* lbw) {
    if (!ctx) return VBX_ERR_INVALID;
    if (!lls || !pi || T <= 0 || S <= 0) FAIL(ctx, VBX_ERR_INVALID, "vbx_forward_backward: bad argument");
    if (!(loopProb >= 0.0 && loopProb <= 1.0)) FAIL(ctx, VBX_ERR_INVALID, "loopProb=%g outside [0,1]", loopProb);
    vbx_batch* b = nullptr;
    int rc = vbx_batch_create(ctx, 1, &T, &S, 32, precision, 1, &b);
    if (rc != VBX_OK) return rc;
    rc = vbx_batch_set_option(b, VBX_OPT_FB_ALGO, fb_algo);
    if (rc == VBX_OK) {
        RecDesc& rd = b->recs[0];
        rd.lp = loopProb;
        rd.Fa = rd.Fb = 1.0;
        std::vector<double> pip(b->Sp, 0.0), ipp(b->Sp, 0.0);
        for (int s = 0; s < S; ++s) {
            pip[s] = pi[s];
            ipp[s] = ip ? ip[s] : pi[s];
        }
        rc = dmalloc(ctx, &b->d_ip, (size_t)b->Sp);
        hipError_t e = hipSuccess;
        if (rc == VBX_OK) e = hipMemcpy(b->d_pi, pip.data(), sizeof(double) * b->Sp, hipMemcpyHostToDevice);
        if (rc == VBX_OK && e == hipSuccess) e = hipMemcpy(b->d_ip, ipp.data(), sizeof(double) * b->Sp, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            ctx->err = std::string("pi upload failed: ") + hipGetErrorString(e);
            rc = VBX_ERR_HIP;
        }
        b->recs_dirty = true;
    }
    if (rc == VBX_OK)
        rc = precision == VBX_PREC_FP64 ? fb_step_impl<double>(b, T, S, lls, gamma, tll, entered, lfw, lbw)
                                        : fb_step_impl<float>(b, T, S, lls, gamma, tll, entered, lfw, lbw);
    vbx_batch_destroy(b);
    return rc;
}

extern "C++" {
namespace {
template <typename R>
int fb_dense_impl(vbx_ctx* ctx, int64_t T, int32_t S, const double* lls, const double* tr, const double* ip, double* gamma,
                  double* tll, double* lfw, double* lbw) {
    int Sp = 16;
    while (Sp < S && Sp < 256) Sp *= 2;
    if (S > 256) Sp = round_up(S, 64);                               // fb_dense_big_kernel: M in HBM, any S
    const size_t cells = (size_t)T * Sp;
    std::vector<R> bm(cells, (R)0), m0((size_t)Sp * Sp, (R)0), m1((size_t)Sp * Sp, (R)0), v0(Sp, (R)0);
    std::vector<double> mr((size_t)T);
    for (int64_t t = 0; t < T; ++t) {
        double m = -INFINITY;
        for (int s = 0; s < S; ++s) m = std::max(m, lls[(size_t)t * S + s]);
        mr[(size_t)t] = m;
        for (int s = 0; s < S; ++s) bm[(size_t)t * Sp + s] = (R)std::exp(lls[(size_t)t * S + s] - m);
    }
    for (int i = 0; i < S; ++i) {
        v0[i] = (R)(ip[i] + 1e-8);                                   // VBx.py:163
        for (int j = 0; j < S; ++j) {
            const R a = (R)(tr[(size_t)i * S + j] + 1e-8);           // VBx.py:158
            m0[(size_t)i * Sp + j] = a;                              // forward: M[k][o] = A[k][o]
            m1[(size_t)j * Sp + i] = a;                              // backward: M[k][o] = A[o][k]
        }
    }
    R *d_m0 = nullptr, *d_m1 = nullptr, *d_b = nullptr, *d_v0 = nullptr, *d_ah = nullptr, *d_bh = nullptr, *d_fs = nullptr, *d_bs = nullptr;
    int rc = dmalloc(ctx, &d_m0, m0.size());
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_m1, m1.size());
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_b, cells);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_v0, (size_t)Sp);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_ah, cells);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_bh, cells);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_fs, (size_t)T);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_bs, (size_t)T);
    auto release = [&]() {
        for (void* p : {(void*)d_m0, (void*)d_m1, (void*)d_b, (void*)d_v0, (void*)d_ah, (void*)d_bh, (void*)d_fs, (void*)d_bs}) ctx_free(ctx, p);
    };
    if (rc != VBX_OK) { release(); return rc; }
    hipStream_t st = ctx->stream;
    hipError_t e = hipMemcpyAsync(d_m0, m0.data(), sizeof(R) * m0.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_m1, m1.data(), sizeof(R) * m1.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_b, bm.data(), sizeof(R) * cells, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_v0, v0.data(), sizeof(R) * Sp, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
#define VBX_FB_DENSE(SP_) hipLaunchKernelGGL((fb_dense_kernel<R, SP_>), dim3(2), dim3(FbDenseCfg<SP_>::kThreads), 0, st, \
                                             d_m0, d_m1, d_b, d_v0, d_ah, d_bh, d_fs, d_bs, (int)T, (int)S)
        switch (Sp) {
            case 16: VBX_FB_DENSE(16); break;
            case 32: VBX_FB_DENSE(32); break;
            case 64: VBX_FB_DENSE(64); break;
            case 128: VBX_FB_DENSE(128); break;
            case 256: VBX_FB_DENSE(256); break;
            default:      // more than 256 states (the register-resident kernel would drop them: round-3 advisor finding)
                hipLaunchKernelGGL((fb_dense_big_kernel<R>), dim3(2), dim3(1024), 0, st, d_m0, d_m1, d_b, d_v0, d_ah, d_bh,
                                   d_fs, d_bs, (int)T, (int)S, Sp);
                break;
        }
#undef VBX_FB_DENSE
        e = hipGetLastError();
    }
    std::vector<R> ah(cells), bh(cells), fs((size_t)T), bs((size_t)T);
    if (e == hipSuccess) e = hipMemcpyAsync(ah.data(), d_ah, sizeof(R) * cells, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(bh.data(), d_bh, sizeof(R) * cells, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(fs.data(), d_fs, sizeof(R) * (size_t)T, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(bs.data(), d_bs, sizeof(R) * (size_t)T, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    release();
    if (e != hipSuccess) FAIL(ctx, VBX_ERR_HIP, "forward_backward (dense): %s", hipGetErrorString(e));
    // lfw[t] = log ahat[t] + sum_{u<=t} (log s_u + m_u);  lbw[t] = log bhat[t] + sum_{u>=t, u<T-1} log q_u + sum_{u>t} m_u
    double cum = 0.0;
    for (int64_t t = 0; t < T; ++t) {
        cum += std::log((double)fs[(size_t)t]) + mr[(size_t)t];
        if (lfw)
            for (int s = 0; s < S; ++s) lfw[(size_t)t * S + s] = std::log((double)ah[(size_t)t * Sp + s]) + cum;
    }
    if (tll) *tll = cum;
    if (lbw) {
        double back = 0.0;
        for (int64_t t = T - 1; t >= 0; --t) {
            if (t < T - 1) back += std::log((double)bs[(size_t)t]) + mr[(size_t)t + 1];
            for (int s = 0; s < S; ++s) lbw[(size_t)t * S + s] = std::log((double)bh[(size_t)t * Sp + s]) + back;
        }
    }
    if (gamma)
        for (int64_t t = 0; t < T; ++t) {
            double tot = 0.0;
            for (int s = 0; s < S; ++s) tot += (double)ah[(size_t)t * Sp + s] * (double)bh[(size_t)t * Sp + s];
            for (int s = 0; s < S; ++s)
                gamma[(size_t)t * S + s] = (double)ah[(size_t)t * Sp + s] * (double)bh[(size_t)t * Sp + s] / tot;
        }
    return VBX_OK;
}
}  // namespace
}  // extern "C++"

int vbx_forward_backward_dense(vbx_ctx* ctx, int64_t T, int32_t S, const double* lls, const double* tr, const double* ip,
                               int precision, double* gamma, double* tll, double* lfw, double* lbw) {
    if (!ctx) return VBX_ERR_INVALID;
    if (!lls || !tr || !ip || T <= 0 || S <= 0) FAIL(ctx, VBX_ERR_INVALID, "vbx_forward_backward_dense: bad argument");
    if (S > vbx::kFbDenseBigMax) FAIL(ctx, VBX_ERR_UNSUPPORTED, "forward_backward (dense): S=%d exceeds %d states", S, vbx::kFbDenseBigMax);
    if (precision != VBX_PREC_FP32 && precision != VBX_PREC_FP64) FAIL(ctx, VBX_ERR_INVALID, "unknown precision %d", precision);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return precision == VBX_PREC_FP64 ? fb_dense_impl<double>(ctx, T, S, lls, tr, ip, gamma, tll, lfw, lbw)
                                      : fb_dense_impl<float>(ctx, T, S, lls, tr, ip, gamma, tll, lfw, lbw);
}

int vbx_mstep(vbx_ctx* ctx, int64_t T, int32_t S, int32_t D, const double* X, const double* Phi, const double* gamma,
              double Fa, double Fb, int precision, double* alpha, double* invL) {
    if (!ctx) return VBX_ERR_INVALID;
    if (!X || !Phi || !gamma) FAIL(ctx, VBX_ERR_INVALID, "vbx_mstep: NULL input");
    vbx_batch* b = nullptr;
    int rc = vbx_batch_create(ctx, 1, &T, &S, D, precision, 1, &b);
    if (rc != VBX_OK) return rc;
    std::vector<double> pi(S, 1.0 / S);
    rc = vbx_batch_set_recording(b, 0, X, VBX_F64, Phi, pi.data(), gamma, VBX_F64, nullptr, nullptr, 0.9, Fa, Fb);
    if (rc == VBX_OK) rc = upload_recs(b);
    if (rc == VBX_OK) {
        if (precision == VBX_PREC_FP64) launch_mstep<double>(b, 0.0); else launch_mstep<float>(b, 0.0);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) {
            ctx->err = std::string("mstep kernels failed: ") + hipGetErrorString(e);
            rc = VBX_ERR_HIP;
        }
    }
    if (rc == VBX_OK) rc = vbx_batch_get_result(b, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, alpha, invL);
    vbx_batch_destroy(b);
    return rc;
}

int vbx_loglik(vbx_ctx* ctx, int64_t T, int32_t S, int32_t D, const double* X, const double* Phi, const double* alpha,
               const double* invL, double Fa, int precision, double* log_p) {
    if (!ctx) return VBX_ERR_INVALID;
    if (!X || !Phi || !alpha || !invL || !log_p) FAIL(ctx, VBX_ERR_INVALID, "vbx_loglik: NULL input");
    vbx_batch* b = nullptr;
    int rc = vbx_batch_create(ctx, 1, &T, &S, D, precision, 1, &b);
    if (rc != VBX_OK) return rc;
    std::vector<double> pi(S, 1.0 / S), g0((size_t)T * S, 1.0 / S);
    rc = vbx_batch_set_recording(b, 0, X, VBX_F64, Phi, pi.data(), g0.data(), VBX_F64, alpha, invL, 0.9, Fa, 1.0);
    if (rc == VBX_OK) rc = upload_recs(b);
    const size_t cells = (size_t)T * b->Sp;
    if (rc == VBX_OK) rc = dmalloc_bytes(ctx, &b->d_lraw, cells * b->rsize);
    if (rc == VBX_OK) {
        auto go = [&](auto tag) {
            using R = decltype(tag);
            auto v = b->view<R>(0.0);
            launch_fin<R>(b, 0.0, 1);
            launch_loglik<R>(b, 0.0, true);
        };
        if (precision == VBX_PREC_FP64) go(double{}); else go(float{});
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) {
            ctx->err = std::string("loglik kernels failed: ") + hipGetErrorString(e);
            rc = VBX_ERR_HIP;
        }
    }
    if (rc == VBX_OK) {
        // add the per-frame constant Fa*G_t of VBx.py:87,97 on the host (f64)
        auto fetch = [&](auto tag) -> int {
            using R = decltype(tag);
            std::vector<R> raw(cells);
            HIPCHK(ctx, hipMemcpy(raw.data(), b->d_lraw, sizeof(R) * cells, hipMemcpyDeviceToHost));
            for (int64_t t = 0; t < T; ++t) {
                double ss = 0.0;
                for (int d = 0; d < D; ++d) ss += X[(size_t)t * D + d] * X[(size_t)t * D + d];
                const double G = -0.5 * (ss + D * std::log(2.0 * M_PI));
                for (int s = 0; s < S; ++s) log_p[(size_t)t * S + s] = (double)raw[(size_t)t * b->Sp + s] + Fa * G;
            }
            return VBX_OK;
        };
        rc = precision == VBX_PREC_FP64 ? fetch(double{}) : fetch(float{});
    }
    vbx_batch_destroy(b);
    return rc;
}

}  // extern "C"
