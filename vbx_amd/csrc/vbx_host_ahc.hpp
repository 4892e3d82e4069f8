// vbx_host_ahc.hpp -- C ABI: the rows next to the path -- score stage of the AHC initialisation, resident x-vectors, average linkage
// (one translation unit with vbx_capi.hip, which includes the parts in order; not a stand-alone header)
#pragma once
// ---------------------------------------------------------------------------------------
// score stage of the AHC initialisation (vbhmm.py:135-138)
// ---------------------------------------------------------------------------------------

struct vbx_scores {
    vbx_ctx* ctx = nullptr;
    long long n = 0;
    double* d_s = nullptr;
    size_t d_s_bytes = 0;
};

extern "C" {

int vbx_scores_destroy(vbx_scores* sc) {
    if (!sc) return VBX_OK;
    (void)hipSetDevice(sc->ctx->device);
    scratch_put(sc->ctx, sc->d_s, sc->d_s_bytes);
    delete sc;
    return VBX_OK;
}

int64_t vbx_scores_count(const vbx_scores* sc) { return sc ? sc->n : 0; }

// x: host pointer (uploaded) or, with on_device, rows already resident in HBM
static int cos_similarity_impl(vbx_ctx* ctx, int64_t T, int32_t D, const double* x, bool on_device, vbx_scores** out) {
    if (T > 200000) FAIL(ctx, VBX_ERR_UNSUPPORTED, "T=%lld: the T x T score matrix would not fit the device", (long long)T);
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int Dp = round_up(D, 16);
    double *d_x = nullptr, *d_xn = nullptr;
    vbx_scores* sc = new vbx_scores();
    sc->ctx = ctx;
    sc->n = (long long)T * T;
    size_t x_bytes = 0, xn_bytes = 0;
    int rc = on_device ? VBX_OK : scratch_get(ctx, &d_x, (size_t)T * D, &x_bytes);
    if (rc == VBX_OK) rc = scratch_get(ctx, &d_xn, (size_t)T * Dp, &xn_bytes);
    if (rc == VBX_OK) rc = scratch_get(ctx, &sc->d_s, (size_t)sc->n, &sc->d_s_bytes);
    hipError_t e = hipSuccess;
    if (rc == VBX_OK) {
        if (!on_device) e = hipMemcpyAsync(d_x, x, sizeof(double) * (size_t)T * D, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(vbx::cos_norm_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, ctx->stream,
                               on_device ? x : d_x, d_xn, (long long)T, (int)D, Dp);
            const unsigned nb = (unsigned)((T + 63) / 64);
            hipLaunchKernelGGL(vbx::cos_gemm_kernel, dim3(nb, nb), dim3(256), 0, ctx->stream, d_xn, sc->d_s, (long long)T, Dp);
            e = hipStreamSynchronize(ctx->stream);
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) {
            ctx->err = std::string("cos_similarity kernels failed: ") + hipGetErrorString(e);
            rc = VBX_ERR_HIP;
        }
    }
    if (!on_device) scratch_put(ctx, d_x, x_bytes);
    scratch_put(ctx, d_xn, xn_bytes);
    if (rc != VBX_OK) {
        vbx_scores_destroy(sc);
        return rc;
    }
    *out = sc;
    return VBX_OK;
}

int vbx_cos_similarity(vbx_ctx* ctx, int64_t T, int32_t D, const double* x, vbx_scores** out) {
    if (!ctx) return VBX_ERR_INVALID;
    if (!x || !out || T <= 0 || D <= 0) FAIL(ctx, VBX_ERR_INVALID, "vbx_cos_similarity: bad argument");
    return cos_similarity_impl(ctx, T, D, x, false, out);
}

// ---- x-vectors of an archive resident in HBM: projections, initial assignments, labels (vbx_frontend.hpp) ------------
struct vbx_xvectors {
    vbx_ctx* ctx = nullptr;
    long long n = 0;
    int Dl = 0, fea_dim = 0;
    double *d_xproj = nullptr, *d_fea = nullptr;
};

static const double* xvectors_fea_rows(const vbx_xvectors* xv, int64_t row0, int64_t T, int D, int device) {
    if (!xv || row0 < 0 || row0 + T > xv->n || D != xv->fea_dim || xv->ctx->device != device) return nullptr;
    return xv->d_fea + row0 * xv->fea_dim;
}

int vbx_xvectors_destroy(vbx_xvectors* xv) {
    if (!xv) return VBX_OK;
    (void)hipSetDevice(xv->ctx->device);
    (void)hipStreamSynchronize(xv->ctx->stream);
    ctx_free(xv->ctx, xv->d_xproj);
    ctx_free(xv->ctx, xv->d_fea);
    delete xv;
    return VBX_OK;
}

int vbx_xvectors_project(vbx_ctx* ctx, int64_t n, int32_t Din, int32_t Dl, int32_t fea_dim, const void* x, int x_dtype,
                         const double* mean1, const double* lda, const double* mean2, const double* plda_mu,
                         const double* plda_tr, vbx_xvectors** out) {
    if (!ctx) return VBX_ERR_INVALID;
    if (!out || !x || !mean1 || !lda || !mean2 || !plda_mu || !plda_tr || n <= 0 || Din <= 0 || Dl <= 0 || fea_dim <= 0 ||
        fea_dim > Dl || (x_dtype != VBX_F32 && x_dtype != VBX_F64))
        FAIL(ctx, VBX_ERR_INVALID, "vbx_xvectors_project: bad argument");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int Kp = round_up(Din, 4), Np = round_up(Dl, 16), Kp2 = round_up(Dl, 4), Np2 = round_up(fea_dim, 16);
    // padded operands: lda [Kp][Np]; plda_tr^T [Kp2][Np2] (column k = k-th output dim); the mean of the second product
    // goes through it: (x - mu) P = x P - mu P
    std::vector<double> ldap((size_t)Kp * Np, 0.0), ptp((size_t)Kp2 * Np2, 0.0), mup(Np2, 0.0), m2p(Np, 0.0);
    for (int k = 0; k < Din; ++k)
        for (int c = 0; c < Dl; ++c) ldap[(size_t)k * Np + c] = lda[(size_t)k * Dl + c];
    for (int c = 0; c < Dl; ++c) m2p[c] = mean2[c];
    for (int k = 0; k < fea_dim; ++k) {
        double acc = 0.0;
        for (int d = 0; d < Dl; ++d) {
            ptp[(size_t)d * Np2 + k] = plda_tr[(size_t)k * Dl + d];
            acc += plda_mu[d] * plda_tr[(size_t)k * Dl + d];
        }
        mup[k] = acc;
    }
    vbx_xvectors* xv = new vbx_xvectors();
    xv->ctx = ctx;
    xv->n = n;
    xv->Dl = Dl;
    xv->fea_dim = fea_dim;
    const size_t esz = x_dtype == VBX_F64 ? 8 : 4;
    void* d_x = nullptr;
    double *d_y1 = nullptr, *d_m1 = nullptr, *d_lda = nullptr, *d_m2 = nullptr, *d_pt = nullptr, *d_mu = nullptr;
    int rc = dmalloc_bytes(ctx, &d_x, (size_t)n * Din * esz);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_y1, (size_t)n * Kp);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_m1, (size_t)Din);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_lda, ldap.size());
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_m2, m2p.size());
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_pt, ptp.size());
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_mu, mup.size());
    if (rc == VBX_OK) rc = dmalloc(ctx, &xv->d_xproj, (size_t)n * Kp2);
    if (rc == VBX_OK) rc = dmalloc(ctx, &xv->d_fea, (size_t)n * fea_dim);
    hipError_t e = hipSuccess;
    if (rc == VBX_OK) {
        hipStream_t st = ctx->stream;
        e = hipMemcpyAsync(d_x, x, (size_t)n * Din * esz, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_m1, mean1, sizeof(double) * Din, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_lda, ldap.data(), sizeof(double) * ldap.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_m2, m2p.data(), sizeof(double) * m2p.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_pt, ptp.data(), sizeof(double) * ptp.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_mu, mup.data(), sizeof(double) * mup.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            const dim3 rows((unsigned)((n + 3) / 4)), blocks((unsigned)((n + 63) / 64));
            if (x_dtype == VBX_F64)
                hipLaunchKernelGGL((vbx::xv_center_norm_kernel<double>), rows, dim3(256), 0, st, (const double*)d_x, d_m1, d_y1, (long long)n, (int)Din, (int)Din, Kp);
            else
                hipLaunchKernelGGL((vbx::xv_center_norm_kernel<float>), rows, dim3(256), 0, st, (const float*)d_x, d_m1, d_y1, (long long)n, (int)Din, (int)Din, Kp);
            hipLaunchKernelGGL(vbx::xv_gemm_kernel, blocks, dim3(256), 0, st, d_y1, d_lda, d_m2, xv->d_xproj, (long long)n, Kp, Np, (int)Dl, Kp2);
            // (the columns Dl .. Kp2 of xproj must be zero for the second product)
            hipLaunchKernelGGL((vbx::xv_center_norm_kernel<double>), rows, dim3(256), 0, st, xv->d_xproj, (const double*)nullptr, xv->d_xproj, (long long)n, (int)Dl, Kp2, Kp2);
            hipLaunchKernelGGL(vbx::xv_gemm_kernel, blocks, dim3(256), 0, st, xv->d_xproj, d_pt, d_mu, xv->d_fea, (long long)n, Kp2, Np2, (int)fea_dim, (int)fea_dim);
            e = hipStreamSynchronize(st);
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) {
            ctx->err = std::string("x-vector projection failed: ") + hipGetErrorString(e);
            rc = VBX_ERR_HIP;
        }
    }
    for (void* p : {d_x, (void*)d_y1, (void*)d_m1, (void*)d_lda, (void*)d_m2, (void*)d_pt, (void*)d_mu}) ctx_free(ctx, p);
    if (rc != VBX_OK) {
        vbx_xvectors_destroy(xv);
        return rc;
    }
    *out = xv;
    return VBX_OK;
}

int vbx_xvectors_get(vbx_xvectors* xv, int which, int64_t row0, int64_t nrows, double* out) {
    if (!xv || !out || row0 < 0 || nrows < 0 || row0 + nrows > xv->n || (which != 0 && which != 1)) return VBX_ERR_INVALID;
    vbx_ctx* ctx = xv->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (which == 1) {
        HIPCHK(ctx, hipMemcpy(out, xv->d_fea + row0 * xv->fea_dim, sizeof(double) * (size_t)nrows * xv->fea_dim, hipMemcpyDeviceToHost));
    } else {
        const int ld = round_up(xv->Dl, 4);
        HIPCHK(ctx, hipMemcpy2D(out, sizeof(double) * xv->Dl, xv->d_xproj + row0 * ld, sizeof(double) * ld, sizeof(double) * xv->Dl,
                                (size_t)nrows, hipMemcpyDeviceToHost));
    }
    return VBX_OK;
}

int vbx_cos_similarity_resident(vbx_ctx* ctx, vbx_xvectors* xv, int64_t row0, int64_t T, vbx_scores** out) {
    if (!ctx) return VBX_ERR_INVALID;
    if (!xv || !out || T <= 0 || row0 < 0 || row0 + T > xv->n || xv->ctx->device != ctx->device)
        FAIL(ctx, VBX_ERR_INVALID, "vbx_cos_similarity_resident: bad argument");
    const int ld = round_up(xv->Dl, 4);               // (the padding columns are zero: part of the rows, no effect on the scores)
    return cos_similarity_impl(ctx, T, ld, xv->d_xproj + row0 * ld, true, out);
}

int vbx_scores_upload(vbx_ctx* ctx, int64_t n, const double* s, vbx_scores** out) {
    if (!ctx) return VBX_ERR_INVALID;
    if (!s || !out || n <= 0) FAIL(ctx, VBX_ERR_INVALID, "vbx_scores_upload: bad argument");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    vbx_scores* sc = new vbx_scores();
    sc->ctx = ctx;
    sc->n = n;
    int rc = scratch_get(ctx, &sc->d_s, (size_t)n, &sc->d_s_bytes);
    if (rc == VBX_OK) {
        hipError_t e = hipMemcpy(sc->d_s, s, sizeof(double) * (size_t)n, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            ctx->err = std::string("score upload failed: ") + hipGetErrorString(e);
            rc = VBX_ERR_HIP;
        }
    }
    if (rc != VBX_OK) {
        vbx_scores_destroy(sc);
        return rc;
    }
    *out = sc;
    return VBX_OK;
}

int vbx_scores_get(vbx_scores* sc, int64_t offset, int64_t count, double* out) {
    if (!sc) return VBX_ERR_INVALID;
    if (!out || offset < 0 || count < 0 || offset + count > sc->n) FAIL(sc->ctx, VBX_ERR_INVALID, "vbx_scores_get: bad range");
    HIPCHK(sc->ctx, hipSetDevice(sc->ctx->device));
    if (count) HIPCHK(sc->ctx, hipMemcpy(out, sc->d_s + offset, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost));
    return VBX_OK;
}

int vbx_linkage_average(int64_t n, const double* condensed, double* Z) {
    if (n < 1 || (n > 1 && (!condensed || !Z))) return VBX_ERR_INVALID;
    if (n > 65536) return VBX_ERR_UNSUPPORTED;               // n^2 doubles of working storage
    try {
        vbx::average_linkage(n, condensed, Z);
    } catch (const std::bad_alloc&) {
        return VBX_ERR_HIP - 100;                            // host allocation failure (no ctx to carry a message)
    }
    return VBX_OK;
}

int vbx_linkage_average_fastcluster(int64_t n, const double* condensed, double* Z) {
    if (n < 1 || (n > 1 && (!condensed || !Z))) return VBX_ERR_INVALID;
    if (n > 65536) return VBX_ERR_UNSUPPORTED;
    try {
        vbx::average_linkage_fastcluster(n, condensed, Z);
    } catch (const std::bad_alloc&) {
        return VBX_ERR_HIP - 100;
    }
    return VBX_OK;
}

int64_t vbx_ark_index(const void* buf, int64_t len, int64_t cap, int64_t* key_off, int32_t* key_len, int64_t* data_off,
                      int32_t* dim, int32_t* elem_size) {
    if (!buf || len < 0 || cap < 0 || (cap > 0 && (!key_off || !key_len || !data_off || !dim || !elem_size))) return -1;
    return vbx::ark_index(static_cast<const unsigned char*>(buf), len, cap, key_off, key_len, data_off, dim, elem_size);
}

int vbx_gather_rows(const void* buf, int64_t len, const int64_t* offsets, int64_t n, int64_t row_bytes, void* out) {
    if (!buf || !out || !offsets || n < 0 || row_bytes < 0) return VBX_ERR_INVALID;
    const unsigned char* src = static_cast<const unsigned char*>(buf);
    unsigned char* dst = static_cast<unsigned char*>(out);
    for (int64_t i = 0; i < n; ++i) {
        if (offsets[i] < 0 || offsets[i] + row_bytes > len) return VBX_ERR_INVALID;
        std::memcpy(dst + i * row_bytes, src + offsets[i], (size_t)row_bytes);
    }
    return VBX_OK;
}

int vbx_fcluster_distance(int64_t n, const double* Z, double t, int32_t* labels) {
    if (n < 1 || !labels || (n > 1 && !Z)) return VBX_ERR_INVALID;
    for (int64_t k = 0; k < n - 1; ++k) {                      // children exist before their parent, ids in range
        const double a = Z[4 * k], b = Z[4 * k + 1];
        if (!(a >= 0 && b >= 0 && a < (double)(n + k) && b < (double)(n + k))) return VBX_ERR_INVALID;
    }
    try {
        vbx::fcluster_distance(n, Z, t, labels);
    } catch (const std::bad_alloc&) {
        return VBX_ERR_HIP - 100;
    }
    return VBX_OK;
}

int vbx_scores_linkage_average(vbx_scores* sc, int64_t T, double* Z) {
    if (!sc || !Z || T < 1 || (long long)T * T != sc->n || T > 0x7fffffffLL / 2) return VBX_ERR_INVALID;
    if (T == 1) return VBX_OK;
    vbx_ctx* ctx = sc->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // Two ways over the matrix (vbx_ahc.hpp):
    //   rounds  all reciprocal nearest-neighbour pairs of the current matrix merged at once, the whole chip on every pass
    //           (VBX_AMD_LINKAGE_DEVICE=rounds, the default from kRoundsFrom clusters): a few dozen rounds for 10 000 x-vectors
    //   chain   SciPy's nearest-neighbour chain on ONE persistent workgroup, bit for bit the host routine (=chain): it
    //           finishes what the rounds leave (the last kRoundsStop = 48 clusters, where a round is all launch latency: handing
    //           over at 384 / 128 / 48 / 16 clusters measured 7.6 / 5.9 / 5.8 / 6.0 ms at T = 10 000 and 3.1 / 1.4 / 1.2 / 1.1 ms
    //           at T = 1025) and is the reference the rounds are tested against
    // The chain runs in stages of n/4 merges with a compaction of the live rows and columns in between; below kStageMin
    // clusters the rest runs in one stage (there a merge costs its four round trips, not the bytes of a row).
    static const long long kStageMin = [] { const char* e = experiment_env("VBX_AMD_LINKAGE_STAGE_MIN"); const long long v = e ? atoll(e) : 0; return v >= 64 ? v : 4096LL; }();
    static const bool staged = [] { const char* e = experiment_env("VBX_AMD_LINKAGE_STAGES"); return !(e && e[0] == '0'); }();
    const char* dev_mode = experiment_env("VBX_AMD_LINKAGE_DEVICE");           // (read per call: tests compare the two in one process)
    const bool rounds_on = !(dev_mode && std::strcmp(dev_mode, "chain") == 0);
    static const long long kRoundsFrom = [] { const char* e = experiment_env("VBX_AMD_LINKAGE_ROUNDS_FROM"); const long long v = e ? atoll(e) : 0; return v >= 4 ? v : 256LL; }();
    static const long long kRoundsStop = [] { const char* e = experiment_env("VBX_AMD_LINKAGE_ROUNDS_STOP"); const long long v = e ? atoll(e) : 0; return v >= 2 ? v : 48LL; }();
    int *d_size = nullptr, *d_size2 = nullptr, *d_chain = nullptr, *d_orig = nullptr, *d_orig2 = nullptr, *d_old = nullptr,
        *d_newidx = nullptr, *d_state = nullptr, *d_nn = nullptr, *d_role = nullptr;
    double *d_alt = nullptr, *d_cmp = nullptr, *d_nnd = nullptr;
    vbx::ChainMergeDev* d_merges = nullptr;
    vbx::RnnPair* d_pairs = nullptr;
    const bool rounds = rounds_on && T >= kRoundsFrom;
    int rc = dmalloc(ctx, &d_size, (size_t)T);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_chain, (size_t)T);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_orig, (size_t)T);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_state, (size_t)4);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_merges, (size_t)(T - 1));
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_size2, (size_t)T);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_orig2, (size_t)T);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_old, (size_t)T);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_newidx, (size_t)T);
    if (rc == VBX_OK && rounds) {
        rc = dmalloc(ctx, &d_nn, (size_t)T);
        if (rc == VBX_OK) rc = dmalloc(ctx, &d_nnd, (size_t)T);
        if (rc == VBX_OK) rc = dmalloc(ctx, &d_role, (size_t)T);
        if (rc == VBX_OK) rc = dmalloc(ctx, &d_pairs, (size_t)(T / 2 + 1));
    }
    std::vector<vbx::ChainMerge> merges((size_t)(T - 1));
    static_assert(sizeof(vbx::ChainMerge) == sizeof(vbx::ChainMergeDev), "merge record layout");
    hipError_t e = hipSuccess;
    if (rc == VBX_OK) {
        hipStream_t st = ctx->stream;
        std::vector<int> ones((size_t)T, 1), iota((size_t)T);
        for (long long i = 0; i < T; ++i) iota[(size_t)i] = (int)i;
        const int zero4[4] = {0, 0, 0, 0};
        e = hipMemcpyAsync(d_size, ones.data(), sizeof(int) * (size_t)T, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_orig, iota.data(), sizeof(int) * (size_t)T, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_state, zero4, sizeof(zero4), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);                 // (the host vectors go out of scope below)
        if (e == hipSuccess) {
            hipLaunchKernelGGL(vbx::linkage_prepare_kernel, dim3((unsigned)T), dim3(256), 0, st, sc->d_s, (long long)T);
            double* cur = sc->d_s;
            int *size_c = d_size, *size_a = d_size2, *orig_c = d_orig, *orig_a = d_orig2;
            long long n_cur = T, done = 0;
            auto compact_into = [&](double* dst, long long n_new) {       // the live clusters move up, in order
                hipLaunchKernelGGL(vbx::linkage_compact_index_kernel, dim3(1), dim3(1024), 0, st, (int)n_cur, size_c, orig_c,
                                   d_chain, d_state, size_a, orig_a, d_old, d_newidx);
                hipLaunchKernelGGL(vbx::linkage_compact_matrix_kernel, dim3((unsigned)n_new), dim3(256), 0, st, cur, (int)n_cur,
                                   dst, (int)n_new, d_old);
                std::swap(size_c, size_a);
                std::swap(orig_c, orig_a);
                n_cur = n_new;
            };
            if (rounds) {
                // rounds of reciprocal pairs on the matrix as it lies (dead rows and columns are skipped, not removed: a
                // round reads n_live x n entries), until few clusters are left or a round hardly merges anything
                int stalled = 0;
                while (e == hipSuccess && T - done > kRoundsStop && stalled < 3) {
                    const int n = (int)T;
                    hipLaunchKernelGGL(vbx::rnn_rowmin_kernel, dim3((unsigned)n), dim3(256), 0, st, cur, n, size_c, d_nn, d_nnd);
                    hipLaunchKernelGGL(vbx::rnn_pairs_kernel, dim3(1), dim3(1024), 0, st, n, size_c, orig_c, d_nn, d_nnd, d_pairs,
                                       d_role, d_merges, d_state);
                    int st_host[4] = {0, 0, 0, 0};
                    e = hipMemcpyAsync(st_host, d_state, sizeof st_host, hipMemcpyDeviceToHost, st);
                    if (e == hipSuccess) e = hipStreamSynchronize(st);
                    if (e != hipSuccess) break;
                    const int np = st_host[3];
                    if (np <= 0) break;                                    // (cannot happen: the smallest pair is reciprocal)
                    hipLaunchKernelGGL(vbx::rnn_rows_kernel, dim3((unsigned)np), dim3(256), 0, st, cur, n, size_c, d_pairs);
                    hipLaunchKernelGGL(vbx::rnn_cols_kernel, dim3((unsigned)n), dim3(256), 0, st, cur, n, size_c, d_role, d_pairs, d_state);
                    hipLaunchKernelGGL(vbx::rnn_canon_kernel, dim3((unsigned)np), dim3(256), 0, st, cur, n, d_pairs, d_state);
                    hipLaunchKernelGGL(vbx::rnn_sizes_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, size_c, d_pairs, d_state);
                    stalled = ((long long)np * 64 < T - done) ? stalled + 1 : 0;
                    done = st_host[2];
                }
                if (e == hipSuccess && done < T - 1) {                     // what is left goes to the chain, compacted
                    const long long n_new = T - done;
                    rc = dmalloc(ctx, &d_cmp, (size_t)(n_new * n_new));
                    if (rc == VBX_OK) {
                        compact_into(d_cmp, n_new);
                        cur = d_cmp;
                    }
                }
            }
            if (rc == VBX_OK && e == hipSuccess && done < T - 1) {
                const bool stages = staged && n_cur >= 2 * kStageMin;
                const long long n_alt = stages ? n_cur - n_cur / 4 : 0;
                if (stages) rc = dmalloc(ctx, &d_alt, (size_t)(n_alt * n_alt));
                double* alt = d_alt;
                while (rc == VBX_OK && done < T - 1) {
                    const long long remaining = T - 1 - done;
                    const long long m = (stages && n_cur >= 2 * kStageMin) ? std::min(remaining, n_cur / 4) : remaining;
                    hipLaunchKernelGGL(vbx::nn_chain_kernel, dim3(1), dim3(1024), 0, st, cur, (int)n_cur, size_c, d_chain, orig_c,
                                       d_state, d_merges, (int)done, (int)(done + m));
                    done += m;
                    if (done < T - 1) {
                        compact_into(alt, n_cur - m);
                        std::swap(cur, alt);
                    }
                }
            }
            if (e == hipSuccess) e = hipGetLastError();
        }
        if (rc == VBX_OK && e == hipSuccess) e = hipMemcpyAsync(merges.data(), d_merges, sizeof(vbx::ChainMerge) * merges.size(), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    for (void* p : {(void*)d_size2, (void*)d_orig, (void*)d_orig2, (void*)d_old, (void*)d_newidx, (void*)d_state, (void*)d_alt,
                    (void*)d_cmp, (void*)d_nn, (void*)d_nnd, (void*)d_role, (void*)d_pairs})
        ctx_free(ctx, p);
    ctx_free(ctx, d_size);
    ctx_free(ctx, d_chain);
    ctx_free(ctx, d_merges);
    if (rc != VBX_OK) return rc;
    if (e != hipSuccess) FAIL(ctx, VBX_ERR_HIP, "device linkage failed: %s", hipGetErrorString(e));
    vbx::finish_linkage(T, merges.data(), Z);
    return VBX_OK;
}

int vbx_scores_get_condensed(vbx_scores* sc, int64_t T, double scale, double* out) {
    if (!sc) return VBX_ERR_INVALID;
    vbx_ctx* ctx = sc->ctx;
    if (!out || T < 1 || (long long)T * T != sc->n) FAIL(ctx, VBX_ERR_INVALID, "vbx_scores_get_condensed: the scores are not a %lld x %lld matrix", (long long)T, (long long)T);
    if (T == 1) return VBX_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t m = (size_t)T * (size_t)(T - 1) / 2;
    double* d_c = nullptr;
    size_t c_bytes = 0;
    int rc = scratch_get(ctx, &d_c, m, &c_bytes);
    if (rc != VBX_OK) return rc;
    hipLaunchKernelGGL(vbx::condense_kernel, dim3((unsigned)(T - 1)), dim3(256), 0, ctx->stream, sc->d_s, d_c, (long long)T, scale);
    hipError_t e = hipMemcpyAsync(out, d_c, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    scratch_put(ctx, d_c, c_bytes);
    if (e != hipSuccess) {
        ctx->err = std::string("vbx_scores_get_condensed: ") + hipGetErrorString(e);
        return VBX_ERR_HIP;
    }
    return VBX_OK;
}

int vbx_scores_two_gmm_calib(vbx_scores* sc, int32_t niters, double* threshold, double* llr) {
    if (!sc) return VBX_ERR_INVALID;
    vbx_ctx* ctx = sc->ctx;
    if (niters < 1 || !threshold) FAIL(ctx, VBX_ERR_INVALID, "vbx_scores_two_gmm_calib: niters must be >= 1");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int nb = (int)std::min<long long>(vbx::kGmmPartials, (sc->n + 255) / 256);
    double *d_par = nullptr, *d_part = nullptr, *d_llr = nullptr;
    size_t par_bytes = 0, part_bytes = 0, llr_bytes = 0;
    int rc = scratch_get(ctx, &d_par, 16, &par_bytes);
    if (rc == VBX_OK) rc = scratch_get(ctx, &d_part, (size_t)nb * 6, &part_bytes);
    if (rc == VBX_OK && llr) rc = scratch_get(ctx, &d_llr, (size_t)sc->n, &llr_bytes);
    double par[16];
    hipError_t e = hipSuccess;
    if (rc == VBX_OK) {
        hipLaunchKernelGGL((vbx::gmm_moment_kernel<0>), dim3(nb), dim3(256), 0, st, sc->d_s, sc->n, d_par, d_part);
        hipLaunchKernelGGL(vbx::gmm_init_kernel, dim3(1), dim3(256), 0, st, d_part, nb, sc->n, d_par, 0);
        hipLaunchKernelGGL((vbx::gmm_moment_kernel<1>), dim3(nb), dim3(256), 0, st, sc->d_s, sc->n, d_par, d_part);
        hipLaunchKernelGGL(vbx::gmm_init_kernel, dim3(1), dim3(256), 0, st, d_part, nb, sc->n, d_par, 1);
        for (int it = 0; it < niters; ++it) {
            hipLaunchKernelGGL(vbx::gmm_pass_kernel, dim3(nb), dim3(256), 0, st, sc->d_s, sc->n, d_par, d_part);
            hipLaunchKernelGGL(vbx::gmm_update_kernel, dim3(1), dim3(256), 0, st, d_part, nb, d_par);
        }
        if (llr) hipLaunchKernelGGL(vbx::gmm_llr_kernel, dim3(nb), dim3(256), 0, st, sc->d_s, sc->n, d_par, d_llr);
        e = hipMemcpyAsync(par, d_par, sizeof(double) * 16, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess && llr) e = hipMemcpy(llr, d_llr, sizeof(double) * (size_t)sc->n, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) {
            ctx->err = std::string("twoGMMcalib kernels failed: ") + hipGetErrorString(e);
            rc = VBX_ERR_HIP;
        }
    }
    if (rc == VBX_OK) {
        // diarization_lib.py:30 with the final weights / means / var
        const double w0 = par[0], w1 = par[1], m0 = par[2], m1 = par[3], var = par[4];
        const double t0 = std::log(w0 * w0 / var) - m0 * m0 / var, t1 = std::log(w1 * w1 / var) - m1 * m1 / var;
        *threshold = -0.5 * (t0 - t1) / (m0 / var - m1 / var);
    }
    scratch_put(ctx, d_par, par_bytes);
    scratch_put(ctx, d_part, part_bytes);
    scratch_put(ctx, d_llr, llr_bytes);
    return rc;
}

}  // extern "C"
