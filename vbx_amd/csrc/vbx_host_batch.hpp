// vbx_host_batch.hpp -- C ABI (include/vbx_hip.h): context, and a batch on ONE stream -- create, options, upload, run, results
// (one translation unit with vbx_capi.hip, which includes the parts in order; not a stand-alone header)
#pragma once
// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

int vbx_abi_version(void) { return VBX_ABI_VERSION; }

const char* vbx_last_error(const vbx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int vbx_create(vbx_ctx** out, int device) {
    if (!out) return VBX_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("no HIP device visible: ") + hipGetErrorString(e);
        return VBX_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        g_create_error = "device index out of range";
        return VBX_ERR_INVALID;
    }
    vbx_ctx* ctx = new vbx_ctx();
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&ctx->prop, device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("device setup failed: ") + hipGetErrorString(e);
        delete ctx;
        return VBX_ERR_HIP;
    }
    if (std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("libvbx_hip.so is built for gfx950 only; device reports ") + ctx->prop.gcnArchName;
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return VBX_ERR_NO_DEVICE;
    }
    *out = ctx;
    return VBX_OK;
}

int vbx_destroy(vbx_ctx* ctx) {
    if (!ctx) return VBX_OK;
    (void)hipSetDevice(ctx->device);
    for (auto& sp : ctx->spare) (void)hipFree(sp.first);
    for (auto& gs : ctx->group_streams) (void)hipStreamDestroy(gs.first);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return VBX_OK;
}

int vbx_device_info(vbx_ctx* ctx, char* name, int cap, int* compute_units, int64_t* hbm_bytes) {
    if (!ctx) return VBX_ERR_INVALID;
    if (name && cap > 0) {
        std::snprintf(name, cap, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    }
    if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
    return VBX_OK;
}

static int leaf_destroy(vbx_batch* b) {
    if (!b) return VBX_OK;
    (void)hipSetDevice(b->ctx->device);
    void* ptrs[] = {b->d_recs, b->d_state, b->d_tile_rec, b->d_tile_t0, b->d_tile_desc, b->d_tile_done, b->d_phi, b->d_sqrt_phi, b->d_gtile,
                    b->d_rho, b->d_gamma, b->d_bmat, b->d_mrow, b->d_ahat, b->d_bhat, b->d_alpha, b->d_invL,
                    b->d_bias, b->d_mpart, b->d_npart, b->d_lraw, b->d_emodel, b->d_pi, b->d_epart, b->d_Li,
                    b->d_xstage, b->d_ip, b->d_fw_scale, b->d_bw_scale, b->d_op, b->d_fbound, b->d_gbound,
                    b->d_opexp, b->d_tllpart, b->d_sfw, b->d_dump, b->d_sop, b->d_sopexp, b->d_sup_rec, b->d_sup_idx,
                    b->d_sop2, b->d_sopexp2, b->d_sup2_rec, b->d_sup2_idx,
                    b->d_gamma0, b->d_pi_prev, b->d_oph, b->d_ophexp, b->d_cop, b->d_lppow, b->d_tile_order,
                    b->d_rho_a, b->d_rho_b, b->d_alpha_frag, b->d_rho_e, b->d_rho_amax, b->d_alpha_e};
    (void)hipStreamSynchronize(b->ctx->stream);               // nothing of this batch may still be running when its
    for (void* p : ptrs) ctx_free(b->ctx, p);                 // blocks go back to the spare list
    if (b->d_fetch) ctx_free(b->ctx, b->d_fetch);
    if (b->h_args) (void)vbx_host_free(b->h_args);
    if (b->h_poll) (void)vbx_host_free(b->h_poll);
    if (b->ev_start) (void)hipEventDestroy(b->ev_start);
    if (b->ev_stop) (void)hipEventDestroy(b->ev_stop);
    for (auto& ep : b->ev_pool) {
        (void)hipEventDestroy(ep.a);
        (void)hipEventDestroy(ep.b);
    }
    delete b;
    return VBX_OK;
}

static int leaf_create(vbx_ctx* ctx, int n_rec, const int64_t* T, const int32_t* S, int32_t D, int precision,
                     int max_iters, vbx_batch** out) {
    if (!ctx) return VBX_ERR_INVALID;
    if (!out || !T || !S || n_rec <= 0 || D <= 0 || max_iters < 0) FAIL(ctx, VBX_ERR_INVALID, "vbx_batch_create: bad argument");
    if (precision != VBX_PREC_FP32 && precision != VBX_PREC_FP64) FAIL(ctx, VBX_ERR_INVALID, "unknown precision %d", precision);
    *out = nullptr;
    int smax = 0;
    for (int i = 0; i < n_rec; ++i) {
        if (T[i] <= 0 || T[i] > 0x7fffffffLL / 512) FAIL(ctx, VBX_ERR_INVALID, "recording %d: T=%lld out of range", i, (long long)T[i]);
        if (S[i] <= 0) FAIL(ctx, VBX_ERR_INVALID, "recording %d: S=%d", i, S[i]);
        smax = std::max(smax, (int)S[i]);
    }
    if (smax > VBX_MAX_SPEAKERS)
        FAIL(ctx, VBX_ERR_UNSUPPORTED, "S=%d exceeds VBX_MAX_SPEAKERS=%d", smax, VBX_MAX_SPEAKERS);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    vbx_batch* b = new vbx_batch();
    b->ctx = ctx;
    b->n_rec = n_rec;
    b->D = D;
    b->Dp = round_up(D, 32);
    int sp = 16;
    while (sp < smax) sp *= 2;
    b->Sp = sp;
    b->NT = sp / 16;
    b->precision = precision;
    b->rsize = precision == VBX_PREC_FP64 ? 8 : 4;
    b->max_iters = max_iters;
    b->recs.resize(n_rec);
    b->is_set.assign(n_rec, 0);
    b->split_dirty.assign(n_rec, 1);
    std::vector<int> tile_rec, tile_t0;
    long long row = 0;
    long long maxT = 0;
    for (int i = 0; i < n_rec; ++i) {
        RecDesc& rd = b->recs[i];
        std::memset(&rd, 0, sizeof rd);
        rd.row0 = rd.rho_row0 = row;
        rd.T = (int)T[i];
        rd.S = S[i];
        rd.tile0 = rd.rho_tile0 = (int)tile_rec.size();
        rd.rho_rec = i;
        rd.ntiles = (rd.T + kTileFrames - 1) / kTileFrames;
        for (int tl = 0; tl < rd.ntiles; ++tl) {
            tile_rec.push_back(i);
            tile_t0.push_back(tl * kTileFrames);
        }
        row += rd.T;
        maxT = std::max<long long>(maxT, rd.T);
    }
    b->sum_T = row;
    b->ntiles_total = b->nblocks_chunk = (int)tile_rec.size();
    b->share_src.resize(n_rec);
    for (int i = 0; i < n_rec; ++i) b->share_src[i] = i;
    std::vector<int4> tile_desc;
    for (int t = 0; t < b->ntiles_total; ++t) {
        const RecDesc& rd = b->recs[tile_rec[t]];
        tile_desc.push_back(make_int4(tile_rec[t], tile_t0[t], std::min(kTileFrames, rd.T - tile_t0[t]), (int)(rd.row0 + tile_t0[t])));
    }
    while (tile_desc.size() % 4) tile_desc.push_back(make_int4(0, 0, 0, (int)row));
    const size_t rs = b->rsize;
    const size_t cells = (size_t)b->sum_T * b->Sp;
    int rc = VBX_OK;
#define ALLOC(expr) if (rc == VBX_OK) rc = (expr)
    ALLOC(dmalloc(ctx, &b->d_recs, n_rec));
    ALLOC(dmalloc(ctx, &b->d_state, (size_t)2 * n_rec));
    ALLOC(dmalloc(ctx, &b->d_tile_rec, b->ntiles_total));
    ALLOC(dmalloc(ctx, &b->d_tile_t0, b->ntiles_total));
    const int ntiles_pad = (b->ntiles_total + 3) / 4 * 4;
    ALLOC(dmalloc(ctx, &b->d_tile_desc, ntiles_pad));
    ALLOC(dmalloc(ctx, &b->d_tile_done, ntiles_pad));
    ALLOC(dmalloc(ctx, &b->d_phi, (size_t)n_rec * b->Dp));
    ALLOC(dmalloc(ctx, &b->d_sqrt_phi, b->Dp));
    ALLOC(dmalloc(ctx, &b->d_gtile, b->ntiles_total));
    // (one tile of zero rows after the last recording: kernels may read whole tiles past its end)
    ALLOC(dmalloc_bytes(ctx, &b->d_rho, ((size_t)b->sum_T + kTileFrames) * b->Dp * rs));
    ALLOC(dmalloc_bytes(ctx, &b->d_gamma, cells * rs));
    ALLOC(dmalloc_bytes(ctx, &b->d_bmat, (cells + (size_t)kTileFrames * b->Sp) * rs));     // (+ one tile, like rho)
    ALLOC(dmalloc_bytes(ctx, &b->d_mrow, (size_t)b->sum_T * rs));
    ALLOC(dmalloc_bytes(ctx, &b->d_alpha, (size_t)2 * n_rec * b->Sp * b->Dp * rs));     // (two copies: fin_kernel)
    ALLOC(dmalloc_bytes(ctx, &b->d_invL, (size_t)2 * n_rec * b->Sp * b->Dp * rs));
    ALLOC(dmalloc_bytes(ctx, &b->d_bias, (size_t)4 * n_rec * b->Sp * rs));              // (two copies of bias, then two of bias_lo)
    ALLOC(dmalloc_bytes(ctx, &b->d_mpart, (size_t)b->ntiles_total * b->Sp * b->Dp * rs));
    ALLOC(dmalloc_bytes(ctx, &b->d_npart, (size_t)b->ntiles_total * b->Sp * rs));
    ALLOC(dmalloc(ctx, &b->d_emodel, (size_t)2 * n_rec * b->Sp));
    ALLOC(dmalloc(ctx, &b->d_pi, (size_t)n_rec * b->Sp));
    ALLOC(dmalloc(ctx, &b->d_pi_prev, (size_t)n_rec * b->Sp));
    ALLOC(dmalloc_bytes(ctx, &b->d_gamma0, (size_t)n_rec * b->Sp * rs));
    ALLOC(dmalloc(ctx, &b->d_epart, (size_t)b->ntiles_total * b->Sp));
    ALLOC(dmalloc(ctx, &b->d_Li, (size_t)n_rec * std::max(max_iters, 1)));
    b->xstage_bytes = (size_t)maxT * D * 8;
    ALLOC(dmalloc_bytes(ctx, &b->d_xstage, b->xstage_bytes));
#undef ALLOC
    if (rc != VBX_OK) {
        leaf_destroy(b);
        return rc;
    }
    hipError_t e;
    if ((e = hipMemcpy(b->d_tile_rec, tile_rec.data(), sizeof(int) * tile_rec.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(b->d_tile_t0, tile_t0.data(), sizeof(int) * tile_t0.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(b->d_tile_desc, tile_desc.data(), sizeof(int4) * tile_desc.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemset(b->d_tile_done, 0, sizeof(int) * ntiles_pad)) != hipSuccess ||
        (e = hipMemset(b->d_pi_prev, 0, sizeof(double) * (size_t)n_rec * b->Sp)) != hipSuccess ||
        (e = hipMemset((char*)b->d_bmat + cells * rs, 0, (size_t)kTileFrames * b->Sp * rs)) != hipSuccess ||
        (e = hipMemset(b->d_state, 0, sizeof(RecState) * 2 * n_rec)) != hipSuccess ||
        (e = hipMemset(b->d_gamma, 0, cells * rs)) != hipSuccess ||
        (e = hipMemset((char*)b->d_rho + (size_t)b->sum_T * b->Dp * rs, 0, (size_t)kTileFrames * b->Dp * rs)) != hipSuccess ||
        (e = hipDeviceSynchronize()) != hipSuccess ||      // null-stream memsets vs. our non-blocking stream
        (e = hipEventCreate(&b->ev_start)) != hipSuccess || (e = hipEventCreate(&b->ev_stop)) != hipSuccess) {
        ctx->err = std::string("batch initialisation failed: ") + hipGetErrorString(e);
        leaf_destroy(b);
        return VBX_ERR_HIP;
    }
    *out = b;
    return VBX_OK;
}

static int sync_uploads(vbx_batch* b);

static int leaf_set_option(vbx_batch* b, int option, int64_t value) {
    if (!b) return VBX_ERR_INVALID;
    switch (option) {
        case VBX_OPT_FB_ALGO:
            if (value < VBX_FB_AUTO || value > VBX_FB_CHUNKED) FAIL(b->ctx, VBX_ERR_INVALID, "bad fb algo");
            b->fb_algo = (int)value;
            return VBX_OK;
        case VBX_OPT_CHECK_EVERY:
            if (value < 1) FAIL(b->ctx, VBX_ERR_INVALID, "check_every must be >= 1");
            b->check_every = (int)value;
            return VBX_OK;
        case VBX_OPT_PROFILE:
            b->profile = value == 1 ? ((int64_t)1 << VBX_K_COUNT) - 1 : value < 0 ? 0 : (value >> 1);
            return VBX_OK;
        case VBX_OPT_FUSE:
            if (value < 0 || value > 2) FAIL(b->ctx, VBX_ERR_INVALID, "fuse must be 0, 1 or 2");
            b->fuse = (int)value;
            b->mpart_valid = false;
            return VBX_OK;
        case VBX_OPT_SPLIT_TILES:
            if (value < 0 || value > 2) FAIL(b->ctx, VBX_ERR_INVALID, "split_tiles must be 0 (auto), 1 (on) or 2 (off)");
            b->split_tiles = (int)value;
            return VBX_OK;
        case VBX_OPT_TWO_LEVEL_FROM:
            if (value < 2) FAIL(b->ctx, VBX_ERR_INVALID, "two-level threshold must be >= 2 chunks");
            b->two_level_from = (int)value;
            return VBX_OK;
        case VBX_OPT_SCAN_GROUP:
            if (value < 0 || value > 4096) FAIL(b->ctx, VBX_ERR_INVALID, "scan group must be in [0, 4096]");
            b->scan_group = (int)value;
            return VBX_OK;
        case VBX_OPT_SCAN_GROUP2:
            if (value < 0 || value > 4096) FAIL(b->ctx, VBX_ERR_INVALID, "level-2 scan group must be in [0, 4096]");
            b->scan_group2 = (int)value;
            return VBX_OK;
        case VBX_OPT_THREE_LEVEL_FROM:
            if (value < 4) FAIL(b->ctx, VBX_ERR_INVALID, "three-level threshold must be >= 4 chunks");
            b->three_level_from = (int)value;
            return VBX_OK;
        case VBX_OPT_CHUNK_FRAMES:
            if (value < 0) FAIL(b->ctx, VBX_ERR_INVALID, "chunk_frames must be >= 0");
            b->chunk_frames = (int)value;
            return VBX_OK;
        case VBX_OPT_GEMM:
            if (value != VBX_GEMM_EXACT && value != VBX_GEMM_SPLIT) FAIL(b->ctx, VBX_ERR_INVALID, "VBX_OPT_GEMM takes VBX_GEMM_EXACT or VBX_GEMM_SPLIT");
#ifdef VBX_ISA_UNAUDITED
            // vbx_amd/build.py could not disassemble this library (no llvm-objdump, or VBX_AMD_SKIP_ISA_AUDIT): it may hold the
            // packed-f32 operand form that misreads src1 beside the K = 32 f16 matrix instructions
            if (value == VBX_GEMM_SPLIT) FAIL(b->ctx, VBX_ERR_UNSUPPORTED, "VBX_GEMM_SPLIT: this library was built without the ISA audit (vbx_amd/build.py); rebuild with llvm-objdump available");
#endif
            b->gemm = (int)value;
            return VBX_OK;
        case VBX_OPT_ASYNC_UPLOAD:
            if (value != 0 && value != 1) FAIL(b->ctx, VBX_ERR_INVALID, "VBX_OPT_ASYNC_UPLOAD takes 0 or 1");
            if (!value && b->async_upload)
                if (int rc = sync_uploads(b); rc != VBX_OK) return rc;
            b->async_upload = value != 0;
            return VBX_OK;
        default: FAIL(b->ctx, VBX_ERR_INVALID, "unknown option %d", option);
    }
}

// Uploads (round 6).  A setter only ENQUEUES: the small per-recording arguments travel through a pinned host block of the
// batch (so their copies are asynchronous and nothing on the host goes out of scope under them), x-vectors and initial
// responsibilities are copied as the caller holds them and converted / padded on the device (prep_kernel, pad_gamma_kernel),
// and sum_t G_t -- the one value that comes BACK from an upload -- stays in d_gtile until sync_uploads fetches it for all
// recordings at once.  Without VBX_OPT_ASYNC_UPLOAD every setter ends with sync_uploads (the contract of ABI <= 6: the
// caller's buffers are free when the call returns); with it the synchronize happens once, when the next run begins
// (vbx_batch_run) or in vbx_batch_sync_uploads, and the caller's X / gamma0 must stay valid until then.
static int ensure_host_args(vbx_batch* b) {
    if (b->h_args) return VBX_OK;
    const size_t per = (size_t)2 * b->Dp + b->Sp + 2;         // {Phi, sqrt Phi, pi0, flags: Phi is here, pi0 is here}
    // (from the process-wide pool of pinned blocks, vbx_host_alloc: pinning and unpinning a block per batch cost a single VBx() call
    //  of the example recording 0.25 of its 1.2 ms)
    if (vbx_host_alloc(sizeof(double) * per * b->n_rec, (void**)&b->h_args) != VBX_OK) FAIL(b->ctx, VBX_ERR_HIP, "pinned argument block: %s", g_create_error.c_str());
    std::memset(b->h_args, 0, sizeof(double) * per * b->n_rec);
    b->gsum_pending.assign(b->n_rec, 0);
    b->args_busy.assign(b->n_rec, 0);
    return VBX_OK;
}

static int sync_uploads(vbx_batch* b) {
    vbx_ctx* ctx = b->ctx;
    bool pending = false;
    for (char c : b->gsum_pending) pending = pending || c;
    if (!b->uploads_in_flight && !pending) return VBX_OK;
    if (b->h_args) {                                          // Phi / pi0 of the recordings set since: one launch (scatter_args_kernel)
        const int per = 2 * b->Dp + b->Sp + 2;
        hipLaunchKernelGGL(vbx::scatter_args_kernel, dim3(b->n_rec), dim3(256), 0, ctx->stream, (const double*)b->h_args, per, b->Dp, b->Sp,
                           b->d_phi, b->d_pi);
        HIPCHK(ctx, hipGetLastError());
    }
    std::vector<double> gt;
    if (pending) {
        gt.resize(b->ntiles_total);
        HIPCHK(ctx, hipMemcpyAsync(gt.data(), b->d_gtile, sizeof(double) * gt.size(), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    for (int i = 0; i < b->n_rec && pending; ++i) {
        if (!b->gsum_pending[i]) continue;
        const RecDesc& rd = b->recs[i];
        double gsum = 0.0;
        for (int tl = 0; tl < rd.ntiles; ++tl) gsum += gt[rd.tile0 + tl];
        b->recs[i].gsum = gsum;
        b->gsum_pending[i] = 0;
    }
    for (int i = 0; i < b->n_rec; ++i)                        // recordings that read another one's rows: its sum G as well
        if (b->share_src[i] != i) b->recs[i].gsum = b->recs[b->share_src[i]].gsum;
    if (b->h_args) {
        const size_t per = (size_t)2 * b->Dp + b->Sp + 2;
        for (int i = 0; i < b->n_rec; ++i) b->h_args[(size_t)i * per + per - 2] = b->h_args[(size_t)i * per + per - 1] = 0.0;
    }
    std::fill(b->args_busy.begin(), b->args_busy.end(), 0);
    b->uploads_in_flight = false;
    b->recs_dirty = true;
    return VBX_OK;
}

extern "C++" {
namespace {
template <typename R>
int set_recording_impl(vbx_batch* b, int rec, const void* X, int x_dtype, const double* Phi, const double* pi0,
                       const void* gamma0, int g_dtype, const double* alpha0, const double* invL0) {
    vbx_ctx* ctx = b->ctx;
    RecDesc& rd = b->recs[rec];
    const int D = b->D, Dp = b->Dp, Sp = b->Sp, S = rd.S;
    const long long T = rd.T;
    if (int rc = ensure_host_args(b); rc != VBX_OK) return rc;
    // (the pinned slot of this recording may still be the source of a copy in flight: the same recording set twice in a row)
    if (b->args_busy[rec])
        if (int rc = sync_uploads(b); rc != VBX_OK) return rc;
    const size_t per = (size_t)2 * Dp + Sp + 2;
    double* const phi = b->h_args + (size_t)rec * per;
    double* const sphi = phi + Dp;
    double* const pip = phi + 2 * Dp;
    double* const flags = phi + 2 * Dp + Sp;                  // what sync_uploads' scatter_args_kernel takes from this slot
    bool must_sync = !b->async_upload;
    if (X) {
        for (int d = 0; d < Dp; ++d) phi[d] = sphi[d] = 0.0;     // (padded dims: 0)
        for (int d = 0; d < D; ++d) {
            if (!(Phi[d] > 0.0)) FAIL(ctx, VBX_ERR_INVALID, "Phi[%d] must be positive", d);
            phi[d] = Phi[d];
            sphi[d] = std::sqrt(Phi[d]);
        }
        flags[0] = 1.0;                                       // (Phi goes to the device with the scatter of sync_uploads)
        // X -> staging -> rho, G; prep_kernel reads sqrt(Phi) from the pinned slot itself
        const size_t xbytes = (size_t)T * D * (x_dtype == VBX_F64 ? 8 : 4);
        HIPCHK(ctx, hipMemcpyAsync(b->d_xstage, X, xbytes, hipMemcpyHostToDevice, ctx->stream));
        if (x_dtype == VBX_F64) launch_prep<R, double>(b, rd, sphi); else launch_prep<R, float>(b, rd, sphi);
        HIPCHK(ctx, hipGetLastError());
        b->gsum_pending[rec] = 1;
    } else {
        // shared rho: Phi (and with it sum_t G_t) of the recording this one shares its x-vectors with
        // (src == rec: a clone from another stream group's arena, leaf_set_recording_cloned has put Phi, rho and sum G in place)
        const int src = b->share_src[rec];
        flags[0] = 0.0;
        if (src != rec) {
            const double* sphi_src = b->h_args + (size_t)src * per;
            if (sphi_src[per - 2] != 0.0) {                   // (the source's Phi is itself still on its way: take it from its slot)
                std::memcpy(phi, sphi_src, sizeof(double) * 2 * Dp);
                flags[0] = 1.0;
            } else {
                HIPCHK(ctx, hipMemcpyAsync(b->d_phi + (size_t)rec * Dp, b->d_phi + (size_t)src * Dp, sizeof(double) * Dp, hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        if (src != rec) rd.gsum = b->recs[src].gsum;              // (if the source's is still on its way: sync_uploads copies it again)
        b->gsum_pending[rec] = 0;
    }
    // gamma0 (padded speakers: 0): the rows as the caller holds them -> staging (free again: prep_kernel is ahead in the
    // stream) -> padded and converted on the device
    std::vector<R> gp;
    const size_t gbytes = (size_t)T * S * (g_dtype == VBX_F64 ? 8 : 4);
    if (gbytes <= b->xstage_bytes) {
        HIPCHK(ctx, hipMemcpyAsync(b->d_xstage, gamma0, gbytes, hipMemcpyHostToDevice, ctx->stream));
        const unsigned blocks = (unsigned)(((size_t)T * Sp + 255) / 256);
        if (g_dtype == VBX_F64)
            hipLaunchKernelGGL((vbx::pad_gamma_kernel<R, double>), dim3(blocks), dim3(256), 0, ctx->stream, (const double*)b->d_xstage,
                               (R*)b->d_gamma + rd.row0 * Sp, T, S, Sp);
        else
            hipLaunchKernelGGL((vbx::pad_gamma_kernel<R, float>), dim3(blocks), dim3(256), 0, ctx->stream, (const float*)b->d_xstage,
                               (R*)b->d_gamma + rd.row0 * Sp, T, S, Sp);
        HIPCHK(ctx, hipGetLastError());
    } else {                                                   // (more speakers than feature dims: the staging block is too small)
        if (g_dtype == VBX_F64) pack_matrix<R, double>(gp, (const double*)gamma0, T, S, Sp, (R)0);
        else pack_matrix<R, float>(gp, (const float*)gamma0, T, S, Sp, (R)0);
        HIPCHK(ctx, hipMemcpyAsync((R*)b->d_gamma + rd.row0 * Sp, gp.data(), sizeof(R) * gp.size(), hipMemcpyHostToDevice, ctx->stream));
        must_sync = true;
    }
    for (int s = 0; s < Sp; ++s) pip[s] = s < S ? pi0[s] : 0.0;
    flags[1] = 1.0;
    std::vector<R> ap, ip;
    rd.has_model = (alpha0 && invL0) ? 1 : 0;
    if (rd.has_model) {
        ap.assign((size_t)Sp * Dp, (R)0);
        ip.assign((size_t)Sp * Dp, (R)1);
        for (int s = 0; s < S; ++s)
            for (int d = 0; d < D; ++d) {
                ap[(size_t)s * Dp + d] = (R)alpha0[(size_t)s * D + d];
                ip[(size_t)s * Dp + d] = (R)invL0[(size_t)s * D + d];
            }
        HIPCHK(ctx, hipMemcpyAsync((R*)b->d_alpha + (size_t)rec * Sp * Dp, ap.data(), sizeof(R) * ap.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync((R*)b->d_invL + (size_t)rec * Sp * Dp, ip.data(), sizeof(R) * ip.size(), hipMemcpyHostToDevice, ctx->stream));
        must_sync = true;
    }
    if (b->has_run) {                                         // (a fresh batch: leaf_create has zeroed the states and the flags)
        HIPCHK(ctx, hipMemsetAsync(b->d_state + rec, 0, sizeof(RecState), ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(b->d_state + b->n_rec + rec, 0, sizeof(RecState), ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(b->d_tile_done + rd.tile0, 0, sizeof(int) * rd.ntiles, ctx->stream));
    }
    b->args_busy[rec] = 1;
    b->uploads_in_flight = true;
    b->mirrors_valid = false;
    if (must_sync) return sync_uploads(b);                     // (host vectors go out of scope below)
    return VBX_OK;
}
}  // namespace
}  // extern "C++"

extern "C++" {
namespace {
// recording `rec` from rows already in HBM: fea [T][D] f64 (vbx_xvectors) and the AHC labels; the initial
// responsibilities are built on the device (vbhmm.py:150-152), pi0 = 1/S (VBx.py:76: pi given as an int)
template <typename R>
int set_recording_resident_impl(vbx_batch* b, int rec, const double* d_fea, const int32_t* labels, double hi, double lo,
                                const double* Phi) {
    vbx_ctx* ctx = b->ctx;
    RecDesc& rd = b->recs[rec];
    const int D = b->D, Dp = b->Dp, Sp = b->Sp, S = rd.S;
    const long long T = rd.T;
    std::vector<double> phi(Dp, 0.0), sphi(Dp, 0.0);
    for (int d = 0; d < D; ++d) {
        if (!(Phi[d] > 0.0)) FAIL(ctx, VBX_ERR_INVALID, "Phi[%d] must be positive", d);
        phi[d] = Phi[d];
        sphi[d] = std::sqrt(Phi[d]);
    }
    for (long long t = 0; t < T; ++t)
        if (labels[t] < 0 || labels[t] >= S) FAIL(ctx, VBX_ERR_INVALID, "label %d of frame %lld outside [0, %d)", labels[t], t, S);
    HIPCHK(ctx, hipMemcpyAsync(b->d_phi + (size_t)rec * Dp, phi.data(), sizeof(double) * Dp, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(b->d_sqrt_phi, sphi.data(), sizeof(double) * Dp, hipMemcpyHostToDevice, ctx->stream));
    {
        LaunchScope ls(b, VBX_K_PREP);
        hipLaunchKernelGGL((prep_kernel<R, double>), dim3(rd.ntiles), dim3(256), 0, ctx->stream, d_fea, (const double*)b->d_sqrt_phi,
                           (R*)b->d_rho + rd.row0 * Dp, b->d_gtile + rd.tile0, rd.T, D, Dp);
    }
    HIPCHK(ctx, hipGetLastError());
    std::vector<double> gt(rd.ntiles);
    HIPCHK(ctx, hipMemcpyAsync(gt.data(), b->d_gtile + rd.tile0, sizeof(double) * rd.ntiles, hipMemcpyDeviceToHost, ctx->stream));
    // labels -> staging (the x staging block is free: fea is already on the device) -> gamma
    if (b->xstage_bytes < sizeof(int32_t) * (size_t)T) FAIL(ctx, VBX_ERR_STATE, "staging block too small for the labels");
    HIPCHK(ctx, hipMemcpyAsync(b->d_xstage, labels, sizeof(int32_t) * (size_t)T, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL((vbx::qinit_kernel<R>), dim3((unsigned)((T * Sp + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const int*)b->d_xstage, (R*)b->d_gamma + rd.row0 * Sp, T, S, Sp, hi, lo);
    std::vector<double> pip(Sp, 0.0);
    for (int s = 0; s < S; ++s) pip[s] = 1.0 / S;
    HIPCHK(ctx, hipMemcpyAsync(b->d_pi + (size_t)rec * Sp, pip.data(), sizeof(double) * Sp, hipMemcpyHostToDevice, ctx->stream));
    rd.has_model = 0;
    RecState st;
    std::memset(&st, 0, sizeof st);
    HIPCHK(ctx, hipMemcpyAsync(b->d_state + rec, &st, sizeof st, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(b->d_state + b->n_rec + rec, &st, sizeof st, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(b->d_tile_done + rd.tile0, 0, sizeof(int) * rd.ntiles, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    double gsum = 0.0;
    for (double g : gt) gsum += g;
    rd.gsum = gsum;
    if (!b->gsum_pending.empty()) b->gsum_pending[rec] = 0;
    if (b->h_args) {                                          // (nothing of an earlier upload of this recording may follow)
        const size_t per = (size_t)2 * Dp + Sp + 2;
        b->h_args[(size_t)rec * per + per - 2] = b->h_args[(size_t)rec * per + per - 1] = 0.0;
    }
    b->mirrors_valid = false;
    return VBX_OK;
}

template <typename R>
int get_labels_impl(vbx_batch* b, int rec, int32_t* first, int32_t* second) {
    vbx_ctx* ctx = b->ctx;
    const RecDesc& rd = b->recs[rec];
    const long long T = rd.T;
    int* d_lab = nullptr;
    int rc = dmalloc(ctx, &d_lab, (size_t)2 * T);
    if (rc != VBX_OK) return rc;
    hipLaunchKernelGGL((vbx::top2_kernel<R>), dim3((unsigned)((T + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const R*)b->d_gamma + rd.row0 * b->Sp, d_lab, d_lab + T, T, rd.S, b->Sp);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && first) e = hipMemcpyAsync(first, d_lab, sizeof(int32_t) * (size_t)T, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && second) e = hipMemcpyAsync(second, d_lab + T, sizeof(int32_t) * (size_t)T, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    ctx_free(ctx, d_lab);
    if (e != hipSuccess) FAIL(ctx, VBX_ERR_HIP, "label extraction failed: %s", hipGetErrorString(e));
    return VBX_OK;
}
}  // namespace
}  // extern "C++"

struct vbx_xvectors;
static void own_rho(vbx_batch* b, int rec);
static const double* xvectors_fea_rows(const vbx_xvectors* xv, int64_t row0, int64_t T, int D, int device);

static int leaf_set_recording_resident(vbx_batch* b, int rec, const vbx_xvectors* xv, int64_t row0, const int32_t* labels,
                                       double init_smoothing, const double* Phi, double loopProb, double Fa, double Fb) {
    if (!b) return VBX_ERR_INVALID;
    vbx_ctx* ctx = b->ctx;
    if (rec < 0 || rec >= b->n_rec) FAIL(ctx, VBX_ERR_INVALID, "recording index %d out of range", rec);
    if (!xv || !labels || !Phi) FAIL(ctx, VBX_ERR_INVALID, "vbx_batch_set_recording_resident: NULL input");
    if (!(loopProb >= 0.0 && loopProb <= 1.0)) FAIL(ctx, VBX_ERR_INVALID, "loopProb=%g outside [0,1]", loopProb);
    if (!(Fa > 0.0) || !(Fb > 0.0)) FAIL(ctx, VBX_ERR_INVALID, "Fa and Fb must be positive");
    RecDesc& rd = b->recs[rec];
    const double* d_fea = xvectors_fea_rows(xv, row0, rd.T, b->D, ctx->device);
    if (!d_fea) FAIL(ctx, VBX_ERR_INVALID, "rows [%lld, +%d) x %d dims are not in the resident x-vectors", (long long)row0, rd.T, b->D);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rd.lp = loopProb;
    rd.Fa = Fa;
    rd.Fb = Fb;
    own_rho(b, rec);
    // softmax(smoothing * onehot) row (vbhmm.py:152, scipy.special.softmax: exp(x - max) / sum)
    const double z = std::exp(-init_smoothing), den = 1.0 + (rd.S - 1) * z;
    const double hi = 1.0 / den, lo = z / den;
    int rc = b->precision == VBX_PREC_FP64 ? set_recording_resident_impl<double>(b, rec, d_fea, labels, hi, lo, Phi)
                                            : set_recording_resident_impl<float>(b, rec, d_fea, labels, hi, lo, Phi);
    if (rc != VBX_OK) return rc;
    b->is_set[rec] = 1;
    b->recs_dirty = true;
    b->mpart_valid = false;
    return VBX_OK;
}

static int leaf_get_labels(vbx_batch* b, int rec, int32_t* first, int32_t* second) {
    if (!b) return VBX_ERR_INVALID;
    if (rec < 0 || rec >= b->n_rec) FAIL(b->ctx, VBX_ERR_INVALID, "recording index %d out of range", rec);
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    return b->precision == VBX_PREC_FP64 ? get_labels_impl<double>(b, rec, first, second)
                                         : get_labels_impl<float>(b, rec, first, second);
}

// `rec` gets (back) a rho of its own: recordings that read its rows so far are unset, and it leaves the group it was in
static void own_rho(vbx_batch* b, int rec) {
    for (int i = 0; i < b->n_rec; ++i)
        if (i != rec && b->share_src[i] == rec) {
            b->share_src[i] = i;
            b->recs[i].rho_row0 = b->recs[i].row0;
            b->recs[i].rho_tile0 = b->recs[i].tile0;
            b->recs[i].rho_rec = i;
            b->is_set[i] = 0;
            b->order_dirty = true;
        }
    if (b->share_src[rec] != rec) b->order_dirty = true;
    b->share_src[rec] = rec;
    b->recs[rec].rho_row0 = b->recs[rec].row0;
    b->recs[rec].rho_tile0 = b->recs[rec].tile0;
    b->recs[rec].rho_rec = rec;
    b->split_dirty[rec] = 1;                                  // (every caller is about to give `rec` new x-vectors)
}

static int leaf_set_recording(vbx_batch* b, int rec, const void* X, int x_dtype, const double* Phi, const double* pi0,
                            const void* gamma0, int g_dtype, const double* alpha0, const double* invL0,
                            double loopProb, double Fa, double Fb) {
    if (!b) return VBX_ERR_INVALID;
    vbx_ctx* ctx = b->ctx;
    if (rec < 0 || rec >= b->n_rec) FAIL(ctx, VBX_ERR_INVALID, "recording index %d out of range", rec);
    if (!X || !Phi || !pi0 || !gamma0) FAIL(ctx, VBX_ERR_INVALID, "vbx_batch_set_recording: NULL input");
    if ((x_dtype != VBX_F32 && x_dtype != VBX_F64) || (g_dtype != VBX_F32 && g_dtype != VBX_F64))
        FAIL(ctx, VBX_ERR_INVALID, "bad element type");
    if ((alpha0 == nullptr) != (invL0 == nullptr)) { alpha0 = nullptr; invL0 = nullptr; }   // VBx.py:94 needs both
    if (!(loopProb >= 0.0 && loopProb <= 1.0)) FAIL(ctx, VBX_ERR_INVALID, "loopProb=%g outside [0,1]", loopProb);
    if (!(Fa > 0.0) || !(Fb > 0.0)) FAIL(ctx, VBX_ERR_INVALID, "Fa and Fb must be positive");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    RecDesc& rd = b->recs[rec];
    rd.lp = loopProb;
    rd.Fa = Fa;
    rd.Fb = Fb;
    own_rho(b, rec);
    int rc = b->precision == VBX_PREC_FP64
                 ? set_recording_impl<double>(b, rec, X, x_dtype, Phi, pi0, gamma0, g_dtype, alpha0, invL0)
                 : set_recording_impl<float>(b, rec, X, x_dtype, Phi, pi0, gamma0, g_dtype, alpha0, invL0);
    if (rc != VBX_OK) return rc;
    b->is_set[rec] = 1;
    b->recs_dirty = true;
    b->mpart_valid = false;
    return VBX_OK;
}

// recording `rec` on the x-vectors (rho, Phi, sum G) of recording `src` of the same batch: an Fa / Fb / loopProb sweep
// over one recording keeps one rho in HBM
static int leaf_set_recording_shared(vbx_batch* b, int rec, int src, const double* pi0, const void* gamma0, int g_dtype,
                                     const double* alpha0, const double* invL0, double loopProb, double Fa, double Fb) {
    if (!b) return VBX_ERR_INVALID;
    vbx_ctx* ctx = b->ctx;
    if (rec < 0 || rec >= b->n_rec || src < 0 || src >= b->n_rec || src == rec)
        FAIL(ctx, VBX_ERR_INVALID, "vbx_batch_set_recording_shared: recording %d / source %d out of range", rec, src);
    if (!pi0 || !gamma0) FAIL(ctx, VBX_ERR_INVALID, "vbx_batch_set_recording_shared: NULL input");
    if (g_dtype != VBX_F32 && g_dtype != VBX_F64) FAIL(ctx, VBX_ERR_INVALID, "bad element type");
    if (!b->is_set[src]) FAIL(ctx, VBX_ERR_STATE, "recording %d (the source of the x-vectors) has not been set", src);
    if (b->recs[src].T != b->recs[rec].T)
        FAIL(ctx, VBX_ERR_INVALID, "recording %d has %d frames, its source %d has %d", rec, b->recs[rec].T, src, b->recs[src].T);
    if ((alpha0 == nullptr) != (invL0 == nullptr)) { alpha0 = nullptr; invL0 = nullptr; }
    if (!(loopProb >= 0.0 && loopProb <= 1.0)) FAIL(ctx, VBX_ERR_INVALID, "loopProb=%g outside [0,1]", loopProb);
    if (!(Fa > 0.0) || !(Fb > 0.0)) FAIL(ctx, VBX_ERR_INVALID, "Fa and Fb must be positive");
    // (a source that reads the rows of `rec` itself would be unset by own_rho below and leave `rec` pointing at rows that
    //  hold nothing: round-3 advisor finding)
    if (b->share_src[src] == rec)
        FAIL(ctx, VBX_ERR_STATE, "recording %d reads the x-vectors of recording %d: it cannot be that recording's source", src, rec);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    own_rho(b, rec);                                          // (whoever shared with `rec` must be set again)
    const int owner = b->share_src[src];                      // a source that shares itself: its owner
    RecDesc& rd = b->recs[rec];
    rd.lp = loopProb;
    rd.Fa = Fa;
    rd.Fb = Fb;
    rd.rho_row0 = b->recs[owner].row0;
    rd.rho_tile0 = b->recs[owner].tile0;
    rd.rho_rec = owner;
    b->share_src[rec] = owner;
    b->order_dirty = true;
    // the rows of rho this recording leaves unused lie behind another recording's: the last chunk of that one reads a whole
    // tile (finite values that meet gamma = 0, vbx_chunk_post.hpp), so they must not hold whatever the block held before
    HIPCHK(ctx, hipMemsetAsync((char*)b->d_rho + (size_t)rd.row0 * b->Dp * b->rsize, 0,
                               (size_t)std::min(rd.T, kTileFrames) * b->Dp * b->rsize, ctx->stream));
    int rc = b->precision == VBX_PREC_FP64
                 ? set_recording_impl<double>(b, rec, nullptr, VBX_F64, nullptr, pi0, gamma0, g_dtype, alpha0, invL0)
                 : set_recording_impl<float>(b, rec, nullptr, VBX_F64, nullptr, pi0, gamma0, g_dtype, alpha0, invL0);
    if (rc != VBX_OK) return rc;
    b->is_set[rec] = 1;
    b->recs_dirty = true;
    b->mpart_valid = false;
    return VBX_OK;
}

// recording `rec` of batch `b` on a COPY of the x-vectors (rho, Phi, sum G) of recording `src` of batch `from` -- another
// sub-batch of the same stream group, i.e. another device arena: a sweep over one recording that runs on several streams
// keeps one rho per stream (288 GB of HBM: a copy of 100 MB buys a stream of its own), shared by the points of that stream
static int leaf_set_recording_cloned(vbx_batch* b, int rec, vbx_batch* from, int src, const double* pi0, const void* gamma0,
                                     int g_dtype, const double* alpha0, const double* invL0, double loopProb, double Fa, double Fb) {
    if (!b || !from) return VBX_ERR_INVALID;
    vbx_ctx* ctx = b->ctx;
    if (rec < 0 || rec >= b->n_rec || src < 0 || src >= from->n_rec)
        FAIL(ctx, VBX_ERR_INVALID, "vbx_batch_set_recording_shared: recording %d / source %d out of range", rec, src);
    if (!pi0 || !gamma0) FAIL(ctx, VBX_ERR_INVALID, "vbx_batch_set_recording_shared: NULL input");
    if (g_dtype != VBX_F32 && g_dtype != VBX_F64) FAIL(ctx, VBX_ERR_INVALID, "bad element type");
    if (!from->is_set[src]) FAIL(ctx, VBX_ERR_STATE, "the source of the x-vectors has not been set");
    if (from->recs[src].T != b->recs[rec].T)
        FAIL(ctx, VBX_ERR_INVALID, "recording %d has %d frames, its source has %d", rec, b->recs[rec].T, from->recs[src].T);
    if (from->precision != b->precision || from->Dp != b->Dp || from->rsize != b->rsize) FAIL(ctx, VBX_ERR_STATE, "sub-batches of one group differ in layout");
    if ((alpha0 == nullptr) != (invL0 == nullptr)) { alpha0 = nullptr; invL0 = nullptr; }
    if (!(loopProb >= 0.0 && loopProb <= 1.0)) FAIL(ctx, VBX_ERR_INVALID, "loopProb=%g outside [0,1]", loopProb);
    if (!(Fa > 0.0) || !(Fb > 0.0)) FAIL(ctx, VBX_ERR_INVALID, "Fa and Fb must be positive");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // the source lives on another stream: its rows (and its sum G) must be complete before they are copied from
    if (int rc = sync_uploads(from); rc != VBX_OK) {
        ctx->err = from->ctx->err;
        return rc;
    }
    const RecDesc& sd = from->recs[from->share_src[src]];     // the rows `src` reads
    RecDesc& rd = b->recs[rec];
    rd.lp = loopProb;
    rd.Fa = Fa;
    rd.Fb = Fb;
    own_rho(b, rec);
    HIPCHK(ctx, hipMemcpyAsync((char*)b->d_rho + (size_t)rd.row0 * b->Dp * b->rsize, (const char*)from->d_rho + (size_t)sd.row0 * b->Dp * b->rsize,
                               (size_t)rd.T * b->Dp * b->rsize, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(b->d_phi + (size_t)rec * b->Dp, from->d_phi + (size_t)src * b->Dp, sizeof(double) * b->Dp, hipMemcpyDeviceToDevice, ctx->stream));
    rd.gsum = from->recs[src].gsum;
    int rc = b->precision == VBX_PREC_FP64
                 ? set_recording_impl<double>(b, rec, nullptr, VBX_F64, nullptr, pi0, gamma0, g_dtype, alpha0, invL0)
                 : set_recording_impl<float>(b, rec, nullptr, VBX_F64, nullptr, pi0, gamma0, g_dtype, alpha0, invL0);
    if (rc != VBX_OK) return rc;
    b->is_set[rec] = 1;
    b->recs_dirty = true;
    b->mpart_valid = false;
    return VBX_OK;
}

// VBX_OPT_GEMM = split: the f16 copies of rho (vbx_split.hpp) of every recording whose x-vectors have changed since they
// were made -- largest magnitude, power-of-two scale, then the two fragment-ordered copies; the recordings that share a
// rho read their owner's tiles and scale (RecDesc::rho_tile0 / rho_rec).
static int prepare_split(vbx_batch* b) {
    if (!split_wanted(b)) return VBX_OK;
    vbx_ctx* ctx = b->ctx;
    if (!b->d_rho_a) {
        const size_t tile_bytes = (size_t)kTileFrames * b->Dp * 4;
        int rc = dmalloc_bytes(ctx, &b->d_rho_a, (size_t)b->ntiles_total * tile_bytes);
        if (rc == VBX_OK) rc = dmalloc_bytes(ctx, &b->d_rho_b, (size_t)b->ntiles_total * tile_bytes);
        if (rc == VBX_OK) rc = dmalloc_bytes(ctx, &b->d_alpha_frag, (size_t)2 * b->n_rec * b->Sp * b->Dp * 6);
        if (rc == VBX_OK) rc = dmalloc(ctx, &b->d_rho_e, (size_t)b->n_rec);
        if (rc == VBX_OK) rc = dmalloc(ctx, &b->d_rho_amax, (size_t)2 * b->n_rec);
        if (rc == VBX_OK) rc = dmalloc(ctx, &b->d_alpha_e, (size_t)2 * b->n_rec * b->Sp);
        if (rc != VBX_OK) return rc;
        HIPCHK(ctx, hipMemsetAsync(b->d_alpha_frag, 0, (size_t)2 * b->n_rec * b->Sp * b->Dp * 6, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(b->d_alpha_e, 0, sizeof(int) * 2 * b->n_rec * b->Sp, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(b->d_rho_e, 0, sizeof(int) * b->n_rec, ctx->stream));
        b->split_dirty.assign(b->n_rec, 1);
        b->split_bad.assign(b->n_rec, 0);
    }
    const size_t tile_halfs = (size_t)kTileFrames * b->Dp * 2;
    std::vector<int> fresh;
    for (int i = 0; i < b->n_rec; ++i) {
        if (b->share_src[i] != i || !b->split_dirty[i]) continue;
        const RecDesc& rd = b->recs[i];
        const float* rho = (const float*)b->d_rho + rd.row0 * b->Dp;
        static const int init[2] = {0, 0x7f800000};              // {largest = 0, smallest frame maximum = +inf}
        HIPCHK(ctx, hipMemcpyAsync(b->d_rho_amax + 2 * i, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
        LaunchScope ls(b, VBX_K_PREP);
        hipLaunchKernelGGL(rho_absmax_kernel, dim3(rd.ntiles), dim3(256), 0, ctx->stream, rho, rd.T, b->Dp, b->d_rho_amax + 2 * i);
        hipLaunchKernelGGL(rho_split_kernel, dim3(rd.ntiles), dim3(256), 0, ctx->stream, rho, rd.T, b->Dp,
                           (const int*)(b->d_rho_amax + 2 * i), b->d_rho_e + i, (_Float16*)b->d_rho_a + (size_t)rd.tile0 * tile_halfs,
                           (_Float16*)b->d_rho_b + (size_t)rd.tile0 * tile_halfs);
        b->split_dirty[i] = 0;
        fresh.push_back(i);
    }
    HIPCHK(ctx, hipGetLastError());
    if (!fresh.empty()) {
        // one power-of-two scale per recording: does it cover the recording's frames?  (once per upload; the copy waits for
        // the kernels above)
        std::vector<int> range((size_t)2 * b->n_rec);
        HIPCHK(ctx, hipMemcpyAsync(range.data(), b->d_rho_amax, sizeof(int) * range.size(), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int i : fresh) {
            float hi, lo;
            memcpy(&hi, &range[2 * i], 4);
            memcpy(&lo, &range[2 * i + 1], 4);
            b->split_bad[i] = (hi > 0.0f && lo < hi && lo * (float)(1 << kSplitRangeBits) < hi) ? 1 : 0;
        }
        b->split_declined = false;
        for (int i = 0; i < b->n_rec; ++i) b->split_declined = b->split_declined || (b->share_src[i] == i && b->split_bad[i]);
    }
    return VBX_OK;
}

// One run = begin (checks, tables, start event) -> max_iters x { launch one iteration; now and then look at the
// convergence flags } -> end (stop event, wait, timings).  Split so that a stream group can interleave its kids.
static int run_begin(vbx_batch* b, int max_iters) {
    vbx_ctx* ctx = b->ctx;
    if (max_iters < 0) FAIL(ctx, VBX_ERR_INVALID, "max_iters < 0");
    for (int i = 0; i < b->n_rec; ++i)
        if (!b->is_set[i]) FAIL(ctx, VBX_ERR_STATE, "recording %d has not been set", i);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = choose_fb_algo(b, false);
    if (rc != VBX_OK) return rc;
    if ((rc = sync_uploads(b)) != VBX_OK) return rc;          // (uploads enqueued with VBX_OPT_ASYNC_UPLOAD: sum G of every recording)
    b->mirrors_valid = false;
    b->has_run = true;
    rc = upload_recs(b);
    if (rc != VBX_OK) return rc;
    std::fill(b->k_ms, b->k_ms + VBX_K_COUNT, 0.0);
    std::fill(b->k_launches, b->k_launches + VBX_K_COUNT, 0);
    b->ev_used = 0;
    b->iters_launched = 0;
    HIPCHK(ctx, hipEventRecord(b->ev_start, ctx->stream));
    // (inside the timed run and under VBX_K_PREP: the first run after an upload pays two more passes over rho in split mode)
    return prepare_split(b);
}

static void run_launch(vbx_batch* b, double epsilon) {
    b->run_epsilon = epsilon;
    if (b->precision == VBX_PREC_FP64) launch_iteration<double>(b, epsilon);
    else launch_iteration<float>(b, epsilon);
    ++b->iters_launched;
}

// have all recordings of this batch converged?  (waits for the iterations launched so far)
static int run_all_done(vbx_batch* b, bool* all_done) {
    vbx_ctx* ctx = b->ctx;
    // (into pinned memory: a copy into pageable memory goes through the runtime's staging path -- a question cost 19 us of idle
    //  GPU, now 12: one recording of T = 10 000 at the default of a question every four iterations 51.2 -> 49.2 us per
    //  iteration (46.4 without the test), eight recordings 66.3 -> 62.0; round 6)
    const size_t bytes = sizeof(RecState) * (size_t)b->n_rec;
    if (!b->h_poll && vbx_host_alloc(bytes, (void**)&b->h_poll) != VBX_OK) b->h_poll = nullptr;
    std::vector<RecState> pageable;
    RecState* st = b->h_poll;
    if (!st) {
        pageable.resize(b->n_rec);
        st = pageable.data();
    }
    HIPCHK(ctx, hipMemcpyAsync(st, b->d_state + (size_t)b->state_cur * b->n_rec, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *all_done = true;
    for (int i = 0; i < b->n_rec; ++i) *all_done = *all_done && st[i].done;
    return VBX_OK;
}

// The fused path keeps gamma on the chip; what a caller can ask for (VBx.py:126) is written here, once, from the b,
// boundary vectors and priors of every recording's last iteration (vbx_chunk_post.hpp, REPLAY).
extern "C++" {
namespace {
template <typename R> void launch_gamma_replay(vbx_batch* b) {
    b->fused_now = true;
    auto v = b->view<R>(0.0);
    LaunchScope ls(b, VBX_K_POST);
    switch (b->Sp) {
        case 16: launch_chunk_post<R, 16, true>(b, v); break;
        case 32: launch_chunk_post<R, 32, true>(b, v); break;
        case 64: launch_chunk_post<R, 64, true>(b, v); break;
        default: break;
    }
}
}  // namespace
}  // extern "C++"

static int run_end(vbx_batch* b) {
    vbx_ctx* ctx = b->ctx;
    if (b->fin_pending) {                     // the last iteration launched: ELBO, pi, history, convergence
        if (b->precision == VBX_PREC_FP64) launch_fin<double>(b, b->run_epsilon, 2);
        else launch_fin<float>(b, b->run_epsilon, 2);
        b->fin_pending = false;
    }
    if (b->gamma_stale) {
        if (b->precision == VBX_PREC_FP64) launch_gamma_replay<double>(b);
        else launch_gamma_replay<float>(b);
        b->gamma_stale = false;
    }
    HIPCHK(ctx, hipEventRecord(b->ev_stop, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    if (b->launch_rc != VBX_OK) {             // (a launch helper had no instance for Sp: nothing ran -- an error, not a no-op)
        const int rc = b->launch_rc;
        b->launch_rc = VBX_OK;
        FAIL(ctx, rc, "no kernel instance for a padded state count of %d", b->Sp);
    }
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, b->ev_start, b->ev_stop));
    b->last_ms = ms;
    return collect_profile(b);
}

static int leaf_run(vbx_batch* b, int max_iters, double epsilon) {
    int rc = run_begin(b, max_iters);
    if (rc != VBX_OK) return rc;
    const bool can_stop = epsilon > -1e299;
    for (int it = 0; it < max_iters; ++it) {
        run_launch(b, epsilon);
        if (can_stop && ((it + 1) % b->check_every == 0) && it + 1 < max_iters) {
            bool all_done = false;
            if ((rc = run_all_done(b, &all_done)) != VBX_OK) return rc;
            if (all_done) break;
        }
    }
    return run_end(b);
}

// Results (round 6).  The small ones -- state, priors, ELBO history of EVERY recording -- are fetched once after a run into host
// mirrors (three copies per run instead of four round trips per recording); the responsibilities and the speaker models are
// unpadded and widened on the device and copied straight into the caller's arrays.  fetch_enqueue only enqueues (its copies
// are asynchronous when the destination is pinned memory: vbx_host_alloc), fetch_finish waits: vbx_batch_get_result is one
// of each, vbx_batch_get_results enqueues a whole list before it waits once per stream.
static int fetch_mirrors(vbx_batch* b) {
    if (b->mirrors_valid) return VBX_OK;
    vbx_ctx* ctx = b->ctx;
    if (int rc = sync_uploads(b); rc != VBX_OK) return rc;   // (results asked for before any run: the priors are still on their way)
    b->model_mirrors_valid = false;
    b->h_state.resize(b->n_rec);
    b->h_pi.resize((size_t)b->n_rec * b->Sp);
    b->h_Li.resize((size_t)b->n_rec * std::max(b->max_iters, 1));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(b->h_state.data(), b->d_state + (size_t)b->state_cur * b->n_rec, sizeof(RecState) * b->n_rec, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(b->h_pi.data(), b->d_pi, sizeof(double) * b->h_pi.size(), hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(b->h_Li.data(), b->d_Li, sizeof(double) * b->h_Li.size(), hipMemcpyDeviceToHost));
    b->mirrors_valid = true;
    return VBX_OK;
}

// alpha / invL of every recording, unpadded and widened on the device, in two copies (instead of two launches and two copies per
// recording): only when a caller asks for a model
extern "C++" {
namespace {
template <typename R> int fetch_model_mirrors(vbx_batch* b) {
    if (b->model_mirrors_valid) return VBX_OK;
    vbx_ctx* ctx = b->ctx;
    const size_t cells = (size_t)b->n_rec * b->Sp * b->D;
    std::vector<int> copy_of(b->n_rec);
    // the model of the last iteration that ran, n_iters - 1, lives in copy (n_iters - 1) & 1 (fin_kernel); before any
    // iteration: copy 0, where a caller's alpha / invL went
    for (int i = 0; i < b->n_rec; ++i) copy_of[i] = b->h_state[i].n_iters > 0 ? ((b->h_state[i].n_iters - 1) & 1) : 0;
    int* d_copy = nullptr;
    double* d_out = nullptr;
    int rc = dmalloc(ctx, &d_copy, (size_t)b->n_rec);
    if (rc == VBX_OK) rc = dmalloc(ctx, &d_out, 2 * cells);
    if (rc != VBX_OK) {
        ctx_free(ctx, d_copy);
        return rc;
    }
    b->h_alpha.resize(cells);
    b->h_invL.resize(cells);
    hipError_t e = hipMemcpyAsync(d_copy, copy_of.data(), sizeof(int) * b->n_rec, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((vbx::unpack_models_kernel<R>), dim3(b->n_rec), dim3(256), 0, ctx->stream, (const R*)b->d_alpha, (const R*)b->d_invL,
                           (const int*)d_copy, (long long)b->n_rec * b->Sp * b->Dp, d_out, d_out + cells, b->n_rec, b->Sp, b->D, b->Dp);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(b->h_alpha.data(), d_out, sizeof(double) * cells, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(b->h_invL.data(), d_out + cells, sizeof(double) * cells, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    ctx_free(ctx, d_copy);
    ctx_free(ctx, d_out);
    if (e != hipSuccess) FAIL(ctx, VBX_ERR_HIP, "fetching the speaker models failed: %s", hipGetErrorString(e));
    b->model_mirrors_valid = true;
    return VBX_OK;
}
}  // namespace
}  // extern "C++"

extern "C++" {
namespace {
template <typename R>
int fetch_enqueue_impl(vbx_batch* b, int rec, double* gamma, double* pi, double* Li, int li_cap, int* n_iters,
                       int* warned, double* alpha, double* invL) {
    vbx_ctx* ctx = b->ctx;
    const RecDesc& rd = b->recs[rec];
    const int Sp = b->Sp, Dp = b->Dp, S = rd.S, D = b->D;
    if (int rc = fetch_mirrors(b); rc != VBX_OK) return rc;
    const RecState& st = b->h_state[rec];
    if (n_iters) *n_iters = st.n_iters;
    if (warned) *warned = st.warned;
    if (pi)
        for (int s = 0; s < S; ++s) pi[s] = b->h_pi[(size_t)rec * Sp + s];
    if (Li && li_cap > 0) {
        const int n = std::min(std::min(st.n_iters, li_cap), b->max_iters);
        for (int k = 0; k < n; ++k) Li[k] = b->h_Li[(size_t)rec * b->max_iters + k];
    }
    // device staging for what is unpacked: the upload staging block of the batch is free between uploads; its three users
    // below follow each other in the stream, so one block serves them all (a copy out of it is ahead of the next kernel into it)
    const size_t cells = (size_t)rd.T * S;
    const size_t need = sizeof(double) * (gamma ? cells : 0);
    double* d_out = (double*)b->d_xstage;
    if (need > b->xstage_bytes) {
        // (more speakers than feature dimensions: a block of its own, kept for the rest of the batch's life)
        if (b->fetch_bytes < need) {
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            if (b->d_fetch) ctx_free(ctx, b->d_fetch);
            b->d_fetch = nullptr;
            b->fetch_bytes = 0;
            int rc = dmalloc_bytes(ctx, &b->d_fetch, need);
            if (rc != VBX_OK) return rc;
            b->fetch_bytes = need;
        }
        d_out = (double*)b->d_fetch;
    }
    if (gamma) {
        hipLaunchKernelGGL((vbx::unpack_gamma_kernel<R>), dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const R*)b->d_gamma + rd.row0 * Sp, d_out, (long long)rd.T, S, Sp);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(gamma, d_out, sizeof(double) * cells, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (alpha || invL) {
        if (int rc = fetch_model_mirrors<R>(b); rc != VBX_OK) return rc;
        for (int sp = 0; sp < S; ++sp) {
            const size_t src = ((size_t)rec * Sp + sp) * D, dst = (size_t)sp * D;
            if (alpha) std::memcpy(alpha + dst, b->h_alpha.data() + src, sizeof(double) * D);
            if (invL) std::memcpy(invL + dst, b->h_invL.data() + src, sizeof(double) * D);
        }
    }
    return VBX_OK;
}
}  // namespace
}  // extern "C++"

static int leaf_fetch_enqueue(vbx_batch* b, int rec, double* gamma, double* pi, double* Li, int li_cap, int* n_iters,
                              int* warned, double* alpha, double* invL) {
    if (!b) return VBX_ERR_INVALID;
    if (rec < 0 || rec >= b->n_rec) FAIL(b->ctx, VBX_ERR_INVALID, "recording index %d out of range", rec);
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    return b->precision == VBX_PREC_FP64
               ? fetch_enqueue_impl<double>(b, rec, gamma, pi, Li, li_cap, n_iters, warned, alpha, invL)
               : fetch_enqueue_impl<float>(b, rec, gamma, pi, Li, li_cap, n_iters, warned, alpha, invL);
}

static int leaf_fetch_finish(vbx_batch* b) {
    HIPCHK(b->ctx, hipStreamSynchronize(b->ctx->stream));
    return VBX_OK;
}

static int leaf_get_result(vbx_batch* b, int rec, double* gamma, double* pi, double* Li, int li_cap, int* n_iters,
                         int* warned, double* alpha, double* invL) {
    const int rc = leaf_fetch_enqueue(b, rec, gamma, pi, Li, li_cap, n_iters, warned, alpha, invL);
    return rc != VBX_OK ? rc : leaf_fetch_finish(b);
}

static int leaf_last_run_ms(vbx_batch* b, double* total_ms, int* iters_launched) {
    if (!b) return VBX_ERR_INVALID;
    if (total_ms) *total_ms = b->last_ms;
    if (iters_launched) *iters_launched = b->iters_launched;
    return VBX_OK;
}

static int leaf_kernel_times(vbx_batch* b, double* ms, int64_t* launches) {
    if (!b) return VBX_ERR_INVALID;
    for (int k = 0; k < VBX_K_COUNT; ++k) {
        if (ms) ms[k] = b->k_ms[k];
        if (launches) launches[k] = b->k_launches[k];
    }
    return VBX_OK;
}

}  // extern "C"
