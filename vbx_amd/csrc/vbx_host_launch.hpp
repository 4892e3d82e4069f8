// vbx_host_launch.hpp -- host runtime: the launch sequence of one VB iteration (VBx.py:91-125) and the device allocator
// (one translation unit with vbx_capi.hip, which includes the parts in order; not a stand-alone header)
#pragma once
namespace {

// ---------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------
struct LaunchScope {   // brackets one kernel launch with events when profiling is on
    vbx_batch* b;
    EventPair* ep = nullptr;
    LaunchScope(vbx_batch* b_, int klass) : b(b_) {
        if (!((b->profile >> klass) & 1)) return;
        if (b->ev_used == b->ev_pool.size()) {
            EventPair p{klass, nullptr, nullptr};
            if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
            b->ev_pool.push_back(p);
        }
        ep = &b->ev_pool[b->ev_used++];
        ep->klass = klass;
        (void)hipEventRecord(ep->a, b->ctx->stream);
    }
    ~LaunchScope() {
        if (ep) (void)hipEventRecord(ep->b, b->ctx->stream);
    }
};

// (debugging aid: VBX_AMD_SPLIT_MASK = 1 / 2 keeps the split GEMM to chunk_loglik / chunk_post only)
static int split_debug_mask() {
    static const int m = [] { const char* e = experiment_env("VBX_AMD_SPLIT_MASK"); return e ? atoi(e) : 3; }();
    return m;
}

#define NT_SWITCH(nt_, BODY)                                   \
    switch (nt_) {                                             \
        case 1: { constexpr int kNT = 1; BODY } break;         \
        case 2: { constexpr int kNT = 2; BODY } break;         \
        case 4: { constexpr int kNT = 4; BODY } break;         \
        case 8: { constexpr int kNT = 8; BODY } break;         \
        case 16: { constexpr int kNT = 16; BODY } break;       \
        default: b->launch_rc = VBX_ERR_UNSUPPORTED; break;    \
    }

template <typename R> void launch_mstep_acc(vbx_batch* b, double eps) {
    auto v = b->view<R>(eps);
    LaunchScope ls(b, VBX_K_MSTEP_ACC);
    const int nt = std::min(b->NT, 16);                       // (more than 256 speakers: blocks of 16 tiles along grid z)
    dim3 grid(b->ntiles_total, b->Dp / 32, b->NT / nt);
    NT_SWITCH(nt, hipLaunchKernelGGL((mstep_acc_kernel<R, kNT>), grid, dim3(64), 0, b->ctx->stream, v);)
}

static int small_kernel_threads(const vbx_batch* b, int from_tiles);

// fin_kernel (vbx_kernels.hpp): mode 1 = start an iteration (M-step), 2 = finish one (ELBO, pi, convergence), 3 = finish
// the previous one and start the next in the same launch.  A launch with a finishing role writes the other state copy.
// More than 1024 states (vbx_big.hpp): switch over NR = Sp / 1024
#define BIG_SWITCH(sp, BODY)                                   \
    switch ((sp) >> 10) {                                      \
        case 2: { constexpr int kNR = 2; BODY } break;         \
        case 4: { constexpr int kNR = 4; BODY } break;         \
        case 8: { constexpr int kNR = 8; BODY } break;         \
        case 16: { constexpr int kNR = 16; BODY } break;       \
        default: break;                                        \
    }

template <typename R> void launch_fin(vbx_batch* b, double eps, int mode) {
    if (b->Sp > 1024) {
        // the finishing role has a kernel of its own there (a thread per speaker does not reach); it runs first and the host
        // swaps the state buffers, so the M-step role that follows finds the state of the iteration it starts
        if (mode & 2) {
            auto v = b->view<R>(eps);
            LaunchScope ls(b, VBX_K_ITER_FIN);
            BIG_SWITCH(b->Sp, hipLaunchKernelGGL((iter_fin_big_kernel<R, kNR>), dim3(b->n_rec), dim3(1024), 0, b->ctx->stream, v);)
            b->state_cur ^= 1;
        }
        if (mode & 1) {
            auto v = b->view<R>(eps);
            LaunchScope ls(b, VBX_K_MSTEP_FIN);
            hipLaunchKernelGGL((fin_kernel<R>), dim3(b->n_rec, b->Sp + 1), dim3(1024), 0, b->ctx->stream, v, 1);
        }
        return;
    }
    auto v = b->view<R>(eps);
    LaunchScope ls(b, mode == 2 ? VBX_K_ITER_FIN : VBX_K_MSTEP_FIN);
    hipLaunchKernelGGL((fin_kernel<R>), dim3(b->n_rec, b->Sp + 1), dim3(small_kernel_threads(b, 80)), 0, b->ctx->stream, v, mode);
    if (mode & 2) b->state_cur ^= 1;
}

template <typename R> void launch_mstep(vbx_batch* b, double eps) {
    launch_mstep_acc<R>(b, eps);
    launch_fin<R>(b, eps, 1);
}

template <typename R> void launch_loglik(vbx_batch* b, double eps, bool raw) {
    auto v = b->view<R>(eps);
    LaunchScope ls(b, VBX_K_LOGLIK);
    R* lraw = raw ? (R*)b->d_lraw : nullptr;
    const int nt = std::min(b->NT, 16);
    NT_SWITCH(nt, hipLaunchKernelGGL((loglik_kernel<R, kNT>), dim3(b->ntiles_total, b->NT / nt), dim3(256), 0,
                                     b->ctx->stream, v, lraw);)
    if (b->NT > nt)          // the row maximum spans several speaker blocks
        hipLaunchKernelGGL((rownorm_kernel<R>), dim3(b->ntiles_total), dim3(256), 0, b->ctx->stream, v);
}

// Block size of the per-recording reductions over tiles (fin_kernel): more threads once a recording has more partials than
// the smaller block fetches in a few rounds (one recording of T = 200 000, rounds 2-3: mstep_fin 42 -> 30 us, iter_fin 33 -> 23 us
// with 1024 threads instead of 256).
static int small_kernel_threads(const vbx_batch* b, int from_tiles) {
    static const int forced = [] { const char* e = experiment_env("VBX_AMD_FIN_THREADS"); const int v = e ? atoi(e) : 0; return (v == 256 || v == 512 || v == 1024) ? v : 0; }();
    if (forced && b->Sp <= 256) return forced;
    int maxtiles = 0;
    for (auto& rd : b->recs) maxtiles = std::max(maxtiles, rd.ntiles);
    // (round 6, one recording, split, us per iteration with 256 / 512 / 1024 threads: T = 12 000 48.7 / 48.0 / 51.3, 20 000 52.9 /
    //  53.6 / 56.1, 50 000 74.0 / 73.4 / 75.9, 100 000 102.8 / 98.0 / 100.2; four of T = 20 000: 65.5 / 65.0 / 68.8 -- the barriers
    //  of a block of sixteen waves cost more than its shorter rounds save until a recording has about a thousand partials)
    if (b->Sp > 256) return 1024;                                      // (iter_fin: a thread per speaker)
    // (T = 150 000 / 200 000 alone: 126.1 / 149.7 with 512 against 125.0 / 148.6 with 1024; the nine-point sweep over T = 200 000
    //  on three streams: 1.390 against 1.407 ms -- the lighter block fits beside the other streams' kernels)
    return (maxtiles > 1200 && b->n_rec == 1) ? 1024 : maxtiles > from_tiles ? 512 : 256;
}

// chunk_post over the tiles of the batch; REPLAY: the instance that only writes the responsibilities
// Does chunk_post walk the last level of the boundary walk itself (FOLD, vbx_chunk_post.hpp)?  Where an iteration is its launches:
// a grouped walk, the group's operators fit the free LDS region, and the batch does not fill the chip (beyond that the extra
// mat-vecs per workgroup cost more than the launch they replace).  VBX_AMD_FOLD_WALK=0 / 1 forces it off / on (A/B runs).
// A batch that does not fill the chip: the small-batch instances of the chunk kernels (FOLD / LAT).  VBX_AMD_FOLD_WALK /
// VBX_AMD_SMALL_BATCH = 0 / 1 force them off / on (A/B runs, under VBX_AMD_EXPERIMENT=1).
static bool small_batch_wanted(const vbx_batch* b) {
    static const int forced = [] { const char* e = experiment_env("VBX_AMD_SMALL_BATCH"); return (e && *e) ? (e[0] == '0' ? 0 : 1) : -1; }();
    if (forced >= 0) return forced == 1;
    return b->ntiles_total <= 2048;
}
template <int SP> bool fold_walk_wanted(const vbx_batch* b) {
    static const int forced = [] { const char* e = experiment_env("VBX_AMD_FOLD_WALK"); return (e && *e) ? (e[0] == '0' ? 0 : 1) : -1; }();
    if (SP > 32 || b->sgroup <= 1 || b->spt != 1 || b->sgroup - 1 > kTileFrames / SP) return false;
    if (forced >= 0) return forced == 1;
    return small_batch_wanted(b);
}

template <typename R, int SP, bool REPLAY> void launch_chunk_post(vbx_batch* b, const BatchView<R>& v) {
    if constexpr (ChunkPostCfg<R, SP>::kFits) {
        constexpr bool kCanFold = SP <= 32;                  // (a group has at least four chunks: three operators in r1)
        const bool fold = kCanFold && b->fold_now;
        if constexpr (std::is_same<R, float>::value && !REPLAY) {
            if (v.rho_b && (split_debug_mask() & 2)) {       // gamma^T rho on the f16 matrix cores (vbx_split.hpp)
                if constexpr (kCanFold) {
                    if (fold) {
                        hipLaunchKernelGGL((chunk_post_kernel<R, SP, false, true, true>), dim3(b->nblocks_chunk), dim3(256), 0, b->ctx->stream, v);
                        return;
                    }
                }
                hipLaunchKernelGGL((chunk_post_kernel<R, SP, false, true>), dim3(b->nblocks_chunk), dim3(256), 0, b->ctx->stream, v);
                return;
            }
        }
        if constexpr (kCanFold) {
            if (fold) {
                hipLaunchKernelGGL((chunk_post_kernel<R, SP, REPLAY, false, true>), dim3(b->nblocks_chunk), dim3(256), 0, b->ctx->stream, v);
                return;
            }
        }
        hipLaunchKernelGGL((chunk_post_kernel<R, SP, REPLAY>), dim3(b->nblocks_chunk), dim3(256), 0, b->ctx->stream, v);
    }
}

template <typename R, int SP> void launch_scan(vbx_batch* b, const BatchView<R>& v, bool fused_post, bool fused_loglik) {
    hipStream_t st = b->ctx->stream;
    bool have_op = false;
    if constexpr (ChunkLoglikCfg<R, SP>::kFits) {
        if (fused_loglik) {      // log-likelihoods and the chunk operators in one pass over rho
            LaunchScope ls(b, VBX_K_CHUNK_LOGLIK);
            bool launched = false;
            // LAT: the whole rho slab of a workgroup in flight before the first product (vbx_chunk_loglik.hpp): batches that do
            // not fill the chip, feature dimensions that fit the registers
            const bool lat = SP <= 32 && b->Dp <= 128 && small_batch_wanted(b);
            if constexpr (std::is_same<R, float>::value) {
                if (v.rho_a && (split_debug_mask() & 1)) {   // rho alpha^T on the f16 matrix cores (vbx_split.hpp)
                    if constexpr (SP <= 32) {
                        if (lat) {
                            hipLaunchKernelGGL((chunk_loglik_kernel<R, SP, true, true>), dim3(b->nblocks_chunk), dim3(256), 0, st, v);
                            launched = true;
                        }
                    }
                    if (!launched) hipLaunchKernelGGL((chunk_loglik_kernel<R, SP, true>), dim3(b->nblocks_chunk), dim3(256), 0, st, v);
                    launched = true;
                }
            }
            // (exact f32 and fp64: LAT measured slower -- 8 recordings 70.9 -> 79.9 / 117.6 -> 120.8 us per iteration together
            //  with chunk_post's counterpart: the extra registers cost the occupancy these batches need; split only)
            if (!launched) hipLaunchKernelGGL((chunk_loglik_kernel<R, SP>), dim3(b->nblocks_chunk), dim3(256), 0, st, v);
            have_op = true;
        }
    }
    if (!have_op) {
        LaunchScope ls(b, VBX_K_FB);
        hipLaunchKernelGGL((scan1_kernel<R, SP>), dim3(b->ntiles_total), dim3(SP * SP / 4), 0, st, v);
    }
    bool fold = false;
    if constexpr (ChunkPostCfg<R, SP>::kFits) fold = fused_post && fold_walk_wanted<SP>(b);
    b->fold_now = fold;                          // (the gamma write-out after the run replays with the same choice)
    {
        LaunchScope ls(b, VBX_K_FB_AUX);
        if (b->sgroup > 1 && b->sgroup2 > 1) {   // very long recordings: groups of groups on top
            hipLaunchKernelGGL((scan_compose_kernel<R, SP>), dim3(b->nsup_total), dim3(256), 0, st, v, 1);
            hipLaunchKernelGGL((scan_compose_kernel<R, SP>), dim3(b->nsup2_total), dim3(256), 0, st, v, 2);
            hipLaunchKernelGGL((scan2_kernel<R, SP>), dim3(b->n_rec, 2), dim3(256), 0, st, v, 4);
            hipLaunchKernelGGL((scan2_kernel<R, SP>), dim3(b->nsup2_total, 2), dim3(256), 0, st, v, 5);
            if (!fold) hipLaunchKernelGGL((scan2_kernel<R, SP>), dim3(b->nsup_total, 2), dim3(256), 0, st, v, 3);
        } else if (b->sgroup > 1) {     // long recordings: group operators, boundaries at the group edges, then inside the groups
            hipLaunchKernelGGL((scan_compose_kernel<R, SP>), dim3(b->nsup_total), dim3(256), 0, st, v, 1);
            hipLaunchKernelGGL((scan2_kernel<R, SP>), dim3(b->n_rec, 2), dim3(256), 0, st, v, 2);
            if (!fold) hipLaunchKernelGGL((scan2_kernel<R, SP>), dim3(b->nsup_total, 2), dim3(256), 0, st, v, 3);
        } else {
            hipLaunchKernelGGL((scan2_kernel<R, SP>), dim3(b->n_rec, 2), dim3(256), 0, st, v, 0);
        }
    }
    if constexpr (ChunkPostCfg<R, SP>::kFits) {
        if (fused_post) {
            LaunchScope ls(b, VBX_K_CHUNK_POST);
            launch_chunk_post<R, SP, false>(b, v);
            return;
        }
    }
    {
        LaunchScope ls(b, VBX_K_FB);
        hipLaunchKernelGGL((scan3_kernel<R, SP>), dim3((b->ntiles_total + 1) / 2), dim3(64), 0, st, v);
    }
}

// 64 < S <= 256: the same three steps with operators that live in HBM (vbx_scan_wide.hpp)
template <typename R, int SP> void launch_scan_wide(vbx_batch* b, const BatchView<R>& v) {
    hipStream_t st = b->ctx->stream;
    {
        LaunchScope ls(b, VBX_K_FB);
        bool launched = false;
        if constexpr (Scan1WideLdsCfg<R, SP>::kFits) {
            static const bool off = [] { const char* e = experiment_env("VBX_AMD_SCAN1_WIDE_LDS"); return e && e[0] == '0'; }();
            if (b->d_lppow && !off) {
                hipLaunchKernelGGL((scan1_wide_lds_kernel<R, SP>), dim3(b->ntiles_total, SP / Scan1WideLdsCfg<R, SP>::COLS), dim3(1024), 0, st, v);
                launched = true;
            }
        }
        if (!launched) hipLaunchKernelGGL((scan1_wide_kernel<R, SP>), dim3(b->ntiles_total, SP / ScanWideCfg<R, SP>::CB), dim3(256), 0, st, v);
    }
    {
        LaunchScope ls(b, VBX_K_FB_AUX);
        if (b->sgroup > 1) {     // group operators, boundaries at the group edges, then inside the groups (as launch_scan does)
            hipLaunchKernelGGL((compose_wide_kernel<R, SP>), dim3(b->nsup_total, SP / ComposeWideCfg<R, SP>::CW), dim3(256), 0, st, v);
            hipLaunchKernelGGL((scan2_wide_kernel<R, SP, 16>), dim3(b->n_rec, 2), dim3(1024), 0, st, v, 2);
            hipLaunchKernelGGL((scan2_wide_kernel<R, SP, 16>), dim3(b->nsup_total, 2), dim3(1024), 0, st, v, 3);
        } else {
            hipLaunchKernelGGL((scan2_wide_kernel<R, SP, 16>), dim3(b->n_rec, 2), dim3(1024), 0, st, v, 0);
        }
    }
    {
        LaunchScope ls(b, VBX_K_FB);
        hipLaunchKernelGGL((scan3_wide_kernel<R, SP>), dim3(b->ntiles_total, 2), dim3(64), 0, st, v);
    }
}

template <typename R> bool fused_loglik_available(const vbx_batch* b) {
    if (!b->use_chunked || b->fuse < 2) return false;
    switch (b->Sp) {
        case 16: return ChunkLoglikCfg<R, 16>::kFits;
        case 32: return ChunkLoglikCfg<R, 32>::kFits;
        case 64: return ChunkLoglikCfg<R, 64>::kFits;
        default: return false;
    }
}

// Can this batch run the fused per-chunk kernels?  (chunked scan + the lattices fit in LDS)
template <typename R> bool fused_available(const vbx_batch* b) {
    if (!b->use_chunked || !b->fuse) return false;
    switch (b->Sp) {
        case 16: return ChunkPostCfg<R, 16>::kFits;
        case 32: return ChunkPostCfg<R, 32>::kFits;
        case 64: return ChunkPostCfg<R, 64>::kFits;
        default: return false;
    }
}

template <typename R> void launch_fb(vbx_batch* b, double eps, bool fused_post = false, bool fused_loglik = false) {
    auto v = b->view<R>(eps);
    if (b->use_chunked) {
        switch (b->Sp) {
            case 16: launch_scan<R, 16>(b, v, fused_post, fused_loglik); return;
            case 32: launch_scan<R, 32>(b, v, fused_post, fused_loglik); return;
            case 64: launch_scan<R, 64>(b, v, fused_post, fused_loglik); return;
            case 128: launch_scan_wide<R, 128>(b, v); return;
            case 256: launch_scan_wide<R, 256>(b, v); return;
            default: break;
        }
    }
    LaunchScope ls(b, VBX_K_FB);
    const int nreg = std::max(1, b->Sp / 64);
    switch (nreg) {
        case 1: hipLaunchKernelGGL((fb_seq_kernel<R, 1>), dim3(b->n_rec), dim3(128), 0, b->ctx->stream, v); break;
        case 2: hipLaunchKernelGGL((fb_seq_kernel<R, 2>), dim3(b->n_rec), dim3(128), 0, b->ctx->stream, v); break;
        case 4: hipLaunchKernelGGL((fb_seq_kernel<R, 4>), dim3(b->n_rec), dim3(128), 0, b->ctx->stream, v); break;
        case 8: hipLaunchKernelGGL((fb_seq_kernel<R, 8>), dim3(b->n_rec), dim3(128), 0, b->ctx->stream, v); break;
        case 16: hipLaunchKernelGGL((fb_seq_kernel<R, 16>), dim3(b->n_rec), dim3(128), 0, b->ctx->stream, v); break;
        default: BIG_SWITCH(b->Sp, hipLaunchKernelGGL((fb_big_kernel<R, kNR>), dim3(b->n_rec, 2), dim3(1024), 0, b->ctx->stream, v);) break;
    }
}

template <typename R> void launch_post(vbx_batch* b, double eps) {
    auto v = b->view<R>(eps);
    LaunchScope ls(b, VBX_K_POST);
    dim3 grid(b->ntiles_total), block(256);
    switch (b->Sp) {
        case 16: hipLaunchKernelGGL((post_kernel<R, 16>), grid, block, 0, b->ctx->stream, v); break;
        case 32: hipLaunchKernelGGL((post_kernel<R, 32>), grid, block, 0, b->ctx->stream, v); break;
        case 64: hipLaunchKernelGGL((post_kernel<R, 64>), grid, block, 0, b->ctx->stream, v); break;
        case 128: hipLaunchKernelGGL((post_kernel<R, 128>), grid, block, 0, b->ctx->stream, v); break;
        case 256: hipLaunchKernelGGL((post_kernel<R, 256>), grid, block, 0, b->ctx->stream, v); break;
        case 512: hipLaunchKernelGGL((post_kernel<R, 512>), grid, block, 0, b->ctx->stream, v); break;
        case 1024: hipLaunchKernelGGL((post_kernel<R, 1024>), grid, block, 0, b->ctx->stream, v); break;
        default: hipLaunchKernelGGL((post_big_kernel<R>), grid, block, 0, b->ctx->stream, v); break;     // > 1024 states (vbx_big.hpp)
    }
}

// Can this batch multiply with f16 operand pairs (VBX_OPT_GEMM = split)?  fp32, both fused per-chunk kernels -- and
// (split_available) x-vectors whose dynamic range one power-of-two scale per recording covers (prepare_split).
static bool split_wanted(const vbx_batch* b) {
    return b->gemm == VBX_GEMM_SPLIT && b->precision == VBX_PREC_FP32 && b->Dp <= kSplitMaxDp &&
           fused_available<float>(b) && fused_loglik_available<float>(b);
}
static bool split_available(const vbx_batch* b) { return split_wanted(b) && !b->split_declined; }

template <typename R> void launch_iteration(vbx_batch* b, double eps) {
    b->fused_now = fused_available<R>(b);
    b->split_now = b->d_rho_a != nullptr && split_available(b);
    // the previous iteration of this run (if any) is finished by the launch that starts this one; the last one of a run
    // by run_end
    const int fin_mode = b->fin_pending ? 3 : 1;
    b->fin_pending = true;
    if (b->fused_now) {
        // chunk_post leaves gamma^T rho of the gamma it has just written in mpart/npart, so only the
        // first iteration after an upload needs the stand-alone accumulation
        if (!b->mpart_valid) launch_mstep_acc<R>(b, eps);
        launch_fin<R>(b, eps, fin_mode);
        const bool fl = fused_loglik_available<R>(b);
        // half-tile re-runs: most where the chains' latency is exposed (one recording 65 -> 58 us per iteration, fp64
        // batches -13 %), a few percent with thousands of f32 tiles in flight (there the operator build is
        // VALU-throughput bound and chunk_loglik pays 5 % for what chunk_post gains) -- never a loss, so on unless asked
        b->half_ops_now = fl && b->split_tiles != 2;
        if (!fl) launch_loglik<R>(b, eps, false);
        launch_fb<R>(b, eps, true, fl);
        b->mpart_valid = true;
        b->gamma_stale = true;
        return;
    }
    launch_mstep_acc<R>(b, eps);
    launch_fin<R>(b, eps, fin_mode);
    launch_loglik<R>(b, eps, false);
    launch_fb<R>(b, eps);
    launch_post<R>(b, eps);
    b->mpart_valid = false;
}

template <typename R, typename XT>
void launch_prep(vbx_batch* b, const RecDesc& rd, const double* sqrt_phi) {
    LaunchScope ls(b, VBX_K_PREP);
    R* rho = (R*)b->d_rho + rd.row0 * b->Dp;
    hipLaunchKernelGGL((prep_kernel<R, XT>), dim3(rd.ntiles), dim3(256), 0, b->ctx->stream,
                       (const XT*)b->d_xstage, sqrt_phi, rho, b->d_gtile + rd.tile0, rd.T,
                       b->D, b->Dp);
}

// (debugging aid: VBX_AMD_POISON=1 fills every block handed out with 0xFF bytes -- NaNs in every floating-point type -- so
//  that a read of memory nobody wrote shows up in the results instead of depending on what the block held before)
static int ctx_alloc_raw(vbx_ctx* ctx, void** p, size_t bytes);
int ctx_alloc(vbx_ctx* ctx, void** p, size_t bytes) {
    static const bool poison = [] { const char* e = std::getenv("VBX_AMD_POISON"); return e && e[0] == '1'; }();
    const int rc = ctx_alloc_raw(ctx, p, bytes);
    if (rc == VBX_OK && poison) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemset(*p, 0xFF, std::max<size_t>(bytes, 16)));
        HIPCHK(ctx, hipDeviceSynchronize());
    }
    return rc;
}

// A block of at least `bytes` bytes: the smallest spare one that fits (and is at most twice too large), else a new one.
static int ctx_alloc_raw(vbx_ctx* user, void** p, size_t bytes) {
    vbx_ctx* ctx = user->pool ? user->pool : user;
    std::lock_guard<std::mutex> lock(ctx->alloc_mutex);
    bytes = std::max<size_t>(bytes, 16);
    if (ctx->recycle) {
        int best = -1;
        for (int i = 0; i < (int)ctx->spare.size(); ++i)
            if (ctx->spare[i].second >= bytes && ctx->spare[i].second <= 2 * bytes + 4096 &&
                (best < 0 || ctx->spare[i].second < ctx->spare[best].second))
                best = i;
        if (best >= 0) {
            *p = ctx->spare[best].first;
            ctx->live[*p] = ctx->spare[best].second;
            ctx->spare_bytes -= ctx->spare[best].second;
            ctx->spare.erase(ctx->spare.begin() + best);
            return VBX_OK;
        }
    }
    HIPCHK(user, hipMalloc(p, bytes));
    ctx->live[*p] = bytes;
    return VBX_OK;
}

// Back to the spare list.  Work queued on the ctx stream that still touches the block stays ordered before its next
// use (every user of the list runs on that stream or has waited for it); beyond 16 GB / 1024 spares the block is freed
// (a batch of 64 recordings on three streams is 1.5 GB in 170 blocks).
void ctx_free(vbx_ctx* user, void* p) {
    if (!p) return;
    vbx_ctx* ctx = user->pool ? user->pool : user;
    std::lock_guard<std::mutex> lock(ctx->alloc_mutex);
    auto it = ctx->live.find(p);
    const size_t bytes = it == ctx->live.end() ? 0 : it->second;
    if (it != ctx->live.end()) ctx->live.erase(it);
    if (!ctx->recycle || bytes == 0 || ctx->spare_bytes + bytes > ((size_t)16 << 30) || ctx->spare.size() >= 1024) {
        (void)hipFree(p);
        return;
    }
    ctx->spare.emplace_back(p, bytes);
    ctx->spare_bytes += bytes;
}

template <typename T> int dmalloc(vbx_ctx* ctx, T** p, size_t count) {
    return ctx_alloc(ctx, (void**)p, std::max<size_t>(count, 1) * sizeof(T));
}
int dmalloc_bytes(vbx_ctx* ctx, void** p, size_t bytes) { return ctx_alloc(ctx, p, bytes); }

template <typename T> int scratch_get(vbx_ctx* ctx, T** p, size_t count, size_t* got_bytes) {
    *got_bytes = 0;
    return ctx_alloc(ctx, (void**)p, std::max<size_t>(count, 1) * sizeof(T));
}
void scratch_put(vbx_ctx* ctx, void* p, size_t) { ctx_free(ctx, p); }

// Decide between the sequential walk and the chunked scan, allocating the scan buffers on first use.
int choose_fb_algo(vbx_batch* b, bool step_api_logs) {
    int maxtiles = 0;
    for (auto& rd : b->recs) maxtiles = std::max(maxtiles, rd.ntiles);
    bool chunked = b->fb_algo == VBX_FB_CHUNKED || (b->fb_algo == VBX_FB_AUTO && maxtiles >= 3);
    if (step_api_logs) chunked = false;      // lfw/lbw reconstruction uses the sequential kernel's scales
    // More than 256 states: an S x S transfer operator per chunk is 1 - 4 MB and its build S^2 operations per frame -- the
    // O(T S) sequential walk (one wavefront per direction, 8 / 16 states per lane) is the better deal there.  The
    // reference takes any S (VBx.py:76-85); this path is about taking it at all, not about speed.
    if (b->Sp > 256) chunked = false;
    if (chunked && !b->d_op) {
        const size_t rs = b->rsize, nt = (size_t)b->ntiles_total, sp = (size_t)b->Sp;
        int rc = dmalloc_bytes(b->ctx, &b->d_op, nt * sp * sp * rs);
        if (rc == VBX_OK) rc = dmalloc(b->ctx, &b->d_opexp, nt * sp);
        if (rc == VBX_OK) rc = dmalloc_bytes(b->ctx, &b->d_fbound, nt * sp * rs);
        if (rc == VBX_OK) rc = dmalloc_bytes(b->ctx, &b->d_gbound, nt * sp * rs);
        if (rc == VBX_OK) rc = dmalloc(b->ctx, &b->d_tllpart, nt);
        if (rc == VBX_OK) rc = dmalloc_bytes(b->ctx, &b->d_sfw, (size_t)b->sum_T * rs);
        if (rc == VBX_OK) rc = dmalloc_bytes(b->ctx, &b->d_dump, 1024 * rs);
        if (rc != VBX_OK) return rc;
    }
    b->use_chunked = chunked;
    const int spt = 1;
    // the forward / backward lattices live in HBM only on the paths that do not keep them in LDS
    const bool fused1 = b->precision == VBX_PREC_FP64 ? fused_available<double>(b) : fused_available<float>(b);
    if (fused1 && chunked && !b->d_oph) {
        const size_t nt = (size_t)b->ntiles_total, sp = (size_t)b->Sp;
        int rc = dmalloc_bytes(b->ctx, &b->d_oph, 2 * nt * sp * sp * b->rsize);
        if (rc == VBX_OK) rc = dmalloc(b->ctx, &b->d_ophexp, 2 * nt * sp);
        if (rc == VBX_OK) rc = dmalloc_bytes(b->ctx, &b->d_cop, (size_t)b->n_rec * sp * b->rsize);
        if (rc == VBX_OK) rc = dmalloc(b->ctx, &b->d_lppow, (size_t)b->n_rec * (kTileFrames + 1));
        if (rc != VBX_OK) return rc;
        b->recs_dirty = true;                    // (the lp^n tables go up with the recording descriptors)
    }
    if (!fused1 && chunked && b->Sp > 64 && b->Sp <= 256 && !b->d_lppow) {      // the wide scan's operator build (scan1_wide_lds_kernel)
        int rc = dmalloc(b->ctx, &b->d_lppow, (size_t)b->n_rec * (kTileFrames + 1));
        if (rc != VBX_OK) return rc;
        b->recs_dirty = true;
    }
    if (!fused1 && !b->d_ahat) {
        const size_t cells = (size_t)b->sum_T * b->Sp;
        int rc = dmalloc_bytes(b->ctx, &b->d_ahat, cells * b->rsize);
        if (rc == VBX_OK) rc = dmalloc_bytes(b->ctx, &b->d_bhat, cells * b->rsize);
        if (rc != VBX_OK) return rc;
    }
    int maxchunks = maxtiles;
    if (spt == 2) {
        maxchunks = 0;
        for (auto& rd : b->recs) maxchunks = std::max(maxchunks, (rd.T + kTileFrames / 2 - 1) / (kTileFrames / 2));
    }
    // two-level walk over the chunk boundaries once the flat chain gets long
    int group = 1;
    if (chunked && b->Sp <= 64) {            // (the wide scan walks the flat chain)
        // up to 16 recordings: the walk is exposed (nothing else to fill the GPU with), and groups of four cut its
        // dependent chain from K to K/4 + 4 + 4 steps (T = 10 000, one recording: 28 -> 19 us per iteration); many
        // recordings: the three launches of the two-level walk cost more than they save until the chain is long
        // Group size: a composition step (S x S times S x S) costs about four walk steps, so the chain
        // g (compose) + K/g (walk) + g (expand) is shortest near g = sqrt(K/5), not sqrt(K) -- measured on one
        // recording, boundary walk per iteration: T = 200 000 (K = 1563): g = 8/12/16/20/24/32/40 -> 229/191/175/178/
        // 185/216/253 us; T = 50 000 (K = 391): g = 8/12/16/24 -> 35/37/41/51 us.
        const int g_auto = std::max(4, (int)std::lround(std::sqrt((double)maxchunks / 5.0)));
        if (b->scan_group >= 2) group = b->scan_group;
        // (round 4, 8 / 16 / 24 / 32 / 64 recordings of T = 10 000 on one stream, groups of 4 against the flat chain: walk
        //  25.3 -> 20.5 / 21.6 / 22.5 / 25.4 / 32.5 us, iteration 73.7 -> 69.1, 95.7 -> 90.0, then no gain: up to 16 recordings)
        else if (b->scan_group == 0 && (maxchunks >= b->two_level_from || (b->n_rec <= 16 && maxchunks >= 32)))
            group = g_auto;
    }
    // The wide scan (64 < Sp <= 256, round 6): a walk step there is 0.95 us (fp32) / 1.6 us (fp64) at Sp = 128 and a product of
    // compose_wide_kernel 2.4 us, so the chain (g - 1) products + K / g + g steps is shortest near sqrt(K / 3.5) -- measured,
    // one recording at S = 128, ms per iteration with the flat chain / groups of 4 / 5 / 6 / 8 at T = 10 000: 0.193 / 0.161 /
    // 0.161 / 0.164 / 0.169 (fp64 0.310 / 0.246 / 0.247 / 0.253 / 0.263); flat / 6 / 8 / 10 / 14 / 20 at T = 50 000: 0.616 / 0.371
    // / 0.359 / 0.356 / 0.354 / 0.371 (fp64 1.103 / 0.677 / 0.640 / 0.637 / 0.633 / 0.671); T = 3000 (24 chunks): 0.123 flat
    // against 0.127 in groups of 4 -- the two extra launches are paid from about 40 chunks.
    if (chunked && b->Sp > 64 && b->Sp <= 256) {
        if (b->scan_group >= 2) group = b->scan_group;
        else if (b->scan_group == 0 && maxchunks >= 40) group = std::max(4, (int)std::lround(std::sqrt((double)maxchunks / 3.5)));
    }
    // Round 6: where chunk_post walks the last level itself (FOLD: groups of at most kTileFrames / Sp + 1 chunks) that level
    // costs no launch, so the automatic group size stops there: one recording of T = 20 000 / 30 000 (g = 6 / 7 before, last
    // level a launch of its own) 57.5 -> 54.9 / 64.0 -> 62.0 us per iteration.
    static const bool fold_off = [] { const char* e = experiment_env("VBX_AMD_FOLD_WALK"); return e && e[0] == '0'; }();
    const int fold_max = kTileFrames / std::max(b->Sp, 1) + 1;
    const bool may_fold = group > 1 && fused1 && spt == 1 && b->Sp <= 32 && !fold_off && small_batch_wanted(b) && b->scan_group == 0;
    if (may_fold) group = std::min(group, fold_max);
    // Third level: with products worth ~4 walk steps the chain 4 (g - 1) + 4 (g2 - 1) + K / (g g2) + g2 + g is shortest
    // near g = g2 = (K / 8)^(1/3) rounded up: K = 1563 (T = 200 000): 7 x 7 -> 94 step equivalents against 173 on two
    // levels; K = 391 (T = 50 000): 56 against 84 -- measured walk 36.0 -> 33.7 us (fp64 47.8 -> 39.8), T = 70 000: 40.5 ->
    // 36.9; K = 235 (T = 30 000): 28.9 -> 31.3, the two extra launches cost more than the shorter chain saves.  From 300.
    int group2 = 1;
    if (group > 1) {
        if (b->scan_group2 >= 2) group2 = b->scan_group2;
        else if (b->scan_group2 == 0 && b->scan_group == 0 && maxchunks >= b->three_level_from) {
            group = group2 = std::max(4, (int)std::ceil(std::cbrt((double)maxchunks / 8.0)) + 1);
            // (a first level that folds, and the second as long as the chain 4 (g2 - 1) + K / (g g2) + g2 likes it: T = 100 000 /
            //  120 000, (6, 6) -> (5, 6): 102.3 -> 100.7 / 106.6 -> 104.4 us; from 1200 chunks the cube root wins: T = 200 000,
            //  (7, 7) 148.6 against (5, 8) 150.6.  A two-level walk with a folded first level loses from 300 chunks:
            //  T = 50 000 81.8 against 75.5, T = 100 000 125 against 102)
            if (may_fold && group > fold_max && maxchunks < 1200) {
                group = fold_max;
                group2 = std::max(5, (int)std::lround(std::sqrt((double)maxchunks / (5.0 * group))));
            }
        }
    }
    if (group != b->sgroup || group2 != b->sgroup2 || spt != b->spt || (group > 1 && !b->d_sop) || (group2 > 1 && !b->d_sop2)) {
        for (void* p : {(void*)b->d_sop, (void*)b->d_sopexp, (void*)b->d_sup_rec, (void*)b->d_sup_idx,
                        (void*)b->d_sop2, (void*)b->d_sopexp2, (void*)b->d_sup2_rec, (void*)b->d_sup2_idx}) ctx_free(b->ctx, p);
        b->d_sop = nullptr; b->d_sopexp = nullptr; b->d_sup_rec = nullptr; b->d_sup_idx = nullptr;
        b->d_sop2 = nullptr; b->d_sopexp2 = nullptr; b->d_sup2_rec = nullptr; b->d_sup2_idx = nullptr;
        b->sgroup = group;
        b->sgroup2 = group2;
        b->spt = spt;
        b->nsup_total = b->nsup2_total = 0;
        if (group > 1) {
            std::vector<int> sup_rec, sup_idx, sup2_rec, sup2_idx;
            for (int i = 0; i < b->n_rec; ++i) {
                b->recs[i].sup0 = (int)sup_rec.size();
                b->recs[i].sup20 = (int)sup2_rec.size();
                const int kc = spt == 2 ? (b->recs[i].T + kTileFrames / 2 - 1) / (kTileFrames / 2) : b->recs[i].ntiles;
                const int ns = (kc + group - 1) / group;
                for (int s = 0; s < ns; ++s) {
                    sup_rec.push_back(i);
                    sup_idx.push_back(s);
                }
                if (group2 > 1)
                    for (int s = 0; s < (ns + group2 - 1) / group2; ++s) {
                        sup2_rec.push_back(i);
                        sup2_idx.push_back(s);
                    }
            }
            b->nsup_total = (int)sup_rec.size();
            b->nsup2_total = (int)sup2_rec.size();
            const size_t sp = (size_t)b->Sp;
            int rc = dmalloc_bytes(b->ctx, &b->d_sop, (size_t)b->nsup_total * sp * sp * b->rsize);
            if (rc == VBX_OK) rc = dmalloc(b->ctx, &b->d_sopexp, (size_t)b->nsup_total * sp);
            if (rc == VBX_OK) rc = dmalloc(b->ctx, &b->d_sup_rec, sup_rec.size());
            if (rc == VBX_OK) rc = dmalloc(b->ctx, &b->d_sup_idx, sup_idx.size());
            if (rc != VBX_OK) return rc;
            HIPCHK(b->ctx, hipMemcpy(b->d_sup_rec, sup_rec.data(), sizeof(int) * sup_rec.size(), hipMemcpyHostToDevice));
            HIPCHK(b->ctx, hipMemcpy(b->d_sup_idx, sup_idx.data(), sizeof(int) * sup_idx.size(), hipMemcpyHostToDevice));
            if (group2 > 1) {
                rc = dmalloc_bytes(b->ctx, &b->d_sop2, (size_t)b->nsup2_total * sp * sp * b->rsize);
                if (rc == VBX_OK) rc = dmalloc(b->ctx, &b->d_sopexp2, (size_t)b->nsup2_total * sp);
                if (rc == VBX_OK) rc = dmalloc(b->ctx, &b->d_sup2_rec, sup2_rec.size());
                if (rc == VBX_OK) rc = dmalloc(b->ctx, &b->d_sup2_idx, sup2_idx.size());
                if (rc != VBX_OK) return rc;
                HIPCHK(b->ctx, hipMemcpy(b->d_sup2_rec, sup2_rec.data(), sizeof(int) * sup2_rec.size(), hipMemcpyHostToDevice));
                HIPCHK(b->ctx, hipMemcpy(b->d_sup2_idx, sup2_idx.data(), sizeof(int) * sup2_idx.size(), hipMemcpyHostToDevice));
            }
            b->recs_dirty = true;        // sup0 / sup20 changed
        }
    }
    return VBX_OK;
}

// Workgroup -> tile table of the per-chunk kernels for a batch in which recordings share a rho (an Fa / Fb sweep over
// one recording).  Block b of a grid runs on XCD b % 8 (observed on gfx950; a speed assumption only, nothing depends on
// it for correctness) and each XCD has its own L2, so the tiles that read the same 128 rows of rho -- chunk c of every
// recording of a sharing group -- get block ids with the same residue and consecutive quotients: they are dispatched
// back to back to one XCD, the first one pulls the rho tile from HBM and the others find it in that L2.  Units (group,
// chunk) are dealt to the XCD with the fewest blocks so far; positions left over at the end hold -1 (the block exits).
int build_tile_order(vbx_batch* b) {
    if (!b->order_dirty) return VBX_OK;
    b->order_dirty = false;
    ctx_free(b->ctx, b->d_tile_order);
    b->d_tile_order = nullptr;
    b->nblocks_chunk = b->ntiles_total;
    bool any = false;
    for (int i = 0; i < b->n_rec; ++i) any = any || b->share_src[i] != i;
    if (!any) return VBX_OK;
    std::vector<std::vector<int>> members(b->n_rec);
    for (int i = 0; i < b->n_rec; ++i) members[b->share_src[i]].push_back(i);
    constexpr int kXcds = 8;
    std::vector<std::vector<int>> lists(kXcds);
    for (int owner = 0; owner < b->n_rec; ++owner) {
        if (members[owner].empty()) continue;
        for (int c = 0; c < b->recs[owner].ntiles; ++c) {
            int x = 0;
            for (int y = 1; y < kXcds; ++y)
                if (lists[y].size() < lists[x].size()) x = y;
            for (int m : members[owner]) lists[x].push_back(b->recs[m].tile0 + c);
        }
    }
    size_t len = 0;
    for (auto& l : lists) len = std::max(len, l.size());
    std::vector<int> order(kXcds * len, -1);
    for (int x = 0; x < kXcds; ++x)
        for (size_t k = 0; k < lists[x].size(); ++k) order[kXcds * k + x] = lists[x][k];
    int rc = dmalloc(b->ctx, &b->d_tile_order, order.size());
    if (rc != VBX_OK) return rc;
    HIPCHK(b->ctx, hipMemcpyAsync(b->d_tile_order, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice, b->ctx->stream));
    HIPCHK(b->ctx, hipStreamSynchronize(b->ctx->stream));
    b->nblocks_chunk = (int)order.size();
    return VBX_OK;
}

int upload_recs(vbx_batch* b) {
    if (int rc = build_tile_order(b); rc != VBX_OK) return rc;
    if (!b->recs_dirty) return VBX_OK;
    HIPCHK(b->ctx, hipMemcpyAsync(b->d_recs, b->recs.data(), sizeof(RecDesc) * b->n_rec, hipMemcpyHostToDevice,
                                  b->ctx->stream));
    std::vector<vbx::LpPow> pw;
    if (b->d_lppow) {        // lp^n = mant * 2^fl for n = 0 .. kTileFrames (vbx_operator.hpp: the scaled recursion's factor)
        pw.resize((size_t)b->n_rec * (kTileFrames + 1));
        for (int i = 0; i < b->n_rec; ++i) {
            const double lp = b->recs[i].lp, l2lp = lp > 0.0 ? std::log2(lp) : 0.0;
            for (int n = 0; n <= kTileFrames; ++n) {
                const double l2 = (double)n * l2lp, fl = std::floor(l2);
                pw[(size_t)i * (kTileFrames + 1) + n] = vbx::LpPow{std::exp2(l2 - fl), (int)fl, 0};
            }
        }
        HIPCHK(b->ctx, hipMemcpyAsync(b->d_lppow, pw.data(), sizeof(vbx::LpPow) * pw.size(), hipMemcpyHostToDevice, b->ctx->stream));
    }
    HIPCHK(b->ctx, hipStreamSynchronize(b->ctx->stream));
    b->recs_dirty = false;
    return VBX_OK;
}

int collect_profile(vbx_batch* b) {
    for (size_t i = 0; i < b->ev_used; ++i) {
        float ms = 0.f;
        HIPCHK(b->ctx, hipEventElapsedTime(&ms, b->ev_pool[i].a, b->ev_pool[i].b));
        b->k_ms[b->ev_pool[i].klass] += ms;
        b->k_launches[b->ev_pool[i].klass] += 1;
    }
    b->ev_used = 0;
    return VBX_OK;
}

// host <-> working precision packing --------------------------------------------------
template <typename R, typename SRC>
void pack_matrix(std::vector<R>& dst, const SRC* src, long long rows, int cols, int cols_p, R pad) {
    dst.assign((size_t)rows * cols_p, pad);
    for (long long r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) dst[(size_t)r * cols_p + c] = (R)src[(size_t)r * cols + c];
}

}  // namespace
