// vbx_host_group.hpp -- C ABI: the public batch API -- a plain batch or a stream group of plain batches -- and the one-shot vbx_run
// (one translation unit with vbx_capi.hip, which includes the parts in order; not a stand-alone header)
#pragma once
extern "C" {

// ---------------------------------------------------------------------------------------
// public batch API: a plain batch (one stream) or a stream group of plain batches
// ---------------------------------------------------------------------------------------
static int auto_streams(int n_rec, long long tiles, int max_iters) {
    const char* env = std::getenv("VBX_AMD_STREAMS");
    if (env && *env) {
        const int k = std::atoi(env);
        if (k >= 1) return std::min(k, std::min(n_rec, 8));
    }
    // measured on 64 recordings of T = 10 000 (NOTES.md, rounds 1-2): 1 / 2 / 3 / 4 streams = 341 / 321 / 312 / 334 us per
    // iteration (three is the robust optimum: the fourth stream brought nothing in any queue configuration tried)
    // -- and only when every stream still has several rounds of workgroups per launch (a chunk = one workgroup)
    if (n_rec >= 24 && tiles >= 1536) return 3;
    // Batches that do not fill the chip (round 6, NOTES.md): an iteration there is five dependent launches, each as long as one
    // of its workgroups lives, and two or three sub-batches of >= 150 chunks each run theirs in each other's shadow -- 8
    // recordings of T = 10 000: 63.9 / 60.9 / 57.4 us per iteration on 1 / 2 / 3 streams (fp64 116.8 / 108.9 / 97.8), 4: 54.2 /
    // 50.9, 16: 95.1 / 82.2 / 76.7, 2 x T = 50 000: 91.9 / 84.0; below 150 chunks per stream nothing (8 x T = 2000: 37.9 / 39.5 /
    // 40.5).  A group costs 0.2-0.5 ms per batch (threads, sub-batches, one more synchronize per stream): it pays from about
    // forty iterations, so a batch created for fewer (max_iters: one VBx_batch call with the reference's default of 10)
    // keeps the rule of rounds 1-5.
    if (max_iters >= 40) {
        const int k = (int)std::min<long long>(std::min(3, n_rec), tiles / 150);
        if (k >= 2) return k;
    }
    return (n_rec >= 12 && tiles >= 768) ? 2 : 1;
}

static void group_stop_threads(vbx_batch* b) {
    if (!b->threads) return;
    {
        std::lock_guard<std::mutex> lock(b->threads->m);
        b->threads->quit = true;
    }
    b->threads->go.notify_all();
    for (auto& w : b->threads->workers) w.join();
    delete b->threads;
    b->threads = nullptr;
}

static void group_start_threads(vbx_batch* b) {
    const int K = (int)b->kids.size();
    GroupThreads* g = new GroupThreads();
    g->rc.assign(K, VBX_OK);
    b->threads = g;
    for (int k = 1; k < K; ++k)
        g->workers.emplace_back([b, g, k]() {
            long long seen = 0;
            while (true) {
                int max_iters;
                double epsilon;
                {
                    std::unique_lock<std::mutex> lock(g->m);
                    g->go.wait(lock, [&] { return g->quit || g->generation != seen; });
                    if (g->quit) return;
                    seen = g->generation;
                    max_iters = g->max_iters;
                    epsilon = g->epsilon;
                }
                const int rc = leaf_run(b->kids[k], max_iters, epsilon);
                {
                    std::lock_guard<std::mutex> lock(g->m);
                    g->rc[k] = rc;
                    if (--g->pending == 0) g->done.notify_one();
                }
            }
        });
}

static void group_clear(vbx_batch* b) {
    group_stop_threads(b);
    for (vbx_batch* k : b->kids) leaf_destroy(k);
    b->kids.clear();
    for (size_t i = 0; i < b->kid_ctx.size(); ++i) {
        for (auto& gs : b->ctx->group_streams)                // back to the ctx (not destroyed: see vbx_ctx)
            if (gs.first == b->kid_ctx[i]->stream && i > 0) gs.second = false;
        delete b->kid_ctx[i];
    }
    b->kid_ctx.clear();
}

static int kid_fail(vbx_batch* b, int kid, int rc) {      // the message lives in the kid's private ctx
    if (rc != VBX_OK) b->ctx->err = b->kid_ctx[kid]->err;
    return rc;
}

// (re)build the kids of a group for K streams; recordings are dealt longest first to the least loaded kid
static int group_build(vbx_batch* b, int K) {
    vbx_ctx* ctx = b->ctx;
    group_clear(b);
    const int n = b->n_rec;
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return b->all_T[p] > b->all_T[q]; });
    std::vector<long long> load(K, 0);
    std::vector<std::vector<int>> members(K);
    for (int i : order) {
        const int k = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        members[k].push_back(i);
        load[k] += b->all_T[i];
    }
    b->kid_of.assign(n, 0);
    b->local_of.assign(n, 0);
    b->root_of.resize(n);
    for (int i = 0; i < n; ++i) b->root_of[i] = i;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // the sub-batches take spare blocks of the parent ctx onto OTHER streams: whatever was queued on a block when it was put
    // back (the lists promise ordering on the ctx's own stream only) must have finished
    HIPCHK(ctx, hipDeviceSynchronize());
    for (int k = 0; k < K; ++k) {
        std::sort(members[k].begin(), members[k].end());
        vbx_ctx* kc = new vbx_ctx();                // the parent's device and stream, block lists of its own
        kc->device = ctx->device;
        kc->stream = ctx->stream;
        kc->prop = ctx->prop;
        kc->pool = ctx;                             // blocks come from and go back to the parent's lists
        if (k > 0) {
            kc->stream = nullptr;
            for (auto& gs : ctx->group_streams)
                if (!gs.second) {
                    gs.second = true;
                    kc->stream = gs.first;
                    break;
                }
            if (!kc->stream) {
                hipError_t e = hipStreamCreateWithFlags(&kc->stream, hipStreamNonBlocking);
                if (e != hipSuccess) {
                    delete kc;
                    group_clear(b);
                    ctx->err = std::string("stream group: hipStreamCreate failed: ") + hipGetErrorString(e);
                    return VBX_ERR_HIP;
                }
                ctx->group_streams.emplace_back(kc->stream, true);
            }
        }
        b->kid_ctx.push_back(kc);
        std::vector<int64_t> Tk;
        std::vector<int32_t> Sk;
        for (size_t j = 0; j < members[k].size(); ++j) {
            const int i = members[k][j];
            b->kid_of[i] = k;
            b->local_of[i] = (int)j;
            Tk.push_back(b->all_T[i]);
            Sk.push_back(b->all_S[i]);
        }
        vbx_batch* kid = nullptr;
        int rc = leaf_create(kc, (int)Tk.size(), Tk.data(), Sk.data(), b->D, b->precision, b->max_iters, &kid);
        if (rc != VBX_OK) {
            ctx->err = kc->err;
            group_clear(b);
            return rc;
        }
        b->kids.push_back(kid);
        for (auto& o : b->options)
            if ((rc = leaf_set_option(kid, o.first, o.second)) != VBX_OK) {
                ctx->err = kc->err;
                group_clear(b);
                return rc;
            }
    }
    group_start_threads(b);
    return VBX_OK;
}

// `rec` is about to get other x-vectors than it has (new ones of its own: `own`; or another recording's): whoever runs on
// a copy of its old rows in ANOTHER sub-batch must be set again (inside its own sub-batch own_rho does the same), and the
// bookkeeping of who shares whose rows follows.  One helper for the three setters (round 6, the advisor's finding: the
// resident setter used to skip this and left clones running on stale x-vectors).
static void group_new_rows(vbx_batch* b, int rec, bool own) {
    const int k = b->kid_of[rec];
    if (b->root_of[rec] == rec)
        for (int i = 0; i < b->n_rec; ++i)
            if (i != rec && b->root_of[i] == rec) {
                b->root_of[i] = i;
                if (b->kid_of[i] != k) b->kids[b->kid_of[i]]->is_set[b->local_of[i]] = 0;
            }
    if (own) b->root_of[rec] = rec;
}

int vbx_batch_create(vbx_ctx* ctx, int n_rec, const int64_t* T, const int32_t* S, int32_t D, int precision,
                     int max_iters, vbx_batch** out) {
    return vbx_batch_create_streams(ctx, n_rec, T, S, D, precision, max_iters, 0, out);
}

int vbx_batch_create_streams(vbx_ctx* ctx, int n_rec, const int64_t* T, const int32_t* S, int32_t D, int precision,
                             int max_iters, int streams, vbx_batch** out) {
    if (!ctx) return VBX_ERR_INVALID;
    if (!out || !T || !S || n_rec <= 0) FAIL(ctx, VBX_ERR_INVALID, "vbx_batch_create: bad argument");
    if (streams < 0 || streams > 8) FAIL(ctx, VBX_ERR_INVALID, "vbx_batch_create_streams: streams takes 0 (auto) .. 8");
    long long tiles = 0;
    for (int i = 0; i < n_rec; ++i) tiles += T[i] > 0 ? (T[i] + kTileFrames - 1) / kTileFrames : 0;
    const int K = streams == 0 ? auto_streams(n_rec, tiles, max_iters) : std::min(streams, n_rec);
    if (K <= 1) return leaf_create(ctx, n_rec, T, S, D, precision, max_iters, out);
    *out = nullptr;
    vbx_batch* b = new vbx_batch();
    b->ctx = ctx;
    b->n_rec = n_rec;
    b->D = D;
    b->precision = precision;
    b->max_iters = max_iters;
    b->all_T.assign(T, T + n_rec);
    b->all_S.assign(S, S + n_rec);
    int rc = group_build(b, K);
    if (rc != VBX_OK) {
        delete b;
        return rc;
    }
    *out = b;
    return VBX_OK;
}

int vbx_batch_destroy(vbx_batch* b) {
    if (!b) return VBX_OK;
    if (b->kids.empty() && b->kid_ctx.empty()) return leaf_destroy(b);
    (void)hipSetDevice(b->ctx->device);
    group_clear(b);
    delete b;
    return VBX_OK;
}

int vbx_batch_set_option(vbx_batch* b, int option, int64_t value) {
    if (!b) return VBX_ERR_INVALID;
    if (option == VBX_OPT_STREAMS) {
        if (value < 0 || value > 8) FAIL(b->ctx, VBX_ERR_INVALID, "VBX_OPT_STREAMS takes 0 (auto) .. 8");
        const bool group = !b->kids.empty();
        long long tiles = 0;
        for (int64_t t : b->all_T) tiles += (t + kTileFrames - 1) / kTileFrames;
        const int want = value == 0 ? auto_streams(b->n_rec, tiles, b->max_iters) : (int)std::min<int64_t>(value, b->n_rec);
        const int have = group ? (int)b->kids.size() : 1;
        if (want == have) return VBX_OK;
        if (!group) FAIL(b->ctx, VBX_ERR_STATE, "VBX_OPT_STREAMS: this batch was created on one stream and cannot be regrouped: "
                                                "create it with vbx_batch_create_streams (or set VBX_AMD_STREAMS before vbx_batch_create)");
        if (b->any_set) FAIL(b->ctx, VBX_ERR_STATE, "VBX_OPT_STREAMS must be set before the first recording");
        return group_build(b, std::max(want, 1));
    }
    if (b->kids.empty()) return leaf_set_option(b, option, value);
    for (size_t k = 0; k < b->kids.size(); ++k) {
        const int rc = kid_fail(b, (int)k, leaf_set_option(b->kids[k], option, value));
        if (rc != VBX_OK) return rc;
    }
    b->options.emplace_back(option, value);
    return VBX_OK;
}

int vbx_batch_set_recording(vbx_batch* b, int rec, const void* X, int x_dtype, const double* Phi, const double* pi0,
                            const void* gamma0, int g_dtype, const double* alpha0, const double* invL0,
                            double loopProb, double Fa, double Fb) {
    if (!b) return VBX_ERR_INVALID;
    if (b->kids.empty())
        return leaf_set_recording(b, rec, X, x_dtype, Phi, pi0, gamma0, g_dtype, alpha0, invL0, loopProb, Fa, Fb);
    if (rec < 0 || rec >= b->n_rec) FAIL(b->ctx, VBX_ERR_INVALID, "recording index %d out of range", rec);
    const int k = b->kid_of[rec];
    {
        // (recordings of DIFFERENT sub-batches may be set from different host threads -- each has its own stream and arena, and
        //  a batch call uploads three times as fast that way, vbx_amd/batch.py -- so the bookkeeping they share is guarded)
        std::lock_guard<std::mutex> lock(b->group_mutex);
        b->any_set = true;
        group_new_rows(b, rec, true);
    }
    return kid_fail(b, k, leaf_set_recording(b->kids[k], b->local_of[rec], X, x_dtype, Phi, pi0, gamma0, g_dtype, alpha0,
                                             invL0, loopProb, Fa, Fb));
}

int vbx_batch_set_recording_shared(vbx_batch* b, int rec, int src_rec, const double* pi0, const void* gamma0, int g_dtype,
                                   const double* alpha0, const double* invL0, double loopProb, double Fa, double Fb) {
    if (!b) return VBX_ERR_INVALID;
    if (b->kids.empty())
        return leaf_set_recording_shared(b, rec, src_rec, pi0, gamma0, g_dtype, alpha0, invL0, loopProb, Fa, Fb);
    if (rec < 0 || rec >= b->n_rec || src_rec < 0 || src_rec >= b->n_rec)
        FAIL(b->ctx, VBX_ERR_INVALID, "vbx_batch_set_recording_shared: recording index out of range");
    if (rec == src_rec) FAIL(b->ctx, VBX_ERR_INVALID, "vbx_batch_set_recording_shared: a recording cannot share with itself");
    // A stream group deals its recordings to sub-batches with a device arena each.  Sharing proper works inside one of them;
    // across two, the first recording of a sub-batch that asks for these x-vectors gets a COPY of the rows (device to
    // device) and owns it, the later ones of that sub-batch share with it: one rho per stream, not one per sweep point.
    const int k = b->kid_of[rec], ks = b->kid_of[src_rec];
    const int root = b->root_of[src_rec];
    if (root == rec) FAIL(b->ctx, VBX_ERR_STATE, "recording %d runs on the x-vectors of recording %d: it cannot be that recording's source", src_rec, rec);
    group_new_rows(b, rec, false);                            // (`rec` had x-vectors of its own: its dependants in other sub-batches)
    b->any_set = true;
    int rc;
    if (ks == k) {
        rc = leaf_set_recording_shared(b->kids[k], b->local_of[rec], b->local_of[src_rec], pi0, gamma0, g_dtype, alpha0, invL0, loopProb, Fa, Fb);
    } else {
        int local_src = -1;                                   // a recording of sub-batch k that already holds these x-vectors
        for (int i = 0; i < b->n_rec && local_src < 0; ++i)
            if (i != rec && b->kid_of[i] == k && b->root_of[i] == root && b->kids[k]->is_set[b->local_of[i]] &&
                b->kids[k]->share_src[b->local_of[i]] != b->local_of[rec])        // (not one that reads `rec`'s own copy: `rec` is
                local_src = b->local_of[i];                                       //  the clone owner being set again -> a fresh clone)
        rc = local_src >= 0
                 ? leaf_set_recording_shared(b->kids[k], b->local_of[rec], local_src, pi0, gamma0, g_dtype, alpha0, invL0, loopProb, Fa, Fb)
                 : leaf_set_recording_cloned(b->kids[k], b->local_of[rec], b->kids[ks], b->local_of[src_rec], pi0, gamma0, g_dtype,
                                             alpha0, invL0, loopProb, Fa, Fb);
    }
    if (rc == VBX_OK) b->root_of[rec] = root;
    return kid_fail(b, k, rc);
}

int vbx_batch_run(vbx_batch* b, int max_iters, double epsilon) {
    if (!b) return VBX_ERR_INVALID;
    if (b->kids.empty()) return leaf_run(b, max_iters, epsilon);
    // One host thread per stream, each running the ordinary loop of its sub-batch: launches (five per iteration and
    // stream) are issued in parallel and the streams drift out of phase by themselves.  Fed round-robin from ONE
    // thread (also with the streams started a fraction of a period apart) the same streams gave no gain at all.
    const int K = (int)b->kids.size();
    // The feeding threads are created with the group and sleep between runs: a VBx() call is a few dozen
    // iterations, and starting three threads (with their first HIP call each) cost as much as two of them.
    GroupThreads& g = *b->threads;
    {
        std::lock_guard<std::mutex> lock(g.m);
        g.max_iters = max_iters;
        g.epsilon = epsilon;
        g.pending = K - 1;
        ++g.generation;
    }
    g.go.notify_all();
    g.rc[0] = leaf_run(b->kids[0], max_iters, epsilon);
    {
        std::unique_lock<std::mutex> lock(g.m);
        g.done.wait(lock, [&] { return g.pending == 0; });
    }
    std::vector<int>& rcs = g.rc;
    b->last_ms = 0.0;
    b->iters_launched = 0;
    for (int k = 0; k < K; ++k) {
        if (rcs[k] != VBX_OK) return kid_fail(b, k, rcs[k]);
        b->last_ms = std::max(b->last_ms, b->kids[k]->last_ms);       // the streams start together
        b->iters_launched = std::max(b->iters_launched, b->kids[k]->iters_launched);
    }
    return VBX_OK;
}

int vbx_batch_get_result(vbx_batch* b, int rec, double* gamma, double* pi, double* Li, int li_cap, int* n_iters,
                         int* warned, double* alpha, double* invL) {
    if (!b) return VBX_ERR_INVALID;
    if (b->kids.empty()) return leaf_get_result(b, rec, gamma, pi, Li, li_cap, n_iters, warned, alpha, invL);
    if (rec < 0 || rec >= b->n_rec) FAIL(b->ctx, VBX_ERR_INVALID, "recording index %d out of range", rec);
    const int k = b->kid_of[rec];
    return kid_fail(b, k, leaf_get_result(b->kids[k], b->local_of[rec], gamma, pi, Li, li_cap, n_iters, warned, alpha, invL));
}

int vbx_batch_sync_uploads(vbx_batch* b) {
    if (!b) return VBX_ERR_INVALID;
    if (b->kids.empty()) return sync_uploads(b);
    for (size_t k = 0; k < b->kids.size(); ++k) {
        const int rc = kid_fail(b, (int)k, sync_uploads(b->kids[k]));
        if (rc != VBX_OK) return rc;
    }
    return VBX_OK;
}

int vbx_batch_get_results(vbx_batch* b, int n, vbx_fetch* items) {
    if (!b) return VBX_ERR_INVALID;
    if (n < 0 || (n > 0 && !items)) FAIL(b->ctx, VBX_ERR_INVALID, "vbx_batch_get_results: bad argument");
    const bool group = !b->kids.empty();
    std::vector<char> touched(group ? b->kids.size() : 1, 0);
    int rc = VBX_OK;
    for (int i = 0; i < n && rc == VBX_OK; ++i) {
        vbx_fetch& f = items[i];
        if (f.rec < 0 || f.rec >= b->n_rec) FAIL(b->ctx, VBX_ERR_INVALID, "recording index %d out of range", f.rec);
        const int k = group ? b->kid_of[f.rec] : 0;
        vbx_batch* leaf = group ? b->kids[k] : b;
        rc = leaf_fetch_enqueue(leaf, group ? b->local_of[f.rec] : f.rec, f.gamma, f.pi, f.Li, f.li_cap, &f.n_iters, &f.warned, f.alpha, f.invL);
        if (group) rc = kid_fail(b, k, rc);
        touched[k] = 1;
    }
    for (size_t k = 0; k < touched.size(); ++k) {              // (also after a failure: nothing may be in flight into the
        if (!touched[k]) continue;                             //  caller's arrays when this returns)
        const int rf = group ? kid_fail(b, (int)k, leaf_fetch_finish(b->kids[k])) : leaf_fetch_finish(b);
        if (rc == VBX_OK) rc = rf;
    }
    return rc;
}

// Pinned host memory for results (and inputs): a copy into or out of such a block is a plain DMA at the link's rate and is
// asynchronous -- a copy into pageable memory goes through the runtime's staging buffers at a fraction of it and blocks.
// Process-wide pool (a block may outlive the ctx it was first used with: the Python wrapper frees it from a finaliser).
namespace {
struct HostPool {
    std::mutex m;
    std::vector<std::pair<void*, size_t>> spare;
    std::unordered_map<void*, size_t> live;
    size_t spare_bytes = 0;
};
HostPool& host_pool() { static HostPool* p = new HostPool(); return *p; }     // (never destroyed: finalisers may run at exit)
}  // namespace

int vbx_host_alloc(size_t bytes, void** out) {
    if (!out) return VBX_ERR_INVALID;
    *out = nullptr;
    bytes = std::max<size_t>(bytes, 64);
    HostPool& hp = host_pool();
    std::lock_guard<std::mutex> lock(hp.m);
    int best = -1;
    for (int i = 0; i < (int)hp.spare.size(); ++i)
        if (hp.spare[i].second >= bytes && hp.spare[i].second <= 2 * bytes + 4096 && (best < 0 || hp.spare[i].second < hp.spare[best].second)) best = i;
    if (best >= 0) {
        *out = hp.spare[best].first;
        hp.live[*out] = hp.spare[best].second;
        hp.spare_bytes -= hp.spare[best].second;
        hp.spare.erase(hp.spare.begin() + best);
        return VBX_OK;
    }
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        g_create_error = std::string("vbx_host_alloc: hipHostMalloc failed: ") + hipGetErrorString(e);
        return VBX_ERR_HIP;
    }
    hp.live[p] = bytes;
    *out = p;
    return VBX_OK;
}

int vbx_host_free(void* p) {
    if (!p) return VBX_OK;
    HostPool& hp = host_pool();
    std::lock_guard<std::mutex> lock(hp.m);
    auto it = hp.live.find(p);
    if (it == hp.live.end()) return VBX_ERR_INVALID;
    const size_t bytes = it->second;
    hp.live.erase(it);
    if (hp.spare_bytes + bytes > ((size_t)2 << 30) || hp.spare.size() >= 512) {
        (void)hipHostFree(p);
        return VBX_OK;
    }
    hp.spare.emplace_back(p, bytes);
    hp.spare_bytes += bytes;
    return VBX_OK;
}

int vbx_batch_set_recording_resident(vbx_batch* b, int rec, const vbx_xvectors* xv, int64_t row0, const int32_t* labels,
                                     double init_smoothing, const double* Phi, double loopProb, double Fa, double Fb) {
    if (!b) return VBX_ERR_INVALID;
    if (b->kids.empty()) return leaf_set_recording_resident(b, rec, xv, row0, labels, init_smoothing, Phi, loopProb, Fa, Fb);
    if (rec < 0 || rec >= b->n_rec) FAIL(b->ctx, VBX_ERR_INVALID, "recording index %d out of range", rec);
    b->any_set = true;
    const int k = b->kid_of[rec];
    group_new_rows(b, rec, true);
    return kid_fail(b, k, leaf_set_recording_resident(b->kids[k], b->local_of[rec], xv, row0, labels, init_smoothing, Phi,
                                                      loopProb, Fa, Fb));
}

int vbx_batch_get_labels(vbx_batch* b, int rec, int32_t* first, int32_t* second) {
    if (!b) return VBX_ERR_INVALID;
    if (b->kids.empty()) return leaf_get_labels(b, rec, first, second);
    if (rec < 0 || rec >= b->n_rec) FAIL(b->ctx, VBX_ERR_INVALID, "recording index %d out of range", rec);
    const int k = b->kid_of[rec];
    return kid_fail(b, k, leaf_get_labels(b->kids[k], b->local_of[rec], first, second));
}

int vbx_batch_last_run_ms(vbx_batch* b, double* total_ms, int* iters_launched) { return leaf_last_run_ms(b, total_ms, iters_launched); }

int vbx_batch_streams(const vbx_batch* b) { return !b ? 0 : b->kids.empty() ? 1 : (int)b->kids.size(); }

int vbx_batch_stream_of(const vbx_batch* b, int rec) {
    if (!b || rec < 0 || rec >= b->n_rec) return -1;
    return b->kids.empty() ? 0 : b->kid_of[rec];
}

int vbx_batch_gemm_in_effect(const vbx_batch* b) {
    if (!b) return VBX_GEMM_EXACT;
    if (b->kids.empty()) return b->split_now ? VBX_GEMM_SPLIT : VBX_GEMM_EXACT;
    // a stream group: the range guard of the split mode (prepare_split) acts per sub-batch; "split" = every one of them
    for (const vbx_batch* k : b->kids)
        if (!k->split_now) return VBX_GEMM_EXACT;
    return VBX_GEMM_SPLIT;
}

int vbx_batch_kernel_times(vbx_batch* b, double* ms, int64_t* launches) {
    if (!b) return VBX_ERR_INVALID;
    if (b->kids.empty()) return leaf_kernel_times(b, ms, launches);
    for (int c = 0; c < VBX_K_COUNT; ++c) {                   // summed over the streams: ms / launches = mean duration of
        double t = 0.0;                                       // one launch (of a kid's share of the recordings)
        int64_t n = 0;
        for (vbx_batch* k : b->kids) {
            t += k->k_ms[c];
            n += k->k_launches[c];
        }
        if (ms) ms[c] = t;
        if (launches) launches[c] = n;
    }
    return VBX_OK;
}

int vbx_run(vbx_ctx* ctx, const vbx_problem* p, vbx_result* r) {
    if (!ctx) return VBX_ERR_INVALID;
    if (!p || !r) FAIL(ctx, VBX_ERR_INVALID, "vbx_run: NULL problem/result");
    vbx_batch* b = nullptr;
    int64_t T = p->T;
    int32_t S = p->S;
    int rc = vbx_batch_create(ctx, 1, &T, &S, p->D, p->precision, p->max_iters, &b);
    if (rc != VBX_OK) return rc;
    rc = vbx_batch_set_option(b, VBX_OPT_FB_ALGO, p->fb_algo);
    if (rc == VBX_OK)
        rc = vbx_batch_set_recording(b, 0, p->X, p->x_dtype, p->Phi, p->pi0, p->gamma0, p->g_dtype, p->alpha0,
                                     p->invL0, p->loopProb, p->Fa, p->Fb);
    if (rc == VBX_OK) rc = vbx_batch_run(b, p->max_iters, p->epsilon);
    if (rc == VBX_OK)
        rc = vbx_batch_get_result(b, 0, r->gamma, r->pi, r->Li, p->max_iters, &r->n_iters, &r->warned, r->alpha,
                                  r->invL);
    if (rc == VBX_OK) r->run_ms = b->last_ms;
    vbx_batch_destroy(b);
    return rc;
}

}  // extern "C"
