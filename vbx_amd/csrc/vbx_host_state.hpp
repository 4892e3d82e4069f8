// vbx_host_state.hpp -- host runtime: the device context (vbx_ctx) and a batch of recordings resident in HBM (vbx_batch)
// (one translation unit with vbx_capi.hip, which includes the parts in order; not a stand-alone header)
#pragma once

namespace {

thread_local std::string g_create_error;

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Environment variables that exist for A/B measurements only (second table): read only when
// VBX_AMD_EXPERIMENT=1, so that a stray variable in a production environment cannot change what the library runs.
inline const char* experiment_env(const char* name) {
    const char* on = std::getenv("VBX_AMD_EXPERIMENT");
    return (on && on[0] == '1') ? std::getenv(name) : nullptr;
}

}  // namespace

// A process gets four hardware compute queues by default; a fifth HIP stream shares one with another, and two busy
// streams on one queue run one after the other (measured: a 4-stream group drops from 211 k to 168 k
// recording-iterations/s when any other stream exists in the process).  Ask for eight before the runtime starts --
// if it has already started (another library initialised HIP first) this does nothing.
// The override is the library's only process-wide side effect; VBX_AMD_HW_QUEUES=0 switches it off (the host application
// keeps whatever it configured), VBX_AMD_HW_QUEUES=<n> asks for another number.  A value the application has already
// put into GPU_MAX_HW_QUEUES is never overwritten.
static const int g_hw_queues_set = [] {
    const char* want = std::getenv("VBX_AMD_HW_QUEUES");
    if (!want || !*want) want = "8";
    return std::strcmp(want, "0") == 0 ? 0 : setenv("GPU_MAX_HW_QUEUES", want, 0);
}();

struct vbx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    std::string err;
    // Device blocks are recycled: the reference's usage is one VBx() call (one batch of ~35 buffers) and one score
    // stage (8 buffers) per recording, and that many hipMalloc / hipFree pairs (each hipFree waits for the device)
    // cost more than the kernels of a short recording.
    std::vector<std::pair<void*, size_t>> spare;              // (block, bytes), kept until vbx_destroy
    size_t spare_bytes = 0;
    std::unordered_map<void*, size_t> live;                    // blocks handed out by ctx_alloc
    // Streams of stream groups are kept for the life of the ctx and handed to one group at a time: the runtime maps a
    // stream to a hardware queue when it is created, and after a few create / destroy cycles two streams of one
    // group ended up on the same queue (measured: 204 k -> 183 k recording-iterations/s for the second batch of a
    // process).
    std::vector<std::pair<hipStream_t, bool>> group_streams;   // (stream, in use)
    bool recycle = true;
    vbx_ctx* pool = nullptr;                                   // the private ctx of a stream-group kid: the ctx whose block lists it
                                                               // allocates from and frees into (round 6: a kid's blocks used to be
                                                               // hipFree'd one by one when its batch closed -- 8 ms per batch of 64)
    std::mutex alloc_mutex;                                    // the block lists: a scores object may be closed by whichever
};                                                             // thread the interpreter's garbage collector runs on

#define HIPCHK(ctx_, call)                                                                    \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            char buf_[512];                                                                    \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                      \
            (ctx_)->err = buf_;                                                                \
            return VBX_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

#define FAIL(ctx_, code_, ...)                    \
    do {                                          \
        char buf_[512];                           \
        snprintf(buf_, sizeof buf_, __VA_ARGS__); \
        (ctx_)->err = buf_;                       \
        return (code_);                           \
    } while (0)

struct EventPair {
    int klass;
    hipEvent_t a, b;
};

struct GroupThreads {
    std::mutex m;
    std::condition_variable go, done;
    long long generation = 0;
    int pending = 0, max_iters = 0;
    double epsilon = 0.0;
    bool quit = false;
    std::vector<int> rc;
    std::vector<std::thread> workers;
};

struct vbx_batch {
    vbx_ctx* ctx = nullptr;
    // Stream groups (VBX_OPT_STREAMS): a batch of many recordings is a parent that owns no device memory but K
    // ordinary batches ("kids"), each with a share of the recordings and its own HIP stream (a private vbx_ctx that
    // differs from the parent's in the stream only).  vbx_batch_run drives every kid from its own host thread: while
    // one kid sits in its latency-bound launches (boundary walk, per-recording reductions) the others keep the CUs
    // busy.
    std::vector<vbx_batch*> kids;
    std::vector<vbx_ctx*> kid_ctx;                // kid_ctx[0] shares the parent's stream
    std::vector<int> kid_of, local_of;            // recording -> kid, index inside the kid
    std::vector<int> root_of;                     // recording -> the recording whose x-vectors it runs on (itself: it was given its own)
    std::vector<int64_t> all_T;
    std::vector<int32_t> all_S;
    std::vector<std::pair<int, int64_t>> options; // options set so far (replayed when the kids are rebuilt)
    int streams = 0;                              // option: 0 auto, >= 1 explicit
    struct GroupThreads* threads = nullptr;       // one sleeping host thread per kid beyond the first
    bool any_set = false;
    std::mutex group_mutex;                       // the bookkeeping of a stream group that setters on different sub-batches share
    // uploads without a synchronize per recording (VBX_OPT_ASYNC_UPLOAD, round 6): the small per-recording arguments go through
    // a pinned host block of the batch, sum_t G_t is fetched for all recordings at once when the next run begins
    bool async_upload = false;
    double* h_args = nullptr;                     // pinned: per recording {Phi[Dp], sqrt Phi[Dp], pi0[Sp]}
    std::vector<char> gsum_pending;               // recording -> its sum_t G_t still lies in d_gtile
    std::vector<char> args_busy;                  // recording -> its slot of h_args may be the source of a copy in flight
    bool has_run = false;                         // an iteration has been launched since the batch was created
    bool uploads_in_flight = false;               // something was enqueued by a setter since the last synchronize
    // host mirrors of the small results, fetched once per run (vbx_batch_get_result then needs no device round trip for them)
    std::vector<RecState> h_state;
    std::vector<double> h_pi, h_Li;
    bool mirrors_valid = false, model_mirrors_valid = false;
    std::vector<double> h_alpha, h_invL;          // [n_rec][Sp][D] (fetch_model_mirrors)
    void* d_fetch = nullptr;                      // device staging of a result fetch that the upload staging block cannot hold
    size_t fetch_bytes = 0;
    int launch_rc = 0;                            // a launch helper found no kernel instance for this batch's padded width (checked in run_end)
    int n_rec = 0, D = 0, Dp = 0, Sp = 0, NT = 0, precision = 0, max_iters = 0;
    size_t rsize = 4;
    long long sum_T = 0;
    int ntiles_total = 0;
    std::vector<RecDesc> recs;
    std::vector<char> is_set;
    bool recs_dirty = true;
    // options
    int fb_algo = VBX_FB_AUTO, check_every = 4, chunk_frames = 0, fuse = 2;
    int split_tiles = 0;                          // option: 0 auto, 1 on, 2 off (VBX_OPT_SPLIT_TILES)
    int gemm = VBX_GEMM_EXACT;                    // option VBX_OPT_GEMM: how the fp32 path multiplies (vbx_split.hpp)
    bool split_now = false;                       // in effect for the launches being issued: f16 operand pairs
    std::vector<char> split_dirty;                // recording -> its rho has changed since its f16 copies were made
    std::vector<char> split_bad;                  // recording -> its rho spans more than kSplitRangeBits between frames (rho_absmax_kernel)
    bool split_declined = false;                  // ... for any recording: the batch multiplies exactly (vbx_batch_gemm_in_effect says so)
    void *d_rho_a = nullptr, *d_rho_b = nullptr, *d_alpha_frag = nullptr;
    int *d_rho_e = nullptr, *d_rho_amax = nullptr, *d_alpha_e = nullptr;
    int64_t profile = 0;                          // bit k: bracket launches of kernel class k with HIP events
    bool mpart_valid = false;                     // mpart/npart hold gamma^T rho of the current gamma (fused path)
    bool gamma_stale = false;                     // fused iterations have run since gamma was last written (run_end replays)
    bool fused_now = false;                       // in effect for the launches being issued: the fused per-chunk kernels
    bool fold_now = false;                        // ... and chunk_post walks the last level of the boundary walk itself (FOLD)
    bool half_ops_now = false;                    // ... and chunk_loglik builds the half-tile operators chunk_post splits its re-run with
    void* d_gamma0 = nullptr;
    double* d_pi_prev = nullptr;
    // device memory
    RecDesc* d_recs = nullptr;
    RecState* d_state = nullptr;                  // two copies of [n_rec] (fin_kernel): the latest one is d_state + state_cur * n_rec
    int state_cur = 0;
    bool fin_pending = false;                     // an iteration has been launched whose finishing role has not run yet
    double run_epsilon = 0.0;
    int *d_tile_rec = nullptr, *d_tile_t0 = nullptr, *d_tile_done = nullptr;
    // recordings that share a rho (vbx_batch_set_recording_shared): who shares with whom, and the workgroup -> tile table
    // that puts the chunks reading one rho tile side by side on one XCD
    std::vector<int> share_src;                   // recording -> the recording whose rho it reads (itself: owns its rho)
    int* d_tile_order = nullptr;
    int nblocks_chunk = 0;                        // grid of the per-chunk kernels (ntiles_total, or the padded table)
    bool order_dirty = false;
    int4* d_tile_desc = nullptr;
    double *d_phi = nullptr, *d_sqrt_phi = nullptr, *d_gtile = nullptr;
    void *d_rho = nullptr, *d_gamma = nullptr, *d_bmat = nullptr, *d_mrow = nullptr, *d_ahat = nullptr,
         *d_bhat = nullptr, *d_alpha = nullptr, *d_invL = nullptr, *d_bias = nullptr, *d_mpart = nullptr,
         *d_npart = nullptr, *d_lraw = nullptr;
    double *d_emodel = nullptr, *d_pi = nullptr, *d_epart = nullptr, *d_Li = nullptr;
    double* d_ip = nullptr;                       // step-level API only; VBx() uses pi (VBx.py:99)
    void *d_fw_scale = nullptr, *d_bw_scale = nullptr;   // step-level API only
    // chunked scan
    void *d_op = nullptr, *d_fbound = nullptr, *d_gbound = nullptr;
    int* d_opexp = nullptr;
    RecState* h_poll = nullptr;                   // pinned host copy of the states for the stop test (run_all_done)
    void* d_cop = nullptr;                        // c of the operator recursion per recording (mstep_fin -> chunk_loglik)
    vbx::LpPow* d_lppow = nullptr;                // lp^n tables of the recordings (host-computed)
    void* d_oph = nullptr;                        // half-tile operators of the fused path (chunk_loglik -> chunk_post)
    int* d_ophexp = nullptr;
    double* d_tllpart = nullptr;
    void* d_sfw = nullptr;
    void* d_dump = nullptr;
    bool use_chunked = false;
    // two-level boundary walk
    int scan_group = 0;                           // option: 0 auto, 1 flat, >= 2 chunks per group
    int two_level_from = 160;
    int sgroup = 1, nsup_total = 0;               // in effect
    int spt = 1;                                  // scan chunks per tile in effect (2: fused kernels, half-tile operators)
    void* d_sop = nullptr;
    int *d_sopexp = nullptr, *d_sup_rec = nullptr, *d_sup_idx = nullptr;
    // third level of the walk (very long recordings): groups of sgroup2 groups
    int scan_group2 = 0;                          // option: 0 auto, 1 off, >= 2 groups per level-2 group
    int three_level_from = 300;                   // chunks from which the automatic choice adds the third level
    int sgroup2 = 1, nsup2_total = 0;             // in effect
    void* d_sop2 = nullptr;
    int *d_sopexp2 = nullptr, *d_sup2_rec = nullptr, *d_sup2_idx = nullptr;
    void* d_xstage = nullptr;
    size_t xstage_bytes = 0;
    // timing
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    double last_ms = 0.0;
    int iters_launched = 0;
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    double k_ms[VBX_K_COUNT] = {0};
    int64_t k_launches[VBX_K_COUNT] = {0};

    template <typename R> BatchView<R> view(double epsilon) const {
        BatchView<R> v;
        v.n_rec = n_rec; v.Sp = Sp; v.Dp = Dp; v.D = D; v.max_iters = max_iters;
        v.ntiles_total = ntiles_total;
        v.recs = d_recs; v.state = d_state + (size_t)state_cur * n_rec; v.state_out = d_state + (size_t)(state_cur ^ 1) * n_rec;
        v.model_stride = (long long)n_rec * Sp * Dp; v.vec_stride = n_rec * Sp;
        v.tile_rec = d_tile_rec; v.tile_t0 = d_tile_t0; v.tile_desc = d_tile_desc; v.tile_done = d_tile_done;
        v.tile_order = d_tile_order;
        v.phi = d_phi;
        v.rho = (R*)d_rho; v.gamma = (R*)d_gamma; v.bmat = (R*)d_bmat; v.mrow = (R*)d_mrow;
        v.ahat = (R*)d_ahat; v.bhat = (R*)d_bhat; v.alpha = (R*)d_alpha; v.invL = (R*)d_invL;
        v.bias = (R*)d_bias; v.bias_lo = (R*)d_bias + (size_t)2 * n_rec * Sp; v.emodel = d_emodel; v.pi = d_pi; v.mpart = (R*)d_mpart;
        v.npart = (R*)d_npart; v.epart = d_epart; v.Li = d_Li; v.epsilon = epsilon;
        v.ip = d_ip ? d_ip : d_pi; v.fw_scale = (R*)d_fw_scale; v.bw_scale = (R*)d_bw_scale;
        v.oph = fused_now && half_ops_now ? (R*)d_oph : nullptr; v.ophexp = d_ophexp;
        v.cop = fused_now ? (R*)d_cop : nullptr; v.lppow = d_lppow;
        v.op = (R*)d_op; v.opexp = d_opexp; v.fbound = (R*)d_fbound; v.gbound = (R*)d_gbound;
        v.tllpart = use_chunked ? d_tllpart : nullptr; v.sfw = (R*)d_sfw; v.dump = (R*)d_dump;
        v.sop = (R*)d_sop; v.sopexp = d_sopexp; v.sup_rec = d_sup_rec; v.sup_idx = d_sup_idx;
        v.sgroup = sgroup; v.nsup_total = nsup_total; v.spt = spt;
        v.sop2 = (R*)d_sop2; v.sopexp2 = d_sopexp2; v.sup2_rec = d_sup2_rec; v.sup2_idx = d_sup2_idx;
        v.sgroup2 = sgroup2; v.nsup2_total = nsup2_total;
        v.gamma0 = fused_now ? (R*)d_gamma0 : nullptr; v.pi_prev = d_pi_prev;
        const bool sp = split_now && fused_now;
        v.rho_a = sp ? (const _Float16*)d_rho_a : nullptr; v.rho_b = sp ? (const _Float16*)d_rho_b : nullptr;
        v.rho_e = sp ? d_rho_e : nullptr; v.alpha_frag = sp ? (_Float16*)d_alpha_frag : nullptr; v.alpha_e = sp ? d_alpha_e : nullptr;
        return v;
    }
};
