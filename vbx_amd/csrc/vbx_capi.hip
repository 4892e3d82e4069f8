// vbx_capi.hip -- host runtime + C ABI (include/vbx_hip.h) of libvbx_hip.so.
//
// The runtime owns: the device context (one device, one stream), the HBM arena of a batch
// of recordings, the launch sequence of one VB iteration (reference: VBx/VBx.py:91-125) and
// the convergence bookkeeping.  Nothing here computes on the CPU except argument packing
// (padding to Sp/Dp, f64 -> working precision) -- there is no CPU fallback.
#include "../../include/vbx_hip.h"
#include "vbx_kernels.hpp"
#include "vbx_scan.hpp"
#include "vbx_scan_wide.hpp"
#include "vbx_fb_dense.hpp"
#include "vbx_chunk_loglik.hpp"
#include "vbx_chunk_post.hpp"
#include "vbx_linkage.hpp"
#include "vbx_ahc.hpp"
#include "vbx_frontend.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace vbx;

namespace {

thread_local std::string g_create_error;

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

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
    bool recycle = true;                                       // false for the private ctx of a stream-group kid
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
    std::vector<int64_t> all_T;
    std::vector<int32_t> all_S;
    std::vector<std::pair<int, int64_t>> options; // options set so far (replayed when the kids are rebuilt)
    int streams = 0;                              // option: 0 auto, >= 1 explicit
    struct GroupThreads* threads = nullptr;       // one sleeping host thread per kid beyond the first
    bool any_set = false;
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
        v.bias = (R*)d_bias; v.emodel = d_emodel; v.pi = d_pi; v.mpart = (R*)d_mpart;
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
    static const int m = [] { const char* e = std::getenv("VBX_AMD_SPLIT_MASK"); return e ? atoi(e) : 3; }();
    return m;
}

#define NT_SWITCH(nt_, BODY)                                   \
    switch (nt_) {                                             \
        case 1: { constexpr int kNT = 1; BODY } break;         \
        case 2: { constexpr int kNT = 2; BODY } break;         \
        case 4: { constexpr int kNT = 4; BODY } break;         \
        case 8: { constexpr int kNT = 8; BODY } break;         \
        case 16: { constexpr int kNT = 16; BODY } break;       \
        default: break;                                        \
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
template <typename R> void launch_fin(vbx_batch* b, double eps, int mode) {
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

// Block size of the per-recording reductions over tiles (mstep_fin, iter_fin): 1024 threads once a recording has more
// partials than the smaller block fetches in a few rounds (one recording of T = 200 000: mstep_fin 42 -> 30 us, iter_fin
// 33 -> 23 us; at T = 50 000 mstep_fin is no faster with 1024 threads, iter_fin 9 -> 8 us).
static int small_kernel_threads(const vbx_batch* b, int from_tiles) {
    int maxtiles = 0;
    for (auto& rd : b->recs) maxtiles = std::max(maxtiles, rd.ntiles);
    return (maxtiles > from_tiles || b->Sp > 256) ? 1024 : 256;        // (iter_fin: a thread per speaker)
}

// chunk_post over the tiles of the batch; REPLAY: the instance that only writes the responsibilities
template <typename R, int SP, bool REPLAY> void launch_chunk_post(vbx_batch* b, const BatchView<R>& v) {
    if constexpr (ChunkPostCfg<R, SP>::kFits) {
        if constexpr (std::is_same<R, float>::value && !REPLAY) {
            if (v.rho_b && (split_debug_mask() & 2)) {       // gamma^T rho on the f16 matrix cores (vbx_split.hpp)
                hipLaunchKernelGGL((chunk_post_kernel<R, SP, false, true>), dim3(b->nblocks_chunk), dim3(256), 0, b->ctx->stream, v);
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
            if constexpr (std::is_same<R, float>::value) {
                if (v.rho_a && (split_debug_mask() & 1)) {   // rho alpha^T on the f16 matrix cores (vbx_split.hpp)
                    hipLaunchKernelGGL((chunk_loglik_kernel<R, SP, true>), dim3(b->nblocks_chunk), dim3(256), 0, st, v);
                    launched = true;
                }
            }
            if (!launched) hipLaunchKernelGGL((chunk_loglik_kernel<R, SP>), dim3(b->nblocks_chunk), dim3(256), 0, st, v);
            have_op = true;
        }
    }
    if (!have_op) {
        LaunchScope ls(b, VBX_K_FB);
        hipLaunchKernelGGL((scan1_kernel<R, SP>), dim3(b->ntiles_total), dim3(SP * SP / 4), 0, st, v);
    }
    {
        LaunchScope ls(b, VBX_K_FB_AUX);
        if (b->sgroup > 1 && b->sgroup2 > 1) {   // very long recordings: groups of groups on top
            hipLaunchKernelGGL((scan_compose_kernel<R, SP>), dim3(b->nsup_total), dim3(256), 0, st, v, 1);
            hipLaunchKernelGGL((scan_compose_kernel<R, SP>), dim3(b->nsup2_total), dim3(256), 0, st, v, 2);
            hipLaunchKernelGGL((scan2_kernel<R, SP>), dim3(b->n_rec, 2), dim3(256), 0, st, v, 4);
            hipLaunchKernelGGL((scan2_kernel<R, SP>), dim3(b->nsup2_total, 2), dim3(256), 0, st, v, 5);
            hipLaunchKernelGGL((scan2_kernel<R, SP>), dim3(b->nsup_total, 2), dim3(256), 0, st, v, 3);
        } else if (b->sgroup > 1) {     // long recordings: group operators, boundaries at the group edges, then inside the groups
            hipLaunchKernelGGL((scan_compose_kernel<R, SP>), dim3(b->nsup_total), dim3(256), 0, st, v, 1);
            hipLaunchKernelGGL((scan2_kernel<R, SP>), dim3(b->n_rec, 2), dim3(256), 0, st, v, 2);
            hipLaunchKernelGGL((scan2_kernel<R, SP>), dim3(b->nsup_total, 2), dim3(256), 0, st, v, 3);
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
        hipLaunchKernelGGL((scan1_wide_kernel<R, SP>), dim3(b->ntiles_total, SP / ScanWideCfg<R, SP>::CB), dim3(256), 0, st, v);
    }
    {
        LaunchScope ls(b, VBX_K_FB_AUX);
        hipLaunchKernelGGL((scan2_wide_kernel<R, SP, 16>), dim3(b->n_rec, 2), dim3(1024), 0, st, v);
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
        default: break;
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
        default: break;
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
void launch_prep(vbx_batch* b, const RecDesc& rd) {
    LaunchScope ls(b, VBX_K_PREP);
    R* rho = (R*)b->d_rho + rd.row0 * b->Dp;
    hipLaunchKernelGGL((prep_kernel<R, XT>), dim3(rd.ntiles), dim3(256), 0, b->ctx->stream,
                       (const XT*)b->d_xstage, (const double*)b->d_sqrt_phi, rho, b->d_gtile + rd.tile0, rd.T,
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
static int ctx_alloc_raw(vbx_ctx* ctx, void** p, size_t bytes) {
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
    HIPCHK(ctx, hipMalloc(p, bytes));
    ctx->live[*p] = bytes;
    return VBX_OK;
}

// Back to the spare list.  Work queued on the ctx stream that still touches the block stays ordered before its next
// use (every user of the list runs on that stream or has waited for it); beyond 4 GB / 256 spares the block is freed.
void ctx_free(vbx_ctx* ctx, void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(ctx->alloc_mutex);
    auto it = ctx->live.find(p);
    const size_t bytes = it == ctx->live.end() ? 0 : it->second;
    if (it != ctx->live.end()) ctx->live.erase(it);
    if (!ctx->recycle || bytes == 0 || ctx->spare_bytes + bytes > ((size_t)4 << 30) || ctx->spare.size() >= 256) {
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
    // Third level: with products worth ~4 walk steps the chain 4 (g - 1) + 4 (g2 - 1) + K / (g g2) + g2 + g is shortest
    // near g = g2 = (K / 8)^(1/3) rounded up: K = 1563 (T = 200 000): 7 x 7 -> 94 step equivalents against 173 on two
    // levels; K = 391 (T = 50 000): 56 against 84 -- measured walk 36.0 -> 33.7 us (fp64 47.8 -> 39.8), T = 70 000: 40.5 ->
    // 36.9; K = 235 (T = 30 000): 28.9 -> 31.3, the two extra launches cost more than the shorter chain saves.  From 300.
    int group2 = 1;
    if (group > 1) {
        if (b->scan_group2 >= 2) group2 = b->scan_group2;
        else if (b->scan_group2 == 0 && b->scan_group == 0 && maxchunks >= b->three_level_from) {
            group = group2 = std::max(4, (int)std::ceil(std::cbrt((double)maxchunks / 8.0)) + 1);
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
    ALLOC(dmalloc_bytes(ctx, &b->d_bias, (size_t)2 * n_rec * b->Sp * rs));
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
        default: FAIL(b->ctx, VBX_ERR_INVALID, "unknown option %d", option);
    }
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
    // Phi, sqrt(Phi) (padded dims: 0)
    std::vector<double> phi(Dp, 0.0), sphi(Dp, 0.0);
    for (int d = 0; X && d < D; ++d) {
        if (!(Phi[d] > 0.0)) FAIL(ctx, VBX_ERR_INVALID, "Phi[%d] must be positive", d);
        phi[d] = Phi[d];
        sphi[d] = std::sqrt(Phi[d]);
    }
    std::vector<double> gt;
    if (X) {
        HIPCHK(ctx, hipMemcpyAsync(b->d_phi + (size_t)rec * Dp, phi.data(), sizeof(double) * Dp, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(b->d_sqrt_phi, sphi.data(), sizeof(double) * Dp, hipMemcpyHostToDevice, ctx->stream));
        // X -> staging -> rho, G
        const size_t xbytes = (size_t)T * D * (x_dtype == VBX_F64 ? 8 : 4);
        HIPCHK(ctx, hipMemcpyAsync(b->d_xstage, X, xbytes, hipMemcpyHostToDevice, ctx->stream));
        if (x_dtype == VBX_F64) launch_prep<R, double>(b, rd); else launch_prep<R, float>(b, rd);
        HIPCHK(ctx, hipGetLastError());
        gt.resize(rd.ntiles);
        HIPCHK(ctx, hipMemcpyAsync(gt.data(), b->d_gtile + rd.tile0, sizeof(double) * rd.ntiles, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        // shared rho: Phi (and with it sum_t G_t) of the recording this one shares its x-vectors with
        const int src = b->share_src[rec];
        HIPCHK(ctx, hipMemcpyAsync(b->d_phi + (size_t)rec * Dp, b->d_phi + (size_t)src * Dp, sizeof(double) * Dp, hipMemcpyDeviceToDevice, ctx->stream));
    }
    // gamma0, pi0 (padded speakers: 0)
    std::vector<R> gp;
    if (g_dtype == VBX_F64) pack_matrix<R, double>(gp, (const double*)gamma0, T, S, Sp, (R)0);
    else pack_matrix<R, float>(gp, (const float*)gamma0, T, S, Sp, (R)0);
    HIPCHK(ctx, hipMemcpyAsync((R*)b->d_gamma + rd.row0 * Sp, gp.data(), sizeof(R) * gp.size(), hipMemcpyHostToDevice, ctx->stream));
    std::vector<double> pip(Sp, 0.0);
    for (int s = 0; s < S; ++s) pip[s] = pi0[s];
    HIPCHK(ctx, hipMemcpyAsync(b->d_pi + (size_t)rec * Sp, pip.data(), sizeof(double) * Sp, hipMemcpyHostToDevice, ctx->stream));
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
    }
    RecState st;
    std::memset(&st, 0, sizeof st);
    HIPCHK(ctx, hipMemcpyAsync(b->d_state + rec, &st, sizeof st, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(b->d_state + b->n_rec + rec, &st, sizeof st, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(b->d_tile_done + rd.tile0, 0, sizeof(int) * rd.ntiles, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // host vectors go out of scope below
    if (X) {
        double gsum = 0.0;
        for (double g : gt) gsum += g;
        rd.gsum = gsum;
    } else {
        rd.gsum = b->recs[b->share_src[rec]].gsum;
    }
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
        if (rc == VBX_OK) rc = dmalloc_bytes(ctx, &b->d_alpha_frag, (size_t)2 * b->n_rec * b->Sp * b->Dp * 4);
        if (rc == VBX_OK) rc = dmalloc(ctx, &b->d_rho_e, (size_t)b->n_rec);
        if (rc == VBX_OK) rc = dmalloc(ctx, &b->d_rho_amax, (size_t)2 * b->n_rec);
        if (rc == VBX_OK) rc = dmalloc(ctx, &b->d_alpha_e, (size_t)2 * b->n_rec * b->Sp);
        if (rc != VBX_OK) return rc;
        HIPCHK(ctx, hipMemsetAsync(b->d_alpha_frag, 0, (size_t)2 * b->n_rec * b->Sp * b->Dp * 4, ctx->stream));
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
    std::vector<RecState> st(b->n_rec);
    HIPCHK(ctx, hipMemcpyAsync(st.data(), b->d_state + (size_t)b->state_cur * b->n_rec, sizeof(RecState) * b->n_rec, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *all_done = true;
    for (auto& s : st) *all_done = *all_done && s.done;
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

extern "C++" {
namespace {
template <typename R>
int get_result_impl(vbx_batch* b, int rec, double* gamma, double* pi, double* Li, int li_cap, int* n_iters,
                    int* warned, double* alpha, double* invL) {
    vbx_ctx* ctx = b->ctx;
    const RecDesc& rd = b->recs[rec];
    const int Sp = b->Sp, Dp = b->Dp, S = rd.S, D = b->D;
    RecState st;
    HIPCHK(ctx, hipMemcpy(&st, b->d_state + (size_t)b->state_cur * b->n_rec + rec, sizeof st, hipMemcpyDeviceToHost));
    if (n_iters) *n_iters = st.n_iters;
    if (warned) *warned = st.warned;
    if (gamma) {
        std::vector<R> g((size_t)rd.T * Sp);
        HIPCHK(ctx, hipMemcpy(g.data(), (R*)b->d_gamma + rd.row0 * Sp, sizeof(R) * g.size(), hipMemcpyDeviceToHost));
        for (long long t = 0; t < rd.T; ++t)
            for (int s = 0; s < S; ++s) gamma[(size_t)t * S + s] = (double)g[(size_t)t * Sp + s];
    }
    if (pi) {
        std::vector<double> p(Sp);
        HIPCHK(ctx, hipMemcpy(p.data(), b->d_pi + (size_t)rec * Sp, sizeof(double) * Sp, hipMemcpyDeviceToHost));
        for (int s = 0; s < S; ++s) pi[s] = p[s];
    }
    if (Li && li_cap > 0) {
        const int n = std::min(std::min(st.n_iters, li_cap), b->max_iters);
        if (n > 0) HIPCHK(ctx, hipMemcpy(Li, b->d_Li + (size_t)rec * b->max_iters, sizeof(double) * n, hipMemcpyDeviceToHost));
    }
    if (alpha || invL) {
        // the model of the last iteration that ran, n_iters - 1, lives in copy (n_iters - 1) & 1 (fin_kernel); before any
        // iteration: copy 0, where a caller's alpha / invL went
        const size_t copy = st.n_iters > 0 ? (size_t)((st.n_iters - 1) & 1) * b->n_rec * Sp * Dp : 0;
        std::vector<R> a((size_t)Sp * Dp), il((size_t)Sp * Dp);
        HIPCHK(ctx, hipMemcpy(a.data(), (R*)b->d_alpha + copy + (size_t)rec * Sp * Dp, sizeof(R) * a.size(), hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(il.data(), (R*)b->d_invL + copy + (size_t)rec * Sp * Dp, sizeof(R) * il.size(), hipMemcpyDeviceToHost));
        for (int s = 0; s < S; ++s)
            for (int d = 0; d < D; ++d) {
                if (alpha) alpha[(size_t)s * D + d] = (double)a[(size_t)s * Dp + d];
                if (invL) invL[(size_t)s * D + d] = (double)il[(size_t)s * Dp + d];
            }
    }
    return VBX_OK;
}
}  // namespace
}  // extern "C++"

static int leaf_get_result(vbx_batch* b, int rec, double* gamma, double* pi, double* Li, int li_cap, int* n_iters,
                         int* warned, double* alpha, double* invL) {
    if (!b) return VBX_ERR_INVALID;
    if (rec < 0 || rec >= b->n_rec) FAIL(b->ctx, VBX_ERR_INVALID, "recording index %d out of range", rec);
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    HIPCHK(b->ctx, hipStreamSynchronize(b->ctx->stream));
    return b->precision == VBX_PREC_FP64
               ? get_result_impl<double>(b, rec, gamma, pi, Li, li_cap, n_iters, warned, alpha, invL)
               : get_result_impl<float>(b, rec, gamma, pi, Li, li_cap, n_iters, warned, alpha, invL);
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

// ---------------------------------------------------------------------------------------
// public batch API: a plain batch (one stream) or a stream group of plain batches
// ---------------------------------------------------------------------------------------
static int auto_streams(int n_rec, long long tiles) {
    const char* env = std::getenv("VBX_AMD_STREAMS");
    if (env && *env) {
        const int k = std::atoi(env);
        if (k >= 1) return std::min(k, std::min(n_rec, 8));
    }
    // measured on 64 recordings of T = 10 000 (NOTES.md, rounds 1-2): 1 / 2 / 3 / 4 streams = 341 / 321 / 312 / 334 us per
    // iteration (three is the robust optimum: the fourth stream brought nothing in any queue configuration tried)
    // -- and only when every stream still has several rounds of workgroups per launch (a chunk = one workgroup)
    return (n_rec >= 24 && tiles >= 1536) ? 3 : (n_rec >= 12 && tiles >= 768) ? 2 : 1;
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
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int k = 0; k < K; ++k) {
        std::sort(members[k].begin(), members[k].end());
        vbx_ctx* kc = new vbx_ctx();                // the parent's device and stream, block lists of its own
        kc->device = ctx->device;
        kc->stream = ctx->stream;
        kc->prop = ctx->prop;
        kc->recycle = false;
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

int vbx_batch_create(vbx_ctx* ctx, int n_rec, const int64_t* T, const int32_t* S, int32_t D, int precision,
                     int max_iters, vbx_batch** out) {
    if (!ctx) return VBX_ERR_INVALID;
    if (!out || !T || !S || n_rec <= 0) FAIL(ctx, VBX_ERR_INVALID, "vbx_batch_create: bad argument");
    long long tiles = 0;
    for (int i = 0; i < n_rec; ++i) tiles += T[i] > 0 ? (T[i] + kTileFrames - 1) / kTileFrames : 0;
    const int K = auto_streams(n_rec, tiles);
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
        const int want = value == 0 ? auto_streams(b->n_rec, tiles) : (int)std::min<int64_t>(value, b->n_rec);
        const int have = group ? (int)b->kids.size() : 1;
        if (want == have) return VBX_OK;
        if (!group) FAIL(b->ctx, VBX_ERR_STATE, "VBX_OPT_STREAMS: this batch was created as a plain batch (set VBX_AMD_STREAMS "
                                                "before vbx_batch_create, or create it with >= 16 recordings)");
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
    b->any_set = true;
    const int k = b->kid_of[rec];
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
    // a stream group deals its recordings to sub-batches with a device arena each: sharing works inside one of them
    const int k = b->kid_of[rec];
    if (b->kid_of[src_rec] != k)
        FAIL(b->ctx, VBX_ERR_UNSUPPORTED, "recordings %d and %d live in different stream sub-batches: create the batch with "
                                          "VBX_OPT_STREAMS = 1 (VBX_AMD_STREAMS=1) to share x-vectors between them", rec, src_rec);
    b->any_set = true;
    return kid_fail(b, k, leaf_set_recording_shared(b->kids[k], b->local_of[rec], b->local_of[src_rec], pi0, gamma0, g_dtype,
                                                    alpha0, invL0, loopProb, Fa, Fb));
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

int vbx_batch_set_recording_resident(vbx_batch* b, int rec, const vbx_xvectors* xv, int64_t row0, const int32_t* labels,
                                     double init_smoothing, const double* Phi, double loopProb, double Fa, double Fb) {
    if (!b) return VBX_ERR_INVALID;
    if (b->kids.empty()) return leaf_set_recording_resident(b, rec, xv, row0, labels, init_smoothing, Phi, loopProb, Fa, Fb);
    if (rec < 0 || rec >= b->n_rec) FAIL(b->ctx, VBX_ERR_INVALID, "recording index %d out of range", rec);
    b->any_set = true;
    const int k = b->kid_of[rec];
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

int vbx_batch_gemm_in_effect(const vbx_batch* b) {
    if (!b) return VBX_GEMM_EXACT;
    const vbx_batch* leaf = b->kids.empty() ? b : b->kids[0];
    return leaf->split_now ? VBX_GEMM_SPLIT : VBX_GEMM_EXACT;
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
        rc = get_result_impl<R>(b, 0, gamma, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr);
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

// ---------------------------------------------------------------------------------------
// score stage of the AHC initialisation (vbhmm.py:135-138)
// ---------------------------------------------------------------------------------------
}  // extern "C"

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
    static const long long kStageMin = [] { const char* e = getenv("VBX_AMD_LINKAGE_STAGE_MIN"); const long long v = e ? atoll(e) : 0; return v >= 64 ? v : 4096LL; }();
    static const bool staged = [] { const char* e = getenv("VBX_AMD_LINKAGE_STAGES"); return !(e && e[0] == '0'); }();
    const char* dev_mode = getenv("VBX_AMD_LINKAGE_DEVICE");           // (read per call: tests compare the two in one process)
    const bool rounds_on = !(dev_mode && std::strcmp(dev_mode, "chain") == 0);
    static const long long kRoundsFrom = [] { const char* e = getenv("VBX_AMD_LINKAGE_ROUNDS_FROM"); const long long v = e ? atoll(e) : 0; return v >= 4 ? v : 256LL; }();
    static const long long kRoundsStop = [] { const char* e = getenv("VBX_AMD_LINKAGE_ROUNDS_STOP"); const long long v = e ? atoll(e) : 0; return v >= 2 ? v : 48LL; }();
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


#ifdef VBX_PHASE_CLOCKS
// instrumentation builds only: the per-tile phase stamps of chunk_post_mid_kernel (tile, {wave 0, wave 2}, 8)
extern "C" int vbx_debug_clocks(long long* out, int n_words) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vbx::g_phase_clocks), (size_t)n_words * sizeof(long long));
}
#endif

// debugging aid (not part of the ABI): copy one of a plain batch's per-tile arrays to the host
//   which = 0 mpart [tiles][Sp][Dp] R, 1 npart [tiles][Sp] R, 2 epart [tiles][Sp] f64, 3 tllpart [tiles] f64, 4 gamma0 [n_rec][Sp] R
extern "C" long long vbx_debug_fetch(vbx_batch* b, int which, void* out, long long cap_bytes) {
    if (!b || !b->kids.empty()) return -1;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    const size_t nt = (size_t)b->ntiles_total, sp = (size_t)b->Sp, dp = (size_t)b->Dp, rs = b->rsize;
    const void* src = nullptr;
    size_t bytes = 0;
    switch (which) {
        case 0: src = b->d_mpart; bytes = nt * sp * dp * rs; break;
        case 1: src = b->d_npart; bytes = nt * sp * rs; break;
        case 2: src = b->d_epart; bytes = nt * sp * 8; break;
        case 3: src = b->d_tllpart; bytes = nt * 8; break;
        case 4: src = b->d_gamma0; bytes = (size_t)b->n_rec * sp * rs; break;
        default: return -1;
    }
    if (!src || (long long)bytes > cap_bytes) return -(long long)bytes;
    if (hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long long)bytes;
}
