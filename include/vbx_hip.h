/*
 * vbx_hip.h -- C ABI of libvbx_hip.so, the MI355X (gfx950) implementation of the VBx
 * variational-Bayes HMM E/M loop.
 *
 * The reference has no FFI: the path is the pure-Python function
 *     VBx(X, Phi, loopProb, Fa, Fb, pi, gamma, maxIters, epsilon, alphaQInit, ref, plot,
 *         return_model, alpha, invL) -> (gamma, pi, Li[, alpha, invL])
 * at /root/reference/VBx/VBx.py:27-126, called from /root/reference/VBx/vbhmm.py:154-158.
 * The entry points below are what a ctypes binding for that function binds; the Python
 * mirror of the reference interface (vbx_amd/VBx.py) is a thin marshalling layer on top.
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a negative
 *     VBX_ERR_* code and never throws; vbx_last_error() gives the message.
 *   - the caller owns every host buffer; inputs are read-only; outputs are written
 *     into caller-allocated arrays.  Device memory is owned by the ctx / batch objects.
 *   - matrices are C-contiguous row-major, exactly as NumPy hands them over:
 *     X[T][D], gamma[T][S] (frames x speakers, VBx.py:82,85), alpha/invL[S][D].
 *   - a ctx is bound to one device and one HIP stream; not thread-safe per ctx.
 *   - "recording" = one x-vector sequence = one VBx() call in the reference.
 */
#ifndef VBX_HIP_H
#define VBX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VBX_ABI_VERSION 7

/* error codes */
#define VBX_OK 0
#define VBX_ERR_INVALID (-1)     /* bad argument (shape, NULL, range)                      */
#define VBX_ERR_UNSUPPORTED (-2) /* e.g. S > VBX_MAX_SPEAKERS                              */
#define VBX_ERR_HIP (-3)         /* a HIP runtime call failed; see vbx_last_error          */
#define VBX_ERR_NO_DEVICE (-4)   /* no gfx950-class GPU visible                            */
#define VBX_ERR_STATE (-5)       /* call order violated (run before every recording is set) */

#define VBX_MAX_SPEAKERS 16384 /* states per recording (the reference: any; up to 64 fused kernels, 256 wide scan, 1024 one-wavefront
                                  walk, beyond a workgroup-wide walk: vbx_big.hpp) */

/* element types of caller buffers */
#define VBX_F32 0
#define VBX_F64 1

/* arithmetic of the device path (bench.py "dtype") */
#define VBX_PREC_FP32 0 /* f32 storage + f32 MFMA; f64 for every reduction over T and the ELBO */
#define VBX_PREC_FP64 1 /* f64 storage + f64 MFMA; matches the reference's iteration count   */

/* forward-backward algorithm selector (vbx_batch_set_option VBX_OPT_FB_ALGO) */
#define VBX_FB_AUTO 0
#define VBX_FB_SEQUENTIAL 1 /* one wavefront per direction walks all T frames           */
#define VBX_FB_CHUNKED 2    /* exact chunked parallel scan over T (transfer operators)   */

/* options for vbx_batch_set_option */
#define VBX_OPT_FB_ALGO 1
#define VBX_OPT_CHECK_EVERY 2   /* iterations launched between two convergence polls (default 4) */
#define VBX_OPT_PROFILE 3       /* 0: off; 1: bracket every kernel launch with HIP events;
                                   2*mask: only the kernel classes whose bit (1 << VBX_K_*) is set in mask */
#define VBX_OPT_CHUNK_FRAMES 4  /* frames per scan chunk for VBX_FB_CHUNKED (0 = auto)         */
#define VBX_OPT_SCAN_GROUP 6    /* chunks per group of the two-level boundary walk: 0 auto (sqrt(chunks / 5) once a
                                   recording has >= 160 chunks, or >= 32 in a batch of <= 16; at most 128 / Sp + 1 where
                                   chunk_post walks the last level itself: up to 32 padded states in a batch of <= 2048
                                   chunks; 64 < Sp <= 256: sqrt(chunks / 3.5) from 40 chunks), 1 flat chain, >= 2 explicit */
#define VBX_OPT_SCAN_GROUP2 12   /* groups per level-2 group of the THREE-level walk: 0 auto (from 300 chunks per recording --
                                   T = 38 400 -- both group sizes become (chunks / 8)^(1/3) + 1), 1 off, >= 2 explicit (on top of
                                   the VBX_OPT_SCAN_GROUP in effect).  Where chunk_post walks the last level itself and the
                                   cube root is too long for it, up to 1200 chunks: 128 / Sp + 1 and sqrt(chunks / (5 group)) */
#define VBX_OPT_THREE_LEVEL_FROM 13 /* chunk count from which VBX_OPT_SCAN_GROUP2 = 0 adds the third level            */
#define VBX_OPT_SPLIT_TILES 11  /* fused path: tiles re-run as two halves side by side (four 64-frame chains per tile instead of two
                                   128-frame ones; chunk_loglik hands the half-tile operators over).  0 = auto (on), 1 = on,
                                   2 = off; env VBX_AMD_SPLIT_TILES overrides                                             */
#define VBX_OPT_STREAMS 10      /* HIP streams of a batch: its recordings are dealt to that many sub-batches, one
                                   iteration of each is launched stream after stream, so the latency-bound launches of
                                   one overlap the bandwidth-bound ones of the others.  0 = auto (3 from 24 recordings and
                                   1536 chunks; a batch created for max_iters >= 40: min(3, recordings, chunks / 150) if
                                   that is >= 2; else 2 from 12 recordings and 768 chunks, else 1; env VBX_AMD_STREAMS overrides).  Before the
                                   first recording. */
#define VBX_OPT_TWO_LEVEL_FROM 8 /* chunk count from which VBX_OPT_SCAN_GROUP = 0 picks the two-level walk           */
#define VBX_OPT_GEMM 14         /* how the fp32 path multiplies rho alpha^T (VBx.py:97) and gamma^T rho (VBx.py:96):
                                   VBX_GEMM_EXACT (default) v_mfma_f32_16x16x4_f32, the exact f32 product at the f32 vector
                                   rate; VBX_GEMM_SPLIT v_mfma_f32_16x16x32_f16 on error-compensated f16 operand pairs
                                   (x 2^e = hi + lo, three products, f32 accumulation: 2^-22 per product; rho kept in HBM
                                   as such pairs in MFMA fragment order -- vbx_amd/csrc/vbx_split.hpp).  fp32 batches on
                                   the fused kernels (S <= 64) only; ignored elsewhere.  env VBX_AMD_GEMM=exact|split   */
#define VBX_OPT_ASYNC_UPLOAD 15 /* 1: vbx_batch_set_recording / _shared only ENQUEUE the upload (no synchronize per recording):
                                   the caller's X and gamma0 must stay valid until the next vbx_batch_run or
                                   vbx_batch_sync_uploads returns.  0 (default): every setter returns with the caller's
                                   buffers free, as in ABI <= 6.  What a batch call saves: 64 recordings of T = 10 000 go up in
                                   ONE synchronize instead of 64 (ABI 7) */
#define VBX_GEMM_EXACT 0
#define VBX_GEMM_SPLIT 1
#define VBX_OPT_FUSE 5          /* per-chunk fused kernels when the lattices fit in LDS: 0 none, 1 chunk_post,
                                   2 (default) chunk_post + chunk_loglik.  On the fused path the responsibilities are
                                   written once, when vbx_batch_run returns (they are not needed between iterations) */

typedef struct vbx_ctx vbx_ctx;
typedef struct vbx_batch vbx_batch;

/* kernel classes reported by vbx_batch_kernel_times (one slot per HIP kernel) */
enum {
    VBX_K_PREP = 0,       /* rho = X*sqrt(Phi), sum_t G_t            VBx.py:87-89  */
    VBX_K_MSTEP_ACC = 1,  /* gamma^T rho partial sums (MFMA)         VBx.py:96     */
    VBX_K_MSTEP_FIN = 2,  /* invL, alpha, per-speaker bias           VBx.py:95-97  */
    VBX_K_LOGLIK = 3,     /* rho alpha^T + bias, row max, exp (MFMA) VBx.py:97     */
    VBX_K_FB = 4,         /* forward-backward recursion              VBx.py:146-175 */
    VBX_K_FB_AUX = 5,     /* chunk-boundary propagation (chunked scan only)         */
    VBX_K_POST = 6,       /* gamma, pi statistics                    VBx.py:101-103,174 */
    VBX_K_ITER_FIN = 7,   /* ELBO, pi update, convergence test       VBx.py:100-105,122-125 */
    VBX_K_CHUNK_LOGLIK = 8, /* fused: log-likelihoods + chunk transfer operator  VBx.py:97,167-171 */
    VBX_K_CHUNK_POST = 9, /* fused: chunk re-run + gamma + pi statistics + next gamma^T rho  VBx.py:96,101-103,167-174 */
    VBX_K_COUNT = 10
};

int vbx_abi_version(void);

/* ---- context -------------------------------------------------------------------- */
int vbx_create(vbx_ctx** out, int device);
int vbx_destroy(vbx_ctx* ctx);
/* message of the last failure on this ctx (ctx == NULL: last failure of vbx_create). */
const char* vbx_last_error(const vbx_ctx* ctx);
/* device name (NUL-terminated, truncated to cap), compute units, HBM bytes. */
int vbx_device_info(vbx_ctx* ctx, char* name, int cap, int* compute_units, int64_t* hbm_bytes);

/* ---- batch of recordings resident in HBM -------------------------------------------
 * T[b] frames, S[b] speakers (HMM states) per recording, common feature dim D.
 * max_iters bounds the ELBO history kept on the device. */
int vbx_batch_create(vbx_ctx* ctx, int n_rec, const int64_t* T, const int32_t* S, int32_t D,
                     int precision, int max_iters, vbx_batch** out);
/* The same with the number of HIP streams (sub-batches, VBX_OPT_STREAMS) chosen at creation: 0 = the library's choice as
 * vbx_batch_create makes it (VBX_OPT_STREAMS: three streams from 24 recordings and 1536 chunks, for max_iters >= 40 also two or
 * three from 150 chunks per stream, else two from 12 recordings and 768 chunks, else one; env VBX_AMD_STREAMS overrides), 1 .. 8 = that many (at most one per recording).  A batch created on one stream by the
 * automatic choice cannot be regrouped later (VBX_OPT_STREAMS regroups stream groups only): a sweep over one long recording
 * -- few "recordings", many chunks -- asks for its streams here (vbx_batch_set_recording_shared).  ABI 6. */
int vbx_batch_create_streams(vbx_ctx* ctx, int n_rec, const int64_t* T, const int32_t* S, int32_t D,
                             int precision, int max_iters, int streams, vbx_batch** out);
int vbx_batch_destroy(vbx_batch* batch);
int vbx_batch_set_option(vbx_batch* batch, int option, int64_t value);

/* Upload one recording and its hyper-parameters; computes rho and sum_t G_t on the device.
 *   X      [T][D]  x_dtype            (VBx.py:27 "X")
 *   Phi    [D]     f64                (VBx.py:27 "Phi")
 *   pi0    [S]     f64                (VBx.py:76-77; the caller expands an int pi to ones/pi)
 *   gamma0 [T][S]  g_dtype            (VBx.py:79-85; the caller draws the RNG init)
 *   alpha0, invL0 [S][D] f64 or both NULL (VBx.py:94: skip the first M-step when given)  */
int vbx_batch_set_recording(vbx_batch* batch, int b, const void* X, int x_dtype, const double* Phi,
                            const double* pi0, const void* gamma0, int g_dtype,
                            const double* alpha0, const double* invL0, double loopProb, double Fa,
                            double Fb);

/* Recording b on the x-vectors of recording src_b of the same batch (same T; src_b set before): a sweep of Fa / Fb /
 * loopProb over one recording -- the hyper-parameter grids of DIHARD2_run.sh:42-47, AMI_run.sh:44-49,
 * CALLHOME_run.sh:42-47 around one VBx.py:87-89 rho -- keeps ONE rho in HBM, and the per-chunk kernels run the chunks that
 * read the same rows of it side by side on one XCD, so that HBM delivers them once per kernel instead of once per sweep
 * point.  Everything else (pi0, gamma0, alpha0 / invL0, results) is per recording as in vbx_batch_set_recording.  Setting
 * src_b again unsets the recordings that share with it.  In a batch that runs on several streams (VBX_OPT_STREAMS) every
 * stream's sub-batch keeps ONE copy of the rows (made device to device when its first recording asks for them), shared by
 * the sweep points dealt to that stream: the latency-bound launches of one stream (boundary walk, per-recording
 * reductions) then hide behind the per-chunk kernels of the others. */
int vbx_batch_set_recording_shared(vbx_batch* batch, int b, int src_b, const double* pi0, const void* gamma0, int g_dtype,
                                   const double* alpha0, const double* invL0, double loopProb, double Fa, double Fb);

/* Run up to max_iters VB iterations on every recording (VBx.py:91-125).  A recording
 * stops on its own when ii > 0 and ELBO - previous < epsilon (VBx.py:122-125); its state
 * is frozen from then on.  Returns after the device work has completed. */
int vbx_batch_run(vbx_batch* batch, int max_iters, double epsilon);

/* Results of recording b.  Any pointer may be NULL to skip that output.
 *   gamma [T][S] f64, pi [S] f64, Li [n_iters] f64 (ELBO per iteration, VBx.py:105),
 *   n_iters, warned (1 iff the last ELBO step was negative: the reference prints
 *   'WARNING: Value of auxiliary function has decreased!', VBx.py:123-124),
 *   alpha/invL [S][D] f64 (VBx.py:126 return_model). */
int vbx_batch_get_result(vbx_batch* batch, int b, double* gamma, double* pi, double* Li, int li_cap,
                         int* n_iters, int* warned, double* alpha, double* invL);

/* Wait for the uploads enqueued under VBX_OPT_ASYNC_UPLOAD (vbx_batch_run does the same when it begins).  ABI 7. */
int vbx_batch_sync_uploads(vbx_batch* batch);

/* Results of MANY recordings with one synchronize per stream: every item is what vbx_batch_get_result takes for recording
 * `rec` (any pointer may be NULL), n_iters / warned are filled in.  The copies into `gamma`, `alpha`, `invL` are plain DMA when
 * those point into memory from vbx_host_alloc (pinned), staged and blocking otherwise.  Replaces the per-recording loop a
 * batched caller of the reference would write around VBx.py:126's return.  ABI 7. */
typedef struct {
    int32_t rec;
    double* gamma;   /* [T][S] or NULL */
    double* pi;      /* [S] or NULL */
    double* Li;      /* [li_cap] or NULL */
    int32_t li_cap;
    int32_t n_iters; /* out */
    int32_t warned;  /* out */
    double* alpha;   /* [S][D] or NULL */
    double* invL;    /* [S][D] or NULL */
} vbx_fetch;
int vbx_batch_get_results(vbx_batch* batch, int n, vbx_fetch* items);

/* Pinned (page-locked) host memory from a process-wide pool: the destination of choice for vbx_batch_get_results and a fast
 * source for vbx_batch_set_recording.  A freed block goes back to the pool (at most 2 GB are kept).  A failure's message is
 * vbx_last_error(NULL).  ABI 7. */
int vbx_host_alloc(size_t bytes, void** out);
int vbx_host_free(void* p);

/* Wall time of the last vbx_batch_run measured with HIP events on the batch's stream
 * (ms), and the number of iterations that were launched. */
int vbx_batch_last_run_ms(vbx_batch* batch, double* total_ms, int* iters_launched);
/* With VBX_OPT_PROFILE=1: summed HIP-event time (ms) and launch count per kernel class
 * for the last run; arrays of VBX_K_COUNT entries. */
int vbx_batch_kernel_times(vbx_batch* batch, double* ms, int64_t* launches);
/* Number of HIP streams (sub-batches) this batch runs on: VBX_OPT_STREAMS in effect. */
int vbx_batch_streams(const vbx_batch* b);
/* The stream (sub-batch, 0 .. vbx_batch_streams - 1) recording b was dealt to; -1: no such recording.  vbx_batch_set_recording may
 * be called from different host threads for recordings of DIFFERENT streams (each stream has its own device arena): that is
 * how a batch call keeps the host link busy.  ABI 7. */
int vbx_batch_stream_of(const vbx_batch* b, int rec);
/* VBX_GEMM_EXACT or VBX_GEMM_SPLIT: how the iterations of the last vbx_batch_run multiplied (VBX_OPT_GEMM asks, the
 * batch's precision and kernels decide: fp64 batches, S > 64 and the unfused kernels always answer VBX_GEMM_EXACT -- and so
 * does a batch that holds a recording whose frames span more than 2^10 in magnitude, which one power-of-two scale per
 * recording cannot carry at 22 bits: vbx_split.hpp.  In a batch on several streams the guard acts per sub-batch and the
 * answer is VBX_GEMM_SPLIT only if every sub-batch multiplied that way). */
int vbx_batch_gemm_in_effect(const vbx_batch* b);

/* ---- one-shot: a single recording, host buffers in / out (= one reference VBx call) ---- */
typedef struct {
    int64_t T;
    int32_t D, S;
    const void* X;        /* [T][D] */
    int32_t x_dtype;      /* VBX_F32 / VBX_F64 */
    const double* Phi;    /* [D] */
    const double* pi0;    /* [S] */
    const void* gamma0;   /* [T][S] */
    int32_t g_dtype;
    const double* alpha0; /* [S][D] or NULL */
    const double* invL0;  /* [S][D] or NULL */
    double loopProb, Fa, Fb;
    int32_t max_iters;
    double epsilon;
    int32_t precision;    /* VBX_PREC_* */
    int32_t fb_algo;      /* VBX_FB_* */
} vbx_problem;

typedef struct {
    double* gamma;   /* [T][S] */
    double* pi;      /* [S] */
    double* Li;      /* [max_iters] */
    double* alpha;   /* [S][D] or NULL */
    double* invL;    /* [S][D] or NULL */
    int32_t n_iters;
    int32_t warned;
    double run_ms;   /* device time of the iteration loop */
} vbx_result;

int vbx_run(vbx_ctx* ctx, const vbx_problem* problem, vbx_result* result);

/* ---- step-level entry points (used by the parity tests) ------------------------------
 * forward_backward (VBx.py:146-175) for transition matrices of the form VBx.py:98 builds,
 *   tr = I*loopProb + (1-loopProb)*pi      (every column j constant off the diagonal),
 * and initial-state probabilities ip (NULL: ip = pi, as VBx.py:99 passes them).
 *   lls [T][S] f64, pi [S], ip [S]  ->  gamma [T][S], tll, entered [S], lfw [T][S], lbw [T][S]
 * entered[j] = sum_{t>=1} exp(LSE_i lfw[t-1,i] + lls[t,j] + lbw[t,j] - tll) is the statistic
 * of VBx.py:101-103.  Any output pointer may be NULL. */
int vbx_forward_backward(vbx_ctx* ctx, int64_t T, int32_t S, const double* lls, const double* pi,
                         const double* ip, double loopProb, int precision, int fb_algo, double* gamma,
                         double* tll, double* entered, double* lfw, double* lbw);
/* forward_backward (VBx.py:146-175) for ANY transition matrix tr [S][S] (row i: from state i) and initial-state
 * probabilities ip [S]; eps = 1e-8 is added to both as VBx.py:158,163 do.  lls [T][S] -> gamma [T][S] (state
 * posteriors), tll, lfw [T][S], lbw [T][S]; any output pointer may be NULL.  One dependent S x S mat-vec per frame. */
int vbx_forward_backward_dense(vbx_ctx* ctx, int64_t T, int32_t S, const double* lls, const double* tr, const double* ip,
                               int precision, double* gamma, double* tll, double* lfw, double* lbw);
/* M-step (VBx.py:95-96): gamma [T][S], X [T][D], Phi [D] -> alpha, invL [S][D]. */
int vbx_mstep(vbx_ctx* ctx, int64_t T, int32_t S, int32_t D, const double* X, const double* Phi,
              const double* gamma, double Fa, double Fb, int precision, double* alpha, double* invL);
/* per-frame log-likelihoods (VBx.py:97): X, Phi, alpha, invL -> log_p [T][S] (G included). */
int vbx_loglik(vbx_ctx* ctx, int64_t T, int32_t S, int32_t D, const double* X, const double* Phi,
               const double* alpha, const double* invL, double Fa, int precision, double* log_p);

/* ---- the steps of the driver either side of VBx(), for all x-vectors of an archive at once -------------------------
 * vbhmm.py:125-129  xproj = l2_norm( l2_norm(x - mean1) lda - mean2 )       [n][Dl]       input of cos_similarity
 * vbhmm.py:153      fea   = (xproj - plda_mu) plda_tr^T [:, :fea_dim]        [n][fea_dim]  input of VBx()
 * Both stay resident in HBM (vbx_xvectors); x [n][Din] x_dtype, mean1 [Din], lda [Din][Dl], mean2 [Dl],
 * plda_mu [Dl], plda_tr [Dl][Dl] (row k = k-th output dim, i.e. the array vbhmm.py:113 calls plda_tr): f64. */
typedef struct vbx_xvectors vbx_xvectors;
int vbx_xvectors_project(vbx_ctx* ctx, int64_t n, int32_t Din, int32_t Dl, int32_t fea_dim, const void* x, int x_dtype,
                         const double* mean1, const double* lda, const double* mean2, const double* plda_mu,
                         const double* plda_tr, vbx_xvectors** out);
/* rows [row0, row0 + nrows) of xproj (which = 0, [nrows][Dl]) or fea (which = 1, [nrows][fea_dim]) to the host. */
int vbx_xvectors_get(vbx_xvectors* xv, int which, int64_t row0, int64_t nrows, double* out);
int vbx_xvectors_destroy(vbx_xvectors* xv);
/* cos_similarity (diarization_lib.py:190-213) of the resident rows [row0, row0 + T) of xproj: nothing is uploaded. */
int vbx_cos_similarity_resident(vbx_ctx* ctx, vbx_xvectors* xv, int64_t row0, int64_t T, struct vbx_scores** out);
/* Recording b of a batch from resident rows of fea and the AHC labels [T] (in [0, S[b])): the initial responsibilities
 * softmax(init_smoothing * onehot(labels)) of vbhmm.py:150-152 are built on the device, pi0 = 1/S (VBx.py:76). */
int vbx_batch_set_recording_resident(vbx_batch* batch, int b, const vbx_xvectors* xv, int64_t row0, const int32_t* labels,
                                     double init_smoothing, const double* Phi, double loopProb, double Fa, double Fb);
/* vbhmm.py:160-162: np.argsort(-gamma, axis=1)[:, 0] and [:, 1] of recording b, [T] each (second: -1 when S == 1);
 * equal responsibilities keep their index order (numpy leaves the order of ties unspecified).  Either may be NULL. */
int vbx_batch_get_labels(vbx_batch* batch, int b, int32_t* first, int32_t* second);

/* ---- score stage of the AHC initialisation (next row upstream of VBx(): vbhmm.py:135-138) -------------
 * cos_similarity(x)            /root/reference/VBx/diarization_lib.py:190-213
 * twoGMMcalib_lin(s, niters)   /root/reference/VBx/diarization_lib.py:13-31
 * The T x T score matrix stays resident in HBM between the two calls (vbx_scores). */
typedef struct vbx_scores vbx_scores;

/* x [T][D] f64 (rows need not be normalised) -> device-resident matrix of cosine similarities. */
int vbx_cos_similarity(vbx_ctx* ctx, int64_t T, int32_t D, const double* x, vbx_scores** out);
/* A device-resident copy of n host scores (for callers that bring their own scores). */
int vbx_scores_upload(vbx_ctx* ctx, int64_t n, const double* s, vbx_scores** out);
/* Number of scores held (T*T after vbx_cos_similarity). */
int64_t vbx_scores_count(const vbx_scores* sc);
/* Copy count scores starting at offset to the host (out [count]). */
int vbx_scores_get(vbx_scores* sc, int64_t offset, int64_t count, double* out);
/* The strict upper triangle of scale * S, row by row -- the vector form scipy.spatial.distance.squareform(m,
 * checks=False) gives -- of a T x T score matrix: vbhmm.py:139 `scr_mx = squareform(-scr_mx, checks=False)` is
 * scale = -1.  out: [T (T - 1) / 2]. */
int vbx_scores_get_condensed(vbx_scores* sc, int64_t T, double scale, double* out);

/* vbhmm.py:139-141 on the device: average linkage of the distances -S, S = the T x T matrix of similarities held by sc,
 * by a nearest-neighbour chain that one persistent workgroup walks on the matrix where it lies (no condensed copy, no
 * PCIe traffic but the T - 1 merges).  Z [T - 1][4]: the linkage matrix vbx_linkage_average gives for
 * squareform(-S), bit for bit.  CONSUMES the scores: afterwards sc holds the distances of the merged clusters. */
int vbx_scores_linkage_average(vbx_scores* sc, int64_t T, double* Z);

/* Average-linkage clustering of n observations from their condensed distance vector [n (n - 1) / 2] (host code;
 * nearest-neighbour chain, see vbx_linkage.hpp for why it is not a kernel): vbhmm.py:140-141
 * `fastcluster.linkage(scr_mx, method='average')`.  Z: [n - 1][4] = (cluster a, cluster b, distance, members), the
 * linkage matrix of scipy.cluster.hierarchy / fastcluster, bit for bit SciPy's.  Pure function, safe to call from
 * several threads at once (one recording per thread); needs no vbx_ctx. */
int vbx_linkage_average(int64_t n, const double* condensed, double* Z);
/* The same clustering in the form fastcluster 1.2 itself computes it (the package vbhmm.py:33,140-141 imports; SciPy is
 * what this repository's fixtures were made with because fastcluster is not installed): weights divided before the update
 * (d = s a + t b, s = n_a / (n_a + n_b)) and fastcluster's own chain bookkeeping (vbx_linkage.hpp).  Same tree and
 * distances equal to rounding wherever no two candidate distances tie; selected by VBX_AMD_LINKAGE=fastcluster in the
 * Python layer.  Restated from the published source -- parity with the package itself is unpinned. */
int vbx_linkage_average_fastcluster(int64_t n, const double* condensed, double* Z);
/* Flat clusters of the linkage matrix Z cut at cophenetic distance t: vbhmm.py:145-146 `fcluster(lin_mat, t,
 * criterion='distance')`.  labels: [n], numbered from 1 in SciPy's order (depth-first from the root). */
int vbx_fcluster_distance(int64_t n, const double* Z, double t, int32_t* labels);

/* Index of a binary Kaldi vector archive held in memory (vbhmm.py:117 `kaldi_io.read_vec_flt_ark`): offsets and
 * lengths of the key and of the data of every entry ('FV ' float32 / 'DV ' float64 vectors).  Returns the number
 * of entries; -1: not a well-formed binary vector archive; -2: more than `cap` entries. */
int64_t vbx_ark_index(const void* buf, int64_t len, int64_t cap, int64_t* key_off, int32_t* key_len, int64_t* data_off,
                      int32_t* dim, int32_t* elem_size);
/* n rows of row_bytes bytes each, row i starting at buf + offsets[i], packed into out (the vectors of one recording
 * out of an indexed archive: `np.array(xvecs)` of vbhmm.py:123). */
int vbx_gather_rows(const void* buf, int64_t len, const int64_t* offsets, int64_t n, int64_t row_bytes, void* out);
/* Two-Gaussian shared-variance EM over all the scores: threshold, and (llr != NULL) the linearly calibrated
 * log-odds of every score, [vbx_scores_count]. */
int vbx_scores_two_gmm_calib(vbx_scores* sc, int32_t niters, double* threshold, double* llr);
int vbx_scores_destroy(vbx_scores* sc);

#ifdef __cplusplus
}
#endif
#endif /* VBX_HIP_H */
