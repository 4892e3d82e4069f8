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
#include "vbx_big.hpp"
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

// The host side in reading order (each part opens and closes its own extern "C" block):
#include "vbx_host_state.hpp"    // vbx_ctx, vbx_batch
#include "vbx_host_launch.hpp"   // launch helpers of an iteration, allocator
#include "vbx_host_batch.hpp"    // context API; a batch on one stream
#include "vbx_host_group.hpp"    // stream groups, the public vbx_batch_* entry points, vbx_run
#include "vbx_host_steps.hpp"    // step-level API of the parity tests
#include "vbx_host_ahc.hpp"      // AHC score stage, x-vector front end, linkage

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
