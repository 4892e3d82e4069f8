// placement_probe.hip -- how does the dispatcher spread a grid that does NOT fill the chip?  N workgroups of 256 threads, each
// holding `lds` bytes of LDS and spinning ~`us` microseconds, record the CU they ran on (HW_REG_HW_ID / XCC_ID) and their start
// time; the host prints the histogram "workgroups per CU" and how many ran concurrently.  A small batch of the VB loop is such a
// grid (632 workgroups for 8 recordings on 256 CUs): if the hardware packs them onto a part of the chip, the kernels of a small
// batch pay for it.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/placement_probe tools/placement_probe.hip && /tmp/placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void probe(unsigned* hw, unsigned* xcc, long long* t0, long long* t1, int spin, int lds_words) {
    extern __shared__ int dyn[];
    if (threadIdx.x < (unsigned)lds_words && lds_words > 0) dyn[threadIdx.x] = threadIdx.x;
    long long a = wall_clock64();
    unsigned h, x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0000001f + 0.5f;
    long long b = wall_clock64();
    if (threadIdx.x == 0) {
        hw[blockIdx.x] = h;
        xcc[blockIdx.x] = x;
        t0[blockIdx.x] = a;
        t1[blockIdx.x] = b;
    }
    if (v == 12345.f) dyn[0] = 1;
}

int main() {
    const int counts[] = {79, 158, 256, 316, 632, 1264};
    const int ldss[] = {17 * 1024, 36 * 1024, 60 * 1024};
    unsigned *hw, *xcc;
    long long *t0, *t1;
    hipMalloc(&hw, 4096 * 4); hipMalloc(&xcc, 4096 * 4); hipMalloc(&t0, 4096 * 8); hipMalloc(&t1, 4096 * 8);
    for (int lds : ldss) {
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (int n : counts) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(n), dim3(256), lds, 0, hw, xcc, t0, t1, 20000, 64);
            hipDeviceSynchronize();
            std::vector<unsigned> h(n), x(n);
            std::vector<long long> a(n), b(n);
            hipMemcpy(h.data(), hw, n * 4, hipMemcpyDeviceToHost); hipMemcpy(x.data(), xcc, n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(a.data(), t0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), t1, n * 8, hipMemcpyDeviceToHost);
            std::map<unsigned, int> per_cu;
            for (int i = 0; i < n; ++i) {
                // HW_ID: [11:8] CU_ID, [12] SH_ID, [15:13] SE_ID (gfx9 layout); XCC_ID [3:0]
                const unsigned key = ((x[i] & 0xf) << 12) | (((h[i] >> 13) & 7) << 8) | (((h[i] >> 12) & 1) << 7) | ((h[i] >> 8) & 0xf);
                per_cu[key]++;
            }
            std::map<int, int> hist;
            for (auto& kv : per_cu) hist[kv.second]++;
            const long long start = *std::min_element(a.begin(), a.end()), end = *std::max_element(b.begin(), b.end());
            long long longest = 0, first_round = 0;
            for (int i = 0; i < n; ++i) { longest = std::max(longest, b[i] - a[i]); if (a[i] - start < (end - start) / 4) ++first_round; }
            printf("lds %2d KB  n %4d: CUs used %3zu  workgroups-per-CU histogram {", lds / 1024, n, per_cu.size());
            for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second);
            printf(" }  span %.1f us, longest workgroup %.1f us, started in the first quarter %lld\n", (end - start) / 100.0, longest / 100.0, first_round);
        }
    }
    return 0;
}
