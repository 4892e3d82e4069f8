// mall_probe.hip -- how fast does a grid that does NOT fill the chip stream a working set that fits the Infinity Cache?  n
// workgroups of 256 threads read 80 KB each (what a chunk of the VB loop reads: rho + b), all loads of a thread in flight at once,
// twenty launches over the SAME bytes (8 recordings: 51 MB; 64 recordings: 404 MB, beyond the 256 MB cache).  Prints GB/s.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mall_probe tools/mall_probe.hip && /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NL>
__global__ __launch_bounds__(256) void stream(const f4* __restrict__ src, float* out) {
    const f4* p = src + (long long)blockIdx.x * (256 * NL) + threadIdx.x;
    f4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = p[i * 256];
    float s = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    if (s == 1.2345f) out[blockIdx.x] = s;
}
int main() {
    const size_t bytes = (size_t)5056 * 81920;
    f4* buf; float* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 1 << 20);
    hipMemset(buf, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int n : {79, 158, 316, 632, 1264, 2528, 5056}) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(stream<20>, dim3(n), dim3(256), 0, 0, buf, out);
        hipEventRecord(a);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(stream<20>, dim3(n), dim3(256), 0, 0, buf, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double us = 1e3 * ms / 20, mb = n * 81920 / 1e6;
        printf("n %4d workgroups, %6.1f MB per launch (same bytes every launch): %.2f us per launch back to back, %.0f GB/s\n", n, mb, us, mb / us * 1e3 / 1e3 * 1e3);
    }
    return 0;
}
