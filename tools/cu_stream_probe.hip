// cu_stream_probe.hip -- what can ONE workgroup (one CU) pull from memory?  (the boundary walk of the
// wide chunked scan reads an S x S operator per chain step on one CU and its step time goes with those bytes, 42 GB/s.)
// One workgroup of 1024 (or 256) threads streams `bytes` with 16-byte loads, `UNROLL` loads in flight per thread:
//   fresh     a buffer of 256 MB read once, front to back (HBM; every page new to the CU)
//   region    a 4 MB / 64 KB region read over and over (the XCD's L2 / the CU's own L1; pages known)
//
//   hipcc --offload-arch=gfx950 -O3 -o cu_stream_probe tools/cu_stream_probe.hip && ./cu_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int UNROLL>
__global__ __launch_bounds__(1024) void stream(const float4* __restrict__ src, size_t region_vecs, size_t total_vecs, float* sink) {      // (region_vecs a power of two)
    const size_t mask = region_vecs - 1;
    float4 acc = {0, 0, 0, 0};
    const size_t step = (size_t)blockDim.x * UNROLL;
    for (size_t base = 0; base < total_vecs; base += step) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = src[(base + (size_t)u * blockDim.x + threadIdx.x) & mask];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == -1.f) *sink = 1.f;
}

template <int UNROLL> void run(const float4* d, float* sink, int threads, size_t region_bytes, size_t total_bytes, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(stream<UNROLL>, dim3(1), dim3(threads), 0, 0, d, region_bytes / 16, (size_t)(4 << 20) / 16, sink);     // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(stream<UNROLL>, dim3(1), dim3(threads), 0, 0, d, region_bytes / 16, total_bytes / 16, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %4d threads, %2d loads of 16 bytes in flight per thread: %7.1f GB/s\n", what, threads, UNROLL, total_bytes / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main() {
    const size_t big = (size_t)256 << 20;
    float4* d;
    float* sink;
    hipMalloc(&d, big);
    hipMalloc(&sink, 4);
    hipMemset(d, 0, big);
    for (int threads : {1024, 256}) {
        run<4>(d, sink, threads, big, big, "fresh 256 MB, front to back");
        run<16>(d, sink, threads, big, big, "fresh 256 MB, front to back");
        run<4>(d, sink, threads, (size_t)4 << 20, big / 2, "a 4 MB region over and over (L2)");
        run<16>(d, sink, threads, (size_t)4 << 20, big / 2, "a 4 MB region over and over (L2)");
        run<4>(d, sink, threads, (size_t)16 << 10, big / 2, "a 16 KB region over and over (the CU's L1)");
        run<16>(d, sink, threads, (size_t)16 << 10, big / 2, "a 16 KB region over and over (the CU's L1)");
    }
    return 0;
}
