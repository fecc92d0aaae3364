// hbm_probe.hip — what this box's HBM delivers to a plain streaming kernel with the SAME read/write byte mix as
// camal::post_kernel: the honest denominator beside the 8 TB/s spec (bench.py: roofline_post.copy_gbs).
// NR read streams and NW write streams of n floats each, 16-byte accesses, grid-stride, nothing else.
// Built by __graft_entry__.build() into tools/libhbm_probe.so; measurement tool only, not part of the product.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int NR, int NW>
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) v[r] = src[(size_t)r * n4 + i];
        float4 s = v[0];
#pragma unroll
        for (int r = 1; r < NR; ++r) { s.x += v[r].x; s.y += v[r].y; s.z += v[r].z; s.w += v[r].w; }
#pragma unroll
        for (int w = 0; w < NW; ++w) dst[(size_t)w * n4 + i] = s;
    }
}

// The same streams with U independent 16-byte loads per read stream in flight per lane before the first use (the
// plain grid-stride kernel above keeps ONE per stream: its copy stops at 4.6-4.8 TB/s where the guide measures
// 6.29 TB/s for a float4 copy on this part).  A workgroup owns a contiguous span of U * 256 float4 per iteration.
template <int NR, int NW, int U>
__global__ __launch_bounds__(256) void stream_kernel_u(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t span = (size_t)U * 256;
    for (size_t base = (size_t)blockIdx.x * span; base < n4; base += (size_t)gridDim.x * span) {
        float4 v[U][NR];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
#pragma unroll
            for (int r = 0; r < NR; ++r) v[u][r] = i < n4 ? src[(size_t)r * n4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            float4 s = v[u][0];
#pragma unroll
            for (int r = 1; r < NR; ++r) { s.x += v[u][r].x; s.y += v[u][r].y; s.z += v[u][r].z; s.w += v[u][r].w; }
            if (i < n4) {
#pragma unroll
                for (int w = 0; w < NW; ++w) dst[(size_t)w * n4 + i] = s;
            }
        }
    }
}

// Average seconds per launch of a kernel that reads nr and writes nw streams of n floats (n % 4 == 0), over `reps`
// launches after 3 warm-up launches, timed with HIP events on `stream`.  src holds nr*n floats, dst nw*n.  < 0: error.
// unroll: 1 = the plain kernel; 2, 4, 8 = that many independent loads per stream in flight per lane
static double probe_impl(const float* src, float* dst, size_t n, int nr, int nw, int blocks, int reps, int unroll, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const size_t n4 = n / 4;
    auto launch = [&]() -> bool {
#define CASE(R, W) if (nr == R && nw == W) { \
        if (unroll <= 1) hipLaunchKernelGGL((stream_kernel<R, W>), dim3(blocks), dim3(256), 0, s, (const float4*)src, (float4*)dst, n4); \
        else if (unroll == 2) hipLaunchKernelGGL((stream_kernel_u<R, W, 2>), dim3(blocks), dim3(256), 0, s, (const float4*)src, (float4*)dst, n4); \
        else if (unroll == 4) hipLaunchKernelGGL((stream_kernel_u<R, W, 4>), dim3(blocks), dim3(256), 0, s, (const float4*)src, (float4*)dst, n4); \
        else hipLaunchKernelGGL((stream_kernel_u<R, W, 8>), dim3(blocks), dim3(256), 0, s, (const float4*)src, (float4*)dst, n4); \
        return true; }
        CASE(1, 1) CASE(2, 2) CASE(3, 2) CASE(4, 2) CASE(5, 2) CASE(6, 2) CASE(1, 2) CASE(2, 1) CASE(6, 1)
#undef CASE
        return false;
    };
    for (int i = 0; i < 3; ++i) if (!launch()) return -1.0;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -2.0;
    hipEventRecord(a, s);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, s);
    if (hipEventSynchronize(b) != hipSuccess) return -3.0;
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return (double)ms * 1e-3 / reps;
}

extern "C" double hbm_probe(const float* src, float* dst, size_t n, int nr, int nw, int blocks, int reps, void* stream) {
    return probe_impl(src, dst, n, nr, nw, blocks, reps, 1, stream);
}
extern "C" double hbm_probe_u(const float* src, float* dst, size_t n, int nr, int nw, int blocks, int reps, int unroll, void* stream) {
    return probe_impl(src, dst, n, nr, nw, blocks, reps, unroll, stream);
}
