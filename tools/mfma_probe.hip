// mfma_probe.hip — what this box's matrix pipes deliver to a loop that does nothing but MFMA on registers (no LDS, no
// memory): the measured ceiling beside the guide's 2.5 PF (bf16) / 157.3 TF (fp32) figures, which are quoted at the
// 2.4 GHz peak clock.  NACC independent accumulator tiles per wave, `waves` waves per workgroup, `blocks` workgroups.
// Built by __graft_entry__.build() into tools/libmfma_probe.so; measurement tool only, not part of the product.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// RND: every accumulator tile has its own A and B operands with random signs and mantissas (the operand buses and the
// multipliers toggle as they do on real activations); else all MFMAs read the same small integers.
__device__ __forceinline__ unsigned probe_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// a bf16 pair with random sign and mantissa bits, exponents 2^-2 .. 2^1 (finite products, sums stay small)
__device__ __forceinline__ unsigned probe_bf16_pair(unsigned h) {
    const unsigned lo = (h & 0x80ffu) | (((h >> 8) & 3u) + 125u) << 7, hi = ((h >> 16) & 0x80ffu) | (((h >> 24) & 3u) + 125u) << 7;
    return (lo & 0xffffu) | (hi << 16);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool RND, bool OWN = RND>
__global__ __launch_bounds__(512) void mfma_bf16_kernel(float* out, int iters, long long* cycles) {
    f32x4 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NOP = OWN ? NACC : 1;  // OWN without RND: a register pair per tile, all holding the same small integers
    bf16x8 a[NOP], b[NOP];
#pragma unroll
    for (int j = 0; j < NOP; ++j) {
        if (RND) {
            u32x4 ua, ub;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ua[k] = probe_bf16_pair(probe_hash((threadIdx.x * 64 + j * 8 + k) * 2 + 1));
                ub[k] = probe_bf16_pair(probe_hash((threadIdx.x * 64 + j * 8 + k) * 2 + 0x10001));
            }
            a[j] = __builtin_bit_cast(bf16x8, ua);
            b[j] = __builtin_bit_cast(bf16x8, ub);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) { a[j][k] = (__bf16)(float)((threadIdx.x + k) & 7); b[j][k] = (__bf16)(float)((threadIdx.x * 3 + k) & 3); }
            asm volatile("" : "+v"(a[j]), "+v"(b[j]));  // one register set per j even where the values are equal
        }
    }
    const long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j % NOP], b[j % NOP], acc[j], 0, 0, 0);
    }
    const long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 12345.678f) out[0] = s;  // keeps the accumulators alive
    if (cycles != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = c1 - c0;
}

template <int NACC>
__global__ __launch_bounds__(512) void mfma_f32_kernel(float* out, int iters, long long* cycles) {
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
    const float a = (float)(threadIdx.x & 7), b = (float)(threadIdx.x & 3);
    const long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    const long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][5];
    if (s == 12345.678f) out[0] = s;
    if (cycles != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = c1 - c0;
}

// TFLOP/s of `reps` launches (after 2 warm-up launches) of blocks x waves waves, each issuing iters x 8 (bf16) or x 4 (fp32) independent MFMAs.
// kind 0: v_mfma_f32_16x16x32_bf16 (16384 flop each), 1: v_mfma_f32_32x32x2_f32 (4096 flop each), 2: kind 0 on random
// operands, 3: kind 0 with a register pair per tile as in 2 but the constant values of 0.  < 0: error.
// seconds_out (optional): average seconds per launch.
extern "C" double mfma_probe(int kind, int blocks, int waves, int iters, int reps, float* scratch, double* seconds_out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (waves < 1 || waves > 8 || blocks < 1 || iters < 1 || reps < 1) return -1.0;
    auto launch = [&]() {
        if (kind == 0) hipLaunchKernelGGL((mfma_bf16_kernel<8, false>), dim3(blocks), dim3(waves * 64), 0, s, scratch, iters, (long long*)nullptr);
        else if (kind == 2) hipLaunchKernelGGL((mfma_bf16_kernel<8, true>), dim3(blocks), dim3(waves * 64), 0, s, scratch, iters, (long long*)nullptr);
        else if (kind == 3) hipLaunchKernelGGL((mfma_bf16_kernel<8, false, true>), dim3(blocks), dim3(waves * 64), 0, s, scratch, iters, (long long*)nullptr);
        else hipLaunchKernelGGL((mfma_f32_kernel<4>), dim3(blocks), dim3(waves * 64), 0, s, scratch, iters, (long long*)nullptr);
    };
    launch(); launch();
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -2.0;
    hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) return -3.0;
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    const double sec = (double)ms * 1e-3 / reps;
    if (seconds_out) *seconds_out = sec;
    const double flop = (double)blocks * waves * iters * (kind != 1 ? 8.0 * 16384.0 : 4.0 * 4096.0);
    return flop / sec / 1e12;
}
