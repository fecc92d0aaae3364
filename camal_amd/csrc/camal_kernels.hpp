// camal_kernels.hpp — gfx950 (CDNA4) device code for the CamAL inference path.
//
// Data layout in HBM:
//   activations  [window][time][channel]   (channels-last; 16-byte chunks of 4 fp32 / 8 bf16
//                                            channels are the unit every load/store moves)
//   conv weights  packed per layer in MFMA A-fragment order
//                 [m-tile of 32 out-channels][tap][chunk-pair][lane 0..63][16 bytes]
//   CAM rows      [member][window][time] fp32, pooled partial sums [member][window][tile][2]
//
// Implicit GEMM view of one ConvBlock (reference camal_resnet.py:81-86, BN folded):
//   D[co][t] = sum_{tap j} sum_{ci} W'[co][ci][j] * X[clamp(t + j - k/2)][ci]
//   M = out-channels (MFMA rows), N = time (MFMA columns, one column per lane),
//   K = (tap, in-channel).  The time tile plus its k-1 halo rows is staged once into LDS
//   (replicate padding = clamping the staged row's source position, per layer, exactly as
//   the reference pads each layer's own input), and every tap re-reads it at a row offset.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace camal {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename DT> struct Elem;
template <> struct Elem<float> { static constexpr int PER_CHUNK = 4; static constexpr int K_PER_PAIR = 8; };
template <> struct Elem<__bf16> { static constexpr int PER_CHUNK = 8; static constexpr int K_PER_PAIR = 16; };

// XOR swizzle of the 16-byte chunk index inside an LDS row so that the 16 lanes of one
// ds_read_b128 group (rows distinct mod 16, same logical chunk) hit 16 different slots of
// the 256-byte bank row.  Rows of 128 B hold two rows per bank row, hence the shift.
template <int CHUNKS_PER_ROW> __device__ __forceinline__ int swz(int row) {
    if (CHUNKS_PER_ROW >= 16) return row & 15;
    if (CHUNKS_PER_ROW == 4) return (row >> 2) & 3;  // 64-byte rows (32 bf16 channels): four rows per bank row
    return (row >> 1) & 7;
}

// One wave's accumulator block = 32 out-channels x 32 columns, as the MFMA shape of the data type lays it out:
//   fp32, v_mfma_f32_32x32x2_f32: one 32 x 32 tile.  Lane (r = l & 31, h = l >> 5); element 4g + i (g = 0..3)
//         is channel 8g + 4h + i of column r.
//   bf16, v_mfma_f32_16x16x32_bf16: four 16 x 16 tiles v[2 mi + nj] (channels 16 mi.., columns 16 nj..).  Lane
//         (c = l & 15, q = l >> 4); element i of tile g = 2 mi + nj is channel 16 mi + 4q + i of column 16 nj + c.
//         (The 16x16x32 shape: the chip holds a higher clock under it than under 32x32x16 at equal cycles per
//         FLOP, MI355X_MICROARCH.md "DVFS give-back" (7); measured here +6 % on the fused blocks.)
// Either way a lane holds, for every group g, 4 consecutive channels chan(g) + 0..3 of column col(g): the
// epilogues below are written against get/chan/col only.
template <typename DT> struct Frag;
template <> struct Frag<float> {
    static constexpr int NCOL = 1;  // distinct columns a lane holds
    f32x16 v;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = 0.f;
    }
    __device__ __forceinline__ float get(int g, int i) const { return v[4 * g + i]; }
    __device__ __forceinline__ void set(int g, int i, float x) { v[4 * g + i] = x; }
    static __device__ __forceinline__ int chan(int g, int lane) { return 8 * g + 4 * (lane >> 5); }
    static __device__ __forceinline__ int col(int g, int lane) { return lane & 31; }
    static __device__ __forceinline__ int colslot(int g) { return 0; }
    static __device__ __forceinline__ bool col_writer(int lane) { return lane < 32; }
    // sum over the lanes that hold the same column
    static __device__ __forceinline__ float col_reduce(float x) { return x + __shfl_xor(x, 32); }
};
template <> struct Frag<__bf16> {
    static constexpr int NCOL = 2;
    f32x4 v[4];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[g][i] = 0.f;
    }
    __device__ __forceinline__ float get(int g, int i) const { return v[g][i]; }
    __device__ __forceinline__ void set(int g, int i, float x) { v[g][i] = x; }
    static __device__ __forceinline__ int chan(int g, int lane) { return 16 * (g >> 1) + 4 * (lane >> 4); }
    // Which of the fragment's 16 columns lane i = l & 15 carries is free (the MFMA only pairs B column i with
    // D column i), and so is which 8-channel block of a 32-channel group k-block q = l >> 4 carries (A and B
    // agree on kblock).  A ds_read_b128 is served in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...:
    // lanes i in 4..11 of one k-block with lanes i in 0-3,12-15 of the next.  With columns 0,2,..,14 on lanes
    // 4..11, the odd ones on the others, and k-blocks (0,1,2,3) -> channel blocks (0,2,1,3), the 16 lanes of
    // every group land on 16 different 16-byte slots of the 256-byte bank row for every tap offset (XOR
    // swizzle of swz<>): conflict-free, where the identity mapping is 2-way conflicted on every odd tap.
    static __device__ __forceinline__ int bcol(int lane) {
        const int i = lane & 15;
        return (i >= 4 && i < 12) ? 2 * (i - 4) : (i < 4 ? 2 * i + 1 : 2 * i - 15);
    }
    static __device__ __forceinline__ int kblock(int lane) { const int q = lane >> 4; return ((q & 1) << 1) | (q >> 1); }
    static __device__ __forceinline__ int col(int g, int lane) { return 16 * (g & 1) + bcol(lane); }
    static __device__ __forceinline__ int colslot(int g) { return g & 1; }
    static __device__ __forceinline__ bool col_writer(int lane) { return lane < 16; }
    static __device__ __forceinline__ float col_reduce(float x) {
        x += __shfl_xor(x, 16);
        return x + __shfl_xor(x, 32);
    }
};

__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
// two floats -> one dword of two bf16 (round to nearest even) in ONE v_cvt_pk_bf16_f32: hipcc turns a 4-vector
// conversion into four single-value converts and two v_perm_b32 (3x the instructions; the bf16 epilogues are
// VALU-issue bound), a 2-vector conversion into the packed instruction
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void store4(__bf16* p, float a, float b, float c, float d) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = {pk_bf16(a, b), pk_bf16(c, d)};
    *reinterpret_cast<u32x2*>(p) = v;
}
// relu of 4 values, stored in the activation type.  bf16: round first, then clamp the packed pairs as signed
// 16-bit integers (a negative bf16 is a negative int16; -0 and negative NaNs become +0 like everything below 0)
__device__ __forceinline__ void store4_relu(float* p, float a, float b, float c, float d) {
    store4(p, fmaxf(a, 0.f), fmaxf(b, 0.f), fmaxf(c, 0.f), fmaxf(d, 0.f));
}
__device__ __forceinline__ void store4_relu(__bf16* p, float a, float b, float c, float d) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = {pk_bf16(a, b), pk_bf16(c, d)};
    s16x4 w = __builtin_bit_cast(s16x4, v);
    const s16x4 z = {0, 0, 0, 0};
    w = __builtin_elementwise_max(w, z);
    *reinterpret_cast<s16x4*>(p) = w;
}

__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const __bf16* p, float (&v)[4]) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}

// ---------------------------------------------------------------------------------------
// Stem: the Cin = 1 layers of block 1 on the VALU (K = k <= 63 is not MFMA-shaped).
//   a1[t][co] = relu(b'[co] + sum_j w'[co][j] * x[clamp(t + j - k/2)])   (camal_resnet.py:14, ConvBlock 0)
//   r1[t][co] = wr'[co] * x[t] + br'[co]                                  (residual, :51-54)
// One workgroup = 64 timesteps x C channels (C = mid_channels: 64, or 32 / 128); thread = 4 channels x (C / 16)
// timesteps.
// ---------------------------------------------------------------------------------------
template <typename DT, int C = 64>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x,      // [B][L]
                                                   const float* __restrict__ w,      // [ks][C]
                                                   const float* __restrict__ bias,   // [C]
                                                   const float* __restrict__ wr,     // [C]
                                                   const float* __restrict__ br,     // [C]
                                                   DT* __restrict__ a1, DT* __restrict__ r1,  // [B][L][C]
                                                   int L, int ks, bool zero_pad = false) {
    constexpr int CG = C / 4, TQ = 256 / CG, NQ = 64 / TQ;  // channel groups, time slots, timesteps per thread
    static_assert(CG * TQ == 256 && NQ * TQ == 64 && NQ >= 1, "C must be 16 .. 256, a power of two");
    __shared__ float xs[64 + 64];
    const int b = blockIdx.y, t0 = blockIdx.x * 64, pad = ks >> 1;
    const float* xb = x + (size_t)b * L;
    for (int i = threadIdx.x; i < 64 + ks - 1; i += 256) {
        const int p = t0 - pad + i;
        const int pc = p < 0 ? 0 : (p > L - 1 ? L - 1 : p);
        // zero_pad: padding="same" of classifiers/resnet.py:62 (zeros) instead of the replicate mode
        xs[i] = (zero_pad && pc != p) ? 0.f : xb[pc];
    }
    __syncthreads();
    const int cg = threadIdx.x % CG, tq = threadIdx.x / CG;
    float acc[NQ][4];
    float bv[4], wrv[4], brv[4];
    load4(bias + 4 * cg, bv); load4(wr + 4 * cg, wrv); load4(br + 4 * cg, brv);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[q][c] = bv[c];
    for (int j = 0; j < ks; ++j) {
        float wv[4];
        load4(w + j * C + 4 * cg, wv);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float xv = xs[tq + TQ * q + j];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[q][c] = fmaf(wv[c], xv, acc[q][c]);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        int t = t0 + tq + TQ * q;
        if (t < L) {
            size_t o = ((size_t)b * L + t) * C + 4 * cg;
            store4(a1 + o, fmaxf(acc[q][0], 0.f), fmaxf(acc[q][1], 0.f), fmaxf(acc[q][2], 0.f), fmaxf(acc[q][3], 0.f));
            float xv = xs[tq + TQ * q + pad];
            store4(r1 + o, fmaf(wrv[0], xv, brv[0]), fmaf(wrv[1], xv, brv[1]), fmaf(wrv[2], xv, brv[2]),
                   fmaf(wrv[3], xv, brv[3]));
        }
    }
}

// ---------------------------------------------------------------------------------------
// Implicit-GEMM Conv1d(replicate) + folded BN + [ReLU] + [residual add] on MFMA.
//
// Persistent workgroups of 5 waves.  Waves 0-3 compute: wave w owns out-channel tile
// (w % MT) and NT column tiles of 32 timesteps.  Wave 4 is a LOADER: while the others run
// the MFMA loop on tile i from one LDS buffer it streams tile i+1 (+ its k-1 halo rows)
// into the other with global_load_lds (LDS-DMA, no VGPRs), so HBM latency never sits in
// front of the matrix pipe and the loader's vmcnt does not serialise the compute waves'
// weight loads.  One workgroup barrier per tile hands the buffers over.
//   HEAD = false: writes out[b][t][co]   (+ res[b][t][co] added AFTER the ReLU, :56-59)
//   HEAD = true : last ConvBlock; the 128 x L feature map is never written.  Emits the two
//                 CAM rows' values W_fc[k] . relu(.) per timestep: cam[b][t] for k = cam_row
//                 (class_activation_map.py:42-48) and per-tile sums over t of both rows
//                 (mean_t(W_fc[k].F) + b_k == Linear(mean_t F), camal_resnet.py:23).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    // LDS destination = wave-uniform base + lane * 16; the global source is per lane.
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Weight loads of the K loop, hidden from hipcc (guide §5.7 item 1, form (ii)): wave-uniform base in
// scalar registers + the lane's 32-bit byte offset, and the waits counted by hand.  hipcc's own
// bookkeeping put a vmcnt(0) at the head of the tap loop (the ring's prologue order and a spilled 64-bit
// lane pointer reloaded there), which drained the ring once per tap, and kept one 64-bit lane address
// per layer alive (spills under the 168-register cap).  A destination is valid only after w_wait.
// The base goes through an s_mov_b64 inside the asm: a VMEM instruction that reads an SGPR which a VALU
// instruction (v_readlane of a spilled SGPR, v_readfirstlane) wrote fewer than 5 wait states earlier reads
// the OLD value, and hipcc's hazard pass does not look at the operands of inline asm (seen: a v_readlane
// pair right in front of the load, a wild address, a GPU memory fault).  SALU reads of such an SGPR are
// interlocked, and a SALU-written SGPR is safe for VMEM at once.
__device__ __forceinline__ void w_load(u32x4& dst, const void* sbase, unsigned lane_off) {
    unsigned long long tmp;
    asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx4 %0, %2, %1"
                 : "=v"(dst), "=&s"(tmp)
                 : "v"(lane_off), "s"(sbase)
                 : "memory");
}
template <int N> __device__ __forceinline__ void w_wait(u32x4& r) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N) : "memory");
}

// K loop of one conv layer for one wave, fp32: acc[nt] += sum over taps and chunk pairs of
// A (weights, 32 out-channels, streamed from L2 in packed order through a register ring that keeps
// AR-1 loads in flight) x B (activations read from the LDS tile at row R0 + tap + 32 nt, one chunk
// pair ahead).  Software-pipelined by hand; ring slots are static after unrolling.
// wp is WAVE-UNIFORM (the lane's 16 bytes are added inside).  No other vector-memory operation may be
// issued by the caller's wave between entry and exit: the ring's waits count from the youngest load.
template <int CIN, int NT, int MTW>
__device__ __forceinline__ void mma_layer_f32(const char* buf, int Rb, int ks, const uint4* wp, Frag<float> (&acc)[MTW][NT]) {
    // MTW = out-channel tiles per wave (consecutive in the packed weights: stride ks*NCP*64 uint4)
    constexpr int RB = CIN * 4, CPR = RB / 16, NCP = CIN / 8;
#ifndef CAMAL_AR_F32
#define CAMAL_AR_F32 2
#endif
    constexpr int AR = CAMAL_AR_F32 < NCP ? CAMAL_AR_F32 : NCP;
    static_assert(NCP % AR == 0 && NCP % 2 == 0, "ring sizes must divide the chunk pairs per tap");
    const unsigned loff = (threadIdx.x & 63) * 16;
    const int h = (threadIdx.x & 63) >> 5;
    const int R0 = Rb + (threadIdx.x & 31);  // this lane's first row: one column per lane
    const int total = ks * NCP;
    const size_t mstride = (size_t)total * 64;
    u32x4 a[AR][MTW];
    uint4 bq[2][NT];
#pragma unroll
    for (int d = 0; d < AR - 1; ++d)
#pragma unroll
        for (int m = 0; m < MTW; ++m) w_load(a[d][m], wp + m * mstride + (size_t)(d < total ? d : total - 1) * 64, loff);
    {
        const char* rowp = buf + R0 * RB;
        const int col = (h ^ swz<CPR>(R0)) << 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[0][nt] = *reinterpret_cast<const uint4*>(rowp + nt * 32 * RB + col);
    }
#pragma unroll 1
    for (int j = 0; j < ks; ++j) {
        const int row = R0 + j, rown = R0 + (j + 1 < ks ? j + 1 : j);
        const int s = swz<CPR>(row), sn = swz<CPR>(rown);
        const char* rowp = buf + row * RB;
        const char* rowpn = buf + rown * RB;
#pragma unroll
        for (int cp = 0; cp < NCP; ++cp) {
            int idx = j * NCP + cp + (AR - 1);
            idx = idx < total ? idx : total - 1;  // past the end: a repeated load keeps the wait counts uniform
#pragma unroll
            for (int m = 0; m < MTW; ++m) w_load(a[(cp + AR - 1) % AR][m], wp + m * mstride + (size_t)idx * 64, loff);
            const char* rp = cp + 1 < NCP ? rowp : rowpn;
            const int col = cp + 1 < NCP ? ((2 * (cp + 1) + h) ^ s) << 4 : (h ^ sn) << 4;
            // issue order pinned by hand: each LDS read of the NEXT step's fragment rides in the shadow of
            // the MFMAs of this step, instead of a separate load phase during which the matrix pipe idles.
            // hipcc left alone sinks every load to just before its use; sched_group_barrier made it drop the
            // register double buffer.  4 k-steps per chunk; consecutive MFMAs alternate accumulators (k-step
            // major) so that no MFMA waits on the result of the one issued just before it
#pragma unroll
            for (int ks4 = 0; ks4 < 4; ++ks4) {
                if (ks4 < NT)
                    bq[(cp + 1) & 1][ks4] = *reinterpret_cast<const uint4*>(rp + ks4 * 32 * RB + col);
                if (NT > 4 && ks4 == 3)
#pragma unroll
                    for (int nt = 4; nt < NT; ++nt)
                        bq[(cp + 1) & 1][nt] = *reinterpret_cast<const uint4*>(rp + nt * 32 * RB + col);
                __builtin_amdgcn_sched_barrier(0);
                if (ks4 == 0)  // this step's weights: all but the AR-1 younger steps' loads have landed
#pragma unroll
                    for (int m = 0; m < MTW; ++m) w_wait<(AR - 1) * MTW>(a[cp % AR][m]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float* bf = reinterpret_cast<const float*>(&bq[cp & 1][nt]);
#pragma unroll
                    for (int m = 0; m < MTW; ++m) {
                        const float* af = reinterpret_cast<const float*>(&a[cp % AR][m]);
                        acc[m][nt].v = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks4], bf[ks4], acc[m][nt].v, 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // the repeated tail loads are still in flight: nothing may reuse their registers before they land
#pragma unroll
    for (int d = 0; d < AR; ++d)
#pragma unroll
        for (int m = 0; m < MTW; ++m) w_wait<0>(a[d][m]);
}

// The same for bf16 on v_mfma_f32_16x16x32_bf16.  K is walked in groups of 32 input channels of one tap
// (NG = CIN/32 groups per tap); a group takes two A fragments (16 out-channels x 32 k: lane (i = l & 15,
// q = l >> 4) carries channels 8 kblock(q)..+7 of the group; packed [tap][group][mi][lane][16 B], 2 KiB per
// group, consecutive groups consecutive in memory) through a ring of 4 groups (3 in flight), and 2 NT B
// fragments (16 columns x 32 k: lane (i, q) reads the 16-byte chunk 4 group + kblock(q) of row
// Rb + bcol(i) + tap + 16 f) which rotate through ONE register set: the read for the next group's fragment f
// is issued right behind this group's MFMAs on fragment f, 14 MFMAs before its use.
// Two waves share a SIMD and a 16x16x32 MFMA leaves only 8 issue cycles per MFMA for everything else, so
// the loop is kept lean: the weight pointer is a scalar that advances by 2 KiB per group (the second
// fragment through the instruction's offset; no clamp at the end: the ring runs 3 groups past the layer,
// into the next out-channel tile or the padding make_layer adds), one hand-counted wait per group, and the
// B address of a tap's later groups is the first one's XOR (group << 6) (swz<> only touches those bits).
typedef __attribute__((address_space(3))) const u32x4* lds_u4p;
__device__ __forceinline__ u32x4 lds_read16(unsigned addr) { return *(lds_u4p)(uintptr_t)addr; }

// both fragments of one group; the base goes through s_mov_b64 (see w_load)
__device__ __forceinline__ void w_load2(u32x4& d0, u32x4& d1, const void* sbase, unsigned lane_off) {
    unsigned long long tmp;
    asm volatile("s_mov_b64 %2, %4\n\tglobal_load_dwordx4 %0, %3, %2\n\tglobal_load_dwordx4 %1, %3, %2 offset:1024"
                 : "=v"(d0), "=v"(d1), "=&s"(tmp)
                 : "v"(lane_off), "s"(sbase)
                 : "memory");
}
template <int N> __device__ __forceinline__ void w_wait2(u32x4& r0, u32x4& r1) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(N) : "memory");
}

// group in ring slot S (mod 4): issue the loads of the group 3 ahead, wait for this group's fragments, 2 MFMAs per B
// fragment, each followed by the read of the next group's fragment (at LDS address bnext) into the same registers.
// LAST: the layer's last group.  Nothing is left to read; instead the epilogue of fragment f - 1 (whose two
// accumulator tiles are final) is issued behind the MFMAs on fragment f: an epilogue is VALU and LDS-store work that
// the matrix pipe would otherwise sit out (1.85 k cycles per layer and tile after a 512-cycle group).
struct NoEpi {
    static constexpr bool active = false;
    template <int NT> __device__ __forceinline__ void operator()(int, Frag<__bf16> (&)[NT]) const {}
};
template <int RB, int NT, int S, bool LAST = false, typename Epi = NoEpi>
__device__ __forceinline__ void bf16_group(const char*& wptr, unsigned loff, unsigned bnext, u32x4 (&a)[4][2],
                                           u32x4 (&b)[2 * NT], Frag<__bf16> (&acc)[NT], const Epi& epi = Epi()) {
    constexpr int NF = 2 * NT, SL = S % 4;
    w_load2(a[(SL + 3) % 4][0], a[(SL + 3) % 4][1], wptr, loff);
    wptr += 2048;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        __builtin_amdgcn_sched_barrier(0);
        if (f == 0) w_wait2<6>(a[SL][0], a[SL][1]);  // all but the 3 younger groups' loads have landed
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
            acc[f >> 1].v[2 * mi + (f & 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                *reinterpret_cast<const bf16x8*>(&a[SL][mi]), *reinterpret_cast<const bf16x8*>(&b[f]),
                acc[f >> 1].v[2 * mi + (f & 1)], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LAST) b[f] = lds_read16(bnext + f * 16 * RB);
        else if (f > 0) epi(f - 1, acc);
    }
    if constexpr (LAST) {
        __builtin_amdgcn_sched_barrier(0);
        epi(NF - 1, acc);
    }
}

// LDS address of the lane's chunk of group 0 in row `row` of the tile at LDS address `base`
template <int RB>
__device__ __forceinline__ unsigned b_addr(unsigned base, int row, int q) {
    return base + row * RB + ((q ^ swz<RB / 16>(row)) << 4);
}

// the NG groups of one tap, first ring slot S0: cur / nxt = b_addr of this tap / of the tap that follows
// (LAST: the layer's last tap, see bf16_group)
template <int CIN, int NT, int S0, bool LAST = false, typename Epi = NoEpi>
__device__ __forceinline__ void bf16_tap(const char*& wptr, unsigned loff, unsigned cur, unsigned nxt, u32x4 (&a)[4][2],
                                         u32x4 (&b)[2 * NT], Frag<__bf16> (&acc)[NT], const Epi& epi = Epi()) {
    constexpr int RB = CIN * 2, NG = CIN / 32;
    bf16_group<RB, NT, S0>(wptr, loff, cur ^ (1u << 6), a, b, acc);
    if constexpr (NG == 2) {
        bf16_group<RB, NT, S0 + 1, LAST, Epi>(wptr, loff, nxt, a, b, acc, epi);
    } else {
        bf16_group<RB, NT, S0 + 1>(wptr, loff, cur ^ (2u << 6), a, b, acc);
        bf16_group<RB, NT, S0 + 2>(wptr, loff, cur ^ (3u << 6), a, b, acc);
        bf16_group<RB, NT, S0 + 3, LAST, Epi>(wptr, loff, nxt, a, b, acc, epi);
    }
}

template <int CIN, int NT, int MTW>
__device__ __forceinline__ void mma_layer_bf16(const char* buf, int Rb, int ks, const uint4* wp, Frag<__bf16> (&accm)[MTW][NT]) {
    static_assert(MTW == 1, "one out-channel tile per wave");
    Frag<__bf16> (&acc)[NT] = accm[0];
    constexpr int RB = CIN * 2, NG = CIN / 32;
    static_assert(NG == 1 || NG == 2 || NG == 4 || NG == 8, "32, 64, 128 or 256 input channels");
    constexpr int NF = 2 * NT;
    const int lane = threadIdx.x & 63;
    const unsigned loff = lane * 16;
    const int q = Frag<__bf16>::kblock(lane);
    const int R0 = Rb + Frag<__bf16>::bcol(lane);
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)buf;
    const char* wptr = reinterpret_cast<const char*>(wp);
    u32x4 a[4][2];
    u32x4 b[NF];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        w_load2(a[d][0], a[d][1], wptr, loff);
        wptr += 2048;
    }
    unsigned cur = b_addr<RB>(base, R0, q);
#pragma unroll
    for (int f = 0; f < NF; ++f) {  // issue order pinned: see mma_stream_bf16
        b[f] = lds_read16(cur + f * 16 * RB);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NG == 4) {
#pragma unroll 1
        for (int j = 0; j < ks; ++j) {
            const unsigned nxt = j + 1 < ks ? b_addr<RB>(base, R0 + j + 1, q) : cur;  // past the end: any valid address
            bf16_tap<CIN, NT, 0>(wptr, loff, cur, nxt, a, b, acc);
            cur = nxt;
        }
    } else if constexpr (NG == 8) {
        // 256 input channels (mid_channels = 128): eight groups per tap, two turns of the ring; a tap's later groups sit at
        // the first one's address XOR (group << 6) (512-byte rows: swz<> only touches the bits below)
#pragma unroll 1
        for (int j = 0; j < ks; ++j) {
            const unsigned nxt = j + 1 < ks ? b_addr<RB>(base, R0 + j + 1, q) : cur;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g == 0) bf16_group<RB, NT, 0>(wptr, loff, cur ^ (1u << 6), a, b, acc);
                else if (g == 1) bf16_group<RB, NT, 1>(wptr, loff, cur ^ (2u << 6), a, b, acc);
                else if (g == 2) bf16_group<RB, NT, 2>(wptr, loff, cur ^ (3u << 6), a, b, acc);
                else if (g == 3) bf16_group<RB, NT, 3>(wptr, loff, cur ^ (4u << 6), a, b, acc);
                else if (g == 4) bf16_group<RB, NT, 0>(wptr, loff, cur ^ (5u << 6), a, b, acc);
                else if (g == 5) bf16_group<RB, NT, 1>(wptr, loff, cur ^ (6u << 6), a, b, acc);
                else if (g == 6) bf16_group<RB, NT, 2>(wptr, loff, cur ^ (7u << 6), a, b, acc);
                else bf16_group<RB, NT, 3>(wptr, loff, nxt, a, b, acc);
            }
            cur = nxt;
        }
    } else if constexpr (NG == 1) {
        // 32 input channels (mid_channels = 32): one group per tap, four taps per turn of the ring, then the 1 or 3 taps
        // an odd kernel size leaves (2 only if the caller ever passes an even one)
        int j = 0;
#pragma unroll 1
        for (; j + 4 <= ks; j += 4) {
            const unsigned a1 = b_addr<RB>(base, R0 + j + 1, q), a2 = b_addr<RB>(base, R0 + j + 2, q);
            const unsigned a3 = b_addr<RB>(base, R0 + j + 3, q);
            const unsigned a4 = j + 4 < ks ? b_addr<RB>(base, R0 + j + 4, q) : a3;
            bf16_group<RB, NT, 0>(wptr, loff, a1, a, b, acc);
            bf16_group<RB, NT, 1>(wptr, loff, a2, a, b, acc);
            bf16_group<RB, NT, 2>(wptr, loff, a3, a, b, acc);
            bf16_group<RB, NT, 3>(wptr, loff, a4, a, b, acc);
            cur = a4;
        }
        const int rem = ks - j;  // wave-uniform
        if (rem >= 1) {
            const unsigned a1 = rem > 1 ? b_addr<RB>(base, R0 + j + 1, q) : cur;
            bf16_group<RB, NT, 0>(wptr, loff, a1, a, b, acc);
            if (rem >= 2) {
                const unsigned a2 = rem > 2 ? b_addr<RB>(base, R0 + j + 2, q) : a1;
                bf16_group<RB, NT, 1>(wptr, loff, a2, a, b, acc);
                if (rem >= 3) bf16_group<RB, NT, 2>(wptr, loff, a2, a, b, acc);
            }
        }
    } else {
        int j = 0;
#pragma unroll 1
        for (; j + 2 <= ks; j += 2) {  // two taps = 4 groups per body: static ring slots
            const unsigned mid = b_addr<RB>(base, R0 + j + 1, q);
            const unsigned nxt = j + 2 < ks ? b_addr<RB>(base, R0 + j + 2, q) : mid;
            bf16_tap<CIN, NT, 0>(wptr, loff, cur, mid, a, b, acc);
            bf16_tap<CIN, NT, 2>(wptr, loff, mid, nxt, a, b, acc);
            cur = nxt;
        }
        if (j < ks) bf16_tap<CIN, NT, 0>(wptr, loff, cur, cur, a, b, acc);  // the odd last tap
    }
    // the ring's last loads (past the layer's end) are still in flight: nothing may reuse their registers before they land
#pragma unroll
    for (int d = 0; d < 4; ++d) w_wait2<0>(a[d][0], a[d][1]);
}

// The weight ring of the fused blocks never stops: the host lays the A fragments of ALL the layers of a block out
// as one stream per out-channel tile, [layer 1 | layer 2 | layer 3 | 1x1 skip | first 3 groups of layer 1 again]
// (camal_hip.hip: make_block_stream), so the 3 groups the ring holds in flight at the end of a layer ARE the next
// layer's first groups (after the last layer: the next tile's) and no layer starts by waiting an L2 round trip.
// The ring slot of a layer's first group is then static per layer (PH): 0 where every layer has 4 groups per tap;
// with 64 input channels (2 groups per tap, odd first kernel) layer 1 ends on slot 2.
struct WRing {
    u32x4 a[4][2];
    const char* wptr;  // next group to load (wave-uniform)
};
__device__ __forceinline__ void ring_begin(WRing& r, const char* base, unsigned loff) {
    r.wptr = base;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        w_load2(r.a[d][0], r.a[d][1], r.wptr, loff);
        r.wptr += 2048;
    }
}
__device__ __forceinline__ void ring_drain(WRing& r) {
#pragma unroll
    for (int d = 0; d < 4; ++d) w_wait2<0>(r.a[d][0], r.a[d][1]);
}

// One layer off the running ring: the products add to acc.  Epi (optional): the layer's epilogue, called once per B
// fragment f = 0 .. 2 NT - 1 from inside the last group (bf16_group LAST) when the two accumulator tiles
// acc[f >> 1].v[(f & 1)], .v[2 + (f & 1)] are final; with 64 input channels the kernel size must then be odd.
template <int CIN, int NT, int PH, typename Epi = NoEpi>
__device__ __forceinline__ void mma_stream_bf16(const char* buf, int Rb, int ks, WRing& r, Frag<__bf16> (&acc)[NT],
                                                const Epi& epi = Epi()) {
    constexpr int RB = CIN * 2, NG = CIN / 32;
    static_assert(NG == 2 || NG == 4, "64 or 128 input channels");
    constexpr int NF = 2 * NT;
    constexpr bool EPI = Epi::active;
    const int lane = threadIdx.x & 63;
    const unsigned loff = lane * 16;
    const int q = Frag<__bf16>::kblock(lane);
    const int R0 = Rb + Frag<__bf16>::bcol(lane);
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)buf;
    u32x4 b[NF];
    unsigned cur = b_addr<RB>(base, R0, q);
    // in fragment order, pinned: the waits hipcc puts in the loop come from the issue order it merges at the loop
    // header, and with these reads left to its scheduler fragment 0 was issued second to last: lgkmcnt(1), a full
    // LDS round trip, at the top of every tap
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        b[f] = lds_read16(cur + f * 16 * RB);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NG == 4) {
        const int body = EPI ? ks - 1 : ks;
#pragma unroll 1
        for (int j = 0; j < body; ++j) {
            const unsigned nxt = j + 1 < ks ? b_addr<RB>(base, R0 + j + 1, q) : cur;  // past the end: any valid address
            bf16_tap<CIN, NT, PH>(r.wptr, loff, cur, nxt, r.a, b, acc);
            cur = nxt;
        }
        if constexpr (EPI) bf16_tap<CIN, NT, PH, true, Epi>(r.wptr, loff, cur, cur, r.a, b, acc, epi);
    } else {
        int j = 0;
#pragma unroll 1
        for (; j + 2 <= ks; j += 2) {  // two taps = 4 groups per body: static ring slots
            const unsigned mid = b_addr<RB>(base, R0 + j + 1, q);
            const unsigned nxt = j + 2 < ks ? b_addr<RB>(base, R0 + j + 2, q) : mid;
            bf16_tap<CIN, NT, PH>(r.wptr, loff, cur, mid, r.a, b, acc);
            bf16_tap<CIN, NT, PH + 2>(r.wptr, loff, mid, nxt, r.a, b, acc);
            cur = nxt;
        }
        // the odd last tap: the ring ends 2 slots on
        if (EPI || j < ks) bf16_tap<CIN, NT, PH, EPI, Epi>(r.wptr, loff, cur, cur, r.a, b, acc, epi);
    }
}

// Epi of a layer whose output goes on to the next layer through LDS: relu(acc) of fragment f -> CH-channel LDS rows,
// what store_act_lds does for the whole tile (same addresses: one per (column half nj, channel half mi), the column
// tiles are constant offsets from them).
template <int NT, int CH = 128>
struct ActStoreEpi {
    static constexpr bool active = true;
    char* pg[2][2];
    __device__ __forceinline__ ActStoreEpi(char* dst, int row0, int mt, int lane) {
        using F = Frag<__bf16>;
        constexpr int RB = CH * 2, CPR = RB / 16;
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int g = 2 * mi + nj;
                const int row = row0 + F::col(g, lane);
                const int boff = (32 * mt + F::chan(g, lane)) * 2;
                pg[nj][mi] = dst + row * RB + (((boff >> 4) ^ swz<CPR>(row)) << 4) + (boff & 15);
            }
    }
    __device__ __forceinline__ void operator()(int f, Frag<__bf16> (&acc)[NT]) const {
        constexpr int RB = CH * 2;
        const int nt = f >> 1, nj = f & 1;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const f32x4 v = acc[nt].v[2 * mi + nj];
            store4_relu(reinterpret_cast<__bf16*>(pg[nj][mi] + nt * 32 * RB), v[0], v[1], v[2], v[3]);
        }
    }
};

// Rb: wave-uniform first row of the wave's columns in the LDS tile (the lane's column is added inside)
template <typename DT, int CIN, int NT, int MTW = 1>
__device__ __forceinline__ void mma_layer(const char* buf, int Rb, int ks, const uint4* wp, Frag<DT> (&acc)[MTW][NT]) {
    if constexpr (sizeof(DT) == 4) mma_layer_f32<CIN, NT, MTW>(buf, Rb, ks, wp, acc);
    else mma_layer_bf16<CIN, NT, MTW>(buf, Rb, ks, wp, acc);
}

// A value the compiler must take as new at this point: what is computed from it cannot be hoisted out of
// the tile loop.  Used on the lane index in front of the epilogues: hipcc otherwise computes every epilogue
// address once per kernel, keeps them live across the K loops and spills them (60 registers in the fused
// blocks under their 168-register cap).
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// core.py:308-310 on one CAM value: clip below at 0, NaN and +inf -> 0
__device__ __forceinline__ float clip_cam(float v) {
    // np.clip(a_min=0) then nan_to_num(nan=0, posinf=0): only positive finite values survive, and those
    // are exactly the bit patterns below +inf's when read as unsigned (sign bit set = negative or -0)
    return __float_as_uint(v) < 0x7f800000u ? v : 0.f;
}

#ifdef CAMAL_STAMPS
// Diagnostic builds only: s_memtime stamps of one tile per workgroup and wave (tools/stamps*.py)
#define CAMAL_NSTAMP 18
__device__ unsigned long long g_stamps[256][9][CAMAL_NSTAMP];
// CAMAL_STAMPS == 3: the fp32 per-layer kernel of the 128 -> 128, k = 3 ConvBlock with the skip add (tools/stamps_conv.py)
#define CSTAMP(i) do { if (CAMAL_STAMPS == 3 && sizeof(DT) == 4 && CIN == 128 && HAS_RES && it == 2 && blockIdx.x < 256 && (threadIdx.x & 63) == 0) \
    g_stamps[blockIdx.x][threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CSTAMP(i) do {} while (0)
#endif

// Tile order of conv_kernel's persistent loop.  A window's last tile is short when L is just past a multiple of the tile
// width (the data gradient's L + 2 pad = 516 rows: eight 64-row tiles + 4 rows); in plain (window, tile) order those
// cheap tiles are spread through the rounds and the last, partly filled round still consists of full tiles.  With a short
// tail the full tiles of all windows come first and the short ones last: 1152 tiles on 512 workgroups take 2 + a short
// round instead of 3.  Wave-uniform arithmetic, the same in the loader and the compute waves.
__device__ __forceinline__ void conv_tile_coords(int tile, int tpw, int total_tiles, int L, int TN, int& b, int& tx) {
    const int rem = L - (tpw - 1) * TN;  // rows of a window's last tile
    if (tpw > 1 && 2 * rem <= TN) {
        const int nb = total_tiles / tpw, full = nb * (tpw - 1);
        if (tile < full) { b = tile / (tpw - 1); tx = tile - b * (tpw - 1); }
        else { b = tile - full; tx = tpw - 1; }
    } else {
        b = tile / tpw;
        tx = tile - b * tpw;
    }
}

template <typename DT, int CIN, int COUT, int NT, int NWC>
struct ConvGeom {
    static constexpr int ES = sizeof(DT);
    static constexpr int RB = CIN * ES;                        // LDS row bytes
    static constexpr int CPR = RB / 16;                        // 16-byte chunks per row
    static constexpr int NCP = CIN / Elem<DT>::K_PER_PAIR;     // chunk pairs per tap
    static constexpr int MT = COUT / 32;                       // out-channel tiles
    static constexpr int NG = NWC / MT;                        // column groups across compute waves
    static constexpr int THREADS = (NWC + 1) * 64;             // + the loader wave
    static constexpr int TN = NG * NT * 32;                    // timesteps per tile
    static constexpr int HEAD_BYTES = (MT * TN * 2 + 3 * NWC) * 4;  // CAM-head reduction scratch
    __host__ __device__ static constexpr int buf_bytes(int ks) { return (((TN + ks - 1) * RB) + 1023) & ~1023; }
};

template <typename DT, int CIN, int COUT, int NT, int NWC, bool RELU, bool HAS_RES, bool HEAD>
__global__ __launch_bounds__((NWC + 1) * 64) void conv_kernel(const DT* __restrict__ in,     // [B][L][CIN]
                                                            const uint4* __restrict__ wpack,
                                                            const float* __restrict__ bias,  // [COUT]
                                                            const DT* __restrict__ res,    // [B][L][COUT]
                                                            DT* __restrict__ out,          // [B][L][COUT]
                                                            const float* __restrict__ wfc, // [2][COUT]
                                                            float* __restrict__ cam_out,   // [B][L]
                                                            float* __restrict__ part_out,  // [B][tiles_per_window][2]
                                                            int L, int ks, int cam_row, int tiles_per_window,
                                                            int total_tiles,
                                                            // generalisation used by the training path (dgrad):
                                                            // input row for (output t, tap j) = t + j - in_off,
                                                            // valid in [0, L_in); outside: clamp (replicate) or,
                                                            // if zero_row != null, a row of zeros.
                                                            int L_in, int in_off, const DT* __restrict__ zero_row,
                                                            // HEAD, optional: per-window max of the clipped CAM row
                                                            // (core.py:311), folded in with one atomic max per tile
                                                            float* __restrict__ cmax) {
    using G = ConvGeom<DT, CIN, COUT, NT, NWC>;
    constexpr int RB = G::RB, CPR = G::CPR, NCP = G::NCP, MT = G::MT, TN = G::TN;
    static_assert((MT == 1 || MT == 2 || MT == 4 || MT == 8) && NWC % MT == 0, "COUT is 32, 64, 128 or 256, one 32-channel tile per wave");
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bufb = G::buf_bytes(ks);
    float* red = reinterpret_cast<float*>(lds + 2 * bufb);  // HEAD only

    if (wave == NWC) {
        // ------------------------------ loader wave ------------------------------------
        const int pieces = bufb >> 10;
        auto stage = [&](int tile, char* buf) {
            int b, tx;
            conv_tile_coords(tile, tiles_per_window, total_tiles, L, TN, b, tx);
            const int t0 = tx * TN;
            const char* inb = reinterpret_cast<const char*>(in + (size_t)b * L_in * CIN);
            for (int pc = 0; pc < pieces; ++pc) {
                const int q = pc * 64 + lane;
                const int row = q / CPR, c = (q % CPR) ^ swz<CPR>(row);
                const int p = t0 - in_off + row;
                // replicate padding of THIS layer = clamp of the source row (camal_resnet.py:83)
                const int pcl = p < 0 ? 0 : (p > L_in - 1 ? L_in - 1 : p);
                const char* src = inb + (size_t)pcl * RB + c * 16;
                if (zero_row != nullptr && pcl != p) src = reinterpret_cast<const char*>(zero_row) + c * 16;
                glds16(src, buf + pc * 1024);
            }
        };
        int tile = blockIdx.x;
        if (tile < total_tiles) stage(tile, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int it = 0; tile < total_tiles; ++it, tile += gridDim.x) {
            const int next = tile + gridDim.x;
            if (next < total_tiles) stage(next, lds + ((it + 1) & 1) * bufb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (HEAD) { __syncthreads(); __syncthreads(); }
            __syncthreads();
        }
        return;
    }

    // ---------------------------------- compute waves --------------------------------------
    using F = Frag<DT>;
    const int mt = wave % MT, ng = wave / MT;
    const uint4* wp = wpack + (size_t)mt * ks * NCP * 64;  // wave-uniform; mma_layer adds the lane
    const int Rb = ng * NT * 32;  // first column of this wave inside the tile

    float bv[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) load4(bias + 32 * mt + F::chan(g, lane), bv[g]);

    __syncthreads();  // first tile has landed
    int it = 0;
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x, ++it) {
        const char* buf = lds + (it & 1) * bufb;
        int b, tx;
        conv_tile_coords(tile, tiles_per_window, total_tiles, L, TN, b, tx);
        const int t0 = tx * TN;

        F accw[1][NT];
        F (&acc)[NT] = accw[0];
        CSTAMP(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt].zero();

        // A window's last tile is often nearly empty (the data gradient's extended range L + 2 pad = 516 rows is eight
        // 64-row tiles + 4 rows): a wave whose columns all lie past the end skips its K loop, one that needs only its
        // first 32-column tile runs the K loop on that tile alone (fp32, two column tiles per wave).  Wave-uniform.
        const int cols_left = L - t0 - Rb;  // of this wave's NT * 32 columns, how many exist
        if constexpr (sizeof(DT) == 4 && NT == 2 && !HEAD) {
            if (cols_left > 32) mma_layer<DT, CIN, NT, 1>(buf, Rb, ks, wp, accw);
            else if (cols_left > 0) mma_layer<DT, CIN, 1, 1>(buf, Rb, ks, wp, reinterpret_cast<F (&)[1][1]>(accw));
        } else {
            mma_layer<DT, CIN, NT, 1>(buf, Rb, ks, wp, accw);
        }
        CSTAMP(1);

        // ---- epilogue.  acc[nt] group g: column t0 + Rb + 32 nt + col(g), channels 32 mt + chan(g) + 0..3
        if constexpr (!HEAD) {
            // residual rows first: all loads of the tile in flight before the first add
            float rv[HAS_RES ? NT : 1][4][4];
            if constexpr (HAS_RES) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        int t = t0 + Rb + nt * 32 + F::col(g, lane);
                        t = t < L ? t : L - 1;
                        load4(res + ((size_t)b * L + t) * COUT + 32 * mt + F::chan(g, lane), rv[nt][g]);
                    }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int t = t0 + Rb + nt * 32 + F::col(g, lane);
                    if (t < L) {
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v[i] = acc[nt].get(g, i) + bv[g][i];
                            if (RELU) v[i] = fmaxf(v[i], 0.f);
                            if constexpr (HAS_RES) v[i] += rv[nt][g][i];
                        }
                        store4(out + ((size_t)b * L + t) * COUT + 32 * mt + F::chan(g, lane), v[0], v[1], v[2], v[3]);
                    }
                }
        } else {
            float w0[4][4], w1[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                load4(wfc + 32 * mt + F::chan(g, lane), w0[g]);
                load4(wfc + COUT + 32 * mt + F::chan(g, lane), w1[g]);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float c0[F::NCOL], c1[F::NCOL];
#pragma unroll
                for (int cs = 0; cs < F::NCOL; ++cs) c0[cs] = c1[cs] = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float f = fmaxf(acc[nt].get(g, i) + bv[g][i], 0.f);
                        c0[F::colslot(g)] = fmaf(w0[g][i], f, c0[F::colslot(g)]);
                        c1[F::colslot(g)] = fmaf(w1[g][i], f, c1[F::colslot(g)]);
                    }
#pragma unroll
                for (int cs = 0; cs < F::NCOL; ++cs) {
                    const float s0 = F::col_reduce(c0[cs]), s1 = F::col_reduce(c1[cs]);
                    if (F::col_writer(lane)) {
                        const int tl = Rb + nt * 32 + F::col(cs, lane);
                        red[(mt * TN + tl) * 2 + 0] = s0;
                        red[(mt * TN + tl) * 2 + 1] = s1;
                    }
                }
            }
            __syncthreads();
            float s0 = 0.f, s1 = 0.f, mx = 0.f;
            for (int tl = threadIdx.x; tl < TN; tl += NWC * 64) {
                float c0 = 0.f, c1 = 0.f;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    c0 += red[(m * TN + tl) * 2 + 0];
                    c1 += red[(m * TN + tl) * 2 + 1];
                }
                const int t = t0 + tl;
                if (t < L) {
                    const float cv = cam_row ? c1 : c0;
                    cam_out[(size_t)b * L + t] = cv;
                    mx = fmaxf(mx, clip_cam(cv));
                    s0 += c0;
                    s1 += c1;
                }
            }
            // tile sums of both rows (+ max of the clipped CAM): wave shuffle tree, then NWC partials through LDS
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                s0 += __shfl_xor(s0, o);
                s1 += __shfl_xor(s1, o);
                mx = fmaxf(mx, __shfl_xor(mx, o));
            }
            float* red2 = red + MT * TN * 2;
            if (lane == 0) { red2[3 * wave] = s0; red2[3 * wave + 1] = s1; red2[3 * wave + 2] = mx; }
            __syncthreads();
            if (threadIdx.x == 0) {
                float* po = part_out + ((size_t)b * tiles_per_window + tx) * 2;
                float p0 = 0.f, p1 = 0.f, pm = 0.f;
#pragma unroll
                for (int wv = 0; wv < NWC; ++wv) { p0 += red2[3 * wv]; p1 += red2[3 * wv + 1]; pm = fmaxf(pm, red2[3 * wv + 2]); }
                po[0] = p0;
                po[1] = p1;
                // non-negative floats order like their bit patterns
                if (cmax != nullptr && pm > 0.f) atomicMax(reinterpret_cast<unsigned int*>(cmax) + b, __float_as_uint(pm));
            }
        }
        CSTAMP(2);
        __syncthreads();  // next tile landed (loader) and this buffer is free to refill
        CSTAMP(3);
    }
}

// ---------------------------------------------------------------------------------------
// Fused ResNetBlock: the three ConvBlocks of block 2 or block 3 (camal_resnet.py:26-59) in ONE
// kernel, intermediate activations never leave LDS.
//
// NWC compute waves (wave = 32 out-channels x NT*32 columns; 4 channel tiles x NWC/4 column groups)
// + 1 loader wave, persistent over time tiles of N = (NWC/4)*32*NT computed columns.  Every layer
// computes all N columns at the same positions p = t0 - 4 + c; the columns a layer cannot know
// (its halo) are garbage that shrinks inward by the layer's padding, so T = N - 8 columns are
// valid after the (k,7,3) stack and tiles advance by T.  Two LDS buffers P, Q alternate roles:
//     X(P) --conv k--> A1(Q) --conv 7--> A2(P) --conv 3--> epilogue     (next tile: X in Q)
// and the loader prefetches the next tile's X into the buffer A1 just vacated while conv 3 runs.
// Replicate padding is per layer: X rows are clamped by the loader; after
// conv k / conv 7 the first / last tile of a window overwrites the rows just outside [0, L)
// of A1 / A2 with the edge row (fix_edges) before the next layer reads them.
//   HAS_RES (block 2): out = relu(conv3) + BN(Conv1x1(X)) — the loader re-stages the centre rows
//     of X beside the prefetch; the 1x1 product is accumulated by MFMA on top of the already
//     rectified accumulator, so no second accumulator is needed.
//   HEAD (block 3): CAM head as in conv_kernel; the 128 x L feature map is never written.
// bf16 runs NWC = 4 (one wave per SIMD, wave = 32 out-channels x all N = 256 columns, NT = 8), takes its weights from
// ONE stream per out-channel tile through all layers and tiles (WRing / mma_stream_bf16), issues the LDS-store
// epilogues of the first two layers from inside their last K-loop group (ActStoreEpi) and computes the CAM head
// by MFMA on the accumulators; fp32 runs NWC = 8 with a weight ring per layer (mma_layer).
// ---------------------------------------------------------------------------------------
// bf16 output rows in 16-byte stores: a lane holds, for the fragment pair (mi, nj = 0 / 1), 4 channels (8 bytes) of
// column bcol and of column 16 + bcol; lanes l and l ^ 16 hold neighbouring 4-channel groups.  One
// v_permlane16_swap per dword hands the even k-block rows the neighbour's half of column nj = 0 and the odd ones
// the neighbour's half of column nj = 1: every lane then owns 8 consecutive channels of ONE column (the epilogue
// of the fused blocks is store-issue bound; half as many, twice as wide stores).
// in: x = the lane's 4 values of (mi, nj = 0), y = of (mi, nj = 1).  out: 16 bytes for column
// 16 * (q & 1) + bcol, channels 16 mi + 4 (q & ~1) .. + 7, q = lane >> 4.
__device__ __forceinline__ u32x4 pair_rows_bf16(const float (&x)[4], const float (&y)[4]) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 ux = {pk_bf16(x[0], x[1]), pk_bf16(x[2], x[3])}, uy = {pk_bf16(y[0], y[1]), pk_bf16(y[2], y[3])};
    const u32x2 r0 = __builtin_amdgcn_permlane16_swap(ux[0], uy[0], false, false);
    const u32x2 r1 = __builtin_amdgcn_permlane16_swap(ux[1], uy[1], false, false);
    u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
    return o;
}

// The accumulators of a layer start from its (BN-folded) bias, read from the LDS copy `cst` of the bias vector:
// the epilogue is then a bare relu, and no vector-memory load sits between a K loop and its epilogue.
template <typename DT, int NT>
__device__ __forceinline__ void init_acc_bias(Frag<DT> (&acc)[NT], const float* cst, int mt, int lane) {
    using F = Frag<DT>;
    float bv[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) load4(cst + 32 * mt + F::chan(g, lane), bv[g]);
    if constexpr (sizeof(DT) == 2) {
        // 64-bit moves: half the VALU issue slots of 16 NT v_mov_b32 (the init is exposed once per layer and tile)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x2 lo = {bv[g][0], bv[g][1]}, hi = {bv[g][2], bv[g][3]};
                f32x2 l2, h2;
                asm volatile("v_mov_b64 %0, %1" : "=v"(l2) : "v"(lo));
                asm volatile("v_mov_b64 %0, %1" : "=v"(h2) : "v"(hi));
                acc[nt].v[g] = f32x4{l2[0], l2[1], h2[0], h2[1]};
            }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[nt].set(g, i, bv[g][i]);
    }
}

// relu(acc) of this wave's 32-channel x NT*32-column tile -> CH-channel LDS rows; row0 = LDS row of the wave's
// first column (wave-uniform).  One address per group g (swz<> of a row does not change when 16 or 32 rows are
// added, so the column tiles are constant offsets from it).
template <typename DT, int NT, int CH = 128>
__device__ __forceinline__ void store_act_lds(char* dst, int row0, const Frag<DT> (&acc)[NT], int mt, int lane) {
    using F = Frag<DT>;
    constexpr int RB = CH * (int)sizeof(DT), CPR = RB / 16;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = row0 + F::col(g, lane);
        const int boff = (32 * mt + F::chan(g, lane)) * (int)sizeof(DT);
        char* pg = dst + row * RB + (((boff >> 4) ^ swz<CPR>(row)) << 4) + (boff & 15);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            store4_relu(reinterpret_cast<DT*>(pg + nt * 32 * RB), acc[nt].get(g, 0), acc[nt].get(g, 1), acc[nt].get(g, 2),
                        acc[nt].get(g, 3));
    }
}

// rows just outside the window take the edge row's values (replicate padding of THIS layer's output)
// (zero: they take zeros instead, the zero "same" padding of classifiers/resnet.py:62)
template <typename DT, int CH = 128>
__device__ __forceinline__ void fix_edges(char* buf, int guard, int g, int c_first, int c_last, bool lo, bool hi, int lane,
                                          bool zero = false) {
    constexpr int RB = CH * (int)sizeof(DT), CPR = RB / 16;
    if (lane < CPR) {
        if (lo) {
            const int src = c_first + guard;
            uint4 v = *reinterpret_cast<const uint4*>(buf + src * RB + ((lane ^ swz<CPR>(src)) << 4));
            if (zero) v = make_uint4(0u, 0u, 0u, 0u);
            for (int d = 1; d <= g; ++d) {
                const int dst = src - d;
                if (dst >= 0) *reinterpret_cast<uint4*>(buf + dst * RB + ((lane ^ swz<CPR>(dst)) << 4)) = v;
            }
        }
        if (hi) {
            const int src = c_last + guard;
            uint4 v = *reinterpret_cast<const uint4*>(buf + src * RB + ((lane ^ swz<CPR>(src)) << 4));
            if (zero) v = make_uint4(0u, 0u, 0u, 0u);
            for (int d = 1; d <= g; ++d) {
                const int dst = src + d;
                *reinterpret_cast<uint4*>(buf + dst * RB + ((lane ^ swz<CPR>(dst)) << 4)) = v;
            }
        }
    }
}

#ifdef CAMAL_STAMPS
// Diagnostic build only: s_memtime stamps of the second tile of every workgroup (block 3 kernel), per wave.
#define STAMP(i) do { if ((CAMAL_STAMPS == 2 ? !HEAD : HEAD) && it == 1 && blockIdx.x < 256 && (threadIdx.x & 63) == 0) \
    g_stamps[blockIdx.x][threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP1(i) do { if (it == 1 && blockIdx.x < 256 && (threadIdx.x & 63) == 0) \
    g_stamps[blockIdx.x][threadIdx.x >> 6][10 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do {} while (0)
#define STAMP1(i) do {} while (0)
#endif

struct BlockArgs {
    const void* in;   // [B][L][CIN]
    void* out;        // [B][L][128]   (block 2)
    const uint4 *w1, *w2, *w3, *wr;
    const float *b1, *b2, *b3, *br;
    const float* wfc;  // [2][128]     (block 3)
    float* cam;        // [B][L]
    float* part;       // [B][tiles_per_window][2]
    float* cmax;       // [B] or null: running max of the clipped CAM row (atomic max per head tile)
    int L, ks1, cam_row, tiles_per_window, total_tiles;
    const void* zero_row;  // ZPAD kernels: 512 B of zeros = what lies outside the window (classifiers/resnet.py:62)
    const char* wstream;   // bf16: the block's A fragments as one stream per out-channel tile (WRing), wstride bytes apart
    int wstride;
};

template <typename DT, int CIN, int NT, int NWC, int MTW = 1>
struct BlockGeom {
    static constexpr int ES = sizeof(DT);
    static constexpr int N = (NWC * MTW / 4) * 32 * NT;  // computed columns per tile
    static constexpr int THREADS = (NWC + 1) * 64;
    static constexpr int T = N - 8;            // valid columns per tile (halo 4 each side after convs 7 and 3)
    static constexpr int RBI = CIN * ES;       // X row bytes
    static constexpr int RBA = 128 * ES;       // A1 / A2 row bytes
    static constexpr int HEAD_BYTES = (4 * N * 2 + 3 * NWC) * 4;
    static constexpr int CST_FLOATS = 5 * 128;  // LDS copies of b1, b2, b3 and (br | the two head rows)
    __host__ __device__ static constexpr int x_bytes(int ks) { return (((N + ks - 1) * RBI) + 1023) & ~1023; }
    __host__ __device__ static constexpr int xc_bytes() { return ((N * RBI) + 1023) & ~1023; }
    // one buffer must hold X, or A1 (N + 6 rows), or [next X | centre rows of X] (HAS_RES)
    __host__ __device__ static constexpr int buf_bytes(int ks, bool has_res) {
        int b = x_bytes(ks) + (has_res ? xc_bytes() : 0);
        const int a = (((N + 6) * RBA) + 1023) & ~1023;
        return b > a ? b : a;
    }
};

// KS2 / ZPAD: the second ConvBlock's kernel size and the padding mode are compile-time (7, replicate = CamALResNet;
// 5, zeros = the ResNet variant): as run-time values they cost the block-2 kernel 1.5 %.
template <typename DT, int CIN, int NT, int NWC, int MTW, bool HAS_RES, bool HEAD, int KS2 = 7, bool ZPAD = false>
__global__ __launch_bounds__((NWC + 1) * 64) void block_kernel(BlockArgs a) {
    using G = BlockGeom<DT, CIN, NT, NWC, MTW>;
    constexpr int N = G::N, T = G::T, RBI = G::RBI, CPRI = RBI / 16;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = a.L, ks1 = a.ks1, pad1 = ks1 >> 1;
    const int bufb = G::buf_bytes(ks1, HAS_RES);
    const int xb = G::x_bytes(ks1);
    float* red = reinterpret_cast<float*>(lds + 2 * bufb);
    float* cst = reinterpret_cast<float*>(lds + 2 * bufb + (HEAD ? G::HEAD_BYTES : 0));  // [5][128]
    const int tpw = a.tiles_per_window;
    if (threadIdx.x < 128) {  // visible after the barrier that publishes the first X tile
        const int i = threadIdx.x;
        cst[i] = a.b1[i];
        cst[128 + i] = a.b2[i];
        cst[256 + i] = a.b3[i];
        if (HEAD) { cst[384 + i] = a.wfc[i]; cst[512 + i] = a.wfc[128 + i]; }
        else if (HAS_RES) cst[384 + i] = a.br[i];
    }

    if (wave == NWC) {
        // ------------------------------ loader wave ------------------------------------
        // rows r of the X tile <-> positions t0 - 4 - pad1 + r, clamped into the window
        auto stage = [&](int tile, char* buf, int row_shift, int bytes) {
            const int b = tile / tpw, t0 = (tile - b * tpw) * T;
            const char* inb = reinterpret_cast<const char*>(a.in) + (size_t)b * L * RBI;
            const int pieces = bytes >> 10;
            for (int pc = 0; pc < pieces; ++pc) {
                const int q = pc * 64 + lane;
                const int row = q / CPRI, c = (q % CPRI) ^ swz<CPRI>(row);
                const int p = t0 - 4 - pad1 + row_shift + row;
                const int pcl = p < 0 ? 0 : (p > L - 1 ? L - 1 : p);
                const char* src = inb + (size_t)pcl * RBI + c * 16;
                if (ZPAD && pcl != p) src = reinterpret_cast<const char*>(a.zero_row) + c * 16;
                glds16(src, buf + pc * 1024);
            }
        };
        int tile = blockIdx.x;
        if (tile < a.total_tiles) stage(tile, lds, 0, xb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int it = 0; tile < a.total_tiles; ++it, tile += gridDim.x) {
            const int b = tile / tpw, t0 = (tile - b * tpw) * T;
            const bool lo = t0 == 0, hi = L - 1 - t0 + 4 < N;  // position L-1 sits among this tile's columns
            char* Q = lds + ((it + 1) & 1) * bufb;
            __syncthreads();                       // B1: A1 complete
            if (lo || hi) __syncthreads();         // after fix_edges(A1)
            __syncthreads();                       // B2: A2 complete, A1 dead -> Q is free
            if (lo || hi) __syncthreads();         // after fix_edges(A2)
            const int next = tile + gridDim.x;
            if (HAS_RES) {
                stage(tile, Q + xb, pad1, G::xc_bytes());   // centre rows of THIS tile's X (row = column)
                if (next < a.total_tiles) {
                    stage(next, Q, 0, xb);
                    // counted wait: the centre rows were issued first and the prefetch is at least
                    // N * RBI / 1024 pieces (N rows), so that many outstanding means the centre rows landed
                    // while most of the prefetch stays in flight
                    constexpr int PF = N * RBI / 1024 > 63 ? 63 : N * RBI / 1024;  // fewer is stricter: still correct
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();                   // BR: centre rows landed
            } else if (next < a.total_tiles) {
                stage(next, Q, 0, xb);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                       // END: next X landed (HEAD: and the CAM partials are in `red`)
            if constexpr (HEAD) {
                // Second half of the CAM head, off the compute waves' critical path (they are already in the
                // next tile's first K loop; `red` is not written again before three more barriers that this
                // wave joins): per column the sum of the 4 channel tiles' partial dot products = the two CAM
                // rows' values; the row asked for goes out, both rows' tile sums (for the logits) and the
                // tile maximum of the clipped CAM (core.py:311) follow by a fixed shuffle tree.
                float s0 = 0.f, s1 = 0.f, mx = 0.f;
                for (int c = lane; c < N; c += 64) {
                    float c0 = 0.f, c1 = 0.f;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        c0 += red[(m * N + c) * 2 + 0];
                        c1 += red[(m * N + c) * 2 + 1];
                    }
                    const int p = t0 - 4 + c;
                    if (c >= 4 && c < 4 + T && p < L) {
                        const float cv = a.cam_row ? c1 : c0;
                        a.cam[(size_t)b * L + p] = cv;
                        mx = fmaxf(mx, clip_cam(cv));
                        s0 += c0;
                        s1 += c1;
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    s0 += __shfl_xor(s0, o);
                    s1 += __shfl_xor(s1, o);
                    mx = fmaxf(mx, __shfl_xor(mx, o));
                }
                if (lane == 0) {
                    float* po = a.part + ((size_t)b * tpw + (tile - b * tpw)) * 2;
                    po[0] = s0;
                    po[1] = s1;
                    // non-negative floats order like their bit patterns
                    if (a.cmax != nullptr && mx > 0.f) atomicMax(reinterpret_cast<unsigned int*>(a.cmax) + b, __float_as_uint(mx));
                }
            }
        }
        return;
    }

    // ---------------------------------- compute waves --------------------------------------
    // wave = MTW out-channel tiles (32 each) x NT column tiles; 4/MTW waves across channels
    using F = Frag<DT>;
    constexpr int WM = 4 / MTW;
    const int mt0 = (wave % WM) * MTW, ng = wave / WM;
    const int colw = ng * NT * 32;  // this wave's first column (wave-uniform)
    constexpr int NCPI = CIN / Elem<DT>::K_PER_PAIR, NCPA = 128 / Elem<DT>::K_PER_PAIR;
    // wave-uniform weight bases; mma_layer adds the lane
    const uint4* w1 = a.w1 + (size_t)mt0 * ks1 * NCPI * 64;
    constexpr int ks2 = KS2, pad2 = KS2 >> 1;
    constexpr bool zpad = ZPAD;
    const uint4* w2 = a.w2 + (size_t)mt0 * ks2 * NCPA * 64;
    const uint4* w3 = a.w3 + (size_t)mt0 * 3 * NCPA * 64;
    auto start_acc = [&](F (&acc)[MTW][NT], int layer) {  // accumulators = the layer's bias
#pragma unroll
        for (int m = 0; m < MTW; ++m) init_acc_bias<DT, NT>(acc[m], cst + 128 * layer, mt0 + m, lane);
    };
    // bf16: one weight stream through all the layers of all the tiles (WRing); PH2 = the ring slot layers 2.. start on
    constexpr bool STREAM = sizeof(DT) == 2;
    static_assert(!STREAM || (MTW == 1 && ((CIN == 128 && !HAS_RES) || (CIN == 64 && HAS_RES))), "ring phases");
    constexpr int PH2 = CIN == 64 ? 2 : 0;
    const char* wbase = a.wstream + (size_t)mt0 * a.wstride;
    WRing ring;
    if constexpr (STREAM) ring_begin(ring, wbase, lane * 16);

    // bf16 CAM head (below): the two head rows as a 16x16x32 A operand, rows 0 and 1, hi and lo bf16 halves
    u32x4 head_hi = {0u, 0u, 0u, 0u}, head_lo = {0u, 0u, 0u, 0u};
    if constexpr (HEAD && sizeof(DT) == 2) {
        const int row = lane & 15, q = lane >> 4;
        if (row < 2) {
            float wa[4], wb[4], r[8];
            load4(a.wfc + row * 128 + 32 * mt0 + 4 * q, wa);        // k-values 0..3 of k-block q: channels 4q..
            load4(a.wfc + row * 128 + 32 * mt0 + 16 + 4 * q, wb);   // k-values 4..7: channels 16 + 4q..
            const float w[8] = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                head_hi[j] = pk_bf16(w[2 * j], w[2 * j + 1]);
                r[2 * j] = w[2 * j] - __uint_as_float(head_hi[j] << 16);
                r[2 * j + 1] = w[2 * j + 1] - __uint_as_float(head_hi[j] & 0xffff0000u);
                head_lo[j] = pk_bf16(r[2 * j], r[2 * j + 1]);
            }
        }
    }

    __syncthreads();  // first X landed
    int it = 0;
    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x, ++it) {
        char* P = lds + (it & 1) * bufb;         // X, later A2
        char* Q = lds + ((it + 1) & 1) * bufb;   // A1, later [next X | X centre]
        const int b = tile / tpw, tx = tile - b * tpw, t0 = tx * T;
        const bool lo = t0 == 0, hi = L - 1 - t0 + 4 < N;  // position L-1 sits among this tile's columns
        const int c_first = 4 - t0, c_last = L - 1 - t0 + 4;  // columns of positions 0 and L-1

        F acc[MTW][NT];
        // ---- ConvBlock(k): X(P) -> A1(Q), rows = column + pad2 (3 for the 7-tap second layer)
        STAMP(0);
        start_acc(acc, 0);
        if constexpr (STREAM) {  // the LDS-store epilogue rides inside the last group of the K loop
            const ActStoreEpi<NT> epi(Q, colw + pad2, mt0, opaque(lane));
            mma_stream_bf16<CIN, NT, 0>(P, colw, ks1, ring, acc[0], epi);
            STAMP(1);
        } else {
            mma_layer<DT, CIN, NT, MTW>(P, colw, ks1, w1, acc);
            STAMP(1);
#pragma unroll
            for (int m = 0; m < MTW; ++m) store_act_lds<DT, NT>(Q, colw + pad2, acc[m], mt0 + m, opaque(lane));
        }
        STAMP(2);
        __syncthreads();  // B1
        if (lo || hi) {
            if (wave == 0) fix_edges<DT>(Q, pad2, pad2, c_first, c_last, lo, hi, lane, zpad);
            __syncthreads();
        }
        STAMP(3);
        // ---- ConvBlock(7 | 5): A1(Q) -> A2(P), rows = column + 1
        start_acc(acc, 1);
        if constexpr (STREAM) {
            const ActStoreEpi<NT> epi(P, colw + 1, mt0, opaque(lane));
            mma_stream_bf16<128, NT, PH2>(Q, colw, ks2, ring, acc[0], epi);
            STAMP(4);
        } else {
            mma_layer<DT, 128, NT, MTW>(Q, colw, ks2, w2, acc);
            STAMP(4);
#pragma unroll
            for (int m = 0; m < MTW; ++m) store_act_lds<DT, NT>(P, colw + 1, acc[m], mt0 + m, opaque(lane));
        }
        STAMP(5);
        __syncthreads();  // B2
        if (lo || hi) {
            if (wave == 0) fix_edges<DT>(P, 1, 1, c_first, c_last, lo, hi, lane, zpad);
            __syncthreads();
        }
        STAMP(6);
        // ---- ConvBlock(3): A2(P) -> registers
        start_acc(acc, 2);
        if constexpr (STREAM) {
            mma_stream_bf16<128, NT, PH2>(P, colw, 3, ring, acc[0]);
            if constexpr (!HAS_RES) ring.wptr = wbase + 3 * 2048;  // the ring already holds the next tile's first 3 groups
        } else {
            mma_layer<DT, 128, NT, MTW>(P, colw, 3, w3, acc);
        }
        STAMP(7);

        float bv[4][4];
        const int le = opaque(lane);  // epilogue addresses are computed here, not once per kernel
        if constexpr (!HEAD) {
            // relu(conv3 + b3) in registers (the bias came in with the accumulator), then the 1x1 skip
            // accumulates on top by MFMA
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[m][nt].set(g, i, fmaxf(acc[m][nt].get(g, i), 0.f));
            if constexpr (HAS_RES) {
                __syncthreads();  // BR: centre rows of X landed in Q + xb
                if constexpr (STREAM) {
                    mma_stream_bf16<CIN, NT, PH2>(Q + xb, colw, 1, ring, acc[0]);
                    ring.wptr = wbase + 3 * 2048;  // the ring already holds the next tile's first 3 groups
                } else {
                    const uint4* wr = a.wr + (size_t)mt0 * NCPI * 64;
                    mma_layer<DT, CIN, NT, MTW>(Q + xb, colw, 1, wr, acc);
                }
            }
            DT* outp = reinterpret_cast<DT*>(a.out);
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int mt = mt0 + m;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (HAS_RES) load4(cst + 384 + 32 * mt + F::chan(g, le), bv[g]);
                    else bv[g][0] = bv[g][1] = bv[g][2] = bv[g][3] = 0.f;
                }
                if constexpr (sizeof(DT) == 2) {
                    const int q = le >> 4;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi) {
                            float vx[4], vy[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                vx[i] = acc[m][nt].get(2 * mi, i) + bv[2 * mi][i];
                                vy[i] = acc[m][nt].get(2 * mi + 1, i) + bv[2 * mi + 1][i];
                            }
                            const u32x4 o = pair_rows_bf16(vx, vy);
                            const int c = colw + nt * 32 + F::col(q & 1, le);
                            const int p = t0 - 4 + c;
                            if (c >= 4 && c < 4 + T && p < L)
                                *reinterpret_cast<u32x4*>(outp + ((size_t)b * L + p) * 128 + 32 * mt + 16 * mi + 4 * (q & ~1)) = o;
                        }
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int c = colw + nt * 32 + F::col(g, le);
                            const int p = t0 - 4 + c;
                            if (c >= 4 && c < 4 + T && p < L)
                                store4(outp + ((size_t)b * L + p) * 128 + 32 * mt + F::chan(g, le), acc[m][nt].get(g, 0) + bv[g][0],
                                       acc[m][nt].get(g, 1) + bv[g][1], acc[m][nt].get(g, 2) + bv[g][2], acc[m][nt].get(g, 3) + bv[g][3]);
                        }
                }
            }
        } else if constexpr (sizeof(DT) == 2) {
            // CAM head on the matrix core: per 16-column tile the two head rows' dot products over the wave's 32
            // channels are ONE 16x16x32 MFMA whose B operand is the tile's accumulators themselves (relu, rounded
            // to bf16: the lane's 8 channels 4q.. and 16 + 4q.. of its column are the 8 k-values of k-block q) and
            // whose A operand holds the head rows in rows 0 and 1 (hi + lo bf16 halves: the weights stay exact to
            // 2^-17) -- no per-element fma, no cross-lane reduction; lanes 0..15 hold rows 0 / 1 of their column.
            // (Issued from inside the last group of the K loop, like the LDS-store epilogues: measured slower,
            // 15.2 -> 15.8 ms per step, the hi -> lo MFMA pairs are dependent and the registers spill.)
            const int mt = mt0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj) {
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const f32x4 v0 = acc[0][nt].v[nj], v1 = acc[0][nt].v[2 + nj];
                    const u32x4 raw = {pk_bf16(v0[0], v0[1]), pk_bf16(v0[2], v0[3]), pk_bf16(v1[0], v1[1]), pk_bf16(v1[2], v1[3])};
                    const s16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, raw), z8));
                    f32x4 d = {0.f, 0.f, 0.f, 0.f};
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, head_hi), bf, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, head_lo), bf, d, 0, 0, 0);
                    if (le < 16) {
                        const int c = colw + nt * 32 + F::col(nj, le);
                        *reinterpret_cast<float2*>(red + (mt * N + c) * 2) = make_float2(d[0], d[1]);
                    }
                }
        } else {
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int mt = mt0 + m;
                float w0[4][4], w1v[4][4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    load4(cst + 384 + 32 * mt + F::chan(g, le), w0[g]);
                    load4(cst + 512 + 32 * mt + F::chan(g, le), w1v[g]);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float c0[F::NCOL], c1[F::NCOL];
#pragma unroll
                    for (int cs = 0; cs < F::NCOL; ++cs) c0[cs] = c1[cs] = 0.f;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float f = fmaxf(acc[m][nt].get(g, i), 0.f);
                            c0[F::colslot(g)] = fmaf(w0[g][i], f, c0[F::colslot(g)]);
                            c1[F::colslot(g)] = fmaf(w1v[g][i], f, c1[F::colslot(g)]);
                        }
#pragma unroll
                    for (int cs = 0; cs < F::NCOL; ++cs) {
                        const float r0 = F::col_reduce(c0[cs]), r1 = F::col_reduce(c1[cs]);
                        if (F::col_writer(le)) {
                            const int c = colw + nt * 32 + F::col(cs, le);
                            red[(mt * N + c) * 2 + 0] = r0;
                            red[(mt * N + c) * 2 + 1] = r1;
                        }
                    }
                }
            }
            // the loader wave finishes the head behind the END barrier (sums over channel tiles, CAM row, tile sums)
        }
        STAMP(8);
        __syncthreads();  // END
        STAMP(9);
    }
    if constexpr (STREAM) ring_drain(ring);  // the last loads are still in flight
}

// ---------------------------------------------------------------------------------------
// Fused ResNetBlock 1 (camal_resnet.py:13-14,26-59 with in_channels = 1): the aggregate x goes in, the
// 64-channel block output comes out; ConvBlock(k) on 1 channel runs on the VALU (stem_kernel's arithmetic,
// same order), ConvBlock(7) and ConvBlock(3) on MFMA through LDS, and the skip BN(Conv1x1(x)) is a
// per-element fma in the epilogue.  Per timestep this reads 4 bytes and writes one activation row, where
// the per-layer path (stem, two conv launches) moves six activation rows through HBM and is bound by it.
//
// 8 waves, no loader wave (the tile's input is N + k - 1 floats): wave = channel half (32 of 64) x
// quarter of the N computed columns.  Columns, halos and the per-layer replicate fix-ups are those of
// block_kernel: positions p = t0 - 4 + c, T = N - 8 valid columns, A1 rows = column + 3, A2 rows =
// column + 1.  The stem weights are wave-uniform: they come through scalar loads, one x value per lane,
// column and tap from LDS.  bf16 runs 512-column tiles (NT = 4: four MFMAs per streamed weight fragment,
// as in block_kernel), fp32 256-column tiles (LDS).
// ---------------------------------------------------------------------------------------
struct Block1Args {
    const float* x;    // [B][L]
    void* out;         // [B][L][64]
    const uint4 *w2, *w3;            // packed ConvBlock(7), ConvBlock(3) weights (64 -> 64)
    const float *b2, *b3;
    int L, ks1, tiles_per_window, total_tiles;
    // bf16: the Cin = 1 ConvBlock's weights as MFMA A fragments, split in two bf16 halves (w = hi + lo):
    // [channel tile mt][tap group g of 32][mi][hi | lo][lane][16 B]; lane (i = l & 15, q = l >> 4), element jj
    // <-> w[32 mt + 16 mi + i][tap 32 g + 8 q + jj] (0 past the kernel size).  Null: the VALU stem.
    const uint4* stem_pack;
};

template <typename DT, int NT, int NWC = 8>
struct Block1Geom {
    static constexpr int ES = sizeof(DT);
    static constexpr int THREADS = NWC * 64;
    static constexpr int N = (NWC / 2) * 32 * NT;   // computed columns per tile
    static constexpr int T = N - 8;
    static constexpr int RB = 64 * ES;              // activation row bytes
    static constexpr int XS = N + 64;               // floats per x buffer (N + k - 1 <= N + 62)
    static constexpr int A1_BYTES = (((N + 6) * RB) + 1023) & ~1023;
    static constexpr int A2_BYTES = (((N + 2) * RB) + 1023) & ~1023;
    static constexpr int CST_OFF = A1_BYTES + A2_BYTES + 2 * XS * 4;  // LDS copies of b2, b3, wr, br, b1: [5][64] floats
    // bf16 (MFMA stem): the tile's x row once more as bf16 halves, [hi | lo][shift 0..7][XS]: copy s holds the row from
    // element s on, so that the 8 values from ANY element e on are one 16-byte aligned read of copy e % 8
    static constexpr int XB_OFF = CST_OFF + 5 * 64 * 4;
    static constexpr int LDS_BYTES = XB_OFF + (ES == 2 ? 2 * 8 * XS * 2 : 0);
};

// NWC = 8: one workgroup per CU, 512-column tiles (bf16).  NWC = 4: half the tile, two workgroups per CU.
// KS2 / ZPAD as in block_kernel (5, zeros: classifiers/resnet.py's block).
template <typename DT, int NT, int NWC, int KS2 = 7, bool ZPAD = false>
__global__ __launch_bounds__(NWC * 64) void block1_kernel(Block1Args a, const float* __restrict__ stem_w /*[k][64]*/,
                                                     const float* __restrict__ stem_b, const float* __restrict__ res_w,
                                                     const float* __restrict__ res_b) {
    using G = Block1Geom<DT, NT, NWC>;
    constexpr int TH = G::THREADS;
    constexpr int N = G::N, T = G::T, NCP = 64 / Elem<DT>::K_PER_PAIR;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* A1 = lds;
    char* A2 = lds + G::A1_BYTES;
    float* xs = reinterpret_cast<float*>(lds + G::A1_BYTES + G::A2_BYTES);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = a.L, ks1 = a.ks1, pad1 = ks1 >> 1, tpw = a.tiles_per_window;
    const int nx = N + ks1 - 1;  // x values per tile: positions t0 - 4 - pad1 + i, clamped (replicate padding of x)

    auto load_x = [&](int tile, int i) {
        const int b = tile / tpw, t0 = (tile - b * tpw) * T;
        const int p = t0 - 4 - pad1 + i;
        const int pcl = p < 0 ? 0 : (p > L - 1 ? L - 1 : p);
        const float v = a.x[(size_t)b * L + pcl];
        return (ZPAD && pcl != p) ? 0.f : v;
    };
    float* cst = reinterpret_cast<float*>(lds + G::CST_OFF);
    if (threadIdx.x < 64) {  // visible after the barrier below
        const int i = threadIdx.x;
        cst[i] = a.b2[i];
        cst[64 + i] = a.b3[i];
        cst[128 + i] = res_w[i];
        cst[192 + i] = res_b[i];
        cst[256 + i] = stem_b[i];
    }
    // MFMA stem (bf16): x value i of the tile as bf16 halves into the 8 shifted copies.  The values past the kernel's
    // last tap meet zero weights and must be finite: the copies start from zeros and only i < nx is ever written.
    __bf16* xb = reinterpret_cast<__bf16*>(lds + G::XB_OFF);
    const bool mfma_stem = sizeof(DT) == 2 && a.stem_pack != nullptr;
    auto put_x = [&](int i, float v) {
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
#pragma unroll
        for (int sft = 0; sft < 8; ++sft)
            if (i - sft >= 0) {
                xb[sft * G::XS + i - sft] = h;
                xb[(8 + sft) * G::XS + i - sft] = l;
            }
    };
    if (mfma_stem) {
        for (int i = threadIdx.x; i < 16 * G::XS / 2; i += TH) reinterpret_cast<unsigned*>(xb)[i] = 0u;
        __syncthreads();
    }
    constexpr int XPT = (G::XS + TH - 1) / TH;  // x values per thread
    if ((int)blockIdx.x < a.total_tiles)
#pragma unroll
        for (int u = 0; u < XPT; ++u)
            if ((int)threadIdx.x + TH * u < nx) {
                const float v = load_x(blockIdx.x, threadIdx.x + TH * u);
                xs[threadIdx.x + TH * u] = v;
                if (mfma_stem) put_x(threadIdx.x + TH * u, v);
            }
    __syncthreads();

    // stem role: channel half sh, 64 columns;  MFMA role: channel tile mt, NT column tiles
    using F = Frag<DT>;
    constexpr int SC = NT / 2;  // stem columns per lane (a wave covers N/4 = 64 SC columns)
    const int sh = wave & 1, scol = (wave >> 1) * 64 * SC + lane;
    const int mt = wave & 1, ng = wave >> 1;
    const int colw = ng * NT * 32;
    constexpr int pad2 = KS2 >> 1;
    const uint4* w2 = a.w2 + (size_t)mt * KS2 * NCP * 64;
    const uint4* w3 = a.w3 + (size_t)mt * 3 * NCP * 64;
    const float* sw = stem_w + 32 * sh;

    int it = 0;
    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x, ++it) {
        const float* xc = xs + (it & 1) * G::XS;
        float* xn = xs + ((it + 1) & 1) * G::XS;
        const int b = tile / tpw, t0 = (tile - b * tpw) * T;
        const bool lo = t0 == 0, hi = L - 1 - t0 + 4 < N;
        const int c_first = 4 - t0, c_last = L - 1 - t0 + 4;
        // next tile's x: issued now, parked in a register until the end of the tile
        const int next = tile + gridDim.x;
        float xpre[XPT];
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            xpre[u] = 0.f;
            if (next < a.total_tiles && (int)threadIdx.x + TH * u < nx) xpre[u] = load_x(next, threadIdx.x + TH * u);
        }

        STAMP1(0);
        // ---- ConvBlock(k), 1 -> 64: a1[c][co] = relu(b[co] + sum_j w[j][co] x[c + j])  -> A1 row c + pad2
        bool stem_done = false;
        if constexpr (sizeof(DT) == 2) {
            if (mfma_stem) {
                // On MFMA (v_mfma_f32_16x16x32_bf16, K = taps in groups of 32): D[co][c] += W[co][j] X[j][c] with
                // X[j][c] = x[c + j] read straight from the shifted bf16 copies of the x row in LDS (an im2col that is never stored).  x and w are
                // fp32: each goes in as two bf16 halves (hi = bf16(v), lo = bf16(v - hi)) and three products
                // (hi hi + lo hi + hi lo) keep 16 bits of each, where the VALU stem this replaces kept 24 — the result is
                // rounded to bf16 for the next layer either way.  The wave keeps its MFMA role (channel tile mt, columns
                // colw ..): 4.5 k (k = 7) to 9.4 k (k = 25) cycles of VALU per tile become 48 MFMAs + the operand build.
                F sacc[NT];
                init_acc_bias<DT, NT>(sacc, cst + 256, mt, lane);
                const int ngrp = (ks1 + 31) >> 5;
                const int bi = F::bcol(lane), bq = lane >> 4;
                for (int g = 0; g < ngrp; ++g) {
                    const uint4* sp = a.stem_pack + ((size_t)(mt * ngrp + g) * 4) * 64 + lane;
                    u32x4 ah[2], al[2];
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        ah[mi] = *reinterpret_cast<const u32x4*>(sp + (2 * mi) * 64);
                        al[mi] = *reinterpret_cast<const u32x4*>(sp + (2 * mi + 1) * 64);
                    }
#pragma unroll
                    for (int f = 0; f < 2 * NT; ++f) {
                        const int e = colw + 16 * f + bi + 32 * g + 8 * bq;  // first of the lane's 8 x values
                        const __bf16* xp = xb + (e & 7) * G::XS + (e & ~7);
                        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(xp);
                        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(xp + 8 * G::XS);
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi) {
                            f32x4& d = sacc[f >> 1].v[2 * mi + (f & 1)];
                            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[mi]), bh, d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[mi]), bh, d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[mi]), bl, d, 0, 0, 0);
                        }
                    }
                }
                store_act_lds<DT, NT, 64>(A1, colw + pad2, sacc, mt, opaque(lane));
                stem_done = true;
            }
        }
        if (!stem_done) {  // VALU: wave-uniform weights through scalar loads, one x value per lane, column and tap
            float sacc[SC][32];
#pragma unroll
            for (int u = 0; u < SC; ++u)
#pragma unroll
                for (int i = 0; i < 32; ++i) sacc[u][i] = stem_b[32 * sh + i];
            for (int j = 0; j < ks1; ++j) {
                float xv[SC];
#pragma unroll
                for (int u = 0; u < SC; ++u) xv[u] = xc[scol + 64 * u + j];
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const float wv = sw[j * 64 + i];
#pragma unroll
                    for (int u = 0; u < SC; ++u) sacc[u][i] = fmaf(wv, xv[u], sacc[u][i]);
                }
            }
            constexpr int CPR = G::RB / 16;
#pragma unroll
            for (int u = 0; u < SC; ++u) {
                const int row = scol + 64 * u + pad2, s = swz<CPR>(row);
#pragma unroll
                for (int q = 0; q < 8; ++q) {  // 8 groups of 4 channels
                    const int boff = (32 * sh + 4 * q) * (int)sizeof(DT);
                    DT* p = reinterpret_cast<DT*>(A1 + row * G::RB + (((boff >> 4) ^ s) << 4) + (boff & 15));
                    store4(p, fmaxf(sacc[u][4 * q], 0.f), fmaxf(sacc[u][4 * q + 1], 0.f), fmaxf(sacc[u][4 * q + 2], 0.f),
                           fmaxf(sacc[u][4 * q + 3], 0.f));
                }
            }
        }
        STAMP1(1);
        __syncthreads();
        if (lo || hi) {
            if (wave == 0) fix_edges<DT, 64>(A1, pad2, pad2, c_first, c_last, lo, hi, lane, ZPAD);
            __syncthreads();
        }
        STAMP1(2);
        F acc[1][NT];
        // ---- ConvBlock(7): A1 -> A2, rows = column + 1   (accumulators start from the bias)
        init_acc_bias<DT, NT>(acc[0], cst, mt, lane);
        mma_layer<DT, 64, NT, 1>(A1, colw, KS2, w2, acc);
        STAMP1(3);
        store_act_lds<DT, NT, 64>(A2, colw + 1, acc[0], mt, opaque(lane));
        __syncthreads();
        if (lo || hi) {
            if (wave == 0) fix_edges<DT, 64>(A2, 1, 1, c_first, c_last, lo, hi, lane, ZPAD);
            __syncthreads();
        }
        // ---- ConvBlock(3): A2 -> registers; out = relu(. + b3) + (wr x + br)   (skip added after the ReLU)
        STAMP1(4);
        init_acc_bias<DT, NT>(acc[0], cst + 64, mt, lane);
        mma_layer<DT, 64, NT, 1>(A2, colw, 3, w3, acc);
        STAMP1(5);
        {
            float wr[4][4], br[4][4];
            const int le = opaque(lane);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                load4(cst + 128 + 32 * mt + F::chan(g, le), wr[g]);
                load4(cst + 192 + 32 * mt + F::chan(g, le), br[g]);
            }
            DT* outp = reinterpret_cast<DT*>(a.out);
            if constexpr (sizeof(DT) == 2) {
                const int q = le >> 4;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int cx = colw + nt * 32 + F::col(0, le);  // the lane's columns of nj = 0 and (cx + 16) of nj = 1
                    const float xv0 = xc[cx + pad1], xv1 = xc[cx + 16 + pad1];
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        float vx[4], vy[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            vx[i] = fmaxf(acc[0][nt].get(2 * mi, i), 0.f) + fmaf(wr[2 * mi][i], xv0, br[2 * mi][i]);
                            vy[i] = fmaxf(acc[0][nt].get(2 * mi + 1, i), 0.f) + fmaf(wr[2 * mi + 1][i], xv1, br[2 * mi + 1][i]);
                        }
                        const u32x4 o = pair_rows_bf16(vx, vy);
                        const int c = cx + 16 * (q & 1);
                        const int p = t0 - 4 + c;
                        if (c >= 4 && c < 4 + T && p < L)
                            *reinterpret_cast<u32x4*>(outp + ((size_t)b * L + p) * 64 + 32 * mt + 16 * mi + 4 * (q & ~1)) = o;
                    }
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = colw + nt * 32 + F::col(g, le);
                        const int p = t0 - 4 + c;
                        if (c >= 4 && c < 4 + T && p < L) {
                            const float xv = xc[c + pad1];
                            float v[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = fmaxf(acc[0][nt].get(g, i), 0.f) + fmaf(wr[g][i], xv, br[g][i]);
                            store4(outp + ((size_t)b * L + p) * 64 + 32 * mt + F::chan(g, le), v[0], v[1], v[2], v[3]);
                        }
                    }
            }
        }
        STAMP1(6);
#pragma unroll
        for (int u = 0; u < XPT; ++u)
            if (next < a.total_tiles && (int)threadIdx.x + TH * u < nx) {
                xn[threadIdx.x + TH * u] = xpre[u];
                if (mfma_stem) put_x(threadIdx.x + TH * u, xpre[u]);  // the copies were last read by this tile's stem
            }
        __syncthreads();  // END: A1/A2 free, next x in place
        STAMP1(7);
    }
}

// CAM of a hooked layer other than the last block (class_activation_map.py:25-31,42-48 with last_conv_layer =
// layers[1]): cam[r] = sum_c W_fc[label][c] * F[r][c] over the C channels of block 2's output F [rows][C]
// (channels-last, DT).  One row per lane, 16-byte chunks; HBM-bound, off the hot path.
template <typename DT, int C>
__global__ __launch_bounds__(256) void cam_dot_kernel(const DT* __restrict__ feat, const float* __restrict__ wrow,
                                                      float* __restrict__ cam, size_t rows) {
    __shared__ float w[C];
    for (int i = threadIdx.x; i < C; i += 256) w[i] = wrow[i];
    __syncthreads();
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
        const DT* f = feat + r * C;
        float acc = 0.f;
#pragma unroll 4
        for (int c = 0; c < C; c += 4) {
            float v[4];
            load4(f + c, v);
            acc = fmaf(w[c], v[0], acc); acc = fmaf(w[c + 1], v[1], acc);
            acc = fmaf(w[c + 2], v[2], acc); acc = fmaf(w[c + 3], v[3], acc);
        }
        cam[r] = acc;
    }
}

// Zero fill as a kernel: camal_localize clears the row maxima with it.  (hipMemsetAsync on this ROCm is not replayed as
// part of a captured HIP graph: a replay on new inputs kept the maxima of the capture-time call.)
__global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
}

// logits[b][k] = (sum over tiles of part[b][tile][k]) / L + b_fc[k]   (camal_resnet.py:23)
__device__ __forceinline__ void member_logits(const float* part, int ntiles, int L, const float* bfc, float& l0,
                                              float& l1) {
    float s0 = 0.f, s1 = 0.f;
    for (int i = 0; i < ntiles; ++i) { s0 += part[2 * i]; s1 += part[2 * i + 1]; }
    l0 = s0 / (float)L + bfc[0];
    l1 = s1 / (float)L + bfc[1];
}

__global__ void logits_kernel(const float* __restrict__ part, const float* __restrict__ bfc,
                              float* __restrict__ logits, int B, int ntiles, int L) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float l0, l1;
    member_logits(part + (size_t)b * ntiles * 2, ntiles, L, bfc, l0, l1);
    logits[2 * b] = l0;
    logits[2 * b + 1] = l1;
}

// ---------------------------------------------------------------------------------------
// Fused ensemble post-processing (HBM-bound: reads E CAM rows + the aggregate once, writes
// states [+ power, + average CAM]).  Grid = (time tiles of POST_TILE, windows); a workgroup
// owns one tile of one window, one dword per lane per access (256 contiguous bytes per wave
// instruction).  Restates, in the reference's order and precisions:
//   core.py:305      prob += softmax(logits)[1]             (float32 softmax, float64 sum)
//   core.py:308-317  if argmax==1: c = clip(cam,0); nan/inf->0; if max(c)>0: pred += c/max(c)
//   core.py:320-324  prob/=E; pred/=E; pred *= round_half_even(prob)
//   core.py:429-437  ma = convolve(pred, ones(w),'same')/w; s = 2/(1+exp(-ma*x))-1; round(s)
//   core.py:216-218  power = min(int(state)*p_on, x*scale_mul+scale_add)
// max(c) per member and window comes from the CAM-head epilogue (cmax, one atomic max per
// head tile), so the rows are read exactly once; OWN_MAX = true (camal_postprocess on rows
// the caller brought) scans the window's rows for it instead.
// ---------------------------------------------------------------------------------------
#define CAMAL_MAX_MEMBERS 16
#ifndef POST_THREADS
#define POST_THREADS 128
#endif
#define POST_PER_LANE 4
#ifndef POST_MU
#define POST_MU 4  // members per load round
#endif
#define POST_TILE (POST_THREADS * POST_PER_LANE)

struct PostArgs {
    const float* cam;      // member e, window b: cam + e*cam_estride + b*L
    const float* logits;   // logits + e*lg_estride + 2*b                 (OWN_MAX only)
    const double* hdr;     // [B][E + 1] window headers from header_kernel  (!OWN_MAX)
    size_t cam_estride, lg_estride;
    const float* x;        // [B][L]
    const double* x64;     // [B][L] or null
    const int8_t* y;       // [B] or null                                  (OWN_MAX only)
    double* prob_out;      // [B]                                          (OWN_MAX only)
    float* avg_out;        // [B][L] or null
    float* state_out;      // [B][L]   (one of state_out / state8_out)
    int8_t* state8_out;    // [B][L] or null: the states as one byte each (state_code), for callers that ship them over PCIe
    float* power_out;      // [B][L] or null
    int E, B, L, w_ma, gate_mode, pred_f64;
    double p_on, scale_mul, scale_add;
};

// Window headers for post_kernel, one thread per window (camal_localize: replaces one logits launch per
// member and pass, and the softmax / vote every post_kernel tile used to redo):
//   logits[e][b][k] = sum over head tiles of part / L + b_fc      (camal_resnet.py:23)
//   p1 = softmax(logits)[1] in float32, label = p1 > p0           (class_activation_map.py:59-64)
//   prob = sum_e p1 / E in float64; gates (core.py:292-300); vote = round_half_even(prob)   (core.py:305,320-324)
//   hdr[b] = { vote (0: the window's prediction is 0), per member 1/max(clip(cam)) or 0 if it adds nothing }
struct HeaderArgs {
    const float* part;      // member e, window b: part + e*part_estride + b*ntiles[e]*2
    const float* cmax;      // cmax + e*cmax_estride + b
    float* logits;          // out: logits + e*lg_estride + 2*b
    size_t part_estride, cmax_estride, lg_estride;
    const float* bfc[CAMAL_MAX_MEMBERS];  // each member's head bias [2]
    int ntiles[CAMAL_MAX_MEMBERS];
    const int8_t* y;        // [B] or null
    double* prob_out;       // [B]
    double* hdr;            // [B][E + 1]
    int E, B, L, gate_mode;
};

__global__ __launch_bounds__(128) void header_kernel(HeaderArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    double rinv[CAMAL_MAX_MEMBERS];
    double psum = 0.0;
    int lab0 = 0;
    for (int e = 0; e < a.E; ++e) {
        float l0, l1;
        member_logits(a.part + e * a.part_estride + (size_t)b * a.ntiles[e] * 2, a.ntiles[e], a.L, a.bfc[e], l0, l1);
        float* lg = a.logits + e * a.lg_estride + 2 * (size_t)b;
        lg[0] = l0;
        lg[1] = l1;
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        const int lab = p1 > p0 ? 1 : 0;
        if (e == 0) lab0 = lab;
        psum += (double)p1;
        const float cm = a.cmax[e * a.cmax_estride + b];
        rinv[e] = (lab && cm > 0.f) ? 1.0 / (double)cm : 0.0;
    }
    bool skip = false;
    if (a.y) skip = a.y[b] < 1;
    else if (a.gate_mode == 1) skip = lab0 < 1;
    const double prob = skip ? 0.0 : psum / (double)a.E;
    const double vote = rint(prob);  // np.round: half to even
    const bool pass = !skip && vote != 0.0;
    a.prob_out[b] = prob;
    double* h = a.hdr + (size_t)b * (a.E + 1);
    h[0] = pass ? vote : 0.0;
    for (int e = 0; e < a.E; ++e) h[1 + e] = pass ? rinv[e] : 0.0;
}

// np.round(2/(1+exp(-z)) - 1) with z = (sum/w)*x  (core.py:431-437), values -1, -0, +0, 1.
// The rounded value only depends on which side of +-ln 3 (and of 0) z falls, so the exponential
// and the two float64 divisions run only where that is in doubt: within 1e-6 of +-ln 3, for
// 1e-17 <= |z| < 1e-9 (where exp(-z) may or may not round to 1, which decides -0 against +0),
// and for NaN.  Everywhere else the answer of the full float64 evaluation is implied:
//   |z| < 1e-17: exp(-z) == 1 -> +0;  else the sign of z, and magnitude 1 iff |z| > ln 3.
__device__ __forceinline__ double attention_state(double sum, int w, double inv_w, double x) {
    constexpr double LN3 = 1.0986122886681098;
    const double zf = (sum * inv_w) * x;  // within a few ulp of (sum / w) * x
    const double az = fabs(zf);
    if (az < 1e-17) return 0.0;
    if (az >= 1e-9 && fabs(az - LN3) > 1e-6) return copysign(az > LN3 ? 1.0 : 0.0, zf);
    const double z = (sum / (double)w) * x;
    return rint(2.0 * (1.0 / (1.0 + exp(-z))) - 1.0);
}

// np.round's four possible states in one byte: 0 -> 0, 1 -> 1, -1 -> -1, -0.0 -> -128 (the sign bit alone): what
// camal_localize_i8 writes, 1 byte per timestamp instead of 4 (include/camal_hip.h: CAMAL_STATE_NEG_ZERO)
__device__ __forceinline__ int state_code(double st) {
    if (st == 0.0) return signbit(st) ? -128 : 0;
    return (int)st;
}
__device__ __forceinline__ unsigned pack_codes(const int (&c)[4]) {
    return (unsigned)(c[0] & 0xff) | ((unsigned)(c[1] & 0xff) << 8) | ((unsigned)(c[2] & 0xff) << 16) | ((unsigned)(c[3] & 0xff) << 24);
}

// NS values of one window row starting at byte offset off0 (VEC: NS consecutive floats in one 16-byte load,
// the row is then 16-byte aligned) or at the NS byte offsets off[] (one dword each)
template <int NS, bool VEC>
__device__ __forceinline__ void load_row(const float* rowp, const unsigned (&off)[NS], float (&v)[NS]) {
    const char* c = reinterpret_cast<const char*>(rowp);
    if constexpr (VEC) {
        static_assert(NS == 4, "one float4 per lane");
        const float4 t = *reinterpret_cast<const float4*>(c + off[0]);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < NS; ++j) v[j] = *reinterpret_cast<const float*>(c + off[j]);
    }
}

// pred[t] of core.py:308-324 at NS positions of window row `row`: members that voted 'on' add
// clip(cam)/max(clip(cam)) in member order (float32 sum when the caller's data was float32,
// float64 otherwise), /E, * vote.  c/max as float32(float64(c) * (1/float64(max))): 53 bits
// cover the correctly rounded float32 quotient of two float32 numbers.
// rinv[e] = 1/max of member e, or 0 when the member adds nothing (voted off, max 0, window off): wave-uniform.
template <int NS, bool F64, int MU, bool VEC>
__device__ __forceinline__ void ensemble_pred(const PostArgs& a, const double* rinv, size_t row, const unsigned (&off)[NS],
                                              const bool (&ok)[NS], double vote, double (&out)[NS],
                                              bool do_halo, unsigned hoff, bool hok, double& hout) {
    // off[j]: BYTE offset of the position inside the window row, already clamped into it (a position that
    // is not wanted still loads a valid address; ok[j] = false zeroes its result).  32-bit byte offsets on a
    // wave-uniform row base let the loads use the scalar-base addressing form (no 64-bit lane addresses).
    // do_halo: this lane also computes ONE position of the moving average's halo (byte offset hoff); its loads
    // go out in the same round as the tile's, so the halo costs no memory round trip of its own.
    float accf[NS + 1];
    double accd[NS + 1];
#pragma unroll
    for (int j = 0; j <= NS; ++j) { accf[j] = 0.f; accd[j] = 0.0; }
    // MU members per round: all their loads are in flight before the first is consumed (every round is
    // one more memory round trip on the workgroup's critical path; the host picks MU);
    // a member that is off loads nothing and adds nothing
    for (int e0 = 0; e0 < a.E; e0 += MU) {
        float v[MU][NS + 1];
        double ri[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int e = e0 + u < a.E ? e0 + u : 0;
            ri[u] = e0 + u < a.E ? rinv[e] : 0.0;
            // a member that is off loads nothing (wave-uniform branch).  Branch-free loads that point an off member
            // at the aggregate's row, so that a round's loads never wait for each other, measured 5 % slower.
#pragma unroll
            for (int j = 0; j <= NS; ++j) v[u][j] = 0.f;
            if (ri[u] != 0.0) {
                const float* rp = a.cam + e * a.cam_estride + row;
                float t[NS];
                load_row<NS, VEC>(rp, off, t);
#pragma unroll
                for (int j = 0; j < NS; ++j) v[u][j] = t[j];
                if (do_halo) v[u][NS] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + hoff);
            }
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            if (ri[u] == 0.0) continue;  // wave-uniform: member off
#pragma unroll
            for (int j = 0; j <= NS; ++j) {
                const float q = (float)((double)clip_cam(v[u][j]) * ri[u]);
                if (F64) accd[j] += (double)q;
                else accf[j] += q;
            }
        }
    }
    // pred / len(models): float64 division, or the float32 one by the same float64-product argument
    const double inv_e = 1.0 / (double)a.E;
#pragma unroll
    for (int j = 0; j <= NS; ++j) {
        const double avg = F64 ? accd[j] / (double)a.E : (double)(float)((double)accf[j] * inv_e);
        if (j < NS) out[j] = ok[j] ? avg * vote : 0.0;
        else hout = hok ? avg * vote : 0.0;
    }
}

// OWN_MAX = false (camal_localize): the window's header comes from header_kernel through wave-uniform (scalar)
//   loads, the row maxima from the CAM-head epilogue: the CAM rows are read exactly once, and nothing sits
//   between the kernel's start and its loads.
// OWN_MAX = true (camal_postprocess on rows and logits the caller brought): softmax, vote and a scan of the
//   window's rows for their maxima in the kernel.
// VEC: L % 4 == 0, every lane works on 4 consecutive timesteps through 16-byte loads and stores; else lane t of
//   the tile takes timesteps t, t + POST_THREADS, ... through dwords.
template <bool OWN_MAX, int MU, bool VEC>
__global__ __launch_bounds__(POST_THREADS) void post_kernel(PostArgs a) {
    __shared__ double s_rinv[OWN_MAX ? CAMAL_MAX_MEMBERS : 1];
    __shared__ float s_p1[OWN_MAX ? CAMAL_MAX_MEMBERS : 1];
    __shared__ int s_lab[OWN_MAX ? CAMAL_MAX_MEMBERS : 1];
    __shared__ float s_red[POST_THREADS / 64];
    __shared__ int s_y;
    __shared__ double s_pred[POST_TILE + 64];
    const int b = blockIdx.y, c0 = blockIdx.x * POST_TILE, L = a.L, E = a.E;
    const int tid = threadIdx.x;
    const int n = min(POST_TILE, L - c0);
    const size_t row = (size_t)b * L;

    // positions of this lane inside the tile, and their (clamped) byte offsets inside the window row
    int pos[POST_PER_LANE];
    unsigned off[POST_PER_LANE];
    bool ok[POST_PER_LANE];
#pragma unroll
    for (int j = 0; j < POST_PER_LANE; ++j) {
        pos[j] = VEC ? 4 * tid + j : tid + POST_THREADS * j;
        ok[j] = pos[j] < n;
        // VEC: n is a multiple of 4, a lane is in or out as a whole; an idle lane re-reads the tile's first chunk
        off[j] = 4u * (unsigned)(c0 + (VEC ? (ok[0] ? pos[j] : j) : min(pos[j], n - 1)));
    }

    double vote;
    bool pass;
    const double* rinv;
    float xf[POST_PER_LANE];
    if constexpr (!OWN_MAX) {
        const double* h = a.hdr + (size_t)b * (E + 1);  // wave-uniform address
        vote = h[0];
        pass = vote != 0.0;
        rinv = h + 1;
        load_row<POST_PER_LANE, VEC>(a.x + row, off, xf);
    } else {
        // window header, one lane per member: float32 softmax, label.  Its loads are issued FIRST and
        // unconditionally; the loads of the aggregate (needed whatever the vote) follow.
        const int he = min(tid, E - 1);
        const float* lg = a.logits + he * a.lg_estride + 2 * (size_t)b;
        const float l0 = lg[0], l1 = lg[1];
        const int yv = *(a.y ? a.y + b : reinterpret_cast<const int8_t*>(lg));  // always a valid address
        __builtin_amdgcn_sched_barrier(0);
        load_row<POST_PER_LANE, VEC>(a.x + row, off, xf);
        __builtin_amdgcn_sched_barrier(0);
        if (tid < E) {
            const float m = fmaxf(l0, l1);
            const float e0 = expf(l0 - m), e1 = expf(l1 - m);
            const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
            s_p1[tid] = p1;
            s_lab[tid] = p1 > p0 ? 1 : 0;
            if (tid == 0) s_y = yv;
        }
        __syncthreads();
        bool skip = false;
        if (a.y) skip = s_y < 1;
        else if (a.gate_mode == 1) skip = s_lab[0] < 1;
        double psum = 0.0;
        for (int e = 0; e < E; ++e) psum += (double)s_p1[e];
        const double prob = skip ? 0.0 : psum / (double)E;
        vote = rint(prob);  // np.round: half to even
        if (tid == 0 && blockIdx.x == 0) a.prob_out[b] = prob;
        pass = !skip && vote != 0.0;
        rinv = s_rinv;
        // the rows' maxima (core.py:311): scan the window's rows of the members that voted on
        const int lane = tid & 63, wave = tid >> 6;
        for (int e = 0; e < E; ++e) {
            float m = 0.f;
            if (pass && s_lab[e]) {
                const float* c = a.cam + e * a.cam_estride + row;
                for (int t = tid; t < L; t += POST_THREADS) m = fmaxf(m, clip_cam(c[t]));
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            }
            __syncthreads();
            if (lane == 0) s_red[wave] = m;
            __syncthreads();
            if (tid == 0) {
                float mm = 0.f;
#pragma unroll
                for (int wv = 0; wv < POST_THREADS / 64; ++wv) mm = fmaxf(mm, s_red[wv]);
                s_rinv[e] = mm > 0.f ? 1.0 / (double)mm : 0.0;
            }
        }
        __syncthreads();
    }

    const int w = a.w_ma, hi = (w - 1) / 2, lo = hi - w + 1;  // lo <= 0
    const double inv_w = 1.0 / (double)w;

    double xd[POST_PER_LANE];
    if (a.x64) {
#pragma unroll
        for (int j = 0; j < POST_PER_LANE; ++j) xd[j] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.x64 + row) + 2 * (size_t)off[j]);
    }

    if (pass) {
        // ---- ensemble-averaged, vote-gated CAM of the tile -> LDS slot (position - c0 - lo)
        // np.convolve(p, ones(w), 'same')[t] = sum_{m = t + lo .. t + hi} p[m]: the tile needs pred on
        // [c0 + lo, c0 + n + hi)
        {
            // halo: -lo positions before the tile, hi after it (zero outside the window), one per lane of the
            // first w - 1 lanes, loaded in the same rounds as the tile itself
            const bool do_halo = tid < w - 1;
            const bool left = tid < -lo;
            const int th = left ? c0 + lo + tid : c0 + n + (tid + lo);
            const unsigned hoff = 4u * (unsigned)min(max(th, 0), L - 1);
            const bool hok = th >= 0 && th < L;
            double pr[POST_PER_LANE], hpr = 0.0;
            if (a.pred_f64) ensemble_pred<POST_PER_LANE, true, MU, VEC>(a, rinv, row, off, ok, vote, pr, do_halo, hoff, hok, hpr);
            else ensemble_pred<POST_PER_LANE, false, MU, VEC>(a, rinv, row, off, ok, vote, pr, do_halo, hoff, hok, hpr);
#pragma unroll
            for (int j = 0; j < POST_PER_LANE; ++j)
                if (ok[j]) s_pred[pos[j] - lo] = pr[j];
            if (do_halo) s_pred[left ? tid : n + tid] = hpr;
        }
        __syncthreads();
    }

    // ---- moving average, rectified sigmoid of (ma * aggregate), round, power
    float st4[POST_PER_LANE], av4[POST_PER_LANE], pw4[POST_PER_LANE];
    int sc4[POST_PER_LANE];
#pragma unroll
    for (int j = 0; j < POST_PER_LANE; ++j) {
        st4[j] = av4[j] = pw4[j] = 0.f;
        sc4[j] = 0;
        if (!ok[j]) continue;
        const int i = pos[j];
        double sum = 0.0;
        if (pass)
            for (int m = 0; m < w; ++m) sum += s_pred[i + m];
        const double xv = a.x64 ? xd[j] : (double)xf[j];
        const double st = attention_state(sum, w, inv_w, xv);
        st4[j] = (float)st;
        sc4[j] = state_code(st);
        av4[j] = pass ? (float)s_pred[i - lo] : 0.f;
        const double agg = xv * a.scale_mul + a.scale_add;
        double yh = (double)(int)st * a.p_on;  // astype(int) of -1, -0, 0, 1
        if (yh > agg) yh = agg;
        pw4[j] = (float)yh;
    }
    if constexpr (VEC) {
        if (ok[0]) {
            const size_t o = row + c0 + pos[0];
            if (a.state8_out) *reinterpret_cast<unsigned*>(a.state8_out + o) = pack_codes(sc4);
            else *reinterpret_cast<float4*>(a.state_out + o) = make_float4(st4[0], st4[1], st4[2], st4[3]);
            if (a.avg_out) *reinterpret_cast<float4*>(a.avg_out + o) = make_float4(av4[0], av4[1], av4[2], av4[3]);
            if (a.power_out) *reinterpret_cast<float4*>(a.power_out + o) = make_float4(pw4[0], pw4[1], pw4[2], pw4[3]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < POST_PER_LANE; ++j) {
            if (!ok[j]) continue;
            const size_t o = row + c0 + pos[j];
            if (a.state8_out) a.state8_out[o] = (int8_t)sc4[j];
            else a.state_out[o] = st4[j];
            if (a.avg_out) a.avg_out[o] = av4[j];
            if (a.power_out) a.power_out[o] = pw4[j];
        }
    }
}

// ---- streaming variant of post_kernel for the common case ---------------------------------------------------------
// (camal_localize with header_kernel's window headers, L % 4 == 0, float32 aggregate, moving-average window W <= 9,
// no averaged-CAM output).  Same arithmetic, statement for statement, as post_kernel (bit-identical results); what
// changes is the shape of the memory stream: a WAVE is the unit of work, not a workgroup.  A wave takes one segment of
// POST_SEG = 62 x 4 timesteps of one window; lanes 1..62 own 4 consecutive timesteps each (one 16-byte access per
// row), lanes 0 and 63 compute the 4 positions before / after the segment only to feed the moving average, which
// reaches its neighbours through lane shuffles: no LDS, no barrier, no halo round, every load of an item in flight at
// once, and a grid-stride loop instead of one small workgroup per tile.
#define POST_SEG_LANES 62
#define POST_SEG (POST_SEG_LANES * 4)

template <int NE, bool F64, int W>
__global__ __launch_bounds__(256) void post_stream_kernel(PostArgs a, int segs_per_window, int total_items) {
    constexpr int HI = (W - 1) / 2, LO = HI - W + 1;  // np.convolve(..., 'same'): m in [t + LO, t + HI]
    static_assert(LO >= -4 && HI <= 4, "one neighbour lane on each side covers the window");
    const int lane = threadIdx.x & 63;
    const int L = a.L;
    const int nwaves = gridDim.x * 4;
    const double inv_w = 1.0 / (double)W, inv_e = 1.0 / (double)NE;
    for (int item0 = blockIdx.x * 4 + (threadIdx.x >> 6); item0 < total_items; item0 += nwaves) {
        const int item = __builtin_amdgcn_readfirstlane(item0);
        const int b = item / segs_per_window, seg = item - b * segs_per_window;
        const int t0 = seg * POST_SEG + (lane - 1) * 4;
        const bool inrow = t0 >= 0 && t0 < L;  // L % 4 == 0: a lane's four timesteps are in or out together
        const bool owner = inrow && lane >= 1 && lane <= POST_SEG_LANES;
        const size_t row = (size_t)b * L;
        const unsigned off = 4u * (unsigned)min(max(t0, 0), L - 4);
        const double* h = a.hdr + (size_t)b * (NE + 1);  // wave-uniform
        const double vote = h[0];
        const float4 xv4 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.x + row) + off);
        double pr[4] = {0.0, 0.0, 0.0, 0.0};
        if (vote != 0.0) {
            double ri[NE];
            float4 v[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                ri[e] = h[1 + e];
                v[e] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ri[e] != 0.0)  // wave-uniform: a member that adds nothing is not read
                    v[e] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.cam + e * a.cam_estride + row) + off);
            }
            float accf[4] = {0.f, 0.f, 0.f, 0.f};
            double accd[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                if (ri[e] == 0.0) continue;
                const float c[4] = {v[e].x, v[e].y, v[e].z, v[e].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float q = (float)((double)clip_cam(c[j]) * ri[e]);
                    if (F64) accd[j] += (double)q;
                    else accf[j] += q;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double avg = F64 ? accd[j] / (double)NE : (double)(float)((double)accf[j] * inv_e);
                pr[j] = inrow ? avg * vote : 0.0;
            }
        }
        // p[0..3] = the previous lane's four predictions, p[4..7] = this lane's, p[8..11] = the next lane's
        double p[12];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p[4 + j] = pr[j];
            p[j] = -LO > 3 - j ? __shfl_up(pr[j], 1, 64) : 0.0;
            p[8 + j] = HI > j ? __shfl_down(pr[j], 1, 64) : 0.0;
        }
        const float xs[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
        float st4[4], pw4[4];
        int sc4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double sum = 0.0;
#pragma unroll
            for (int m = LO; m <= HI; ++m) sum += p[4 + j + m];
            const double xv = (double)xs[j];
            const double st = attention_state(sum, W, inv_w, xv);
            st4[j] = (float)st;
            sc4[j] = state_code(st);
            const double agg = xv * a.scale_mul + a.scale_add;
            double yh = (double)(int)st * a.p_on;  // astype(int) of -1, -0, 0, 1
            if (yh > agg) yh = agg;
            pw4[j] = (float)yh;
        }
        if (owner) {
            const size_t o = row + t0;
            if (a.state8_out) *reinterpret_cast<unsigned*>(a.state8_out + o) = pack_codes(sc4);
            else *reinterpret_cast<float4*>(a.state_out + o) = make_float4(st4[0], st4[1], st4[2], st4[3]);
            if (a.power_out) *reinterpret_cast<float4*>(a.power_out + o) = make_float4(pw4[0], pw4[1], pw4[2], pw4[3]);
        }
    }
}

// Standalone attention_sigmoid_module (core.py:415-437) on an already averaged soft label:
// ma = convolve(soft, ones(w), 'same')/w ; s = 2/(1+exp(-ma*x))-1 ; round half to even.
__global__ __launch_bounds__(256) void attention_kernel(const double* __restrict__ soft, const double* __restrict__ x,
                                                        double* __restrict__ out, int L, int w) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= L) return;
    const int hi = (w - 1) / 2, lo = hi - w + 1;
    const double* sb = soft + (size_t)b * L;
    double sum = 0.0;
    for (int m = t + lo; m <= t + hi; ++m)
        if (m >= 0 && m < L) sum += sb[m];
    out[(size_t)b * L + t] = attention_state(sum, w, 1.0 / (double)w, x[(size_t)b * L + t]);
}

// ---------------------------------------------------------------------------------------
// Window builder + scaler + weak labels in front of camal_localize: the window loop of
// *_DataBuilder.get_nilm_dataset (src/helpers/data_processing.py:377-400), NILMscaler.transform on the
// aggregate channel (src/helpers/torch_dataset.py:105-111) and nilmdataset_to_clfdataset (:17-33).
//   series [n_rows][T]: row 0 aggregate power, then (power, status) per appliance
//   window i = columns i*stride .. i*stride + W - 1; a window with a NaN in ANY row is dropped (:388),
//   survivors keep their order: flags -> exclusive scan (two levels, fixed order) -> copy.
// HBM-bound: the series is read twice (flag pass, copy pass), the outputs written once.
// ---------------------------------------------------------------------------------------
#define WIN_SCAN_BLOCK 1024

struct WinArgs {
    const float* s32;      // one of s32 / s64
    const double* s64;
    long long T;
    int n_rows, W, stride, n_wins, app;   // app: appliance (1-based) whose status row gives the weak label
    double stat1, stat2;                  // x_scaled = (agg - stat1) / stat2 in float64
    unsigned char* keep;   // [n_wins]   1: no NaN
    int* pos;              // [n_wins]   exclusive scan of keep (after win_scan_*)
    int* blocksum;         // [ceil(n_wins / WIN_SCAN_BLOCK)]
    float* x_out; double* x64_out; int8_t* label_out; float* app_power_out; int8_t* app_state_out;
    int* index_out; int* count_out;
};

__device__ __forceinline__ double win_at(const WinArgs& a, int row, long long t) {
    return a.s64 ? a.s64[(size_t)row * a.T + t] : (double)a.s32[(size_t)row * a.T + t];
}

__global__ __launch_bounds__(256) void win_flag_kernel(WinArgs a) {
    const int i = blockIdx.x;
    const long long t0 = (long long)i * a.stride;
    int bad = 0;
    for (int r = 0; r < a.n_rows; ++r)
        for (int t = threadIdx.x; t < a.W; t += 256) {
            const double v = win_at(a, r, t0 + t);
            bad |= (v != v);
        }
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) a.keep[i] = bad ? 0 : 1;
}

// exclusive scan of keep[] inside blocks of WIN_SCAN_BLOCK windows (one thread per window) + the block totals
__global__ __launch_bounds__(WIN_SCAN_BLOCK) void win_scan_block_kernel(WinArgs a) {
    __shared__ int sh[WIN_SCAN_BLOCK];
    const int i = blockIdx.x * WIN_SCAN_BLOCK + threadIdx.x;
    const int v = i < a.n_wins ? a.keep[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < WIN_SCAN_BLOCK; o <<= 1) {  // Hillis-Steele, inclusive
        const int add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    if (i < a.n_wins) a.pos[i] = sh[threadIdx.x] - v;
    if (threadIdx.x == WIN_SCAN_BLOCK - 1) a.blocksum[blockIdx.x] = sh[threadIdx.x];
}

// exclusive scan of the block totals by ONE block (chunks of 1024 with a running carry) + the survivor count
__global__ __launch_bounds__(WIN_SCAN_BLOCK) void win_scan_top_kernel(WinArgs a, int nblocks) {
    __shared__ int sh[WIN_SCAN_BLOCK];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += WIN_SCAN_BLOCK) {
        const int i = base + threadIdx.x;
        const int v = i < nblocks ? a.blocksum[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < WIN_SCAN_BLOCK; o <<= 1) {
            const int add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        const int c = carry;
        if (i < nblocks) a.blocksum[i] = c + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == WIN_SCAN_BLOCK - 1) carry = c + sh[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) *a.count_out = carry;
}

__global__ __launch_bounds__(256) void win_copy_kernel(WinArgs a) {
    const int i = blockIdx.x;
    if (!a.keep[i]) return;
    const int dst = a.blocksum[i / WIN_SCAN_BLOCK] + a.pos[i];
    const long long t0 = (long long)i * a.stride;
    const size_t o = (size_t)dst * a.W;
    const int prow = 2 * a.app - 1, srow = 2 * a.app;  // rows of the appliance's power and status
    const bool has_app = srow < a.n_rows;
    int any = 0;
    for (int t = threadIdx.x; t < a.W; t += 256) {
        const double xs = (win_at(a, 0, t0 + t) - a.stat1) / a.stat2;
        a.x_out[o + t] = (float)xs;
        if (a.x64_out) a.x64_out[o + t] = xs;
        if (has_app) {
            const double st = win_at(a, srow, t0 + t);
            any |= (st != 0.0);  // data[i, 1, 1, :].any()
            if (a.app_power_out) a.app_power_out[o + t] = (float)win_at(a, prow, t0 + t);
            if (a.app_state_out) a.app_state_out[o + t] = (int8_t)(int)st;
        }
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) {
        if (a.label_out) a.label_out[dst] = any ? 1 : 0;
        if (a.index_out) a.index_out[dst] = i;
    }
}

// ---------------------------------------------------------------------------------------
// Sums behind NILMmetrics (src/helpers/other.py:112-159) and CamAL.test (core.py:201-231) in one pass
// over the flattened [N*L] arrays, so that states and powers never leave the device:
//   y, y_hat -> nan_to_num (nan, +-inf -> 0), float64:  sum|yh-y|, sum(yh-y)^2, sum|y|, sum y^2, sum y,
//   sum yh, sum min(yh,y), sum max(yh,y);  y_state, int(y_hat_state) -> TP, TN, FP, FN (labels other than
//   0/1 fall in none of the four, as in camal_amd/metrics.py).
// HBM-bound: 13 (f32 y) or 17 (f64 y) bytes per element.  Two stages, fixed order: deterministic.
// ---------------------------------------------------------------------------------------
#define NILM_NSUMS 12
#define NILM_THREADS 256
#define NILM_MAX_BLOCKS 2048

struct NilmArgs {
    const float* y32;        // true appliance power (either y32 or y64; both null: no regression sums)
    const double* y64;
    const float* yhat;       // estimated power (float32, post_kernel's power_out)
    const int8_t* ystate;    // true on/off labels (0/1), or null
    const float* yhat_state; // estimated states (post_kernel's state_out: -1, -0, 0, 1)
    size_t n;
    double* partial;         // [gridDim.x][NILM_NSUMS]
};

__device__ __forceinline__ double nan_to_num0(double v) { return (v - v == 0.0) ? v : 0.0; }  // finite ? v : 0

__global__ __launch_bounds__(NILM_THREADS) void nilm_partial_kernel(NilmArgs a) {
    __shared__ double sh[NILM_THREADS / 64][NILM_NSUMS];
    double acc[NILM_NSUMS];
#pragma unroll
    for (int k = 0; k < NILM_NSUMS; ++k) acc[k] = 0.0;
    const bool reg = (a.y32 != nullptr || a.y64 != nullptr) && a.yhat != nullptr;
    const bool cls = a.ystate != nullptr && a.yhat_state != nullptr;
    for (size_t i = (size_t)blockIdx.x * NILM_THREADS + threadIdx.x; i < a.n; i += (size_t)gridDim.x * NILM_THREADS) {
        if (reg) {
            const double y = nan_to_num0(a.y64 ? a.y64[i] : (double)a.y32[i]);
            const double yh = nan_to_num0((double)a.yhat[i]);
            const double d = yh - y;
            acc[0] += fabs(d);
            acc[1] += d * d;
            acc[2] += fabs(y);
            acc[3] += y * y;
            acc[4] += y;
            acc[5] += yh;
            acc[6] += fmin(yh, y);
            acc[7] += fmax(yh, y);
        }
        if (cls) {
            const int ys = a.ystate[i];
            const float hs = a.yhat_state[i];
            const int yh = (hs - hs == 0.f) ? (int)hs : 0;  // astype(int) of -1, -0, 0, 1 (nan_to_num first)
            acc[8] += (double)(ys == 1 && yh == 1);
            acc[9] += (double)(ys == 0 && yh == 0);
            acc[10] += (double)(ys == 0 && yh == 1);
            acc[11] += (double)(ys == 1 && yh == 0);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NILM_NSUMS; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) sh[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NILM_NSUMS) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < NILM_THREADS / 64; ++w) v += sh[w][threadIdx.x];
        a.partial[(size_t)blockIdx.x * NILM_NSUMS + threadIdx.x] = v;
    }
}

// sums[k] = sum over blocks of partial[b][k]: 64 lanes per sum over a strided subset, then a shuffle tree
__global__ __launch_bounds__(NILM_NSUMS * 64) void nilm_final_kernel(const double* __restrict__ partial, int nblocks,
                                                                     double* __restrict__ sums) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double v = 0.0;
    for (int b = lane; b < nblocks; b += 64) v += partial[(size_t)b * NILM_NSUMS + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) sums[k] = v;
}

}  // namespace camal
