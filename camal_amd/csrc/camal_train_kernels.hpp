// camal_train_kernels.hpp — gfx950 device code for the training step of one CamAL ResNet
// (reference: BasedClassifTrainer.__train, src/helpers/torch_trainer.py:273-300, on
// CamALResNet in train() mode, src/models/camal/classifiers/camal_resnet.py).
//
// All tensors fp32, activations [window][time][channel].  Convolutions (forward, dgrad) reuse
// conv_kernel from camal_kernels.hpp; this file adds BatchNorm-train statistics / apply /
// backward, the pooled head + cross-entropy, the MFMA weight-gradient kernel, the Cin=1 stem
// kernels and AdamW.  Every cross-workgroup reduction is two-stage through a partial buffer
// with a fixed summation order, so a step is bitwise reproducible.
#pragma once
#include "camal_kernels.hpp"

namespace camal {

// ---- weight re-layout on the device (weights change every step) -------------------------
// w [Cout][Cin][ks] -> MFMA A-fragment order of conv_kernel<float>:
//   [mt][tap][chunk pair][lane][4]  with lane (i = l&31, h = l>>5), element s
//   flip == 0 (forward):  W'[co' = 32 mt + i][ci' = 8 cp + 4 h + s][tap]        = w[co'][ci'][tap]
//   flip == 1 (dgrad): the adjoint conv has Cout' = Cin, Cin' = Cout and a reversed kernel:
//                         W'[co'][ci'][tap] = w[ci'][co'][ks - 1 - tap]
// One launch packs every MFMA layer of a member in both orientations (blockIdx.y = job): 18 launches of a few
// microseconds each were 2 % of the training step.
struct PackJob {
    const float* w;
    float* out;
    int coutp, cinp, ks, flip;
};
#define PACK_JOBS 18
struct PackJobs {
    PackJob j[PACK_JOBS];
};

__global__ void pack_w_kernel(PackJobs jobs) {
    const PackJob& job = jobs.j[blockIdx.y];
    const float* __restrict__ w = job.w;
    float* __restrict__ out = job.out;
    const int coutp = job.coutp, cinp = job.cinp, ks = job.ks, flip = job.flip;
    const int total = coutp * cinp * ks;
    const int ncp = cinp / 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int s = idx & 3, l = (idx >> 2) & 63;
        int rest = idx >> 8;
        const int cp = rest % ncp; rest /= ncp;
        const int j = rest % ks;
        const int mt = rest / ks;
        const int co = 32 * mt + (l & 31), ci = 8 * cp + 4 * (l >> 5) + s;
        out[idx] = flip ? w[((size_t)ci * coutp + co) * ks + (ks - 1 - j)] : w[((size_t)co * cinp + ci) * ks + j];
    }
}

// ---- stem, train mode: raw conv outputs of the two Cin = 1 layers --------------------------
//   z0[t][co] = b[co] + sum_j w[co][j] x[clamp(t + j - k/2)]     (ConvBlock 0 of block 1)
//   zr[t][co] = br[co] + wr[co] x[t]                              (residual conv of block 1)
template <int C>
__global__ __launch_bounds__(256) void stem_train_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ wr,
                                                         const float* __restrict__ br, float* __restrict__ z0,
                                                         float* __restrict__ zr, int L, int ks, bool zero_pad) {
    constexpr int CG = C / 4, TQ = 256 / CG, NQ = 64 / TQ;  // as stem_kernel: C = mid_channels
    __shared__ float xs[64 + 64];
    const int b = blockIdx.y, t0 = blockIdx.x * 64, pad = ks >> 1;
    const float* xb = x + (size_t)b * L;
    for (int i = threadIdx.x; i < 64 + ks - 1; i += 256) {
        const int p = t0 - pad + i;
        const int pc = p < 0 ? 0 : (p > L - 1 ? L - 1 : p);
        xs[i] = (zero_pad && pc != p) ? 0.f : xb[pc];  // zero_pad: classifiers/resnet.py:62
    }
    __syncthreads();
    const int cg = threadIdx.x % CG, tq = threadIdx.x / CG;
    float acc[NQ][4], bv[4], wrv[4], brv[4];
    load4(bias + 4 * cg, bv); load4(wr + 4 * cg, wrv); load4(br + 4 * cg, brv);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[q][c] = bv[c];
    for (int j = 0; j < ks; ++j) {
        float wv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) wv[c] = w[(4 * cg + c) * ks + j];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float xv = xs[tq + TQ * q + j];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[q][c] = fmaf(wv[c], xv, acc[q][c]);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int t = t0 + tq + TQ * q;
        if (t < L) {
            const size_t o = ((size_t)b * L + t) * C + 4 * cg;
            store4(z0 + o, acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
            const float xv = xs[tq + TQ * q + pad];
            store4(zr + o, fmaf(wrv[0], xv, brv[0]), fmaf(wrv[1], xv, brv[1]), fmaf(wrv[2], xv, brv[2]),
                   fmaf(wrv[3], xv, brv[3]));
        }
    }
}

// ---- BatchNorm1d, train mode ------------------------------------------------------------------
// Batch statistics over the N = B*L rows of z [N][C] (biased variance for the normalisation,
// unbiased for the running update, momentum 0.1 — torch.nn.BatchNorm1d as used at
// camal_resnet.py:53,84).  Stage 1: per-block partial (sum, sum of squares) in double.
#define BN_ROWS_PER_BLOCK 64
#define BN_FINAL_THREADS 1024

// Lane = 4 channels x every G-th row of the block's 64 (16-byte loads, all of a lane's rows in flight at once:
// a channel per lane with dword loads moved 3.8 TB/s).
template <int C>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ z, int N,
                                                               double* __restrict__ partial /*[grid][C][2]*/) {
    constexpr int Q = C / 4, G = 256 / Q, RPL = BN_ROWS_PER_BLOCK / G;  // quads per row, row groups, rows per lane
    static_assert(RPL * G == BN_ROWS_PER_BLOCK, "rows divide over the row groups");
    __shared__ float sh[G][C][2];
    const int c4 = (threadIdx.x % Q) * 4, g = threadIdx.x / Q;
    const int r0 = blockIdx.x * BN_ROWS_PER_BLOCK;
    float v[RPL][4];
#pragma unroll
    for (int u = 0; u < RPL; ++u) {
        const int r = r0 + g + u * G;
        v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.f;
        if (r < N) load4(z + (size_t)r * C + c4, v[u]);
    }
    float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < RPL; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[k] += v[u][k]; ss[k] = fmaf(v[u][k], v[u][k], ss[k]); }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sh[g][c4 + k][0] = s[k]; sh[g][c4 + k][1] = ss[k]; }
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x;
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int k = 0; k < G; ++k) { a += (double)sh[k][c][0]; b += (double)sh[k][c][1]; }
        partial[((size_t)blockIdx.x * C + c) * 2 + 0] = a;
        partial[((size_t)blockIdx.x * C + c) * 2 + 1] = b;
    }
}

// Stage 2, C / BN_FINAL_CPB workgroups of BN_FINAL_CPB channels each: BN_FINAL_THREADS / BN_FINAL_CPB lanes per
// channel each sum a fixed strided subset of the partials, then a fixed-order combine (xor shuffles inside the
// wave, LDS across the waves), so the result does not depend on scheduling.  One workgroup for all channels
// took 17 us per BatchNorm: 2 MB of partials through a single CU's load path; a single lane per channel
// walking every partial was 60 us of load latency.
#define BN_FINAL_CPB 8
template <int C>
__device__ __forceinline__ bool bn_final_reduce(const double* __restrict__ partial, int nblocks, int& c, double& a, double& b) {
    constexpr int G = BN_FINAL_THREADS / BN_FINAL_CPB, NW = BN_FINAL_THREADS / 64;
    static_assert(C % BN_FINAL_CPB == 0 && 64 % BN_FINAL_CPB == 0, "channels per workgroup");
    __shared__ double sh[NW][BN_FINAL_CPB][2];
    const int cl = threadIdx.x % BN_FINAL_CPB, g = threadIdx.x / BN_FINAL_CPB;
    c = blockIdx.x * BN_FINAL_CPB + cl;
    double s = 0.0, ss = 0.0;
#pragma unroll 4
    for (int i = g; i < nblocks; i += G) {
        const double2 t = *reinterpret_cast<const double2*>(partial + ((size_t)i * C + c) * 2);
        s += t.x;
        ss += t.y;
    }
#pragma unroll
    for (int m = BN_FINAL_CPB; m < 64; m <<= 1) {
        s += __shfl_xor(s, m, 64);
        ss += __shfl_xor(ss, m, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < BN_FINAL_CPB) { sh[wave][lane][0] = s; sh[wave][lane][1] = ss; }
    __syncthreads();
    a = 0.0; b = 0.0;
    if (threadIdx.x >= BN_FINAL_CPB) return false;
#pragma unroll
    for (int k = 0; k < NW; ++k) { a += sh[k][cl][0]; b += sh[k][cl][1]; }
    return true;
}

template <int C>
__global__ __launch_bounds__(BN_FINAL_THREADS) void bn_stats_final_kernel(
    const double* __restrict__ partial, int nblocks, int N, float eps, float momentum, float* __restrict__ mean,
    float* __restrict__ invstd, float* __restrict__ running_mean, float* __restrict__ running_var) {
    double s, ss;
    int c;
    if (!bn_final_reduce<C>(partial, nblocks, c, s, ss)) return;
    const double m = s / N;
    double var = ss / N - m * m;
    var = var > 0.0 ? var : 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const float unbiased = (float)(N > 1 ? var * ((double)N / (double)(N - 1)) : var);
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// a = [relu](bn(z)) [+ bn_r(zr)]   (ConvBlock output; block output with the skip added AFTER the ReLU)
template <int C, bool RELU, bool HAS_RES>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ zr,
                                                       const float* __restrict__ mean_r, const float* __restrict__ invstd_r,
                                                       const float* __restrict__ gamma_r, const float* __restrict__ beta_r,
                                                       float* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % (C / 4)) * 4;
        float v[4], m[4], is[4], g[4], bt[4], o[4];
        load4(z + i * 4, v); load4(mean + c, m); load4(invstd + c, is); load4(gamma + c, g); load4(beta + c, bt);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[k] = (v[k] - m[k]) * is[k] * g[k] + bt[k];
            if (RELU) o[k] = fmaxf(o[k], 0.f);
        }
        if constexpr (HAS_RES) {
            load4(zr + i * 4, v); load4(mean_r + c, m); load4(invstd_r + c, is); load4(gamma_r + c, g); load4(beta_r + c, bt);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] += (v[k] - m[k]) * is[k] * g[k] + bt[k];
        }
        store4(out + i * 4, o[0], o[1], o[2], o[3]);
    }
}

// BatchNorm backward, stage 1: per-channel s1 = sum dyh, s2 = sum dyh * xhat with
// dyh = da * [bn(z) > 0] (RELU) and xhat = (z - mean) * invstd.
// BCAST: da[b][t][c] = dpool[b][c] * inv_len (gradient of the temporal mean, camal_resnet.py:23).
template <int C, bool RELU, bool BCAST>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ da, const float* __restrict__ z,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int N, int L, float inv_len, double* __restrict__ partial) {
    constexpr int Q = C / 4, G = 256 / Q, RPL = BN_ROWS_PER_BLOCK / G;  // as bn_stats_partial_kernel
    __shared__ float sh[G][C][2];
    const int c4 = (threadIdx.x % Q) * 4, g = threadIdx.x / Q;
    const int r0 = blockIdx.x * BN_ROWS_PER_BLOCK;
    float m[4], is[4], ga[4], be[4];
    load4(mean + c4, m); load4(invstd + c4, is); load4(gamma + c4, ga); load4(beta + c4, be);
    float zv[RPL][4], dv[RPL][4];
#pragma unroll
    for (int u = 0; u < RPL; ++u) {
        const int r = r0 + g + u * G;
        const int rc = r < N ? r : N - 1;  // a row past the end loads a valid address and counts for nothing
        load4(z + (size_t)rc * C + c4, zv[u]);
        load4(BCAST ? da + (size_t)(rc / L) * C + c4 : da + (size_t)rc * C + c4, dv[u]);
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < RPL; ++u) {
        const bool live = r0 + g + u * G < N;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (zv[u][k] - m[k]) * is[k];
            float d = BCAST ? dv[u][k] * inv_len : dv[u][k];
            if (RELU) d = (xh * ga[k] + be[k]) > 0.f ? d : 0.f;
            d = live ? d : 0.f;
            s1[k] += d;
            s2[k] = fmaf(d, xh, s2[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sh[g][c4 + k][0] = s1[k]; sh[g][c4 + k][1] = s2[k]; }
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x;
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int k = 0; k < G; ++k) { a += (double)sh[k][c][0]; b += (double)sh[k][c][1]; }
        partial[((size_t)blockIdx.x * C + c) * 2 + 0] = a;
        partial[((size_t)blockIdx.x * C + c) * 2 + 1] = b;
    }
}

template <int C>
__global__ __launch_bounds__(BN_FINAL_THREADS) void bn_bwd_final_kernel(const double* __restrict__ partial, int nblocks,
                                                                        float* __restrict__ s1, float* __restrict__ s2,
                                                                        float* __restrict__ dgamma,
                                                                        float* __restrict__ dbeta) {
    double a, b;
    int c;
    if (!bn_final_reduce<C>(partial, nblocks, c, a, b)) return;
    s1[c] = (float)a;
    s2[c] = (float)b;
    dbeta[c] = (float)a;
    dgamma[c] = (float)b;
}

// dz = gamma * invstd * (dyh - s1/N - xhat * s2/N)
template <int C, bool RELU, bool BCAST>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ da, const float* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ s1, const float* __restrict__ s2,
                                                           int N, int L, float inv_len, float* __restrict__ dz) {
    const size_t n4 = (size_t)N * (C / 4);
    const float invN = 1.f / (float)N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % (C / 4)) * 4;
        const size_t r = i / (C / 4);
        float v[4], m[4], is[4], g[4], bt[4], d[4], a1[4], a2[4], o[4];
        load4(z + i * 4, v); load4(mean + c, m); load4(invstd + c, is); load4(gamma + c, g); load4(beta + c, bt);
        load4(s1 + c, a1); load4(s2 + c, a2);
        if (BCAST) {
            load4(da + (r / L) * C + c, d);
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] *= inv_len;
        } else {
            load4(da + i * 4, d);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (v[k] - m[k]) * is[k];
            float dy = d[k];
            if (RELU) dy = (xh * g[k] + bt[k]) > 0.f ? dy : 0.f;
            o[k] = g[k] * is[k] * (dy - a1[k] * invN - xh * a2[k] * invN);
        }
        store4(dz + i * 4, o[0], o[1], o[2], o[3]);
    }
}

// ---- pooled head + cross-entropy (camal_resnet.py:23, torch_trainer.py:288-296) -------------------
// One workgroup per window: pooled = mean_t F; logits = W pooled + b; softmax; CE; argmax;
// dlogits = grad_scale * (p - onehot) / B; dpool = W^T dlogits.  grad_scale: 1, or this rank's share b / B_global of a
// data-parallel step's batch (every gradient of the backward pass is linear in dlogits).
// ext_dlogits != null (camal_train_bwd_dlogits: a caller-supplied training criterion, torch_trainer.py:20,52,290): the
// gradient of the caller's loss with respect to the logits, [B][2], used in place of the cross-entropy's.
#define HEAD_LOSS_THREADS 1024
template <int C>  // C = 2 mid_channels: the feature width the head sees
__global__ __launch_bounds__(HEAD_LOSS_THREADS) void head_loss_kernel(const float* __restrict__ feat /*[B][L][C]*/,
                                                        const float* __restrict__ wfc, const float* __restrict__ bfc,
                                                        const int* __restrict__ labels, int B, int L, float grad_scale,
                                                        float* __restrict__ pooled /*[B][C]*/,
                                                        float* __restrict__ dpool /*[B][C]*/,
                                                        float* __restrict__ logits /*[B][2]*/,
                                                        float* __restrict__ dlogits /*[B][2]*/,
                                                        float* __restrict__ loss /*[B]*/, int* __restrict__ correct,
                                                        const float* __restrict__ ext_dlogits = nullptr) {
    __shared__ float sp[HEAD_LOSS_THREADS];
    __shared__ float sl[4];
    constexpr int TG = HEAD_LOSS_THREADS / C;  // time groups per channel
    const int b = blockIdx.x, c = threadIdx.x % C, tg = threadIdx.x / C;
    const float* f = feat + (size_t)b * L * C;
    float s = 0.f;
#pragma unroll 4
    for (int t = tg; t < L; t += TG) s += f[(size_t)t * C + c];
    sp[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < TG; ++k) a += sp[c + C * k];
        sp[c] = a / (float)L;
    }
    __syncthreads();
    if (threadIdx.x < C) pooled[(size_t)b * C + c] = sp[c];
    if (threadIdx.x < 2) {
        float a = 0.f;
        for (int k = 0; k < C; ++k) a = fmaf(wfc[threadIdx.x * C + k], sp[k], a);
        sl[threadIdx.x] = a + bfc[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l0 = sl[0], l1 = sl[1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m), den = e0 + e1;
        const float p0 = e0 / den, p1 = e1 / den;
        const int y = labels[b];
        loss[b] = -((y ? l1 : l0) - m - logf(den));
        correct[b] = ((l1 > l0) ? 1 : 0) == y;
        logits[2 * b] = l0; logits[2 * b + 1] = l1;
        const float gs = grad_scale / (float)B;
        float d0 = (p0 - (y == 0 ? 1.f : 0.f)) * gs, d1 = (p1 - (y == 1 ? 1.f : 0.f)) * gs;
        if (ext_dlogits) { d0 = ext_dlogits[2 * b] * grad_scale; d1 = ext_dlogits[2 * b + 1] * grad_scale; }
        dlogits[2 * b] = d0; dlogits[2 * b + 1] = d1;
        sl[2] = d0; sl[3] = d1;
    }
    __syncthreads();
    if (threadIdx.x < C) dpool[(size_t)b * C + c] = wfc[c] * sl[2] + wfc[C + c] * sl[3];
}

// grads of the head and the batch loss / accuracy, fixed order over the batch (one workgroup)
__global__ __launch_bounds__(256) void head_reduce_kernel(const float* __restrict__ pooled, const float* __restrict__ dlogits,
                                                          const float* __restrict__ loss, const int* __restrict__ correct,
                                                          int B, int C, float* __restrict__ dwfc /*[2][C]*/,
                                                          float* __restrict__ dbfc /*[2]*/, float* __restrict__ loss_out,
                                                          int* __restrict__ ncorrect_out) {
    for (int idx = threadIdx.x; idx < 2 * C; idx += 256) {  // C = 128: one element per thread, as before
        const int k = idx / C, c = idx - k * C;
        float a = 0.f;
        // 32 windows' loads in flight per lane (the sum keeps its order): 8 at a time was 16 dependent round trips, 33 us
#pragma unroll 32
        for (int b = 0; b < B; ++b) a = fmaf(dlogits[2 * b + k], pooled[(size_t)b * C + c], a);
        dwfc[k * C + c] = a;
    }
    // bias gradient, loss and accuracy: wave 0, lane l takes windows l, l + 64, ..., then a fixed xor tree (one lane
    // walking the batch alone was 40 us of dependent loads)
    if (threadIdx.x < 64) {
        float d0 = 0.f, d1 = 0.f;
        double s = 0.0;
        int n = 0;
        for (int b = threadIdx.x; b < B; b += 64) {
            d0 += dlogits[2 * b];
            d1 += dlogits[2 * b + 1];
            s += (double)loss[b];
            n += correct[b];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            d0 += __shfl_xor(d0, o, 64);
            d1 += __shfl_xor(d1, o, 64);
            s += __shfl_xor(s, o, 64);
            n += __shfl_xor(n, o, 64);
        }
        if (threadIdx.x == 0) {
            dbfc[0] = d0;
            dbfc[1] = d1;
            *loss_out = (float)(s / B);
            *ncorrect_out = n;
        }
    }
}

// ---- evaluation: per-window cross-entropy and predicted label from eval-mode logits (torch_trainer.py:188-193,316-321)
// loss[b] = -log_softmax(logits[b])[label[b]] in float32, pred[b] = argmax (the first maximum, like torch.max)
__global__ __launch_bounds__(256) void ce_eval_kernel(const float* __restrict__ logits, const int* __restrict__ labels,
                                                      int B, float* __restrict__ loss, int* __restrict__ pred) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float l0 = logits[2 * b], l1 = logits[2 * b + 1];
    const float m = fmaxf(l0, l1);
    const float den = expf(l0 - m) + expf(l1 - m);
    loss[b] = -((labels[b] ? l1 : l0) - m - logf(den));
    pred[b] = l1 > l0 ? 1 : 0;
}

// ---- adjoint of the replicate padding ------------------------------------------------------------
// dxp [B][L + 2 pad][C] is the gradient w.r.t. the PADDED input; positions outside [0, L) were copies
// of the edge samples, so their gradients fold onto t = 0 and t = L - 1.
template <int C>
__global__ __launch_bounds__(256) void fold_edges_kernel(const float* __restrict__ dxp, float* __restrict__ dx, int B,
                                                         int L, int pad) {
    const size_t n4 = (size_t)B * L * (C / 4);
    const int Lp = L + 2 * pad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % (C / 4)) * 4;
        const size_t r = i / (C / 4);
        const int b = (int)(r / L), t = (int)(r % L);
        const float* src = dxp + ((size_t)b * Lp) * C + c;
        float v[4], o[4];
        load4(src + (size_t)(t + pad) * C, o);
        if (t == 0)
            for (int q = 0; q < pad; ++q) { load4(src + (size_t)q * C, v); o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3]; }
        if (t == L - 1)
            for (int q = L + pad; q < Lp; ++q) { load4(src + (size_t)q * C, v); o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3]; }
        store4(dx + i * 4, o[0], o[1], o[2], o[3]);
    }
}

// ---- weight gradient on MFMA -----------------------------------------------------------------------
// dw[co][ci][j] = sum_{b,t} dz[b][t][co] * a[b][clamp(t + j - pad)][ci]   (adjoint of camal_resnet.py:82-83)
// GEMM view: M = co, N = ci, K = (b, t) for every tap j.  One workgroup = COUT/32 waves (wave = one
// 32-wide co tile) x one 32-wide ci tile x a group of <= WG_TAPS taps x a split of the batch.
// v_mfma_f32_32x32x2_f32 consumes two timesteps per instruction: lane (i, h) feeds A = dz[2s+h][co_i]
// and B = a[2s+h+j-pad][ci_i]; both tiles sit in LDS as plain rows, the tap is a row offset.
// Partial results go to part[split][j][co][ci]; wgrad_reduce_kernel sums the splits in a fixed order.
#define WG_TAPS 8
#define WG_TT 64
#ifndef CAMAL_WGRAD_CI_FAST
#define CAMAL_WGRAD_CI_FAST 0
#endif

// NTAP (taps of this workgroup's group) is a template parameter: with a run-time count every MFMA sat behind its
// own branch, LDS read and lgkmcnt(0) (the kernel ran at 0.32 of the fp32 peak).
// timesteps per staged tile: 64, or 32 for 256 out-channels (the dz tile must fit the 64 KiB of static LDS)
template <int COUT> struct WgTT { static constexpr int V = COUT > 128 ? 32 : WG_TT; };
// 32-wide ci tiles per workgroup: a workgroup is COUT / 32 waves per ci tile, and with fewer than four waves the CU's
// dispatcher stacks the workgroups on the same SIMDs (the two-wave workgroups of the 64 -> 64 layers shared SIMD 0 / 1
// whatever their number: 60 TF).  Narrow layers take 2 (64 out-channels) or 4 (32) ci tiles: four waves, one per SIMD,
// sharing one dz tile.  64 -> 64, 128 x 510: 7 taps 62 -> 38 us, 3 taps 33 -> 20 us.
template <int CIN, int COUT> struct WgCIW {
    static constexpr int WPT = COUT / 32;  // waves per ci tile
    static constexpr int V = WPT >= 4 ? 1 : ((CIN / 32) < (4 / WPT) ? (CIN / 32) : (4 / WPT));
};

// A prefetch hipcc's wait bookkeeping does not see (as the weight ring of the convolutions, camal_kernels.hpp w_load):
// left to the compiler, the next tile's loads were followed by s_waitcnt vmcnt(10) -- the first of them back -- in front
// of the MFMA loop they were meant to run under.  g_wait0 is the only wait; tools/check_ring.py watches the registers.
__device__ __forceinline__ void g_load16(u32x4& d, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void g_wait0(u32x4& d) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(d) : : "memory"); }

template <int CIN, int COUT, int NTAP>
__device__ __forceinline__ void wgrad_body(float (*s_dz)[COUT], float (*s_a)[32], const float* __restrict__ dz,
                                           const float* __restrict__ ain, float* __restrict__ part, int B, int L, int ks,
                                           int tiles_per_split, int j0, bool zero_pad) {
    constexpr int TT = WgTT<COUT>::V, CIW = WgCIW<CIN, COUT>::V, WPT = COUT / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int cot = wave % WPT, cis = wave / WPT;        // this wave's co tile and ci tile inside the workgroup
    // grid = (splits, tap groups, ci tiles) with the split fastest: workgroup ids are dealt to the 8 XCDs round robin, and
    // the ci tiles / tap groups of ONE split (they all read the same dz rows, and neighbouring 128-byte pieces of the same
    // input rows) sit splits x groups ids apart, a multiple of 8 for the usual split counts: the same XCD, one L2
    // (CAMAL_WGRAD_CI_FAST=1: ci tile fastest, the four readers of a dz tile on four XCDs)
    const int cit = CAMAL_WGRAD_CI_FAST ? blockIdx.x : blockIdx.z;
    const int cig0 = cit * CIW * 32;                     // first input channel of the workgroup
    const int ci0 = cig0 + cis * 32;
    const int arow0 = cis * (TT + WG_TAPS);              // this wave's ci tile inside s_a
    const int split = CAMAL_WGRAD_CI_FAST ? blockIdx.z : blockIdx.x, pad = ks >> 1;
    // a split = a run of consecutive (window, TT-step tile) pairs: finer than whole windows, so that layers with few
    // (ci tile, tap group) pairs (64 channels: 2) still put three workgroups on every CU at a batch of 128
    const int tpw = (L + TT - 1) / TT;
    const int tile_lo = min(B * tpw, split * tiles_per_split), tile_hi = min(B * tpw, tile_lo + tiles_per_split);

    f32x16 acc[NTAP];
#pragma unroll
    for (int j = 0; j < NTAP; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;

    // The next tile's rows travel from global memory into registers while this tile's MFMAs run (one LDS copy of
    // each tile: the waves of the CU's other workgroup cover the short register -> LDS phase).
    constexpr int NTH = COUT * 2 * CIW;
    constexpr int DZ_PER = TT * (COUT / 4) / NTH;                  // float4 of dz per lane (8 with one ci tile)
    constexpr int A_Q1 = (TT + NTAP - 1) * 8, A_Q = A_Q1 * CIW, A_PER = (A_Q + NTH - 1) / NTH;
    static_assert(DZ_PER * NTH == TT * (COUT / 4), "dz tile divides over the workgroup");
    u32x4 rdz[DZ_PER], ra[A_PER];
#pragma unroll
    for (int u = 0; u < DZ_PER; ++u) rdz[u] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < A_PER; ++u) ra[u] = u32x4{0u, 0u, 0u, 0u};
    // fetch only ISSUES loads, always from rows inside the window, from one call site per tile: what lies outside the
    // window becomes zero in stash, a tile later.  With the zeroing here (selects on the loaded values) and one fetch
    // call per branch of "next tile of this window / first tile of the next", hipcc joined the two definitions through
    // register copies behind an s_waitcnt vmcnt(0): every wave sat out the HBM round trip of its prefetch in front of
    // its MFMA loop (tools/stamps_wgrad.py: 10.6 k of a tile's 32.8 k cycles).
    auto fetch = [&](int b, int t0) {
        // dz rows t0 .. t0+TT-1, all COUT channels
#pragma unroll
        for (int u = 0; u < DZ_PER; ++u) {
            const int q = threadIdx.x + u * NTH;
            const int row = q / (COUT / 4), c4 = (q % (COUT / 4)) * 4;
            const int t = min(t0 + row, L - 1);
            g_load16(rdz[u], dz + ((size_t)b * L + t) * COUT + c4);
        }
        // input rows t0 + j0 - pad .. (+ TT + ntap - 2), clamped = replicate padding, 32 channels
#pragma unroll
        for (int u = 0; u < A_PER; ++u) {
            const int q = min((int)threadIdx.x + u * NTH, A_Q - 1);
            const int sub = q / A_Q1, w = q - sub * A_Q1;  // ci tile, chunk inside it
            const int p = t0 + j0 - pad + (w >> 3);
            const int pc = p < 0 ? 0 : (p > L - 1 ? L - 1 : p);
            g_load16(ra[u], ain + ((size_t)b * L + pc) * CIN + cig0 + sub * 32 + (w & 7) * 4);
        }
    };
    auto landed = [&]() {  // every load of the last fetch has arrived
#pragma unroll
        for (int u = 0; u < DZ_PER; ++u) g_wait0(rdz[u]);
#pragma unroll
        for (int u = 0; u < A_PER; ++u) g_wait0(ra[u]);
    };
    // the tile fetched for t0: dz rows past L are zeros; zero_pad: so are the input rows outside the window
    auto stash = [&](int t0) {
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < DZ_PER; ++u) {
            const int q = threadIdx.x + u * NTH;
            const int row = q / (COUT / 4);
            *reinterpret_cast<float4*>(&s_dz[row][(q % (COUT / 4)) * 4]) = t0 + row < L ? __builtin_bit_cast(float4, rdz[u]) : zero4;
        }
#pragma unroll
        for (int u = 0; u < A_PER; ++u) {
            const int q = threadIdx.x + u * NTH;
            if (q < A_Q) {
                const int sub = q / A_Q1, w = q - sub * A_Q1;
                const int p = t0 + j0 - pad + (w >> 3);
                *reinterpret_cast<float4*>(&s_a[sub * (TT + WG_TAPS) + (w >> 3)][(w & 7) * 4]) =
                    (zero_pad && (p < 0 || p > L - 1)) ? zero4 : __builtin_bit_cast(float4, ra[u]);
            }
        }
    };
#if defined(CAMAL_STAMPS) && CAMAL_STAMPS == 4
#define WSTAMP(k) do { if (CIN == 128 && COUT == 128 && NTAP == 7 && tile == 2 && cit == 0 && split < 256 && lane == 0) \
    g_stamps[split][wave][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP(k) do {} while (0)
#endif
    const int ntiles = tile_hi - tile_lo;
    int b = tile_lo / tpw, t0 = (tile_lo - b * tpw) * TT;
    if (ntiles > 0) {
        fetch(b, t0);
        landed();  // before the loop: whatever hipcc does to join these registers with the loop's happens on landed data
    }
    for (int tile = 0; tile < ntiles; ++tile) {
        WSTAMP(0);
        __syncthreads();
        WSTAMP(1);
        landed();
        stash(t0);
        WSTAMP(2);
        __syncthreads();
        WSTAMP(3);
        // the next tile of this window, else the first one of the next window (after the last tile: that one again, unused)
        int nb = b, nt0 = t0 + TT;
        if (nt0 >= L) { nt0 = 0; ++nb; }
        if (tile + 1 >= ntiles) { nb = b; nt0 = t0; }
        fetch(nb, nt0);
        WSTAMP(4);
#pragma unroll 16
        for (int s = 0; s < TT / 2; ++s) {
            const float av = s_dz[2 * s + h][32 * cot + i];
#pragma unroll
            for (int j = 0; j < NTAP; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s_a[arow0 + 2 * s + h + j][i], acc[j], 0, 0, 0);
        }
        WSTAMP(5);
        b = nb;
        t0 = nt0;
    }
    landed();  // unconditionally: the last, unused prefetch is still in flight and nothing may reuse its registers before it lands
    // acc[j][reg]: column ci = ci0 + i, row co = 32 cot + (reg&3) + 8 (reg>>2) + 4 h
#pragma unroll
    for (int j = 0; j < NTAP; ++j) {
            float* po = part + (((size_t)split * ks + (j0 + j)) * COUT + 32 * cot) * CIN + ci0 + i;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) po[(size_t)((reg & 3) + 8 * (reg >> 2) + 4 * h) * CIN] = acc[j][reg];
        }
}

// One launch for all tap groups of a layer: groups blockIdx.y < n_hi take NHI taps, the others NHI - 1 (a launch per
// group size kept two workgroups per CU; one grid keeps three resident: +0.5 % on the step).  PMC: the MFMA pipe is
// 0.69 busy on a single 7- or 8-tap group (512 workgroups), 0.83 on k = 25's four groups.  Groups of at most 4 taps
// to get there on every layer: 3-5 % slower, a dz tile staged per 4 x 32 MFMAs costs more than the occupancy gains.
template <int CIN, int COUT, int NHI>
__global__ __launch_bounds__((COUT * 2 * WgCIW<CIN, COUT>::V)) void wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ ain,
                                                         float* __restrict__ part, int B, int L, int ks,
                                                         int tiles_per_split, int n_hi, bool zero_pad) {
    __shared__ float s_dz[WgTT<COUT>::V][COUT];
    __shared__ float s_a[WgCIW<CIN, COUT>::V * (WgTT<COUT>::V + WG_TAPS)][32];
    const int y = blockIdx.y;
    if (y < n_hi) {
        wgrad_body<CIN, COUT, NHI>(s_dz, s_a, dz, ain, part, B, L, ks, tiles_per_split, y * NHI, zero_pad);
    } else {
        if constexpr (NHI > 1)
            wgrad_body<CIN, COUT, NHI - 1>(s_dz, s_a, dz, ain, part, B, L, ks, tiles_per_split,
                                           n_hi * NHI + (y - n_hi) * (NHI - 1), zero_pad);
    }
}

// grad[co][ci][j] = sum_split part[split][j][co][ci]; bias_grad[0..cout) = 0 (see t_wgrad).  Four ci per lane and
// four lanes per quad, each walking every fourth split with independent 16-byte loads, combined in a fixed order:
// one lane per element walking all splits moved the partials at 1.9 TB/s.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, int cout, int cin,
                                                           int ks, float* __restrict__ grad, float* __restrict__ bias_grad) {
    __shared__ float4 red[4][64];
    const int total = cout * cin * ks;
    const int q = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int idx = (blockIdx.x * 64 + q) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < total) {
#pragma unroll 8
        for (int s = g; s < nsplit; s += 4) {
            const float4 v = *reinterpret_cast<const float4*>(part + (size_t)s * total + idx);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    red[g][q] = a;
    __syncthreads();
    if (g == 0 && idx < total) {
        const float4 b = red[1][q], c = red[2][q], d = red[3][q];
        const int ci = idx % cin, co = (idx / cin) % cout, j = idx / (cin * cout);
        float* o = grad + ((size_t)co * cin + ci) * ks + j;
        o[0] = (a.x + b.x) + (c.x + d.x);
        o[ks] = (a.y + b.y) + (c.y + d.y);
        o[2 * ks] = (a.z + b.z) + (c.z + d.z);
        o[3 * ks] = (a.w + b.w) + (c.w + d.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < cout) bias_grad[threadIdx.x] = 0.f;
}

// Cin = 1 layers: part[b * nseg + seg][j][co] = sum over the segment's t of dz[b][t][co] * x[b][clamp(t + j - pad)]
// (64 out-channels).  One workgroup per (window, segment of STEM_SEG_LEN steps): a window per workgroup left half
// the CUs idle at the batch sizes trained on, and the aggregate was re-read from global memory once per tap.
#define STEM_SEG_LEN 128
__host__ __device__ inline int stem_nseg(int L) { return (L + STEM_SEG_LEN - 1) / STEM_SEG_LEN; }

template <int C>  // C = mid_channels out-channels; 256 / C phases walk the segment's timesteps
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                         float* __restrict__ part, int L, int ks, bool zero_pad) {
    constexpr int PH = 256 / C;
    __shared__ float red[PH][C];
    __shared__ float xs[STEM_SEG_LEN + 64 + 8];
    const int b = blockIdx.x, seg = blockIdx.y, co = threadIdx.x % C, ph = threadIdx.x / C, pad = ks >> 1;
    const int t0 = seg * STEM_SEG_LEN, t1 = min(L, t0 + STEM_SEG_LEN);
    const float* xb = x + (size_t)b * L;
    const float* dzb = dz + (size_t)b * L * C;
    // xs[r] = x[clamp(t0 - pad + r)]; the rows past the last tap only meet accumulators that are never stored
    for (int r = threadIdx.x; r < STEM_SEG_LEN + 64 + 8; r += 256) {
        const int p = t0 - pad + r;
        const int pc = p < 0 ? 0 : (p > L - 1 ? L - 1 : p);
        xs[r] = (zero_pad && pc != p) ? 0.f : xb[pc];
    }
    __syncthreads();
    float* po = part + ((size_t)b * gridDim.y + seg) * ks * C;
    for (int j0 = 0; j0 < ks; j0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll 4
        for (int t = t0 + ph; t < t1; t += PH) {
            const float d = dzb[(size_t)t * C + co];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(d, xs[t - t0 + j0 + j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __syncthreads();
            red[ph][co] = acc[j];
            __syncthreads();
            if (ph == 0 && j0 + j < ks) {
                float v;
                if constexpr (PH == 4) v = (red[0][co] + red[1][co]) + (red[2][co] + red[3][co]);  // the order the 64-channel kernel always used
                else {
                    v = 0.f;
#pragma unroll
                    for (int q = 0; q < PH; ++q) v += red[q][co];
                }
                po[(j0 + j) * C + co] = v;
            }
        }
    }
}

// grad[co][j] = sum_p part[p][j][co], one workgroup per tap: 256 / C lanes per channel walk the partials in strides
// (independent loads), combined in a fixed order
template <int C>
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, int ks,
                                                                float* __restrict__ grad, float* __restrict__ bias_grad) {
    constexpr int PH = 256 / C;
    __shared__ float red[PH][C];
    const int j = blockIdx.x, co = threadIdx.x % C, g = threadIdx.x / C;
    float a = 0.f;
#pragma unroll 8
    for (int p = g; p < nparts; p += PH) a += part[((size_t)p * ks + j) * C + co];
    red[g][co] = a;
    __syncthreads();
    if (g == 0) {
        float v;
        if constexpr (PH == 4) v = (red[0][co] + red[1][co]) + (red[2][co] + red[3][co]);
        else {
            v = 0.f;
#pragma unroll
            for (int q = 0; q < PH; ++q) v += red[q][co];
        }
        grad[co * ks + j] = v;
    }
    if (g == PH - 1 && j == 0) bias_grad[co] = 0.f;
}

// ---- AdamW (torch.optim.AdamW defaults: torch_trainer.py:35; decoupled weight decay) -------------------
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, size_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, float bias_c1, float bias_c2_sqrt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float pi = p[i] * (1.f - lr * weight_decay);
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bias_c2_sqrt + eps;
        p[i] = pi - (lr / bias_c1) * (mi / denom);
    }
}

}  // namespace camal
