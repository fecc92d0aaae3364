// camal_train_impl.hpp — host side of the training entry points (included by camal_hip.hip).
// Replaces, for one ResNet member, the per-batch body of BasedClassifTrainer.__train
// (reference src/helpers/torch_trainer.py:281-298): forward in train() mode (BatchNorm batch
// statistics), mean cross-entropy, backward, AdamW.
#pragma once
#include "camal_train_kernels.hpp"

namespace {

struct TLayer {
    int cin = 0, cout = 0, ks = 0;
    size_t w = 0, b = 0, g = 0, be = 0;  // offsets (floats) into the flat parameter / gradient buffers
    size_t rm = 0, rv = 0;               // offsets into the flat BatchNorm running-statistics buffer
    int zi = 0;                          // index of this layer's z / mean / invstd slot
};

// Flat parameter order = torch's model.parameters() order for CamALResNet (camal_resnet.py:13-19,
// 35-54): per block: 3 x (conv.w, conv.b, bn.gamma, bn.beta), then the residual's four; head last.
struct TLayout {
    TLayer conv[9], res[2];
    size_t wfc = 0, bfc = 0, n_params = 0, n_running = 0;
};

TLayout make_layout(int k, int arch = CAMAL_ARCH_CAMAL, int mid = C1) {
    const int C1 = mid, C2 = 2 * mid;  // this model's widths (shadow the defaults)
    TLayout t;
    size_t p = 0, r = 0;
    int zi = 0;
    auto add = [&](TLayer& l, int cin, int cout, int ks) {
        l.cin = cin; l.cout = cout; l.ks = ks;
        l.w = p; p += (size_t)cout * cin * ks;
        l.b = p; p += cout;
        l.g = p; p += cout;
        l.be = p; p += cout;
        l.rm = r; r += cout;
        l.rv = r; r += cout;
        l.zi = zi++;
    };
    for (int b = 0; b < 3; ++b) {
        for (int i = 0; i < 3; ++i) {
            const ConvShape c = conv_shape(3 * b + i, mid);
            add(t.conv[3 * b + i], c.cin, c.cout, conv_ks(arch, 3 * b + i, k));
        }
        if (b == 0) add(t.res[0], 1, C1, 1);
        if (b == 1) add(t.res[1], C1, C2, 1);
    }
    t.wfc = p; p += 2 * C2;
    t.bfc = p; p += 2;
    t.n_params = p;
    t.n_running = r;
    return t;
}

}  // namespace

struct camal_trainer {
    int device = 0, k = 0, maxB = 0, maxL = 0, num_cu = 256;
    float grad_scale = 1.f;         // camal_trainer_set_grad_scale
    int mid = C1;                   // mid_channels (widths 1 -> mid -> 2 mid -> 2 mid): 64, or 32 / 128
    int c1() const { return mid; }
    int c2() const { return 2 * mid; }
    int arch = CAMAL_ARCH_CAMAL;    // CAMAL_ARCH_RESNET: kernels 7/5/3 and zeros outside the window (classifiers/resnet.py)
    bool zpad() const { return arch == CAMAL_ARCH_RESNET; }
    size_t lds_per_cu = 160 * 1024;
    TLayout lay;
    std::vector<void*> owned;
    // activations: z (pre-BN conv outputs) and a (post BN/ReLU, block outputs) per layer slot
    float* z[11] = {nullptr};
    float* act[9] = {nullptr};      // a0,a1,out1, a3,a4,out2, a6,a7,out3
    float *mean = nullptr, *invstd = nullptr, *s1 = nullptr, *s2 = nullptr;  // [11][256], [256]
    double* partial = nullptr;
    float* wpack[9] = {nullptr};    // forward packing of conv 1..8 (index = conv idx), [0] unused
    float* wpack_res2 = nullptr;
    float* wflip[9] = {nullptr};    // dgrad packing of conv 1..8
    float* wflip_res2 = nullptr;
    float *gA = nullptr, *gB = nullptr, *gZ = nullptr, *gZr = nullptr, *gXP = nullptr;
    float* wg_part = nullptr;
    size_t wg_part_floats = 0;
    float* stem_part = nullptr;
    float *pooled = nullptr, *dpool = nullptr, *logits = nullptr, *dlogits = nullptr, *loss = nullptr;
    int* correct = nullptr;
    float* zeros = nullptr;         // 256 zeros: zero bias for dgrad and the zero row of its padding
};

namespace {

int t_alloc(camal_trainer* t, void** p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes));
    t->owned.push_back(*p);
    return 0;
}

inline int gs_blocks(size_t n, int cap = 4096) { return (int)std::min<size_t>((n + 255) / 256, (size_t)cap); }

// workgroups wanted per (ci tile, tap group) pair of a layer: per_cu workgroups on every CU
int wgrad_want(const camal_trainer* t, const TLayer& l) {
    const int tiles = (l.cin / 32) * ((l.ks + WG_TAPS - 1) / WG_TAPS);
    // measured (5 members, 128 x 510, same box): 2 per CU 13.65 ms, 3 per CU 13.81 (more partials to write and reduce than
    // the occupancy returns), whole windows per split (round 2) 13.72
    static const int per_cu = getenv("CAMAL_WGRAD_PER_CU") ? std::max(1, atoi(getenv("CAMAL_WGRAD_PER_CU"))) : 2;
    return std::max(1, (per_cu * t->num_cu) / std::max(1, tiles));
}

// Splits of the (window, time tile) pairs of a batch, fixed by (B, L) alone: deterministic.  -> the number of splits,
// *tps_out = tiles per split.  Never more than wgrad_want(): the partial buffer is sized for that many.
int wgrad_splits(const camal_trainer* t, const TLayer& l, int B, int L, int* tps_out) {
    const int TT = l.cout > 128 ? WgTT<256>::V : WgTT<128>::V;
    const int total = B * ((L + TT - 1) / TT);
    static const int by_window = getenv("CAMAL_WGRAD_BY_WINDOW") ? atoi(getenv("CAMAL_WGRAD_BY_WINDOW")) : 0;
    int want = std::min(wgrad_want(t, l), total);
    if (by_window) want = std::min(want, B);  // round 2's behaviour (A/B): whole windows per split
    int tps = (total + want - 1) / want;
    if (by_window) tps = (tps + (total / B) - 1) / (total / B) * (total / B);
    *tps_out = tps;
    return (total + tps - 1) / tps;
}

// ---- typed launch helpers -------------------------------------------------------------------
template <int CIN, int COUT, bool HAS_RES>
int t_conv(camal_trainer* t, hipStream_t s, const float* wpack, const float* bias, int ks, const float* in,
           const float* res, float* out, int B, int L_out, int L_in, int in_off, const float* zero_row) {
    return launch_conv_ex<float, CIN, COUT, false, HAS_RES, false>(nullptr, t->num_cu, t->lds_per_cu, 1, s, wpack, bias, ks,
                                                                   in, res, out, nullptr, nullptr, nullptr, B, L_out, L_in,
                                                                   in_off, zero_row);
}

// wpack (forward) and wflip (data gradient) of every MFMA layer, from the step's parameters
int t_pack_all(camal_trainer* t, hipStream_t s, const float* params, bool backward) {
    PackJobs jobs;
    int n = 0, biggest = 0;
    auto add = [&](const TLayer& l, float* fwd, float* flip) {
        jobs.j[n++] = PackJob{params + l.w, fwd, l.cout, l.cin, l.ks, 0};
        if (backward) jobs.j[n++] = PackJob{params + l.w, flip, /*cout'=*/l.cin, /*cin'=*/l.cout, l.ks, 1};
        biggest = std::max(biggest, l.cout * l.cin * l.ks);
    };
    for (int i = 1; i < 9; ++i) add(t->lay.conv[i], t->wpack[i], t->wflip[i]);
    add(t->lay.res[1], t->wpack_res2, t->wflip_res2);
    hipLaunchKernelGGL(pack_w_kernel, dim3(gs_blocks(biggest, 256), n), dim3(256), 0, s, jobs);
    HIP_TRY(hipGetLastError());
    return 0;
}

// (cin, cout) pairs of the MFMA layers for the supported mid_channels (32, 64, 128), forward and adjoint direction
template <typename F>
int dispatch_cc(int cin, int cout, F&& f) {
#define CAMAL_CC(A, B) if (cin == A && cout == B) return f(std::integral_constant<int, A>(), std::integral_constant<int, B>());
    CAMAL_CC(32, 32) CAMAL_CC(32, 64) CAMAL_CC(64, 32) CAMAL_CC(64, 64) CAMAL_CC(64, 128) CAMAL_CC(128, 64)
    CAMAL_CC(128, 128) CAMAL_CC(128, 256) CAMAL_CC(256, 128) CAMAL_CC(256, 256)
#undef CAMAL_CC
    return fail(CAMAL_ERR_STATE, "no kernel for this channel pair");
}

template <typename F>
int dispatch_c(int c, F&& f) {
#define CAMAL_C(A) if (c == A) return f(std::integral_constant<int, A>());
    CAMAL_C(32) CAMAL_C(64) CAMAL_C(128) CAMAL_C(256)
#undef CAMAL_C
    return fail(CAMAL_ERR_STATE, "no kernel for this channel count");
}

// forward conv of layer l (cin >= 32): z = conv(in) + bias
int t_conv_fwd(camal_trainer* t, hipStream_t s, const TLayer& l, const float* params, float* wpack, const float* in,
               float* zout, int B, int L) {
    const float* bias = params + l.b;
    const int pad = l.ks >> 1;
    const float* zr = t->zpad() ? t->zeros : nullptr;  // what the layer reads outside the window
    return dispatch_cc(l.cin, l.cout, [&](auto ci, auto co) {
        return t_conv<decltype(ci)::value, decltype(co)::value, false>(t, s, wpack, bias, l.ks, in, nullptr, zout, B, L, L, pad, zr);
    });
}

template <int C>
int t_bn_stats(camal_trainer* t, hipStream_t s, const float* z, int N, const TLayer& l, float eps, float momentum,
               float* running) {
    const int nb = (N + BN_ROWS_PER_BLOCK - 1) / BN_ROWS_PER_BLOCK;
    hipLaunchKernelGGL(bn_stats_partial_kernel<C>, dim3(nb), dim3(256), 0, s, z, N, t->partial);
    hipLaunchKernelGGL(bn_stats_final_kernel<C>, dim3(C / BN_FINAL_CPB), dim3(BN_FINAL_THREADS), 0, s, t->partial, nb, N, eps, momentum,
                       t->mean + l.zi * 256, t->invstd + l.zi * 256, running ? running + l.rm : nullptr,
                       running ? running + l.rv : nullptr);
    HIP_TRY(hipGetLastError());
    return 0;
}

int t_bn_stats_any(camal_trainer* t, hipStream_t s, const float* z, int N, const TLayer& l, float momentum, float* running) {
    return dispatch_c(l.cout, [&](auto c) { return t_bn_stats<decltype(c)::value>(t, s, z, N, l, 1e-5f, momentum, running); });
}

template <int C, bool RELU, bool HAS_RES>
int t_bn_apply(camal_trainer* t, hipStream_t s, const float* params, const TLayer& l, const float* z, const TLayer* lr,
               const float* zr, float* out, int N) {
    const size_t n4 = (size_t)N * (C / 4);
    hipLaunchKernelGGL((bn_apply_kernel<C, RELU, HAS_RES>), dim3(gs_blocks(n4)), dim3(256), 0, s, z, t->mean + l.zi * 256,
                       t->invstd + l.zi * 256, params + l.g, params + l.be, zr, lr ? t->mean + lr->zi * 256 : nullptr,
                       lr ? t->invstd + lr->zi * 256 : nullptr, lr ? params + lr->g : nullptr,
                       lr ? params + lr->be : nullptr, out, n4);
    HIP_TRY(hipGetLastError());
    return 0;
}

// BatchNorm(+ReLU) backward of layer l: da -> dz, and d gamma / d beta into grads
template <int C, bool RELU, bool BCAST>
int t_bn_bwd(camal_trainer* t, hipStream_t s, const float* params, float* grads, const TLayer& l, const float* da,
             const float* z, float* dz, int N, int L) {
    const int nb = (N + BN_ROWS_PER_BLOCK - 1) / BN_ROWS_PER_BLOCK;
    const float* mean = t->mean + l.zi * 256;
    const float* invstd = t->invstd + l.zi * 256;
    const float inv_len = 1.f / (float)L;
    hipLaunchKernelGGL((bn_bwd_partial_kernel<C, RELU, BCAST>), dim3(nb), dim3(256), 0, s, da, z, mean, invstd,
                       params + l.g, params + l.be, N, L, inv_len, t->partial);
    hipLaunchKernelGGL(bn_bwd_final_kernel<C>, dim3(C / BN_FINAL_CPB), dim3(BN_FINAL_THREADS), 0, s, t->partial, nb, t->s1, t->s2, grads + l.g,
                       grads + l.be);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<C, RELU, BCAST>), dim3(gs_blocks((size_t)N * (C / 4))), dim3(256), 0, s, da, z,
                       mean, invstd, params + l.g, params + l.be, t->s1, t->s2, N, L, inv_len, dz);
    HIP_TRY(hipGetLastError());
    return 0;
}

// weight gradient of a conv layer with cin >= 64 (MFMA), bias gradient = 0 (the bias is
// followed by a train-mode BatchNorm, which removes any per-channel constant: its exact
// gradient is zero; torch's value is rounding noise of sum(dz)).
int t_wgrad(camal_trainer* t, hipStream_t s, const TLayer& l, const float* dz, const float* ain, float* grads, int B, int L) {
    int wps = 1;  // tiles per split
    const int nsplit = wgrad_splits(t, l, B, L, &wps);
    const size_t need = (size_t)nsplit * l.ks * l.cout * l.cin;
    if (need > t->wg_part_floats) return fail(CAMAL_ERR_STATE, "wgrad partial buffer too small");
    // ceil(ks / WG_TAPS) tap groups of nearly equal size (9 = 5 + 4, 25 = 7 + 6 + 6 + 6), all in one grid:
    // the first n_hi groups take base + 1 taps, the others base
    auto launch = [&](auto cin_c, auto cout_c) {
        constexpr int CI = decltype(cin_c)::value, CO = decltype(cout_c)::value;
        const int ng = (l.ks + WG_TAPS - 1) / WG_TAPS, base = l.ks / ng, rem = l.ks % ng;
        const int nhi_taps = rem ? base + 1 : base, n_hi = rem ? rem : ng;
        constexpr int CIW = WgCIW<CI, CO>::V;
        const dim3 g = CAMAL_WGRAD_CI_FAST ? dim3(CI / 32 / CIW, ng, nsplit) : dim3(nsplit, ng, CI / 32 / CIW);
        const dim3 blk(CO * 2 * CIW);
        switch (nhi_taps) {
#define CAMAL_WG_CASE(NT) \
    case NT: hipLaunchKernelGGL((wgrad_kernel<CI, CO, NT>), g, blk, 0, s, dz, ain, t->wg_part, B, L, l.ks, wps, n_hi, t->zpad()); break;
            CAMAL_WG_CASE(1) CAMAL_WG_CASE(2) CAMAL_WG_CASE(3) CAMAL_WG_CASE(4)
            CAMAL_WG_CASE(5) CAMAL_WG_CASE(6) CAMAL_WG_CASE(7) CAMAL_WG_CASE(8)
#undef CAMAL_WG_CASE
        }
        return 0;
    };
    int rc = fail(CAMAL_ERR_STATE, "no weight-gradient kernel for this channel pair");
#define CAMAL_WG_CC(A, B) if (l.cin == A && l.cout == B) rc = launch(std::integral_constant<int, A>(), std::integral_constant<int, B>());
    CAMAL_WG_CC(32, 32) CAMAL_WG_CC(32, 64) CAMAL_WG_CC(64, 64) CAMAL_WG_CC(64, 128) CAMAL_WG_CC(128, 128)
    CAMAL_WG_CC(128, 256) CAMAL_WG_CC(256, 256)
#undef CAMAL_WG_CC
    if (rc) return rc;
    const int total = l.cout * l.cin * l.ks;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total / 4 + 63) / 64), dim3(256), 0, s, t->wg_part, nsplit, l.cout, l.cin,
                       l.ks, grads + l.w, grads + l.b);
    HIP_TRY(hipGetLastError());
    return 0;
}

int t_stem_wgrad(camal_trainer* t, hipStream_t s, const TLayer& l, const float* dz, const float* x, float* grads, int B, int L) {
    const int nseg = stem_nseg(L);
    if (int r = dispatch_c(l.cout, [&](auto c) {
            constexpr int C = decltype(c)::value;
            if constexpr (C <= 128) {
                hipLaunchKernelGGL(stem_wgrad_kernel<C>, dim3(B, nseg), dim3(256), 0, s, dz, x, t->stem_part, L, l.ks, t->zpad());
                hipLaunchKernelGGL(stem_wgrad_reduce_kernel<C>, dim3(l.ks), dim3(256), 0, s, t->stem_part, B * nseg, l.ks,
                                   grads + l.w, grads + l.b);
                return 0;
            } else {
                return fail(CAMAL_ERR_STATE, "no stem kernel for this width");
            }
        })) return r;
    HIP_TRY(hipGetLastError());
    return 0;
}

// data gradient of conv layer l (cin >= 64, ks > 1): dz [B][L][cout] -> d(input) [B][L][cin]
// = adjoint conv with zero padding on the extended range, then fold of the replicate padding.
int t_dgrad(camal_trainer* t, hipStream_t s, const TLayer& l, const float* params, float* wflip, const float* dz,
            float* dx, int B, int L) {
    const int pad = l.ks >> 1, Lp = L + 2 * pad;
    if (t->zpad()) {
        // zero padding has no fold: the adjoint is the "same" correlation with the flipped, transposed weights,
        // dz read as zeros outside the window (ks odd: ks - 1 - pad = pad)
        return dispatch_cc(l.cout, l.cin, [&](auto ci, auto co) {
            return t_conv<decltype(ci)::value, decltype(co)::value, false>(t, s, wflip, t->zeros, l.ks, dz, nullptr, dx, B, L, L, pad, t->zeros);
        });
    }
    if (int r = dispatch_cc(l.cout, l.cin, [&](auto ci, auto co) {
            return t_conv<decltype(ci)::value, decltype(co)::value, false>(t, s, wflip, t->zeros, l.ks, dz, nullptr, t->gXP, B, Lp, L, 2 * pad, t->zeros);
        })) return r;
    const size_t n4 = (size_t)B * L * (l.cin / 4);
    if (int r = dispatch_c(l.cin, [&](auto c) {
            hipLaunchKernelGGL(fold_edges_kernel<decltype(c)::value>, dim3(gs_blocks(n4)), dim3(256), 0, s, t->gXP, dx, B, L, pad);
            return 0;
        })) return r;
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

size_t camal_trainer_param_count(int kernel_size) { return make_layout(kernel_size).n_params; }
size_t camal_trainer_param_count_arch(int kernel_size, int arch) {
    return arch == CAMAL_ARCH_CAMAL || arch == CAMAL_ARCH_RESNET ? make_layout(kernel_size, arch).n_params : 0;
}
size_t camal_trainer_running_count(void) { return make_layout(7).n_running; }
size_t camal_trainer_param_count_widths(int kernel_size, int arch, int mid_channels) {
    if (!mid_supported(mid_channels) || (arch != CAMAL_ARCH_CAMAL && arch != CAMAL_ARCH_RESNET)) return 0;
    return make_layout(kernel_size, arch, mid_channels).n_params;
}
size_t camal_trainer_running_count_widths(int mid_channels) {
    return mid_supported(mid_channels) ? make_layout(7, CAMAL_ARCH_CAMAL, mid_channels).n_running : 0;
}

int camal_trainer_create(int device_id, int kernel_size, int max_batch, int max_len, camal_trainer** out) {
    return camal_trainer_create_arch(device_id, kernel_size, max_batch, max_len, CAMAL_ARCH_CAMAL, out);
}

int camal_trainer_create_arch(int device_id, int kernel_size, int max_batch, int max_len, int arch, camal_trainer** out) {
    return camal_trainer_create_widths(device_id, kernel_size, max_batch, max_len, arch, C1, out);
}

int camal_trainer_create_widths(int device_id, int kernel_size, int max_batch, int max_len, int arch, int mid_channels,
                                camal_trainer** out) {
    if (!out) return fail(CAMAL_ERR_ARG, "null argument");
    if (!mid_supported(mid_channels)) return fail(CAMAL_ERR_ARG, "mid_channels must be 32, 64 or 128");
    const int C2 = 2 * mid_channels;  // this model's widest layer (shadows the default)
    if (arch != CAMAL_ARCH_CAMAL && arch != CAMAL_ARCH_RESNET) return fail(CAMAL_ERR_ARG, "arch must be 0 (CamALResNet) or 1 (ResNet)");
    if (arch == CAMAL_ARCH_RESNET && kernel_size != 7) return fail(CAMAL_ERR_ARG, "the ResNet variant has kernel sizes 7, 5, 3: pass 7");
    if (kernel_size < 1 || kernel_size > 63 || !(kernel_size & 1)) return fail(CAMAL_ERR_ARG, "kernel size must be odd and in 1..63");
    if (max_batch < 1 || max_len < 1) return fail(CAMAL_ERR_ARG, "max_batch and max_len must be >= 1");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(CAMAL_ERR_ARG, "no such device");
    HIP_TRY(hipSetDevice(device_id));
    camal_trainer* t = new camal_trainer();
    t->device = device_id; t->k = kernel_size; t->maxB = max_batch; t->maxL = max_len; t->arch = arch;
    t->mid = mid_channels;
    t->lay = make_layout(kernel_size, arch, mid_channels);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) t->num_cu = prop.multiProcessorCount;
    const size_t N = (size_t)max_batch * max_len;
    int r = 0;
    auto A = [&](auto** p, size_t floats) { if (!r) r = t_alloc(t, (void**)p, floats * 4); };
    const TLayer* zl[11];
    for (int i = 0; i < 9; ++i) zl[t->lay.conv[i].zi] = &t->lay.conv[i];
    zl[t->lay.res[0].zi] = &t->lay.res[0];
    zl[t->lay.res[1].zi] = &t->lay.res[1];
    for (int i = 0; i < 11; ++i) A(&t->z[i], N * zl[i]->cout);
    for (int i = 0; i < 9; ++i) A(&t->act[i], N * t->lay.conv[i].cout);
    A(&t->mean, 11 * 256); A(&t->invstd, 11 * 256); A(&t->s1, 256); A(&t->s2, 256);
    if (!r) r = t_alloc(t, (void**)&t->partial, ((N + BN_ROWS_PER_BLOCK - 1) / BN_ROWS_PER_BLOCK) * 256 * 2 * sizeof(double));
    size_t wg_max = 0;
    for (int i = 1; i < 9; ++i) {
        const TLayer& l = t->lay.conv[i];
        A(&t->wpack[i], (size_t)l.cout * l.cin * l.ks);
        A(&t->wflip[i], (size_t)l.cout * l.cin * l.ks);
        wg_max = std::max(wg_max, (size_t)wgrad_want(t, l) * l.ks * l.cout * l.cin);
    }
    {
        const TLayer& l = t->lay.res[1];
        A(&t->wpack_res2, (size_t)l.cout * l.cin);
        A(&t->wflip_res2, (size_t)l.cout * l.cin);
        wg_max = std::max(wg_max, (size_t)wgrad_want(t, l) * l.cout * l.cin);
    }
    t->wg_part_floats = wg_max;
    A(&t->wg_part, wg_max);
    const int pad_max = std::max(kernel_size >> 1, 3);
    A(&t->gA, N * C2); A(&t->gB, N * C2); A(&t->gZ, N * C2); A(&t->gZr, N * C2);
    A(&t->gXP, (size_t)max_batch * (max_len + 2 * pad_max) * C2);
    A(&t->stem_part, (size_t)max_batch * stem_nseg(max_len) * kernel_size * mid_channels);
    A(&t->pooled, (size_t)max_batch * C2); A(&t->dpool, (size_t)max_batch * C2);
    A(&t->logits, (size_t)max_batch * 2); A(&t->dlogits, (size_t)max_batch * 2); A(&t->loss, (size_t)max_batch);
    if (!r) r = t_alloc(t, (void**)&t->correct, (size_t)max_batch * sizeof(int));
    A(&t->zeros, 256);
    if (!r && hipMemset(t->zeros, 0, 256 * 4) != hipSuccess) r = fail(CAMAL_ERR_HIP, "hipMemset failed");
    if (r) {
        for (void* p : t->owned) (void)hipFree(p);
        delete t;
        return r;
    }
    *out = t;
    return 0;
}

int camal_trainer_destroy(camal_trainer* t) {
    if (!t) return 0;
    (void)hipSetDevice(t->device);
    (void)hipDeviceSynchronize();
    for (void* p : t->owned) (void)hipFree(p);
    delete t;
    return 0;
}

int camal_trainer_set_grad_scale(camal_trainer* t, float scale) {
    if (!t) return fail(CAMAL_ERR_ARG, "null trainer");
    if (!(scale >= 0.f) || !(scale <= 1e6f)) return fail(CAMAL_ERR_ARG, "grad scale must be finite and >= 0");
    t->grad_scale = scale;
    return 0;
}

}  // extern "C"

namespace {

// the step for mid_channels = M (widths 1 -> M -> 2M -> 2M)
// head + loss + the whole backward pass, on the activations the forward pass left in the trainer's workspace.
// ext_dlogits: null = mean cross-entropy on labels y; else the caller's d loss / d logits [B][2] (y may then be null)
template <int M>
int train_bwd_impl(camal_trainer* t, const float* params, const float* x, const int* y, const float* ext_dlogits, int B, int L,
                   float* grads, float* logits_out, float* loss_out, int* ncorrect_out, hipStream_t s);

template <int M>
int train_fwd_bwd_impl(camal_trainer* t, const float* params, float* running, const float* x, const int* y, int B, int L,
                       float momentum, float* grads, float* logits_out, float* loss_out, int* ncorrect_out, hipStream_t s) {
    constexpr int C1 = M, C2 = 2 * M;  // shadow the defaults
    const TLayout& Y = t->lay;
    const int N = B * L;
    int r;

    if ((r = t_pack_all(t, s, params, grads != nullptr))) return r;
    // ------------------------------------ forward (train mode) ------------------------------------
    {   // block 1 stem: the two Cin = 1 convs on the VALU
        const TLayer& l0 = Y.conv[0];
        const TLayer& lr = Y.res[0];
        hipLaunchKernelGGL(stem_train_kernel<C1>, dim3((L + 63) / 64, B), dim3(256), 0, s, x, params + l0.w, params + l0.b,
                           params + lr.w, params + lr.b, t->z[l0.zi], t->z[lr.zi], L, l0.ks, t->zpad());
        HIP_TRY(hipGetLastError());
        if ((r = t_bn_stats_any(t, s, t->z[l0.zi], N, l0, momentum, running))) return r;
        if ((r = t_bn_stats_any(t, s, t->z[lr.zi], N, lr, momentum, running))) return r;
        if ((r = t_bn_apply<C1, true, false>(t, s, params, l0, t->z[l0.zi], nullptr, nullptr, t->act[0], N))) return r;
    }
    for (int b = 0; b < 3; ++b) {
        for (int i = (b == 0 ? 1 : 0); i < 3; ++i) {
            const int li = 3 * b + i;
            const TLayer& l = Y.conv[li];
            const float* in = t->act[li - 1];
            if ((r = t_conv_fwd(t, s, l, params, t->wpack[li], in, t->z[l.zi], B, L))) return r;
            if ((r = t_bn_stats_any(t, s, t->z[l.zi], N, l, momentum, running))) return r;
            if (i == 0 && b == 1) {  // block 2 skip: BN(Conv1x1(block input))
                const TLayer& lr = Y.res[1];
                if ((r = t_conv<C1, C2, false>(t, s, t->wpack_res2, params + lr.b, 1, in, nullptr, t->z[lr.zi], B, L, L, 0, nullptr))) return r;
                if ((r = t_bn_stats_any(t, s, t->z[lr.zi], N, lr, momentum, running))) return r;
            }
            if (i < 2 || b == 2) {
                if (l.cout == C1) r = t_bn_apply<C1, true, false>(t, s, params, l, t->z[l.zi], nullptr, nullptr, t->act[li], N);
                else r = t_bn_apply<C2, true, false>(t, s, params, l, t->z[l.zi], nullptr, nullptr, t->act[li], N);
            } else {  // last ConvBlock of blocks 1, 2: add the skip AFTER the ReLU (camal_resnet.py:56-58)
                const TLayer& lr = Y.res[b];
                if (l.cout == C1) r = t_bn_apply<C1, true, true>(t, s, params, l, t->z[l.zi], &lr, t->z[lr.zi], t->act[li], N);
                else r = t_bn_apply<C2, true, true>(t, s, params, l, t->z[l.zi], &lr, t->z[lr.zi], t->act[li], N);
            }
            if (r) return r;
        }
    }
    return train_bwd_impl<M>(t, params, x, y, nullptr, B, L, grads, logits_out, loss_out, ncorrect_out, s);
}

template <int M>
int train_bwd_impl(camal_trainer* t, const float* params, const float* x, const int* y, const float* ext_dlogits, int B, int L,
                   float* grads, float* logits_out, float* loss_out, int* ncorrect_out, hipStream_t s) {
    constexpr int C1 = M, C2 = 2 * M;  // shadow the defaults
    const TLayout& Y = t->lay;
    const int N = B * L;
    int r;
    // head + loss
    hipLaunchKernelGGL(head_loss_kernel<C2>, dim3(B), dim3(HEAD_LOSS_THREADS), 0, s, t->act[8], params + Y.wfc, params + Y.bfc,
                       y ? y : (const int*)t->correct, B, L, t->grad_scale, t->pooled, t->dpool, t->logits, t->dlogits, t->loss, t->correct,
                       ext_dlogits);
    HIP_TRY(hipGetLastError());
    if (logits_out) HIP_TRY(hipMemcpyAsync(logits_out, t->logits, (size_t)B * 2 * 4, hipMemcpyDeviceToDevice, s));
    if (!grads) return 0;
    hipLaunchKernelGGL(head_reduce_kernel, dim3(1), dim3(256), 0, s, t->pooled, t->dlogits, t->loss, t->correct, B, C2,
                       grads + Y.wfc, grads + Y.bfc, loss_out, ncorrect_out);
    HIP_TRY(hipGetLastError());

    // ----------------------------------------- backward -------------------------------------------
    // da = gradient w.r.t. the current block/ConvBlock output, ping-ponged between gA and gB.
    float* da = t->gA;
    float* dnext = t->gB;
    bool bcast = true;  // the first da is dpool[b][c] / L broadcast over time
    for (int b = 2; b >= 0; --b) {
        if (b < 2) {  // skip branch of blocks 1, 2: BN backward (no ReLU) on dout -> dzr
            const TLayer& lr = Y.res[b];
            if (lr.cout == C1) r = t_bn_bwd<C1, false, false>(t, s, params, grads, lr, da, t->z[lr.zi], t->gZr, N, L);
            else r = t_bn_bwd<C2, false, false>(t, s, params, grads, lr, da, t->z[lr.zi], t->gZr, N, L);
            if (r) return r;
            if (b == 1) { if ((r = t_wgrad(t, s, lr, t->gZr, t->act[2], grads, B, L))) return r; }
            else { if ((r = t_stem_wgrad(t, s, lr, t->gZr, x, grads, B, L))) return r; }
        }
        for (int i = 2; i >= 0; --i) {
            const int li = 3 * b + i;
            const TLayer& l = Y.conv[li];
            if (bcast) r = t_bn_bwd<C2, true, true>(t, s, params, grads, l, t->dpool, t->z[l.zi], t->gZ, N, L);
            else if (l.cout == C1) r = t_bn_bwd<C1, true, false>(t, s, params, grads, l, da, t->z[l.zi], t->gZ, N, L);
            else r = t_bn_bwd<C2, true, false>(t, s, params, grads, l, da, t->z[l.zi], t->gZ, N, L);
            if (r) return r;
            bcast = false;
            if (li == 0) {
                if ((r = t_stem_wgrad(t, s, l, t->gZ, x, grads, B, L))) return r;
                break;
            }
            if ((r = t_wgrad(t, s, l, t->gZ, t->act[li - 1], grads, B, L))) return r;
            if ((r = t_dgrad(t, s, l, params, t->wflip[li], t->gZ, dnext, B, L))) return r;
            if (li == 3) {  // block 2 input also feeds the 1x1 skip: d(out1) += W_r^T dzr (accumulated in place)
                const TLayer& lr = Y.res[1];
                if ((r = t_conv<C2, C1, true>(t, s, t->wflip_res2, t->zeros, 1, t->gZr, dnext, dnext, B, L, L, 0, t->zeros))) return r;
            }
            std::swap(da, dnext);
        }
    }
    return 0;
}

}  // namespace

extern "C" {

int camal_train_fwd_bwd(camal_trainer* t, const float* params, float* running, const float* x, const int* y, int B, int L,
                        float momentum, float* grads, float* logits_out, float* loss_out, int* ncorrect_out, void* stream) {
    if (!t || !params || !x) return fail(CAMAL_ERR_ARG, "null argument");
    if (B < 1 || B > t->maxB || L < 1 || L > t->maxL) return fail(CAMAL_ERR_ARG, "B or L exceeds what the trainer was created for");
    if (grads && (!y || !loss_out || !ncorrect_out)) return fail(CAMAL_ERR_ARG, "labels, loss_out and ncorrect_out are required for the backward pass");
    HIP_TRY(hipSetDevice(t->device));
    hipStream_t s = (hipStream_t)stream;
    if (t->mid == 32) return train_fwd_bwd_impl<32>(t, params, running, x, y, B, L, momentum, grads, logits_out, loss_out, ncorrect_out, s);
    if (t->mid == 128) return train_fwd_bwd_impl<128>(t, params, running, x, y, B, L, momentum, grads, logits_out, loss_out, ncorrect_out, s);
    return train_fwd_bwd_impl<C1>(t, params, running, x, y, B, L, momentum, grads, logits_out, loss_out, ncorrect_out, s);
}

int camal_train_bwd_dlogits(camal_trainer* t, const float* params, const float* x, const int* y, const float* dlogits, int B,
                            int L, float* grads, float* loss_out, int* ncorrect_out, void* stream) {
    if (!t || !params || !x || !dlogits || !grads || !loss_out || !ncorrect_out) return fail(CAMAL_ERR_ARG, "null argument");
    if (B < 1 || B > t->maxB || L < 1 || L > t->maxL) return fail(CAMAL_ERR_ARG, "B or L exceeds what the trainer was created for");
    HIP_TRY(hipSetDevice(t->device));
    hipStream_t s = (hipStream_t)stream;
    // the data-gradient packing of this step's weights (the forward-only call packed the forward direction alone)
    if (int r = t_pack_all(t, s, params, true)) return r;
    if (t->mid == 32) return train_bwd_impl<32>(t, params, x, y, dlogits, B, L, grads, nullptr, loss_out, ncorrect_out, s);
    if (t->mid == 128) return train_bwd_impl<128>(t, params, x, y, dlogits, B, L, grads, nullptr, loss_out, ncorrect_out, s);
    return train_bwd_impl<C1>(t, params, x, y, dlogits, B, L, grads, nullptr, loss_out, ncorrect_out, s);
}

int camal_trainer_peek(camal_trainer* t, int what, int index, int B, int L, float* out_dev, size_t out_floats,
                       int* channels_out, void* stream) {
    if (!t || !channels_out) return fail(CAMAL_ERR_ARG, "null argument");
    if (B < 1 || B > t->maxB || L < 1 || L > t->maxL) return fail(CAMAL_ERR_ARG, "B or L exceeds what the trainer was created for");
    const TLayer* zl[11];
    for (int i = 0; i < 9; ++i) zl[t->lay.conv[i].zi] = &t->lay.conv[i];
    zl[t->lay.res[0].zi] = &t->lay.res[0];
    zl[t->lay.res[1].zi] = &t->lay.res[1];
    const float* src = nullptr;
    size_t n = 0;
    if (what == CAMAL_PEEK_ACT) {
        if (index < 0 || index >= 9) return fail(CAMAL_ERR_ARG, "activation index must be 0..8");
        *channels_out = t->lay.conv[index].cout;
        src = t->act[index];
        n = (size_t)B * L * *channels_out;
    } else {
        if (index < 0 || index >= 11) return fail(CAMAL_ERR_ARG, "slot index must be 0..10");
        *channels_out = zl[index]->cout;
        if (what == CAMAL_PEEK_Z) { src = t->z[index]; n = (size_t)B * L * *channels_out; }
        else if (what == CAMAL_PEEK_MEAN) { src = t->mean + index * 256; n = *channels_out; }
        else if (what == CAMAL_PEEK_INVSTD) { src = t->invstd + index * 256; n = *channels_out; }
        else return fail(CAMAL_ERR_ARG, "what must be one of CAMAL_PEEK_*");
    }
    if (!out_dev) return 0;
    if (out_floats < n) return fail(CAMAL_ERR_ARG, "output buffer too small");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipMemcpyAsync(out_dev, src, n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int camal_ce_eval(const float* logits, const int* labels, int B, float* loss_out, int* pred_out, void* stream) {
    if (!logits || !labels || !loss_out || !pred_out) return fail(CAMAL_ERR_ARG, "null argument");
    if (B < 1) return fail(CAMAL_ERR_ARG, "B must be >= 1");
    hipLaunchKernelGGL(ce_eval_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, labels, B, loss_out, pred_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int camal_adamw_step(float* params, const float* grads, float* m, float* v, size_t n, int step, float lr, float beta1,
                     float beta2, float eps, float weight_decay, void* stream) {
    if (!params || !grads || !m || !v) return fail(CAMAL_ERR_ARG, "null argument");
    if (step < 1) return fail(CAMAL_ERR_ARG, "step counts from 1");
    // bias corrections in double like torch's Python arithmetic (1 - 0.999^t cancels in float for small t)
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(gs_blocks(n, 2048)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n, lr,
                       beta1, beta2, eps, weight_decay, bc1, bc2s);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
