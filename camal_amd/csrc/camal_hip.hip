// camal_hip.hip — host side of libcamal_hip.so: context, weight re-layout, launch sequences.
// C ABI declared in include/camal_hip.h.  gfx950 only.
#include "camal_kernels.hpp"
#include "../../include/camal_hip.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

using namespace camal;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail(CAMAL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));   \
    } while (0)

constexpr int C1 = 64, C2 = 128;  // core.py:37 mid_channels default; CamAL.eval ignores kwargs (:386-390)

// mid_channels the library has kernels for (camal_resnet.py:8,14-16: widths 1 -> mid -> 2 mid -> 2 mid); 64 is what
// CamAL.eval always builds, 32 and 128 serve CamALResNet(mid_channels=...) on its own and CamAL.train's resnet_kwargs
bool mid_supported(int mid) { return mid == 32 || mid == 64 || mid == 128; }

// The 9 ConvBlocks of one member: (Cin, Cout, k) with k<0 meaning "the member's kernel size".
struct ConvShape { int cin, cout, k; };
ConvShape conv_shape(int i, int mid = C1) {
    const int c1 = mid, c2 = 2 * mid;
    const ConvShape t[9] = {{1, c1, -1},  {c1, c1, 7},  {c1, c1, 3},  {c1, c2, -1}, {c2, c2, 7},
                            {c2, c2, 3},  {c2, c2, -1}, {c2, c2, 7},  {c2, c2, 3}};
    return t[i];
}

// Kernel size of ConvBlock i.  CAMAL_ARCH_CAMAL: (k, 7, 3) per block, replicate padding (camal_resnet.py:31,83);
// CAMAL_ARCH_RESNET: (7, 5, 3) per block, zero "same" padding (classifiers/resnet.py:31,62).
int conv_ks(int arch, int i, int k) {
    if (arch == CAMAL_ARCH_RESNET) return i % 3 == 0 ? 7 : (i % 3 == 1 ? 5 : 3);
    return conv_shape(i).k < 0 ? k : conv_shape(i).k;
}

size_t blob_floats(int k, int arch = CAMAL_ARCH_CAMAL, int mid = C1) {
    const int c1 = mid, c2 = 2 * mid;
    size_t n = 0;
    for (int i = 0; i < 9; ++i) {
        const ConvShape c = conv_shape(i, mid);
        int kk = conv_ks(arch, i, k);
        n += (size_t)c.cout * c.cin * kk + c.cout;
    }
    n += c1 * 1 + c1;      // residual 1
    n += (size_t)c2 * c1 + c2;  // residual 2
    n += 2 * c2 + 2;       // head
    return n;
}

// Tile configuration (see conv_kernel).  bf16: 4 compute waves x 4 column tiles, two workgroups per
// CU.  fp32: 4 compute waves x 2 column tiles with two workgroups per CU whenever two double-buffered
// tiles fit the 160 KiB of LDS (independent workgroups drift apart, so one's epilogue overlaps the
// other's MFMAs), else 8 compute waves in one workgroup per CU.
// fp32 128 -> 64 (only the dgrad of block 2's first conv): one column tile per wave.
template <typename DT, int CIN, int COUT> struct TileNT { static constexpr int NT = sizeof(DT) == 4 ? 2 : 4; };
template <> struct TileNT<float, 128, 64> { static constexpr int NT = 1; };
template <> struct TileNT<float, 256, 128> { static constexpr int NT = 1; };  // dgrad of mid_channels = 128's block 2
template <typename DT, int CIN, int COUT, int NWC>
using GeomW = ConvGeom<DT, CIN, COUT, TileNT<DT, CIN, COUT>::NT, NWC>;

// fp32: two 4-wave workgroups per CU if their tiles fit, else one 8-wave workgroup
template <typename DT, int CIN, int COUT, bool HEAD>
bool use_small_wg(size_t lds_per_cu, int ks) {
    if (sizeof(DT) != 4) return true;
    if (COUT / 32 > 4) return false;  // 256 out-channels: eight channel tiles need the eight compute waves
    using G = GeomW<DT, CIN, COUT, (COUT / 32 > 4 ? 8 : 4)>;
    return 2 * (2 * (size_t)G::buf_bytes(ks) + (HEAD ? G::HEAD_BYTES : 0)) <= lds_per_cu;
}

struct Layer {
    int cin = 0, cout = 0, ks = 0;
    void* wpack = nullptr;   // [MT][ks][NCP][64][16 B]
    float* bias = nullptr;   // [cout]
    std::vector<char> host_pack;  // bf16, default widths: kept until the member's block streams are built
};

struct Member {
    int k = 0;
    bool loaded = false;
    float *stem_w = nullptr, *stem_b = nullptr, *res1_w = nullptr, *res1_b = nullptr;
    void* stem_pack = nullptr;  // bf16, default widths: the stem's weights as hi / lo MFMA A fragments (Block1Args::stem_pack)
    Layer conv[8];  // ConvBlocks 1..8 (ConvBlock 0 is the stem)
    Layer res2;
    void *stream2 = nullptr, *stream3 = nullptr;  // bf16 fused blocks 2 / 3: A fragments of all layers, one stream per out-channel tile
    int stride2 = 0, stride3 = 0;                 // bytes between the streams of consecutive out-channel tiles
    float* wfc = nullptr;  // [2][128]
    float* bfc = nullptr;  // [2]
};

struct ProfRec { int cls; hipEvent_t a, b; };

}  // namespace

struct camal_ctx {
    int device = 0, E = 0, precision = 0, chunk = 0 /* 0: auto_chunk(L) */, cam_row = 1;
    int post_windows = 4096;  // windows per post-processing launch (their CAM rows wait in the workspace)
    int num_cu = 256;
    size_t lds_per_cu = 160 * 1024;
    int arch = CAMAL_ARCH_CAMAL;
    int mid = C1;  // mid_channels (camal_resnet.py:8): 64, or 32 / 128 through camal_create_widths (one kernel per ConvBlock, fp32 or bf16)
    float* zero_row = nullptr;  // 1 KiB of zeros: what a zero-padded layer reads outside the window
    int fused = 1;  // blocks 2 and 3 as fused-block kernels (block_kernel); 0: one conv_kernel per layer
    int tap_block = 2;     // camal_forward_cam_tap: 1 = the CAM is taken from block 2's output (into tap_cam)
    float* tap_cam = nullptr;
    int stem_mfma = 1;     // bf16 block1_kernel: the Cin = 1 ConvBlock on MFMA (hi + lo bf16 operands); env CAMAL_STEM_MFMA=0: VALU
    int fused_block1 = 1;  // with fused: block 1 as block1_kernel (env CAMAL_FUSED_BLOCK1=0: stem + two conv_kernels)
    std::vector<Member> m;
    std::vector<void*> owned;  // device allocations freed in destroy
    // workspace
    char* ws = nullptr;
    size_t ws_bytes = 0;
    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;
    int prof_launches[CAMAL_PROF_CLASSES] = {0};
    double prof_ms[CAMAL_PROF_CLASSES] = {0};
};

namespace {

int dev_alloc(camal_ctx* c, void** p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes));
    c->owned.push_back(*p);
    return 0;
}

int upload(camal_ctx* c, void** dst, const void* src, size_t bytes) {
    if (int r = dev_alloc(c, dst, bytes)) return r;
    HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}

inline uint16_t f2bf(float f) {  // round to nearest even, NaN stays NaN
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16);
}

// w [cout][cin][ks] fp32 -> A-fragment order, one coalesced 1 KiB block per fragment:
// fp32 (32x32x2):   [mt][tap][chunk pair cp][lane][16 B]; lane (i = l&31, h = l>>5), element s
//                   <-> W[32 mt + i][8 cp + 4 h + s][tap]
// bf16 (16x16x32):  [mt][tap][group gk of 32 channels][mi][lane][16 B]; lane (i = l&15, q = l>>4), element jj
//                   <-> W[32 mt + 16 mi + i][32 gk + 8 kblock(q) + jj][tap]   (kblock: Frag<__bf16>, LDS banking)
std::vector<char> pack_weights(const float* w, int cout, int cin, int ks, int precision) {
    const int mtn = cout / 32;
    std::vector<char> out((size_t)mtn * ks * (cin / 8) * (precision == CAMAL_FP32 ? 1024 : 512));
    char* dst = out.data();
    for (int mt = 0; mt < mtn; ++mt)
        for (int j = 0; j < ks; ++j) {
            if (precision == CAMAL_FP32) {
                for (int cp = 0; cp < cin / 8; ++cp)
                    for (int l = 0; l < 64; ++l, dst += 16) {
                        const int co = 32 * mt + (l & 31), h = l >> 5;
                        float v[4];
                        for (int s = 0; s < 4; ++s) v[s] = w[((size_t)co * cin + 8 * cp + 4 * h + s) * ks + j];
                        memcpy(dst, v, 16);
                    }
            } else {
                for (int gk = 0; gk < cin / 32; ++gk)
                    for (int mi = 0; mi < 2; ++mi)
                        for (int l = 0; l < 64; ++l, dst += 16) {
                            const int co = 32 * mt + 16 * mi + (l & 15), q = l >> 4;
                            const int kb = ((q & 1) << 1) | (q >> 1);  // Frag<__bf16>::kblock
                            uint16_t v[8];
                            for (int jj = 0; jj < 8; ++jj) v[jj] = f2bf(w[((size_t)co * cin + 32 * gk + 8 * kb + jj) * ks + j]);
                            memcpy(dst, v, 16);
                        }
            }
        }
    return out;
}

int make_layer(camal_ctx* c, Layer& ly, const float* w, const float* b, int cout, int cin, int ks) {
    ly.cin = cin; ly.cout = cout; ly.ks = ks;
    std::vector<char> p = pack_weights(w, cout, cin, ks, c->precision);
    const size_t packed = p.size();
    p.resize(packed + 8192, 0);  // the bf16 weight ring runs 3 groups (6 KiB) past the last out-channel tile's end
    if (int r = upload(c, &ly.wpack, p.data(), p.size())) return r;
    if (c->precision == CAMAL_BF16 && cout == 128 && c->mid == C1) {  // block streams exist for the default widths
        p.resize(packed);
        ly.host_pack.swap(p);
    }
    return upload(c, (void**)&ly.bias, b, cout * sizeof(float));
}

// bf16 fused block: per out-channel tile mt (4 of them) the packed A fragments of the block's layers back to back,
// then the first 3 groups (2 KiB each) of the first layer again: what block_kernel's weight ring (WRing) walks
// without ever restarting.  Frees the layers' host copies.
int make_block_stream(camal_ctx* c, std::vector<Layer*> ls, void** out, int* stride) {
    size_t per_mt = 3 * 2048;
    for (Layer* l : ls) {
        if (l->cout != 128 || l->host_pack.size() != (size_t)4 * l->ks * (l->cin / 32) * 2048)
            return fail(CAMAL_ERR_STATE, "block stream: unexpected layer shape");
        per_mt += l->host_pack.size() / 4;
    }
    std::vector<char> s(4 * per_mt);
    for (int mt = 0; mt < 4; ++mt) {
        char* d = s.data() + mt * per_mt;
        for (Layer* l : ls) {
            const size_t chunk = l->host_pack.size() / 4;
            memcpy(d, l->host_pack.data() + mt * chunk, chunk);
            d += chunk;
        }
        memcpy(d, ls[0]->host_pack.data() + mt * (ls[0]->host_pack.size() / 4), 3 * 2048);
    }
    *stride = (int)per_mt;
    if (int r = upload(c, out, s.data(), s.size())) return r;
    for (Layer* l : ls) std::vector<char>().swap(l->host_pack);
    return 0;
}

// ---- workspace ------------------------------------------------------------------------
struct Workspace {
    void *a1, *b1, *r1, *a2, *b2, *r2;
    float *cam, *part, *logits, *cmax;
    double* hdr;
    int ntiles;
};

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

#ifndef CAMAL_BF16_BLOCK_WAVES
#define CAMAL_BF16_BLOCK_WAVES 4
#endif
#ifndef CAMAL_BF16_BLOCK1_WAVES
#define CAMAL_BF16_BLOCK1_WAVES 8
#endif
template <typename DT> struct BlockNT;
template <> struct BlockNT<float> { static constexpr int NT = 2; static constexpr int NWC = 8; };
template <> struct BlockNT<__bf16> { static constexpr int NWC = CAMAL_BF16_BLOCK_WAVES; static constexpr int NT = 32 / NWC; };

// wave = 32 out-channels x NT column tiles; fp32: NWC = 8, two column groups of 64; bf16: NWC = 4, one wave per SIMD
// over all 256 columns (BlockNT)
template <typename DT, int CIN, int NWC>
using BGeom = BlockGeom<DT, CIN, BlockNT<DT>::NT, NWC, 1>;

// LDS bytes of one fused-block workgroup
template <typename DT, int CIN, int NWC, bool HAS_RES, bool HEAD>
size_t block_lds(int k) {
    using G = BGeom<DT, CIN, NWC>;
    return 2 * (size_t)G::buf_bytes(k, HAS_RES) + (HEAD ? G::HEAD_BYTES : 0) + G::CST_FLOATS * sizeof(float);
}

// One workgroup per CU: the compute waves of BlockNT<DT> + the loader wave; 256-column tiles in bf16 (4 waves of
// 32 channels x 256 columns: two waves of 128 columns per SIMD never shared the matrix pipe evenly and loaded every
// weight fragment twice, -1.6 %), 128 in fp32 (8 waves).  0: the tile does not fit the LDS for this first kernel size.
// (Two 4-wave workgroups of half the tile width per CU, so that one's epilogues and barriers run under the
// other's MFMAs: measured on the 16x16x32 kernels, 10 % slower.)
template <typename DT, int CIN, bool HAS_RES, bool HEAD>
int block_nwc(size_t lds_per_cu, int k) {
    return block_lds<DT, CIN, BlockNT<DT>::NWC, HAS_RES, HEAD>(k) > lds_per_cu ? 0 : BlockNT<DT>::NWC;
}

// can this member's blocks 2/3 run fused (LDS budget depends on the first kernel size)?
bool fused_fits(const camal_ctx* c, int k) {
    if (c->mid != C1) return false;  // the fused-block kernels exist for the default widths
    if (c->precision == CAMAL_FP32)
        return block_nwc<float, C2, false, true>(c->lds_per_cu, k) && block_nwc<float, C1, true, false>(c->lds_per_cu, k);
    return block_nwc<__bf16, C2, false, true>(c->lds_per_cu, k) && block_nwc<__bf16, C1, true, false>(c->lds_per_cu, k);
}

template <typename DT>
int fused_head_tile(size_t lds_per_cu, int k) {
    return BGeom<DT, C2, BlockNT<DT>::NWC>::T;
}

// the per-layer CAM head is the k = 3 ConvBlock on CH = 2 mid channels: its tile configuration follows use_small_wg
template <int CH, typename DT = float>
int layer_head_tile(size_t lds_per_cu) {
    if constexpr (sizeof(DT) == 2) return CH / 32 > 4 ? ConvGeom<DT, CH, CH, 4, 8>::TN : GeomW<DT, CH, CH, 4>::TN;  // as launch_conv_ex picks
    return use_small_wg<float, CH, CH, true>(lds_per_cu, 3) ? GeomW<float, CH, CH, 4>::TN : GeomW<float, CH, CH, 8>::TN;
}

// timesteps per CAM-head tile (decides the layout of the pooled partial sums)
int head_tile_len(const camal_ctx* c, bool fused, int k) {
    if (c->mid == 32) return c->precision == CAMAL_FP32 ? layer_head_tile<64>(c->lds_per_cu) : layer_head_tile<64, __bf16>(c->lds_per_cu);
    if (c->mid == 128) return c->precision == CAMAL_FP32 ? layer_head_tile<256>(c->lds_per_cu) : layer_head_tile<256, __bf16>(c->lds_per_cu);
    if (c->precision == CAMAL_FP32)
        return fused ? fused_head_tile<float>(c->lds_per_cu, k) : layer_head_tile<C2>(c->lds_per_cu);
    return fused ? fused_head_tile<__bf16>(c->lds_per_cu, k) : GeomW<__bf16, C2, C2, 4>::TN;
}

// upper bound over every path (workspace sizing): the smallest head tile is 56 timesteps
int head_tiles(const camal_ctx*, int L) { return (L + 55) / 56; }

// BC: windows per conv pass (activations); PB >= BC: windows per post-processing launch (CAM rows, logits)
size_t workspace_layout(const camal_ctx* c, int BC, int PB, int L, Workspace* w, char* base) {
    const size_t es = c->precision == CAMAL_FP32 ? 4 : 2;
    const size_t n64 = align_up((size_t)BC * L * c->mid * es), n128 = align_up((size_t)BC * L * 2 * c->mid * es);
    const int ntiles = head_tiles(c, L);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += bytes; return (void*)p; };
    Workspace t;
    t.a1 = take(n64); t.b1 = take(n64); t.r1 = take(n64);
    t.a2 = take(n128); t.b2 = take(n128); t.r2 = take(n128);
    t.cam = (float*)take(align_up((size_t)c->E * PB * L * 4));
    t.part = (float*)take(align_up((size_t)c->E * PB * ntiles * 2 * 4));  // [E][PB][ntiles][2]
    t.hdr = (double*)take(align_up((size_t)PB * (c->E + 1) * 8));
    t.logits = (float*)take(align_up((size_t)c->E * PB * 2 * 4));
    t.cmax = (float*)take(align_up((size_t)c->E * PB * 4));
    t.ntiles = ntiles;
    if (w) *w = t;
    return off;
}

int ensure_workspace(camal_ctx* c, int BC, int PB, int L, Workspace* w) {
    size_t need = workspace_layout(c, BC, PB, L, nullptr, nullptr);
    if (need > c->ws_bytes) {
        if (c->ws) HIP_TRY(hipFree(c->ws));
        c->ws = nullptr; c->ws_bytes = 0;
        hipError_t e = hipMalloc((void**)&c->ws, need);
        if (e != hipSuccess) return fail(CAMAL_ERR_NOMEM, "workspace hipMalloc of " + std::to_string(need) + " bytes failed");
        c->ws_bytes = need;
    }
    workspace_layout(c, BC, PB, L, w, c->ws);
    return 0;
}

// ---- profiling ------------------------------------------------------------------------
// Two forms.  ProfScope(ctx, stream, cls): an event pair recorded on the stream around whatever the scope launches.
// ProfScope(ctx, stream, cls, kOneKernel) + launch(...): the scope's ONE kernel carries the pair itself
// (hipExtLaunchKernel's start / stop events = the dispatch's own begin and end stamps, what rocprofv3 reports); an
// event pair recorded around a 13-25 us kernel also times the hand-offs between the stream's packets (about 3 us).
struct OneKernel {};
constexpr OneKernel kOneKernel{};

struct ProfScope {
    camal_ctx* c; hipStream_t s; ProfRec rec; bool on; bool one = false, launched = false;
    void take(int cls) {
        while (c->pool.size() < c->pool_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
            c->pool.push_back(e);
        }
        rec.cls = cls; rec.a = c->pool[c->pool_used++]; rec.b = c->pool[c->pool_used++];
    }
    ProfScope(camal_ctx* ctx, hipStream_t st, int cls) : c(ctx), s(st), on(ctx && ctx->prof) {
        if (!on) return;
        take(cls);
        if (on) (void)hipEventRecord(rec.a, s);
    }
    ProfScope(camal_ctx* ctx, hipStream_t st, int cls, OneKernel) : c(ctx), s(st), on(ctx && ctx->prof), one(true) {
        if (on) take(cls);
    }
    template <typename... P, typename... A, size_t... I>
    void launch_ev(void (*kern)(P...), dim3 grid, dim3 block, size_t lds, std::index_sequence<I...>, A&&... args) {
        std::tuple<P...> params{static_cast<P>(args)...};  // the kernel's own parameter types
        void* ptrs[] = {(void*)&std::get<I>(params)...};
        (void)hipExtLaunchKernel((const void*)kern, grid, block, ptrs, lds, s, rec.a, rec.b, 0);
    }
    // the scope's one launch (kOneKernel form); errors surface through hipGetLastError like hipLaunchKernelGGL's
    template <typename... P, typename... A>
    void launch(void (*kern)(P...), dim3 grid, dim3 block, size_t lds, A&&... args) {
        static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
        if (on && one && !launched) {
            launched = true;
            launch_ev(kern, grid, block, lds, std::index_sequence_for<P...>{}, std::forward<A>(args)...);
        } else {
            hipLaunchKernelGGL(kern, grid, block, lds, s, static_cast<P>(args)...);
        }
    }
    ~ProfScope() {
        if (!on) return;
        if (one) {
            if (launched) c->recs.push_back(rec);
            else c->pool_used -= 2;  // nothing was launched: the pair goes back
            return;
        }
        (void)hipEventRecord(rec.b, s);
        c->recs.push_back(rec);
    }
};

// ---- launches -------------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: every kernel instantiation keeps its
// high-water mark per device (table of kAttrDevices entries, one per instantiation; raised under one mutex, read without it)
constexpr int kAttrDevices = 64;
std::mutex g_attr_mu;
int ensure_lds_attr(const void* kern, size_t lds, size_t* table) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kAttrDevices) return fail(CAMAL_ERR_ARG, "device index beyond the attribute table");
    // fast path without the lock: the mark only ever grows, and it is published after the attribute is set
    if (lds <= __atomic_load_n(&table[dev], __ATOMIC_ACQUIRE)) return 0;
    std::lock_guard<std::mutex> lock(g_attr_mu);
    if (lds > table[dev]) {
        HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        __atomic_store_n(&table[dev], lds, __ATOMIC_RELEASE);
    }
    return 0;
}

// L_out: output rows per window; input rows per window L_in, input row for (t, tap j) = t + j - in_off;
// zero_row == nullptr: out-of-range rows replicate the edge (forward), else they read zeros (dgrad).
// NTX != 0 overrides the column tiles per wave (narrower time tile for first-kernel sizes whose halo does not fit)
template <typename DT, int CIN, int COUT, int NWC, bool RELU, bool HAS_RES, bool HEAD, int NTX = 0>
int launch_conv_w(camal_ctx* c, int num_cu, size_t lds_per_cu, int cam_row, hipStream_t s, const void* wpack,
                  const float* bias, int ks, const void* in, const void* res, void* out, const float* wfc, float* cam,
                  float* part, int BC, int L_out, int L_in, int in_off, const void* zero_row, float* cmax = nullptr) {
    using G = ConvGeom<DT, CIN, COUT, NTX ? NTX : TileNT<DT, CIN, COUT>::NT, NWC>;
    auto kern = conv_kernel<DT, CIN, COUT, G::TN / (32 * G::NG), NWC, RELU, HAS_RES, HEAD>;
    const size_t lds = 2 * (size_t)G::buf_bytes(ks) + (HEAD ? G::HEAD_BYTES : 0);
    if (lds > lds_per_cu) return fail(CAMAL_ERR_ARG, "kernel size too large for the LDS tile");
    static size_t attr_set[kAttrDevices] = {0};  // per instantiation and device
    if (int r = ensure_lds_attr((const void*)kern, lds, attr_set)) return r;
    const int tpw = (L_out + G::TN - 1) / G::TN;
    const long long total = (long long)tpw * BC;
    if (total > 0x7fffffffLL) return fail(CAMAL_ERR_ARG, "too many tiles in one pass");
    static const int per_cu_cap = getenv("CAMAL_CONV_PER_CU") ? std::max(1, atoi(getenv("CAMAL_CONV_PER_CU"))) : 4;
    const int per_cu = std::max(1, std::min(per_cu_cap, (int)(lds_per_cu / lds)));
    const int grid = (int)std::min<long long>(total, (long long)num_cu * per_cu);
    ProfScope ps(c, s, HEAD ? 2 : 1, kOneKernel);
    ps.launch(kern, dim3(grid), dim3(G::THREADS), lds, (const DT*)in, (const uint4*)wpack, bias,
              (const DT*)res, (DT*)out, wfc, cam, part, L_out, ks, cam_row, tpw, (int)total, L_in, in_off,
              (const DT*)zero_row, cmax);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename DT, int CIN, int COUT, bool RELU, bool HAS_RES, bool HEAD>
int launch_conv_ex(camal_ctx* c, int num_cu, size_t lds_per_cu, int cam_row, hipStream_t s, const void* wpack,
                   const float* bias, int ks, const void* in, const void* res, void* out, const float* wfc, float* cam,
                   float* part, int BC, int L_out, int L_in, int in_off, const void* zero_row, float* cmax = nullptr) {
    if constexpr (COUT / 32 <= 4) {
        if (use_small_wg<DT, CIN, COUT, HEAD>(lds_per_cu, ks))
            return launch_conv_w<DT, CIN, COUT, 4, RELU, HAS_RES, HEAD>(c, num_cu, lds_per_cu, cam_row, s, wpack, bias, ks, in, res,
                                                                        out, wfc, cam, part, BC, L_out, L_in, in_off, zero_row, cmax);
    }
    if constexpr (sizeof(DT) == 2 && COUT / 32 > 4) {
        // bf16, 256 out-channels (mid_channels = 128): eight channel tiles = eight compute waves, 128-column tiles; large
        // first kernels on 256 input channels take 64- or 32-column tiles (LDS)
        auto fits = [&](auto g) { using G = decltype(g); return 2 * (size_t)G::buf_bytes(ks) + (HEAD ? G::HEAD_BYTES : 0) <= lds_per_cu; };
        if (fits(ConvGeom<DT, CIN, COUT, 4, 8>()))
            return launch_conv_w<DT, CIN, COUT, 8, RELU, HAS_RES, HEAD, 4>(c, num_cu, lds_per_cu, cam_row, s, wpack, bias, ks, in, res,
                                                                           out, wfc, cam, part, BC, L_out, L_in, in_off, zero_row, cmax);
        if constexpr (!HEAD) {
            if (fits(ConvGeom<DT, CIN, COUT, 2, 8>()))
                return launch_conv_w<DT, CIN, COUT, 8, RELU, HAS_RES, HEAD, 2>(c, num_cu, lds_per_cu, cam_row, s, wpack, bias, ks, in,
                                                                               res, out, wfc, cam, part, BC, L_out, L_in, in_off, zero_row, cmax);
            return launch_conv_w<DT, CIN, COUT, 8, RELU, HAS_RES, HEAD, 1>(c, num_cu, lds_per_cu, cam_row, s, wpack, bias, ks, in,
                                                                           res, out, wfc, cam, part, BC, L_out, L_in, in_off, zero_row, cmax);
        }
    }
    if constexpr (sizeof(DT) == 4) {
        using G8 = GeomW<DT, CIN, COUT, 8>;
        const bool fits8 = 2 * (size_t)G8::buf_bytes(ks) + (HEAD ? G8::HEAD_BYTES : 0) <= lds_per_cu;
        // one column tile per wave already and the eight-wave tile does not fit (256 -> 128 channels, the data gradient of
        // mid_channels = 128's block 2, first kernel sizes from 19): ONE four-wave workgroup per CU on half the columns
        if constexpr (COUT / 32 <= 4 && !HEAD && TileNT<DT, CIN, COUT>::NT == 1) {
            using G4 = GeomW<DT, CIN, COUT, 4>;
            if (!fits8 && 2 * (size_t)G4::buf_bytes(ks) <= lds_per_cu)
                return launch_conv_w<DT, CIN, COUT, 4, RELU, HAS_RES, HEAD>(c, num_cu, lds_per_cu, cam_row, s, wpack, bias, ks, in,
                                                                            res, out, wfc, cam, part, BC, L_out, L_in, in_off, zero_row, cmax);
        }
        if (fits8 || HEAD || TileNT<DT, CIN, COUT>::NT == 1)
            return launch_conv_w<DT, CIN, COUT, 8, RELU, HAS_RES, HEAD>(c, num_cu, lds_per_cu, cam_row, s, wpack, bias, ks, in,
                                                                        res, out, wfc, cam, part, BC, L_out, L_in, in_off, zero_row, cmax);
        // kernel sizes past 33 on 128 channels: one column tile per wave (64-step tiles)
        if constexpr (!HEAD && TileNT<DT, CIN, COUT>::NT != 1)
            return launch_conv_w<DT, CIN, COUT, 8, RELU, HAS_RES, HEAD, 1>(c, num_cu, lds_per_cu, cam_row, s, wpack, bias, ks, in,
                                                                           res, out, wfc, cam, part, BC, L_out, L_in, in_off, zero_row, cmax);
    }
    return fail(CAMAL_ERR_ARG, "kernel size too large for the LDS tile");
}

template <typename DT, int CIN, int COUT, bool RELU, bool HAS_RES, bool HEAD>
int launch_conv(camal_ctx* c, hipStream_t s, const Layer& ly, const void* in, const void* res, void* out,
                const float* wfc, float* cam, float* part, int BC, int L, float* cmax = nullptr) {
    return launch_conv_ex<DT, CIN, COUT, RELU, HAS_RES, HEAD>(c, c->num_cu, c->lds_per_cu, c->cam_row, s, ly.wpack,
                                                              ly.bias, ly.ks, in, res, out, wfc, cam, part, BC, L, L,
                                                              ly.ks >> 1, c->arch == CAMAL_ARCH_RESNET ? c->zero_row : nullptr,
                                                              cmax);
}

// VARIANT: classifiers/resnet.py's block (second kernel 5, zero padding) instead of CamALResNet's (7, replicate)
template <typename DT, int CIN, int NWC, bool HAS_RES, bool HEAD, bool VARIANT = false>
int launch_block_w(camal_ctx* c, hipStream_t s, const Layer& l1, const Layer& l2, const Layer& l3, const Layer* lr,
                   const void* wstream, int wstride, const void* in, void* out, const float* wfc, float* cam, float* part, int BC, int L, float* cmax) {
    constexpr int NTB = BlockNT<DT>::NT;
    using G = BGeom<DT, CIN, NWC>;
    auto kern = block_kernel<DT, CIN, NTB, NWC, 1, HAS_RES, HEAD, VARIANT ? 5 : 7, VARIANT>;
    const size_t lds = block_lds<DT, CIN, NWC, HAS_RES, HEAD>(l1.ks);
    static size_t attr_set[kAttrDevices] = {0};
    if (int r = ensure_lds_attr((const void*)kern, lds, attr_set)) return r;
    BlockArgs a;
    a.in = in; a.out = out;
    a.w1 = (const uint4*)l1.wpack; a.w2 = (const uint4*)l2.wpack; a.w3 = (const uint4*)l3.wpack;
    a.wr = lr ? (const uint4*)lr->wpack : nullptr;
    a.b1 = l1.bias; a.b2 = l2.bias; a.b3 = l3.bias; a.br = lr ? lr->bias : nullptr;
    a.wfc = wfc; a.cam = cam; a.part = part; a.cmax = cmax;
    a.L = L; a.ks1 = l1.ks; a.cam_row = c->cam_row;
    a.zero_row = c->zero_row;
    a.wstream = (const char*)wstream; a.wstride = wstride;
    if (sizeof(DT) == 2 && (wstream == nullptr || (l1.ks & 1) == 0)) return fail(CAMAL_ERR_STATE, "fused block: no weight stream (odd first kernel sizes only)");
    if (l2.ks != (VARIANT ? 5 : 7) || l3.ks != 3) return fail(CAMAL_ERR_STATE, "fused block: unexpected layer kernel sizes");
    a.tiles_per_window = (L + G::T - 1) / G::T;
    const long long total = (long long)a.tiles_per_window * BC;
    if (total > 0x7fffffffLL) return fail(CAMAL_ERR_ARG, "too many tiles in one pass");
    a.total_tiles = (int)total;
    const int per_cu = std::max(1, (int)(c->lds_per_cu / lds));
    const int grid = (int)std::min<long long>(total, (long long)c->num_cu * per_cu);
    ProfScope ps(c, s, HEAD ? 2 : 1, kOneKernel);
    ps.launch(kern, dim3(grid), dim3(G::THREADS), lds, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename DT, int CIN, bool HAS_RES, bool HEAD>
int launch_block(camal_ctx* c, hipStream_t s, const Layer& l1, const Layer& l2, const Layer& l3, const Layer* lr,
                 const void* wstream, int wstride, const void* in, void* out, const float* wfc, float* cam, float* part, int BC, int L,
                 float* cmax = nullptr) {
    const int nwc = block_nwc<DT, CIN, HAS_RES, HEAD>(c->lds_per_cu, l1.ks);
    constexpr int NWC = BlockNT<DT>::NWC;
    if (nwc == NWC && c->arch == CAMAL_ARCH_RESNET)
        return launch_block_w<DT, CIN, NWC, HAS_RES, HEAD, true>(c, s, l1, l2, l3, lr, wstream, wstride, in, out, wfc, cam, part, BC, L, cmax);
    if (nwc == NWC) return launch_block_w<DT, CIN, NWC, HAS_RES, HEAD>(c, s, l1, l2, l3, lr, wstream, wstride, in, out, wfc, cam, part, BC, L, cmax);
    return fail(CAMAL_ERR_ARG, "kernel size too large for the fused block tile");
}

// fused block 1: x [BC][L] -> block output [BC][L][64]
template <typename DT, int NWC, bool VARIANT = false>
int launch_block1_w(camal_ctx* c, hipStream_t s, const Member& m, const float* x, void* out, int BC, int L) {
    constexpr int NTB = sizeof(DT) == 2 ? 32 / NWC : 2;  // column tiles per wave (bf16: 512-column tiles either way)
    using G = Block1Geom<DT, NTB, NWC>;
    auto kern = block1_kernel<DT, NTB, NWC, VARIANT ? 5 : 7, VARIANT>;
    if (m.conv[0].ks != (VARIANT ? 5 : 7) || m.conv[1].ks != 3) return fail(CAMAL_ERR_STATE, "fused block 1: unexpected layer kernel sizes");
    static size_t attr_set[kAttrDevices] = {0};
    if (int r = ensure_lds_attr((const void*)kern, (size_t)G::LDS_BYTES, attr_set)) return r;
    Block1Args a;
    a.x = x; a.out = out;
    a.w2 = (const uint4*)m.conv[0].wpack; a.w3 = (const uint4*)m.conv[1].wpack;
    a.b2 = m.conv[0].bias; a.b3 = m.conv[1].bias;
    a.L = L; a.ks1 = m.k;
    a.stem_pack = c->stem_mfma ? (const uint4*)m.stem_pack : nullptr;
    a.tiles_per_window = (L + G::T - 1) / G::T;
    const long long total = (long long)a.tiles_per_window * BC;
    if (total > 0x7fffffffLL) return fail(CAMAL_ERR_ARG, "too many tiles in one pass");
    a.total_tiles = (int)total;
    const int per_cu = std::max(1, (int)(c->lds_per_cu / G::LDS_BYTES));
    const int grid = (int)std::min<long long>(total, (long long)c->num_cu * per_cu);
    ProfScope ps(c, s, 1, kOneKernel);
    ps.launch(kern, dim3(grid), dim3(G::THREADS), (size_t)G::LDS_BYTES, a, m.stem_w, m.stem_b, m.res1_w, m.res1_b);
    HIP_TRY(hipGetLastError());
    return 0;
}

// 8 waves, one workgroup per CU.  (Two 4-wave workgroups of half the tile width per CU, so that one's VALU stem
// runs under the other's MFMAs: measured, no difference.)
template <typename DT>
int launch_block1(camal_ctx* c, hipStream_t s, const Member& m, const float* x, void* out, int BC, int L) {
    constexpr int NWC = sizeof(DT) == 2 ? CAMAL_BF16_BLOCK1_WAVES : 8;
    if (c->arch == CAMAL_ARCH_RESNET) return launch_block1_w<DT, NWC, true>(c, s, m, x, out, BC, L);
    return launch_block1_w<DT, NWC>(c, s, m, x, out, BC, L);
}

// camal_forward_cam_tap with tap_block = 1: the CAM of block 2's output (in `feat`, [BC * L][CH]) into c->tap_cam
template <typename DT, int CH>
int launch_cam_tap(camal_ctx* c, hipStream_t s, const Member& m, const void* feat, int BC, int L) {
    if (c->tap_block != 1 || c->tap_cam == nullptr) return 0;
    ProfScope ps(c, s, 4);
    const size_t rows = (size_t)BC * L;
    const int grid = (int)std::min<size_t>((rows + 255) / 256, 4096);
    hipLaunchKernelGGL((cam_dot_kernel<DT, CH>), dim3(grid), dim3(256), 0, s, (const DT*)feat, m.wfc + (size_t)c->cam_row * CH,
                       c->tap_cam, rows);
    HIP_TRY(hipGetLastError());
    return 0;
}

// One member's conv stack on BC windows, one kernel per ConvBlock, for mid_channels = M (widths 1 -> M -> 2M -> 2M):
// x -> cam row [BC][L] + pooled partials [BC][ntiles][2].  block1_done: block 1 already ran fused (its output in w.a1)
template <typename DT, int M>
int run_member_layers(camal_ctx* c, hipStream_t s, const Member& m, const float* x, int BC, int L, const Workspace& w,
                      float* cam, float* part, float* cmax, bool block1_done) {
    constexpr int W1 = M, W2 = 2 * M;
    int r;
    if (!block1_done) {
        {
            ProfScope ps(c, s, 0);
            hipLaunchKernelGGL((stem_kernel<DT, W1>), dim3((L + 63) / 64, BC), dim3(256), 0, s, x, m.stem_w, m.stem_b, m.res1_w,
                               m.res1_b, (DT*)w.a1, (DT*)w.r1, L, m.k, c->arch == CAMAL_ARCH_RESNET);
            HIP_TRY(hipGetLastError());
        }
        // block 1: ConvBlock(7), ConvBlock(3) + residual (camal_resnet.py:35-38,56-58)
        if ((r = launch_conv<DT, W1, W1, true, false, false>(c, s, m.conv[0], w.a1, nullptr, w.b1, nullptr, nullptr, nullptr, BC, L))) return r;
        if ((r = launch_conv<DT, W1, W1, true, true, false>(c, s, m.conv[1], w.b1, w.r1, w.a1, nullptr, nullptr, nullptr, BC, L))) return r;
    }
    // block 2: residual BN(Conv1x1) first (no ReLU), then ConvBlock(k), (7), (3)+residual
    if ((r = launch_conv<DT, W1, W2, false, false, false>(c, s, m.res2, w.a1, nullptr, w.r2, nullptr, nullptr, nullptr, BC, L))) return r;
    if ((r = launch_conv<DT, W1, W2, true, false, false>(c, s, m.conv[2], w.a1, nullptr, w.a2, nullptr, nullptr, nullptr, BC, L))) return r;
    if ((r = launch_conv<DT, W2, W2, true, false, false>(c, s, m.conv[3], w.a2, nullptr, w.b2, nullptr, nullptr, nullptr, BC, L))) return r;
    if ((r = launch_conv<DT, W2, W2, true, true, false>(c, s, m.conv[4], w.b2, w.r2, w.a2, nullptr, nullptr, nullptr, BC, L))) return r;
    if ((r = launch_cam_tap<DT, W2>(c, s, m, w.a2, BC, L))) return r;
    // block 3: no skip (camal_resnet.py:40-41,59); last ConvBlock fused with the CAM head
    if ((r = launch_conv<DT, W2, W2, true, false, false>(c, s, m.conv[5], w.a2, nullptr, w.b2, nullptr, nullptr, nullptr, BC, L))) return r;
    if ((r = launch_conv<DT, W2, W2, true, false, false>(c, s, m.conv[6], w.b2, nullptr, w.a2, nullptr, nullptr, nullptr, BC, L))) return r;
    if ((r = launch_conv<DT, W2, W2, true, false, true>(c, s, m.conv[7], w.a2, nullptr, nullptr, m.wfc, cam, part, BC, L, cmax))) return r;
    return 0;
}

// One member's conv stack on BC windows at the default widths: fused-block kernels where they apply
template <typename DT>
int run_member(camal_ctx* c, hipStream_t s, const Member& m, const float* x, int BC, int L, const Workspace& w,
               float* cam, float* part, float* cmax) {
    int r;
    const bool fused = c->fused && fused_fits(c, m.k);
    const bool b1 = fused && c->fused_block1;
    if (b1) {
        // block 1 in one launch: x in, 64-channel block output out
        if ((r = launch_block1<DT>(c, s, m, x, w.a1, BC, L))) return r;
    }
    if (fused) {
        if (!b1) {
            {
                ProfScope ps(c, s, 0);
                hipLaunchKernelGGL((stem_kernel<DT, C1>), dim3((L + 63) / 64, BC), dim3(256), 0, s, x, m.stem_w, m.stem_b, m.res1_w,
                                   m.res1_b, (DT*)w.a1, (DT*)w.r1, L, m.k, c->arch == CAMAL_ARCH_RESNET);
                HIP_TRY(hipGetLastError());
            }
            if ((r = launch_conv<DT, C1, C1, true, false, false>(c, s, m.conv[0], w.a1, nullptr, w.b1, nullptr, nullptr, nullptr, BC, L))) return r;
            if ((r = launch_conv<DT, C1, C1, true, true, false>(c, s, m.conv[1], w.b1, w.r1, w.a1, nullptr, nullptr, nullptr, BC, L))) return r;
        }
        // blocks 2 and 3: one fused kernel each (three ConvBlocks through LDS)
        if ((r = launch_block<DT, C1, true, false>(c, s, m.conv[2], m.conv[3], m.conv[4], &m.res2, m.stream2, m.stride2, w.a1, w.a2, nullptr, nullptr, nullptr, BC, L))) return r;
        if ((r = launch_cam_tap<DT, C2>(c, s, m, w.a2, BC, L))) return r;
        if ((r = launch_block<DT, C2, false, true>(c, s, m.conv[5], m.conv[6], m.conv[7], nullptr, m.stream3, m.stride3, w.a2, nullptr, m.wfc, cam, part, BC, L, cmax))) return r;
        return 0;
    }
    return run_member_layers<DT, C1>(c, s, m, x, BC, L, w, cam, part, cmax, false);
}

// pooled-sum tiles per window of member m's CAM head
int member_ntiles(const camal_ctx* c, const Member& m, int L) {
    const int tl = head_tile_len(c, c->fused && fused_fits(c, m.k), m.k);
    return (L + tl - 1) / tl;
}

// logits == nullptr: the caller finalizes the pooled sums itself (camal_localize: header_kernel)
int run_member_any(camal_ctx* c, hipStream_t s, const Member& m, const float* x, int BC, int L, const Workspace& w,
                   float* cam, float* part, float* logits, float* cmax) {
    int r;
    if (c->mid == 32) r = c->precision == CAMAL_FP32 ? run_member_layers<float, 32>(c, s, m, x, BC, L, w, cam, part, cmax, false)
                                                      : run_member_layers<__bf16, 32>(c, s, m, x, BC, L, w, cam, part, cmax, false);
    else if (c->mid == 128) r = c->precision == CAMAL_FP32 ? run_member_layers<float, 128>(c, s, m, x, BC, L, w, cam, part, cmax, false)
                                                            : run_member_layers<__bf16, 128>(c, s, m, x, BC, L, w, cam, part, cmax, false);
    else r = c->precision == CAMAL_FP32 ? run_member<float>(c, s, m, x, BC, L, w, cam, part, cmax)
                                        : run_member<__bf16>(c, s, m, x, BC, L, w, cam, part, cmax);
    if (r || !logits) return r;
    ProfScope ps(c, s, 4);
    const int tl = head_tile_len(c, c->fused && fused_fits(c, m.k), m.k);
    hipLaunchKernelGGL(logits_kernel, dim3((BC + 127) / 128), dim3(128), 0, s, part, m.bfc, logits, BC, (L + tl - 1) / tl, L);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Windows per conv pass when the caller has not set one: about 2 M timesteps (measured optimum on 1440- and
// 10080-step windows: enough tiles to amortise launch and tail, activations of a pass still mostly on-die
// between the block kernels), at most 1024 windows.
int auto_chunk(const camal_ctx* c, int L) {
    if (c->chunk > 0) return c->chunk;
    return (int)std::max<long long>(16, std::min<long long>(1024, (1LL << 21) / std::max(1, L)));
}

int check_ctx(camal_ctx* c, int B, int L) {
    if (!c) return fail(CAMAL_ERR_ARG, "null context");
    if (B < 1 || L < 1) return fail(CAMAL_ERR_ARG, "B and L must be >= 1");
    for (int e = 0; e < c->E; ++e)
        if (!c->m[e].loaded) return fail(CAMAL_ERR_STATE, "member " + std::to_string(e) + " has no weights");
    HIP_TRY(hipSetDevice(c->device));
    return 0;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
// the state output a lane writes 4 timesteps of at once: 16 bytes of float32, or 4 bytes of int8 codes
bool state_aligned(const PostArgs& a) { return a.state8_out ? ((uintptr_t)a.state8_out & 3) == 0 : aligned16(a.state_out); }

// post_stream_kernel for E members (1..8) and a 5-step moving average (the reference's default, core.py:257)
template <bool F64>
bool launch_post_stream(ProfScope& ps, const PostArgs& a, int blocks) {
    const int segs = (a.L + POST_SEG - 1) / POST_SEG;
    const long long total = (long long)segs * a.B;
    if (total > 0x7fffffffLL) return false;
    const int grid = (int)std::min<long long>((total + 3) / 4, blocks);
    switch (a.E) {
#define CAMAL_PS_CASE(NE) case NE: ps.launch(post_stream_kernel<NE, F64, 5>, dim3(grid), dim3(256), (size_t)0, a, segs, (int)total); return true;
        CAMAL_PS_CASE(1) CAMAL_PS_CASE(2) CAMAL_PS_CASE(3) CAMAL_PS_CASE(4)
        CAMAL_PS_CASE(5) CAMAL_PS_CASE(6) CAMAL_PS_CASE(7) CAMAL_PS_CASE(8)
#undef CAMAL_PS_CASE
    }
    return false;
}

// own_max: no header_kernel ran; the kernel derives the votes and scans the rows for their maxima
int launch_post(camal_ctx* c, hipStream_t s, const PostArgs& a0, bool own_max) {
    // the common case goes to the streaming kernel (wave-per-segment, no LDS): same arithmetic, bit-identical
    static const int stream_mode = getenv("CAMAL_POST_STREAM") ? atoi(getenv("CAMAL_POST_STREAM")) : 1;
    static const int stream_blocks = getenv("CAMAL_POST_STREAM_BLOCKS") ? atoi(getenv("CAMAL_POST_STREAM_BLOCKS")) : 0;
    if (stream_mode && !own_max && a0.hdr && !a0.x64 && !a0.avg_out && a0.w_ma == 5 && a0.L % 4 == 0 && a0.E <= 8 &&
        aligned16(a0.cam) && a0.cam_estride % 4 == 0 && aligned16(a0.x) && state_aligned(a0) && aligned16(a0.power_out)) {
        const int blocks = stream_blocks > 0 ? stream_blocks : (c ? c->num_cu : 256) * 16;  // measured flat from 12 to 64 per CU
        ProfScope ps1(c, s, 3, kOneKernel);
        const bool ok = a0.pred_f64 ? launch_post_stream<true>(ps1, a0, blocks) : launch_post_stream<false>(ps1, a0, blocks);
        if (ok) {
            HIP_TRY(hipGetLastError());
            return 0;
        }
    }
    ProfScope ps(c, s, 3);
    const int tiles = (a0.L + POST_TILE - 1) / POST_TILE;
    constexpr int kMaxY = 32768;  // grid.y limit is 65535
    // 16-byte accesses when every row of every array starts on a 16-byte boundary
    const bool vec = a0.L % 4 == 0 && aligned16(a0.cam) && a0.cam_estride % 4 == 0 && aligned16(a0.x) && state_aligned(a0) &&
                     aligned16(a0.avg_out) && aligned16(a0.power_out);
    for (int b0 = 0; b0 < a0.B; b0 += kMaxY) {
        PostArgs a = a0;
        const size_t o = (size_t)b0 * a0.L;
        a.B = std::min(kMaxY, a0.B - b0);
        a.cam += o; if (a.logits) a.logits += 2 * (size_t)b0; if (a.hdr) a.hdr += (size_t)b0 * (a0.E + 1);
        a.x += o; if (a.x64) a.x64 += o; if (a.y) a.y += b0;
        if (a.prob_out) a.prob_out += b0;
        if (a.state_out) a.state_out += o;
        if (a.state8_out) a.state8_out += o;
        if (a.avg_out) a.avg_out += o; if (a.power_out) a.power_out += o;
        const dim3 grid(tiles, a.B), block(POST_THREADS);
        // 4 members per load round: 5 or 8 per round cost more registers (occupancy) than the round trip
        // they save (measured on E = 5)
        if (own_max) {
            if (vec) hipLaunchKernelGGL((post_kernel<true, POST_MU, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((post_kernel<true, POST_MU, false>), grid, block, 0, s, a);
        } else {
            if (vec) hipLaunchKernelGGL((post_kernel<false, POST_MU, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((post_kernel<false, POST_MU, false>), grid, block, 0, s, a);
        }
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

}  // namespace

extern "C" {

const char* camal_version(void) { return "camal_hip 1 gfx950"; }
const char* camal_last_error(void) { return g_err.c_str(); }
size_t camal_member_blob_floats(int kernel_size) { return blob_floats(kernel_size); }
size_t camal_member_blob_floats_arch(int kernel_size, int arch) {
    return arch == CAMAL_ARCH_CAMAL || arch == CAMAL_ARCH_RESNET ? blob_floats(kernel_size, arch) : 0;
}

int camal_create(int device_id, int n_members, const int* kernel_sizes, int precision, camal_ctx** out) {
    return camal_create_arch(device_id, n_members, kernel_sizes, precision, CAMAL_ARCH_CAMAL, out);
}

int camal_create_arch(int device_id, int n_members, const int* kernel_sizes, int precision, int arch, camal_ctx** out) {
    return camal_create_widths(device_id, n_members, kernel_sizes, precision, arch, C1, out);
}

size_t camal_member_blob_floats_widths(int kernel_size, int arch, int mid_channels) {
    if (!mid_supported(mid_channels) || (arch != CAMAL_ARCH_CAMAL && arch != CAMAL_ARCH_RESNET)) return 0;
    return blob_floats(kernel_size, arch, mid_channels);
}

int camal_create_widths(int device_id, int n_members, const int* kernel_sizes, int precision, int arch, int mid_channels,
                        camal_ctx** out) {
    if (!out || !kernel_sizes) return fail(CAMAL_ERR_ARG, "null argument");
    if (!mid_supported(mid_channels)) return fail(CAMAL_ERR_ARG, "mid_channels must be 32, 64 or 128");
    if (arch != CAMAL_ARCH_CAMAL && arch != CAMAL_ARCH_RESNET) return fail(CAMAL_ERR_ARG, "arch must be 0 (CamALResNet) or 1 (ResNet)");
    if (arch == CAMAL_ARCH_RESNET)
        for (int e = 0; e < n_members; ++e)
            if (kernel_sizes[e] != 7) return fail(CAMAL_ERR_ARG, "the ResNet variant has kernel sizes 7, 5, 3: pass 7");
    if (n_members < 1 || n_members > CAMAL_MAX_MEMBERS) return fail(CAMAL_ERR_ARG, "n_members must be in 1..16");
    if (precision != CAMAL_FP32 && precision != CAMAL_BF16) return fail(CAMAL_ERR_ARG, "precision must be 0 (fp32) or 1 (bf16)");
    for (int e = 0; e < n_members; ++e)
        if (kernel_sizes[e] < 1 || kernel_sizes[e] > 63 || !(kernel_sizes[e] & 1))
            return fail(CAMAL_ERR_ARG, "kernel sizes must be odd and in 1..63");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(CAMAL_ERR_ARG, "no such device");
    HIP_TRY(hipSetDevice(device_id));
    camal_ctx* c = new camal_ctx();
    c->device = device_id; c->E = n_members; c->precision = precision; c->arch = arch; c->mid = mid_channels;
    // fused blocks pay 8 halo columns per tile: a win where the per-layer path is bandwidth/latency
    // bound (bf16), a 6 % loss of MFMA work where it is already MFMA-bound (fp32, 128-column tiles)
    c->fused = precision == CAMAL_BF16;
    if (const char* e = getenv("CAMAL_FUSED")) c->fused = atoi(e) != 0;
    if (const char* e = getenv("CAMAL_FUSED_BLOCK1")) c->fused_block1 = atoi(e) != 0;
    if (const char* e = getenv("CAMAL_STEM_MFMA")) c->stem_mfma = atoi(e) != 0;
    if (const char* e = getenv("CAMAL_POST_WINDOWS")) c->post_windows = std::max(1, atoi(e));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) c->num_cu = prop.multiProcessorCount;
    c->m.resize(n_members);
    for (int e = 0; e < n_members; ++e) c->m[e].k = kernel_sizes[e];
    if (dev_alloc(c, (void**)&c->zero_row, 1024) || hipMemset(c->zero_row, 0, 1024) != hipSuccess) {
        camal_destroy(c);
        return fail(CAMAL_ERR_HIP, "allocating the zero row failed");
    }
    *out = c;
    return 0;
}

int camal_destroy(camal_ctx* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (void* p : c->owned) (void)hipFree(p);
    if (c->ws) (void)hipFree(c->ws);
    for (hipEvent_t e : c->pool) (void)hipEventDestroy(e);
    delete c;
    return 0;
}

int camal_set_fused(camal_ctx* c, int fused) {
    if (!c) return fail(CAMAL_ERR_ARG, "null context");
    c->fused = fused != 0;
    return 0;
}

int camal_set_chunk_windows(camal_ctx* c, int windows) {
    if (!c || windows < 1) return fail(CAMAL_ERR_ARG, "chunk windows must be >= 1");
    c->chunk = windows;
    return 0;
}

int camal_reserve(camal_ctx* c, int B, int L) {
    if (!c) return fail(CAMAL_ERR_ARG, "null context");
    if (B < 1 || L < 1) return fail(CAMAL_ERR_ARG, "B and L must be >= 1");
    HIP_TRY(hipSetDevice(c->device));
    const int BC = std::min(B, auto_chunk(c, L));
    const long long fit = (1LL << 30) / ((long long)c->E * L * 4);
    const int PB = std::min(B, std::max(BC, (int)std::min<long long>(c->post_windows, fit) / BC * BC));  // as camal_localize
    Workspace w;
    return ensure_workspace(c, BC, PB, L, &w);
}

int camal_set_member_weights(camal_ctx* c, int member, const float* f, size_t n_floats) {
    if (!c || !f) return fail(CAMAL_ERR_ARG, "null argument");
    if (member < 0 || member >= c->E) return fail(CAMAL_ERR_ARG, "member index out of range");
    Member& m = c->m[member];
    if (m.loaded) return fail(CAMAL_ERR_STATE, "member weights already set (create a new context to replace them)");
    if (n_floats != blob_floats(m.k, c->arch, c->mid))
        return fail(CAMAL_ERR_ARG, "blob has " + std::to_string(n_floats) + " floats, expected " + std::to_string(blob_floats(m.k, c->arch, c->mid)));
    const int C1 = c->mid, C2 = 2 * c->mid;  // this context's widths (shadow the defaults)
    HIP_TRY(hipSetDevice(c->device));
    const float* p = f;
    int r;
    for (int i = 0; i < 9; ++i) {
        const int cin = conv_shape(i, c->mid).cin, cout = conv_shape(i, c->mid).cout, ks = conv_ks(c->arch, i, m.k);
        const float* w = p; p += (size_t)cout * cin * ks;
        const float* b = p; p += cout;
        if (i == 0) {  // stem: [C1][1][k] -> [k][C1]
            std::vector<float> t((size_t)ks * C1);
            for (int co = 0; co < C1; ++co)
                for (int j = 0; j < ks; ++j) t[(size_t)j * C1 + co] = w[(size_t)co * ks + j];
            if ((r = upload(c, (void**)&m.stem_w, t.data(), t.size() * 4))) return r;
            if ((r = upload(c, (void**)&m.stem_b, b, C1 * 4))) return r;
            if (c->precision == CAMAL_BF16 && C1 == 64) {
                // [mt][tap group g][mi][hi | lo][lane][8 x bf16]: w[32 mt + 16 mi + (l & 15)][32 g + 8 (l >> 4) + jj]
                const int ngrp = (ks + 31) / 32;
                std::vector<uint16_t> sp((size_t)2 * ngrp * 4 * 64 * 8, 0);
                for (int mt = 0; mt < 2; ++mt)
                    for (int g = 0; g < ngrp; ++g)
                        for (int mi = 0; mi < 2; ++mi)
                            for (int l = 0; l < 64; ++l)
                                for (int jj = 0; jj < 8; ++jj) {
                                    const int co = 32 * mt + 16 * mi + (l & 15), tap = 32 * g + 8 * (l >> 4) + jj;
                                    const float v = tap < ks ? w[(size_t)co * ks + tap] : 0.f;
                                    const uint16_t hi = f2bf(v);
                                    uint32_t hb = (uint32_t)hi << 16;
                                    float hf;
                                    memcpy(&hf, &hb, 4);
                                    const size_t o = ((((size_t)(mt * ngrp + g) * 4 + 2 * mi) * 64 + l) * 8) + jj;
                                    sp[o] = hi;
                                    sp[o + 64 * 8] = f2bf(v - hf);
                                }
                if ((r = upload(c, &m.stem_pack, sp.data(), sp.size() * 2))) return r;
            }
        } else if ((r = make_layer(c, m.conv[i - 1], w, b, cout, cin, ks))) return r;
    }
    if ((r = upload(c, (void**)&m.res1_w, p, C1 * 4))) return r;
    p += C1;
    if ((r = upload(c, (void**)&m.res1_b, p, C1 * 4))) return r;
    p += C1;
    if ((r = make_layer(c, m.res2, p, p + (size_t)C2 * C1, C2, C1, 1))) return r;
    p += (size_t)C2 * C1 + C2;
    if (c->precision == CAMAL_BF16 && c->mid == 64) {
        if ((r = make_block_stream(c, {&m.conv[2], &m.conv[3], &m.conv[4], &m.res2}, &m.stream2, &m.stride2))) return r;
        if ((r = make_block_stream(c, {&m.conv[5], &m.conv[6], &m.conv[7]}, &m.stream3, &m.stride3))) return r;
    }
    if ((r = upload(c, (void**)&m.wfc, p, 2 * C2 * 4))) return r;
    p += 2 * C2;
    if ((r = upload(c, (void**)&m.bfc, p, 2 * 4))) return r;
    m.loaded = true;
    return 0;
}

int camal_forward_cam(camal_ctx* c, const float* x, int B, int L, int member, int cam_label, float* logits, float* cam,
                      void* stream) {
    return camal_forward_cam_tap(c, x, B, L, member, cam_label, 2, logits, cam, stream);
}

int camal_forward_cam_tap(camal_ctx* c, const float* x, int B, int L, int member, int cam_label, int tap_block,
                          float* logits, float* cam, void* stream) {
    if (int r = check_ctx(c, B, L)) return r;
    if (tap_block != 1 && tap_block != 2) return fail(CAMAL_ERR_ARG, "tap_block must be 1 (second ResNetBlock) or 2 (last ResNetBlock)");
    if (cam_label != 0 && cam_label != 1) return fail(CAMAL_ERR_ARG, "cam_label must be 0 or 1");
    c->cam_row = cam_label;
    if (!x || !logits) return fail(CAMAL_ERR_ARG, "x_dev and logits_dev are required");
    if (member < -1 || member >= c->E) return fail(CAMAL_ERR_ARG, "member index out of range");
    hipStream_t s = (hipStream_t)stream;
    const int BC = std::min(B, auto_chunk(c, L));
    Workspace w;
    if (int r = ensure_workspace(c, BC, BC, L, &w)) return r;
    const int e0 = member < 0 ? 0 : member, e1 = member < 0 ? c->E : member + 1, EO = e1 - e0;
    for (int c0 = 0; c0 < B; c0 += BC) {
        const int n = std::min(BC, B - c0);
        for (int e = e0; e < e1; ++e) {
            float* camw = w.cam;  // member rows are consumed immediately: one row buffer is enough
            // tap_block 1: the hooked layer is block 2; its CAM goes to r1 (block 1's skip, free once block 1 is done)
            c->tap_block = (tap_block == 1 && cam) ? 1 : 2;
            c->tap_cam = c->tap_block == 1 ? (float*)w.r1 : nullptr;
            const int rr = run_member_any(c, s, c->m[e], x + (size_t)c0 * L, n, L, w, camw, w.part, w.logits, nullptr);
            if (c->tap_block == 1) camw = c->tap_cam;
            c->tap_block = 2; c->tap_cam = nullptr;
            if (rr) return rr;
            HIP_TRY(hipMemcpyAsync(logits + ((size_t)(e - e0) * B + c0) * 2, w.logits, (size_t)n * 2 * 4, hipMemcpyDeviceToDevice, s));
            if (cam) HIP_TRY(hipMemcpyAsync(cam + ((size_t)(e - e0) * B + c0) * L, camw, (size_t)n * L * 4, hipMemcpyDeviceToDevice, s));
        }
    }
    (void)EO;
    return 0;
}

int camal_postprocess(int E, const float* cam, const float* logits, const float* x, const double* x64, const int8_t* y,
                      int B, int L, int w_ma, int gate_mode, int pred_f64, double p_on, double scale_mul,
                      double scale_add, double* prob_out, float* avg_cam_out, float* state_out, float* power_out,
                      void* stream) {
    if (E < 1 || E > CAMAL_MAX_MEMBERS) return fail(CAMAL_ERR_ARG, "n_members must be in 1..16");
    if (!cam || !logits || !x || !prob_out || !state_out) return fail(CAMAL_ERR_ARG, "null argument");
    if (B < 1 || L < 1) return fail(CAMAL_ERR_ARG, "B and L must be >= 1");
    if (w_ma < 1 || w_ma > 64) return fail(CAMAL_ERR_ARG, "w_ma must be in 1..64");
    PostArgs a;
    a.cam = cam; a.logits = logits; a.hdr = nullptr; a.x = x; a.x64 = x64; a.y = y;
    a.cam_estride = (size_t)B * L; a.lg_estride = (size_t)B * 2;
    a.prob_out = prob_out; a.avg_out = avg_cam_out; a.state_out = state_out; a.state8_out = nullptr; a.power_out = power_out;
    a.E = E; a.B = B; a.L = L; a.w_ma = w_ma; a.gate_mode = gate_mode; a.pred_f64 = pred_f64;
    a.p_on = p_on; a.scale_mul = scale_mul; a.scale_add = scale_add;
    return launch_post(nullptr, (hipStream_t)stream, a, true);
}

// camal_localize / camal_localize_i8: the states go out as float32 (state_out) or as one-byte codes (state8_out)
static int localize_impl(camal_ctx* c, const float* x, const double* x64, const int8_t* y, int B, int L, int w_ma,
                         int gate_mode, int pred_f64, double p_on, double scale_mul, double scale_add, double* prob_out,
                         float* avg_cam_out, float* state_out, int8_t* state8_out, float* power_out,
                         float* member_logits_out, float* member_cam_out, void* stream) {
    if (int r = check_ctx(c, B, L)) return r;
    if (!x || !prob_out || (!state_out && !state8_out)) return fail(CAMAL_ERR_ARG, "x_dev, prob_out and state_out are required");
    if (w_ma < 1 || w_ma > 64) return fail(CAMAL_ERR_ARG, "w_ma must be in 1..64");
    if (gate_mode != 0 && gate_mode != 1) return fail(CAMAL_ERR_ARG, "gate_mode must be 0 or 1");
    hipStream_t s = (hipStream_t)stream;
    c->cam_row = 1;  // core.py:302-303 always asks for the CAM of label 1
    const int BC = std::min(B, auto_chunk(c, L)), E = c->E;
    // windows per post-processing launch: a whole number of conv passes, at most post_windows and at most
    // 1 GiB of waiting CAM rows
    const long long fit = (1LL << 30) / ((long long)E * L * 4);
    const int PB = std::min(B, std::max(BC, (int)std::min<long long>(c->post_windows, fit) / BC * BC));
    Workspace w;
    if (int r = ensure_workspace(c, BC, PB, L, &w)) return r;
    for (int g0 = 0; g0 < B; g0 += PB) {
        // one post-processing launch per group of ng windows; the group's CAM rows, logits and row maxima
        // wait in the workspace as [E][ng][..]
        const int ng = std::min(PB, B - g0);
        {
            const size_t nz = (size_t)E * ng;
            hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)std::min<size_t>((nz + 255) / 256, 1024)), dim3(256), 0, s, w.cmax, nz);
            HIP_TRY(hipGetLastError());
        }
        HeaderArgs h;
        for (int c0 = g0; c0 < g0 + ng; c0 += BC) {
            const int n = std::min(BC, g0 + ng - c0);
            for (int e = 0; e < E; ++e) {
                const int nt = member_ntiles(c, c->m[e], L);
                float* camw = w.cam + ((size_t)e * ng + (c0 - g0)) * L;
                // member stride ng * w.ntiles * 2 (upper bound), window stride = the member's own tile count
                float* partw = w.part + (size_t)e * ng * w.ntiles * 2 + (size_t)(c0 - g0) * nt * 2;
                float* cmw = w.cmax + (size_t)e * ng + (c0 - g0);
                if (int r = run_member_any(c, s, c->m[e], x + (size_t)c0 * L, n, L, w, camw, partw, nullptr, cmw)) return r;
                if (member_cam_out)
                    HIP_TRY(hipMemcpyAsync(member_cam_out + ((size_t)e * B + c0) * L, camw, (size_t)n * L * 4, hipMemcpyDeviceToDevice, s));
            }
        }
        // one header launch per group: logits, votes, per-member 1/max
        h.part = w.part; h.cmax = w.cmax; h.logits = w.logits;
        h.part_estride = (size_t)ng * w.ntiles * 2; h.cmax_estride = (size_t)ng; h.lg_estride = (size_t)ng * 2;
        for (int e = 0; e < E; ++e) { h.bfc[e] = c->m[e].bfc; h.ntiles[e] = member_ntiles(c, c->m[e], L); }
        h.y = y ? y + g0 : nullptr;
        h.prob_out = prob_out + g0; h.hdr = w.hdr;
        h.E = E; h.B = ng; h.L = L; h.gate_mode = gate_mode;
        {
            ProfScope ps(c, s, 4);
            hipLaunchKernelGGL(header_kernel, dim3((ng + 127) / 128), dim3(128), 0, s, h);
            HIP_TRY(hipGetLastError());
        }
        if (member_logits_out)
            for (int e = 0; e < E; ++e)
                HIP_TRY(hipMemcpyAsync(member_logits_out + ((size_t)e * B + g0) * 2, w.logits + (size_t)e * ng * 2, (size_t)ng * 2 * 4, hipMemcpyDeviceToDevice, s));
        PostArgs a;
        a.cam = w.cam; a.logits = nullptr; a.hdr = w.hdr;
        a.cam_estride = (size_t)ng * L; a.lg_estride = 0;
        a.x = x + (size_t)g0 * L; a.x64 = x64 ? x64 + (size_t)g0 * L : nullptr;
        a.y = nullptr;
        a.prob_out = nullptr; a.avg_out = avg_cam_out ? avg_cam_out + (size_t)g0 * L : nullptr;
        a.state_out = state_out ? state_out + (size_t)g0 * L : nullptr;
        a.state8_out = state8_out ? state8_out + (size_t)g0 * L : nullptr;
        a.power_out = power_out ? power_out + (size_t)g0 * L : nullptr;
        a.E = E; a.B = ng; a.L = L; a.w_ma = w_ma; a.gate_mode = gate_mode; a.pred_f64 = pred_f64;
        a.p_on = p_on; a.scale_mul = scale_mul; a.scale_add = scale_add;
        if (int r = launch_post(c, s, a, false)) return r;
    }
    return 0;
}

int camal_localize(camal_ctx* c, const float* x, const double* x64, const int8_t* y, int B, int L, int w_ma,
                   int gate_mode, int pred_f64, double p_on, double scale_mul, double scale_add, double* prob_out,
                   float* avg_cam_out, float* state_out, float* power_out, float* member_logits_out,
                   float* member_cam_out, void* stream) {
    return localize_impl(c, x, x64, y, B, L, w_ma, gate_mode, pred_f64, p_on, scale_mul, scale_add, prob_out, avg_cam_out,
                         state_out, nullptr, power_out, member_logits_out, member_cam_out, stream);
}

int camal_localize_i8(camal_ctx* c, const float* x, const double* x64, const int8_t* y, int B, int L, int w_ma,
                      int gate_mode, int pred_f64, double p_on, double scale_mul, double scale_add, double* prob_out,
                      int8_t* state_out, float* power_out, void* stream) {
    if (!state_out) return fail(CAMAL_ERR_ARG, "state_out is required");
    return localize_impl(c, x, x64, y, B, L, w_ma, gate_mode, pred_f64, p_on, scale_mul, scale_add, prob_out, nullptr,
                         nullptr, state_out, power_out, nullptr, nullptr, stream);
}

int camal_attention(const double* soft, const double* x, int B, int L, int w_ma, double* state_out, void* stream) {
    if (!soft || !x || !state_out) return fail(CAMAL_ERR_ARG, "null argument");
    if (B < 1 || L < 1) return fail(CAMAL_ERR_ARG, "B and L must be >= 1");
    if (w_ma < 1) return fail(CAMAL_ERR_ARG, "w_ma must be >= 1");
    hipLaunchKernelGGL(attention_kernel, dim3((L + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, soft, x, state_out, L, w_ma);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int win_scan_blocks(int n_wins) { return (n_wins + WIN_SCAN_BLOCK - 1) / WIN_SCAN_BLOCK; }

size_t camal_windows_workspace_bytes(int n_wins) {
    if (n_wins < 1) return 0;
    return align_up((size_t)n_wins) + align_up((size_t)n_wins * 4) + align_up((size_t)win_scan_blocks(n_wins) * 4);
}

int camal_windows_count(long long T, int W, int stride) {
    if (T < W || W < 1 || stride < 1) return 0;
    const long long n = W == stride ? T / stride : 1 + (T - W) / stride;  // data_processing.py:377-380
    return n > 0x7fffffffLL ? 0x7fffffff : (int)n;
}

// shared body of camal_build_windows / camal_build_windows_static: `workspace` is device memory of at least
// camal_windows_workspace_bytes(n_wins) bytes (checked by the callers)
static int build_windows_impl(const void* series, int series_is_f64, int n_rows, long long T, int W, int stride, int app,
                              double stat1, double stat2, float* x_out, double* x64_out, int8_t* label_out,
                              float* app_power_out, int8_t* app_state_out, int* index_out, int* count_out, int n_wins,
                              void* workspace, void* stream) {
    WinArgs a;
    a.s32 = series_is_f64 ? nullptr : (const float*)series;
    a.s64 = series_is_f64 ? (const double*)series : nullptr;
    a.T = T; a.n_rows = n_rows; a.W = W; a.stride = stride; a.n_wins = n_wins; a.app = app;
    a.stat1 = stat1; a.stat2 = stat2;
    char* ws = (char*)workspace;
    a.keep = (unsigned char*)ws; ws += align_up((size_t)n_wins);
    a.pos = (int*)ws; ws += align_up((size_t)n_wins * 4);
    a.blocksum = (int*)ws;
    a.x_out = x_out; a.x64_out = x64_out; a.label_out = label_out; a.app_power_out = app_power_out;
    a.app_state_out = app_state_out; a.index_out = index_out; a.count_out = count_out;
    hipStream_t s = (hipStream_t)stream;
    const int nb = win_scan_blocks(n_wins);
    hipLaunchKernelGGL(win_flag_kernel, dim3(n_wins), dim3(256), 0, s, a);
    hipLaunchKernelGGL(win_scan_block_kernel, dim3(nb), dim3(WIN_SCAN_BLOCK), 0, s, a);
    hipLaunchKernelGGL(win_scan_top_kernel, dim3(1), dim3(WIN_SCAN_BLOCK), 0, s, a, nb);
    hipLaunchKernelGGL(win_copy_kernel, dim3(n_wins), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int check_windows_args(const void* series, int n_rows, long long T, int W, int stride, int app, double stat2,
                              const float* x_out, const int8_t* label_out, const float* app_power_out,
                              const int8_t* app_state_out, const int* count_out, int* n_wins) {
    if (!series || !x_out || !count_out) return fail(CAMAL_ERR_ARG, "series_dev, x_out and count_out are required");
    if (n_rows < 1 || T < 1 || W < 1 || stride < 1) return fail(CAMAL_ERR_ARG, "n_rows, T, W and stride must be >= 1");
    if (app < 1) return fail(CAMAL_ERR_ARG, "app is the 1-based appliance index");
    if ((label_out || app_power_out || app_state_out) && 2 * app >= n_rows)
        return fail(CAMAL_ERR_ARG, "the series has no (power, status) rows for that appliance");
    if (stat2 == 0.0) return fail(CAMAL_ERR_ARG, "stat2 must not be 0");
    *n_wins = camal_windows_count(T, W, stride);
    if (*n_wins < 1) return fail(CAMAL_ERR_ARG, "the series is shorter than one window");
    return 0;
}

int camal_build_windows(const void* series, int series_is_f64, int n_rows, long long T, int W, int stride, int app,
                        double stat1, double stat2, float* x_out, double* x64_out, int8_t* label_out, float* app_power_out,
                        int8_t* app_state_out, int* index_out, int* count_out, void* workspace, size_t workspace_bytes,
                        void* stream) {
    int n_wins = 0;
    if (int r = check_windows_args(series, n_rows, T, W, stride, app, stat2, x_out, label_out, app_power_out, app_state_out,
                                   count_out, &n_wins)) return r;
    if (!workspace) return fail(CAMAL_ERR_ARG, "workspace_dev is required");
    if (workspace_bytes < camal_windows_workspace_bytes(n_wins)) return fail(CAMAL_ERR_ARG, "workspace smaller than camal_windows_workspace_bytes(n_wins)");
    return build_windows_impl(series, series_is_f64, n_rows, T, W, stride, app, stat1, stat2, x_out, x64_out, label_out,
                              app_power_out, app_state_out, index_out, count_out, n_wins, workspace, stream);
}

// ---- the entry point that allocates nothing (include/camal_hip.h: camal_build_windows_static) ----------------
// Scratch of the window builder (NaN flags, scan positions, block totals) for up to CAMAL_WINDOWS_STATIC_MAX windows,
// as a file-scope device array: every GPU the library runs on holds its own copy until the process ends.  In host
// code the array's name is only a handle for hipGetSymbolAddress; kernels get the resolved per-device address.
constexpr size_t win_ws_bytes(size_t n) {  // = camal_windows_workspace_bytes(n), monotone in n
    return ((n + 255) & ~(size_t)255) + ((n * 4 + 255) & ~(size_t)255) +
           (((n + WIN_SCAN_BLOCK - 1) / WIN_SCAN_BLOCK * 4 + 255) & ~(size_t)255);
}
constexpr size_t kWinScratchBytes = win_ws_bytes(CAMAL_WINDOWS_STATIC_MAX);  // 5 MiB + 4 KiB
static __device__ __align__(256) unsigned char g_win_scratch[kWinScratchBytes];

constexpr int kMaxDevices = 64;
static void* g_win_scratch_ptr[kMaxDevices];  // resolved address of g_win_scratch on each device (0: not yet)
static std::mutex g_win_scratch_mu;

// address of this device's copy of the scratch array; resolved once per device with that device current
static int win_scratch_for_current_device(void** out) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(CAMAL_ERR_ARG, "device index beyond the static scratch table");
    std::lock_guard<std::mutex> lock(g_win_scratch_mu);
    if (!g_win_scratch_ptr[dev]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_win_scratch)));
        if (!p) return fail(CAMAL_ERR_HIP, "hipGetSymbolAddress returned a null scratch address");
        g_win_scratch_ptr[dev] = p;
    }
    *out = g_win_scratch_ptr[dev];
    return 0;
}

size_t camal_windows_static_scratch_bytes(void) { return kWinScratchBytes; }

int camal_build_windows_static(const void* series, int series_is_f64, int n_rows, long long T, int W, int stride, int app,
                               double stat1, double stat2, float* x_out, double* x64_out, int8_t* label_out,
                               float* app_power_out, int8_t* app_state_out, int* index_out, int* count_out, void* stream) {
    int n_wins = 0;
    if (int r = check_windows_args(series, n_rows, T, W, stride, app, stat2, x_out, label_out, app_power_out, app_state_out,
                                   count_out, &n_wins)) return r;
    if (n_wins > CAMAL_WINDOWS_STATIC_MAX)
        return fail(CAMAL_ERR_ARG, "camal_build_windows_static: " + std::to_string(n_wins) + " windows exceed the static scratch's maximum of " +
                                       std::to_string(CAMAL_WINDOWS_STATIC_MAX) + " (use camal_build_windows with a caller workspace)");
    if (camal_windows_workspace_bytes(n_wins) > kWinScratchBytes) return fail(CAMAL_ERR_ARG, "static scratch too small (internal)");
    // The scratch is per GPU: it must be the copy of the GPU that holds the caller's buffers (and runs the stream),
    // which need not be the current device (torch tensors on cuda:1 while device 0 is current).  The device comes from
    // the pointers themselves; the call runs with that device current and puts the caller's device back.
    hipPointerAttribute_t pa_in, pa_out;
    HIP_TRY(hipPointerGetAttributes(&pa_in, series));
    HIP_TRY(hipPointerGetAttributes(&pa_out, x_out));
    if (pa_in.type != hipMemoryTypeDevice || pa_out.type != hipMemoryTypeDevice)
        return fail(CAMAL_ERR_ARG, "camal_build_windows_static: series_dev and x_out must be device memory");
    if (pa_in.device != pa_out.device)
        return fail(CAMAL_ERR_ARG, "camal_build_windows_static: series_dev and x_out live on different GPUs");
    int prev = -1;
    HIP_TRY(hipGetDevice(&prev));
    const int dev = pa_out.device;
    if (dev != prev) HIP_TRY(hipSetDevice(dev));
    void* scratch = nullptr;
    int r = win_scratch_for_current_device(&scratch);
    // every scratch byte a call reads (keep[0..n), pos[0..n), block totals[0..nb)) is written by that call's own
    // kernels first, so the previous call's contents need no clearing
    if (!r)
        r = build_windows_impl(series, series_is_f64, n_rows, T, W, stride, app, stat1, stat2, x_out, x64_out, label_out,
                               app_power_out, app_state_out, index_out, count_out, n_wins, scratch, stream);
    if (dev != prev) (void)hipSetDevice(prev);
    return r;
}

static int nilm_blocks(size_t n) {
    const size_t want = (n + (size_t)NILM_THREADS * 8 - 1) / ((size_t)NILM_THREADS * 8);
    return (int)std::max<size_t>(1, std::min<size_t>(want, NILM_MAX_BLOCKS));
}

size_t camal_nilm_workspace_bytes(size_t n) { return (size_t)nilm_blocks(n) * NILM_NSUMS * sizeof(double); }

int camal_nilm_sums(const float* y32, const double* y64, const float* yhat, const int8_t* ystate, const float* yhat_state,
                    size_t n, double* sums_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!sums_out || !workspace) return fail(CAMAL_ERR_ARG, "sums_out and workspace_dev are required");
    if (n < 1) return fail(CAMAL_ERR_ARG, "n must be >= 1");
    if (y32 && y64) return fail(CAMAL_ERR_ARG, "pass the true power as float32 OR float64, not both");
    if ((y32 || y64) && !yhat) return fail(CAMAL_ERR_ARG, "y given without y_hat");
    if ((ystate != nullptr) != (yhat_state != nullptr)) return fail(CAMAL_ERR_ARG, "y_state and y_hat_state come together");
    if (workspace_bytes < camal_nilm_workspace_bytes(n)) return fail(CAMAL_ERR_ARG, "workspace smaller than camal_nilm_workspace_bytes(n)");
    NilmArgs a;
    a.y32 = y32; a.y64 = y64; a.yhat = yhat; a.ystate = ystate; a.yhat_state = yhat_state; a.n = n;
    a.partial = (double*)workspace;
    const int nb = nilm_blocks(n);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(nilm_partial_kernel, dim3(nb), dim3(NILM_THREADS), 0, s, a);
    hipLaunchKernelGGL(nilm_final_kernel, dim3(1), dim3(NILM_NSUMS * 64), 0, s, a.partial, nb, sums_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

#ifdef CAMAL_STAMPS
// diagnostic build only (tools/stamps.py): copies g_stamps [256][9][16] to the host
int camal_debug_stamps(unsigned long long* out) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(camal::g_stamps), sizeof(unsigned long long) * 256 * 9 * CAMAL_NSTAMP));
    return 0;
}
#endif

int camal_set_profiling(camal_ctx* c, int enabled) {
    if (!c) return fail(CAMAL_ERR_ARG, "null context");
    c->prof = enabled != 0;
    c->recs.clear();
    c->pool_used = 0;
    for (int i = 0; i < CAMAL_PROF_CLASSES; ++i) { c->prof_launches[i] = 0; c->prof_ms[i] = 0; }
    return 0;
}

int camal_get_profile(camal_ctx* c, int* launches, double* ms) {
    if (!c || !launches || !ms) return fail(CAMAL_ERR_ARG, "null argument");
    for (const ProfRec& r : c->recs) {
        HIP_TRY(hipEventSynchronize(r.b));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        c->prof_launches[r.cls] += 1;
        c->prof_ms[r.cls] += t;
    }
    c->recs.clear();
    c->pool_used = 0;
    for (int i = 0; i < CAMAL_PROF_CLASSES; ++i) { launches[i] = c->prof_launches[i]; ms[i] = c->prof_ms[i]; }
    return 0;
}

}  // extern "C"

#include "camal_train_impl.hpp"
