// Model handle of the C ABI (include/vda.h, "handle API"): the seam a C / C++ host uses in place of the reference's
//   VideoDepthAnything(**cfg)                      video_depth.py:38-63      -> vda_create
//   .load_state_dict(sd, strict=True)              run.py:46                 -> vda_load_weight x N + vda_finalize_weights
//   .forward(x)                                    video_depth.py:89-93,161-164 -> vda_forward
// The handle owns the weights (raw fp32 as loaded + the kernel layouts packed on the device) and the launch sequence of one
// forward pass over the per-kernel entry points; nothing here computes on the host. The Python facade
// (video_depth_anything_amd/video_depth.py) calls exactly these functions through ctypes.
//
// Workspace: every intermediate lives in ONE caller-visible block. The launch sequence is written once (Run::forward) and
// executed in two modes: a dry pass that only records each named buffer's largest request (this is vda_workspace_bytes and
// the layout), and the real pass that resolves names to offsets in the block. No allocation happens in a steady-state
// vda_forward (layouts, the resampled pos-embed and the fp32 weight pack are created the first time a shape / precision is
// seen), so a steady-state forward is graph-capturable.
#include "vda_common.h"
#include <string.h>
#include <exception>
#include <type_traits>
#include <array>
#include <map>
#include <string>
#include <vector>

namespace {

constexpr int PATCH = 14, POS_GRID = 37, KPATCH = 640, GN_GROUPS = 32, TEMPORAL_HEADS = 8;
constexpr float ENC_LN_EPS = 1e-6f, GN_EPS = 1e-6f, TMP_LN_EPS = 1e-5f;

inline int pad64(int c) { return (c + 63) / 64 * 64; }

#define VDA_TRY(expr)            \
    do {                         \
        const int rc_ = (expr);  \
        if (rc_ != 0) return rc_; \
    } while (0)

#define VDA_HIP(expr)                                                      \
    do {                                                                   \
        const hipError_t e_ = (expr);                                      \
        if (e_ != hipSuccess) {                                            \
            vda_set_error("%s: %s", #expr, hipGetErrorString(e_));         \
            return 2;                                                      \
        }                                                                  \
    } while (0)

// ------------------------------------------------------------------ weight packing (device)
template <typename T>
__global__ void pack_matrix_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int K, int Npad, int Kpad) {
    const size_t total = (size_t)Npad * Kpad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i / Kpad), k = (int)(i - (size_t)n * Kpad);
        dst[i] = (n < N && k < K) ? (T)src[(size_t)n * K + k] : (T)0.f;
    }
}
// Conv2d weight [Co,Ci,3,3] -> [Copad, (ky,kx,ci) = 9*Cipad]
template <typename T>
__global__ void pack_conv3x3_kernel(const float* __restrict__ src, T* __restrict__ dst, int Co, int Ci, int Copad, int Cipad) {
    const size_t total = (size_t)Copad * 9 * Cipad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ci = (int)(i % Cipad);
        const size_t t = i / Cipad;
        const int tap = (int)(t % 9), co = (int)(t / 9);
        dst[i] = (co < Co && ci < Ci) ? (T)src[((size_t)co * Ci + ci) * 9 + tap] : (T)0.f;
    }
}
// ConvTranspose2d (k == stride) weight [Ci,Co,k,k] -> [(ky,kx,co) = k*k*cpad, ci = cpad]
template <typename T>
__global__ void pack_convt_kernel(const float* __restrict__ src, T* __restrict__ dst, int Ci, int Co, int k, int cpad) {
    const size_t total = (size_t)k * k * cpad * cpad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ci = (int)(i % cpad);
        const size_t t = i / cpad;
        const int co = (int)(t % cpad), tap = (int)(t / cpad);
        dst[i] = (co < Co && ci < Ci) ? (T)src[((size_t)ci * Co + co) * k * k + tap] : (T)0.f;
    }
}
__global__ void expand_convt_bias_kernel(const float* __restrict__ b, float* __restrict__ dst, int Co, int kk, int cpad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kk * cpad) return;
    const int co = i % cpad;
    dst[i] = co < Co ? b[co] : 0.f;
}
// GEGLU proj [2n, K], rows [value(n) | gate(n)] -> interleaved [16 value | 16 gate] per 32 rows (K = 1: the bias)
template <typename T>
__global__ void pack_geglu_kernel(const float* __restrict__ src, T* __restrict__ dst, int n, int K) {
    const size_t total = (size_t)2 * n * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int R = (int)(i / K), k = (int)(i - (size_t)R * K);
        const int blk = R >> 5, within = R & 31;
        const int srow = within < 16 ? blk * 16 + within : n + blk * 16 + (within - 16);
        dst[i] = (T)src[(size_t)srow * K + k];
    }
}

inline unsigned grid_for(size_t items) {
    const size_t b = (items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

#ifndef VDA_LN_FOLD_DEFAULT
#define VDA_LN_FOLD_DEFAULT 1
#endif

// A forward whose split residual stream left fp16's range has no valid result: its depth is overwritten with NaN so that the
// failure is visible in the data as well as in the status (a clamped ReLU would otherwise turn NaN activations into zeros).
// ... and the report goes to a STICKY word in pinned host memory (written only when set: a later, clean forward cannot erase it however
// far the host runs ahead of the device).
__global__ void __launch_bounds__(256) poison_on_overflow_kernel(const int* __restrict__ flag, float* __restrict__ depth, long long n,
                                                                 volatile int* __restrict__ host_report) {
    if (*flag == 0) return;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) depth[i] = __builtin_nanf("");
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *host_report = 1;
        __threadfence_system();
    }
}

struct Raw {
    float* d = nullptr;
    std::vector<int64_t> dims;
    size_t n = 0;
};

// Optional per-launch timing of the GEMM / conv launches of vda_forward (bench.py's roofline): every `every`-th launch of each
// (shape, epilogue) is bracketed by two events on the launch stream; all launches are counted. Bracketing every launch costs
// ~1 ms per ViT-L clip in event packets (measured), 1 in 4 keeps the timed region honest.
struct ProfSample {
    std::string name;
    double flops;
    hipEvent_t e0, e1;
};
struct Profile {
    int every = 0;
    double min_flops = 0.0;                  // launches below it are counted, never bracketed (vda_set_option "profile_min_gflop")
    std::map<std::array<int, 5>, int> seen;
    std::map<std::string, std::pair<long long, double>> launches;      // kernel name -> (launches, algorithmic flops)
    std::vector<ProfSample> samples;
};

struct Layout {
    int gemm_launches = 0;                                      // GEMM / conv launches of one forward (dynamic-schedule counters)
    std::map<std::string, std::pair<size_t, size_t>> bufs;      // name -> (offset, bytes)
    std::vector<std::string> order;
    size_t total = 0;
};

}  // namespace

struct vda_model {
    vda_config cfg;
    int device = 0;
    std::map<std::string, std::vector<int64_t>> spec;
    std::map<std::string, Raw> raw;
    std::map<std::string, void*> mat[2];      // packed matrices per precision
    std::map<std::string, float*> vec;        // fp32 vectors (biases, affine, LayerScale, PE), shared by both precisions
    bool finalized = false, packed[2] = {false, false};
    float oc3_bias = 0.f;
    void* zero_page = nullptr;
    std::map<std::pair<int, int>, float*> pos_cache;
    void* ws = nullptr;
    int64_t ws_bytes = 0;
    bool ws_owned = false;
    std::map<std::array<int, 5>, Layout> layouts;
    std::vector<void*> owned;                 // every hipMalloc of the handle
    int ocp[4] = {0, 0, 0, 0}, Fhp = 0;
    std::array<int, 5> last_key = {0, 0, 0, 0, -1};
    Profile prof;
    int residual_in_ln = 0;                   // vda_set_option("residual_in_ln"): see Run::forward (off: measured slower end to end)
    int dyn_sched = 0;                        // vda_set_option("dyn_sched"): dynamic tile draw in the 8-phase GEMM (vda_gemm_args.sched); for
                                              // processes that share the GPU with communication kernels (multi-rank runs turn it on)
    int ln_fold = VDA_LN_FOLD_DEFAULT;        // vda_set_option("ln_fold"): LayerNorm folded into the encoder GEMMs either side of it (fp16 path)
    int oc1_fused = 1;                        // vda_set_option("oc1_fused"): refinenet1's 2x upsample folded into output_conv1 (fp16 path)
    int head_overlap = 0;                     // vda_set_option("head_overlap"): the head's tap-0..2 work on a side stream under the last encoder blocks.
                                              // OFF: in-process A/B the sign depends on the box (+0.8 % ViT-L / +1.5 % ViT-S on a 52.1 ms box,
                                              // -0.3 .. -0.5 % on a 50.0 ms box; two clips in flight -0.4 %), profiles/r04/head_overlap_ab.txt
    struct Side {
        hipStream_t stream = nullptr;
        hipEvent_t fork = nullptr, join = nullptr;
    };
    std::map<hipStream_t, Side> sides;        // one side stream + event pair per caller stream (two forwards may be in flight on two)
    int mlp_fused = 0;                        // vda_set_option("mlp_fused"): fc1 + GELU + fc2 + residual in one kernel where built (D = 384; needs ln_fold).
                                              // OFF: measured slower than the two GEMM launches (ViT-S clip 8.69 -> 9.00 ms, mlp_fused.hip's header)
    // Split-stream overflow report (ln_fold): one sticky word in pinned host memory, set by the device at the end of a forward whose
    // flag was raised; vda_forward_status / the next vda_forward read (and clear) it.
    volatile int32_t* ovf_host = nullptr;     // hipHostMalloc'ed, device-visible
};

namespace {

using Spec = std::map<std::string, std::vector<int64_t>>;

// Checkpoint inventory (name -> shape) of the flat fp32 state dict run.py:46 loads: dinov2.py:106-168, dpt.py:60-124,
// util/blocks.py:20-32,52-58,124-129, motion_module/motion_module.py:84-100,141-161,194, motion_module/attention.py:81-91,333,374.
void build_spec(const vda_config& c, Spec& s) {
    const int64_t D = c.embed_dim, F = c.features;
    const int64_t oc[4] = {c.out_channels[0], c.out_channels[1], c.out_channels[2], c.out_channels[3]};
    auto put = [&](const std::string& k, std::vector<int64_t> v) { s[k] = std::move(v); };
    const std::string p = "pretrained.";
    put(p + "cls_token", {1, 1, D});
    put(p + "pos_embed", {1, POS_GRID * POS_GRID + 1, D});
    put(p + "mask_token", {1, D});
    put(p + "patch_embed.proj.weight", {D, 3, PATCH, PATCH});
    put(p + "patch_embed.proj.bias", {D});
    for (int i = 0; i < c.depth; ++i) {
        const std::string b = p + "blocks." + std::to_string(i) + ".";
        put(b + "norm1.weight", {D});
        put(b + "norm1.bias", {D});
        put(b + "attn.qkv.weight", {3 * D, D});
        put(b + "attn.qkv.bias", {3 * D});
        put(b + "attn.proj.weight", {D, D});
        put(b + "attn.proj.bias", {D});
        put(b + "ls1.gamma", {D});
        put(b + "norm2.weight", {D});
        put(b + "norm2.bias", {D});
        put(b + "mlp.fc1.weight", {4 * D, D});
        put(b + "mlp.fc1.bias", {4 * D});
        put(b + "mlp.fc2.weight", {D, 4 * D});
        put(b + "mlp.fc2.bias", {D});
        put(b + "ls2.gamma", {D});
    }
    put(p + "norm.weight", {D});
    put(p + "norm.bias", {D});
    const std::string h = "head.";
    for (int i = 0; i < 4; ++i) {
        put(h + "projects." + std::to_string(i) + ".weight", {oc[i], D, 1, 1});
        put(h + "projects." + std::to_string(i) + ".bias", {oc[i]});
    }
    put(h + "resize_layers.0.weight", {oc[0], oc[0], 4, 4});
    put(h + "resize_layers.0.bias", {oc[0]});
    put(h + "resize_layers.1.weight", {oc[1], oc[1], 2, 2});
    put(h + "resize_layers.1.bias", {oc[1]});
    put(h + "resize_layers.3.weight", {oc[3], oc[3], 3, 3});
    put(h + "resize_layers.3.bias", {oc[3]});
    if (c.use_clstoken)                                             // dpt.py:92-98
        for (int i = 0; i < 4; ++i) {
            put(h + "readout_projects." + std::to_string(i) + ".0.weight", {D, 2 * D});
            put(h + "readout_projects." + std::to_string(i) + ".0.bias", {D});
        }
    const std::string sc = h + "scratch.";
    for (int i = 0; i < 4; ++i) put(sc + "layer" + std::to_string(i + 1) + "_rn.weight", {F, oc[i], 3, 3});
    for (int i = 1; i <= 4; ++i) {
        const std::string r = sc + "refinenet" + std::to_string(i) + ".";
        put(r + "out_conv.weight", {F, F, 1, 1});
        put(r + "out_conv.bias", {F});
        for (int u = 1; u <= 2; ++u)
            for (int cc = 1; cc <= 2; ++cc) {
                const std::string k = r + "resConfUnit" + std::to_string(u) + ".conv" + std::to_string(cc);
                put(k + ".weight", {F, F, 3, 3});
                put(k + ".bias", {F});
            }
        if (c.use_bn)                                                // util/blocks.py:60-62 (state_dict keys of nn.BatchNorm2d)
            for (int u = 1; u <= 2; ++u)
                for (int cc = 1; cc <= 2; ++cc) {
                    const std::string k = r + "resConfUnit" + std::to_string(u) + ".bn" + std::to_string(cc) + ".";
                    put(k + "weight", {F});
                    put(k + "bias", {F});
                    put(k + "running_mean", {F});
                    put(k + "running_var", {F});
                    put(k + "num_batches_tracked", {});
                }
    }
    put(sc + "output_conv1.weight", {F / 2, F, 3, 3});
    put(sc + "output_conv1.bias", {F / 2});
    put(sc + "output_conv2.0.weight", {32, F / 2, 3, 3});
    put(sc + "output_conv2.0.bias", {32});
    put(sc + "output_conv2.2.weight", {1, 32, 1, 1});
    put(sc + "output_conv2.2.bias", {1});
    const int64_t tc[4] = {oc[2], oc[3], F, F};                     // dpt_temporal.py:42-51
    for (int m = 0; m < 4; ++m) {
        const int64_t C = tc[m];
        const std::string t = h + "motion_modules." + std::to_string(m) + ".temporal_transformer.";
        put(t + "norm.weight", {C});
        put(t + "norm.bias", {C});
        put(t + "proj_in.weight", {C, C});
        put(t + "proj_in.bias", {C});
        const std::string tb = t + "transformer_blocks.0.";
        for (int a = 0; a < 2; ++a) {
            const std::string ab = tb + "attention_blocks." + std::to_string(a) + ".";
            put(ab + "to_q.weight", {C, C});
            put(ab + "to_k.weight", {C, C});
            put(ab + "to_v.weight", {C, C});
            put(ab + "to_out.0.weight", {C, C});
            put(ab + "to_out.0.bias", {C});
            if (!c.pe_rope) put(ab + "pos_encoder.pe", {1, c.num_frames, C});     // (rope: freqs_cis is not a buffer, motion_module.py:221-224)
            put(tb + "norms." + std::to_string(a) + ".weight", {C});
            put(tb + "norms." + std::to_string(a) + ".bias", {C});
        }
        put(tb + "ff.net.0.proj.weight", {8 * C, C});
        put(tb + "ff.net.0.proj.bias", {8 * C});
        put(tb + "ff.net.2.weight", {C, 4 * C});
        put(tb + "ff.net.2.bias", {C});
        put(tb + "ff_norm.weight", {C});
        put(tb + "ff_norm.bias", {C});
        put(t + "proj_out.weight", {C, C});
        put(t + "proj_out.bias", {C});
    }
}

int dev_alloc(vda_model* h, size_t bytes, void** out) {
    void* p = nullptr;
    VDA_HIP(hipMalloc(&p, bytes < 256 ? 256 : bytes));
    h->owned.push_back(p);
    *out = p;
    return 0;
}

int require_device(const vda_model* h, const char* what) {
    int dev = -1;
    VDA_HIP(hipGetDevice(&dev));
    if (dev != h->device) {
        vda_set_error("%s: the handle lives on device %d but device %d is current (one handle per device; make it current)", what, h->device, dev);
        return 1;
    }
    return 0;
}

// use_bn: BatchNorm2d in inference is y = (x - running_mean) / sqrt(running_var + 1e-5) * weight + bias per channel, directly behind a
// conv (util/blocks.py:80-81,85-86): folded into it, W'[co] = W[co] * s, b' = (b - running_mean) * s + bias, s = weight / sqrt(var + eps)
__global__ void __launch_bounds__(256) fold_bn_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ g,
                                                      const float* __restrict__ beta, const float* __restrict__ rm, const float* __restrict__ rv,
                                                      float* __restrict__ Wo, float* __restrict__ bo, int Co, int per_co) {
    const long long n = (long long)Co * per_co;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int co = (int)(i / per_co);
        const float sc = g[co] / sqrtf(rv[co] + 1e-5f);
        Wo[i] = W[i] * sc;
        if (i - (long long)co * per_co == 0) bo[co] = (b[co] - rm[co]) * sc + beta[co];
    }
}

// ---- packing of one precision (T = h16 / float); vectors are packed once with the first precision
template <typename T>
int pack_all(vda_model* h, int prec) {
    const vda_config& c = h->cfg;
    hipStream_t s = nullptr;
    auto raw = [&](const std::string& k) -> const float* { return h->raw.at(k).d; };
    auto mat = [&](const std::string& key, size_t elems, T** out) -> int {
        void* p = nullptr;
        VDA_TRY(dev_alloc(h, elems * sizeof(T), &p));
        h->mat[prec][key] = p;
        *out = (T*)p;
        return 0;
    };
    auto lin = [&](const std::string& key, const std::string& name, int N, int K, int Npad, int Kpad) -> int {
        T* d = nullptr;
        VDA_TRY(mat(key, (size_t)Npad * Kpad, &d));
        hipLaunchKernelGGL((pack_matrix_kernel<T>), dim3(grid_for((size_t)Npad * Kpad)), dim3(256), 0, s, raw(name), d, N, K, Npad, Kpad);
        return 0;
    };
    auto conv = [&](const std::string& key, const std::string& name, int Co, int Ci, int Copad, int Cipad) -> int {
        T* d = nullptr;
        VDA_TRY(mat(key, (size_t)Copad * 9 * Cipad, &d));
        hipLaunchKernelGGL((pack_conv3x3_kernel<T>), dim3(grid_for((size_t)Copad * 9 * Cipad)), dim3(256), 0, s, raw(name), d, Co, Ci, Copad, Cipad);
        return 0;
    };
    const bool vecs = h->vec.empty();
    auto vecp = [&](const std::string& key, const std::string& name, int n, int npad) -> int {
        if (!vecs) return 0;
        void* p = nullptr;
        VDA_TRY(dev_alloc(h, (size_t)npad * sizeof(float), &p));
        h->vec[key] = (float*)p;
        hipLaunchKernelGGL((pack_matrix_kernel<float>), dim3(grid_for(npad)), dim3(256), 0, s, raw(name), (float*)p, 1, n, 1, npad);
        return 0;
    };
    const int D = c.embed_dim, Fe = c.features;
    const int* oc = c.out_channels;
    const int* ocp = h->ocp;
    const int Fh = Fe / 2, Fhp = h->Fhp;
    const std::string p = "pretrained.";
    VDA_TRY(lin("patch.w", p + "patch_embed.proj.weight", D, 588, D, KPATCH));
    VDA_TRY(vecp("patch.b", p + "patch_embed.proj.bias", D, D));
    VDA_TRY(vecp("cls", p + "cls_token", D, D));
    for (int i = 0; i < c.depth; ++i) {
        const std::string b = p + "blocks." + std::to_string(i) + ".", k = "b" + std::to_string(i) + ".";
        VDA_TRY(vecp(k + "norm1.weight", b + "norm1.weight", D, D));
        VDA_TRY(vecp(k + "norm1.bias", b + "norm1.bias", D, D));
        VDA_TRY(vecp(k + "norm2.weight", b + "norm2.weight", D, D));
        VDA_TRY(vecp(k + "norm2.bias", b + "norm2.bias", D, D));
        VDA_TRY(vecp(k + "attn.qkv.bias", b + "attn.qkv.bias", 3 * D, 3 * D));
        VDA_TRY(vecp(k + "attn.proj.bias", b + "attn.proj.bias", D, D));
        VDA_TRY(vecp(k + "mlp.fc1.bias", b + "mlp.fc1.bias", 4 * D, 4 * D));
        VDA_TRY(vecp(k + "mlp.fc2.bias", b + "mlp.fc2.bias", D, D));
        VDA_TRY(vecp(k + "ls1.gamma", b + "ls1.gamma", D, D));
        VDA_TRY(vecp(k + "ls2.gamma", b + "ls2.gamma", D, D));
        VDA_TRY(lin(k + "attn.qkv.weight", b + "attn.qkv.weight", 3 * D, D, 3 * D, D));
        VDA_TRY(lin(k + "attn.proj.weight", b + "attn.proj.weight", D, D, D, D));
        VDA_TRY(lin(k + "mlp.fc1.weight", b + "mlp.fc1.weight", 4 * D, D, 4 * D, D));
        VDA_TRY(lin(k + "mlp.fc2.weight", b + "mlp.fc2.weight", D, 4 * D, D, 4 * D));
        if constexpr (std::is_same<T, h16>::value) {
            // LayerNorm folded into the Linear behind it (vda.h, VDA_EPI_LN_*): W * diag(ln_w) in fp16, c1 = its row sums, c2 = b + W.ln_b
            const char* pairs[2][3] = {{"attn.qkv", "norm1", nullptr}, {"mlp.fc1", "norm2", nullptr}};
            const int outs[2] = {3 * D, 4 * D};
            for (int j = 0; j < 2; ++j) {
                const std::string lk = pairs[j][0], nk = pairs[j][1];
                T* d = nullptr;
                VDA_TRY(mat(k + lk + ".weight.ln", (size_t)outs[j] * D, &d));
                void *c1 = nullptr, *c2 = nullptr;
                VDA_TRY(dev_alloc(h, (size_t)outs[j] * sizeof(float), &c1));
                VDA_TRY(dev_alloc(h, (size_t)outs[j] * sizeof(float), &c2));
                h->vec[k + lk + ".c1"] = (float*)c1;
                h->vec[k + lk + ".c2"] = (float*)c2;
                VDA_TRY(vda_fold_ln_weight(raw(b + lk + ".weight"), raw(b + lk + ".bias"), raw(b + nk + ".weight"), raw(b + nk + ".bias"), d, (float*)c1,
                                           (float*)c2, outs[j], D, s));
            }
            // the fused MLP kernel (vda_mlp_fused_f16; widths it is built for) takes fc2's weight with its hidden columns permuted
            if (vda_mlp_fused_supported(D, 4 * D)) {
                T* d = nullptr;
                VDA_TRY(mat(k + "mlp.fc2.weight.perm", (size_t)D * 4 * D, &d));
                VDA_TRY(vda_mlp_permute_w2_f16(h->mat[prec].at(k + "mlp.fc2.weight"), d, D, 4 * D, s));
            }
        }
    }
    VDA_TRY(vecp("norm.w", p + "norm.weight", D, D));
    VDA_TRY(vecp("norm.b", p + "norm.bias", D, D));
    const std::string hd = "head.";
    if (c.use_clstoken)
        for (int i = 0; i < 4; ++i) {
            const std::string n = hd + "readout_projects." + std::to_string(i) + ".0", k = "readout" + std::to_string(i);
            VDA_TRY(lin(k + ".w", n + ".weight", D, 2 * D, D, 2 * D));
            VDA_TRY(vecp(k + ".b", n + ".bias", D, D));
        }
    for (int i = 0; i < 4; ++i) {
        const std::string n = hd + "projects." + std::to_string(i), k = "proj" + std::to_string(i);
        VDA_TRY(lin(k + ".w", n + ".weight", oc[i], D, ocp[i], D));
        VDA_TRY(vecp(k + ".b", n + ".bias", oc[i], ocp[i]));
    }
    for (int i = 0; i < 2; ++i) {
        const int kk = i == 0 ? 4 : 2, cp = ocp[i];
        const std::string n = hd + "resize_layers." + std::to_string(i), k = "resize" + std::to_string(i);
        T* d = nullptr;
        VDA_TRY(mat(k + ".w", (size_t)kk * kk * cp * cp, &d));
        hipLaunchKernelGGL((pack_convt_kernel<T>), dim3(grid_for((size_t)kk * kk * cp * cp)), dim3(256), 0, s, raw(n + ".weight"), d, oc[i], oc[i], kk, cp);
        if (vecs) {
            void* bp = nullptr;
            VDA_TRY(dev_alloc(h, (size_t)kk * kk * cp * sizeof(float), &bp));
            h->vec[k + ".b"] = (float*)bp;
            hipLaunchKernelGGL(expand_convt_bias_kernel, dim3((kk * kk * cp + 255) / 256), dim3(256), 0, s, raw(n + ".bias"), (float*)bp, oc[i], kk * kk, cp);
        }
    }
    VDA_TRY(conv("resize3.w", hd + "resize_layers.3.weight", oc[3], oc[3], ocp[3], ocp[3]));
    VDA_TRY(vecp("resize3.b", hd + "resize_layers.3.bias", oc[3], ocp[3]));
    const std::string sc = hd + "scratch.";
    for (int i = 0; i < 4; ++i)
        VDA_TRY(conv("rn" + std::to_string(i + 1) + ".w", sc + "layer" + std::to_string(i + 1) + "_rn.weight", Fe, oc[i], Fe, ocp[i]));
    for (int i = 1; i <= 4; ++i) {
        const std::string r = sc + "refinenet" + std::to_string(i) + ".", k = "ref" + std::to_string(i) + ".";
        VDA_TRY(lin(k + "out.w", r + "out_conv.weight", Fe, Fe, Fe, Fe));
        VDA_TRY(vecp(k + "out.b", r + "out_conv.bias", Fe, Fe));
        for (int u = 1; u <= 2; ++u)
            for (int cc = 1; cc <= 2; ++cc) {
                const std::string rn = r + "resConfUnit" + std::to_string(u) + ".conv" + std::to_string(cc);
                const std::string kn = k + "rcu" + std::to_string(u) + ".c" + std::to_string(cc);
                if (c.use_bn) {
                    // the folded fp32 conv replaces the raw one for packing (kept under its own key: a re-pack in the other
                    // precision finds it again; a re-load of the state dict overwrites it)
                    const std::string bn = r + "resConfUnit" + std::to_string(u) + ".bn" + std::to_string(cc) + ".";
                    Raw& fw = h->raw[rn + ".weight.bn"];
                    Raw& fb = h->raw[rn + ".bias.bn"];
                    if (fw.d == nullptr) {
                        void *pw = nullptr, *pb = nullptr;
                        VDA_TRY(dev_alloc(h, (size_t)Fe * Fe * 9 * sizeof(float), &pw));
                        VDA_TRY(dev_alloc(h, (size_t)Fe * sizeof(float), &pb));
                        fw.d = (float*)pw, fw.n = (size_t)Fe * Fe * 9;
                        fb.d = (float*)pb, fb.n = Fe;
                    }
                    hipLaunchKernelGGL(fold_bn_kernel, dim3(grid_for((size_t)Fe * Fe * 9)), dim3(256), 0, s, raw(rn + ".weight"), raw(rn + ".bias"), raw(bn + "weight"),
                                       raw(bn + "bias"), raw(bn + "running_mean"), raw(bn + "running_var"), fw.d, fb.d, Fe, Fe * 9);
                    VDA_TRY(conv(kn + ".w", rn + ".weight.bn", Fe, Fe, Fe, Fe));
                    VDA_TRY(vecp(kn + ".b", rn + ".bias.bn", Fe, Fe));
                    continue;
                }
                VDA_TRY(conv(kn + ".w", rn + ".weight", Fe, Fe, Fe, Fe));
                VDA_TRY(vecp(kn + ".b", rn + ".bias", Fe, Fe));
            }
    }
    VDA_TRY(conv("oc1.w", sc + "output_conv1.weight", Fh, Fe, Fhp, Fe));
    VDA_TRY(vecp("oc1.b", sc + "output_conv1.bias", Fh, Fhp));
    VDA_TRY(conv("oc2.w", sc + "output_conv2.0.weight", 32, Fh, 32, Fhp));
    VDA_TRY(vecp("oc2.b", sc + "output_conv2.0.bias", 32, 32));
    VDA_TRY(vecp("oc3.w", sc + "output_conv2.2.weight", 32, 32));
    const int tc[4] = {ocp[2], ocp[3], Fe, Fe};
    for (int m = 0; m < 4; ++m) {
        const int Cc = tc[m];
        const std::string t = hd + "motion_modules." + std::to_string(m) + ".temporal_transformer.", k = "tm" + std::to_string(m) + ".";
        VDA_TRY(vecp(k + "gn.w", t + "norm.weight", Cc, Cc));
        VDA_TRY(vecp(k + "gn.b", t + "norm.bias", Cc, Cc));
        VDA_TRY(lin(k + "in.w", t + "proj_in.weight", Cc, Cc, Cc, Cc));
        VDA_TRY(vecp(k + "in.b", t + "proj_in.bias", Cc, Cc));
        VDA_TRY(lin(k + "out.w", t + "proj_out.weight", Cc, Cc, Cc, Cc));
        VDA_TRY(vecp(k + "out.b", t + "proj_out.bias", Cc, Cc));
        const std::string tb = t + "transformer_blocks.0.";
        for (int a = 0; a < 2; ++a) {
            const std::string ab = tb + "attention_blocks." + std::to_string(a) + ".", ka = k + "a" + std::to_string(a) + ".";
            T* d = nullptr;
            VDA_TRY(mat(ka + "qkv.w", (size_t)3 * Cc * Cc, &d));                 // to_q | to_k | to_v fused to one [3C, C]
            const char* names[3] = {"to_q.weight", "to_k.weight", "to_v.weight"};
            for (int j = 0; j < 3; ++j)
                hipLaunchKernelGGL((pack_matrix_kernel<T>), dim3(grid_for((size_t)Cc * Cc)), dim3(256), 0, s, raw(ab + names[j]),
                                   d + (size_t)j * Cc * Cc, Cc, Cc, Cc, Cc);
            VDA_TRY(lin(ka + "out.w", ab + "to_out.0.weight", Cc, Cc, Cc, Cc));
            VDA_TRY(vecp(ka + "out.b", ab + "to_out.0.bias", Cc, Cc));
            if (!c.pe_rope) VDA_TRY(vecp(ka + "pe", ab + "pos_encoder.pe", c.num_frames * Cc, c.num_frames * Cc));
            VDA_TRY(vecp(ka + "ln.w", tb + "norms." + std::to_string(a) + ".weight", Cc, Cc));
            VDA_TRY(vecp(ka + "ln.b", tb + "norms." + std::to_string(a) + ".bias", Cc, Cc));
        }
        VDA_TRY(vecp(k + "ffln.w", tb + "ff_norm.weight", Cc, Cc));
        VDA_TRY(vecp(k + "ffln.b", tb + "ff_norm.bias", Cc, Cc));
        {
            T* d = nullptr;
            VDA_TRY(mat(k + "ff1.w", (size_t)8 * Cc * Cc, &d));
            hipLaunchKernelGGL((pack_geglu_kernel<T>), dim3(grid_for((size_t)8 * Cc * Cc)), dim3(256), 0, s, raw(tb + "ff.net.0.proj.weight"), d, 4 * Cc, Cc);
            if (vecs) {
                void* bp = nullptr;
                VDA_TRY(dev_alloc(h, (size_t)8 * Cc * sizeof(float), &bp));
                h->vec[k + "ff1.b"] = (float*)bp;
                hipLaunchKernelGGL((pack_geglu_kernel<float>), dim3(grid_for((size_t)8 * Cc)), dim3(256), 0, s, raw(tb + "ff.net.0.proj.bias"), (float*)bp, 4 * Cc, 1);
            }
        }
        VDA_TRY(lin(k + "ff2.w", tb + "ff.net.2.weight", Cc, 4 * Cc, Cc, 4 * Cc));
        VDA_TRY(vecp(k + "ff2.b", tb + "ff.net.2.bias", Cc, Cc));
    }
    VDA_HIP(hipGetLastError());
    VDA_HIP(hipStreamSynchronize(s));
    h->packed[prec] = true;
    return 0;
}

// ------------------------------------------------------------------ one forward pass
struct Run {
    vda_model* h;
    int prec;                 // VDA_PREC_F16 / VDA_PREC_F32
    bool dry;                 // sizing pass: record buffer requests, launch nothing
    Layout* lay;
    hipStream_t s;
    size_t ab;                // bytes per activation element
    int32_t* sched = nullptr; // dynamic-schedule counters: 8 per GEMM launch of the forward, zeroed at its start
    int nsched = 0;           // launches so far (dry pass: the count that sizes the block)

    void* buf(const std::string& name, size_t elems, size_t esize) {
        const size_t bytes = (elems * esize + 255) & ~(size_t)255;
        if (dry) {
            auto it = lay->bufs.find(name);
            if (it == lay->bufs.end()) {
                lay->bufs[name] = {0, bytes};
                lay->order.push_back(name);
            } else if (it->second.second < bytes) {
                it->second.second = bytes;
            }
            return (void*)(uintptr_t)256;          // never dereferenced: every launch helper returns early in a dry pass
        }
        return (char*)h->ws + lay->bufs.at(name).first;
    }
    void* act(const std::string& name, size_t elems) { return buf(name, elems, ab); }
    float* f32(const std::string& name, size_t elems) { return (float*)buf(name, elems, 4); }
    // packed weights by key; a dry pass may run before this precision's pack exists and never dereferences them
    const void* W(const std::string& k) const { return dry ? nullptr : h->mat[prec].at(k); }
    const float* V(const std::string& k) const { return dry ? nullptr : h->vec.at(k); }

    // A large dense GEMM runs as two launches when that quantises better on this device (vda_gemm_plan_split: whole rounds of
    // 256-row tiles + the remainder on 192-row tiles). Split HERE rather than inside vda_gemm_f16 so that each launch has its own
    // dynamic-schedule counters and its own profile bracket (kernel name, FLOPs of its rows).
    int gemm(vda_gemm_args a) {
        if (prec == VDA_PREC_F16) {
            const int m1 = vda_gemm_plan_split(a.M, a.N, a.K, a.epilogue, a.a_mode);
            if (m1 < a.M) {
                if (dry) {
                    VDA_TRY(gemm_one(a));
                    return gemm_one(a);
                }
                vda_gemm_args p1, p2;
                a.zero_page = h->zero_page;
                VDA_TRY(vda_gemm_row_range(&a, 0, m1, &p1));
                VDA_TRY(vda_gemm_row_range(&a, m1, a.M - m1, &p2));
                p1.tile_rows = 256;
                p2.tile_rows = 192;
                VDA_TRY(gemm_one(p1));
                return gemm_one(p2);
            }
        }
        return gemm_one(a);
    }
    int gemm_one(vda_gemm_args a) {
        if (prec == VDA_PREC_F16 && h->dyn_sched) {
            if (!dry && sched != nullptr) a.sched = sched + 8 * nsched;
            ++nsched;
        }
        if (dry) return 0;
        a.zero_page = h->zero_page;
        if (a.lda == 0) a.lda = a.K;
        if (a.ldc == 0) a.ldc = a.N;
        Profile& pf = h->prof;
        if (pf.every <= 0) return prec == VDA_PREC_F32 ? vda_gemm_f32(&a, s) : vda_gemm_f16(&a, s);
        const std::array<int, 5> key = {a.M, a.N, a.K, a.epilogue, a.a_mode};
        const int n = pf.seen[key]++;
        const bool timed = n % pf.every == 0 && 2.0 * a.M * a.N * a.K >= pf.min_flops;
        ProfSample smp;
        if (timed) {
            VDA_HIP(hipEventCreate(&smp.e0));
            VDA_HIP(hipEventCreate(&smp.e1));
            VDA_HIP(hipEventRecord(smp.e0, s));
        }
        VDA_TRY(prec == VDA_PREC_F32 ? vda_gemm_f32(&a, s) : vda_gemm_f16(&a, s));
        const std::string name = prec == VDA_PREC_F32 ? std::string("gemm_f32_kernel") : std::string(vda_gemm_last_kernel());
        const double flops = 2.0 * a.M * a.N * a.K;
        auto& tot = pf.launches[name];
        tot.first += 1;
        tot.second += flops;
        if (timed) {
            VDA_HIP(hipEventRecord(smp.e1, s));
            smp.name = name;
            smp.flops = flops;
            pf.samples.push_back(smp);
        }
        return 0;
    }
    int dense(const void* A, const void* Wm, void* out, int epi, int M, int N, int K, const float* bias = nullptr, const void* res = nullptr,
              const float* gamma = nullptr, int ldc = 0) {
        vda_gemm_args a = {};
        a.A = A, a.W = Wm, a.out = out, a.bias = bias, a.res = res, a.gamma = gamma;
        a.M = M, a.N = N, a.K = K, a.ldc = ldc, a.a_mode = VDA_A_DENSE, a.epilogue = epi;
        return gemm(a);
    }
    int conv3x3(const void* x, const std::string& wname, void* out, int B, int H, int Wd, int Cin, int Cout, int epi, int stride,
                const float* bias, bool relu_in = false, const void* res = nullptr, const void* res2 = nullptr) {
        const int Ho = (H - 1) / stride + 1, Wo = (Wd - 1) / stride + 1;
        vda_gemm_args a = {};
        a.A = x, a.W = W(wname), a.out = out, a.bias = bias, a.res = res, a.res2 = res2;
        a.M = B * Ho * Wo, a.N = Cout, a.K = 9 * Cin, a.a_mode = VDA_A_CONV3X3, a.epilogue = epi, a.relu_in = relu_in ? 1 : 0;
        a.cB = B, a.cH = H, a.cW = Wd, a.cCin = Cin, a.cHo = Ho, a.cWo = Wo, a.cStride = stride;
        return gemm(a);
    }
    int layernorm(const float* x, void* out, const float* w, const float* b, float eps, int rows, int D, int group = 0, int skip = 0,
                  const float* pe = nullptr, int pe_rows = 0, int pe_steps = 0) {
        if (dry) return 0;
        return prec == VDA_PREC_F32 ? vda_layernorm_f32_f32(x, (float*)out, w, b, eps, rows, D, group, skip, pe, pe_rows, pe_steps, s)
                                    : vda_layernorm_f32_f16(x, out, w, b, eps, rows, D, group, skip, pe, pe_rows, pe_steps, s);
    }
    int bilinear(const void* x, void* out, int B, int hh, int ww, int H, int Wd, int C) {
        if (dry) return 0;
        return prec == VDA_PREC_F32 ? vda_bilinear_nhwc_f32((const float*)x, (float*)out, nullptr, B, hh, ww, H, Wd, C, s)
                                    : vda_bilinear_nhwc_f16(x, out, nullptr, B, hh, ww, H, Wd, C, s);
    }

    // motion_module.py:102-126,164-177 on token-major x [B*T*hw, C]; returns a new activation buffer
    int temporal(int m, const void* x, int B, int T, int hw, int Cc, const std::string& tag, void** result) {
        const std::string k = "tm" + std::to_string(m) + ".";
        const int BT = B * T, rows = BT * hw;
        int chunks = (hw + 31) / 32;
        chunks = chunks < 1 ? 1 : (chunks > 16 ? 16 : chunks);
        float* part = f32("gn_partial", (size_t)BT * chunks * GN_GROUPS * 2);
        void* g = act("tm_g", (size_t)rows * Cc);
        if (!dry)
            VDA_TRY(prec == VDA_PREC_F32 ? vda_groupnorm_nhwc_f32((const float*)x, (float*)g, V(k + "gn.w"), V(k + "gn.b"), GN_EPS, BT, hw, Cc, GN_GROUPS, part, chunks, s)
                                         : vda_groupnorm_nhwc_f16(x, g, V(k + "gn.w"), V(k + "gn.b"), GN_EPS, BT, hw, Cc, GN_GROUPS, part, chunks, s));
        float* hs = f32("tm_hs", (size_t)rows * Cc);
        VDA_TRY(dense(g, W(k + "in.w"), hs, VDA_EPI_BIAS_F32, rows, Cc, Cc, V(k + "in.b")));
        void* n = act("tm_n", (size_t)rows * Cc);
        void* qkv = act("tm_qkv", (size_t)rows * 3 * Cc);
        void* ao = act("tm_ao", (size_t)rows * Cc);
        for (int a = 0; a < 2; ++a) {
            const std::string ka = k + "a" + std::to_string(a) + ".";
            const bool rope = h->cfg.pe_rope != 0;                   // motion_module.py:221-224: no additive encoding, q and k rotated instead
            VDA_TRY(layernorm(hs, n, V(ka + "ln.w"), V(ka + "ln.b"), TMP_LN_EPS, rows, Cc, 0, 0, rope ? nullptr : V(ka + "pe"), hw, T));
            VDA_TRY(dense(n, W(ka + "qkv.w"), qkv, VDA_EPI_BIAS_F16, rows, 3 * Cc, Cc));
            for (int b = 0; b < B && !dry; ++b) {
                const size_t r0 = (size_t)b * T * hw;
                if (rope)
                    VDA_TRY(prec == VDA_PREC_F32 ? vda_rope_qk_f32((float*)qkv + r0 * 3 * Cc, T, hw, Cc, s) : vda_rope_qk_f16((h16*)qkv + r0 * 3 * Cc, T, hw, Cc, s));
                VDA_TRY(prec == VDA_PREC_F32
                            ? vda_temporal_attention_f32((const float*)qkv + r0 * 3 * Cc, (float*)ao + r0 * Cc, T, hw, Cc, TEMPORAL_HEADS, s)
                            : vda_temporal_attention_f16((const h16*)qkv + r0 * 3 * Cc, (h16*)ao + r0 * Cc, T, hw, Cc, TEMPORAL_HEADS, s));
            }
            VDA_TRY(dense(ao, W(ka + "out.w"), hs, VDA_EPI_SCALE_RES_F32, rows, Cc, Cc, V(ka + "out.b"), hs));
        }
        VDA_TRY(layernorm(hs, n, V(k + "ffln.w"), V(k + "ffln.b"), TMP_LN_EPS, rows, Cc));
        void* gg = act("tm_gg", (size_t)rows * 4 * Cc);
        VDA_TRY(dense(n, W(k + "ff1.w"), gg, VDA_EPI_GEGLU_F16, rows, 8 * Cc, Cc, V(k + "ff1.b"), nullptr, nullptr, 4 * Cc));
        void* hh = act("tm_hh", (size_t)rows * Cc);
        VDA_TRY(dense(gg, W(k + "ff2.w"), hh, VDA_EPI_SCALE_RES_F32_H, rows, Cc, 4 * Cc, V(k + "ff2.b"), hs));
        void* out = act(tag, (size_t)rows * Cc);
        VDA_TRY(dense(hh, W(k + "out.w"), out, VDA_EPI_RES_F16, rows, Cc, Cc, V(k + "out.b"), x));
        *result = out;
        return 0;
    }

    // util/blocks.py:68-91: conv2(relu(conv1(relu(x)))) + x (+ res2: the fusion block's skip add)
    int rcu(int i, int u, const void* x, void* out, int B, int H, int Wd, int Fe, const void* res2 = nullptr) {
        const std::string k = "ref" + std::to_string(i) + ".rcu" + std::to_string(u) + ".";
        void* y = act("rcu_y", (size_t)B * H * Wd * Fe);
        VDA_TRY(conv3x3(x, k + "c1.w", y, B, H, Wd, Fe, Fe, VDA_EPI_BIAS_RELU_F16, 1, V(k + "c1.b"), true));
        VDA_TRY(conv3x3(y, k + "c2.w", out, B, H, Wd, Fe, Fe, VDA_EPI_RES_F16, 1, V(k + "c2.b"), false, x, res2));
        return 0;
    }
    // util/blocks.py:135-162 with out_conv moved in front of the (commuting) bilinear resize
    // upsample = false: the caller's next conv does the resize itself (vda_conv3x3_up2_f16); *result is the out_conv output at H x Wd
    int fusion(int i, const void* x0, const void* x1, int B, int H, int Wd, int Ho, int Wo, int Fe, const std::string& tag, void** result,
               bool upsample = true) {
        const size_t rows = (size_t)B * H * Wd;
        const void* sm = x0;
        if (x1 != nullptr) {
            void* t = act("fus_s", rows * Fe);
            VDA_TRY(rcu(i, 1, x1, t, B, H, Wd, Fe, x0));
            sm = t;
        }
        void* r = act("fus_r", rows * Fe);
        VDA_TRY(rcu(i, 2, sm, r, B, H, Wd, Fe));
        void* c = act(upsample ? "fus_c" : tag, rows * Fe);
        const std::string k = "ref" + std::to_string(i) + ".out.";
        VDA_TRY(dense(r, W(k + "w"), c, VDA_EPI_BIAS_F16, (int)rows, Fe, Fe, V(k + "b")));
        if (!upsample) {
            *result = c;
            return 0;
        }
        void* out = act(tag, (size_t)B * Ho * Wo * Fe);
        VDA_TRY(bilinear(c, out, B, H, Wd, Ho, Wo, Fe));
        *result = out;
        return 0;
    }

    int forward(const float* x, float* depth, int B, int T, int H, int Wd) {
        const vda_config& c = h->cfg;
        const int BT = B * T, ph = H / PATCH, pw = Wd / PATCH;
        const int P = ph * pw, D = c.embed_dim, NH = c.num_heads, Nt = P + 1, rows = BT * Nt;
        // dynamic-schedule counters of every GEMM launch below: sized by the dry pass, zeroed here once per forward
        if (prec == VDA_PREC_F16 && h->dyn_sched) {
            const int cap = dry ? 4096 : lay->gemm_launches;
            sched = (int32_t*)buf("sched", (size_t)8 * (cap > 0 ? cap : 1), 4);
            if (!dry) VDA_HIP(hipMemsetAsync(sched, 0, (size_t)32 * lay->gemm_launches, s));
            nsched = 0;
        }
        // ---- encoder (dinov2.py:212-219, dinov2_layers/block.py:105-106)
        void* a0 = act("a0", (size_t)BT * P * KPATCH);
        if (!dry) {
            // the K padding (588 -> 640) must be zero: the buffer is shared with nothing else, but the block is caller memory
            VDA_HIP(hipMemsetAsync(a0, 0, (size_t)BT * P * KPATCH * ab, s));
            VDA_TRY(prec == VDA_PREC_F32 ? vda_patchify_f32_f32(x, (float*)a0, BT, H, Wd, KPATCH, s) : vda_patchify_f32_f16(x, a0, BT, H, Wd, KPATCH, s));
        }
        float* tok = f32("tok", (size_t)rows * D);
        const float* pos = dry ? nullptr : h->pos_cache.at({H, Wd});
        {
            vda_gemm_args a = {};
            a.A = a0, a.W = W("patch.w"), a.out = tok, a.bias = V("patch.b"), a.pos = pos;
            a.M = BT * P, a.N = D, a.K = KPATCH, a.a_mode = VDA_A_DENSE, a.epilogue = VDA_EPI_PATCH_F32, a.P = P;
            VDA_TRY(gemm(a));
        }
        if (!dry) VDA_TRY(vda_cls_rows_f32(tok, V("cls"), pos, BT, P, D, s));
        void* xn = act("xn", (size_t)rows * D);
        void* qkv = act("qkv", (size_t)rows * 3 * D);
        void* ao = act("ao", (size_t)rows * D);
        void* hid = nullptr;                     // (allocated below, unless the fused MLP kernel makes it unnecessary)
        void* taps[4] = {nullptr, nullptr, nullptr, nullptr};
        int ntap = 0;
        // Residual add of a block's two projections (attn.proj, mlp.fc2). Default: the GEMM's own fp32 in-place epilogue
        // (VDA_EPI_SCALE_RES_F32). Option residual_in_ln (fp16 path): the projection stores its bias-added output y as fp16 and
        // x += gamma * y rides on the LayerNorm that follows (vda_layernorm_residual_f32_f16). Measured on ViT-L 1x32x518x518: the
        // projection GEMMs get 15 % faster (1040 vs 902 TFLOP/s) but the LayerNorm doubles its bytes (12 B per element at
        // 5.05 TB/s): +1.1 ms per clip net (57.9 vs 56.8 ms). Kept as an A/B option, off.
        const bool defer = prec == VDA_PREC_F16 && h->residual_in_ln != 0;
        // Option ln_fold (fp16 path, default): no LayerNorm pass at all inside the encoder. The residual stream lives as two fp16
        // planes (x = hi + lo; VDA_EPI_SCALE_RES_SPLIT reads and writes both, 8 bytes per element like the fp32 stream, and leaves
        // per-row partial statistics); qkv / fc1 take the hi plane as their A operand with LayerNorm's affine folded into their
        // weights and apply rstd * (acc - mean * c1) + c2 in their epilogue (VDA_EPI_LN_*). Only the four taps still run a LayerNorm.
        const bool fold = prec == VDA_PREC_F16 && h->ln_fold != 0 && !defer && D % 64 == 0;
        // mlp.py:35-41 + block.py:106: the whole MLP branch in one kernel where vda_mlp_fused_f16 is built for the width (ViT-S): hid
        // stays in registers, one launch instead of two. An OPTION, off by default: in-process A/B it loses to the two launches.
        const bool mlp1 = fold && h->mlp_fused != 0 && vda_mlp_fused_supported(D, 4 * D) != 0;
        if (!mlp1) hid = act("hid", (size_t)rows * 4 * D);
        void* thi = fold ? buf("tok_hi", (size_t)rows * D, 2) : nullptr;
        void* tlo = fold ? buf("tok_lo", (size_t)rows * D, 2) : nullptr;
        float* lnpart = fold ? f32("ln_part", (size_t)rows * (D / 64) * 2) : nullptr;
        float* lnstat = fold ? f32("ln_stat", (size_t)rows * 2) : nullptr;
        // overflow flag of this forward (see vda_model::ovf_host): raised by vda_ln_stats_finalize when a row's statistics are not
        // finite, i.e. when a token moved further than 65 504 from its own mean and saturated the fp16 hi plane
        int32_t* ovf = fold ? (int32_t*)buf("overflow", 1, 4) : nullptr;
        if (fold && !dry) VDA_HIP(hipMemsetAsync(ovf, 0, 4, s));
        // The planes hold every token RELATIVE TO ITS OWN MEAN: taken out here, and again by every residual epilogue (a.pos below:
        // the mean the preceding LayerNorm statistics found), so the operand plane's fp16 rounding is relative to the token's
        // spread whatever offset the stream carries (tests/_outliers.py "offset": mean / sigma ~ 20 cost 15x the standalone
        // LayerNorm's error before this). Every reader of the stream is a LayerNorm - invariant to a per-row shift.
        if (fold && !dry) VDA_TRY(vda_split_center_stats_f32(tok, thi, tlo, lnstat, ENC_LN_EPS, rows, D, s));
        auto ln_gemm = [&](const std::string& wk, void* out, int epi, int N) -> int {         // LayerNorm(x) @ W^T + b on the hi plane
            vda_gemm_args a = {};
            a.A = thi, a.W = W(wk + ".weight.ln"), a.out = out, a.bias = V(wk + ".c2"), a.gamma = V(wk + ".c1"), a.stats = lnstat;
            a.M = rows, a.N = N, a.K = D, a.a_mode = VDA_A_DENSE, a.epilogue = epi;
            return gemm(a);
        };
        auto res_gemm = [&](const void* A, const std::string& wk, const std::string& gk, int K, bool stats_next) -> int {   // x += gamma * (A @ W^T + b)
            vda_gemm_args a = {};
            a.A = A, a.W = W(wk + ".weight"), a.out = thi, a.out2 = tlo, a.res = thi, a.res2 = tlo, a.bias = V(wk + ".bias"), a.gamma = V(gk);
            a.stats = lnpart;
            a.pos = lnstat;                      // re-centre by the mean the LayerNorm before this branch saw
            a.M = rows, a.N = D, a.K = K, a.a_mode = VDA_A_DENSE, a.epilogue = VDA_EPI_SCALE_RES_SPLIT;
            VDA_TRY(gemm(a));
            // (after the last block nothing reads the statistics: that finalize runs for its overflow check alone, 5 us per clip)
            (void)stats_next;
            if (!dry) VDA_TRY(vda_ln_stats_finalize(lnpart, lnstat, ENC_LN_EPS, rows, D / 64, ovf, s));
            return 0;
        };
        auto tap_ln = [&](void* out, int group, int skip) -> int {
            if (dry) return 0;
            return fold ? vda_layernorm_split_f16(thi, tlo, out, V("norm.w"), V("norm.b"), ENC_LN_EPS, rows, D, group, skip, s)
                        : vda_layernorm_f32_f16(tok, out, V("norm.w"), V("norm.b"), ENC_LN_EPS, rows, D, group, skip, nullptr, 0, 0, s);
        };
        void* yb = defer ? act("ybuf", (size_t)rows * D) : nullptr;
        auto ln_res = [&](const float* gamma, void* out, const float* w, const float* b, int group, int skip) -> int {
            if (dry) return 0;
            return vda_layernorm_residual_f32_f16(tok, yb, gamma, out, w, b, ENC_LN_EPS, rows, D, group, skip, s);
        };
        bool xn_ready = false;                 // norm1 of block i already ran (fused with block i-1's fc2 residual)
        // ---- the part of the head that needs taps 0..2 only (dpt_temporal.py:55-69 for i = 0, 1, 2; :75 motion module 0 on layer_3;
        // :78-80 layer{1,2,3}_rn): 3.8 of the head's 12.5 TFLOP at ViT-L. Option head_overlap runs it on a SIDE stream as soon as tap 2
        // exists, under the remaining encoder blocks (ViT-L: blocks 18-23), so that its kernels fill the tail rounds and epilogue
        // bursts of theirs - what two clips in flight do for a video, inside ONE clip. Same kernels, same arithmetic: bit-identical.
        const int* ocp = h->ocp;
        const int Fe = c.features, Fhp = h->Fhp;
        const int h1 = 4 * ph, w1 = 4 * pw, h2 = 2 * ph, w2 = 2 * pw, h4 = (ph - 1) / 2 + 1, w4 = (pw - 1) / 2 + 1;
        void *l1r = nullptr, *l2r = nullptr, *l3r = nullptr;
        auto head_early = [&]() -> int {
            void* t0 = act("t0", (size_t)BT * P * ocp[0]);
            VDA_TRY(dense(taps[0], W("proj0.w"), t0, VDA_EPI_BIAS_F16, BT * P, ocp[0], D, V("proj0.b")));
            void* l1 = act("l1", (size_t)BT * h1 * w1 * ocp[0]);
            {
                vda_gemm_args a = {};
                a.A = t0, a.W = W("resize0.w"), a.out = l1, a.bias = V("resize0.b");
                a.M = BT * P, a.N = 16 * ocp[0], a.K = ocp[0], a.ldc = ocp[0], a.a_mode = VDA_A_DENSE, a.epilogue = VDA_EPI_CONVT_F16;
                a.tK = 4, a.tH = ph, a.tW = pw, a.tCout = ocp[0];
                VDA_TRY(gemm(a));
            }
            void* t1 = act("t1", (size_t)BT * P * ocp[1]);
            VDA_TRY(dense(taps[1], W("proj1.w"), t1, VDA_EPI_BIAS_F16, BT * P, ocp[1], D, V("proj1.b")));
            void* l2 = act("l2", (size_t)BT * h2 * w2 * ocp[1]);
            {
                vda_gemm_args a = {};
                a.A = t1, a.W = W("resize1.w"), a.out = l2, a.bias = V("resize1.b");
                a.M = BT * P, a.N = 4 * ocp[1], a.K = ocp[1], a.ldc = ocp[1], a.a_mode = VDA_A_DENSE, a.epilogue = VDA_EPI_CONVT_F16;
                a.tK = 2, a.tH = ph, a.tW = pw, a.tCout = ocp[1];
                VDA_TRY(gemm(a));
            }
            void* l3 = act("l3", (size_t)BT * P * ocp[2]);
            VDA_TRY(dense(taps[2], W("proj2.w"), l3, VDA_EPI_BIAS_F16, BT * P, ocp[2], D, V("proj2.b")));
            void* l3t = nullptr;
            VDA_TRY(temporal(0, l3, B, T, P, ocp[2], "l3t", &l3t));                  // dpt_temporal.py:75
            l1r = act("l1r", (size_t)BT * h1 * w1 * Fe);
            VDA_TRY(conv3x3(l1, "rn1.w", l1r, BT, h1, w1, ocp[0], Fe, VDA_EPI_BIAS_F16, 1, nullptr));
            l2r = act("l2r", (size_t)BT * h2 * w2 * Fe);
            VDA_TRY(conv3x3(l2, "rn2.w", l2r, BT, h2, w2, ocp[1], Fe, VDA_EPI_BIAS_F16, 1, nullptr));
            l3r = act("l3r", (size_t)BT * P * Fe);
            VDA_TRY(conv3x3(l3t, "rn3.w", l3r, BT, ph, pw, ocp[2], Fe, VDA_EPI_BIAS_F16, 1, nullptr));
            return 0;
        };
        bool early_done = false;
        vda_model::Side* side = nullptr;
        for (int i = 0; i < c.depth; ++i) {
            const std::string k = "b" + std::to_string(i) + ".";
            if (fold) {
                VDA_TRY(ln_gemm(k + "attn.qkv", qkv, VDA_EPI_LN_BIAS_F16, 3 * D));
            } else {
                if (!xn_ready) VDA_TRY(layernorm(tok, xn, V(k + "norm1.weight"), V(k + "norm1.bias"), ENC_LN_EPS, rows, D));
                xn_ready = false;
                VDA_TRY(dense(xn, W(k + "attn.qkv.weight"), qkv, VDA_EPI_BIAS_F16, rows, 3 * D, D, V(k + "attn.qkv.bias")));
            }
            if (!dry) VDA_TRY(prec == VDA_PREC_F32 ? vda_attention_f32((const float*)qkv, (float*)ao, BT, Nt, NH, s) : vda_attention_f16(qkv, ao, BT, Nt, NH, s));
            if (fold) {
                VDA_TRY(res_gemm(ao, k + "attn.proj", k + "ls1.gamma", D, true));
            } else if (defer) {
                VDA_TRY(dense(ao, W(k + "attn.proj.weight"), yb, VDA_EPI_BIAS_F16, rows, D, D, V(k + "attn.proj.bias")));
                VDA_TRY(ln_res(V(k + "ls1.gamma"), xn, V(k + "norm2.weight"), V(k + "norm2.bias"), 0, 0));
            } else {
                VDA_TRY(dense(ao, W(k + "attn.proj.weight"), tok, VDA_EPI_SCALE_RES_F32, rows, D, D, V(k + "attn.proj.bias"), tok, V(k + "ls1.gamma")));
                VDA_TRY(layernorm(tok, xn, V(k + "norm2.weight"), V(k + "norm2.bias"), ENC_LN_EPS, rows, D));
            }
            if (mlp1) {
                if (!dry)
                    VDA_TRY(vda_mlp_fused_f16(thi, lnstat, W(k + "mlp.fc1.weight.ln"), V(k + "mlp.fc1.c1"), V(k + "mlp.fc1.c2"), W(k + "mlp.fc2.weight.perm"),
                                              V(k + "mlp.fc2.bias"), V(k + "ls2.gamma"), thi, tlo, lnpart, rows, D, 4 * D, rows, s));
            } else if (fold) VDA_TRY(ln_gemm(k + "mlp.fc1", hid, VDA_EPI_LN_GELU_F16, 4 * D));
            else VDA_TRY(dense(xn, W(k + "mlp.fc1.weight"), hid, VDA_EPI_BIAS_GELU_F16, rows, 4 * D, D, V(k + "mlp.fc1.bias")));
            bool is_tap = false;
            for (int t = 0; t < 4; ++t) is_tap = is_tap || c.taps[t] == i;
            const bool last = i + 1 == c.depth;
            void* tp = (is_tap && ntap < 4) ? act("tap" + std::to_string(ntap), (size_t)BT * P * D) : nullptr;
            if (fold) {
                if (mlp1) {
                    if (!dry) VDA_TRY(vda_ln_stats_finalize(lnpart, lnstat, ENC_LN_EPS, rows, D / 64, ovf, s));
                } else {
                    VDA_TRY(res_gemm(hid, k + "mlp.fc2", k + "ls2.gamma", 4 * D, !last));
                }
                if (tp != nullptr) VDA_TRY(tap_ln(tp, Nt, 1));                                             // final norm, cls dropped
            } else if (defer) {
                VDA_TRY(dense(hid, W(k + "mlp.fc2.weight"), yb, VDA_EPI_BIAS_F16, rows, D, 4 * D, V(k + "mlp.fc2.bias")));
                if (last && tp != nullptr) {
                    VDA_TRY(ln_res(V(k + "ls2.gamma"), tp, V("norm.w"), V("norm.b"), Nt, 1));            // residual + final norm, cls dropped
                } else {
                    const std::string kn = "b" + std::to_string(last ? i : i + 1) + ".";                   // residual + the next block's norm1
                    VDA_TRY(ln_res(V(k + "ls2.gamma"), xn, V(kn + "norm1.weight"), V(kn + "norm1.bias"), 0, 0));
                    xn_ready = true;
                    if (tp != nullptr) VDA_TRY(layernorm(tok, tp, V("norm.w"), V("norm.b"), ENC_LN_EPS, rows, D, Nt, 1));
                }
            } else {
                VDA_TRY(dense(hid, W(k + "mlp.fc2.weight"), tok, VDA_EPI_SCALE_RES_F32, rows, D, 4 * D, V(k + "mlp.fc2.bias"), tok, V(k + "ls2.gamma")));
                if (tp != nullptr) VDA_TRY(layernorm(tok, tp, V("norm.w"), V("norm.b"), ENC_LN_EPS, rows, D, Nt, 1));     // final norm, cls dropped
            }
            if (tp != nullptr) {
                if (c.use_clstoken) {
                    // dpt_temporal.py:56-59: the tap becomes GELU(Linear([patch token, cls])); the final norm above dropped the cls
                    // row, so norm the whole token matrix again (cls kept) and gather [patch | cls] rows for one K = 2D GEMM
                    void* full = act("rd_full", (size_t)rows * D);
                    VDA_TRY(fold ? tap_ln(full, 0, 0) : layernorm(tok, full, V("norm.w"), V("norm.b"), ENC_LN_EPS, rows, D));
                    void* cat = act("rd_cat", (size_t)BT * P * 2 * D);
                    if (!dry)
                        VDA_TRY(prec == VDA_PREC_F32 ? vda_readout_concat_f32((const float*)full, (float*)cat, BT, P, D, s)
                                                     : vda_readout_concat_f16(full, cat, BT, P, D, s));
                    const std::string kr = "readout" + std::to_string(ntap);
                    VDA_TRY(dense(cat, W(kr + ".w"), tp, VDA_EPI_BIAS_GELU_F16, BT * P, D, 2 * D, V(kr + ".b")));
                }
                taps[ntap++] = tp;
                if (ntap == 3 && !last && !early_done) {
                    if (dry || !h->head_overlap) {
                        // (dry pass: the buffers are requested here so that both orders of execution find them laid out)
                        if (dry) {
                            VDA_TRY(head_early());
                            early_done = true;
                        }
                    } else {
                        vda_model::Side& sd = h->sides[s];
                        if (sd.stream == nullptr) {
                            VDA_HIP(hipStreamCreateWithFlags(&sd.stream, hipStreamNonBlocking));
                            VDA_HIP(hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming));
                            VDA_HIP(hipEventCreateWithFlags(&sd.join, hipEventDisableTiming));
                        }
                        side = &sd;
                        VDA_HIP(hipEventRecord(sd.fork, s));               // taps 0..2 are behind this point of the caller's stream
                        VDA_HIP(hipStreamWaitEvent(sd.stream, sd.fork, 0));
                        hipStream_t main_stream = s;
                        s = sd.stream;
                        const int rc = head_early();
                        s = main_stream;
                        if (rc != 0) return rc;
                        VDA_HIP(hipEventRecord(sd.join, sd.stream));
                        early_done = true;
                    }
                }
            }
        }
        if (ntap != 4) {
            vda_set_error("vda_forward: the configuration's taps are not four distinct block indices below depth");
            return 1;
        }
        // ---- head: reassemble (dpt_temporal.py:55-69), temporal modules on layer_3 / layer_4 (:75-76), layer_rn (:78-81)
        if (!early_done) VDA_TRY(head_early());
        else if (side != nullptr) VDA_HIP(hipStreamWaitEvent(s, side->join, 0));     // the side stream's part is done before anything reads it
        void* t3 = act("t3", (size_t)BT * P * ocp[3]);
        VDA_TRY(dense(taps[3], W("proj3.w"), t3, VDA_EPI_BIAS_F16, BT * P, ocp[3], D, V("proj3.b")));
        void* l4 = act("l4", (size_t)BT * h4 * w4 * ocp[3]);
        VDA_TRY(conv3x3(t3, "resize3.w", l4, BT, ph, pw, ocp[3], ocp[3], VDA_EPI_BIAS_F16, 2, V("resize3.b")));
        void* l4t = nullptr;
        VDA_TRY(temporal(1, l4, B, T, h4 * w4, ocp[3], "l4t", &l4t));
        void* l4r = act("l4r", (size_t)BT * h4 * w4 * Fe);
        VDA_TRY(conv3x3(l4t, "rn4.w", l4r, BT, h4, w4, ocp[3], Fe, VDA_EPI_BIAS_F16, 1, nullptr));
        void *p4 = nullptr, *p4t = nullptr, *p3 = nullptr, *p3t = nullptr, *p2 = nullptr, *p1 = nullptr;
        VDA_TRY(fusion(4, l4r, nullptr, BT, h4, w4, ph, pw, Fe, "p4", &p4));
        VDA_TRY(temporal(2, p4, B, T, P, Fe, "p4t", &p4t));
        VDA_TRY(fusion(3, p4t, l3r, BT, ph, pw, h2, w2, Fe, "p3", &p3));
        VDA_TRY(temporal(3, p3, B, T, h2 * w2, Fe, "p3t", &p3t));
        VDA_TRY(fusion(2, p3t, l2r, BT, h2, w2, h1, w1, Fe, "p2", &p2));
        // refinenet1's 2x upsample (util/blocks.py:156-160) is folded into output_conv1 on the fp16 path: path_1 at 8 ph x 8 pw (1.4 GB
        // per ViT-L clip) never exists; "p1c" is refinenet1's out_conv output at 4 ph x 4 pw (bilinear and the 1x1 conv commute)
        const bool up_fused = prec == VDA_PREC_F16 && h->oc1_fused != 0 && Fhp <= 128 && Fe % 16 == 0;
        VDA_TRY(fusion(1, p2, l1r, BT, h1, w1, 2 * h1, 2 * w1, Fe, up_fused ? "p1c" : "p1", &p1, !up_fused));
        // ---- output convs (dpt.py:117-124, dpt_temporal.py:93-100)
        const int hh = 2 * h1, ww = 2 * w1;
        void* o1 = act("o1", (size_t)BT * hh * ww * Fhp);
        if (up_fused) {
            if (!dry) VDA_TRY(vda_conv3x3_up2_f16(p1, W("oc1.w"), V("oc1.b"), o1, BT, h1, w1, Fe, Fhp, Fhp, s));
        } else
            VDA_TRY(conv3x3(p1, "oc1.w", o1, BT, hh, ww, Fe, Fhp, VDA_EPI_BIAS_F16, 1, V("oc1.b")));
        if (prec == VDA_PREC_F16) {
            // bilinear to (H,W) + output_conv2 (3x3 -> ReLU -> 1x1 -> ReLU) fused: the upsampled tensor never exists
            if (!dry) VDA_TRY(vda_depth_tail_f16(o1, W("oc2.w"), V("oc2.b"), V("oc3.w"), h->oc3_bias, depth, h->zero_page, BT, hh, ww, H, Wd, Fhp, s));
        } else {
            // fp32 operands: the same three steps unfused (speed is secondary on this path; 288 GB of HBM hold the 518^2 tensor)
            void* up = act("tail_up", (size_t)BT * H * Wd * Fhp);
            VDA_TRY(bilinear(o1, up, BT, hh, ww, H, Wd, Fhp));
            void* c2 = act("tail_c2", (size_t)BT * H * Wd * 32);
            VDA_TRY(conv3x3(up, "oc2.w", c2, BT, H, Wd, Fhp, 32, VDA_EPI_BIAS_RELU_F16, 1, V("oc2.b")));
            if (!dry) VDA_TRY(vda_head_out_f32_f32((const float*)c2, V("oc3.w"), h->oc3_bias, depth, (long long)BT * H * Wd, 32, s));
        }
        // video_depth.py:162-163: bilinear to (H,W) is the identity here (H == 14*ph) and the final ReLU is idempotent.
        if (fold && !dry) {
            hipLaunchKernelGGL(poison_on_overflow_kernel, dim3(256), dim3(256), 0, s, (const int*)ovf, depth, (long long)BT * H * Wd, (volatile int*)h->ovf_host);
            VDA_LAUNCH_CHECK();
        }
        return 0;
    }
};

int get_layout(vda_model* h, int B, int T, int H, int W, int prec, Layout** out) {
    const std::array<int, 5> key = {B, T, H, W, prec};
    auto it = h->layouts.find(key);
    if (it == h->layouts.end()) {
        Layout lay;
        Run r{h, prec, true, &lay, nullptr, prec == VDA_PREC_F32 ? (size_t)4 : (size_t)2};
        VDA_TRY(r.forward(nullptr, nullptr, B, T, H, W));
        lay.gemm_launches = r.nsched;
        size_t off = 0;
        for (const std::string& n : lay.order) {
            lay.bufs[n].first = off;
            off += lay.bufs[n].second;
        }
        lay.total = off;
        it = h->layouts.emplace(key, std::move(lay)).first;
    }
    *out = &it->second;
    return 0;
}

int check_shape(const vda_model* h, int B, int T, int H, int W, int prec) {
    VDA_REQUIRE(prec == VDA_PREC_F16 || prec == VDA_PREC_F32, "vda_forward: precision %d (VDA_PREC_F16 = 0, VDA_PREC_F32 = 1)", prec);
    VDA_REQUIRE(B > 0 && T > 0, "vda_forward: empty clip B=%d T=%d", B, T);
    VDA_REQUIRE(H > 0 && H % PATCH == 0, "Input image height %d is not a multiple of patch height %d", H, PATCH);       // patch_embed.py:73
    VDA_REQUIRE(W > 0 && W % PATCH == 0, "Input image width %d is not a multiple of patch width: %d", W, PATCH);         // patch_embed.py:74
    VDA_REQUIRE(T <= h->cfg.num_frames, "vda_forward: T=%d exceeds temporal_max_len=%d", T, h->cfg.num_frames);
    VDA_REQUIRE((long long)B * T * ((H / PATCH) * (W / PATCH) + 1) * 4 * h->cfg.embed_dim < (1ll << 31), "vda_forward: clip too large for 32-bit row offsets (B*T*tokens*4*embed_dim)");
    return 0;
}

}  // namespace

extern "C" int vda_create(const vda_config* cfg, vda_model** out) {
    VDA_REQUIRE(cfg && out, "vda_create: null argument");
    VDA_REQUIRE(cfg->embed_dim > 0 && cfg->embed_dim % 64 == 0 && cfg->num_heads > 0 && cfg->embed_dim == cfg->num_heads * 64,
                "vda_create: embed_dim=%d must be num_heads=%d x 64 (the attention kernels are built for head dim 64)", cfg->embed_dim, cfg->num_heads);
    VDA_REQUIRE(cfg->depth > 0 && cfg->features > 0 && cfg->features % 64 == 0, "vda_create: depth=%d features=%d (features must be a multiple of 64)", cfg->depth, cfg->features);
    VDA_REQUIRE(cfg->num_frames > 0 && cfg->num_frames <= 32, "vda_create: num_frames=%d must be in 1..32", cfg->num_frames);
    VDA_REQUIRE((cfg->use_clstoken | cfg->use_bn | cfg->pe_rope) >> 1 == 0, "vda_create: use_clstoken / use_bn / pe_rope are 0 or 1");
    for (int i = 0; i < 4; ++i) {
        VDA_REQUIRE(cfg->out_channels[i] > 0 && cfg->out_channels[i] % 8 == 0, "vda_create: out_channels[%d]=%d must be a positive multiple of 8", i, cfg->out_channels[i]);
        VDA_REQUIRE(cfg->taps[i] >= 0 && cfg->taps[i] < cfg->depth && (i == 0 || cfg->taps[i] > cfg->taps[i - 1]), "vda_create: taps must be increasing block indices below depth");
    }
    VDA_REQUIRE(cfg->out_channels[2] % 64 == 0 && cfg->out_channels[3] % 64 == 0,
                "vda_create: out_channels[2..3] carry the temporal modules (GroupNorm over the true width): multiples of 64 only");
    vda_model* h = new vda_model();
    h->cfg = *cfg;
    if (hipGetDevice(&h->device) != hipSuccess) {
        delete h;
        vda_set_error("vda_create: no HIP device (this library has no CPU path)");
        return 2;
    }
    for (int i = 0; i < 4; ++i) h->ocp[i] = pad64(cfg->out_channels[i]);
    h->Fhp = (cfg->features / 2 + 31) / 32 * 32;      // output_conv1's width: only the depth tail (32-channel passes) consumes it
    build_spec(h->cfg, h->spec);
    if (dev_alloc(h, 256, &h->zero_page) != 0 || hipMemset(h->zero_page, 0, 256) != hipSuccess) {
        for (void* p : h->owned) (void)hipFree(p);
        delete h;
        vda_set_error("vda_create: device allocation failed");
        return 2;
    }
    void* ring = nullptr;
    if (hipHostMalloc(&ring, 64, hipHostMallocDefault) != hipSuccess) {
        for (void* p : h->owned) (void)hipFree(p);
        delete h;
        vda_set_error("vda_create: pinned host allocation failed");
        return 2;
    }
    memset(ring, 0, 64);
    h->ovf_host = (volatile int32_t*)ring;
    *out = h;
    return 0;
}

extern "C" int vda_destroy(vda_model* h) {
    if (h == nullptr) return 0;
    if (h->ovf_host) (void)hipHostFree((void*)h->ovf_host);
    for (auto& kv : h->sides) {
        if (kv.second.fork) (void)hipEventDestroy(kv.second.fork);
        if (kv.second.join) (void)hipEventDestroy(kv.second.join);
        if (kv.second.stream) (void)hipStreamDestroy(kv.second.stream);
    }
    for (void* p : h->owned) (void)hipFree(p);
    delete h;
    return 0;
}

extern "C" int vda_num_weights(const vda_model* h) { return h ? (int)h->spec.size() : 0; }

static int vda_load_weight_impl(vda_model* h, const char* name, const void* ptr, const int64_t* dims, int ndim, int dtype) {
    VDA_REQUIRE(h && name && ptr && (dims || ndim == 0), "vda_load_weight: null argument");
    VDA_REQUIRE(dtype == VDA_DTYPE_F32, "vda_load_weight(%s): dtype %d (only VDA_DTYPE_F32 = 0 checkpoints are defined, run.py:46)", name, dtype);
    VDA_TRY(require_device(h, "vda_load_weight"));
    auto it = h->spec.find(name);
    VDA_REQUIRE(it != h->spec.end(), "Unexpected key(s) in state_dict: \"%s\". ", name);
    const std::vector<int64_t>& want = it->second;
    bool same = (size_t)ndim == want.size();
    for (int i = 0; same && i < ndim; ++i) same = dims[i] == want[i];
    if (!same) {
        std::string got = "(", exp = "(";
        for (int i = 0; i < ndim; ++i) got += (i ? ", " : "") + std::to_string(dims[i]);
        for (size_t i = 0; i < want.size(); ++i) exp += (i ? ", " : "") + std::to_string(want[i]);
        vda_set_error("size mismatch for %s: copying a param with shape %s) from checkpoint, the shape in current model is %s).", name, got.c_str(), exp.c_str());
        return 1;
    }
    size_t n = 1;
    for (int64_t d : want) n *= (size_t)d;
    Raw& r = h->raw[name];
    if (r.d == nullptr) {
        void* p = nullptr;
        VDA_TRY(dev_alloc(h, n * sizeof(float), &p));
        r.d = (float*)p;
        r.dims = want;
        r.n = n;
    }
    VDA_HIP(hipMemcpy(r.d, ptr, n * sizeof(float), hipMemcpyDefault));       // host or device source
    h->finalized = false;
    return 0;
}

static int vda_finalize_weights_impl(vda_model* h) {
    VDA_REQUIRE(h, "vda_finalize_weights: null handle");
    VDA_TRY(require_device(h, "vda_finalize_weights"));
    std::string missing;
    for (const auto& kv : h->spec)
        if (h->raw.find(kv.first) == h->raw.end()) missing += (missing.empty() ? "\"" : ", \"") + kv.first + "\"";
    if (!missing.empty()) {
        if (missing.size() > 400) missing = missing.substr(0, 400) + " ...";
        vda_set_error("Missing key(s) in state_dict: %s. ", missing.c_str());
        return 1;
    }
    // a re-load replaces every layout: drop the packed copies (their memory stays owned by the handle until vda_destroy)
    h->mat[0].clear();
    h->mat[1].clear();
    h->vec.clear();
    h->pos_cache.clear();
    h->packed[0] = h->packed[1] = false;
    VDA_HIP(hipMemcpy(&h->oc3_bias, h->raw.at("head.scratch.output_conv2.2.bias").d, sizeof(float), hipMemcpyDeviceToHost));
    VDA_TRY(pack_all<h16>(h, VDA_PREC_F16));
    h->finalized = true;
    return 0;
}

// Everything a forward of this shape / precision needs beyond the workspace: the fp32 weight pack (first fp32 use) and the
// positional embedding at this grid (dinov2.py:179-210). Called by vda_forward; callable up front to keep the first
// forward allocation-free.
static int vda_prepare_impl(vda_model* h, int B, int T, int H, int W, int precision) {
    VDA_REQUIRE(h && h->finalized, "vda_prepare: load every weight and call vda_finalize_weights first");
    VDA_TRY(require_device(h, "vda_prepare"));
    VDA_TRY(check_shape(h, B, T, H, W, precision));
    if (!h->packed[precision]) VDA_TRY(pack_all<float>(h, VDA_PREC_F32));
    const std::pair<int, int> key = {H, W};
    if (h->pos_cache.find(key) == h->pos_cache.end()) {
        const int ph = H / PATCH, pw = W / PATCH, D = h->cfg.embed_dim;
        const float* pe = h->raw.at("pretrained.pos_embed").d;
        void* p = nullptr;
        VDA_TRY(dev_alloc(h, (size_t)(1 + ph * pw) * D * sizeof(float), &p));
        if (ph * pw == POS_GRID * POS_GRID && H == W) {
            // dinov2.py:183-184. A device-to-device hipMemcpy may return before the copy has run, and the first forward reads the
            // cache on the caller's (non-blocking) stream: complete it here, as the resample branch does
            VDA_HIP(hipMemcpyAsync(p, pe, (size_t)(1 + ph * pw) * D * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
        } else {
            VDA_TRY(vda_pos_embed_resample_f32(pe, (float*)p, POS_GRID, ph, pw, D, nullptr));
        }
        VDA_HIP(hipStreamSynchronize(nullptr));
        h->pos_cache[key] = (float*)p;
    }
    Layout* lay = nullptr;
    return get_layout(h, B, T, H, W, precision, &lay);
}

static int64_t vda_workspace_bytes_impl(vda_model* h, int B, int T, int H, int W, int precision) {
    if (h == nullptr || !h->finalized) {
        vda_set_error("vda_workspace_bytes: load every weight and call vda_finalize_weights first");
        return -1;
    }
    if (check_shape(h, B, T, H, W, precision) != 0) return -1;
    Layout* lay = nullptr;
    if (get_layout(h, B, T, H, W, precision, &lay) != 0) return -1;
    return (int64_t)lay->total;
}

extern "C" int vda_set_workspace(vda_model* h, void* ptr, int64_t bytes) {
    VDA_REQUIRE(h, "vda_set_workspace: null handle");
    VDA_REQUIRE(((uintptr_t)ptr & 255) == 0, "vda_set_workspace: the block must be 256-byte aligned");
    h->ws = ptr;
    h->ws_bytes = ptr ? bytes : 0;
    h->ws_owned = false;
    return 0;
}

// Overflow reports that have ARRIVED (their forward's stream work has completed): collected and cleared. A word still in flight
// reads 0 and is seen by a later call. Status 4 + a message naming the way out.
static int collect_overflow(vda_model* h, const char* who) {
    if (h->ovf_host[0] == 0) return 0;
    h->ovf_host[0] = 0;
    vda_set_error("%s: in a forward since the last check the split residual stream left fp16's range (a token further than 65504 from its own "
                  "mean): that forward's depth was overwritten with NaN. vda_set_option(h, \"ln_fold\", 0) keeps the stream in fp32 (the reference's "
                  "form, no such limit); fp32=True avoids it as well", who);
    return 4;
}

extern "C" int vda_forward_status(vda_model* h) {
    VDA_REQUIRE(h != nullptr, "vda_forward_status: null handle");
    return collect_overflow(h, "vda_forward_status");
}

static int vda_forward_impl(vda_model* h, const float* in, float* out, int B, int T, int H, int W, int precision, vda_stream_t stream) {
    VDA_REQUIRE(h && in && out, "vda_forward: null argument");
    VDA_REQUIRE(h->finalized, "vda_forward: load every weight and call vda_finalize_weights first");
    VDA_TRY(collect_overflow(h, "vda_forward"));          // an EARLIER forward's report, if nobody asked (never silent)
    VDA_TRY(vda_prepare(h, B, T, H, W, precision));
    Layout* lay = nullptr;
    VDA_TRY(get_layout(h, B, T, H, W, precision, &lay));
    if (h->ws == nullptr || (h->ws_owned && (size_t)h->ws_bytes < lay->total)) {
        // no caller block: the handle keeps one of its own, grown to the largest shape seen
        void* p = nullptr;
        VDA_HIP(hipMalloc(&p, lay->total));
        if (h->ws_owned && h->ws) {
            VDA_HIP(hipDeviceSynchronize());
            (void)hipFree(h->ws);
            for (auto& q : h->owned)
                if (q == h->ws) q = p;
        } else {
            h->owned.push_back(p);
        }
        h->ws = p;
        h->ws_bytes = (int64_t)lay->total;
        h->ws_owned = true;
    }
    VDA_REQUIRE((size_t)h->ws_bytes >= lay->total, "vda_forward: workspace of %lld bytes, this shape needs %lld (vda_workspace_bytes)",
                (long long)h->ws_bytes, (long long)lay->total);
    Run r{h, precision, false, lay, (hipStream_t)stream, precision == VDA_PREC_F32 ? (size_t)4 : (size_t)2};
    h->last_key = {B, T, H, W, precision};
    return r.forward(in, out, B, T, H, W);
}

// Debug / parity hook: copy a named intermediate of the LAST forward (same workspace) into `dst`: "tap0".."tap3" (final-norm'd
// patch tokens), "l1", "l2", "l3t", "l4t" (reassembled + temporal layers), "p4t", "p3t", "p2", "p1" (fusion pyramid).
static int vda_debug_copy_impl(vda_model* h, const char* name, void* dst, int64_t bytes, vda_stream_t stream) {
    VDA_REQUIRE(h && name && dst, "vda_debug_copy: null argument");
    auto it = h->layouts.find(h->last_key);
    VDA_REQUIRE(it != h->layouts.end() && h->ws, "vda_debug_copy: no forward has run");
    auto b = it->second.bufs.find(name);
    VDA_REQUIRE(b != it->second.bufs.end(), "vda_debug_copy: no buffer named %s", name);
    VDA_REQUIRE(bytes > 0 && (size_t)bytes <= b->second.second, "vda_debug_copy: %s holds %lld bytes", name, (long long)b->second.second);
    VDA_HIP(hipMemcpyAsync(dst, (char*)h->ws + b->second.first, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// Tuning / A-B switches of the launch sequence: "residual_in_ln" (default 0), "ln_fold" (default 1), "dyn_sched" (default 0), "oc1_fused"
// (default 1), "mlp_fused" (default 0), "head_overlap" (default 0): see Run::forward.
extern "C" int vda_set_option(vda_model* h, const char* name, int value) {
    VDA_REQUIRE(h && name, "vda_set_option: null argument");
    if (strcmp(name, "residual_in_ln") == 0) {
        h->residual_in_ln = value;
        h->layouts.clear();                  // the workspace layout depends on it
        return 0;
    }
    if (strcmp(name, "profile_min_gflop") == 0) {       // vda_profile_start brackets only launches of at least this many GFLOP
        h->prof.min_flops = 1e9 * value;
        return 0;
    }
    if (strcmp(name, "dyn_sched") == 0) {
        h->dyn_sched = value;
        h->layouts.clear();
        return 0;
    }
    if (strcmp(name, "ln_fold") == 0) {
        h->ln_fold = value;
        h->layouts.clear();
        return 0;
    }
    if (strcmp(name, "head_overlap") == 0) {
        h->head_overlap = value != 0;
        return 0;
    }
    if (strcmp(name, "mlp_fused") == 0) {
        h->mlp_fused = value != 0;
        h->layouts.clear();                  // hid is not allocated with it
        return 0;
    }
    if (strcmp(name, "oc1_fused") == 0) {
        h->oc1_fused = value;
        h->layouts.clear();
        return 0;
    }
    vda_set_error("vda_set_option: unknown option %s", name);
    return 1;
}

// ---- bench.py's per-kernel timing of the GEMM / conv launches inside vda_forward
extern "C" int vda_profile_start(vda_model* h, int every) {
    VDA_REQUIRE(h && every > 0, "vda_profile_start: bad arguments");
    VDA_REQUIRE(h->prof.samples.empty(), "vda_profile_start: a profile is already open");
    const double keep = h->prof.min_flops;
    h->prof = Profile();
    h->prof.every = every;
    h->prof.min_flops = keep;
    return 0;
}

// Closes the profile and writes one JSON object {kernel name: {"launches", "flops", "timed", "timed_ms", "timed_flops"}} into
// `json` (NUL-terminated, at most `cap` bytes). Synchronises on the recorded events.
static int vda_profile_stop_impl(vda_model* h, char* json, int cap) {
    VDA_REQUIRE(h && json && cap > 2, "vda_profile_stop: bad arguments");
    Profile& pf = h->prof;
    struct Agg {
        long long timed = 0;
        double ms = 0.0, flops = 0.0;
    };
    std::map<std::string, Agg> agg;
    for (ProfSample& sm : pf.samples) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(sm.e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, sm.e0, sm.e1);
        (void)hipEventDestroy(sm.e0);
        (void)hipEventDestroy(sm.e1);
        if (e != hipSuccess) continue;
        Agg& a = agg[sm.name];
        a.timed += 1;
        a.ms += ms;
        a.flops += sm.flops;
    }
    std::string out = "{";
    bool first = true;
    for (const auto& kv : pf.launches) {
        const Agg& a = agg[kv.first];
        char line[512];
        snprintf(line, sizeof(line), "%s\"%s\": {\"launches\": %lld, \"flops\": %.6e, \"timed\": %lld, \"timed_ms\": %.6f, \"timed_flops\": %.6e}",
                 first ? "" : ", ", kv.first.c_str(), kv.second.first, kv.second.second, a.timed, a.ms, a.flops);
        out += line;
        first = false;
    }
    out += "}";
    h->prof = Profile();
    VDA_REQUIRE((int)out.size() + 1 <= cap, "vda_profile_stop: the report needs %d bytes", (int)out.size() + 1);
    memcpy(json, out.c_str(), out.size() + 1);
    return 0;
}

extern "C" int vda_load_weight(vda_model* h, const char* name, const void* ptr, const int64_t* dims, int ndim, int dtype) {
    try {
        return vda_load_weight_impl(h, name, ptr, dims, ndim, dtype);
    } catch (const std::exception& e) {        // no exception crosses the C ABI
        vda_set_error("vda_load_weight: %s", e.what());
        return 3;
    }
}

extern "C" int vda_finalize_weights(vda_model* h) {
    try {
        return vda_finalize_weights_impl(h);
    } catch (const std::exception& e) {        // no exception crosses the C ABI
        vda_set_error("vda_finalize_weights: %s", e.what());
        return 3;
    }
}

extern "C" int vda_prepare(vda_model* h, int B, int T, int H, int W, int precision) {
    try {
        return vda_prepare_impl(h, B, T, H, W, precision);
    } catch (const std::exception& e) {        // no exception crosses the C ABI
        vda_set_error("vda_prepare: %s", e.what());
        return 3;
    }
}

extern "C" int vda_forward(vda_model* h, const float* in, float* out, int B, int T, int H, int W, int precision, vda_stream_t stream) {
    try {
        return vda_forward_impl(h, in, out, B, T, H, W, precision, stream);
    } catch (const std::exception& e) {        // no exception crosses the C ABI
        vda_set_error("vda_forward: %s", e.what());
        return 3;
    }
}

extern "C" int vda_debug_copy(vda_model* h, const char* name, void* dst, int64_t bytes, vda_stream_t stream) {
    try {
        return vda_debug_copy_impl(h, name, dst, bytes, stream);
    } catch (const std::exception& e) {        // no exception crosses the C ABI
        vda_set_error("vda_debug_copy: %s", e.what());
        return 3;
    }
}

extern "C" int vda_profile_stop(vda_model* h, char* json, int cap) {
    try {
        return vda_profile_stop_impl(h, json, cap);
    } catch (const std::exception& e) {        // no exception crosses the C ABI
        vda_set_error("vda_profile_stop: %s", e.what());
        return 3;
    }
}

extern "C" int64_t vda_workspace_bytes(vda_model* h, int B, int T, int H, int W, int precision) {
    try {
        return vda_workspace_bytes_impl(h, B, T, H, W, precision);
    } catch (const std::exception& e) {
        vda_set_error("vda_workspace_bytes: %s", e.what());
        return -1;
    }
}
