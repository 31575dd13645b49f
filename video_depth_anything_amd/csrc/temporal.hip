// Temporal attention of the motion modules for gfx950: for every pixel, attention over the T <= 32
// frames of the clip, 8 heads of d = C/8 channels. 0.03 TFLOP per ViT-L clip against ~1.2 GB of
// q/k/v traffic: the kernel is HBM/latency-bound, so it is organised around coalesced 16-byte
// loads of the frame-major [T*hw, 3C] rows (no "(b f) d c -> (b d) f c" transpose is ever
// materialised) and does its 32x32 score blocks on the VALU out of LDS.
//
// Two kernels. Head dims 32 / 64 / 128 (ViT-L: 1024/8 and 256/8) run on MFMA (tattn_mfma_kernel below): one WAVE per
// (pixel, head), the spatial attention's single-tile case - S^T = K.Q^T with the query on the lane, in-lane softmax + one
// lane^32 exchange, the exponentiated accumulators reused in place as the B operand of O^T = V^T.P^T, V^T by
// ds_read_b64_tr_b16 - 16 + 2D/16... MFMAs instead of ~2000 VALU MACs per lane. Other head dims (ViT-S: 24, 48, 8) keep the
// VALU kernel: workgroup = one pixel x `hg` heads (hg*d <= 256 channels). Wave work item = (head, 16 queries):
// lane (i = l&15, jq = l>>4) owns query i and keys 8jq..8jq+7; softmax is 8 in-lane values + two
// cross-lane exchanges; P goes through a per-wave LDS scratch so each lane can then produce d/4
// output channels of its query.
#include "vda_common.h"

namespace {

constexpr int TMAX = 32;

template <int D, typename T>
__global__ void __launch_bounds__(256) tattn_kernel(const T* __restrict__ qkv, T* __restrict__ out, int Tn, int hw, int C,
                                                    int hg) {
    extern __shared__ __attribute__((aligned(16))) char smem_t[];
    const int CB = hg * D, RS = CB + 8;                      // channels per block, padded LDS row stride (elements)
    T* lq = reinterpret_cast<T*>(smem_t);
    T* lk = lq + TMAX * RS;
    T* lv = lk + TMAX * RS;
    float* lp = reinterpret_cast<float*>(lv + TMAX * RS);   // [4 waves][16][33]

    const int p = blockIdx.x, g = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- stage q, k, v of this pixel / head group: rows f = 0..Tn-1 (zero-filled above Tn)
    const int vpr = CB >> 3;                                 // 8-element vectors per (row, tensor)
    for (int idx = tid; idx < TMAX * 3 * vpr; idx += 256) {
        const int f = idx / (3 * vpr), rem = idx - f * 3 * vpr;
        const int which = rem / vpr, v = rem - which * vpr;
        float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (f < Tn) load8(qkv + ((size_t)f * hw + p) * (3 * (size_t)C) + which * C + g * CB + v * 8, x);
        store8(lq + which * TMAX * RS + f * RS + v * 8, x);
    }
    __syncthreads();

    const int i16 = lane & 15, jq = lane >> 4;
    float* myp = lp + wave * 16 * 33;
    const float scale = rsqrtf((float)D);
    const int items = hg * 2;
    for (int it0 = 0; it0 < items; it0 += 4) {
        const int it = it0 + wave;
        const bool act = it < items;
        const int hh = it >> 1, qi = (it & 1) * 16 + i16;    // head in group, query frame
        if (act) {
            float qv[D / 8][8];
            const T* qrow = lq + qi * RS + hh * D;
#pragma unroll
            for (int c = 0; c < D / 8; ++c) load8(qrow + c * 8, qv[c]);
            float s[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const T* krow = lk + (jq * 8 + t) * RS + hh * D;
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < D / 8; ++c) {
                    float kv[8];
                    load8(krow + c * 8, kv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) a += qv[c][e] * kv[e];
                }
                s[t] = (jq * 8 + t < Tn) ? a * scale : -1e30f;
            }
            float mx = s[0];
#pragma unroll
            for (int t = 1; t < 8; ++t) mx = fmaxf(mx, s[t]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                s[t] = __expf(s[t] - mx);
                sum += s[t];
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int t = 0; t < 8; ++t) myp[i16 * 33 + jq * 8 + t] = s[t] * inv;
        }
        __syncthreads();
        if (act && qi < Tn) {
            constexpr int DQ = D / 4;                        // channels per lane
            float o[DQ];
#pragma unroll
            for (int c = 0; c < DQ; ++c) o[c] = 0.f;
            typedef T T2 __attribute__((ext_vector_type(2)));
            const T* vcol = lv + hh * D + jq * DQ;
            for (int j = 0; j < Tn; ++j) {
                const float pj = myp[i16 * 33 + j];
                const T* vr = vcol + j * RS;
#pragma unroll
                for (int c = 0; c < DQ; c += 2) {
                    const T2 vv = *reinterpret_cast<const T2*>(vr + c);
                    o[c] += pj * (float)vv[0];
                    o[c + 1] += pj * (float)vv[1];
                }
            }
            T* op = out + ((size_t)qi * hw + p) * C + g * CB + hh * D + jq * DQ;
#pragma unroll
            for (int c = 0; c < DQ; c += 2) {
                const T2 ov = {(T)o[c], (T)o[c + 1]};
                *reinterpret_cast<T2*>(op + c) = ov;
            }
        }
        __syncthreads();
    }
}

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

// ---- MFMA form: one wave per (pixel, head), 4 heads per workgroup, no cross-wave traffic (each wave stages, computes and
// stores its own head, so LDS ordering is the wave's own lgkmcnt).
//   LDS per wave: q (later v) and k as [32 frames][D channels] fp16 rows of 2D + 48 (80 at D = 64) bytes, see RSB.
template <int D>
__global__ void __launch_bounds__(256) tattn_mfma_kernel(const h16* __restrict__ qkv, h16* __restrict__ out, int T, int hw, int C, int heads) {
    constexpr int RSB = 2 * D + (D == 64 ? 80 : 48);          // row bytes: an ODD number of 16-byte slots (16 consecutive rows of one column hit
                                                              // 16 different slots of the 256-byte bank row: conflict-free ds_read_b128) whose multiples
                                                              // 0..3 stay >= 32 bytes apart mod 256 (the 4-row ds_read_b64_tr_b16 blocks)
    constexpr int TEN = TMAX * RSB;                          // one tensor of one head
    extern __shared__ __attribute__((aligned(16))) char smem_t[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x, head = blockIdx.y * 4 + wave;
    if (head >= heads) return;                               // wave-uniform; no workgroup barrier below
    char* base = smem_t + wave * 2 * TEN;

    // ---- stage q, k of (pixel, head): 16-byte loads, D/8 lanes per (frame, tensor) segment. v is fetched now but parked in
    // registers: it takes q's LDS rows once the scores are done (2 tensors of LDS per wave instead of 3: 5 -> 8 waves per CU at D = 128).
    constexpr int VPS = D / 8;                                // 16-byte vectors per segment
    constexpr int SPI = 64 / VPS;                            // segments per wave instruction
    constexpr int NIT = TMAX / SPI;                          // instructions per tensor
    const int sv = lane % VPS, ss = lane / VPS;
    h16x8 vreg[NIT];
    {
#pragma unroll
        for (int it = 0; it < 2 * NIT; ++it) {
            const int seg = it * SPI + ss;                   // seg = which * 32 + frame
            const int which = seg >> 5, f = seg & 31;
            h16x8 x = {0, 0, 0, 0, 0, 0, 0, 0};
            if (f < T) x = *reinterpret_cast<const h16x8*>(qkv + ((size_t)f * hw + p) * (3 * (size_t)C) + which * C + head * D + sv * 8);
            *reinterpret_cast<h16x8*>(base + which * TEN + f * RSB + sv * 16) = x;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int f = it * SPI + ss;
            vreg[it] = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (f < T) vreg[it] = *reinterpret_cast<const h16x8*>(qkv + ((size_t)f * hw + p) * (3 * (size_t)C) + 2 * C + head * D + sv * 8);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    const int r = lane & 31, h = lane >> 5;
    const char* qb = base;
    const char* kb = base + TEN;
    const char* vb = base;                                  // v replaces q after the scores

    // ---- S^T[key][query] = K . Q^T
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
        const h16x8 kf = *reinterpret_cast<const h16x8*>(kb + r * RSB + (ks * 16 + h * 8) * 2);
        const h16x8 qf = *reinterpret_cast<const h16x8*>(qb + r * RSB + (ks * 16 + h * 8) * 2);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf, s, 0, 0, 0);
    }
    // q's rows are consumed (the MFMAs above have read their operands once they issue; lgkmcnt covers the reads): park v there
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < NIT; ++it) *reinterpret_cast<h16x8*>(base + (it * SPI + ss) * RSB + sv * 16) = vreg[it];
    // register e is key (e&3) + 8*(e>>2) + 4h of query r
    constexpr float LOG2E = 1.4426950408889634f;
    const float c2 = rsqrtf((float)D) * LOG2E;
    float mx = -1e30f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int key = (e & 3) + 8 * (e >> 2) + 4 * h;
        s[e] = key < T ? s[e] * c2 : -1e30f;
        mx = fmaxf(mx, s[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    h16x8 pf[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(s[e] - mx);
        sum += pv;
        pf[e >> 3][e & 7] = (h16)pv;
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // v rows written (own wave) before the transposed reads
    // ---- O^T[ch][query] = V^T . P^T: 2 steps of 16 keys per 32-channel block (k order of pf: key 16s + 8(j>>2) + 4h + (j&3))
    const int i = lane & 15, qq = i >> 2, pp = i & 3;
#pragma unroll
    for (int c = 0; c < D / 32; ++c) {
        f32x16 o;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.f;
        const int col = c * 32 + 16 * ((lane >> 4) & 1) + 4 * pp;
#pragma unroll
        for (int kstep = 0; kstep < 2; ++kstep) {
            h16x8 vf;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int key = kstep * 16 + half * 8 + 4 * h + qq;
                const fp16x4_t v4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((VDA_LDS_AS fp16x4_t*)(vb + key * RSB + col * 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) vf[half * 4 + e] = (h16)v4[e];
            }
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kstep], o, 0, 0, 0);
        }
        // lane holds query (frame) r, channels c*32 + (e&3) + 8*(e>>2) + 4h
        if (r < T) {
            h16* op = out + ((size_t)r * hw + p) * C + head * D + c * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h16x4 ov = {(h16)(o[4 * g + 0] * inv), (h16)(o[4 * g + 1] * inv), (h16)(o[4 * g + 2] * inv), (h16)(o[4 * g + 3] * inv)};
                *reinterpret_cast<h16x4*>(op + 8 * g) = ov;
            }
        }
    }
}

template <int D>
int launch_tattn_mfma(const h16* qkv, h16* out, int T, int hw, int C, int heads, hipStream_t s) {
    constexpr size_t smem = (size_t)4 * 2 * TMAX * (2 * D + (D == 64 ? 80 : 48));
    static_assert(smem <= 160 * 1024, "LDS budget");
    static VdaKernelDeviceState dev_state;
    if (vda_prepare_kernel(reinterpret_cast<const void*>(&tattn_mfma_kernel<D>), (int)smem, dev_state) < 0) return 2;
    hipLaunchKernelGGL((tattn_mfma_kernel<D>), dim3(hw, (heads + 3) / 4), dim3(256), smem, s, qkv, out, T, hw, C, heads);
    VDA_LAUNCH_CHECK();
    return 0;
}

template <int D, typename T>
int launch_tattn(const T* qkv, T* out, int Tn, int hw, int C, int heads, hipStream_t s) {
    // head group: as many heads per workgroup as keep q, k, v of the group (3 x 32 rows) within 64 KiB of LDS
    constexpr int max_cb = sizeof(T) == 2 ? 256 : 128;
    int hg = 1;
    while (hg * 2 <= heads && hg * 2 * D <= max_cb) hg *= 2;
    const int CB = hg * D, RS = CB + 8;
    const size_t smem = (size_t)3 * TMAX * RS * sizeof(T) + 4 * 16 * 33 * sizeof(float);
    static VdaKernelDeviceState dev_state;
    if (vda_prepare_kernel(reinterpret_cast<const void*>(&tattn_kernel<D, T>), 64 * 1024, dev_state) < 0) return 2;
    hipLaunchKernelGGL((tattn_kernel<D, T>), dim3(hw, heads / hg), dim3(256), smem, s, qkv, out, Tn, hw, C, hg);
    VDA_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int tattn_valu_dispatch(const T* q, T* o, int Tn, int hw, int C, int heads, hipStream_t s) {
    switch (C / heads) {
        case 8: return launch_tattn<8, T>(q, o, Tn, hw, C, heads, s);
        case 16: return launch_tattn<16, T>(q, o, Tn, hw, C, heads, s);
        case 24: return launch_tattn<24, T>(q, o, Tn, hw, C, heads, s);
        case 32: return launch_tattn<32, T>(q, o, Tn, hw, C, heads, s);
        case 48: return launch_tattn<48, T>(q, o, Tn, hw, C, heads, s);
        case 64: return launch_tattn<64, T>(q, o, Tn, hw, C, heads, s);
        case 128: return launch_tattn<128, T>(q, o, Tn, hw, C, heads, s);
        default: break;
    }
    vda_set_error("vda_temporal_attention: unsupported head dim %d", C / heads);
    return 1;
}

}  // namespace

static int g_tattn_variant = 1;   // 1: MFMA kernel for head dims 32/64/128; 0: VALU kernel everywhere (A/B, cross-check)

// pe = 'rope' (motion_module.py:254-257): q and k of the fused projection rotated in place, channel pair (2i, 2i+1) by
// frame * 10000^(-2i/C). A streaming pass over two thirds of qkv; one thread = 4 pairs (16 bytes of fp16) of q and the same of k.
namespace {
template <typename T>
__global__ void __launch_bounds__(256) rope_qk_kernel(T* __restrict__ qkv, int Tn, int hw, int C) {
    const int per_row = C / 8;                                  // 8-channel vectors per row of q (and of k)
    const long long total = (long long)Tn * hw * per_row;
    const float lnb = -9.210340371976184f / (float)C;           // -ln(10000) / C
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / per_row;
        const int v = (int)(i - row * per_row);
        const int t = (int)(row / hw);
        float cs[4], sn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float freq = expf(lnb * (float)(2 * (v * 4 + e)));          // 1 / 10000^(2i / C), i = v*4 + e
            sincosf((float)t * freq, &sn[e], &cs[e]);
        }
#pragma unroll
        for (int part = 0; part < 2; ++part) {                  // q, then k
            T* p = qkv + row * 3 * C + part * C + v * 8;
            float x[8], y[8];
            load8(p, x);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[2 * e] = x[2 * e] * cs[e] - x[2 * e + 1] * sn[e];
                y[2 * e + 1] = x[2 * e] * sn[e] + x[2 * e + 1] * cs[e];
            }
            store8(p, y);
        }
    }
}
template <typename T>
int rope_launch(T* qkv, int Tn, int hw, int C, vda_stream_t stream) {
    VDA_REQUIRE(qkv != nullptr && Tn > 0 && hw > 0 && C > 0 && C % 8 == 0, "vda_rope_qk: bad arguments (C=%d must be a multiple of 8)", C);
    VDA_REQUIRE(((uintptr_t)qkv & 15) == 0, "vda_rope_qk: 16-byte alignment required");
    const long long total = (long long)Tn * hw * (C / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL((rope_qk_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, qkv, Tn, hw, C);
    VDA_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int vda_rope_qk_f16(void* qkv, int T, int hw, int C, vda_stream_t stream) { return rope_launch<h16>((h16*)qkv, T, hw, C, stream); }
extern "C" int vda_rope_qk_f32(float* qkv, int T, int hw, int C, vda_stream_t stream) { return rope_launch<float>(qkv, T, hw, C, stream); }

extern "C" int vda_temporal_attention_set_variant(int v) {
    g_tattn_variant = v;
    return 0;
}

extern "C" int vda_temporal_attention_f16(const void* qkv, void* out, int T, int hw, int C, int heads, vda_stream_t stream) {
    VDA_REQUIRE(qkv && out, "vda_temporal_attention: null pointer");
    VDA_REQUIRE(T > 0 && T <= TMAX, "vda_temporal_attention: T=%d must be in 1..%d", T, TMAX);
    VDA_REQUIRE(hw > 0 && heads > 0 && (heads & (heads - 1)) == 0 && C % heads == 0, "vda_temporal_attention: bad geometry C=%d heads=%d", C, heads);
    VDA_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0 && C % 8 == 0, "vda_temporal_attention: alignment");
    const h16* q = (const h16*)qkv;
    h16* o = (h16*)out;
    hipStream_t s = (hipStream_t)stream;
    if (g_tattn_variant) {
        switch (C / heads) {                     // MFMA form
            case 32: return launch_tattn_mfma<32>(q, o, T, hw, C, heads, s);
            case 64: return launch_tattn_mfma<64>(q, o, T, hw, C, heads, s);
            case 128: return launch_tattn_mfma<128>(q, o, T, hw, C, heads, s);
            default: break;
        }
    }
    return tattn_valu_dispatch<h16>(q, o, T, hw, C, heads, s);
}

// fp32-operand form (the reference's fp32=True path): the VALU kernel on fp32 q / k / v, fp32 out.
extern "C" int vda_temporal_attention_f32(const float* qkv, float* out, int T, int hw, int C, int heads, vda_stream_t stream) {
    VDA_REQUIRE(qkv && out, "vda_temporal_attention_f32: null pointer");
    VDA_REQUIRE(T > 0 && T <= TMAX, "vda_temporal_attention_f32: T=%d must be in 1..%d", T, TMAX);
    VDA_REQUIRE(hw > 0 && heads > 0 && (heads & (heads - 1)) == 0 && C % heads == 0, "vda_temporal_attention_f32: bad geometry C=%d heads=%d", C, heads);
    VDA_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0 && C % 8 == 0, "vda_temporal_attention_f32: alignment");
    return tattn_valu_dispatch<float>(qkv, out, T, hw, C, heads, (hipStream_t)stream);
}
