// Temporal attention of the motion modules for gfx950: for every pixel, attention over the T <= 32
// frames of the clip, 8 heads of d = C/8 channels. 0.03 TFLOP per ViT-L clip against ~1.2 GB of
// q/k/v traffic: the kernel is HBM/latency-bound, so it is organised around coalesced 16-byte
// loads of the frame-major [T*hw, 3C] rows (no "(b f) d c -> (b d) f c" transpose is ever
// materialised) and does its 32x32 score blocks on the VALU out of LDS.
//
// Workgroup = one pixel x `hg` heads (hg*d <= 256 channels). Wave work item = (head, 16 queries):
// lane (i = l&15, jq = l>>4) owns query i and keys 8jq..8jq+7; softmax is 8 in-lane values + two
// cross-lane exchanges; P goes through a per-wave LDS scratch so each lane can then produce d/4
// output channels of its query.
#include "vda_common.h"

namespace {

constexpr int TMAX = 32;

template <int D>
__global__ void __launch_bounds__(256) tattn_kernel(const h16* __restrict__ qkv, h16* __restrict__ out, int T, int hw, int C,
                                                    int hg) {
    extern __shared__ __attribute__((aligned(16))) char smem_t[];
    const int CB = hg * D, RS = CB + 8;                      // channels per block, padded LDS row stride (halves)
    h16* lq = reinterpret_cast<h16*>(smem_t);
    h16* lk = lq + TMAX * RS;
    h16* lv = lk + TMAX * RS;
    float* lp = reinterpret_cast<float*>(lv + TMAX * RS);   // [4 waves][16][33]

    const int p = blockIdx.x, g = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- stage q, k, v of this pixel / head group: rows f = 0..T-1 (zero-filled above T)
    const int vpr = CB >> 3;                                 // 16-byte vectors per (row, tensor)
    for (int idx = tid; idx < TMAX * 3 * vpr; idx += 256) {
        const int f = idx / (3 * vpr), rem = idx - f * 3 * vpr;
        const int which = rem / vpr, v = rem - which * vpr;
        h16x8 x = {0, 0, 0, 0, 0, 0, 0, 0};
        if (f < T) x = *reinterpret_cast<const h16x8*>(qkv + ((size_t)f * hw + p) * (3 * (size_t)C) + which * C + g * CB + v * 8);
        *reinterpret_cast<h16x8*>(lq + which * TMAX * RS + f * RS + v * 8) = x;
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
            h16x8 qv[D / 8];
            const h16* qrow = lq + qi * RS + hh * D;
#pragma unroll
            for (int c = 0; c < D / 8; ++c) qv[c] = *reinterpret_cast<const h16x8*>(qrow + c * 8);
            float s[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const h16* krow = lk + (jq * 8 + t) * RS + hh * D;
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < D / 8; ++c) {
                    const h16x8 kv = *reinterpret_cast<const h16x8*>(krow + c * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) a += (float)qv[c][e] * (float)kv[e];
                }
                s[t] = (jq * 8 + t < T) ? a * scale : -1e30f;
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
        if (act && qi < T) {
            constexpr int DQ = D / 4;                        // channels per lane
            float o[DQ];
#pragma unroll
            for (int c = 0; c < DQ; ++c) o[c] = 0.f;
            const h16* vcol = lv + hh * D + jq * DQ;
            for (int j = 0; j < T; ++j) {
                const float pj = myp[i16 * 33 + j];
                const h16* vr = vcol + j * RS;
#pragma unroll
                for (int c = 0; c < DQ; c += 2) {
                    const h16x2 vv = *reinterpret_cast<const h16x2*>(vr + c);
                    o[c] += pj * (float)vv[0];
                    o[c + 1] += pj * (float)vv[1];
                }
            }
            h16* op = out + ((size_t)qi * hw + p) * C + g * CB + hh * D + jq * DQ;
#pragma unroll
            for (int c = 0; c < DQ; c += 2) {
                h16x2 ov = {(h16)o[c], (h16)o[c + 1]};
                *reinterpret_cast<h16x2*>(op + c) = ov;
            }
        }
        __syncthreads();
    }
}

template <int D>
int launch_tattn(const h16* qkv, h16* out, int T, int hw, int C, int heads, hipStream_t s) {
    int hg = 1;
    while (hg * 2 <= heads && hg * 2 * D <= 256) hg *= 2;
    const int CB = hg * D, RS = CB + 8;
    const size_t smem = (size_t)3 * TMAX * RS * sizeof(h16) + 4 * 16 * 33 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&tattn_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((tattn_kernel<D>), dim3(hw, heads / hg), dim3(256), smem, s, qkv, out, T, hw, C, hg);
    VDA_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int vda_temporal_attention_f16(const void* qkv, void* out, int T, int hw, int C, int heads, vda_stream_t stream) {
    VDA_REQUIRE(qkv && out, "vda_temporal_attention: null pointer");
    VDA_REQUIRE(T > 0 && T <= TMAX, "vda_temporal_attention: T=%d must be in 1..%d", T, TMAX);
    VDA_REQUIRE(hw > 0 && heads > 0 && (heads & (heads - 1)) == 0 && C % heads == 0, "vda_temporal_attention: bad geometry C=%d heads=%d", C, heads);
    VDA_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0 && C % 8 == 0, "vda_temporal_attention: alignment");
    const h16* q = (const h16*)qkv;
    h16* o = (h16*)out;
    hipStream_t s = (hipStream_t)stream;
    switch (C / heads) {
        case 8: return launch_tattn<8>(q, o, T, hw, C, heads, s);
        case 16: return launch_tattn<16>(q, o, T, hw, C, heads, s);
        case 24: return launch_tattn<24>(q, o, T, hw, C, heads, s);
        case 32: return launch_tattn<32>(q, o, T, hw, C, heads, s);
        case 48: return launch_tattn<48>(q, o, T, hw, C, heads, s);
        case 64: return launch_tattn<64>(q, o, T, hw, C, heads, s);
        case 128: return launch_tattn<128>(q, o, T, hw, C, heads, s);
        default: break;
    }
    vda_set_error("vda_temporal_attention: unsupported head dim %d", C / heads);
    return 1;
}
