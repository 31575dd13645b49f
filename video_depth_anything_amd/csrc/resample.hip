// Streaming (HBM-bound) layout and resampling kernels for gfx950: bilinear align_corners=True
// resizes, patch gather for the patch-embed GEMM, cls rows, the 32->1 output projection and the
// uint8 -> normalised fp32 frame conversion. All use 16-byte per-lane accesses where the layout
// allows; grids are capped and grid-strided.
#include "vda_common.h"

namespace {

__device__ __forceinline__ void lerp_coord(int dst, int in, int outn, int& i0, int& i1, float& w1) {
    // align_corners=True source coordinate: dst * (in-1)/(out-1)
    const float scale = outn > 1 ? (float)(in - 1) / (float)(outn - 1) : 0.f;
    const float src = scale * (float)dst;
    i0 = min((int)src, in - 1);
    i1 = min(i0 + 1, in - 1);
    w1 = src - (float)i0;
}

// Block = TWO consecutive output rows (b, 2k), (b, 2k+1): when upsampling they read the same source rows or rows shifted by one, so a
// thread fetches three source rows' pixel pairs once (6 x 16 B; 4 when both rows sit between the same two source rows) and writes two
// outputs - the kernel is bound by its load instructions, not by bytes (4 loads per 16-byte store in the one-row form: 3.1 TB/s).
// A thread walks (X, 8-channel vector) pairs with 32-bit index math only (the flat-index version spent its time in 64-bit div/mod).
// Per output the arithmetic is unchanged: same expression, same operands.
template <typename T>
__global__ void __launch_bounds__(256) bilinear_nhwc_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                            const T* __restrict__ add, int B, int h, int w, int H, int W, int C) {
    const int nv = C >> 3;
    const int Y = 2 * blockIdx.x, b = blockIdx.y;
    const bool two = Y + 1 < H;                          // (odd H: the last block has one row)
    int ya0, ya1, yb0, yb1;
    float wya, wyb;
    lerp_coord(Y, h, H, ya0, ya1, wya);
    lerp_coord(two ? Y + 1 : Y, h, H, yb0, yb1, wyb);
    // source rows needed: ya0, ya1, yb0, yb1, a subset of {ya0, ya0 + 1, ya0 + 2} when upsampling; otherwise (downsampling: rows
    // further apart) the second output row simply loads its own pair
    const bool share = yb0 >= ya0 && yb1 <= ya0 + 2;
    const int r2i = min(ya0 + 2, h - 1);
    const T* base = in + (size_t)b * h * w * C;
    const T* r0 = base + (size_t)ya0 * w * C;
    const T* r1 = base + (size_t)min(ya0 + 1, h - 1) * w * C;
    const T* r2 = base + (size_t)r2i * w * C;
    const bool need_r2 = share && (yb1 == ya0 + 2 && r2i == ya0 + 2);
    const T* rb0 = base + (size_t)yb0 * w * C;
    const T* rb1 = base + (size_t)yb1 * w * C;
    const size_t orow = ((size_t)b * H + Y) * W * C;
    const float xscale = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const int n = W * nv;
    const bool fixed_v = (256 % nv) == 0;
    const int v_fixed = threadIdx.x % nv, x_first = threadIdx.x / nv, x_step = 256 / nv;
    for (int i = threadIdx.x, Xw = x_first; i < n; i += 256, Xw += x_step) {
        const int X = fixed_v ? Xw : i / nv, v = fixed_v ? v_fixed : i - X * nv;
        const float src = xscale * (float)X;
        const int x0 = min((int)src, w - 1), x1 = min(x0 + 1, w - 1);
        const float wx = src - (float)x0;
        const int o0 = x0 * C + v * 8, o1 = x1 * C + v * 8;
        float p[3][2][8];                                // [source row ya0 + k][x0 / x1]
        load8(r0 + o0, p[0][0]);
        load8(r0 + o1, p[0][1]);
        load8(r1 + o0, p[1][0]);
        load8(r1 + o1, p[1][1]);
        float q[2][2][8];                                // second output row's pair when it cannot be served from p
        if (two && !share) {
            load8(rb0 + o0, q[0][0]);
            load8(rb0 + o1, q[0][1]);
            load8(rb1 + o0, q[1][0]);
            load8(rb1 + o1, q[1][1]);
        } else if (need_r2) {
            load8(r2 + o0, p[2][0]);
            load8(r2 + o1, p[2][1]);
        }
        float ad[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, o[8];
        {   // row Y: source rows ya0 (p[0]) and ya1 (p[ya1 - ya0])
            const int k1 = ya1 - ya0;
            if (add) load8(add + orow + (size_t)i * 8, ad);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float top = p[0][0][e] * (1.f - wx) + p[0][1][e] * wx;
                const float b1 = k1 ? p[1][0][e] : p[0][0][e], b2 = k1 ? p[1][1][e] : p[0][1][e];
                const float bot = b1 * (1.f - wx) + b2 * wx;
                o[e] = top * (1.f - wya) + bot * wya + ad[e];
            }
            store8(out + orow + (size_t)i * 8, o);
        }
        if (two) {
            const size_t orow2 = orow + (size_t)W * C;
            if (add) load8(add + orow2 + (size_t)i * 8, ad);
            const int k0 = yb0 - ya0, k1 = yb1 - ya0;      // 0..2 when share
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t1, t2, b1, b2;
                if (share) {
                    t1 = k0 == 0 ? p[0][0][e] : k0 == 1 ? p[1][0][e] : p[2][0][e];
                    t2 = k0 == 0 ? p[0][1][e] : k0 == 1 ? p[1][1][e] : p[2][1][e];
                    b1 = k1 == 0 ? p[0][0][e] : k1 == 1 ? p[1][0][e] : p[2][0][e];
                    b2 = k1 == 0 ? p[0][1][e] : k1 == 1 ? p[1][1][e] : p[2][1][e];
                } else {
                    t1 = q[0][0][e], t2 = q[0][1][e], b1 = q[1][0][e], b2 = q[1][1][e];
                }
                const float top = t1 * (1.f - wx) + t2 * wx;
                const float bot = b1 * (1.f - wx) + b2 * wx;
                o[e] = top * (1.f - wyb) + bot * wyb + ad[e];
            }
            store8(out + orow2 + (size_t)i * 8, o);
        }
    }
}

__global__ void __launch_bounds__(256) bilinear_plane_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int h,
                                                             int w, int H, int W, int relu) {
    const size_t total = (size_t)B * H * W;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int X = (int)(idx % W);
        const size_t t = idx / W;
        const int Y = (int)(t % H);
        const int b = (int)(t / H);
        int y0, y1, x0, x1;
        float wy, wx;
        lerp_coord(Y, h, H, y0, y1, wy);
        lerp_coord(X, w, W, x0, x1, wx);
        const float* base = in + (size_t)b * h * w;
        const float top = base[(size_t)y0 * w + x0] * (1.f - wx) + base[(size_t)y0 * w + x1] * wx;
        const float bot = base[(size_t)y1 * w + x0] * (1.f - wx) + base[(size_t)y1 * w + x1] * wx;
        float o = top * (1.f - wy) + bot * wy;
        if (relu) o = fmaxf(o, 0.f);
        out[idx] = o;
    }
}

// Block = one patch row of one frame: reads 3*14 image rows (coalesced), scatters them into the
// pw patch rows of the GEMM A matrix (column = c*196 + ky*14 + kx).
template <typename OT>
__global__ void __launch_bounds__(256) patchify_kernel(const float* __restrict__ x, OT* __restrict__ out, int H, int W, int Kpad) {
    const int pw = W / 14, ph = H / 14;
    const int py = blockIdx.x, b = blockIdx.y;
    // two horizontally adjacent pixels per thread (W and the patch width are even: a pair never straddles two patches): 8-byte loads,
    // and 4-byte stores for the fp16 form instead of 2-byte ones (86 -> 43 us per ViT-L clip, tools/patchify_bench.py; same values)
    const int n2 = 3 * 14 * (W >> 1);
    for (int idx = threadIdx.x; idx < n2; idx += 256) {
        const int c = idx / (14 * (W >> 1)), rem = idx - c * 14 * (W >> 1);
        const int ky = rem / (W >> 1), xx = (rem - ky * (W >> 1)) * 2;
        const int px = xx / 14, kx = xx - px * 14;
        if (px >= pw) continue;
        const float2 v = *reinterpret_cast<const float2*>(x + (((size_t)b * 3 + c) * H + py * 14 + ky) * W + xx);
        OT* o = out + ((size_t)(b * ph + py) * pw + px) * Kpad + c * 196 + ky * 14 + kx;
        if constexpr (sizeof(OT) == 2) {
            typedef OT ot2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<ot2*>(o) = ot2{(OT)v.x, (OT)v.y};
        } else {
            o[0] = (OT)v.x;
            o[1] = (OT)v.y;
        }
    }
}

__global__ void cls_rows_kernel(float* __restrict__ tok, const float* __restrict__ cls, const float* __restrict__ pos, int B, int P,
                                int D) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * D) return;
    const int b = idx / D, c = idx - b * D;
    tok[(size_t)b * (P + 1) * D + c] = cls[c] + pos[c];
}

template <typename T>
__global__ void __launch_bounds__(256) head_out_kernel(const T* __restrict__ in, const float* __restrict__ w, float bias,
                                                       float* __restrict__ out, long long rows, int Cpad) {
    // 4 lanes per row, 8 channels each (32 live channels), reduced with two shuffles.
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = gid >> 2;
    const int part = (int)(gid & 3);
    float a = 0.f;
    if (row < (size_t)rows) {
        float x[8];
        load8(in + row * Cpad + part * 8, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) a += x[e] * w[part * 8 + e];
    }
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    if (row < (size_t)rows && part == 0) out[row] = fmaxf(a + bias, 0.f);
}

__global__ void __launch_bounds__(256) normalize_u8_kernel(const uint8_t* __restrict__ f, float* __restrict__ out, int n, int H, int W) {
    const size_t hw = (size_t)H * W, total = (size_t)n * hw;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t fr = idx / hw, pix = idx - fr * hw;
        const uint8_t* p = f + idx * 3;
        // reference arithmetic: float32(x)/255 (fp32), then (v - mean)/std in float64, cast to fp32
        const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (float)p[c] / 255.0f;
            out[(fr * 3 + c) * hw + pix] = (float)(((double)v - mean[c]) / stdv[c]);
        }
    }
}

// Same arithmetic as normalize_u8_kernel, but output frame i is source frame idx[i] of a video resident in HBM:
// the window's gather (video_depth.py:197-201, key-frame refill included) without a host round trip.
__global__ void __launch_bounds__(256) gather_normalize_u8_kernel(const uint8_t* __restrict__ f, const int* __restrict__ idx,
                                                                  float* __restrict__ out, int n, int H, int W) {
    const size_t hw = (size_t)H * W, total = (size_t)n * hw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t fr = i / hw, pix = i - fr * hw;
        const uint8_t* p = f + ((size_t)idx[fr] * hw + pix) * 3;
        const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (float)p[c] / 255.0f;
            out[(fr * 3 + c) * hw + pix] = (float)(((double)v - mean[c]) / stdv[c]);
        }
    }
}

// ---- bicubic resampling (a = -0.75, half-pixel centres, taps clamped to the border): the definition shared by
// cv2.resize(INTER_CUBIC) (util/transform.py:113) and F.interpolate(mode='bicubic', align_corners=False) (dinov2.py:196-203).
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    c[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

// Window gather + resize-to-network-size + normalise in one pass (video_depth.py:197-201, util/transform.py:109-147):
// output pixel (i, :, Y, X) of the fp32 NCHW clip = normalise( bicubic( frame idx[i] / 255 ) ). The reference resizes the
// [0,1] image and then normalises; so does this (the horizontal pass first, like cv2's separable filter).
__global__ void __launch_bounds__(256) gather_resize_normalize_kernel(const uint8_t* __restrict__ video, const int* __restrict__ idx,
                                                                      float* __restrict__ out, int n, int H0, int W0, int H, int W,
                                                                      float sy, float sx) {
    const size_t hw = (size_t)H * W, total = (size_t)n * hw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t fr = i / hw, pix = i - fr * hw;
        const int Y = (int)(pix / W), X = (int)(pix - (size_t)Y * W);
        const float fy = ((float)Y + 0.5f) * sy - 0.5f, fx = ((float)X + 0.5f) * sx - 0.5f;
        const float y0f = floorf(fy), x0f = floorf(fx);
        float cy[4], cx[4];
        cubic_coeffs(fy - y0f, cy);
        cubic_coeffs(fx - x0f, cx);
        const int iy = (int)y0f, ix = (int)x0f;
        const uint8_t* f = video + (size_t)idx[fr] * H0 * W0 * 3;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), H0 - 1);
            float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xx = min(max(ix - 1 + b, 0), W0 - 1);
                const uint8_t* p = f + ((size_t)yy * W0 + xx) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) row[c] += cx[b] * ((float)p[c] / 255.0f);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] += cy[a] * row[c];
        }
        const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(fr * 3 + c) * hw + pix] = (float)(((double)acc[c] - mean[c]) / stdv[c]);
    }
}

// Positional-embedding grid [1 + g*g, D] -> [1 + ph*pw, D] (dinov2.py:185-210): row 0 (cls) copied, the g x g patch grid
// resampled with the reference's explicit scale factors (inv_sy = 1 / ((ph + 0.1) / g), likewise x). Token-major, so the
// channel axis is contiguous: one thread per (token, channel).
__global__ void __launch_bounds__(256) pos_embed_resample_kernel(const float* __restrict__ pe, float* __restrict__ out, int g, int ph,
                                                                 int pw, int D, float inv_sy, float inv_sx) {
    const size_t total = (size_t)(1 + ph * pw) * D;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int tok = (int)(i / D), d = (int)(i - (size_t)tok * D);
        if (tok == 0) {
            out[i] = pe[d];
            continue;
        }
        const int Y = (tok - 1) / pw, X = (tok - 1) - Y * pw;
        const float fy = inv_sy * ((float)Y + 0.5f) - 0.5f, fx = inv_sx * ((float)X + 0.5f) - 0.5f;
        const float y0f = floorf(fy), x0f = floorf(fx);
        float cy[4], cx[4];
        cubic_coeffs(fy - y0f, cy);
        cubic_coeffs(fx - x0f, cx);
        const int iy = (int)y0f, ix = (int)x0f;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), g - 1);
            float row = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xx = min(max(ix - 1 + b, 0), g - 1);
                row += cx[b] * pe[(size_t)(1 + yy * g + xx) * D + d];
            }
            acc += cy[a] * row;
        }
        out[i] = acc;
    }
}

// Readout input of the use_clstoken head (dpt_temporal.py:56-59): tokens [frames*(P+1), D] with the cls token at row 0 of every
// frame -> [frames*P, 2D] = cat(patch token, that frame's cls token). 8 elements per lane.
template <typename T>
__global__ void __launch_bounds__(256) readout_concat_kernel(const T* __restrict__ tok, T* __restrict__ out, int frames, int P, int D) {
    const int nv = D >> 3;
    const size_t total = (size_t)frames * P * 2 * nv;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int v = (int)(i % (2 * nv));
        const size_t r = i / (2 * nv);
        const int f = (int)(r / P), p = (int)(r - (size_t)f * P);
        const T* src = v < nv ? tok + ((size_t)f * (P + 1) + 1 + p) * D + v * 8 : tok + (size_t)f * (P + 1) * D + (v - nv) * 8;
        float x[8];
        load8(src, x);
        store8(out + r * 2 * D + (size_t)v * 8, x);
    }
}

inline unsigned capped_grid(size_t work_items) {
    size_t blocks = (work_items + 255) / 256;
    const size_t cap = 256 * 16;
    return (unsigned)(blocks < cap ? (blocks ? blocks : 1) : cap);
}

}  // namespace

template <typename T>
static int bilinear_nhwc_launch(const T* in, T* out, const T* add, int B, int h, int w, int H, int W, int C, vda_stream_t stream) {
    VDA_REQUIRE(in && out, "vda_bilinear_nhwc: null pointer");
    VDA_REQUIRE(B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "vda_bilinear_nhwc: bad geometry");
    VDA_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)add & 15) == 0, "vda_bilinear_nhwc: alignment");
    VDA_REQUIRE(B <= 65535 && (long long)w * C < (1ll << 31) && (long long)W * C < (1ll << 31), "vda_bilinear_nhwc: row too large");
    hipLaunchKernelGGL((bilinear_nhwc_kernel<T>), dim3((H + 1) / 2, B), dim3(256), 0, (hipStream_t)stream, in, out, add, B, h, w, H, W, C);
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_bilinear_nhwc_f16(const void* in, void* out, const void* add, int B, int h, int w, int H, int W, int C,
                                     vda_stream_t stream) {
    return bilinear_nhwc_launch<h16>((const h16*)in, (h16*)out, (const h16*)add, B, h, w, H, W, C, stream);
}

extern "C" int vda_bilinear_nhwc_f32(const float* in, float* out, const float* add, int B, int h, int w, int H, int W, int C,
                                     vda_stream_t stream) {
    return bilinear_nhwc_launch<float>(in, out, add, B, h, w, H, W, C, stream);
}

extern "C" int vda_bilinear_plane_f32(const float* in, float* out, int B, int h, int w, int H, int W, int relu, vda_stream_t stream) {
    VDA_REQUIRE(in && out, "vda_bilinear_plane: null pointer");
    VDA_REQUIRE(B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "vda_bilinear_plane: bad geometry");
    hipLaunchKernelGGL(bilinear_plane_kernel, dim3(capped_grid((size_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, in, out, B, h, w,
                       H, W, relu);
    VDA_LAUNCH_CHECK();
    return 0;
}

template <typename OT>
static int patchify_launch(const float* x, OT* out, int B, int H, int W, int Kpad, vda_stream_t stream) {
    VDA_REQUIRE(x && out, "vda_patchify: null pointer");
    VDA_REQUIRE(B > 0 && H > 0 && W > 0 && H % 14 == 0 && W % 14 == 0, "vda_patchify: H=%d W=%d must be multiples of 14", H, W);
    VDA_REQUIRE(Kpad >= 588 && Kpad % 2 == 0, "vda_patchify: Kpad=%d < 588 or odd", Kpad);
    VDA_REQUIRE(((uintptr_t)x & 7) == 0 && ((uintptr_t)out & 3) == 0, "vda_patchify: x 8-byte, out 4-byte aligned");
    hipLaunchKernelGGL((patchify_kernel<OT>), dim3(H / 14, B), dim3(256), 0, (hipStream_t)stream, x, out, H, W, Kpad);
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_patchify_f32_f16(const float* x, void* out, int B, int H, int W, int Kpad, vda_stream_t stream) {
    return patchify_launch<h16>(x, (h16*)out, B, H, W, Kpad, stream);
}

extern "C" int vda_patchify_f32_f32(const float* x, float* out, int B, int H, int W, int Kpad, vda_stream_t stream) {
    return patchify_launch<float>(x, out, B, H, W, Kpad, stream);
}

extern "C" int vda_cls_rows_f32(float* tok, const float* cls, const float* pos, int B, int P, int D, vda_stream_t stream) {
    VDA_REQUIRE(tok && cls && pos && B > 0 && P > 0 && D > 0, "vda_cls_rows: bad arguments");
    hipLaunchKernelGGL(cls_rows_kernel, dim3((B * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, tok, cls, pos, B, P, D);
    VDA_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int head_out_launch(const T* in, const float* w, float bias, float* out, long long rows, int Cpad, vda_stream_t stream) {
    VDA_REQUIRE(in && w && out && rows > 0, "vda_head_out: bad arguments");
    VDA_REQUIRE(Cpad >= 32 && Cpad % 8 == 0 && ((uintptr_t)in & 15) == 0, "vda_head_out: Cpad=%d must be >=32, multiple of 8", Cpad);
    const size_t threads = (size_t)rows * 4;
    hipLaunchKernelGGL((head_out_kernel<T>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, w, bias, out,
                       rows, Cpad);
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_head_out_f16_f32(const void* in, const float* w, float bias, float* out, int rows, int Cpad, vda_stream_t stream) {
    return head_out_launch<h16>((const h16*)in, w, bias, out, rows, Cpad, stream);
}

extern "C" int vda_head_out_f32_f32(const float* in, const float* w, float bias, float* out, long long rows, int Cpad, vda_stream_t stream) {
    return head_out_launch<float>(in, w, bias, out, rows, Cpad, stream);
}

extern "C" int vda_normalize_u8_f32(const uint8_t* frames, float* out, int n, int H, int W, vda_stream_t stream) {
    VDA_REQUIRE(frames && out && n > 0 && H > 0 && W > 0, "vda_normalize_u8: bad arguments");
    hipLaunchKernelGGL(normalize_u8_kernel, dim3(capped_grid((size_t)n * H * W)), dim3(256), 0, (hipStream_t)stream, frames, out, n, H, W);
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_gather_normalize_u8_f32(const uint8_t* video, const int32_t* idx, float* out, int n, int n_video, int H, int W,
                                           vda_stream_t stream) {
    VDA_REQUIRE(video && idx && out && n > 0 && n_video > 0 && H > 0 && W > 0, "vda_gather_normalize_u8: bad arguments");
    hipLaunchKernelGGL(gather_normalize_u8_kernel, dim3(capped_grid((size_t)n * H * W)), dim3(256), 0, (hipStream_t)stream, video,
                       (const int*)idx, out, n, H, W);
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_gather_resize_normalize_u8_f32(const uint8_t* video, const int32_t* idx, float* out, int n, int n_video, int H0,
                                                  int W0, int H, int W, vda_stream_t stream) {
    VDA_REQUIRE(video && idx && out && n > 0 && n_video > 0 && H0 > 0 && W0 > 0 && H > 0 && W > 0, "vda_gather_resize_normalize_u8: bad arguments");
    const float sy = (float)H0 / (float)H, sx = (float)W0 / (float)W;
    hipLaunchKernelGGL(gather_resize_normalize_kernel, dim3(capped_grid((size_t)n * H * W)), dim3(256), 0, (hipStream_t)stream, video,
                       (const int*)idx, out, n, H0, W0, H, W, sy, sx);
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_pos_embed_resample_f32(const float* pe, float* out, int g, int ph, int pw, int D, vda_stream_t stream) {
    VDA_REQUIRE(pe && out && g > 0 && ph > 0 && pw > 0 && D > 0, "vda_pos_embed_resample: bad arguments");
    // dinov2.py:194-203: scale_factor = ((ph + 0.1) / g, (pw + 0.1) / g) in Python floats; ATen maps a destination index with
    // the fp32 value of 1 / scale_factor
    const float inv_sy = (float)(1.0 / ((double)(ph + 0.1) / (double)g)), inv_sx = (float)(1.0 / ((double)(pw + 0.1) / (double)g));
    hipLaunchKernelGGL(pos_embed_resample_kernel, dim3(capped_grid((size_t)(1 + ph * pw) * D)), dim3(256), 0, (hipStream_t)stream, pe, out,
                       g, ph, pw, D, inv_sy, inv_sx);
    VDA_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int readout_concat_launch(const T* tok, T* out, int frames, int P, int D, vda_stream_t stream) {
    VDA_REQUIRE(tok && out && frames > 0 && P > 0 && D > 0 && D % 8 == 0, "vda_readout_concat: bad arguments");
    VDA_REQUIRE(((uintptr_t)tok & 15) == 0 && ((uintptr_t)out & 15) == 0, "vda_readout_concat: 16-byte alignment required");
    hipLaunchKernelGGL((readout_concat_kernel<T>), dim3(capped_grid((size_t)frames * P * 2 * (D / 8))), dim3(256), 0, (hipStream_t)stream, tok, out,
                       frames, P, D);
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_readout_concat_f16(const void* tok, void* out, int frames, int P, int D, vda_stream_t stream) {
    return readout_concat_launch<h16>((const h16*)tok, (h16*)out, frames, P, D, stream);
}

extern "C" int vda_readout_concat_f32(const float* tok, float* out, int frames, int P, int D, vda_stream_t stream) {
    return readout_concat_launch<float>(tok, out, frames, P, D, stream);
}
