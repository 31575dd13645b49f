// Depth tail of the DPT head for gfx950 (dpt.py:118-122, dpt_temporal.py:93-100, video_depth.py:162-163):
//     [bilinear align_corners resize h x w -> H x W]  ->  3x3 conv C -> 32 (+bias, ReLU)  ->  1x1 conv 32 -> 1 (+bias, ReLU)
// as ONE kernel with the input read once. Run as an implicit GEMM this stage re-reads every input pixel nine
// times through L2 to feed a 32-wide output (staging-bound, 290 TFLOP/s, plus a 2.2 GB upsampled tensor written
// and read back); here a workgroup keeps the (8+2) x (32+2) pixel patch it needs in LDS and forms all nine taps from it.
//
//   workgroup = 8 x 32 output pixels, 4 waves; a wave owns 2 rows = two 32-pixel MFMA column blocks
//   patch in LDS: [pixel][64 channels] per 64-channel pass (128 B rows, 16-byte chunks XOR-swizzled by (pixel>>1)&7:
//                 conflict-free for the 32-consecutive-pixel ds_read_b128 fragment at any offset)
//   patch fill:   SRC_UP = 0: 16-byte LDS-DMA straight from the tensor (zero page outside the image = conv padding)
//                 SRC_UP = 1: the bilinear resize is evaluated here from the low-res tensor and written with
//                             ds_write_b128 - the upsampled tensor never exists in memory
//   MFMA:         v_mfma_f32_32x32x16_f16, A = weights [32 cout][16 k] from L2 (72 KiB, every workgroup reads the same),
//                 B = patch [16 k][32 pixels]; D[cout][pixel]: a lane owns one pixel and 16 of its 32 couts
//   epilogue:     bias + ReLU, dot with the 32->1 weights in-lane + one lane^32 exchange, bias + ReLU, fp32 store
//                 (32 consecutive pixels per store instruction = full 128-byte lines)
#include "vda_common.h"

namespace {

constexpr int TH = 8, TW = 32;                 // output tile
constexpr int PH = TH + 2, PW = TW + 2;        // patch with halo
constexpr int NPIX = PH * PW;                  // 340
constexpr int NPIECE = (NPIX + 7) / 8;         // 1-KiB DMA pieces (8 pixels x 128 B) per pass
constexpr int CC = 64;                         // channels per pass
constexpr int PATCH_BYTES = NPIECE * 1024;

__device__ __forceinline__ int p_swz(int q) { return (q >> 1) & 7; }

template <int SRC_UP>
__global__ void __launch_bounds__(256) depth_tail_kernel(const h16* __restrict__ in, const h16* __restrict__ w2, const float* __restrict__ b2,
                                                         const float* __restrict__ w3, float b3, float* __restrict__ out,
                                                         const h16* __restrict__ zero_page, int h, int w, int H, int W, int C) {
    __shared__ __attribute__((aligned(16))) char patch[PATCH_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;
    const int px = lane & 31, hh = lane >> 5;

    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    const float ys = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, xs = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;

    for (int c0 = 0; c0 < C; c0 += CC) {
        if (c0 > 0) __syncthreads();                         // everyone done reading the previous pass's patch
        // ---- fill the patch for channels c0 .. c0+63
        if constexpr (SRC_UP == 0) {
            const int lq = lane >> 3, lpos = lane & 7;
            for (int piece = wave; piece < NPIECE; piece += 4) {
                const int q = piece * 8 + lq;                // patch pixel
                const int py = q / PW, pxx = q - py * PW;
                const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
                const bool ok = q < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const int schk = (lpos ^ p_swz(q)) * 8;
                const h16* src = ok ? in + (((size_t)b * H + iy) * W + ix) * C + c0 + schk : zero_page + schk;
                glds16(src, patch + piece * 1024);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // bilinear (align_corners=True) from the low-res tensor, 8 channels per item
            for (int it = tid; it < NPIX * 8; it += 256) {
                const int q = it >> 3, ch = it & 7;
                const int py = q / PW, pxx = q - py * PW;
                const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
                h16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    const float sy = ys * (float)iy, sx = xs * (float)ix;
                    const int ya = min((int)sy, h - 1), yb = min(ya + 1, h - 1);
                    const int xa = min((int)sx, w - 1), xb = min(xa + 1, w - 1);
                    const float wy = sy - (float)ya, wx = sx - (float)xa;
                    const h16* base = in + (size_t)b * h * w * C + c0 + ch * 8;
                    const h16x8 a00 = *reinterpret_cast<const h16x8*>(base + ((size_t)ya * w + xa) * C);
                    const h16x8 a01 = *reinterpret_cast<const h16x8*>(base + ((size_t)ya * w + xb) * C);
                    const h16x8 a10 = *reinterpret_cast<const h16x8*>(base + ((size_t)yb * w + xa) * C);
                    const h16x8 a11 = *reinterpret_cast<const h16x8*>(base + ((size_t)yb * w + xb) * C);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float top = (float)a00[e] * (1.f - wx) + (float)a01[e] * wx;
                        const float bot = (float)a10[e] * (1.f - wx) + (float)a11[e] * wx;
                        o[e] = (h16)(top * (1.f - wy) + bot * wy);      // same rounding point as the standalone resize (fp16 tensor)
                    }
                }
                *reinterpret_cast<h16x8*>(patch + q * 128 + ((ch ^ p_swz(q)) << 4)) = o;
            }
        }
        __syncthreads();

        // ---- 9 taps x 4 k-steps: D[cout][pixel] += W2[cout][tap, c0 + 16ks ..] . patch[pixel + tap][..]
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const h16* wrow = w2 + (size_t)px * (9 * C) + tap * C + c0 + hh * 8;      // A operand: row = cout (lane & 31)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h16x8 wf = *reinterpret_cast<const h16x8*>(wrow + ks * 16);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int q = (wave * 2 + r + ky) * PW + px + kx;                  // patch pixel of this lane's output pixel
                    const h16x8 pf = *reinterpret_cast<const h16x8*>(patch + q * 128 + (((2 * ks + hh) ^ p_swz(q)) << 4));
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, pf, acc[r], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: lane = pixel (lane & 31) of row r; registers = couts 8*(e>>2) + 4*hh + (e&3)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = 8 * (e >> 2) + 4 * hh + (e & 3);
            s += fmaxf(acc[r][e] + b2[co], 0.f) * w3[co];
        }
        s += __shfl_xor(s, 32, 64);
        const int oy = y0 + wave * 2 + r, ox = x0 + px;
        if (hh == 0 && oy < H && ox < W) out[((size_t)b * H + oy) * W + ox] = fmaxf(s + b3, 0.f);
    }
}

}  // namespace

extern "C" int vda_depth_tail_f16(const void* in, const void* w2, const float* b2, const float* w3, float b3, float* out,
                                  const void* zero_page, int B, int h, int w, int H, int W, int C, vda_stream_t stream) {
    VDA_REQUIRE(in && w2 && b2 && w3 && out && zero_page, "vda_depth_tail: null pointer");
    VDA_REQUIRE(B > 0 && B <= 65535 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % CC == 0, "vda_depth_tail: bad geometry (C=%d must be a multiple of %d)", C, CC);
    VDA_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)w2 & 15) == 0, "vda_depth_tail: 16-byte alignment required");
    const dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, B);
    hipStream_t s = (hipStream_t)stream;
    if (h == H && w == W)
        hipLaunchKernelGGL((depth_tail_kernel<0>), grid, dim3(256), 0, s, (const h16*)in, (const h16*)w2, b2, w3, b3, out, (const h16*)zero_page, h, w,
                           H, W, C);
    else
        hipLaunchKernelGGL((depth_tail_kernel<1>), grid, dim3(256), 0, s, (const h16*)in, (const h16*)w2, b2, w3, b3, out, (const h16*)zero_page, h, w,
                           H, W, C);
    VDA_LAUNCH_CHECK();
    return 0;
}
