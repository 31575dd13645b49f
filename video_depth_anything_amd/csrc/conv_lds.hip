// 3x3 / stride-1 / pad-1 convolution for NARROW outputs (Cout = 32 or 64: the ViT-S head, util/blocks.py:20-32,79-84 and
// dpt.py:117) on NHWC fp16, as a patch-in-LDS direct convolution instead of an implicit GEMM.
//
// Why: with 64 output channels the implicit GEMM's tile is 128 x 64 and each 64-deep K tile gives a wave only 16 MFMAs between
// two barriers, while every input pixel is re-read nine times through L2 (300-450 TFLOP/s measured, tools/gemm_ab.py). Here a
// workgroup keeps the (8+2) x (32+2) pixel patch it needs in LDS and forms all nine taps from it - the structure of the depth
// tail (tail.hip), with a generic epilogue:
//
//   workgroup = 8 x 32 output pixels, 4 waves; a wave owns 2 output rows = two 32-pixel MFMA column blocks
//   pass      = 32 input channels: LDS holds the patch [340 pixels][32 ch] and the weights [9 taps x Cout][32 ch]
//               (64-byte rows, 16-byte chunks XOR-swizzled by (row >> 2) & 3: conflict-free ds_read_b128), both by LDS-DMA
//   MFMA      : v_mfma_f32_32x32x16_f16, A = weights [32 cout][16 k], B = patch [16 k][32 pixels]; D[cout][pixel]:
//               patch row R serves (output row R, ky=0), (R-1, ky=1), (R-2, ky=2)
//   epilogue  : + bias, + residual(s), ReLU, fp16 NHWC store (a lane owns one pixel and 4 consecutive channels per register group)
#include "vda_common.h"

namespace {

constexpr int TH = 8, TW = 32;                 // output tile
constexpr int PH = TH + 2, PW = TW + 2;        // patch with halo
constexpr int NPIX = PH * PW;                  // 340
constexpr int CC = 32;                         // channels per pass
constexpr int ROWB = CC * 2;                   // 64-byte LDS rows
constexpr int NP_PATCH = (NPIX + 15) / 16;     // 1-KiB DMA pieces (16 rows x 64 B)
constexpr int PATCH_BYTES = NP_PATCH * 1024;

__device__ __forceinline__ int swz(int r) { return (r >> 2) & 3; }

template <int CB>                              // 32-channel output blocks: Cout = 32 * CB
__global__ void __launch_bounds__(256) conv3x3_lds_kernel(const h16* __restrict__ in, const h16* __restrict__ wt, const float* __restrict__ bias,
                                                          const h16* __restrict__ res, const h16* __restrict__ res2, h16* __restrict__ out,
                                                          const h16* __restrict__ zero_page, int H, int W, int C, int N, int ldc, int relu_in,
                                                          int relu_out, int tiles_x, int tiles_y, int ntiles) {
    constexpr int NP_W = 9 * 32 * CB / 16;     // weight pieces per pass
    constexpr int W_BYTES = NP_W * 1024;
    __shared__ __attribute__((aligned(16))) char lds[PATCH_BYTES + W_BYTES];
    char* const patch = lds;
    char* const wl = lds + PATCH_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroups are dealt to the 8 XCDs round-robin: give XCD x the contiguous tile range [x * per_xcd, (x+1) * per_xcd)
    const int per_xcd = gridDim.x >> 3;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= ntiles) return;                               // uniform per workgroup
    const int tx = tile % tiles_x, tyb = tile / tiles_x;
    const int ty = tyb % tiles_y, b = tyb / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    const int px = lane & 31, hh = lane >> 5;

    f32x16 acc[2][CB];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][cb][e] = 0.f;

    const h16 floor_v = relu_in ? (h16)0.f : (h16)(-65504.f);
    h16x8 relu_thr;
#pragma unroll
    for (int e = 0; e < 8; ++e) relu_thr[e] = floor_v;

    const int lr = lane >> 2, lp = lane & 3;
    for (int c0 = 0; c0 < C; c0 += CC) {
        if (c0 > 0) __syncthreads();                         // everyone done reading the previous pass's patch / weights
        // weights of this pass: rows R = tap * (32 * CB) + cout, 32 channels each (rows of couts >= N read row N - 1: never stored)
        for (int piece = wave; piece < NP_W; piece += 4) {
            const int R = piece * 16 + lr;
            const int tap = R / (32 * CB), co = min(R - tap * (32 * CB), N - 1);
            glds16(wt + co * (9 * C) + tap * C + c0 + ((lp ^ swz(R)) << 3), wl + piece * 1024);
        }
        // patch of this pass (zero page outside the image = the conv's padding)
        for (int piece = wave; piece < NP_PATCH; piece += 4) {
            const int q = piece * 16 + lr;
            const int py = q / PW, pxx = q - py * PW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
            const bool ok = q < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int sc = (lp ^ swz(q)) << 3;
            const h16* src = ok ? in + ((b * H + iy) * W + ix) * C + c0 + sc : zero_page + sc;
            glds16(src, patch + piece * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // LDS-DMA landed before the barrier publishes it
        __syncthreads();

#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h16x8 P[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = (wave * 2 + j) * PW + px + kx;
                    P[j] = *reinterpret_cast<const h16x8*>(patch + q * ROWB + (((2 * ks + hh) ^ swz(q)) << 4));
                    P[j] = __builtin_elementwise_max(P[j], relu_thr);
                }
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    h16x8 Wf[3];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int R = (ky * 3 + kx) * (32 * CB) + cb * 32 + px;      // A operand: row = cout (lane & 31)
                        Wf[ky] = *reinterpret_cast<const h16x8*>(wl + R * ROWB + (((2 * ks + hh) ^ swz(R)) << 4));
                    }
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int r = 0; r < 2; ++r) acc[r][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[ky], P[r + ky], acc[r][cb], 0, 0, 0);
                }
            }
        }
    }

    // epilogue: lane = pixel (lane & 31) of row r; registers 4g..4g+3 of block cb = channels cb*32 + 8g + 4hh .. +3
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oy = y0 + wave * 2 + r, ox = x0 + px;
        if (oy >= H || ox >= W) continue;
        const size_t row = ((size_t)(b * H + oy) * W + ox) * ldc;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = cb * 32 + 8 * g + 4 * hh;
                if (n >= N) continue;
                f32x4 v = {acc[r][cb][4 * g], acc[r][cb][4 * g + 1], acc[r][cb][4 * g + 2], acc[r][cb][4 * g + 3]};
                if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
                if (res) {
                    const h16x4 a = *reinterpret_cast<const h16x4*>(res + row + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)a[e];
                }
                if (res2) {
                    const h16x4 a = *reinterpret_cast<const h16x4*>(res2 + row + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)a[e];
                }
                if (relu_out) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                const h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                *reinterpret_cast<h16x4*>(out + row + n) = o;
            }
    }
}

}  // namespace

// Called by vda_gemm_f16's dispatcher (gemm.hip) for VDA_A_CONV3X3 problems this kernel covers; returns -1 when it does not.
int vda_conv3x3_lds(const vda_gemm_args& a, hipStream_t s) {
    if (a.a_mode != VDA_A_CONV3X3 || a.cStride != 1 || a.N > 64 || a.N % 4 != 0 || a.cCin % CC != 0 || a.ldc % 4 != 0) return -1;
    if (a.epilogue != VDA_EPI_BIAS_F16 && a.epilogue != VDA_EPI_BIAS_RELU_F16 && a.epilogue != VDA_EPI_RES_F16) return -1;
    const int tiles_x = (a.cW + TW - 1) / TW, tiles_y = (a.cH + TH - 1) / TH;
    const long long ntiles = (long long)tiles_x * tiles_y * a.cB;
    if (ntiles >= (1ll << 30)) return -1;
    const dim3 grid((unsigned)((ntiles + 7) / 8 * 8));
    const h16* res = a.epilogue == VDA_EPI_RES_F16 ? (const h16*)a.res : nullptr;
    const h16* res2 = a.epilogue == VDA_EPI_RES_F16 ? (const h16*)a.res2 : nullptr;
    const int relu_out = a.epilogue == VDA_EPI_BIAS_RELU_F16 ? 1 : 0;
    if (a.N <= 32)
        hipLaunchKernelGGL((conv3x3_lds_kernel<1>), grid, dim3(256), 0, s, (const h16*)a.A, (const h16*)a.W, a.bias, res, res2, (h16*)a.out,
                           (const h16*)a.zero_page, a.cH, a.cW, a.cCin, a.N, a.ldc, a.relu_in & 1, relu_out, tiles_x, tiles_y, (int)ntiles);
    else
        hipLaunchKernelGGL((conv3x3_lds_kernel<2>), grid, dim3(256), 0, s, (const h16*)a.A, (const h16*)a.W, a.bias, res, res2, (h16*)a.out,
                           (const h16*)a.zero_page, a.cH, a.cW, a.cCin, a.N, a.ldc, a.relu_in & 1, relu_out, tiles_x, tiles_y, (int)ntiles);
    VDA_LAUNCH_CHECK();
    return 0;
}
