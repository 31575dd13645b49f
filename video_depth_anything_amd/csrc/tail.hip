// Depth tail of the DPT head for gfx950 (dpt.py:118-122, dpt_temporal.py:93-100, video_depth.py:162-163):
//     [bilinear align_corners resize h x w -> H x W]  ->  3x3 conv C -> 32 (+bias, ReLU)  ->  1x1 conv 32 -> 1 (+bias, ReLU)
// as ONE kernel with the input read once. Run as an implicit GEMM this stage re-reads every input pixel nine
// times through L2 to feed a 32-wide output (staging-bound, 290 TFLOP/s, plus a 2.2 GB upsampled tensor written
// and read back); here a workgroup keeps the (8+2) x (32+2) pixel patch it needs in LDS and forms all nine taps from it.
//
//   workgroup = 8 x 32 output pixels, 4 waves; a wave owns 2 output rows = two 32-pixel MFMA column blocks
//   pass      = 32 input channels: LDS holds the patch [340 pixels][32 ch] and the weights [9 taps x 32 cout][32 ch]
//               (64-byte rows; 16-byte chunks XOR-swizzled by (row >> 2) & 3: any 16 consecutive rows of one chunk
//               cover the sixteen 16-byte slots of a bank row = conflict-free ds_read_b128). 40 KiB -> 3 workgroups
//               per CU, so one workgroup's fill overlaps its neighbours' MFMA phase.
//   weights   : LDS-DMA per pass (L2 -> LDS, 18 KiB). Re-reading them from L2 per MFMA made v1 L2-bandwidth-bound.
//   patch fill: SRC_UP = 0: 16-byte LDS-DMA straight from the tensor (zero page outside the image = conv padding)
//               SRC_UP = 1: the bilinear resize is evaluated here: 4 gathers from the low-res tensor (offsets and
//                           weights precomputed once per workgroup), packed-fp16 lerp, ds_write_b128 - the
//                           upsampled tensor never exists in memory
//   MFMA      : v_mfma_f32_32x32x16_f16, A = weights [32 cout][16 k], B = patch [16 k][32 pixels]; D[cout][pixel].
//               Patch row R serves (output row R, ky=0), (R-1, ky=1), (R-2, ky=2): per (kx, k-step) a wave reads
//               4 patch-row fragments + 3 weight fragments for 6 MFMAs.
//   epilogue  : bias + ReLU, dot with the 32->1 weights in-lane + one lane^32 exchange, bias + ReLU, fp32 store
//               (32 consecutive pixels per store instruction = full 128-byte lines)
//   launch    : 1-D grid, workgroup id -> tile remapped so that the workgroups of one XCD take consecutive tiles
//               (neighbouring tiles share halo / source pixels in that XCD's L2).
#include "vda_common.h"

namespace {

constexpr int TH = 8, TW = 32;                 // output tile
constexpr int PH = TH + 2, PW = TW + 2;        // patch with halo
constexpr int NPIX = PH * PW;                  // 340
constexpr int CC = 32;                         // channels per pass
constexpr int ROWB = CC * 2;                   // 64-byte LDS rows
constexpr int NP_PATCH = (NPIX + 15) / 16;     // 1-KiB DMA pieces (16 rows x 64 B)
constexpr int PATCH_BYTES = NP_PATCH * 1024;
constexpr int NP_W = 9 * 32 / 16;
constexpr int W_BYTES = NP_W * 1024;
constexpr int NK = (NPIX + 63) / 64;           // bilinear items per thread and pass

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

// ds_read_b128 is served 16 lanes at a time against one 256-byte bank row = sixteen 16-byte slots. Row r of 64 bytes puts chunk c
// at slot 4 * (r & 3) + c: 16 consecutive rows of one chunk must hit 16 different slots, so rows r, r+4, r+8, r+12 need 4 different
// chunk positions -> XOR with (r >> 2) & 3. (The first version used (r >> 1) & 3: 8 slots hit twice, 44 % conflict cycles measured.)
__device__ __forceinline__ int swz(int r) { return (r >> 2) & 3; }

// a + (b - a) * w on 8 halfs as four v_pk_add_f16 + four v_pk_fma_f16 (w = one VGPR holding the weight twice)
__device__ __forceinline__ h16x8 lerp8(const h16x8 a, const h16x8 b, const h16x2 w) {
    h16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const h16x2 av = {a[2 * i], a[2 * i + 1]}, bv = {b[2 * i], b[2 * i + 1]};
        const h16x2 o = av + (bv - av) * w;
        r[2 * i] = o[0];
        r[2 * i + 1] = o[1];
    }
    return r;
}

template <int SRC_UP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) depth_tail_kernel(const h16* __restrict__ in, const h16* __restrict__ w2, const float* __restrict__ b2,
                                                         const float* __restrict__ w3, float b3, float* __restrict__ out,
                                                         const h16* __restrict__ zero_page, int h, int w, int H, int W, int C, int tiles_x,
                                                         int tiles_y, int ntiles) {
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

    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    // ---- bilinear (align_corners=True) geometry of this thread's patch pixels: the same for every pass
    const int ch = tid & 3;                                   // 16-byte chunk (8 channels) of the pass
    int off00[NK], dxo[NK], dyo[NK];                          // halfs: (ya, xa) corner, +1 column, +1 row; off00 < 0 = outside the image
    h16x2 wxh[NK], wyh[NK];                                   // the lerp weight in both halves of one VGPR
    if constexpr (SRC_UP == 1) {
        const float ys = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, xs = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int q = (tid >> 2) + 64 * k;
            const int py = q / PW, pxx = q - py * PW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
            const bool ok = q < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const float sy = ys * (float)iy, sx = xs * (float)ix;
            const int ya = max(min((int)sy, h - 1), 0), yb = min(ya + 1, h - 1);
            const int xa = max(min((int)sx, w - 1), 0), xb = min(xa + 1, w - 1);
            off00[k] = ok ? ((b * h + ya) * w + xa) * C + ch * 8 : -1;
            dxo[k] = ok ? (xb - xa) * C : 0;
            dyo[k] = ok ? (yb - ya) * w * C : 0;
            const h16 wx = (h16)(sx - (float)xa), wy = (h16)(sy - (float)ya);
            wxh[k] = h16x2{wx, wx};
            wyh[k] = h16x2{wy, wy};
        }
    }

    for (int c0 = 0; c0 < C; c0 += CC) {
        if (c0 > 0) __syncthreads();                         // everyone done reading the previous pass's patch / weights
        // ---- weights of this pass: rows R = tap * 32 + cout, 32 channels each
        {
            const int lr = lane >> 2, lp = lane & 3;
            for (int piece = wave; piece < NP_W; piece += 4) {
                const int R = piece * 16 + lr;
                const int tap = R >> 5, co = R & 31;
                glds16(w2 + co * (9 * C) + tap * C + c0 + ((lp ^ swz(R)) << 3), wl + piece * 1024);
            }
        }
        // ---- patch of this pass
        if constexpr (SRC_UP == 0) {
            const int lr = lane >> 2, lp = lane & 3;
            for (int piece = wave; piece < NP_PATCH; piece += 4) {
                const int q = piece * 16 + lr;
                const int py = q / PW, pxx = q - py * PW;
                const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
                const bool ok = q < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const int sc = (lp ^ swz(q)) << 3;
                const h16* src = ok ? in + ((b * H + iy) * W + ix) * C + c0 + sc : zero_page + sc;
                glds16(src, patch + piece * 1024);
            }
        } else {
            // three batches of 2 patch pixels: all 8 gathers of a batch are issued before its first lerp (loads are
            // unconditional - pixels outside the image read the tensor's first bytes and are zeroed afterwards)
#pragma unroll
            for (int k0 = 0; k0 < NK; k0 += 2) {
                h16x8 a00[2], a01[2], a10[2], a11[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = k0 + j;
                    const unsigned o00 = (unsigned)(max(off00[k], 0) + c0);       // uniform base + 32-bit lane offset
                    a00[j] = *reinterpret_cast<const h16x8*>(in + (size_t)o00);
                    a01[j] = *reinterpret_cast<const h16x8*>(in + (size_t)(o00 + (unsigned)dxo[k]));
                    a10[j] = *reinterpret_cast<const h16x8*>(in + (size_t)(o00 + (unsigned)dyo[k]));
                    a11[j] = *reinterpret_cast<const h16x8*>(in + (size_t)(o00 + (unsigned)(dyo[k] + dxo[k])));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = k0 + j;
                    const int q = (tid >> 2) + 64 * k;
                    const h16x8 top = lerp8(a00[j], a01[j], wxh[k]);
                    const h16x8 bot = lerp8(a10[j], a11[j], wxh[k]);
                    h16x8 o = lerp8(top, bot, wyh[k]);
                    if (off00[k] < 0) o = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    if (q < NPIX) *reinterpret_cast<h16x8*>(patch + q * ROWB + ((ch ^ swz(q)) << 4)) = o;
                }
                asm volatile("" ::: "memory");                                   // keep the next batch's gathers behind this batch's lerps (VGPR budget)
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // LDS-DMA landed before the barrier publishes it
        __syncthreads();

        // ---- 3 kx x 2 k-steps: D[cout][pixel] += W2[cout][(ky,kx), c0 + 16ks ..] . patch[row + ky][pixel + kx][..]
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h16x8 P[4], Wf[3];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = (wave * 2 + j) * PW + px + kx;
                    P[j] = *reinterpret_cast<const h16x8*>(patch + q * ROWB + (((2 * ks + hh) ^ swz(q)) << 4));
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int R = (ky * 3 + kx) * 32 + px;                  // A operand: row = cout (lane & 31)
                    Wf[ky] = *reinterpret_cast<const h16x8*>(wl + R * ROWB + (((2 * ks + hh) ^ swz(R)) << 4));
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int r = 0; r < 2; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[ky], P[r + ky], acc[r], 0, 0, 0);
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
    VDA_REQUIRE(B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % CC == 0, "vda_depth_tail: bad geometry (C=%d must be a multiple of %d)", C, CC);
    VDA_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)w2 & 15) == 0, "vda_depth_tail: 16-byte alignment required");
    VDA_REQUIRE((double)B * h * w * C < 2147483647.0 && (double)B * H * W < 2147483647.0, "vda_depth_tail: tensor exceeds 32-bit element offsets");
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const long long ntiles = (long long)tiles_x * tiles_y * B;
    VDA_REQUIRE(ntiles < (1ll << 30), "vda_depth_tail: too many tiles");
    const dim3 grid((unsigned)((ntiles + 7) / 8 * 8));
    hipStream_t s = (hipStream_t)stream;
    if (h == H && w == W)
        hipLaunchKernelGGL((depth_tail_kernel<0>), grid, dim3(256), 0, s, (const h16*)in, (const h16*)w2, b2, w3, b3, out, (const h16*)zero_page, h, w,
                           H, W, C, tiles_x, tiles_y, (int)ntiles);
    else
        hipLaunchKernelGGL((depth_tail_kernel<1>), grid, dim3(256), 0, s, (const h16*)in, (const h16*)w2, b2, w3, b3, out, (const h16*)zero_page, h, w,
                           H, W, C, tiles_x, tiles_y, (int)ntiles);
    VDA_LAUNCH_CHECK();
    return 0;
}
