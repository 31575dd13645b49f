// output_conv1 with the fusion pyramid's last 2x upsample folded in (dpt.py:117 applied to util/blocks.py:156-160's
// F.interpolate(scale_factor=2, mode="bilinear", align_corners=True) of refinenet1):
//     out[B, 2h, 2w, N] = conv3x3( bilinear2x(in[B, h, w, C]) ) + bias          NHWC fp16, N = 32 * CB <= 128
// as ONE kernel: the upsampled tensor (1.4 GB per ViT-L clip, written once and re-read nine times through L2 by the implicit GEMM)
// never exists. The structure is the patch-in-LDS direct convolution of conv_lds.hip / tail.hip, re-cut so that every staging
// step runs UNDER the MFMAs of the step before it:
//
//   workgroup = 16 x 32 output pixels, 8 waves (one workgroup per CU); a wave owns 2 output rows x all N channels
//               (acc = 2 x CB x 16 fp32 registers)
//   k-step    = 16 input channels = one K of v_mfma_f32_32x32x16_f16; every LDS image has 32-byte rows, the two 16-byte chunks
//               XOR-swizzled by (row >> 3) & 1 (rows 8 and 24 apart land in one ds_read_b128 lane group: conflict-free)
//   two-deep rings, all advanced once per k-step behind ONE barrier:
//     w[2]      9 taps x N couts x 16 ch of the step           <- LDS-DMA (L2 hits: every workgroup reads the same weights)
//     src[2]    the 11 x 19 source pixels under the patch      <- LDS-DMA, TWO steps ahead
//     patch[2]  the (16+2) x (32+2) upsampled pixels x 16 ch   <- interpolated from src[] by the VALU ONE step ahead, between
//               the MFMA groups of the running step (fp32 arithmetic on the four corners, one rounding to fp16: what the
//               unfused path would have read back from memory, up to the last bit of the fp32 sum)
//   MFMA      : A = weights [32 cout][16 k], B = patch [16 k][32 pixels], D[cout][pixel]; patch row R serves (output row R,
//               ky = 0), (R-1, ky = 1), (R-2, ky = 2): per kx a wave reads 4 patch + 3 CB weight fragments for 6 CB MFMAs.
//   epilogue  : + bias, fp16 NHWC store.
#include <type_traits>

#include "vda_common.h"

namespace {

constexpr int TH = 16, TW = 32;                // output tile
constexpr int PH = TH + 2, PW = TW + 2;        // patch with halo
constexpr int NPIX = PH * PW;                  // 612
constexpr int KC = 16;                         // channels per k-step
constexpr int ROWB = KC * 2;                   // 32-byte LDS rows
// source pixels under a patch at scale (h-1)/(2h-1) < 1/2: rows floor(ys*(y0-1)) .. floor(ys*(y0+16)) + 1, i.e. at most
// ceil(17/2) + 2 = 11 (columns: ceil(33/2) + 2 = 19)
constexpr int SH = 11, SW = 19, NSRC = SH * SW;
constexpr int NP_SRC = (NSRC + 31) / 32;       // 1-KiB DMA pieces (32 rows x 32 B)
constexpr int ZERO_OFF = NP_SRC * 1024;        // 16 bytes of zeros behind the pieces: what pixels outside the image interpolate from
constexpr int SRC_BYTES = NP_SRC * 1024 + 64;
constexpr int NP_PATCH = (NPIX + 31) / 32;
constexpr int PATCH_BYTES = NP_PATCH * 1024;
constexpr int NT = 512;
constexpr int NITEM = NPIX * 2;                // (patch pixel, 16-byte chunk) items per k-step
constexpr int NI = (NITEM + NT - 1) / NT;      // 3 per thread

__device__ __forceinline__ int swz(int r) { return (r >> 3) & 1; }
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ swz(row)) << 4); }
// [r4] the PATCH image is swizzled by the pixel's COLUMN in the patch, (col >> 3) & 1, instead of its row index: conflict-free for the
// same reason (the lanes of a ds_read_b128 group that share a row-mod-8 are 8 or 24 columns apart), and a reading lane's swizzle then
// depends on pixel + kx only: the patch fragment addresses of a k-step are 3 lane constants + immediates, the weight fragment addresses
// one (rows R = 32 (...) + cout: (R >> 3) & 1 = (cout >> 3) & 1), instead of ~70 VALU instructions per k-step and wave.
__device__ __forceinline__ int patch_off(int q, int chunk) {
    const int col = q % PW;
    return q * ROWB + ((chunk ^ ((col >> 3) & 1)) << 4);
}

template <int CB, int DBG = 0>
__global__ void __launch_bounds__(NT) conv3x3_up2_kernel(const h16* __restrict__ in, const h16* __restrict__ wt, const float* __restrict__ bias,
                                                         h16* __restrict__ out, int h, int w, int C, int N, int ldc, int tiles_x,
                                                         int tiles_y, int ntiles) {
    constexpr int W_ROWS = 9 * 32 * CB, NP_W = W_ROWS / 32, W_BYTES = NP_W * 1024;
    constexpr int WJ = (NP_W + 7) / 8;         // weight pieces per wave
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const wl = lds;                                        // [2][W_BYTES]
    char* const patch = lds + 2 * W_BYTES;                       // [2][PATCH_BYTES]
    char* const srcb = lds + 2 * W_BYTES + 2 * PATCH_BYTES;      // [2][SRC_BYTES]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = 2 * h, W = 2 * w;
    // workgroups are dealt to the 8 XCDs round-robin: give XCD x the contiguous tile range [x * per_xcd, (x+1) * per_xcd)
    const int per_xcd = gridDim.x >> 3;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= ntiles) return;                                  // uniform per workgroup
    const int tx = tile % tiles_x, tyb = tile / tiles_x;
    const int ty = tyb % tiles_y, b = tyb / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    const int px = lane & 31, hh = lane >> 5;
    const bool wide = (N & 7) == 0 && (ldc & 7) == 0 && ((uintptr_t)out & 15) == 0;     // uniform

    // ---- geometry (the same for every k-step). align_corners=True: src = dst * (in-1)/(out-1), as lerp_coord() of resample.hip
    const float ys = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, xs = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const int sy0 = min((int)(ys * (float)max(y0 - 1, 0)), h - 1), sx0 = min((int)(xs * (float)max(x0 - 1, 0)), w - 1);
    int ia[NI], ib[NI];                         // LDS offsets of the four corners (16 bits each)
    float wxs[NI], wys[NI];
    // Entry k of these arrays is item k in waves 0..3 and item k - 1 (mod 3) in waves 4..7: the interpolation slots of a k-step are
    // I M I M I M in the early waves and M I M I M I in the late ones (below), and with the late waves' entries rotated every slot
    // uses ONE entry index in both - a select between two entries per slot (round 3) made hipcc keep the arrays in scratch memory and
    // reload 5 dwords per slot (vector-memory loads whose waits, vmcnt being in order, also drained the step's LDS-DMA: the weights of
    // the NEXT step had to land before the running step's interpolation could go on).
    const bool early = wave < 4;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int item = tid + NT * (early ? k : (k + NI - 1) % NI);
        const int q = item >> 1, c = item & 1;
        const int py = q / PW, pxx = q - py * PW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
        const bool ok = item < NITEM && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const float sy = ys * (float)iy, sx = xs * (float)ix;
        const int ya = ok ? min((int)sy, h - 1) : sy0, xa = ok ? min((int)sx, w - 1) : sx0;
        const int yb = min(ya + 1, h - 1), xb = min(xa + 1, w - 1);
        wys[k] = sy - (float)ya;
        wxs[k] = sx - (float)xa;
        const int fa = min(max(ya - sy0, 0), SH - 1), fb = min(max(yb - sy0, 0), SH - 1);
        const int ga = min(max(xa - sx0, 0), SW - 1), gb = min(max(xb - sx0, 0), SW - 1);
        ia[k] = lds_off(fa * SW + ga, c) | (lds_off(fa * SW + gb, c) << 16);
        ib[k] = lds_off(fb * SW + ga, c) | (lds_off(fb * SW + gb, c) << 16);
        if (!ok) ia[k] = ib[k] = ZERO_OFF | (ZERO_OFF << 16);      // outside the image: four reads of the zero chunk = the conv's padding
    }
    // DMA sources: a piece = 32 LDS rows x 32 B; lane -> (row lr of the piece, LDS chunk lp), fetching source chunk lp ^ swz(row)
    const int lr = lane >> 1, lp = lane & 1;
    int src_goff;                               // halfs, k-step 0
    {
        const int row = min(wave, NP_SRC - 1) * 32 + lr;
        const int fy = row / SW, fx = row - fy * SW;
        const int gy = min(sy0 + fy, h - 1), gx = min(sx0 + fx, w - 1);
        src_goff = ((b * h + gy) * w + gx) * C + ((lp ^ swz(row)) << 3);
    }
    // weights: piece p holds rows R = 32 p + lr, i.e. tap p / CB and couts (p % CB) * 32 + lr: only the lane's swizzled chunk is kept
    // in a register, the rest is a handful of integer ops per piece and k-step (registers are what this kernel is short of)
    const int w_lane = (lp ^ swz(lr)) << 3;
    auto stage_w = [&](int step, char* buf) {
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            if (NP_W % 8 != 0 && wave + 8 * j >= NP_W) break;                       // wave-uniform
            const int piece = wave + 8 * j;
            const int co = min((piece % CB) * 32 + lr, N - 1);                       // couts >= N read row N - 1: never stored
            glds16_opaque(wt + (unsigned)(co * (9 * C) + (piece / CB) * C + step * KC + w_lane), buf + piece * 1024);
        }
    };
    auto stage_src = [&](int step, char* buf) {
        if (wave < NP_SRC) glds16_opaque(in + (unsigned)(src_goff + step * KC), buf + wave * 1024);
    };
    // one (patch pixel, chunk) item: out = a00*(1-wx)(1-wy) + a01*wx(1-wy) + a10*(1-wx)wy + a11*wx*wy in fp32 (four v_fma_mix per
    // channel), one rounding to fp16. Branch-free: items past the patch write into its padding rows, pixels outside the image
    // write zeros (the conv's padding).
    // where an item lands in the patch image: the same in every k-step. Items past the patch (the last 312 threads' third item) land
    // in the patch buffer's 28 padding rows
    int woff[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int item = tid + NT * (early ? k : (k + NI - 1) % NI), q = item < NITEM ? item >> 1 : NPIX + (lane & 15);
        woff[k] = patch_off(q, item & 1);
    }
    auto interp_item = [&](int wo, int iav, int ibv, float wx, float wy, const char* sb, char* pb) {
        const h16x8 a00 = *reinterpret_cast<const h16x8*>(sb + (iav & 0xffff)), a01 = *reinterpret_cast<const h16x8*>(sb + ((unsigned)iav >> 16));
        const h16x8 a10 = *reinterpret_cast<const h16x8*>(sb + (ibv & 0xffff)), a11 = *reinterpret_cast<const h16x8*>(sb + ((unsigned)ibv >> 16));
        const float ux = 1.f - wx, uy = 1.f - wy;
        const float w00 = ux * uy, w01 = wx * uy, w10 = ux * wy, w11 = wx * wy;
        h16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (h16)((float)a00[e] * w00 + (float)a01[e] * w01 + (float)a10[e] * w10 + (float)a11[e] * w11);
        *reinterpret_cast<h16x8*>(pb + wo) = o;
    };
    auto interp = [&](int k, const char* sb, char* pb) {
        if constexpr (DBG == 1) return;                        // TIMING EXPERIMENT 1 (results invalid): no interpolation
        interp_item(woff[k], ia[k], ib[k], wxs[k], wys[k], sb, pb);
    };

    f32x16 acc[2][CB];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][cb][e] = 0.f;

    const int nk = C / KC;
    // ---- prologue: weights and source of step 0, source of step 1; patch of step 0. The zero chunk behind each source buffer's
    // DMA pieces is written here once.
    if (tid < 2) *reinterpret_cast<h16x8*>(srcb + tid * SRC_BYTES + ZERO_OFF) = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
    stage_w(0, wl);
    stage_src(0, srcb);
    if (nk > 1) stage_src(1, srcb + SRC_BYTES);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NI; ++k) interp(k, srcb, patch);
    __syncthreads();

    // One k-step: the MFMAs of step j from patch[cur] / w[cur], and (INTERP) the patch of step j + 1 interpolated beside them.
    // A kx group's 6 CB MFMAs are one scheduling region (the next 32-cout block's weight fragments are requested before the running
    // block's six MFMAs). The three interpolation items sit BETWEEN the groups, one slot earlier in waves 0..3 than in waves 4..7
    // (I M I M I M against M I M I M I): waves w and w + 4 share a SIMD, so one partner's VALU phase runs under the other's MFMAs.
    // fragment addresses: lane constants (one per kx for the patch, one for the weights) + immediates
    int pbase[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) pbase[kx] = (wave * 2 * PW + px + kx) * ROWB + ((hh ^ (((px + kx) >> 3) & 1)) << 4);
    const int wbase = lds_off(px, hh);
    auto mfma_group = [&](int kx, const char* pc, const char* wc) {
        if constexpr (DBG == 2) return;                        // TIMING EXPERIMENT 2 (results invalid): no MFMA groups
        h16x8 P[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) P[i] = *reinterpret_cast<const h16x8*>(pc + pbase[kx] + i * PW * ROWB);
        h16x8 Wf[CB][3];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) Wf[cb][ky] = *reinterpret_cast<const h16x8*>(wc + wbase + ((ky * 3 + kx) * CB + cb) * 32 * ROWB);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[r][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[cb][ky], P[r + ky], acc[r][cb], 0, 0, 0);
        // ---- the order the scheduler is asked for
        __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);                          // P + the first block's weights
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if (cb + 1 < CB) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);     // the next block's weights
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                                          // nothing moves across groups (register pressure)
    };
    for (int j = 0; j + 1 < nk; ++j) {
        // the barrier behind us published w[cur], patch[cur], src[nxt]; w[nxt], patch[nxt], src[cur] are free
        stage_w(j + 1, wl + ((j + 1) & 1) * W_BYTES);
        if (j + 2 < nk) stage_src(j + 2, srcb + (j & 1) * SRC_BYTES);
        const int cur = j & 1, nxt = cur ^ 1;
        const char* const pc = patch + cur * PATCH_BYTES;
        const char* const wc = wl + cur * W_BYTES;
        const char* const sb = srcb + nxt * SRC_BYTES;
        char* const pb = patch + nxt * PATCH_BYTES;
        if (early) interp(0, sb, pb);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(0, pc, wc);
        interp(1, sb, pb);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(1, pc, wc);
        interp(2, sb, pb);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(2, pc, wc);
        if (!early) interp(0, sb, pb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this step's LDS-DMA landed before the barrier publishes it
        __syncthreads();
    }
    {
        const int cur = (nk - 1) & 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) mfma_group(kx, patch + cur * PATCH_BYTES, wl + cur * W_BYTES);
    }
    static_assert(NI == 3, "the interpolation items ride on the three kx groups");

    // ---- epilogue: lane = pixel (lane & 31) of row r; registers 4g..4g+3 of block cb = channels cb*32 + 8g + 4hh .. +3
    // [r4] WIDE form (N, ldc multiples of 8, out 16-byte aligned: every shape the model runs): the two lanes of a pixel (hh = 0 / 1)
    // hold the interleaved 4-channel groups 8g + 4hh of a 32-channel block; one v_permlane32_swap per register hands the lower lane
    // channels 0..15 and the upper lane 16..31, and each lane stores 2 x 16 bytes instead of 4 x 8: half the store instructions and
    // no 8-byte fragments for L2 to merge (the 8-byte form wrote 851 MB for 718 MB of output, profiles/r03).
    if (wide) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int oy = y0 + wave * 2 + r, ox = x0 + px;
            const bool inb = oy < H && ox < W;                  // (the swaps below run on every lane)
            const size_t row = ((size_t)(b * H + min(oy, H - 1)) * W + min(ox, W - 1)) * ldc;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                unsigned pk[4][2];                              // channel groups g = 0..3 as packed fp16 pairs
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = cb * 32 + 8 * g + 4 * hh;
                    f32x4 v = {acc[r][cb][4 * g], acc[r][cb][4 * g + 1], acc[r][cb][4 * g + 2], acc[r][cb][4 * g + 3]};
                    if (bias && n < N) v += *reinterpret_cast<const f32x4*>(bias + n);
                    const h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                    pk[g][0] = reinterpret_cast<const unsigned*>(&o)[0];
                    pk[g][1] = reinterpret_cast<const unsigned*>(&o)[1];
                }
                // A = groups 0, 1; B = groups 2, 3: the upper lanes' A goes to the lower lanes' B and back
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(pk[g][e], pk[g + 2][e], false, false);
                        pk[g][e] = sw[0];
                        pk[g + 2][e] = sw[1];
                    }
#pragma unroll
                for (int j = 0; j < 2; ++j) {                   // lower lane: channels 8j..8j+7 of the block, upper lane: 16 + 8j ..
                    const int n = cb * 32 + 16 * hh + 8 * j;
                    if (inb && n < N) {
                        const uint4 o = {pk[j][0], pk[j][1], pk[j + 2][0], pk[j + 2][1]};
                        *reinterpret_cast<uint4*>(out + row + n) = o;
                    }
                }
            }
        }
        return;
    }
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
                const h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                *reinterpret_cast<h16x4*>(out + row + n) = o;
            }
    }
}

static int g_up2_variant = 0;
template <int CB, int DBG = 0>
int launch_up2(const h16* in, const h16* wt, const float* bias, h16* out, int B, int h, int w, int C, int N, int ldc, hipStream_t s) {
    constexpr int smem = 2 * (9 * CB * 1024 + PATCH_BYTES + SRC_BYTES);
    static_assert(smem <= 160 * 1024, "LDS budget");
    static VdaKernelDeviceState dev_state;
    if (vda_prepare_kernel(reinterpret_cast<const void*>(&conv3x3_up2_kernel<CB, DBG>), smem, dev_state) < 0) return 2;
    const int tiles_x = (2 * w + TW - 1) / TW, tiles_y = (2 * h + TH - 1) / TH;
    const long long ntiles = (long long)tiles_x * tiles_y * B;
    VDA_REQUIRE(ntiles < (1ll << 30), "vda_conv3x3_up2: too many tiles");
    hipLaunchKernelGGL((conv3x3_up2_kernel<CB, DBG>), dim3((unsigned)((ntiles + 7) / 8 * 8)), dim3(NT), smem, s, in, wt, bias, out, h, w, C, N, ldc,
                       tiles_x, tiles_y, (int)ntiles);
    VDA_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int vda_conv3x3_up2_set_variant(int v) {
    g_up2_variant = v;
    return 0;
}

extern "C" int vda_conv3x3_up2_f16(const void* in, const void* w, const float* bias, void* out, int B, int h, int wd, int C, int N, int ldc,
                                   vda_stream_t stream) {
    VDA_REQUIRE(in && w && out, "vda_conv3x3_up2: null pointer");
    VDA_REQUIRE(B > 0 && h > 0 && wd > 0 && C > 0 && C % KC == 0, "vda_conv3x3_up2: bad geometry (C=%d must be a multiple of %d)", C, KC);
    VDA_REQUIRE(N > 0 && N <= 128 && N % 4 == 0 && ldc >= N && ldc % 4 == 0, "vda_conv3x3_up2: N=%d (at most 128, a multiple of 4), ldc=%d", N, ldc);
    VDA_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)out & 7) == 0 && ((uintptr_t)bias & 15) == 0,
                "vda_conv3x3_up2: alignment (16 bytes; out 8)");
    VDA_REQUIRE((double)B * h * wd * C < 2147483647.0 && (double)N * 9 * C < 2147483647.0 && (double)B * 4 * h * wd < 2147483647.0,
                "vda_conv3x3_up2: tensor exceeds 32-bit element offsets");
    hipStream_t s = (hipStream_t)stream;
    const h16 *ip = (const h16*)in, *wp = (const h16*)w;
    if (N <= 32) return launch_up2<1>(ip, wp, bias, (h16*)out, B, h, wd, C, N, ldc, s);
    if (N <= 64) return launch_up2<2>(ip, wp, bias, (h16*)out, B, h, wd, C, N, ldc, s);
    if (g_up2_variant == 1) return launch_up2<4, 1>(ip, wp, bias, (h16*)out, B, h, wd, C, N, ldc, s);      // timing experiments (tools/conv_up_variants.py)
    if (g_up2_variant == 2) return launch_up2<4, 2>(ip, wp, bias, (h16*)out, B, h, wd, C, N, ldc, s);
    return launch_up2<4>(ip, wp, bias, (h16*)out, B, h, wd, C, N, ldc, s);
}
