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


// ---------------------------------------------------------------------------------------------------------------------------
// [r4] The same convolution for C = 64 input channels (every 3x3 conv of the ViT-S head but the fused output_conv1) as a PERSISTENT
// kernel with SPECIALISED waves - the structure the depth tail got this round (tail.hip), without the interpolation:
//   The kernel above fills 22 KiB of patch and 36 KiB of weights per 32-channel pass and workgroup, waits, runs 72 MFMAs per wave, and
//   does it again for the second pass: 448 TFLOP/s on the 148^2 maps (115 us), two workgroups per CU.
//   Here ALL 64 channels of a pixel are one 128-byte LDS row: the whole weight tensor (9 taps x 64 cout x 64 ch = 72 KiB) is loaded ONCE
//   per workgroup, the (8+2) x (32+2) patch of the NEXT tile (43 KiB) arrives by LDS-DMA from four fill waves while four MFMA waves
//   (2 output rows each, 144 MFMAs per tile, fragments of group g + 1 requested before the MFMAs of group g) work on the current one;
//   one barrier per tile. LDS = 72 + 2 x 43 KiB. Rows are swizzled by the pixel's COLUMN ((col >> 1) & 7 on the eight 16-byte chunks:
//   conflict-free ds_read_b128, fragment addresses = lane constants + immediates).
//   Same MFMA order per output element as the kernel above (channel half, kx, k-step, ky): bit-identical results
//   (tests/test_kernels_gpu.py holds that; vda_conv_lds_set_variant(1) = the kernel above for every shape).
namespace c64 {
constexpr int C64 = 64, ROW64 = C64 * 2;       // 128-byte rows
constexpr int NP64 = (NPIX + 7) / 8;           // 43 DMA pieces of 8 rows
constexpr int PATCH64 = NP64 * 1024;           // 43 KiB
constexpr int MMA_W = 4, FILL_W = 4, NT64 = (MMA_W + FILL_W) * 64;
__device__ __forceinline__ int swz64(int col) { return (col >> 1) & 7; }

template <int CB, int DBG = 0>
__global__ void __launch_bounds__(NT64) conv3x3_c64_kernel(const h16* __restrict__ in, const h16* __restrict__ wt, const float* __restrict__ bias,
                                                           const h16* __restrict__ res, const h16* __restrict__ res2, h16* __restrict__ out,
                                                           const h16* __restrict__ zero_page, int H, int W, int N, int ldc, int relu_in,
                                                           int relu_out, int tiles_x, int tiles_y, int ntiles) {
    constexpr int W_ROWS = 9 * 32 * CB, NP_W = W_ROWS / 8, W_BYTES = NP_W * 1024;       // 36 KiB per 32 output channels
    extern __shared__ __attribute__((aligned(16))) char lds64[];
    char* const wl = lds64;
    char* const patch = lds64 + W_BYTES;                        // [2][PATCH64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per_xcd = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tb = (int)((long long)ntiles * xcd / 8), te = (int)((long long)ntiles * (xcd + 1) / 8);
    if (tb + slot >= te) return;                               // uniform per workgroup
    const int nmy = (te - tb - slot + per_xcd - 1) / per_xcd;
    auto bar = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };
    // ---- weights, once: rows R = tap * (32 * CB) + cout (couts >= N read row N - 1: never stored), swizzled by (cout >> 1) & 7
    {
        const int lr = lane >> 3, lp = lane & 7;
        for (int piece = wave; piece < NP_W; piece += NT64 / 64) {
            const int R = piece * 8 + lr;
            const int tap = R / (32 * CB), co = R - tap * (32 * CB);
            glds16_opaque(wt + (min(co, N - 1) * (9 * C64) + tap * C64 + ((lp ^ swz64(co)) << 3)), wl + piece * 1024);
        }
    }

    if (wave >= MMA_W) {
        // =================================================== fill waves ===================================================
        const int fw = wave - MMA_W;
        const int lr = lane >> 3, lp = lane & 7;
        auto fill = [&](int t, char* pb) __attribute__((always_inline)) {
            const int tx = t % tiles_x, tyb = t / tiles_x;
            const int ty = tyb % tiles_y, b = tyb / tiles_y;
            const int x0 = tx * TW, y0 = ty * TH;
            if constexpr (DBG == 1) return;                    // TIMING EXPERIMENT 1: no patch DMA
            for (int piece = fw; piece < NP64; piece += FILL_W) {
                const int q = piece * 8 + lr;
                const int py = q / PW, pxx = q - py * PW;
                const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
                const bool ok = q < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const int sc = (lp ^ swz64(pxx)) << 3;
                const h16* src = ok ? in + ((size_t)((b * H + iy) * W + ix) * C64 + sc) : zero_page + sc;      // zero page = the conv's padding
                glds16_opaque(src, pb + piece * 1024);
            }
        };
        int tile = tb + slot;
        fill(tile, patch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();
        for (int k = 0; k < nmy; ++k) {
            if (k + 1 < nmy) {
                tile += per_xcd;
                fill(tile, patch + ((k + 1) & 1) * PATCH64);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bar();
        }
        return;
    }

    // ======================================================= MFMA waves =======================================================
    const int px = lane & 31, hh = lane >> 5;
    const h16 floor_v = relu_in ? (h16)0.f : (h16)(-65504.f);
    h16x8 relu_thr;
#pragma unroll
    for (int e = 0; e < 8; ++e) relu_thr[e] = floor_v;
    f32x16 acc[2][CB];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][cb][e] = 0.f;
    // fragment addresses: lane constants (per kx and k-step: the XOR swizzle is not an immediate) + immediates for rows / taps / blocks
    int pbase[3][4], wbase[4];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) pbase[kx][ks] = (wave * 2 * PW + px + kx) * ROW64 + (((2 * ks + hh) ^ swz64(px + kx)) << 4);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wbase[ks] = px * ROW64 + (((2 * ks + hh) ^ swz64(px)) << 4);
    struct Frag {
        h16x8 P[4], Wf[CB][3];
    };
    // group g = (channel half, kx, k-step inside the half): the accumulation order of conv3x3_lds_kernel's two passes
    auto load_frag = [&](int g, const char* pc, Frag& f) __attribute__((always_inline)) {
        const int half = g / 6, rem = g - half * 6, kx = rem >> 1, ks = half * 2 + (rem & 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) f.P[j] = *reinterpret_cast<const h16x8*>(pc + pbase[kx][ks] + j * PW * ROW64);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) f.Wf[cb][ky] = *reinterpret_cast<const h16x8*>(wl + wbase[ks] + ((ky * 3 + kx) * (32 * CB) + cb * 32) * ROW64);
    };
    auto mma = [&](Frag& f) __attribute__((always_inline)) {
        if constexpr (DBG == 3) return;                        // TIMING EXPERIMENT 3: no MFMAs (fill + epilogue + barriers)
        // (the input ReLU is applied where the fragment is USED: at the load it would make the wave wait for the prefetch it just issued)
#pragma unroll
        for (int j = 0; j < 4; ++j) f.P[j] = __builtin_elementwise_max(f.P[j], relu_thr);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[r][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.Wf[cb][ky], f.P[r + ky], acc[r][cb], 0, 0, 0);
    };
    const bool wide = DBG != 2 && (N & 7) == 0 && (ldc & 7) == 0 && (((uintptr_t)out | (uintptr_t)res | (uintptr_t)res2) & 15) == 0;      // uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (this wave's share of the weights)
    bar();
    int tile = tb + slot;
    for (int k = 0; k < nmy; ++k) {
        const char* const pc = patch + (k & 1) * PATCH64;
        // The residual rows of this tile are requested NOW, in the wide (store) layout, and land under the tile's 144 MFMAs: with one
        // MFMA wave per SIMD nothing else would hide their HBM latency, and an epilogue that loads them where it uses them paid it four
        // times per tile (fill + epilogue alone, no MFMAs: 84 of 112 us at 148^2). The second residual tensor (the fusion blocks' skip
        // add: 3 of ViT-S's 18 convs) is read in the epilogue: its registers are the ones this prefetch takes.
        const int txp = tile % tiles_x, tybp = tile / tiles_x;
        const int typ = tybp % tiles_y, bp = tybp / tiles_y;
        uint4 rpre[2][CB][2];
        if (wide && res != nullptr) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int oy = typ * TH + wave * 2 + r, ox = txp * TW + px;
                const size_t row = ((size_t)(bp * H + min(oy, H - 1)) * W + min(ox, W - 1)) * ldc;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int j = 0; j < 2; ++j) rpre[r][cb][j] = *reinterpret_cast<const uint4*>(res + row + min(cb * 32 + 16 * hh + 8 * j, N - 8));
            }
        }
        Frag f[2];
        load_frag(0, pc, f[0]);
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            if (g + 1 < 12) {
                load_frag(g + 1, pc, f[(g + 1) & 1]);
                __builtin_amdgcn_sched_group_barrier(0x100, 4 + 3 * CB, 0);      // the next group's fragments are requested first ...
            }
            mma(f[g & 1]);
            __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);                  // ... then this group's input ReLU (16 packed max) ...
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * CB, 0);              // ... and its MFMAs
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: lane = pixel (lane & 31) of row r; registers 4g..4g+3 of block cb = channels cb*32 + 8g + 4hh .. +3.
        // WIDE form (N, ldc multiples of 8, 16-byte aligned tensors: every shape the model runs): the two lanes of a pixel hold the
        // interleaved 4-channel groups 8g + 4hh; one v_permlane32_swap per register gives the lower lane channels 0..15 of the block and
        // the upper lane 16..31, so residuals are READ and the result is WRITTEN 16 bytes per lane (the residual registers are swapped
        // back into the accumulator layout first: the swap is its own inverse). With 8-byte accesses the epilogue was the whole kernel:
        // 110 of 132 us at 148^2 (vda_conv_lds_set_variant 3), in this kernel and in the per-pass one alike.
        const int tx = tile % tiles_x, tyb = tile / tiles_x;
        const int ty = tyb % tiles_y, b = tyb / tiles_y;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int oy = ty * TH + wave * 2 + r, ox = tx * TW + px;
            const bool inb = oy < H && ox < W;
            const size_t row = ((size_t)(b * H + min(oy, H - 1)) * W + min(ox, W - 1)) * ldc;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                if (wide) {
                    // residuals: 2 x 16 bytes per lane and tensor in the STORE layout, swapped back into the accumulator layout
                    unsigned rr[2][4][2];                       // [tensor][group g][packed pair]
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const h16* rp = t == 0 ? res : res2;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int n = cb * 32 + 16 * hh + 8 * j;
                            uint4 q = {0u, 0u, 0u, 0u};
                            if (t == 0) {
                                if (rp != nullptr) q = rpre[r][cb][j];           // (rows / columns outside the tensor: clamped reads, never stored)
                            } else if (rp != nullptr && inb && n < N) {
                                q = *reinterpret_cast<const uint4*>(rp + row + n);
                            }
                            rr[t][j][0] = q.x;
                            rr[t][j][1] = q.y;
                            rr[t][j + 2][0] = q.z;
                            rr[t][j + 2][1] = q.w;
                        }
                        if (rp != nullptr) {
#pragma unroll
                            for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                                for (int e = 0; e < 2; ++e) {
                                    const auto sw = __builtin_amdgcn_permlane32_swap(rr[t][g2][e], rr[t][g2 + 2][e], false, false);
                                    rr[t][g2][e] = sw[0];
                                    rr[t][g2 + 2][e] = sw[1];
                                }
                        }
                    }
                    unsigned pk[4][2];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int n = cb * 32 + 8 * gq + 4 * hh;
                        f32x4 v = {acc[r][cb][4 * gq], acc[r][cb][4 * gq + 1], acc[r][cb][4 * gq + 2], acc[r][cb][4 * gq + 3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[r][cb][4 * gq + e] = 0.f;
                        if (bias && n < N) v += *reinterpret_cast<const f32x4*>(bias + n);
                        if (res) {
                            const h16x4 a = *reinterpret_cast<const h16x4*>(&rr[0][gq][0]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)a[e];
                        }
                        if (res2) {
                            const h16x4 a = *reinterpret_cast<const h16x4*>(&rr[1][gq][0]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)a[e];
                        }
                        if (relu_out) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        const h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                        pk[gq][0] = reinterpret_cast<const unsigned*>(&o)[0];
                        pk[gq][1] = reinterpret_cast<const unsigned*>(&o)[1];
                    }
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(pk[g2][e], pk[g2 + 2][e], false, false);
                            pk[g2][e] = sw[0];
                            pk[g2 + 2][e] = sw[1];
                        }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int n = cb * 32 + 16 * hh + 8 * j;
                        if (inb && n < N) {
                            const uint4 o = {pk[j][0], pk[j][1], pk[j + 2][0], pk[j + 2][1]};
                            *reinterpret_cast<uint4*>(out + row + n) = o;
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int n = cb * 32 + 8 * gq + 4 * hh;
                    f32x4 v = {acc[r][cb][4 * gq], acc[r][cb][4 * gq + 1], acc[r][cb][4 * gq + 2], acc[r][cb][4 * gq + 3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[r][cb][4 * gq + e] = 0.f;
                    if (!inb || n >= N || DBG == 2) continue;          // (TIMING EXPERIMENT 2: no epilogue memory traffic)
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
        tile += per_xcd;
        bar();
    }
}
}  // namespace c64

}  // namespace

static int g_conv_lds_variant = 0;
extern "C" int vda_conv_lds_set_variant(int v) {      // 0 = default (the C = 64 kernel where it applies), 1 = conv3x3_lds_kernel for every shape (A/B)
    g_conv_lds_variant = v;
    return 0;
}

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
    if (a.cCin == c64::C64 && a.N > 32 && g_conv_lds_variant != 1 && (double)a.cB * a.cH * a.cW * c64::C64 < 2147483647.0) {
        // C = 64 -> 33..64 channels: persistent, specialised waves, resident weights (c64::conv3x3_c64_kernel). In one process against the
        // per-pass kernel (tools/conv_lds_ab.py, profiles/r04/vits_conv_c64_ab.txt): 148^2 with a residual 129 -> 103 us, 74^2 39.4 -> 35.9,
        // 37^2 a tie, 19^2 13.3 -> 11.6; ViT-S forward 8.15 -> 8.05 ms. 32 output channels (not a shape of the model) lose on it
        // (one MFMA wave per SIMD has half the work per tile) and stay on the per-pass kernel.
        const int cb = a.N <= 32 ? 1 : 2;
        const int smem = 9 * 32 * cb * c64::ROW64 + 2 * c64::PATCH64;
        static VdaKernelDeviceState st1, st2;
        const void* fn = cb == 1 ? reinterpret_cast<const void*>(&c64::conv3x3_c64_kernel<1>) : reinterpret_cast<const void*>(&c64::conv3x3_c64_kernel<2>);
        const int ncu = vda_prepare_kernel(fn, smem, cb == 1 ? st1 : st2);
        if (ncu < 0) return 2;
        const int g64 = (int)(ntiles < ncu ? (ntiles + 7) / 8 * 8 : ncu);
        if (cb == 1)
            hipLaunchKernelGGL((c64::conv3x3_c64_kernel<1>), dim3(g64), dim3(c64::NT64), smem, s, (const h16*)a.A, (const h16*)a.W, a.bias, res, res2, (h16*)a.out,
                               (const h16*)a.zero_page, a.cH, a.cW, a.N, a.ldc, a.relu_in & 1, relu_out, tiles_x, tiles_y, (int)ntiles);
        else if (g_conv_lds_variant == 2)
            hipLaunchKernelGGL((c64::conv3x3_c64_kernel<2, 1>), dim3(g64), dim3(c64::NT64), smem, s, (const h16*)a.A, (const h16*)a.W, a.bias, res, res2, (h16*)a.out,
                               (const h16*)a.zero_page, a.cH, a.cW, a.N, a.ldc, a.relu_in & 1, relu_out, tiles_x, tiles_y, (int)ntiles);
        else if (g_conv_lds_variant == 4)
            hipLaunchKernelGGL((c64::conv3x3_c64_kernel<2, 3>), dim3(g64), dim3(c64::NT64), smem, s, (const h16*)a.A, (const h16*)a.W, a.bias, res, res2, (h16*)a.out,
                               (const h16*)a.zero_page, a.cH, a.cW, a.N, a.ldc, a.relu_in & 1, relu_out, tiles_x, tiles_y, (int)ntiles);
        else if (g_conv_lds_variant == 3)
            hipLaunchKernelGGL((c64::conv3x3_c64_kernel<2, 2>), dim3(g64), dim3(c64::NT64), smem, s, (const h16*)a.A, (const h16*)a.W, a.bias, res, res2, (h16*)a.out,
                               (const h16*)a.zero_page, a.cH, a.cW, a.N, a.ldc, a.relu_in & 1, relu_out, tiles_x, tiles_y, (int)ntiles);
        else
            hipLaunchKernelGGL((c64::conv3x3_c64_kernel<2>), dim3(g64), dim3(c64::NT64), smem, s, (const h16*)a.A, (const h16*)a.W, a.bias, res, res2, (h16*)a.out,
                               (const h16*)a.zero_page, a.cH, a.cW, a.N, a.ldc, a.relu_in & 1, relu_out, tiles_x, tiles_y, (int)ntiles);
        VDA_LAUNCH_CHECK();
        return 0;
    }
    if (a.N <= 32)
        hipLaunchKernelGGL((conv3x3_lds_kernel<1>), grid, dim3(256), 0, s, (const h16*)a.A, (const h16*)a.W, a.bias, res, res2, (h16*)a.out,
                           (const h16*)a.zero_page, a.cH, a.cW, a.cCin, a.N, a.ldc, a.relu_in & 1, relu_out, tiles_x, tiles_y, (int)ntiles);
    else
        hipLaunchKernelGGL((conv3x3_lds_kernel<2>), grid, dim3(256), 0, s, (const h16*)a.A, (const h16*)a.W, a.bias, res, res2, (h16*)a.out,
                           (const h16*)a.zero_page, a.cH, a.cW, a.cCin, a.N, a.ldc, a.relu_in & 1, relu_out, tiles_x, tiles_y, (int)ntiles);
    VDA_LAUNCH_CHECK();
    return 0;
}
