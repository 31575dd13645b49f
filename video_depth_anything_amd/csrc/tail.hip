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

// The bilinear sample as ONE weighted sum, a00 w00 + a01 w01 + a10 w10 + a11 w11 in packed fp16 (v_pk_mul + 3 v_pk_fma per channel pair:
// 16 VALU instructions per 8 channels against 24 for the nested lerp the first version used); the weights are formed in fp32 and
// rounded once. Both kernels below use it, so their results stay bit-identical.
// align_corners=True source position of output coordinate (iy, ix): corner indices and fractional offsets. No fused multiply-add,
// so that the two kernels (which inline this into different surroundings) compute the same bits.
__device__ __forceinline__ void bilinear_coord(float ys, float xs, int iy, int ix, int h, int w, int& ya, int& yb, int& xa, int& xb, float& wx,
                                               float& wy) {
#pragma clang fp contract(off)
    const float sy = ys * (float)iy, sx = xs * (float)ix;
    ya = max(min((int)sy, h - 1), 0);
    yb = min(ya + 1, h - 1);
    xa = max(min((int)sx, w - 1), 0);
    xb = min(xa + 1, w - 1);
    wx = sx - (float)xa;
    wy = sy - (float)ya;
}

struct BiW {
    h16x2 w00, w01, w10, w11;
};
__device__ __forceinline__ BiW bilinear_weights(float wx, float wy) {
#pragma clang fp contract(off)                          // one operation sequence in every caller: the two kernels agree to the bit
    const float ux = 1.f - wx, uy = 1.f - wy;
    float pa = ux * uy, pb = wx * uy, pc = ux * wy, pd = wx * wy;
    // (the products are rounded to fp32 FIRST: without this fence the backend may fuse multiply and conversion into one v_fma_mixlo_f16 -
    // a single rounding - in one kernel and not in the other: 1 fp16 ulp on a weight at ~1 pixel in 5 000, found by the bit-identity test)
    asm volatile("" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));
    const h16 a = (h16)pa, b = (h16)pb, c = (h16)pc, d = (h16)pd;
    return BiW{h16x2{a, a}, h16x2{b, b}, h16x2{c, c}, h16x2{d, d}};
}
__device__ __forceinline__ h16x8 bilinear8(const h16x8 a00, const h16x8 a01, const h16x8 a10, const h16x8 a11, const BiW& q) {
    h16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const h16x2 p00 = {a00[2 * i], a00[2 * i + 1]}, p01 = {a01[2 * i], a01[2 * i + 1]};
        const h16x2 p10 = {a10[2 * i], a10[2 * i + 1]}, p11 = {a11[2 * i], a11[2 * i + 1]};
        h16x2 o = p00 * q.w00;
        o = __builtin_elementwise_fma(p01, q.w01, o);
        o = __builtin_elementwise_fma(p10, q.w10, o);
        o = __builtin_elementwise_fma(p11, q.w11, o);
        r[2 * i] = o[0];
        r[2 * i + 1] = o[1];
    }
    return r;
}

template <int SRC_UP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) depth_tail_kernel(const h16* __restrict__ in, const h16* __restrict__ w2, const float* __restrict__ b2,
                                                         const float* __restrict__ w3, float b3, float* __restrict__ out,
                                                         const h16* __restrict__ zero_page, int h, int w, int H, int W, int C, int tiles_x,
                                                         int tiles_y, int ntiles, float ys, float xs) {
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
    float wxf[NK], wyf[NK];                                   // the sample's fractional position
    if constexpr (SRC_UP == 1) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int q = (tid >> 2) + 64 * k;
            const int py = q / PW, pxx = q - py * PW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
            const bool ok = q < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            int ya, yb, xa, xb;
            bilinear_coord(ys, xs, iy, ix, h, w, ya, yb, xa, xb, wxf[k], wyf[k]);
            off00[k] = ok ? ((b * h + ya) * w + xa) * C + ch * 8 : -1;
            dxo[k] = ok ? (xb - xa) * C : 0;
            dyo[k] = ok ? (yb - ya) * w * C : 0;
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
                    h16x8 o = bilinear8(a00[j], a01[j], a10[j], a11[j], bilinear_weights(wxf[k], wyf[k]));
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


// ---------------------------------------------------------------------------------------------------------------------------
// v2 (round 4), the resizing form the product runs (296^2 -> 518^2 at ViT-L): PERSISTENT workgroups with SPECIALISED waves - eight run
// MFMAs, eight build patches - and the source pixels under a patch staged ONCE in LDS.
//   v1 above refills 18 KiB of weights and the patch per pass and per workgroup, waits for them, and only then runs 36 MFMAs per wave:
//   MFMA busy 29 %, LDS active 22 %, TA busy 47 % (profiles/r04) - no pipe is the limit, the fill -> barrier -> MFMA chain of each
//   workgroup is, and three workgroups per CU do not hide it. Measured on the way here (profiles/r04/depth_tail_v2.txt):
//     - a persistent form with every wave doing both jobs (VALU work cut from 13 to 7 instructions per MFMA): -6 %; its parts ADD
//       (MFMA phase alone 488 us, + interpolation 207, + gathers 193): a wave issues one instruction stream;
//     - the same with separate MFMA and fill waves: -5 %, and "gathers only, no interpolation" still 902 us of 985: the four 16-byte
//       gathers per (pixel, chunk) item - 9 792 lane-loads per pass for 1 008 distinct source chunks - cost ~46 cycles per wave
//       instruction in the texture path, whatever else the CU does.
//   So the source region of a patch (12 x 21 pixels at 296 -> 518) comes in by LDS-DMA, one 16-byte chunk once, two passes ahead,
//   and the interpolation reads its four corners from LDS.
//   workgroup = 16 x 32 output pixels, 16 waves, one workgroup per CU, walking its XCD's share of the tiles:
//     waves 0..7  own 2 output rows each: per pass six groups (kx, k-step) of 6 MFMAs, the fragments of group g + 1 requested before the
//                 MFMAs of group g; no VALU work, no global loads; the tile's epilogue after its last pass
//     waves 8..15 build the patch of the NEXT unit (five (pixel, 16-byte chunk) items per lane: 4 ds_read_b128, 16 packed FMAs, one
//                 ds_write_b128), then put the weights of the next unit and the source region of the unit after it in flight
//   LDS       = 2 x weights of one pass [9 taps x 32 cout][32 ch] (18 KiB) + 2 x patch [640 rows][32 ch] (40 KiB; 18 x 34 = 612
//               pixels, the rest padding so every item has a row) + 2 x source region [<= 256 pixels][32 ch] (16 KiB + a zero row:
//               what pixels outside the image interpolate from = the conv's padding) = 148 KiB.
//   unit      = one pass (32 channels) of one tile; ONE barrier per unit publishes patch / weights / source of the next.
//   arithmetic: the same interpolation, the same MFMA order per output pixel as v1 - results are bit-identical to v1
//               (tests/test_kernels_gpu.py holds that).
namespace v2 {
constexpr int TH2 = 16, TW2 = 32, PH2 = TH2 + 2, PW2 = TW2 + 2, NPIX2 = PH2 * PW2;      // 612
constexpr int MMA_WAVES = 8, FILL_WAVES = 8, NT2 = (MMA_WAVES + FILL_WAVES) * 64;        // 1024 threads
constexpr int FILL_T = FILL_WAVES * 64;
constexpr int NKF = (NPIX2 * 4 + FILL_T - 1) / FILL_T;                                   // 5 items per fill lane and pass
constexpr int PROWS2 = NKF * (FILL_T / 4);                                               // 640 rows
constexpr int PATCH_BYTES2 = PROWS2 * ROWB;                                              // 40 KiB
constexpr int W_PASS_BYTES = 9 * 32 * ROWB;                                              // 18 KiB per pass
constexpr int SRC_ROWS = 256, ZERO_OFF = SRC_ROWS * ROWB, SRC_BYTES = ZERO_OFF + ROWB;   // 16 KiB + the zero row
constexpr int SMEM2 = 2 * W_PASS_BYTES + 2 * PATCH_BYTES2 + 2 * SRC_BYTES;               // 148 KiB
// patch rows are swizzled by their COLUMN in the patch, (col >> 2) & 3, not by the row index as in v1: conflict-free for the same
// reason (a ds_read_b128 lane group's 16 columns still land on 16 different 16-byte slots), and a reading lane's swizzle then depends
// on (pixel + kx) only - the 24 patch fragment addresses of a pass are 6 lane constants + immediates, the 18 weight fragment
// addresses 2 (v1 recomputes ~4 VALU instructions per fragment and pass)
__device__ __forceinline__ int swzc(int col) { return (col >> 2) & 3; }

// the source rows / columns a 16 x 32 tile's patch can touch, exactly as the kernel computes them (host side: sizes the source region)
inline int src_extent(float scale, int tile, int halo_lo, int out_n, int in_n) {
    int ext = 1;
    for (int o0 = 0; o0 < out_n; o0 += tile) {
        const int lo = o0 + halo_lo < 0 ? 0 : o0 + halo_lo;
        const int hi = o0 + tile < out_n - 1 ? o0 + tile : out_n - 1;
        int a = (int)(scale * (float)lo), b = (int)(scale * (float)hi) + 1;
        a = a < in_n - 1 ? a : in_n - 1;
        b = b < in_n - 1 ? b : in_n - 1;
        ext = b - a + 1 > ext ? b - a + 1 : ext;
    }
    return ext;
}

template <int DBG>
__global__ void __launch_bounds__(NT2) depth_tail_up_kernel(const h16* __restrict__ in, const h16* __restrict__ w2, const float* __restrict__ b2,
                                                            const float* __restrict__ w3, float b3, float* __restrict__ out, int h, int w, int H,
                                                            int W, int C, int tiles_x, int tiles_y, int ntiles, float ys, float xs, int SH, int SW) {
    extern __shared__ __attribute__((aligned(16))) char lds2[];
    char* const wring = lds2;                                   // [2][W_PASS_BYTES]
    char* const patch = lds2 + 2 * W_PASS_BYTES;                // [2][PATCH_BYTES2]
    char* const srcb = patch + 2 * PATCH_BYTES2;                // [2][SRC_BYTES]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npass = C / CC;
    const bool tail_prio = (SH >> 16) == 0;                    // (A/B switch riding on an argument's unused upper half: set = off)
    SH &= 0xffff;
    // the XCD's contiguous share of the tiles, dealt to its workgroups tile by tile: the workgroups of an XCD work on neighbouring tiles
    // at any moment (shared halo and source pixels in that XCD's L2)
    const int per_xcd = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tb = (int)((long long)ntiles * xcd / 8), te = (int)((long long)ntiles * (xcd + 1) / 8);
    if (tb + slot >= te) return;                               // uniform per workgroup
    const int nmy = (te - tb - slot + per_xcd - 1) / per_xcd;
    const int nu = nmy * npass;                                // units of this workgroup
    auto bar = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };
    // first source row / column under the patch of the tile at (y0, x0)
    auto src_origin = [&](int y0, int x0, int& sy0, int& sx0) __attribute__((always_inline)) {
        sy0 = min((int)(ys * (float)max(y0 - 1, 0)), h - 1);
        sx0 = min((int)(xs * (float)max(x0 - 1, 0)), w - 1);
    };

    if (wave >= MMA_WAVES) {
        // =================================================== fill waves ===================================================
        // The fill waves go AHEAD of the MFMA waves at the issue arbiter: they are the critical path to the barrier (their pass takes
        // longer than the 36 MFMAs of an MFMA wave), and an MFMA wave loses nothing by yielding an issue slot while its matrix pipe is
        // busy. 773 -> 740 us in one process (priority 2: 796 -> 772; the MFMA waves ahead instead: 777 -> 799). Variant bit 3 = off (A/B).
        if (tail_prio) __builtin_amdgcn_s_setprio(3);
        const int fw = wave - MMA_WAVES;                       // 0..7
        const int ft = tid - MMA_WAVES * 64;                   // 0..511
        const int ch = ft & 3, lr = lane >> 2;
        if (ft < 8) *reinterpret_cast<h16x8*>(srcb + (ft >> 2) * SRC_BYTES + ZERO_OFF + (ft & 3) * 16) = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
        // ---- DMA side: this wave's pieces (16 LDS rows x 64 B) of the weights (3 of 18) and of the source region (2 of <= 16)
        const int np_src = (SH * SW + 15) >> 4;
        unsigned w_goff[3], s_goff[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int R = (fw + 8 * j) * 16 + lr;              // row = tap * 32 + cout, swizzled by (R >> 2) & 3 = (cout >> 2) & 3
            w_goff[j] = (unsigned)((R & 31) * (9 * C) + (R >> 5) * C + ((ch ^ swz(R)) << 3));
        }
        auto dma_weights = [&](int pass, char* wb) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (fw + 8 * j < 18) glds16_opaque(w2 + (w_goff[j] + (unsigned)(pass * CC)), wb + (fw + 8 * j) * 1024);
        };
        auto dma_geometry = [&](int t) __attribute__((always_inline)) {
            asm volatile("; source geometry of a new tile" ::: "memory");     // (not speculable: keeps the once-per-tile work behind its branch)
            const int tx = t % tiles_x, tyb = t / tiles_x;
            const int ty = tyb % tiles_y, b = tyb / tiles_y;
            int sy0, sx0;
            src_origin(ty * TH2, tx * TW2, sy0, sx0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = (fw + 8 * j) * 16 + lr;
                const int fy = r / SW, fx = r - fy * SW;
                const int gy = min(sy0 + fy, h - 1), gx = min(sx0 + fx, w - 1);     // (rows past the region: some valid pixel, never read)
                s_goff[j] = (unsigned)(((b * h + gy) * w + gx) * C + ch * 8);
            }
        };
        auto dma_src = [&](int pass, char* sb) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (fw + 8 * j < np_src) glds16_opaque(in + (s_goff[j] + (unsigned)(pass * CC)), sb + (fw + 8 * j) * 1024);
        };
        // ---- interpolation side: this lane's five items; the LDS position in the patch is the same in every tile
        int woff[NKF];
#pragma unroll
        for (int k = 0; k < NKF; ++k) {
            const int q = (ft >> 2) + (FILL_T / 4) * k;
            const int py = q / PW2, pxx = q - py * PW2;
            woff[k] = q * ROWB + ((ch ^ swzc(pxx)) << 4);
        }
        unsigned ia[NKF], ib[NKF];                             // LDS offsets of the four corners in the source region, 16 bits each
        BiW bw[NKF];
        auto interp_geometry = [&](int t) __attribute__((always_inline)) {
            asm volatile("; patch geometry of a new tile" ::: "memory");      // (hipcc if-converted ~260 VALU instructions into every pass without it)
            const int tx = t % tiles_x, tyb = t / tiles_x;
            const int ty = tyb % tiles_y;
            const int x0 = tx * TW2, y0 = ty * TH2;
            int sy0, sx0;
            src_origin(y0, x0, sy0, sx0);
#pragma unroll
            for (int k = 0; k < NKF; ++k) {
                const int q = (ft >> 2) + (FILL_T / 4) * k;
                const int py = q / PW2, pxx = q - py * PW2;
                const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
                const bool ok = q < NPIX2 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                int ya, yb, xa, xb;
                float wx, wy;
                bilinear_coord(ys, xs, iy, ix, h, w, ya, yb, xa, xb, wx, wy);
                const int fa = min(max(ya - sy0, 0), SH - 1), fb = min(max(yb - sy0, 0), SH - 1);
                const int ga = min(max(xa - sx0, 0), SW - 1), gb = min(max(xb - sx0, 0), SW - 1);
                const unsigned c16 = (unsigned)ch * 16u;
                const unsigned z = (unsigned)ZERO_OFF + c16;
                ia[k] = ok ? ((unsigned)(fa * SW + ga) * ROWB + c16) | (((unsigned)(fa * SW + gb) * ROWB + c16) << 16) : (z | (z << 16));
                ib[k] = ok ? ((unsigned)(fb * SW + ga) * ROWB + c16) | (((unsigned)(fb * SW + gb) * ROWB + c16) << 16) : (z | (z << 16));
                bw[k] = bilinear_weights(wx, wy);
            }
        };
        auto interp = [&](const char* sb, char* pb) __attribute__((always_inline)) {
            if constexpr (DBG == 1) return;                    // TIMING EXPERIMENT 1: no interpolation (DMA + barriers only)
#pragma unroll
            for (int k = 0; k < NKF; ++k) {
                const h16x8 a00 = *reinterpret_cast<const h16x8*>(sb + (ia[k] & 0xffffu)), a01 = *reinterpret_cast<const h16x8*>(sb + (ia[k] >> 16));
                const h16x8 a10 = *reinterpret_cast<const h16x8*>(sb + (ib[k] & 0xffffu)), a11 = *reinterpret_cast<const h16x8*>(sb + (ib[k] >> 16));
                if constexpr (DBG == 3) {                      // TIMING EXPERIMENT 3: the LDS traffic of the interpolation without its arithmetic
                    uint4 x = *reinterpret_cast<const uint4*>(&a00);
                    const uint4 y = *reinterpret_cast<const uint4*>(&a01), z = *reinterpret_cast<const uint4*>(&a10), t = *reinterpret_cast<const uint4*>(&a11);
                    x.x ^= y.x ^ z.x ^ t.x; x.y ^= y.y ^ z.y ^ t.y; x.z ^= y.z ^ z.z ^ t.z; x.w ^= y.w ^ z.w ^ t.w;
                    *reinterpret_cast<uint4*>(pb + woff[k]) = x;
                    continue;
                }
                *reinterpret_cast<h16x8*>(pb + woff[k]) = bilinear8(a00, a01, a10, a11, bw[k]);
            }
        };
        // unit v of this workgroup = (tile tb + slot + (v / npass) * per_xcd, pass v % npass); the two sides run at different units
        int tileD = tb + slot, passD = 0;                      // DMA side: the unit whose source region is fetched next
        int tileI = tb + slot, passI = 0;                      // interpolation side: the unit whose patch is built next
        auto advance = [&](int& t, int& ps) __attribute__((always_inline)) {
            if (++ps == npass) {
                ps = 0;
                t += per_xcd;
            }
        };
        // ---- prologue: weights and source of unit 0; its patch; source of unit 1
        dma_geometry(tileD);
        dma_weights(0, wring);
        dma_src(0, srcb);
        advance(tileD, passD);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();
        interp_geometry(tileI);
        interp(srcb, patch);
        advance(tileI, passI);
        if (nu > 1) {
            if (passD == 0) dma_geometry(tileD);
            dma_src(passD, srcb + SRC_BYTES);
            advance(tileD, passD);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();
        if constexpr (DBG >= 4) {                              // TIMING EXPERIMENTS 4..6: the MFMA side alone
            for (int u = 0; u < nu; ++u)
                if (DBG != 6) bar();
            return;
        }
        for (int u = 0; u < nu; ++u) {
            // patch of unit u + 1 from source[(u + 1) & 1] (landed before the barrier behind us)
            // first the DMA - weights of unit u + 1 (its ring slot was last read in unit u - 1), source of unit u + 2 into the region
            // unit u's patch was built from - so that it lands under the interpolation
            if (u + 1 < nu) dma_weights(passI, wring + ((u + 1) & 1) * W_PASS_BYTES);
            if (u + 2 < nu) {
                if (passD == 0) dma_geometry(tileD);
                dma_src(passD, srcb + (u & 1) * SRC_BYTES);
                advance(tileD, passD);
            }
            if (u + 1 < nu) {
                if (passI == 0) interp_geometry(tileI);
                interp(srcb + ((u + 1) & 1) * SRC_BYTES, patch + ((u + 1) & 1) * PATCH_BYTES2);
                advance(tileI, passI);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bar();
        }
        return;
    }

    // ======================================================= MFMA waves =======================================================
    const int px = lane & 31, hh = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
    // fragment addresses: lane constants + immediates
    int pbase[3][2], wbase[2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) pbase[kx][ks] = (wave * 2 * PW2 + px + kx) * ROWB + (((2 * ks + hh) ^ swzc(px + kx)) << 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wbase[ks] = px * ROWB + (((2 * ks + hh) ^ swz(px)) << 4);
    struct Frag {
        h16x8 P[4], Wf[3];
    };
    auto load_frag = [&](int g, const char* pc, const char* wc, Frag& f) __attribute__((always_inline)) {
        const int kx = g >> 1, ks = g & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) f.P[j] = *reinterpret_cast<const h16x8*>(pc + pbase[kx][ks] + j * PW2 * ROWB);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) f.Wf[ky] = *reinterpret_cast<const h16x8*>(wc + wbase[ks] + (ky * 3 + kx) * 32 * ROWB);
    };
    auto mma = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.Wf[ky], f.P[r + ky], acc[r], 0, 0, 0);
    };
    int u = 0;
    auto unit = [&]() __attribute__((always_inline)) {
        if constexpr (DBG == 2) {                              // TIMING EXPERIMENT 2: no MFMA phase (the fill waves' own pace)
            ++u;
            return;
        }
        const char* const pc = patch + (u & 1) * PATCH_BYTES2;
        const char* const wc = wring + (u & 1) * W_PASS_BYTES;
        Frag f[2];
        load_frag(0, pc, wc, f[0]);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (g + 1 < 6 && DBG != 5) {
                load_frag(g + 1, pc, wc, f[(g + 1) & 1]);
                __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);      // the next group's fragments are requested first ...
            }
            mma(f[DBG == 5 ? 0 : g & 1]);                               // (experiment 5: no fragment reads but the first group's)
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);          // ... then this group's six MFMAs
            __builtin_amdgcn_sched_barrier(0);
        }
        ++u;
    };
    bar();                                                     // (prologue: weights and source of unit 0 landed)
    bar();                                                     // (prologue: patch of unit 0 built)
    int tile = tb + slot;
    for (int k = 0; k < nmy; ++k) {
        for (int pass = 0; pass + 1 < npass; ++pass) {
            unit();
            if (DBG != 6) bar();
        }
        unit();
        {
            // ---- epilogue of the tile: lane = pixel (lane & 31) of row r; registers = couts 8*(e>>2) + 4*hh + (e&3)
            const int tx = tile % tiles_x, tyb = tile / tiles_x;
            const int ty = tyb % tiles_y, b = tyb / tiles_y;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = 8 * (e >> 2) + 4 * hh + (e & 3);
                    s += fmaxf(acc[r][e] + b2[co], 0.f) * w3[co];
                    acc[r][e] = 0.f;
                }
                s += __shfl_xor(s, 32, 64);
                const int oy = ty * TH2 + wave * 2 + r, ox = tx * TW2 + px;
                if (hh == 0 && oy < H && ox < W) out[((size_t)b * H + oy) * W + ox] = fmaxf(s + b3, 0.f);
            }
        }
        tile += per_xcd;
        if (DBG != 6) bar();
    }
}
}  // namespace v2

}  // namespace

static int g_tail_variant = 0;
extern "C" int vda_depth_tail_set_variant(int v) {
    g_tail_variant = v;
    return 0;
}

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
    // align_corners=True scale, computed once on the host (an in-kernel division is not guaranteed the same bits in two kernels)
    const float ys = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, xs = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    // v2: persistent, specialised waves, source region staged in LDS (tools/tail_bench.py; variant 1 = v1 for A/B). Needs the region
    // under a 18 x 34 patch to fit 256 pixels - any upsampling by >= ~1.3 does (1.75 everywhere in the product)
    const int SH = v2::src_extent(ys, v2::TH2, -1, H, h), SW = v2::src_extent(xs, v2::TW2, -1, W, w);
    if ((h != H || w != W) && SH * SW <= v2::SRC_ROWS && (g_tail_variant & 7) != 1) {
        const int tx2 = (W + v2::TW2 - 1) / v2::TW2, ty2 = (H + v2::TH2 - 1) / v2::TH2;
        const long long nt2 = (long long)tx2 * ty2 * B;
        VDA_REQUIRE(nt2 < (1ll << 30), "vda_depth_tail: too many tiles");
        static VdaKernelDeviceState dev_state[8];
        const int dbg = (g_tail_variant >> 4) & 7;              // timing experiments (results invalid), tools/tail_variants.py
        using KFn = void (*)(const h16*, const h16*, const float*, const float*, float, float*, int, int, int, int, int, int, int, int, float, float, int, int);
        static const KFn fns[8] = {&v2::depth_tail_up_kernel<0>, &v2::depth_tail_up_kernel<1>, &v2::depth_tail_up_kernel<2>, &v2::depth_tail_up_kernel<3>,
                                   &v2::depth_tail_up_kernel<4>, &v2::depth_tail_up_kernel<5>, &v2::depth_tail_up_kernel<6>, &v2::depth_tail_up_kernel<0>};
        const KFn fn = fns[dbg];
        const int ncu = vda_prepare_kernel(reinterpret_cast<const void*>(fn), v2::SMEM2, dev_state[dbg]);
        if (ncu < 0) return 2;
        const int grid2 = (int)(nt2 < ncu ? (nt2 + 7) / 8 * 8 : ncu);
        hipLaunchKernelGGL(fn, dim3(grid2), dim3(v2::NT2), v2::SMEM2, s, (const h16*)in, (const h16*)w2, b2, w3, b3, out, h, w, H, W, C,
                           tx2, ty2, (int)nt2, ys, xs, SH | ((g_tail_variant & 8) ? 1 << 16 : 0), SW);
        VDA_LAUNCH_CHECK();
        return 0;
    }
    if (h == H && w == W)
        hipLaunchKernelGGL((depth_tail_kernel<0>), grid, dim3(256), 0, s, (const h16*)in, (const h16*)w2, b2, w3, b3, out, (const h16*)zero_page, h, w,
                           H, W, C, tiles_x, tiles_y, (int)ntiles, ys, xs);
    else
        hipLaunchKernelGGL((depth_tail_kernel<1>), grid, dim3(256), 0, s, (const h16*)in, (const h16*)w2, b2, w3, b3, out, (const h16*)zero_page, h, w,
                           H, W, C, tiles_x, tiles_y, (int)ntiles, ys, xs);
    VDA_LAUNCH_CHECK();
    return 0;
}
