// One ViT block's MLP branch on the split residual stream in ONE kernel (fp16 path, LayerNorm folded):
//     x += gamma * (fc2(GELU(fc1(LayerNorm(x)))) + b2)            dinov2_layers/mlp.py:35-41, block.py:105-106, layer_scale.py:27-28
// for embedding widths whose output row fits a wave's accumulators (D = 384: ViT-S). Replaces the pair
// {fc1 + LN-fold + GELU -> hid [M, 4D] in HBM, fc2 + split-residual epilogue} of the unfused path (two launches, 5 + 3 rounds of the
// chip for 4.03 + 2.68 rounds of tiles at M = 43 840, hid written and read back: 270 MB).
//
// Structure (gfx950, 8 waves, one workgroup per CU, 16x16x32 fp16 MFMA with swapped operands as everywhere in this library):
//   * a wave owns 16 ROWS for the whole kernel. Its A operand (the hi plane, 16 x 384) sits in registers as 12 k-step fragments,
//     its output row block O[16 x 384] in 96 fp32 accumulators. Waves split rows only, so nothing a wave computes is needed by
//     another wave: no exchange through LDS, no transposition.
//   * the hidden dimension goes by in 24 chunks of 64. Per chunk: S[16 x 64] = A . W1c^T (48 MFMAs), LayerNorm fold + GELU on S in
//     registers (the row's (mean, rstd) are LANE constants in this layout: lane -> row), and S - rounded to fp16 - IS the A operand of
//     O += H . W2c^T (48 MFMAs): accumulator register e of column subtile j holds hidden unit 16 j + 4 (lane >> 4) + e, an MFMA
//     fragment wants 8 consecutive k per lane, so the two subtiles (2q, 2q + 1) of a lane make the fragment of k-step q if W2's
//     hidden columns are stored in the matching order - a pack-time permutation inside every 32 (vda_mlp_permute_w2_f16): hid never
//     exists, not even in LDS.
//   * what streams through LDS is the weights only: per chunk W1c [64 x 384] and W2c [384 x 64], 48 KiB each, by 16-byte LDS-DMA
//     into a ring of three 48 KiB buffers (tile t in buffer t % 3, issued two phases ahead, one counted vmcnt + one barrier per
//     phase); c1 / c2 of the LayerNorm fold (12 KiB) are staged once. 156 KiB of LDS.
//   * every MFMA needs one ds_read_b128 (a weight fragment; the activations are in registers): the LDS pipe and the matrix pipe are
//     equally loaded - the price of 16-row wave tiles; the two waves of a SIMD overlap one's reads and GELU with the other's MFMAs.
//   * rows: workgroups 0 .. n_full - 1 take 128 rows each; the remainder goes out in 64-ROW workgroups (waves 4..7 only help with
//     the DMA), so that the second, partial round of the chip runs one wave per SIMD - about half the time of a full one.
// Results are deterministic (fixed summation order, no atomics) and a row's result does not depend on its position.
//
// MEASURED (round 4, tools/mlp_fused_bench.py and tools/option_ab2.py, one process): the kernel is CORRECT and SLOWER than the pair
// it replaces - M = 32 768 (one round of 128-row workgroups) 139.8 us against 129.0 us, the 43 840-row clip 239 us against 180 us,
// the ViT-S forward 8.69 -> 9.00 ms with it. 302 MFLOP per workgroup in 140 us is 26 % of a CU's MFMA rate: with 16-row wave tiles
// every resource is asked for about one chunk-time of work per chunk at once - the matrix pipe (192 MFMAs x 16 cycles per SIMD =
// 3.1 k cycles), the LDS read port (832 ds_read_b128 = 3.3 k cycles), the LDS-DMA writes (96 KiB: 1.5 k) and the L2 channels (every CU
// of an XCD pulling the same 768 lines at the same moment: ~3 k) - and what the hardware overlaps of that is far from all of it
// (11.7 k cycles per chunk measured). The unfused pair keeps hid in the Infinity Cache anyway (135 MB at ViT-S's widths), so the
// HBM round trip the fusion removes was not there to be saved. vda_set_option("mlp_fused", 1) runs it; the default is off. A form
// with taller wave tiles (fewer weight-fragment reads per MFMA) needs O[32 x 384] = 192 accumulator registers per wave.
#include "vda_common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int D = 384, HID = 1536, HC = 64, NCHUNK = HID / HC;
// RS = 16-row subtiles per wave: RS = 1 -> 8 waves x 16 rows (two waves per SIMD, every MFMA needs its own weight-fragment read);
// RS = 2 -> 4 waves x 32 rows, ONE wave per SIMD with the whole 512-entry register file (O: 192 accumulators, A: 96 registers):
// a weight fragment feeds two MFMAs, so the LDS read port carries half the load. Measured (VDA_MLP_RS=2): 329 us for the clip against
// 224 us for RS = 1 - one wave per SIMD has nobody to hide its LDS round trips and GELU behind. RS = 1 is what the option runs.
constexpr int TILE_BYTES = 48 * 1024;                 // W1 chunk [64 x 384] = 6 K tiles of [64][128 B]; W2 chunk [384][128 B]
constexpr int C_BYTES = 2 * HID * 4;                  // c1 | c2
constexpr int SMEM = 3 * TILE_BYTES + C_BYTES;
constexpr int KSTEPS = D / 32;                        // 12 k-steps of GEMM 1
constexpr int NJ1 = HC / 16;                          // 4 column subtiles of S
constexpr int NJ2 = D / 16;                           // 24 column subtiles of O

template <int KEEP>
__device__ __forceinline__ void vm_wait_keep() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
}

struct Args {
    const h16* hi_in;         // A operand: the hi plane [M, D]
    const float* stats;       // [M, 2] (mean, rstd) of the LayerNorm in front of fc1; mean is also what the stream is re-centred by
    const h16* w1;            // [HID, D] fc1 weight with LayerNorm's scale folded in (vda_fold_ln_weight)
    const float* c1;          // [HID] row sums of w1
    const float* c2;          // [HID] b1 + W1 . ln_b
    const h16* w2p;           // [D, HID] fc2 weight, hidden columns permuted (vda_mlp_permute_w2_f16)
    const float* b2;          // [D]
    const float* gamma;       // [D] LayerScale
    h16* hi;                  // planes of the residual stream, updated in place (hi may be hi_in)
    h16* lo;
    float* part;              // [D / 64, stats_ld, 2] partial row statistics of the updated stream
    int M, stats_ld, n_full;
};

template <int RS>
__global__ void __launch_bounds__(512 / RS) mlp_fused_kernel(const Args p) {
    constexpr int NW = 8 / RS, NT = NW * 64;
    constexpr int PIECES = TILE_BYTES / 1024 / NW;        // DMA pieces (1 KiB) per wave and tile: 6 or 12
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fh = lane >> 4;
    const int bid = blockIdx.x;
    // rows of this workgroup: 128, or 64 in the tail (waves 4..7 then only issue DMA and keep the barriers)
    const bool full = bid < p.n_full;
    const int row0 = full ? bid * 128 : p.n_full * 128 + (bid - p.n_full) * 64;
    const int wrow = row0 + wave * 16 * RS;
    const bool active = (full || wave < NW / 2) && wrow < p.M;       // wave-uniform
    int m[RS];                                                       // this lane's rows (clamped: rows past M are computed and dropped)
#pragma unroll
    for (int r = 0; r < RS; ++r) m[r] = min(wrow + r * 16 + frow, p.M - 1);

    // ---- DMA: tile t = 2 c + (0: W1 chunk c | 1: W2 chunk c) -> ring buffer t % 3. A piece = 8 rows x 128 B; lane -> (row lrow, chunk).
    const int lrow = lane >> 3;
    const int src_chk = ((lane & 7) ^ ((((wave & 1) << 2) + (lrow >> 1)) & 7)) * 8;     // halves (the swizzle on the SOURCE address)
    auto issue = [&](int t) {
        if (t >= 2 * NCHUNK) return;
        char* buf = smem + (t % 3) * TILE_BYTES;
        const int c = t >> 1;
        if ((t & 1) == 0) {
            // W1 chunk: piece (row group g, kt): rows n = 64 c + 8 g + lrow of K tile kt -> buf + kt * 8 KiB + g * 1 KiB; g = wave (+ NW ...)
#pragma unroll
            for (int gi = 0; gi < 8 / NW; ++gi) {
                const int g = wave + NW * gi;
                const h16* src = p.w1 + (size_t)(c * HC + g * 8 + lrow) * D + src_chk;
#pragma unroll
                for (int kt = 0; kt < 6; ++kt) glds16(src + kt * 64, buf + kt * 8192 + g * 1024);
            }
        } else {
            // W2 chunk: piece i of the wave: rows n = 8 (wave + NW i) + lrow, the chunk's 64 (permuted) hidden columns
            const h16* src = p.w2p + (size_t)(wave * 8 + lrow) * HID + c * HC + src_chk;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) glds16(src + (size_t)i * 8 * NW * HID, buf + (wave + NW * i) * 1024);
        }
    };
    // raw barrier (the LDS-DMA stays in flight across it); the empty asm statements and sched_barriers keep LDS accesses on their
    // side of it (gemm8p_kernel.h)
    auto bar = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };
    issue(0);
    issue(1);

    // ---- the wave's A fragments (k-step s: row frow, k = 32 s + 8 fh .. + 8) and the row's LayerNorm statistics
    h16x8 a[RS][KSTEPS];
    float nmean[RS], rstd[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const h16* ap = p.hi_in + (size_t)m[r] * D + fh * 8;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) a[r][s] = *reinterpret_cast<const h16x8*>(ap + s * 32);
        const float2 st = *reinterpret_cast<const float2*>(p.stats + 2 * (size_t)m[r]);
        nmean[r] = -st.x, rstd[r] = st.y;
    }
    // c1 | c2 into LDS (fp32, 12 KiB): 3072 floats by 512 threads
    {
        float* cl = reinterpret_cast<float*>(smem + 3 * TILE_BYTES);
#pragma unroll
        for (int i = 0; i < 2 * HID / NT; ++i) {
            const int idx = tid + i * NT;
            cl[idx] = idx < HID ? p.c1[idx] : p.c2[idx - HID];
        }
    }
    // everything loaded through registers has landed (the two tiles issued above stay in flight: 2 x PIECES younger operations)
    // (hipcc waits vmcnt(0) for the loads above anyway - a VGPR-destination load beside LDS-DMA, cdna_hip_programming.md 4(b) -
    // which costs one exposed L2 round trip per workgroup, once)
    f32x4 acc[RS][NJ2];
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
        for (int j = 0; j < NJ2; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* cl = reinterpret_cast<const float*>(smem + 3 * TILE_BYTES);
    const int fsw = (frow >> 1) & 7;                  // ((row >> 1) & 7) of a 16-row subtile's row frow (16 j + frow: 8 j vanishes mod 8)

    h16x8 hf[RS][HC / 32];                            // the chunk's GELU output as the A fragments of GEMM 2
    for (int c = 0; c < NCHUNK; ++c) {
        // ===== phase 2c: S = A . W1c^T, LayerNorm fold + GELU
        vm_wait_keep<PIECES>();                       // tile 2c landed (mine); tile 2c + 1 stays in flight
        bar();                                        // ... and everyone's; buffer (2c + 2) % 3 = (2c - 1) % 3 is no longer read
        issue(2 * c + 2);
        if (active) {
            const char* wb = smem + ((2 * c) % 3) * TILE_BYTES;
            f32x4 s[RS][NJ1];
#pragma unroll
            for (int r = 0; r < RS; ++r)
#pragma unroll
                for (int j = 0; j < NJ1; ++j) s[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const char* kb = wb + (ks >> 1) * 8192 + ((((ks & 1) * 4 + fh) ^ fsw) << 4);
#pragma unroll
                for (int j = 0; j < NJ1; ++j) {
                    const h16x8 w = *reinterpret_cast<const h16x8*>(kb + (j * 16 + frow) * 128);
#pragma unroll
                    for (int r = 0; r < RS; ++r) s[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, a[r][ks], s[r][j], 0, 0, 0);
                }
            }
            // LayerNorm fold (gemm_epilogue.h, VDA_EPI_LN_GELU_F16: rstd * (acc - mean * c1) + c2) and GELU, 16 values per lane and row
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                float v[NJ1 * 4];
#pragma unroll
                for (int j = 0; j < NJ1; ++j) {
                    const int n = c * HC + j * 16 + fh * 4;
                    const f32x4 k1 = *reinterpret_cast<const f32x4*>(cl + n), k2 = *reinterpret_cast<const f32x4*>(cl + HID + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j * 4 + e] = fmaf(rstd[r], fmaf(nmean[r], k1[e], s[r][j][e]), k2[e]);
                }
                gelu_erf_n(v);
#pragma unroll
                for (int q = 0; q < HC / 32; ++q)
#pragma unroll
                    for (int t = 0; t < 8; ++t) hf[r][q][t] = vda_gemm::to_h16(v[(2 * q + (t >> 2)) * 4 + (t & 3)]);
            }
        }
        // ===== phase 2c + 1: O += H . W2c^T
        if (c + 1 < NCHUNK) vm_wait_keep<PIECES>();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last tile: nothing younger is in flight
        bar();
        issue(2 * c + 3);
        if (active) {
            const char* wb = smem + ((2 * c + 1) % 3) * TILE_BYTES;
#pragma unroll
            for (int q = 0; q < HC / 32; ++q) {
                const char* kb = wb + (((q * 4 + fh) ^ fsw) << 4);
#pragma unroll
                for (int j = 0; j < NJ2; ++j) {
                    const h16x8 w = *reinterpret_cast<const h16x8*>(kb + (j * 16 + frow) * 128);
#pragma unroll
                    for (int r = 0; r < RS; ++r) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, hf[r][q], acc[r][j], 0, 0, 0);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (nothing is in flight: the last two issue() calls were no-ops)
    if (!active) return;

    // ---- epilogue: x' = (hi + lo) - mean + gamma * (O + b2), re-split into the planes, partial statistics per 64 columns.
    // Lane -> row frow, columns 16 j + 4 fh + e: 8-byte accesses, a row's four lane groups complete 32-byte runs; once per workgroup.
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const bool ok = wrow + r * 16 + frow < p.M;
        const size_t rbase = (size_t)m[r] * D + fh * 4;
#pragma unroll
        for (int b = 0; b < D / 64; ++b) {
            float v[16];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = b * 4 + jj, n = j * 16 + fh * 4;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b2 + n), gg = *reinterpret_cast<const f32x4*>(p.gamma + n);
                const h16x4 xh = *reinterpret_cast<const h16x4*>(p.hi + rbase + j * 16), xl = *reinterpret_cast<const h16x4*>(p.lo + rbase + j * 16);
                h16x4 oh, ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = fmaf(gg[e], acc[r][j][e] + bb[e], ((float)xh[e] + (float)xl[e]) + nmean[r]);
                    h16 hh, ll;
                    vda_gemm::split_h16(x, hh, ll);
                    oh[e] = hh;
                    ol[e] = ll;
                    v[jj * 4 + e] = x;
                }
                if (ok) {
                    *reinterpret_cast<h16x4*>(p.hi + rbase + j * 16) = oh;
                    *reinterpret_cast<h16x4*>(p.lo + rbase + j * 16) = ol;
                }
            }
            // (sum, centred sum of squares) of the row's 64 columns: 16 in this lane, the rest in lanes frow + 16, + 32, + 48
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += v[i];
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.f / 64.f);
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float d = v[i] - mean;
                sq = fmaf(d, d, sq);
            }
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (ok && fh == 0) *reinterpret_cast<float2*>(p.part + ((size_t)b * p.stats_ld + (wrow + r * 16 + frow)) * 2) = float2{sum, sq};
        }
    }
}

// w2p[o, 32 s + 8 g + t] = w2[o, 32 s + 16 (t >> 2) + 4 g + (t & 3)]   (s = 32-block of the hidden dimension, g = 0..3, t = 0..7)
__global__ void __launch_bounds__(256) permute_w2_kernel(const h16* __restrict__ w2, h16* __restrict__ w2p, int Dd, int Hh) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)Dd * Hh) return;
    const int o = (int)(i / Hh), k = (int)(i - (long long)o * Hh);
    const int s = k >> 5, g = (k >> 3) & 3, t = k & 7;
    w2p[i] = w2[(size_t)o * Hh + s * 32 + 16 * (t >> 2) + 4 * g + (t & 3)];
}

}  // namespace

extern "C" int vda_mlp_fused_supported(int Dd, int Hh) { return Dd == D && Hh == HID ? 1 : 0; }

extern "C" int vda_mlp_permute_w2_f16(const void* w2, void* w2p, int Dd, int Hh, vda_stream_t stream) {
    VDA_REQUIRE(w2 && w2p && w2 != w2p && Dd > 0 && Hh > 0 && Hh % 32 == 0, "vda_mlp_permute_w2_f16: bad arguments (hidden width must be a multiple of 32)");
    const long long n = (long long)Dd * Hh;
    hipLaunchKernelGGL(permute_w2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const h16*)w2, (h16*)w2p, Dd, Hh);
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_mlp_fused_f16(const void* hi_in, const float* stats, const void* w1, const float* c1, const float* c2, const void* w2p, const float* b2,
                                 const float* gamma, void* hi, void* lo, float* part, int M, int Dd, int Hh, int stats_ld, vda_stream_t stream) {
    VDA_REQUIRE(hi_in && stats && w1 && c1 && c2 && w2p && b2 && gamma && hi && lo && part, "vda_mlp_fused_f16: null argument");
    VDA_REQUIRE(Dd == D && Hh == HID, "vda_mlp_fused_f16: built for D = %d, hidden = %d (got %d, %d): vda_mlp_fused_supported", D, HID, Dd, Hh);
    VDA_REQUIRE(M > 0 && stats_ld >= M, "vda_mlp_fused_f16: M=%d stats_ld=%d", M, stats_ld);
    VDA_REQUIRE((((uintptr_t)hi_in | (uintptr_t)w1 | (uintptr_t)w2p | (uintptr_t)hi | (uintptr_t)lo | (uintptr_t)c1 | (uintptr_t)c2 | (uintptr_t)b2 | (uintptr_t)gamma) & 15) == 0 &&
                    (((uintptr_t)stats | (uintptr_t)part) & 7) == 0,
                "vda_mlp_fused_f16: operands must be 16-byte aligned (statistics 8)");
    static const int rs = getenv("VDA_MLP_RS") ? atoi(getenv("VDA_MLP_RS")) : 1;       // A/B: 1 = 8 waves x 16 rows (default), 2 = 4 waves x 32 rows
    static VdaKernelDeviceState dev_state1, dev_state2;
    const int ncu = rs == 1 ? vda_prepare_kernel(reinterpret_cast<const void*>(&mlp_fused_kernel<1>), SMEM, dev_state1)
                            : vda_prepare_kernel(reinterpret_cast<const void*>(&mlp_fused_kernel<2>), SMEM, dev_state2);
    if (ncu < 0) return 2;
    // full 128-row workgroups for at most one round of the chip; what is left goes out in 64-row workgroups
    Args a;
    a.hi_in = (const h16*)hi_in, a.stats = stats, a.w1 = (const h16*)w1, a.c1 = c1, a.c2 = c2, a.w2p = (const h16*)w2p, a.b2 = b2, a.gamma = gamma;
    a.hi = (h16*)hi, a.lo = (h16*)lo, a.part = part, a.M = M, a.stats_ld = stats_ld;
    a.n_full = M / 128 < ncu ? M / 128 : ncu;
    const int tail = M - a.n_full * 128;
    const int grid = a.n_full + (tail + 63) / 64;
    if (rs == 1) hipLaunchKernelGGL(mlp_fused_kernel<1>, dim3(grid), dim3(512), SMEM, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mlp_fused_kernel<2>, dim3(grid), dim3(256), SMEM, (hipStream_t)stream, a);
    VDA_LAUNCH_CHECK();
    return 0;
}
