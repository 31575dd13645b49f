// fp32-operand GEMM / implicit-GEMM conv for gfx950: the `fp32=True` path of the reference
// (video_depth.py:203-205 disables autocast; run.py:31 --fp32; benchmark/infer/infer.py:58 always passes it).
//
//   out = epilogue( A[M,K] * W[N,K]^T )      fp32 operands, fp32 accumulate, fp32 activations out
//
// v_mfma_f32_32x32x2_f32: exact fp32 products, one rounding per accumulate (bit-for-bit a k-ordered fmaf chain), at
// 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak - 1/16 of the fp16 MFMA rate, so this kernel is MFMA-bound by a wide margin and
// is kept simple: 4 waves, BM x BN x 16 tile, double-buffered 16-byte LDS-DMA, per-lane epilogue.
//
// LDS image per operand tile: [rows][4 chunks of 16 B] (64-byte rows), chunk ^= (row >> 2) & 3 on the DMA source address and
// on the read (any ds_read_b128 lane group then covers the sixteen 16-byte slots of a bank row: conflict-free).
// K order inside a 16-wide K step is permuted identically for both operands (lane half h of fragment jj holds k = 8jj + 4h + e,
// e = 0..3 over four MFMAs), which a contraction does not see: one ds_read_b128 feeds four MFMAs.
// The MFMA is issued with W as the A operand and the activation tile as B, so a lane owns one output ROW m = lane & 31 and,
// in accumulator registers 4g..4g+3, four CONSECUTIVE columns n = 8g + 4h: bias / LayerScale / residual are 16-byte accesses.
#include "vda_common.h"
#include <type_traits>

namespace {

constexpr int BK = 16;                 // floats per K step
constexpr int ROW_BYTES = BK * 4;      // 64 B per tile row

__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// One row m, four consecutive columns n..n+3 (v); g = the gate columns n+16.. (GEGLU only). Every activation is fp32 here:
// the *_F16 epilogue ids keep their meaning (what is fused) and write fp32.
template <int EPI>
__device__ __forceinline__ void store_one_f32(const vda_gemm_args& p, int m, int n, f32x4 v, f32x4 g) {
    if (m >= p.M || n >= p.N) return;
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
    float* out = (float*)p.out;
    if constexpr (EPI == VDA_EPI_BIAS_F16 || EPI == VDA_EPI_BIAS_F32) {
        *reinterpret_cast<f32x4*>(out + (size_t)m * p.ldc + n) = v;
    } else if constexpr (EPI == VDA_EPI_BIAS_GELU_F16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gelu_exact(v[i]);
        *reinterpret_cast<f32x4*>(out + (size_t)m * p.ldc + n) = v;
    } else if constexpr (EPI == VDA_EPI_BIAS_RELU_F16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        *reinterpret_cast<f32x4*>(out + (size_t)m * p.ldc + n) = v;
    } else if constexpr (EPI == VDA_EPI_SCALE_RES_F32 || EPI == VDA_EPI_SCALE_RES_F32_H) {
        if (p.gamma) v *= *reinterpret_cast<const f32x4*>(p.gamma + n);
        const size_t off = (size_t)m * p.ldc + n;
        v += *reinterpret_cast<const f32x4*>((const float*)p.res + off);
        *reinterpret_cast<f32x4*>(out + off) = v;
    } else if constexpr (EPI == VDA_EPI_RES_F16) {
        const size_t off = (size_t)m * p.ldc + n;
        v += *reinterpret_cast<const f32x4*>((const float*)p.res + off);
        if (p.res2) v += *reinterpret_cast<const f32x4*>((const float*)p.res2 + off);
        *reinterpret_cast<f32x4*>(out + off) = v;
    } else if constexpr (EPI == VDA_EPI_GEGLU_F16) {
        if (p.bias) g += *reinterpret_cast<const f32x4*>(p.bias + n + 16);
        const int oc = (n >> 5) * 16 + (n & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= gelu_exact(g[i]);
        *reinterpret_cast<f32x4*>(out + (size_t)m * p.ldc + oc) = v;
    } else if constexpr (EPI == VDA_EPI_PATCH_F32) {
        const int f = m / p.P, q = m - f * p.P;
        v += *reinterpret_cast<const f32x4*>(p.pos + (size_t)(1 + q) * p.N + n);
        *reinterpret_cast<f32x4*>(out + ((size_t)f * (p.P + 1) + 1 + q) * p.ldc + n) = v;
    } else if constexpr (EPI == VDA_EPI_CONVT_F16) {
        const int k = p.tK, Co = p.tCout;
        const int tap = n / Co, co = n - tap * Co;
        const int ky = tap / k, kx = tap - ky * k;
        const int hw = p.tH * p.tW;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / p.tW, x = rem - y * p.tW;
        const size_t orow = ((size_t)b * p.tH * k + (size_t)y * k + ky) * ((size_t)p.tW * k) + (size_t)x * k + kx;
        *reinterpret_cast<f32x4*>(out + orow * p.ldc + co) = v;
    }
}

template <int BM, int BN, int WM, int WN, int AMODE>
__global__ void __launch_bounds__(WM * WN * 64) gemm_f32_kernel(const vda_gemm_args p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile
    constexpr int MI = WTM / 32, NI = WTN / 32;   // 32x32 subtiles per wave
    constexpr int AP = BM / 16, WP = BN / 16;     // 1-KiB DMA pieces (16 rows x 64 B) per tile
    constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES, STAGE = A_BYTES + W_BYTES;
    static_assert(MI >= 1 && NI >= 1, "wave tile must hold a 32x32 subtile");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // block -> tile, XCD-aware: blocks that share an XCD (bid % 8) take a contiguous run of tiles, N fastest
    const int nbn = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int bm = t / nbn, bn = t - bm * nbn;
    const int m0 = bm * BM, n0 = bn * BN;

    // per-lane DMA geometry: LDS position (row lr of the piece, chunk lane & 3) <- source chunk (lane & 3) ^ swz(row)
    const int lr = lane >> 2;
    const int lchk = ((lane & 3) ^ swz(lr)) * 4;               // floats
    const float* A = (const float*)p.A;
    const float* W = (const float*)p.W;

    auto stage = [&](int kt, char* buf) {
        const int k0 = kt * BK;
        int tap = 0, ci0 = 0, ky = 0, kx = 0;
        if constexpr (AMODE == VDA_A_CONV3X3) {
            tap = k0 / p.cCin;
            ci0 = k0 - tap * p.cCin;
            ky = tap / 3;
            kx = tap - ky * 3;
        }
        for (int piece = wave; piece < AP; piece += NW) {
            int m = m0 + piece * 16 + lr;
            const float* src;
            if constexpr (AMODE == VDA_A_DENSE) {
                m = min(m, p.M - 1);
                src = A + (size_t)m * p.lda + k0 + lchk;
            } else {
                const bool row_ok = m < p.M;
                m = min(m, p.M - 1);
                const int hw = p.cHo * p.cWo;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / p.cWo, ox = rem - oy * p.cWo;
                const int iy = oy * p.cStride - 1 + ky, ix = ox * p.cStride - 1 + kx;
                const bool ok = row_ok && (unsigned)iy < (unsigned)p.cH && (unsigned)ix < (unsigned)p.cW;
                src = ok ? A + (((size_t)b * p.cH + iy) * p.cW + ix) * p.cCin + ci0 + lchk : (const float*)p.zero_page + lchk;
            }
            glds16(src, buf + piece * 1024);
        }
        for (int piece = wave; piece < WP; piece += NW) {
            const int n = min(n0 + piece * 16 + lr, p.N - 1);
            glds16(W + (size_t)n * p.K + k0 + lchk, buf + A_BYTES + piece * 1024);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    const bool relu = AMODE == VDA_A_CONV3X3 && (p.relu_in & 1);

    auto compute = [&](const char* buf) {
        const char* At = buf + (wm * WTM + fr) * ROW_BYTES;
        const char* Wt = buf + A_BYTES + (wn * WTN + fr) * ROW_BYTES;
        const int sw = swz(fr);                                  // (row >> 2) & 3 with row = 32 * subtile + fr
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int coff = ((2 * jj + fh) ^ sw) << 4;
            f32x4 af[MI], wf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                af[i] = *reinterpret_cast<const f32x4*>(At + i * 32 * ROW_BYTES + coff);
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) af[i][e] = fmaxf(af[i][e], 0.f);
                }
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) wf[j] = *reinterpret_cast<const f32x4*>(Wt + j * 32 * ROW_BYTES + coff);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][e], af[i][e], acc[i][j], 0, 0, 0);
        }
    };

    const int nt = p.K / BK;
    stage(0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nt; ++kt) {
        if (kt + 1 < nt) stage(kt + 1, smem + (cur ^ 1) * STAGE);
        compute(smem + cur * STAGE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // LDS-DMA landed before the barrier publishes it
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: register e of subtile (i, j) is row n = 8*(e>>2) + 4*fh + (e&3) of W, column m = fr of the activation tile
    auto run = [&](auto epi_tag) {
        constexpr int EPI = decltype(epi_tag)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wm * WTM + i * 32 + fr;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int nb = n0 + wn * WTN + j * 32 + 4 * fh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (EPI == VDA_EPI_GEGLU_F16) {
                        if (g >= 2) continue;                      // g = 2, 3 are the gate rows of g = 0, 1
                    }
                    const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    const int gg = EPI == VDA_EPI_GEGLU_F16 ? g + 2 : g;
                    const f32x4 gt = {acc[i][j][4 * gg], acc[i][j][4 * gg + 1], acc[i][j][4 * gg + 2], acc[i][j][4 * gg + 3]};
                    store_one_f32<EPI>(p, m, nb + 8 * g, v, gt);
                }
            }
        }
    };
    switch (p.epilogue) {
        case VDA_EPI_BIAS_F16: run(std::integral_constant<int, VDA_EPI_BIAS_F16>{}); break;
        case VDA_EPI_BIAS_GELU_F16: run(std::integral_constant<int, VDA_EPI_BIAS_GELU_F16>{}); break;
        case VDA_EPI_BIAS_RELU_F16: run(std::integral_constant<int, VDA_EPI_BIAS_RELU_F16>{}); break;
        case VDA_EPI_SCALE_RES_F32: run(std::integral_constant<int, VDA_EPI_SCALE_RES_F32>{}); break;
        case VDA_EPI_RES_F16: run(std::integral_constant<int, VDA_EPI_RES_F16>{}); break;
        case VDA_EPI_GEGLU_F16: run(std::integral_constant<int, VDA_EPI_GEGLU_F16>{}); break;
        case VDA_EPI_PATCH_F32: run(std::integral_constant<int, VDA_EPI_PATCH_F32>{}); break;
        case VDA_EPI_CONVT_F16: run(std::integral_constant<int, VDA_EPI_CONVT_F16>{}); break;
        case VDA_EPI_BIAS_F32: run(std::integral_constant<int, VDA_EPI_BIAS_F32>{}); break;
        case VDA_EPI_SCALE_RES_F32_H: run(std::integral_constant<int, VDA_EPI_SCALE_RES_F32_H>{}); break;
        default: break;
    }
}

template <int BM, int BN, int WM, int WN, int AMODE>
int launch(const vda_gemm_args& a, hipStream_t s) {
    constexpr int smem = 2 * (BM + BN) * ROW_BYTES;
    static_assert(smem <= 64 * 1024, "fits the default dynamic LDS limit: no per-device attribute to set");
    const long long nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    if (nbm * nbn >= (1ll << 31)) {
        vda_set_error("vda_gemm_f32: grid too large");
        return 1;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, AMODE>), dim3((unsigned)(nbm * nbn)), dim3(WM * WN * 64), smem, s, a);
    VDA_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int vda_gemm_f32(const vda_gemm_args* args, vda_stream_t stream) {
    VDA_REQUIRE(args != nullptr, "vda_gemm_f32: null args");
    const vda_gemm_args& a = *args;
    VDA_REQUIRE(a.A && a.W && a.out, "vda_gemm_f32: null operand");
    VDA_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "vda_gemm_f32: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    VDA_REQUIRE(a.K % BK == 0, "vda_gemm_f32: K=%d must be a multiple of %d (pad at pack time)", a.K, BK);
    VDA_REQUIRE(a.N % 4 == 0 && a.ldc % 4 == 0, "vda_gemm_f32: N=%d and ldc=%d must be multiples of 4", a.N, a.ldc);
    VDA_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0 && ((uintptr_t)a.out & 15) == 0,
                "vda_gemm_f32: operands must be 16-byte aligned");
    VDA_REQUIRE(a.epilogue >= 0 && a.epilogue <= VDA_EPI_SCALE_RES_F32_H, "vda_gemm_f32: bad epilogue %d", a.epilogue);
    if (a.a_mode == VDA_A_DENSE) {
        VDA_REQUIRE(a.relu_in == 0, "vda_gemm_f32: relu_in is only built for the conv A operand");
        VDA_REQUIRE(a.lda >= a.K && a.lda % 4 == 0, "vda_gemm_f32: lda=%d must be >= K and a multiple of 4", a.lda);
    } else if (a.a_mode == VDA_A_CONV3X3) {
        VDA_REQUIRE(a.zero_page != nullptr, "vda_gemm_f32: conv needs zero_page");
        VDA_REQUIRE(a.cCin % BK == 0 && a.K == 9 * a.cCin, "vda_gemm_f32: conv needs Cin%%16==0 and K==9*Cin (Cin=%d K=%d)", a.cCin, a.K);
        VDA_REQUIRE(a.cStride == 1 || a.cStride == 2, "vda_gemm_f32: conv stride %d", a.cStride);
        VDA_REQUIRE(a.cHo == (a.cH + 2 - 3) / a.cStride + 1 && a.cWo == (a.cW + 2 - 3) / a.cStride + 1,
                    "vda_gemm_f32: conv output size mismatch");
        VDA_REQUIRE(a.M == a.cB * a.cHo * a.cWo, "vda_gemm_f32: conv M=%d != B*Ho*Wo", a.M);
    } else {
        VDA_REQUIRE(false, "vda_gemm_f32: bad a_mode %d", a.a_mode);
    }
    switch (a.epilogue) {
        case VDA_EPI_SCALE_RES_F32:
        case VDA_EPI_SCALE_RES_F32_H:
        case VDA_EPI_RES_F16:
            VDA_REQUIRE(a.res != nullptr, "vda_gemm_f32: residual epilogue needs res");
            break;
        case VDA_EPI_GEGLU_F16:
            VDA_REQUIRE(a.N % 32 == 0, "vda_gemm_f32: GEGLU needs N%%32==0");
            break;
        case VDA_EPI_PATCH_F32:
            VDA_REQUIRE(a.pos != nullptr && a.P > 0 && a.M % a.P == 0, "vda_gemm_f32: patch epilogue needs pos and M%%P==0");
            break;
        case VDA_EPI_CONVT_F16:
            VDA_REQUIRE(a.tK > 0 && a.tCout > 0 && a.tCout % 4 == 0 && a.N == a.tK * a.tK * a.tCout && a.M % (a.tH * a.tW) == 0,
                        "vda_gemm_f32: bad ConvTranspose geometry");
            break;
        default:
            break;
    }
    hipStream_t s = (hipStream_t)stream;
    // tile by output width: the depth tail's 3x3 conv has N = 32, the ViT-S head N = 64
    if (a.a_mode == VDA_A_DENSE) {
        if (a.N <= 32) return launch<256, 32, 4, 1, VDA_A_DENSE>(a, s);
        if (a.N <= 64) return launch<128, 64, 2, 2, VDA_A_DENSE>(a, s);
        return launch<128, 128, 2, 2, VDA_A_DENSE>(a, s);
    }
    if (a.N <= 32) return launch<256, 32, 4, 1, VDA_A_CONV3X3>(a, s);
    if (a.N <= 64) return launch<128, 64, 2, 2, VDA_A_CONV3X3>(a, s);
    return launch<128, 128, 2, 2, VDA_A_CONV3X3>(a, s);
}
