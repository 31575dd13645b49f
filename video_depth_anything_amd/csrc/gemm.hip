// fp16 MFMA GEMM / implicit-GEMM conv for gfx950 (MI355X).
//
//   out = epilogue( A[M,K] * W[N,K]^T )          fp16 operands, fp32 accumulate
//
// Both operands are K-contiguous, which is exactly the v_mfma_f32_16x16x32_f16 fragment
// shape (8 consecutive k per lane), so tiles go HBM -> LDS with 16-byte
// global_load_lds (no VGPR round trip) and LDS -> VGPR with ds_read_b128.
//
// Tile: BM x BN x 64, 4 waves (2 x 2), each wave (BM/2) x (BN/2) as 16x16 MFMA subtiles.
// LDS image per operand tile: [rows][8 chunks of 16 B] with chunk ^= (row & 7). The DMA
// writes LDS linearly (wave-uniform base + lane*16), so the swizzle is applied to the
// per-lane SOURCE address and again on the ds_read address (same involution both sides).
// The MFMA is issued with W as the "A" operand and the activation tile as "B", so a lane's
// 4 accumulator registers are 4 CONSECUTIVE output columns of one row: bias / LayerScale /
// residual are read and the result stored as one 8- or 16-byte access per subtile.
//
// A-operand generators: dense rows, or the 3x3/pad-1 window of an NHWC tensor
// (K ordered (ky,kx,ci); a 64-wide K step never straddles a tap because Cin % 64 == 0).
// Rows past M / N are clamped on load and dropped on store.
#include "gemm_epilogue.h"

namespace {
using vda_gemm::store_one;

constexpr int BK = 64;                 // halves per K step
constexpr int ROW_BYTES = BK * 2;      // 128 B per tile row
constexpr int NWAVES = 4;
constexpr int NTHREADS = NWAVES * 64;

template <int BM, int BN, int AMODE>
__global__ void __launch_bounds__(NTHREADS) gemm_kernel(const vda_gemm_args p) {
    constexpr int WTM = BM / 2, WTN = BN / 2;     // wave tile
    constexpr int MI = WTM / 16, NI = WTN / 16;   // 16x16 subtiles per wave
    constexpr int AJ = BM / 8 / NWAVES;           // 1-KiB DMA pieces per wave, A tile
    constexpr int WJ = BN / 8 / NWAVES;
    constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES, STAGE = A_BYTES + W_BYTES;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- block -> tile, XCD-aware: blocks that share an XCD (bid % 8) get a contiguous run of
    // tiles with the N index fastest, so an A row panel stays in that XCD's L2 across its N tiles.
    const int nbn = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int bm = t / nbn, bn = t - bm * nbn;
    const int m0 = bm * BM, n0 = bn * BN;

    // ---- per-lane DMA sources
    const int lrow = lane >> 3;                   // row inside an 8-row piece
    const int lchk = ((lane & 7) ^ lrow) * 8;     // swizzled source chunk, in halves
    const h16* a_src[AJ];
    int a_pix[AJ], a_iy0[AJ], a_ix0[AJ];
    const h16* w_src[WJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        int m = m0 + (wave + NWAVES * j) * 8 + lrow;
        if constexpr (AMODE == VDA_A_DENSE) {
            m = min(m, p.M - 1);
            a_src[j] = (const h16*)p.A + (size_t)m * p.lda + lchk;
        } else {
            const bool ok = m < p.M;
            m = min(m, p.M - 1);
            const int hw = p.cHo * p.cWo;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.cWo, ox = rem - oy * p.cWo;
            a_pix[j] = b * p.cH * p.cW;
            a_iy0[j] = ok ? oy * p.cStride - 1 : -100000;
            a_ix0[j] = ox * p.cStride - 1;
            a_src[j] = (const h16*)p.A + lchk;
        }
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int n = min(n0 + (wave + NWAVES * j) * 8 + lrow, p.N - 1);
        w_src[j] = (const h16*)p.W + (size_t)n * p.K + lchk;
    }

    auto stage = [&](int kt, char* buf) {
        const int k0 = kt * BK;
        if constexpr (AMODE == VDA_A_DENSE) {
#pragma unroll
            for (int j = 0; j < AJ; ++j) glds16(a_src[j] + k0, buf + (wave + NWAVES * j) * 1024);
        } else {
            const int tap = k0 / p.cCin, ci0 = k0 - tap * p.cCin;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const int iy = a_iy0[j] + ky, ix = a_ix0[j] + kx;
                const bool ok = (unsigned)iy < (unsigned)p.cH && (unsigned)ix < (unsigned)p.cW;
                const h16* src = ok ? a_src[j] + ((size_t)(a_pix[j] + iy * p.cW + ix) * p.cCin + ci0)
                                    : (const h16*)p.zero_page + lchk;
                glds16(src, buf + (wave + NWAVES * j) * 1024);
            }
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) glds16(w_src[j] + k0, buf + A_BYTES + (wave + NWAVES * j) * 1024);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row = lane&15 (so row&7 == lane&7), k-chunk = ks*4 + (lane>>4)
    const int frow = lane & 15, fchk = lane >> 4, fsw = lane & 7;
    const h16 relu_floor = p.relu_in ? (h16)0.f : (h16)(-65504.f);
    h16x8 relu_thr;
#pragma unroll
    for (int e = 0; e < 8; ++e) relu_thr[e] = relu_floor;

    auto compute = [&](const char* buf) {
        const char* At = buf + (wm * WTM + frow) * ROW_BYTES;
        const char* Wt = buf + A_BYTES + (wn * WTN + frow) * ROW_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + fchk) ^ fsw) << 4;
            h16x8 af[MI], wf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const h16x8*>(At + i * 16 * ROW_BYTES + coff);
#pragma unroll
            for (int j = 0; j < NI; ++j) wf[j] = *reinterpret_cast<const h16x8*>(Wt + j * 16 * ROW_BYTES + coff);
            if constexpr (AMODE == VDA_A_CONV3X3) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = __builtin_elementwise_max(af[i], relu_thr);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane holds, per subtile, 4 consecutive columns of one row
    const int em = m0 + wm * WTM + (lane & 15);
    const int en = n0 + wn * WTN + (lane >> 4) * 4;
    auto run = [&](auto epi_tag) {
        constexpr int EPI = decltype(epi_tag)::value;
        if constexpr (EPI == VDA_EPI_GEGLU_F16) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; j += 2) store_one<EPI>(p, em + i * 16, en + j * 16, acc[i][j], acc[i][j + 1]);
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) store_one<EPI>(p, em + i * 16, en + j * 16, acc[i][j], acc[i][j]);
        }
    };
    vda_gemm::dispatch_epilogue(p.epilogue, run);
}

template <int BM, int BN, int AMODE>
int launch(const vda_gemm_args& a, hipStream_t s) {
    constexpr int smem = 2 * (BM + BN) * ROW_BYTES;
    static VdaKernelDeviceState dev_state;
    if (vda_prepare_kernel(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, AMODE>), smem, dev_state) < 0) return 2;
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_kernel<BM, BN, AMODE>), dim3(nbm * nbn), dim3(NTHREADS), smem, s, a);
    VDA_LAUNCH_CHECK();
    return 0;
}

// Partial row statistics of the split stream for the kernels whose epilogue is not row-layout (small problems only):
// part[j, m, :] = (sum, centred sum of squares) of hi + lo over columns 64j..64j+63, one lane per (row, 64-column block).
__global__ void __launch_bounds__(256) split_partials_kernel(const h16* __restrict__ hi, const h16* __restrict__ lo, float* __restrict__ part,
                                                             int M, int N, int ldc, int ld) {
    const int np = N >> 6;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * np) return;
    const int m = (int)(i / np), j = (int)(i - (long long)m * np);
    const h16* a = hi + (size_t)m * ldc + j * 64;
    const h16* b = lo + (size_t)m * ldc + j * 64;
    float sum = 0.f;
    for (int e = 0; e < 64; ++e) sum += (float)a[e] + (float)b[e];
    const float mean = sum * (1.f / 64.f);
    float sq = 0.f;
    for (int e = 0; e < 64; ++e) {
        const float d = ((float)a[e] + (float)b[e]) - mean;
        sq = fmaf(d, d, sq);
    }
    part[2 * ((size_t)j * ld + m)] = sum;             // ld = rows per column block of the array (stats_ld: M of the WHOLE GEMM for a row range)
    part[2 * ((size_t)j * ld + m) + 1] = sq;
}

}  // namespace

// gemm256_*.hip: return -1 when the (A mode, epilogue) pair is not instantiated for the large tile
int vda_gemm256_dense_bn256(const vda_gemm_args& a, hipStream_t s);
int vda_gemm256_dense_bn128(const vda_gemm_args& a, hipStream_t s);
int vda_gemm256_conv_bn256(const vda_gemm_args& a, hipStream_t s);
int vda_gemm256_conv_bn128(const vda_gemm_args& a, hipStream_t s);
// gemm256s_*.hip: the same tiles on v_mfma_f32_16x16x32_f16
int vda_gemm256s_dense_bn256(const vda_gemm_args& a, hipStream_t s);
int vda_gemm256s_dense_bn128(const vda_gemm_args& a, hipStream_t s);
int vda_gemm256s_conv_bn256(const vda_gemm_args& a, hipStream_t s);
int vda_gemm256s_conv_bn128(const vda_gemm_args& a, hipStream_t s);
int vda_gemm256s_dense_bn128_bm192(const vda_gemm_args& a, hipStream_t s);      // 192 x 128 tiles, six waves; -1 = epilogue not built
int vda_gemm256s_dense_bn128_bm192_x2(const vda_gemm_args& a, hipStream_t s);   // the same tile, TWO workgroups per CU (variant 11); -1 = not built
int vda_gemm256s_dense_bn384_bm192(const vda_gemm_args& a, hipStream_t s);      // 192 x 384 tiles, twelve waves; -1 = epilogue not built

// gemm8p_*.hip: 256 x 256 tile, 8-phase two-group schedule
int vda_gemm8p_dense_bn256(const vda_gemm_args& a, hipStream_t s);
int vda_gemm8p_conv_bn256(const vda_gemm_args& a, hipStream_t s);
int vda_gemm8p_dense_bn256_sched(const vda_gemm_args& a, hipStream_t s, int sched);
int vda_gemm8p_dense_bn128(const vda_gemm_args& a, hipStream_t s);
int vda_gemm8p_dense_bn256_bm192(const vda_gemm_args& a, hipStream_t s);         // 192 x 256 tiles; -1 = epilogue not built
int vda_gemm8p_conv_bn128(const vda_gemm_args& a, hipStream_t s);

// conv_lds.hip: patch-in-LDS direct 3x3 convolution for narrow outputs; -1 when the problem is not one it covers
int vda_conv3x3_lds(const vda_gemm_args& a, hipStream_t s);

static int vda_gemm256s_launch(const vda_gemm_args& a, int bn, hipStream_t s) {
    if (a.a_mode == VDA_A_DENSE) return bn == 256 ? vda_gemm256s_dense_bn256(a, s) : vda_gemm256s_dense_bn128(a, s);
    return bn == 256 ? vda_gemm256s_conv_bn256(a, s) : vda_gemm256s_conv_bn128(a, s);
}

static int vda_gemm256_launch(const vda_gemm_args& a, int bn, hipStream_t s) {
    if (a.a_mode == VDA_A_DENSE) return bn == 256 ? vda_gemm256_dense_bn256(a, s) : vda_gemm256_dense_bn128(a, s);
    return bn == 256 ? vda_gemm256_conv_bn256(a, s) : vda_gemm256_conv_bn128(a, s);
}

// the 128-row kernel (any size, any epilogue)
static int launch_small(const vda_gemm_args& a, hipStream_t s) {
    const bool narrow = a.N <= 64;
    if (a.a_mode == VDA_A_DENSE) {
        const int rc = narrow ? launch<128, 64, VDA_A_DENSE>(a, s) : launch<128, 128, VDA_A_DENSE>(a, s);
        if (rc == 0 && a.epilogue == VDA_EPI_SCALE_RES_SPLIT) {
            // this kernel's epilogue does not own whole row segments: the partial statistics come from a pass over the planes
            const long long items = (long long)a.M * (a.N >> 6);
            hipLaunchKernelGGL(split_partials_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, (const h16*)a.out, (const h16*)a.out2, a.stats, a.M,
                               a.N, a.ldc, a.stats_ld ? a.stats_ld : a.M);
            VDA_LAUNCH_CHECK();
        }
        return rc;
    }
    return narrow ? launch<128, 64, VDA_A_CONV3X3>(a, s) : launch<128, 128, VDA_A_CONV3X3>(a, s);
}

static int g_gemm_variant = -1;   // tuning / A-B hook, see the dispatch in vda_gemm_f16

static thread_local const char* g_last_kernel = "";

extern "C" const char* vda_gemm_last_kernel(void) { return g_last_kernel; }

extern "C" int vda_gemm_set_variant(int v) {
    g_gemm_variant = v;
    return 0;
}

static int g_gemm_debug = 0;      // diagnostic switches of the 8-phase kernel (bit 0 clock stamps into pos, bit 1 L2-blocked tile order)

extern "C" int vda_gemm_set_debug(int flags) {
    g_gemm_debug = flags & 0xff;
    return 0;
}

static int device_cus() {
    static thread_local int ncu = 0;
    if (ncu == 0) {
        int dev = 0, cu = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
        ncu = cu < 8 ? 8 : (cu & ~7);
    }
    const int cap = g_vda_max_wgs;             // vda_set_max_wgs: the planner's rounds are rounds of the CAPPED grid
    return (cap >= 8 && cap < ncu) ? (cap & ~7) : ncu;
}

static bool bm192_epilogue(int epilogue) {
    return epilogue == VDA_EPI_BIAS_F16 || epilogue == VDA_EPI_SCALE_RES_F32 || epilogue == VDA_EPI_SCALE_RES_SPLIT || epilogue == VDA_EPI_LN_BIAS_F16;
}

// Rows [0, M1) on 256 x 256 tiles in whole rounds of the chip, rows [M1, M) on 192 x 256 tiles. Cost model, calibrated on the MI355X
// (tools/gemm_ab.py -1,5,1029, round 3): a round of 256-row tiles takes t = 1.6 us per K tile + 5 us; a round of 192-row tiles 0.84 t
// (not 0.75: the load sections of the two-group schedule shrink less than its MFMA sections); the second launch costs ~10 us
// (boundary, pipeline refill, no cross-launch prefetch). What that leaves: K = 4096 (fc2: 3 rounds -> 2 + 0.84: 315 -> 301 us with the
// plain epilogue, 357 -> 348 with the split-residual one) pays; K = 1024 (qkv 9 -> 6 + 3 x 0.84: 260 -> 272 us; proj) does not - a
// partial last round already runs faster than a full one, which a round count cannot see: K < 2048 never splits. M1 == M unless
// the model predicts at least 1.5 % (VDA_GEMM_SPLIT=0: never; 2: whenever the round count says so, any K - the A/B switch).
extern "C" int vda_gemm_plan_split(int M, int N, int K, int epilogue, int a_mode) {
    static const int allow = getenv("VDA_GEMM_SPLIT") ? atoi(getenv("VDA_GEMM_SPLIT")) : 1;
    static const int r192 = getenv("VDA_GEMM_R192") ? atoi(getenv("VDA_GEMM_R192")) : 84;
    if (!allow || a_mode != VDA_A_DENSE || !bm192_epilogue(epilogue) || M < 4096 || N < 256 || K < (allow == 2 ? 256 : 2048)) return M;
    const long long ncu = device_cus(), nbn = (N + 255) / 256;
    const long long rt = (M + 255) / 256;
    auto rounds = [&](long long tiles) { return (tiles + ncu - 1) / ncu; };
    const double t = 1.6 * (K / 64) + 5.0, launch = allow == 2 ? 0.0 : 10.0;         // us (VDA_GEMM_SPLIT=2: A/B, split whenever rounds say so)
    const double single = (double)rounds(rt * nbn) * t;
    double best = single * 0.985;
    int best_m1 = M;
    for (long long r = 8; r * 256 < M; ++r) {               // (both parts stay on the 8-phase kernels: >= 2048 rows each, so a row's
        const long long m2 = M - r * 256;                   // arithmetic - K order, bias in the accumulators' start - is the same in either)
        if (m2 < 2048) break;
        const double cost = (double)rounds(r * nbn) * t + 0.01 * r192 * (double)rounds(((m2 + 191) / 192) * nbn) * t + launch;
        if (cost < best) {
            best = cost;
            best_m1 = (int)(r * 256);
        }
    }
    return best_m1;
}

// args of the row range [r0, r0 + rows) of a dense GEMM (the layout rules of vda_gemm_plan_split's comment in vda.h)
static vda_gemm_args row_range(const vda_gemm_args& a0, int r0, int rows) {
    vda_gemm_args a = a0;
    if (a.lda == 0) a.lda = a.K;       // vda.h: the public row-range helper reads 0 as "dense, K" (vda_gemm_f16 itself reads lda == 0 as a
    if (a.ldc == 0) a.ldc = a.N;       // broadcast row and therefore never row-splits such a GEMM on its own: see the guard at its split)
    vda_gemm_args p = a;
    auto adv = [&](const void* q, size_t bytes_per_row) -> const void* { return q ? (const char*)q + (size_t)r0 * bytes_per_row : nullptr; };
    const bool f32_out = a.epilogue == VDA_EPI_SCALE_RES_F32;
    const size_t ob = f32_out ? 4 : 2;
    p.A = adv(a.A, (size_t)a.lda * 2);
    p.out = const_cast<void*>(adv(a.out, (size_t)a.ldc * ob));
    if (a.epilogue == VDA_EPI_SCALE_RES_F32) p.res = adv(a.res, (size_t)a.ldc * 4);
    if (a.epilogue == VDA_EPI_SCALE_RES_SPLIT) {
        p.res = adv(a.res, (size_t)a.ldc * 2);
        p.res2 = adv(a.res2, (size_t)a.ldc * 2);
        p.out2 = const_cast<void*>(adv(a.out2, (size_t)a.ldc * 2));
        p.stats = (float*)const_cast<void*>(adv(a.stats, 8));
        p.stats_ld = a.stats_ld ? a.stats_ld : a.M;
        if (a.pos != nullptr && (const void*)a.pos != a.zero_page) p.pos = (const float*)adv(a.pos, 8);
    }
    if (a.epilogue == VDA_EPI_LN_BIAS_F16) p.stats = (float*)const_cast<void*>(adv(a.stats, 8));
    p.M = rows;
    return p;
}

extern "C" int vda_gemm_row_range(const vda_gemm_args* args, int r0, int rows, vda_gemm_args* out) {
    VDA_REQUIRE(args && out && r0 >= 0 && rows > 0 && r0 + rows <= args->M, "vda_gemm_row_range: bad range");
    VDA_REQUIRE(args->a_mode == VDA_A_DENSE && bm192_epilogue(args->epilogue), "vda_gemm_row_range: dense A and a row-splittable epilogue only");
    *out = row_range(*args, r0, rows);
    return 0;
}

static int vda_gemm_f16_impl(const vda_gemm_args* args, vda_stream_t stream, bool may_split);

extern "C" int vda_gemm_f16(const vda_gemm_args* args, vda_stream_t stream) { return vda_gemm_f16_impl(args, stream, true); }

static int vda_gemm_f16_impl(const vda_gemm_args* args, vda_stream_t stream, bool may_split) {
    VDA_REQUIRE(args != nullptr, "vda_gemm_f16: null args");
    vda_gemm_args a = *args;
    VDA_REQUIRE(a.A && a.W && a.out, "vda_gemm_f16: null operand");
    VDA_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "vda_gemm_f16: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    VDA_REQUIRE(a.K % BK == 0, "vda_gemm_f16: K=%d must be a multiple of %d (pad at pack time)", a.K, BK);
    VDA_REQUIRE(a.N % 4 == 0 && a.ldc % 4 == 0, "vda_gemm_f16: N=%d and ldc=%d must be multiples of 4", a.N, a.ldc);
    VDA_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0 && ((uintptr_t)a.out & 15) == 0,
                "vda_gemm_f16: operands must be 16-byte aligned");
    VDA_REQUIRE(a.epilogue >= 0 && a.epilogue <= VDA_EPI_LN_GELU_F16, "vda_gemm_f16: bad epilogue %d", a.epilogue);
    if (a.a_mode == VDA_A_DENSE) {
        VDA_REQUIRE(a.relu_in == 0, "vda_gemm_f16: relu_in is only built for the conv A operand");
        VDA_REQUIRE((a.lda >= a.K || a.lda == 0) && a.lda % 8 == 0, "vda_gemm_f16: lda=%d must be >= K (or 0 = broadcast row) and a multiple of 8", a.lda);
    } else if (a.a_mode == VDA_A_CONV3X3) {
        VDA_REQUIRE(a.zero_page != nullptr, "vda_gemm_f16: conv needs zero_page");
        VDA_REQUIRE(a.cCin % BK == 0 && a.K == 9 * a.cCin, "vda_gemm_f16: conv needs Cin%%64==0 and K==9*Cin (Cin=%d K=%d)", a.cCin, a.K);
        VDA_REQUIRE(a.cStride == 1 || a.cStride == 2, "vda_gemm_f16: conv stride %d", a.cStride);
        VDA_REQUIRE(a.cHo == (a.cH + 2 - 3) / a.cStride + 1 && a.cWo == (a.cW + 2 - 3) / a.cStride + 1,
                    "vda_gemm_f16: conv output size mismatch");
        VDA_REQUIRE(a.M == a.cB * a.cHo * a.cWo, "vda_gemm_f16: conv M=%d != B*Ho*Wo", a.M);
        VDA_REQUIRE((long long)a.cB * a.cH * a.cW * a.cCin < (1ll << 31), "vda_gemm_f16: conv input too large for 32-bit pixel index");
    } else {
        VDA_REQUIRE(false, "vda_gemm_f16: bad a_mode %d", a.a_mode);
    }
    switch (a.epilogue) {
        case VDA_EPI_SCALE_RES_F32:
        case VDA_EPI_SCALE_RES_F32_H:
        case VDA_EPI_RES_F16:
            VDA_REQUIRE(a.res != nullptr, "vda_gemm_f16: residual epilogue needs res");
            break;
        case VDA_EPI_GEGLU_F16:
            VDA_REQUIRE(a.N % 32 == 0, "vda_gemm_f16: GEGLU needs N%%32==0");
            break;
        case VDA_EPI_SCALE_RES_SPLIT:
            VDA_REQUIRE(a.res != nullptr && a.res2 != nullptr && a.out2 != nullptr && a.stats != nullptr,
                        "vda_gemm_f16: the split-residual epilogue needs res, res2 (hi / lo planes), out2 and stats");
            VDA_REQUIRE(a.N % 64 == 0 && a.ldc % 8 == 0 && a.a_mode == VDA_A_DENSE, "vda_gemm_f16: the split-residual epilogue needs a dense A operand, N%%64==0 and ldc%%8==0");
            VDA_REQUIRE(((uintptr_t)a.res & 15) == 0 && ((uintptr_t)a.res2 & 15) == 0 && ((uintptr_t)a.out2 & 15) == 0 && ((uintptr_t)a.stats & 7) == 0,
                        "vda_gemm_f16: split-residual planes must be 16-byte aligned");
            // re-centring rows (pos = [M, 2] (mean, rstd), optional): the kernels load pos[m * P] unconditionally
            // (pos == zero_page is this function's own "no re-centring" form coming back through the row split below: stride 0 stays)
            if (a.pos != nullptr && (const void*)a.pos != a.zero_page) {
                VDA_REQUIRE(((uintptr_t)a.pos & 7) == 0, "vda_gemm_f16: pos (re-centring statistics) must be 8-byte aligned");
                a.P = 2;
            } else {
                VDA_REQUIRE(a.zero_page != nullptr, "vda_gemm_f16: the split-residual epilogue without pos needs zero_page");
                a.pos = (const float*)a.zero_page;
                a.P = 0;
            }
            break;
        case VDA_EPI_LN_BIAS_F16:
        case VDA_EPI_LN_GELU_F16:
            VDA_REQUIRE(a.stats != nullptr && a.gamma != nullptr && a.bias != nullptr && a.a_mode == VDA_A_DENSE,
                        "vda_gemm_f16: a LayerNorm-folded epilogue needs a dense A operand, stats (mean, rstd rows), gamma (= c1) and bias (= c2)");
            VDA_REQUIRE(((uintptr_t)a.stats & 7) == 0, "vda_gemm_f16: stats must be 8-byte aligned");
            break;
        case VDA_EPI_PATCH_F32:
            VDA_REQUIRE(a.pos != nullptr && a.P > 0 && a.M % a.P == 0, "vda_gemm_f16: patch epilogue needs pos and M%%P==0");
            break;
        case VDA_EPI_CONVT_F16:
            VDA_REQUIRE(a.tK > 0 && a.tCout > 0 && a.tCout % 4 == 0 && a.N == a.tK * a.tK * a.tCout && a.M % (a.tH * a.tW) == 0,
                        "vda_gemm_f16: bad ConvTranspose geometry");
            break;
        default:
            break;
    }
    hipStream_t s = (hipStream_t)stream;
    // Narrow-output 3x3 convs (Cout <= 64: the ViT-S head) run as a patch-in-LDS direct convolution instead of an implicit GEMM
    // (variant 7 forces it, any other explicit variant or VDA_CONV_LDS=0 keeps the GEMM: A/B and cross-checks).
    static const int conv_lds = getenv("VDA_CONV_LDS") ? atoi(getenv("VDA_CONV_LDS")) : 1;
    if (a.a_mode == VDA_A_CONV3X3 && ((g_gemm_variant < 0 && conv_lds) || g_gemm_variant == 7)) {
        const int rc = vda_conv3x3_lds(a, s);
        if (rc >= 0) {
            g_last_kernel = a.N <= 32 ? "conv3x3_lds_kernel<1>" : "conv3x3_lds_kernel<2>";
            return rc;
        }
    }
    const bool fits32 = (a.a_mode == VDA_A_DENSE ? (long long)a.M * a.lda : 0ll) + a.K < (1ll << 31) && (long long)a.N * a.K < (1ll << 31);
    VDA_REQUIRE(fits32 || g_gemm_variant == 0 || g_gemm_variant < 0, "vda_gemm_f16: operand too large for the 256-row kernel's 32-bit offsets");
    if (!fits32) return launch_small(a, s);
    // variants: -1 auto; 0 = 128-row tiles; 1 / 2 = 256x256 / 256x128 on 32x32x16 MFMA; 3 / 4 = the same on 16x16x32 MFMA;
    // 5 = 256x256 8-phase two-group schedule (16x16x32 MFMA)
    int big = 0, small_mfma = 1;
    bool eight = g_gemm_variant >= 5 && (g_gemm_variant & 15) == 5;      // upper bits: A/B switches of the 8-phase kernel
    if (eight) big = 256;
    if (g_gemm_variant == 1 || g_gemm_variant == 3) big = 256;
    if (g_gemm_variant == 2 || g_gemm_variant == 4 || g_gemm_variant == 8 || g_gemm_variant == 10 || g_gemm_variant == 11) big = 128;
    if (g_gemm_variant == 1 || g_gemm_variant == 2) small_mfma = 0;
    if (g_gemm_variant == 9) {               // 9 = 256x128 8-phase two-group schedule
        eight = true;
        big = 128;
    }
    static const int big_min_n = getenv("VDA_GEMM_BIG_MIN_N") ? atoi(getenv("VDA_GEMM_BIG_MIN_N")) : 192;      // A/B hook
    if (g_gemm_variant < 0 && a.N >= big_min_n && a.M >= 2048) {
        // large-tile kernel; BN picked for the smaller padded width
        const int pad256 = (a.N + 255) / 256 * 256, pad128 = (a.N + 127) / 128 * 128;
        big = pad128 < pad256 ? 128 : 256;
        // 256 x 256: the 8-phase two-group schedule (tools/gemm_ab.py, in-process A/B: -4..-15 % on every encoder shape and
        // epilogue). The conv A operand too since it is gathered by bounds-checked buffer loads (one scalar tap offset per K tile,
        // no pointer select): +7..10 % over the one-barrier kernel on the head's 256-channel convs (it had been 8-16 % SLOWER with
        // per-stage address math in its load sections).
        // 256 x 128 on the same schedule (variant 9 / VDA_GEMM_8P128=1) is built and tested but NOT the default: in-process A/B on
        // ViT-S's N = 384 / 1152 / 1536 GEMMs (K = 384: six K tiles) and on output_conv1 it is -4..+2 % against the one-barrier kernel
        static const int eight128 = getenv("VDA_GEMM_8P128") ? atoi(getenv("VDA_GEMM_8P128")) : 0;
        if (big == 256 || eight128) eight = true;
    }
    if (big && (a.N % 8 != 0 || a.ldc % 8 != 0)) {
        VDA_REQUIRE(g_gemm_variant < 0, "vda_gemm_f16: the 256-row kernel needs N and ldc to be multiples of 8");
        big = 0;                            // its epilogue owns 8-column (16-byte) row segments
    }
    if (big) {
        vda_gemm_args a8 = a;
        static const int stagger = getenv("VDA_GEMM_STAGGER") ? atoi(getenv("VDA_GEMM_STAGGER")) : 0;      // 1 / 2 = start stagger (off: see gemm8p_kernel.h)
        if (!stagger) a8.relu_in |= 16 << 8;
        if (stagger == 2) a8.relu_in |= 32 << 8;             // A/B: panel-aligned phases
        if (eight && g_gemm_variant > 0) a8.relu_in = (a.relu_in & 0xff) | (((g_gemm_variant >> 4) & 0xff) << 8);   // A/B switches
        a8.relu_in |= g_gemm_debug << 16;                      // vda_gemm_set_debug (0 unless a diagnostic tool set it)
        // Non-temporal output stores (8-phase kernel, fp16 row stores) for outputs that no cache will hand to the next kernel:
        // VDA_GEMM_NT_MB (default 192; 0 = never) megabytes and up. ViT-L: qkv, hid, the GEGLU hidden, the 148^2 conv maps; ViT-S: none.
        static const long long nt_mb = getenv("VDA_GEMM_NT_MB") ? atoll(getenv("VDA_GEMM_NT_MB")) : 192;
        if (nt_mb > 0 && (long long)a.M * a.N * 2 >= nt_mb * 1000000ll) a8.relu_in |= 1 << 24;
        const int sched8 = (eight && g_gemm_variant > 0) ? ((g_gemm_variant >> 5) & 3) : 0;   // A/B: variant 5 + 32 * sched
        // 192-row tiles when they quantise better on this device: rounds of 256-row tiles against 3/4-size rounds of 192-row tiles
        // (ViT-S proj / fc2: 3 against 2.25; variant 8 forces them). Only the one-barrier 256 x 128 family has the shape.
        bool tall192 = g_gemm_variant == 8 && a.a_mode == VDA_A_DENSE;
        const int ncu = device_cus();
        if (g_gemm_variant < 0 && !eight && big == 128 && a.a_mode == VDA_A_DENSE) {
            const long long nbn = (a.N + 127) / 128;
            const long long r256 = (((a.M + 255) / 256) * nbn + ncu - 1) / ncu, r192 = (((a.M + 191) / 192) * nbn + ncu - 1) / ncu;
            static const int allow192 = getenv("VDA_GEMM_BM192") ? atoi(getenv("VDA_GEMM_BM192")) : 1;
            tall192 = allow192 && r192 * 3 * 100 < r256 * 4 * 85;                 // at least 15 % fewer tile-time units (a 192-row tile costs ~0.8, not 0.75, of a 256-row one)
        }
        // Few large tiles leave most of the chip idle for a whole K loop: when the 256-row tiling fills at most half of the CUs the
        // 128-row kernel (four times the tiles, two workgroups per CU) is faster - the head's 19x19 maps: rn4 (46 tiles, K = 9216)
        // 223 -> 133 us, the refinenet4 convs 62 -> 39 us (tools/gemm_ab.py, AB_SHAPES=small). VDA_GEMM_SMALL_GRID=0 switches it off.
        static const int small_grid = getenv("VDA_GEMM_SMALL_GRID") ? atoi(getenv("VDA_GEMM_SMALL_GRID")) : 1;
        if (g_gemm_variant < 0 && small_grid && ((long long)(a.M + 255) / 256) * ((a.N + big - 1) / big) * 2 <= ncu) {
            g_last_kernel = a.a_mode == VDA_A_DENSE ? (a.N <= 64 ? "gemm_kernel<128, 64, 0>" : "gemm_kernel<128, 128, 0>")
                                                    : (a.N <= 64 ? "gemm_kernel<128, 64, 1>" : "gemm_kernel<128, 128, 1>");
            return launch_small(a, s);
        }
        // N a multiple of 384 (ViT-S's embedding width: proj / fc2 N = 384, qkv N = 1152): 192 x 384 tiles on twelve waves, one tile
        // per row panel and column third - A and the residual rows are read once, 229 row panels are one round of the chip.
        // Built, tested (variant 10) and NOT the default (VDA_GEMM_BN384=1 enables it): in-process A/B (tools/gemm_ab.py
        // AB_SHAPES=vits -1,3,8,10, round 3) it wins with a plain epilogue (fc2's shape 70.7 -> 52.4 us) and not with the ones the
        // model uses - split residual: fc2 80.0 -> 79.6, proj 38.2 -> 43.2 us; qkv + LayerNorm 61.5 -> 64.1 - and the ViT-S forward is
        // 8.86 -> 9.15 ms with it: a single round puts every CU's residual epilogue (270 MB for fc2) on HBM at the same moment with no
        // K loop anywhere to hide behind.
        // [r4] The split-residual instantiation had been reloading two K-loop registers from scratch in every K tile (168 VGPRs; a
        // scratch reload is a vector-memory load whose in-order wait drains the tile's LDS-DMA) and spilling 76 registers around its
        // epilogue. With the epilogue's lane-derived addresses kept out of the K loop's live set and half-size row groups (gemm256s_kernel.h)
        // fc2's shape is 86.9 -> 69.5 us with the split residual and 82.1 -> 63.0 with the fp32 one, proj's is a tie (41.6 / 41.4),
        // and with the LayerNorm-folded epilogues it now wins on qkv (N = 1152: 60.4 -> 55.5 us) and fc1 (N = 1536, against the 8-phase
        // 256 x 256 tile: 107.0 -> 91.1 us) (profiles/r04/vits_bn384_ab.txt). DEFAULT (VDA_GEMM_BN384 unset or 2): N = 384 with
        // K >= 1024 (ViT-S's fc2, any built epilogue) and N <= 1536 with a LayerNorm-folded epilogue (ViT-S's qkv and fc1);
        // 1 = every N % 384 == 0 up to 1152, 0 = never.
        static const int wide384 = getenv("VDA_GEMM_BN384") ? atoi(getenv("VDA_GEMM_BN384")) : 2;
        const bool ln_epi = a.epilogue == VDA_EPI_LN_BIAS_F16 || a.epilogue == VDA_EPI_LN_GELU_F16;
        const bool auto384 = g_gemm_variant < 0 && a.tile_rows == 0 &&
                             (wide384 == 1 ? (!eight && a.N <= 1152)
                                            : wide384 == 2 ? ((a.N == 384 && a.K >= 1024) || (ln_epi && a.N <= 1536))
                                            : wide384 == 3 ? (a.N == 384 && a.K >= 1024)                                     // (A/B: fc2 only)
                                            : wide384 == 4 ? ((a.N == 384 && a.K >= 1024) || (ln_epi && a.N == 1536)) : false);   // (A/B: fc2 + fc1)
        if (a.a_mode == VDA_A_DENSE && a.N % 384 == 0 && (auto384 || g_gemm_variant == 10)) {
            const int rc384 = vda_gemm256s_dense_bn384_bm192(a8, s);
            if (rc384 >= 0) {
                static thread_local char name384[64];
                snprintf(name384, sizeof(name384), "gemm256s_kernel<384, %d, %d, 192, 1>", a.a_mode, a.epilogue);
                g_last_kernel = name384;
                return rc384;
            }
        }
        // Row split (vda_gemm_plan_split): whole rounds of 256-row tiles + one launch of 192-row tiles for the remainder. Applied here
        // when the caller left it to the dispatcher (no per-launch sched counters: those belong to ONE launch).
        if (eight && big == 256 && a.a_mode == VDA_A_DENSE && g_gemm_variant < 0 && may_split && a.tile_rows == 0 && a.sched == nullptr && a.lda != 0) {
            const int m1 = vda_gemm_plan_split(a.M, a.N, a.K, a.epilogue, a.a_mode);
            if (m1 < a.M) {
                vda_gemm_args p1 = row_range(a, 0, m1), p2 = row_range(a, m1, a.M - m1);
                p2.tile_rows = 192;
                const int rc1 = vda_gemm_f16_impl(&p1, stream, false);
                if (rc1 != 0) return rc1;
                return vda_gemm_f16_impl(&p2, stream, false);
            }
        }
        if (eight && big == 256 && a.a_mode == VDA_A_DENSE && (a.tile_rows == 192 || g_gemm_variant == 5 + 16 * 64)) {
            const int rc192 = vda_gemm8p_dense_bn256_bm192(a8, s);
            if (rc192 >= 0) {
                static thread_local char name192[64];
                snprintf(name192, sizeof(name192), "gemm8p_kernel<256, %d, %d, 1, 192, %s>", a.a_mode, a.epilogue, a.sched ? "true" : "false");
                g_last_kernel = name192;
                return rc192;
            }
        }
        if (g_gemm_variant == 11 && a.a_mode == VDA_A_DENSE) {        // 192 x 128, two workgroups per CU (A/B only, see gemm256s_kernel.h)
            const int rcx2 = vda_gemm256s_dense_bn128_bm192_x2(a8, s);
            if (rcx2 >= 0) {
                static thread_local char namex2[64];
                snprintf(namex2, sizeof(namex2), "gemm256s_kernel<128, %d, %d, 192, 2>", a.a_mode, a.epilogue);
                g_last_kernel = namex2;
                return rcx2;
            }
        }
        if (tall192) {
            const int rc192 = vda_gemm256s_dense_bn128_bm192(a8, s);
            if (rc192 >= 0) {
                static thread_local char name192[64];
                snprintf(name192, sizeof(name192), "gemm256s_kernel<128, %d, %d, 192, 1>", a.a_mode, a.epilogue);
                g_last_kernel = name192;
                return rc192;
            }
        }
        const int rc = !eight ? (small_mfma ? vda_gemm256s_launch(a8, big, s) : vda_gemm256_launch(a, big, s))
                       : big == 128 ? (a.a_mode == VDA_A_DENSE ? vda_gemm8p_dense_bn128(a8, s) : vda_gemm8p_conv_bn128(a8, s))
                       : a.a_mode == VDA_A_DENSE ? (sched8 ? vda_gemm8p_dense_bn256_sched(a8, s, sched8) : vda_gemm8p_dense_bn256(a8, s))
                                                 : vda_gemm8p_conv_bn256(a8, s);
        if (rc >= 0) {
            // exact instantiation name as rocprofv3 prints it: gemm256[s]_kernel<BN, a_mode, epilogue> / gemm8p_kernel<...>
            static thread_local char name[64];
            // (every template argument, defaults included, as the profiler prints them)
            const int sch = sched8 == 1 ? 0 : sched8 == 2 ? 2 : 1;
            if (eight) snprintf(name, sizeof(name), "gemm8p_kernel<%d, %d, %d, %d, 256, %s>", big, a.a_mode, a.epilogue, sch, (a.sched && sch == 1) ? "true" : "false");
            else if (small_mfma) snprintf(name, sizeof(name), "gemm256s_kernel<%d, %d, %d, 256, 1>", big, a.a_mode, a.epilogue);
            else snprintf(name, sizeof(name), "gemm256_kernel<%d, %d, %d>", big, a.a_mode, a.epilogue);
            g_last_kernel = name;
            return rc;
        }                                   // -1: pair not built for the large tile, use the 128-row kernel
    }
    g_last_kernel = a.a_mode == VDA_A_DENSE ? (a.N <= 64 ? "gemm_kernel<128, 64, 0>" : "gemm_kernel<128, 128, 0>")
                                            : (a.N <= 64 ? "gemm_kernel<128, 64, 1>" : "gemm_kernel<128, 128, 1>");
    return launch_small(a, s);
}
