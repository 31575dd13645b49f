// Epilogues shared by the GEMM kernels: the caller hands one row m and 4 consecutive output columns n..n+3
// (fp32 accumulators); bias / activation / LayerScale / residual / layout scatter happen here.
#pragma once
#include "vda_common.h"
#include <type_traits>

namespace vda_gemm {

// fp32 -> fp16 of an epilogue result, always as "round to fp32, then round to fp16". Left alone, hipcc fuses the last fp32
// fma/mul of SOME elements with the conversion (v_fma_mix{lo,hi}_f16: one rounding) and not of others (v_fmac_f32 +
// v_cvt_pk_f16_f32: two roundings) - which ones depends on the unrolled copy, so identical rows could differ by one fp16
// ulp with their position in the tile (found by the duplicated-clip test). The empty asm makes the fp32 value opaque.
__device__ __forceinline__ h16 to_h16(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return (h16)v;
}

template <int EPI>
inline constexpr bool is_ln_epi = (EPI == VDA_EPI_LN_BIAS_F16 || EPI == VDA_EPI_LN_GELU_F16);

// x = hi + lo with hi = fp16(x), lo = fp16(x - hi): the split fp32 residual stream (VDA_EPI_SCALE_RES_SPLIT)
__device__ __forceinline__ void split_h16(float x, h16& hi, h16& lo) {
    hi = to_h16(x);
    lo = to_h16(x - (float)hi);
}

template <int CTRL>
__device__ __forceinline__ float epi_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// sum over the 8 lanes (aligned group) that own one row's 64 columns in the row-layout epilogue
__device__ __forceinline__ float sum8(float v) {
    v += epi_dpp<0xB1>(v);                  // quad_perm [1,0,3,2]
    v += epi_dpp<0x4E>(v);                  // quad_perm [2,3,0,1]
    v += epi_dpp<0x141>(v);                 // row_half_mirror
    return v;
}

template <int EPI>
__device__ __forceinline__ void store_one(const vda_gemm_args& p, int m, int n, f32x4 v, f32x4 g) {
    // v: accumulators for columns n..n+3 of row m (g: gate accumulators, GEGLU only).
    if (m >= p.M || n >= p.N) return;
    if constexpr (is_ln_epi<EPI>) {
        const float mu = p.stats[2 * (size_t)m], rstd = p.stats[2 * (size_t)m + 1];
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(p.gamma + n), c2 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = fmaf(rstd, fmaf(-mu, c1[i], v[i]), c2[i]);
            if constexpr (EPI == VDA_EPI_LN_GELU_F16) v[i] = gelu_erf(v[i]);
        }
        h16x4 o = {to_h16(v[0]), to_h16(v[1]), to_h16(v[2]), to_h16(v[3])};
        *reinterpret_cast<h16x4*>((h16*)p.out + (size_t)m * p.ldc + n) = o;
        return;
    }
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
    if constexpr (EPI == VDA_EPI_SCALE_RES_SPLIT) {
        // (the partial row statistics are written by a separate pass for this layout: see vda_gemm_f16)
        const size_t off = (size_t)m * p.ldc + n;
        const h16x4 rh = *reinterpret_cast<const h16x4*>((const h16*)p.res + off), rl = *reinterpret_cast<const h16x4*>((const h16*)p.res2 + off);
        const float ctr = p.pos[(size_t)m * p.P];           // re-centring: see load_row_aux
        h16x4 oh, ol;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float g = p.gamma ? p.gamma[n + i] : 1.f;
            const float x = fmaf(g, v[i], ((float)rh[i] + (float)rl[i]) - ctr);
            h16 a, b;
            split_h16(x, a, b);
            oh[i] = a;
            ol[i] = b;
        }
        *reinterpret_cast<h16x4*>((h16*)p.out + off) = oh;
        *reinterpret_cast<h16x4*>((h16*)p.out2 + off) = ol;
        return;
    }
    if constexpr (EPI == VDA_EPI_BIAS_F16 || EPI == VDA_EPI_BIAS_GELU_F16 || EPI == VDA_EPI_BIAS_RELU_F16) {
        if constexpr (EPI == VDA_EPI_BIAS_GELU_F16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
        }
        if constexpr (EPI == VDA_EPI_BIAS_RELU_F16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        h16x4 o = {to_h16(v[0]), to_h16(v[1]), to_h16(v[2]), to_h16(v[3])};
        *reinterpret_cast<h16x4*>((h16*)p.out + (size_t)m * p.ldc + n) = o;
    } else if constexpr (EPI == VDA_EPI_SCALE_RES_F32) {
        if (p.gamma) v *= *reinterpret_cast<const f32x4*>(p.gamma + n);
        const size_t off = (size_t)m * p.ldc + n;
        v += *reinterpret_cast<const f32x4*>((const float*)p.res + off);
        *reinterpret_cast<f32x4*>((float*)p.out + off) = v;
    } else if constexpr (EPI == VDA_EPI_RES_F16) {
        const size_t off = (size_t)m * p.ldc + n;
        h16x4 r = *reinterpret_cast<const h16x4*>((const h16*)p.res + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += (float)r[i];
        if (p.res2) {
            h16x4 r2 = *reinterpret_cast<const h16x4*>((const h16*)p.res2 + off);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += (float)r2[i];
        }
        h16x4 o = {to_h16(v[0]), to_h16(v[1]), to_h16(v[2]), to_h16(v[3])};
        *reinterpret_cast<h16x4*>((h16*)p.out + off) = o;
    } else if constexpr (EPI == VDA_EPI_GEGLU_F16) {
        // n indexes the interleaved weight rows [16 value | 16 gate] per 32; g belongs to n + 16.
        if (p.bias) g += *reinterpret_cast<const f32x4*>(p.bias + n + 16);
        const int oc = (n >> 5) * 16 + (n & 15);
        h16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = to_h16(v[i] * gelu_erf(g[i]));
        *reinterpret_cast<h16x4*>((h16*)p.out + (size_t)m * p.ldc + oc) = o;
    } else if constexpr (EPI == VDA_EPI_PATCH_F32) {
        const int f = m / p.P, q = m - f * p.P;
        v += *reinterpret_cast<const f32x4*>(p.pos + (size_t)(1 + q) * p.N + n);
        *reinterpret_cast<f32x4*>((float*)p.out + ((size_t)f * (p.P + 1) + 1 + q) * p.ldc + n) = v;
    } else if constexpr (EPI == VDA_EPI_CONVT_F16) {
        // m = (b, y, x) over the tH x tW input; n = (ky*k + kx)*Cout + co (bias pre-expanded to N).
        const int k = p.tK, Co = p.tCout;
        const int tap = n / Co, co = n - tap * Co;
        const int ky = tap / k, kx = tap - ky * k;
        const int hw = p.tH * p.tW;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / p.tW, x = rem - y * p.tW;
        const size_t orow = ((size_t)b * p.tH * k + (size_t)y * k + ky) * ((size_t)p.tW * k) + (size_t)x * k + kx;
        h16x4 o = {to_h16(v[0]), to_h16(v[1]), to_h16(v[2]), to_h16(v[3])};
        *reinterpret_cast<h16x4*>((h16*)p.out + orow * p.ldc + co) = o;
    } else if constexpr (EPI == VDA_EPI_BIAS_F32) {
        *reinterpret_cast<f32x4*>((float*)p.out + (size_t)m * p.ldc + n) = v;
    } else if constexpr (EPI == VDA_EPI_SCALE_RES_F32_H) {
        if (p.gamma) v *= *reinterpret_cast<const f32x4*>(p.gamma + n);
        const size_t off = (size_t)m * p.ldc + n;
        v += *reinterpret_cast<const f32x4*>((const float*)p.res + off);
        h16x4 o = {to_h16(v[0]), to_h16(v[1]), to_h16(v[2]), to_h16(v[3])};
        *reinterpret_cast<h16x4*>((h16*)p.out + off) = o;
    }
}


// ---------------------------------------------------------------------------------------------
// Row-layout epilogues (used after the accumulators have been transposed through LDS): the caller
// hands NC consecutive columns n..n+NC-1 of ONE row m, so bias / LayerScale / residual reads and the
// output store are contiguous 16-byte (or 32-byte) accesses and a wave instruction covers full lines.
// fp16-output epilogues take NC = 8, fp32-output ones NC = 4. `g` = gate columns (GEGLU only).
template <int EPI>
struct RowTraits {
    static constexpr bool f32_out = (EPI == VDA_EPI_SCALE_RES_F32 || EPI == VDA_EPI_BIAS_F32 || EPI == VDA_EPI_PATCH_F32);
    static constexpr int NC = f32_out ? 4 : 8;
};

__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = a[i];
        v[4 + i] = b[i];
    }
}

// NT: a non-temporal store (`global_store_dwordx4 ... nt`). For outputs far larger than the caches can hand to the next kernel (ViT-L's
// hid 359 MB, qkv 270 MB, the 148 x 148 conv maps 359 MB) it keeps the streamed lines from displacing the A / W panels in L2 and
// leaves fewer dirty lines behind at the kernel boundary: fc1 -2.0 %, qkv -2.8 %, the ViT-L forward 50.21 -> 49.67 ms (+1.1 %) in
// one process with two builds. ONLY for these full-line 16-byte row stores and only for big outputs: on ViT-S (hid 135 MB: the Infinity
// Cache hands it to fc2) it costs 0.6 %, and on the 8-byte-per-lane stores of the attention / conv kernels - which rely on L2 to merge
// them into lines - it costs 3.5 % of the ViT-L forward and 14 % of ViT-S (profiles/r04/nt_stores_ab.txt).
template <bool NT = false>
__device__ __forceinline__ void store8h(h16* p, const float (&v)[8]) {
    h16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = to_h16(v[i]);
    if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<h16x8*>(p));
    else *reinterpret_cast<h16x8*>(p) = o;
}

// Per-lane column constants (bias / LayerScale for the lane's NC columns), loaded once per tile.
template <int NC>
struct ColConst {
    float bias[NC];
    float gamma[NC];
    float gbias[NC];     // GEGLU: bias of the gate columns n+16..
};

// BIAS = false: the kernel started its accumulators at the bias (the 8-phase kernel does), so the epilogue neither loads nor adds it.
template <int EPI, int NC, bool BIAS = true>
__device__ __forceinline__ void load_col_const(const vda_gemm_args& p, int n, ColConst<NC>& c) {
    const bool ok = n < p.N;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        c.bias[i] = 0.f;
        c.gamma[i] = 1.f;
        c.gbias[i] = 0.f;
    }
    if (!ok) return;
    if ((BIAS || is_ln_epi<EPI>) && p.bias) {
#pragma unroll
        for (int i = 0; i < NC; i += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p.bias + n + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) c.bias[i + e] = t[e];
        }
        if constexpr (EPI == VDA_EPI_GEGLU_F16) {
            if ((n & 31) < 16) {        // value lanes only: for a gate lane n+16.. runs past the end of bias
#pragma unroll
            for (int i = 0; i < NC; i += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(p.bias + n + 16 + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) c.gbias[i + e] = t[e];
            }
            }
        }
    }
    if constexpr (EPI == VDA_EPI_SCALE_RES_F32 || EPI == VDA_EPI_SCALE_RES_F32_H || EPI == VDA_EPI_SCALE_RES_SPLIT || is_ln_epi<EPI>) {
        if (p.gamma) {
#pragma unroll
            for (int i = 0; i < NC; i += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(p.gamma + n + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) c.gamma[i + e] = t[e];
            }
        }
    }
}

// Row-dependent operands fetched from memory (residuals, pos-embed). Loaded for a whole 32-row block BEFORE any
// of the block's stores is issued: vmcnt retires in order and counts stores, so interleaving load/store pairs
// would serialise one memory round trip per row group.
struct RowAux {
    f32x4 f0, f1;
    h16x8 h0, h1;
    float s0, s1;        // LayerNorm-folded epilogues: the row's (mean, rstd)
    mutable float o0, o1;   // VDA_EPI_SCALE_RES_SPLIT: (sum, centred sum of squares) of the row's 64 columns, for the caller to store
};

// [column block][row] layout of the partial statistics: the 8 rows a wave instruction covers are one 64-byte run
__device__ __forceinline__ void store_split_stats(const vda_gemm_args& p, int m, int n, const RowAux& x) {
    *reinterpret_cast<float2*>(p.stats + ((size_t)(n >> 6) * (p.stats_ld ? p.stats_ld : p.M) + m) * 2) = float2{x.o0, x.o1};   // (stats_ld: a row range of a larger GEMM)
}

// GUARD = false: the caller has established that every row / column of the wave's tile is inside the matrix. The bounds
// test is not free: it puts every row's loads and stores in their own basic block, and hipcc then waits vmcnt(0) at each
// (the number of stores issued so far is unknown at the join), which serialises one memory round trip per row.
template <int EPI, bool GUARD = true>
__device__ __forceinline__ void load_row_aux(const vda_gemm_args& p, int m, int n, RowAux& x) {
    if (GUARD && (m >= p.M || n >= p.N)) return;
    if constexpr (EPI == VDA_EPI_SCALE_RES_F32) {
        x.f0 = *reinterpret_cast<const f32x4*>((const float*)p.res + (size_t)m * p.ldc + n);
    } else if constexpr (EPI == VDA_EPI_SCALE_RES_F32_H) {
        const float* r = (const float*)p.res + (size_t)m * p.ldc + n;
        x.f0 = *reinterpret_cast<const f32x4*>(r);
        x.f1 = *reinterpret_cast<const f32x4*>(r + 4);
    } else if constexpr (EPI == VDA_EPI_RES_F16) {
        const size_t off = (size_t)m * p.ldc + n;
        x.h0 = *reinterpret_cast<const h16x8*>((const h16*)p.res + off);
        if (p.res2) x.h1 = *reinterpret_cast<const h16x8*>((const h16*)p.res2 + off);
    } else if constexpr (EPI == VDA_EPI_PATCH_F32) {
        const int q = m % p.P;
        x.f0 = *reinterpret_cast<const f32x4*>(p.pos + (size_t)(1 + q) * p.N + n);
    } else if constexpr (EPI == VDA_EPI_SCALE_RES_SPLIT) {
        const size_t off = (size_t)m * p.ldc + n;
        // (read once, by this lane only: non-temporal, like the lo plane's store below - proj alone 134.6 -> 129.6 us, the ViT-L
        // forward 51.93 -> 51.87 ms in one process; hi is stored normally: the next GEMM stages it as its A operand)
        x.h0 = __builtin_nontemporal_load(reinterpret_cast<const h16x8*>((const h16*)p.res + off));
        x.h1 = __builtin_nontemporal_load(reinterpret_cast<const h16x8*>((const h16*)p.res2 + off));
        // Re-centring of the split stream (vda.h): pos = the (mean, rstd) rows the preceding LayerNorm-folded GEMM consumed; the
        // row's mean is taken out of the stream as the update goes in, so the planes always hold the token relative to (about)
        // its own mean and the fp16 rounding of the operand plane is relative to the token's spread, not to its offset. Every
        // reader of the stream is a LayerNorm (shift-invariant per row). No pos: the dispatcher points it at a zero page with
        // P = 0 (an unconditional load: a branch around it would make hipcc drain vmcnt per row, cdna_hip_programming.md 5, trap c).
        x.s0 = p.pos[(size_t)m * p.P];
    } else if constexpr (is_ln_epi<EPI>) {
        const float2 st = *reinterpret_cast<const float2*>(p.stats + 2 * (size_t)m);
        x.s0 = st.x;
        x.s1 = st.y;
    }
}

template <int EPI, bool GUARD = true, bool BIAS = true, bool NT = false>
__device__ __forceinline__ void finish_row8(const vda_gemm_args& p, int m, int n, float (&v)[8], float (&g)[8],
                                            const ColConst<8>& c, const RowAux& x) {
    if (GUARD && (m >= p.M || n >= p.N)) return;
    if constexpr (BIAS && !is_ln_epi<EPI>) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += c.bias[i];
    }
    if constexpr (is_ln_epi<EPI>) {
        // LayerNorm applied AFTER the GEMM: acc = sum_k hi[m,k] * (W[n,k] * ln_w[k]); c.gamma = c1 (row sums of the folded W), c.bias = c2
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(x.s1, fmaf(-x.s0, c.gamma[i], v[i]), c.bias[i]);
        if constexpr (EPI == VDA_EPI_LN_GELU_F16) gelu_erf_n(v);
        store8h<NT>((h16*)p.out + (size_t)m * p.ldc + n, v);
    } else if constexpr (EPI == VDA_EPI_SCALE_RES_SPLIT) {
        // the row's 64 columns of this wave tile sit in 8 consecutive lanes (all inside the matrix together: N % 64 == 0)
        h16x8 oh, ol;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = fmaf(c.gamma[i], v[i], ((float)x.h0[i] + (float)x.h1[i]) - x.s0);    // hi + lo is exact in fp32; s0: re-centring
            h16 a, b;
            split_h16(v[i], a, b);
            oh[i] = a;
            ol[i] = b;
        }
        const size_t off = (size_t)m * p.ldc + n;
        *reinterpret_cast<h16x8*>((h16*)p.out + off) = oh;
        __builtin_nontemporal_store(ol, reinterpret_cast<h16x8*>((h16*)p.out2 + off));
        // partial statistics of the row's 64 columns (all 8 lanes end up with the same pair); the caller stores them
        sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        sum = sum8(sum);
        const float mean = sum * (1.f / 64.f);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float d = v[i] - mean;
            sq = fmaf(d, d, sq);
        }
        x.o0 = sum;
        x.o1 = sum8(sq);
    } else if constexpr (EPI == VDA_EPI_BIAS_F16) {
        store8h<NT>((h16*)p.out + (size_t)m * p.ldc + n, v);
    } else if constexpr (EPI == VDA_EPI_BIAS_GELU_F16) {
        gelu_erf_n(v);
        store8h<NT>((h16*)p.out + (size_t)m * p.ldc + n, v);
    } else if constexpr (EPI == VDA_EPI_BIAS_RELU_F16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
        store8h<NT>((h16*)p.out + (size_t)m * p.ldc + n, v);
    } else if constexpr (EPI == VDA_EPI_RES_F16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += (float)x.h0[i];
        if (p.res2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += (float)x.h1[i];
        }
        store8h<NT>((h16*)p.out + (size_t)m * p.ldc + n, v);
    } else if constexpr (EPI == VDA_EPI_SCALE_RES_F32_H) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = fmaf(c.gamma[i], v[i], x.f0[i]);                 // explicit fma: identical rounding in every instantiation
            v[4 + i] = fmaf(c.gamma[4 + i], v[4 + i], x.f1[i]);
        }
        store8h<NT>((h16*)p.out + (size_t)m * p.ldc + n, v);
    } else if constexpr (EPI == VDA_EPI_GEGLU_F16) {
        // n is a VALUE column group (n % 32 < 16); g holds columns n+16.. (the gates)
        if constexpr (BIAS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] += c.gbias[i];
        }
        gelu_erf_n(g);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= g[i];
        store8h<NT>((h16*)p.out + (size_t)m * p.ldc + ((n >> 5) * 16 + (n & 15)), v);
    } else if constexpr (EPI == VDA_EPI_CONVT_F16) {
        const int k = p.tK, Co = p.tCout;
        const int tap = n / Co, co = n - tap * Co;
        const int ky = tap / k, kx = tap - ky * k;
        const int hw = p.tH * p.tW;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / p.tW, xx = rem - y * p.tW;
        const size_t orow = ((size_t)b * p.tH * k + (size_t)y * k + ky) * ((size_t)p.tW * k) + (size_t)xx * k + kx;
        store8h<NT>((h16*)p.out + orow * p.ldc + co, v);
    }
}

template <int EPI, bool GUARD = true, bool BIAS = true>
__device__ __forceinline__ void finish_row4(const vda_gemm_args& p, int m, int n, f32x4 v, const ColConst<4>& c, const RowAux& x) {
    if (GUARD && (m >= p.M || n >= p.N)) return;
    if constexpr (BIAS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += c.bias[i];
    }
    if constexpr (EPI == VDA_EPI_SCALE_RES_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaf(c.gamma[i], v[i], x.f0[i]);   // explicit fma: identical rounding in every instantiation
        *reinterpret_cast<f32x4*>((float*)p.out + (size_t)m * p.ldc + n) = v;
    } else if constexpr (EPI == VDA_EPI_BIAS_F32) {
        *reinterpret_cast<f32x4*>((float*)p.out + (size_t)m * p.ldc + n) = v;
    } else if constexpr (EPI == VDA_EPI_PATCH_F32) {
        const int f = m / p.P, q = m - f * p.P;
        v += x.f0;
        *reinterpret_cast<f32x4*>((float*)p.out + ((size_t)f * (p.P + 1) + 1 + q) * p.ldc + n) = v;
    }
}

// Run `f(tag)` with the runtime epilogue id lifted to a compile-time constant.
template <class F>
__device__ __forceinline__ void dispatch_epilogue(int epilogue, F&& f) {
    switch (epilogue) {
        case VDA_EPI_BIAS_F16: f(std::integral_constant<int, VDA_EPI_BIAS_F16>{}); break;
        case VDA_EPI_BIAS_GELU_F16: f(std::integral_constant<int, VDA_EPI_BIAS_GELU_F16>{}); break;
        case VDA_EPI_BIAS_RELU_F16: f(std::integral_constant<int, VDA_EPI_BIAS_RELU_F16>{}); break;
        case VDA_EPI_SCALE_RES_F32: f(std::integral_constant<int, VDA_EPI_SCALE_RES_F32>{}); break;
        case VDA_EPI_RES_F16: f(std::integral_constant<int, VDA_EPI_RES_F16>{}); break;
        case VDA_EPI_GEGLU_F16: f(std::integral_constant<int, VDA_EPI_GEGLU_F16>{}); break;
        case VDA_EPI_PATCH_F32: f(std::integral_constant<int, VDA_EPI_PATCH_F32>{}); break;
        case VDA_EPI_CONVT_F16: f(std::integral_constant<int, VDA_EPI_CONVT_F16>{}); break;
        case VDA_EPI_BIAS_F32: f(std::integral_constant<int, VDA_EPI_BIAS_F32>{}); break;
        case VDA_EPI_SCALE_RES_F32_H: f(std::integral_constant<int, VDA_EPI_SCALE_RES_F32_H>{}); break;
        case VDA_EPI_SCALE_RES_SPLIT: f(std::integral_constant<int, VDA_EPI_SCALE_RES_SPLIT>{}); break;
        case VDA_EPI_LN_BIAS_F16: f(std::integral_constant<int, VDA_EPI_LN_BIAS_F16>{}); break;
        case VDA_EPI_LN_GELU_F16: f(std::integral_constant<int, VDA_EPI_LN_GELU_F16>{}); break;
        default: break;
    }
}

}  // namespace vda_gemm
