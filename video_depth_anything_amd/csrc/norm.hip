// LayerNorm (fp32 rows -> fp16) and per-frame GroupNorm (NHWC fp16 -> token-major fp16) for gfx950.
// Both are HBM-bound streaming kernels: 16-byte accesses, statistics in fp32, one pass over the
// data for LayerNorm (row kept in registers), two launches with a fixed-order (deterministic)
// partial-sum tree for GroupNorm.
#include "vda_common.h"

namespace {

// ---------------------------------------------------------------- LayerNorm
// A row is split over LPR lanes, 8 consecutive elements per lane per chunk (two 16-byte loads in, one 16-byte
// store out); a wave holds 64/LPR rows, so small D (temporal modules: 64..256) still fills the wave. The row
// stays in registers; statistics are fp32, two-pass (mean, then centred sum of squares). Cross-lane sums use
// DPP (quad_perm / row_half_mirror / row_mirror) and v_readlane, not ds_bpermute: no LDS round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_xchg(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

template <int LPR>
__device__ __forceinline__ float segment_sum(float v, int lane) {
    v += dpp_xchg<0xB1>(v);                 // quad_perm [1,0,3,2]
    v += dpp_xchg<0x4E>(v);                 // quad_perm [2,3,0,1]
    v += dpp_xchg<0x141>(v);                // row_half_mirror: 8 lanes
    if constexpr (LPR >= 16) v += dpp_xchg<0x140>(v);    // row_mirror: 16 lanes
    if constexpr (LPR == 32) {
        const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) +
                         __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
        const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) +
                         __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
        v = lane < 32 ? lo : hi;
    }
    if constexpr (LPR == 64) {
        v = (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) +
             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16))) +
            (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) +
             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48)));
    }
    return v;
}

// RES: the row is first updated in place, in[r,:] += gamma * y[r,:] (y fp16: the bias-added output of the preceding projection,
// gamma = LayerScale or NULL = 1) - the encoder's residual add (block.py:105-106) rides on the LayerNorm that follows it, so the
// projection GEMM stores 2 bytes per element instead of reading and writing the fp32 stream in its (fabric-bound) epilogue.
template <int LPR, int NCH, typename OT, bool RES>
__global__ void __launch_bounds__(256) layernorm_kernel(float* __restrict__ in, OT* __restrict__ out,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float eps, int rows, int D, int group, int skip,
                                                        const float* __restrict__ pe, int pe_rows_per_step, int pe_steps,
                                                        const h16* __restrict__ y, const float* __restrict__ gamma) {
    constexpr int RPW = 64 / LPR;                                  // rows per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, rsel = lane / LPR;
    const int row = (blockIdx.x * 4 + wave) * RPW + rsel;
    const bool row_ok = row < rows;
    const int nchunk = D >> 3;                                     // 8-element chunks per row
    float* src = in + (size_t)(row_ok ? row : 0) * D;
    f32x4 v[NCH][2];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = sub + c * LPR;
        const bool ok = row_ok && ch < nchunk;
        v[c][0] = v[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) {
            v[c][0] = *reinterpret_cast<const f32x4*>(src + ch * 8);
            v[c][1] = *reinterpret_cast<const f32x4*>(src + ch * 8 + 4);
            if constexpr (RES) {
                const h16x8 yy = *reinterpret_cast<const h16x8*>(y + (size_t)row * D + ch * 8);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 g = {1.f, 1.f, 1.f, 1.f};
                    if (gamma) g = *reinterpret_cast<const f32x4*>(gamma + ch * 8 + h * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[c][h][e] = fmaf(g[e], (float)yy[h * 4 + e], v[c][h][e]);
                    *reinterpret_cast<f32x4*>(src + ch * 8 + h * 4) = v[c][h];
                }
            }
        }
        sum += ((v[c][0][0] + v[c][0][1]) + (v[c][0][2] + v[c][0][3])) + ((v[c][1][0] + v[c][1][1]) + (v[c][1][2] + v[c][1][3]));
    }
    const float mean = segment_sum<LPR>(sum, lane) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const bool ok = sub + c * LPR < nchunk;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = ok ? v[c][h][e] - mean : 0.f;
                sq += d * d;
            }
    }
    const float rstd = rsqrtf(segment_sum<LPR>(sq, lane) / (float)D + eps);
    if (!row_ok) return;
    int orow = row;
    if (group > 0) {
        const int g = row / group, i = row - g * group;
        if (i < skip) return;                                       // dropped row (cls token)
        orow = g * (group - skip) + (i - skip);
    }
    const float* per = pe ? pe + (size_t)((row / pe_rows_per_step) % pe_steps) * D : nullptr;
    OT* dst = out + (size_t)orow * D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = sub + c * LPR;
        if (ch < nchunk) {
            float o[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 ww = *reinterpret_cast<const f32x4*>(w + ch * 8 + h * 4);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b + ch * 8 + h * 4);
                f32x4 y = (v[c][h] - mean) * rstd * ww + bb;
                if (per) y += *reinterpret_cast<const f32x4*>(per + ch * 8 + h * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[h * 4 + e] = y[e];
            }
            store8(dst + ch * 8, o);
        }
    }
}

// ---------------------------------------------------------------- GroupNorm
// Stage 1: grid (chunks, frames). A block sums x and x^2 per channel over its rows of one frame,
// folds channels into groups in a fixed order, writes partial[frame][chunk][group][2].
template <typename T>
__global__ void __launch_bounds__(256) groupnorm_partial_kernel(const T* __restrict__ in, float* __restrict__ partial,
                                                                int hw, int C, int groups, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [rows_par][C][2] then [C][2]
    const int frame = blockIdx.y, chunk = blockIdx.x;
    const int nv = C >> 3;                       // 16-byte vectors per row
    const int rows_par = 256 / nv;               // rows processed concurrently (>= 2 for C <= 1024)
    const int tid = threadIdx.x;
    const int rl = tid / nv, vi = tid - rl * nv;
    const int rows_per_chunk = (hw + chunks - 1) / chunks;
    const int r0 = chunk * rows_per_chunk, r1 = min(hw, r0 + rows_per_chunk);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    if (rl < rows_par) {
        const T* base = in + (size_t)frame * hw * C + vi * 8;
        for (int r = r0 + rl; r < r1; r += rows_par) {
            float x[8];
            load8(base + (size_t)r * C, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s[e] += x[e];
                q[e] += x[e] * x[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[((size_t)rl * C + vi * 8 + e) * 2 + 0] = s[e];
            red[((size_t)rl * C + vi * 8 + e) * 2 + 1] = q[e];
        }
    }
    __syncthreads();
    float* chs = red + (size_t)rows_par * C * 2;                 // per-channel totals
    for (int c = tid; c < C; c += 256) {
        float a = 0.f, bq = 0.f;
        for (int k = 0; k < rows_par; ++k) {
            a += red[((size_t)k * C + c) * 2 + 0];
            bq += red[((size_t)k * C + c) * 2 + 1];
        }
        chs[c * 2 + 0] = a;
        chs[c * 2 + 1] = bq;
    }
    __syncthreads();
    if (tid < groups) {
        const int cpg = C / groups;
        float a = 0.f, bq = 0.f;
        for (int k = 0; k < cpg; ++k) {
            a += chs[(tid * cpg + k) * 2 + 0];
            bq += chs[(tid * cpg + k) * 2 + 1];
        }
        float* p = partial + (((size_t)frame * chunks + chunk) * groups + tid) * 2;
        p[0] = a;
        p[1] = bq;
    }
}

// Stage 2: grid (row blocks, frames). Combine the frame's partials (double, fixed order), normalise.
template <typename T>
__global__ void __launch_bounds__(256) groupnorm_apply_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                              const float* __restrict__ w, const float* __restrict__ b,
                                                              const float* __restrict__ partial, float eps, int hw, int C,
                                                              int groups, int chunks, int rows_per_block) {
    __shared__ float mean_s[64], rstd_s[64];
    const int frame = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / groups;
    if (tid < groups) {
        double a = 0.0, q = 0.0;
        for (int k = 0; k < chunks; ++k) {
            const float* p = partial + (((size_t)frame * chunks + k) * groups + tid) * 2;
            a += (double)p[0];
            q += (double)p[1];
        }
        const double n = (double)hw * cpg;
        const double mean = a / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[tid] = (float)mean;
        rstd_s[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int nv = C >> 3;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(hw, r0 + rows_per_block);
    const size_t fbase = (size_t)frame * hw * C;
    for (int idx = tid; idx < (r1 - r0) * nv; idx += 256) {
        const int r = r0 + idx / nv, vi = idx % nv;
        const size_t off = fbase + (size_t)r * C + vi * 8;
        float x[8], y[8];
        load8(in + off, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = vi * 8 + e;
            const int g = c / cpg;
            y[e] = (x[e] - mean_s[g]) * rstd_s[g] * w[c] + b[c];
        }
        store8(out + off, y);
    }
}

// ---------------------------------------------------------------- LayerNorm folded into the neighbouring GEMMs
// The fp32 residual stream of the encoder is kept as two fp16 planes, x = hi + lo (vda.h, VDA_EPI_SCALE_RES_SPLIT).
//   MODE 0 (vda_split_stats_f32): fp32 rows -> hi, lo planes + stat[row] = (mean, rstd): the entry into the split stream.
//   MODE 1 (vda_layernorm_split_f16): LayerNorm(hi + lo) * w + b -> fp16, group / skip as layernorm_kernel (the taps).
//   MODE 2 (vda_split_center_stats_f32): as MODE 0 with the row's mean taken out: hi + lo = x - mean(x), stat[row] = (0, rstd).
// Same row-in-registers, two-pass fp32 statistics as layernorm_kernel.
template <int LPR, int NCH, int MODE>
__global__ void __launch_bounds__(256) ln_split_kernel(const float* __restrict__ xin, const h16* __restrict__ hin, const h16* __restrict__ lin,
                                                       h16* __restrict__ o0, h16* __restrict__ o1, float* __restrict__ stat,
                                                       const float* __restrict__ w, const float* __restrict__ b, float eps, int rows, int D,
                                                       int group, int skip) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, rsel = lane / LPR;
    const int row = (blockIdx.x * 4 + wave) * RPW + rsel;
    const bool row_ok = row < rows;
    const int nchunk = D >> 3;
    const size_t base = (size_t)(row_ok ? row : 0) * D;
    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = sub + c * LPR;
        const bool ok = row_ok && ch < nchunk;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
        if (ok) {
            if constexpr (MODE != 1) {
                load8(xin + base + ch * 8, v[c]);
            } else {
                const h16x8 a = *reinterpret_cast<const h16x8*>(hin + base + ch * 8), l = *reinterpret_cast<const h16x8*>(lin + base + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c][e] = (float)a[e] + (float)l[e];
            }
        }
        sum += ((v[c][0] + v[c][1]) + (v[c][2] + v[c][3])) + ((v[c][4] + v[c][5]) + (v[c][6] + v[c][7]));
    }
    const float mean = segment_sum<LPR>(sum, lane) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const bool ok = sub + c * LPR < nchunk;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = ok ? v[c][e] - mean : 0.f;
            sq += d * d;
        }
    }
    const float rstd = rsqrtf(segment_sum<LPR>(sq, lane) / (float)D + eps);
    if (!row_ok) return;
    if constexpr (MODE != 1) {
        // MODE 2: the planes hold x - mean (the mean of what they hold is 0 up to fp32 rounding: 1e-7 of the row's spread)
        const float ctr = MODE == 2 ? mean : 0.f;
        if (sub == 0) *reinterpret_cast<float2*>(stat + 2 * (size_t)row) = float2{mean - ctr, rstd};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = sub + c * LPR;
            if (ch < nchunk) {
                h16x8 a, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[c][e] - ctr;
                    a[e] = (h16)d;
                    l[e] = (h16)(d - (float)a[e]);
                }
                *reinterpret_cast<h16x8*>(o0 + base + ch * 8) = a;
                *reinterpret_cast<h16x8*>(o1 + base + ch * 8) = l;
            }
        }
    } else {
        int orow = row;
        if (group > 0) {
            const int g = row / group, i = row - g * group;
            if (i < skip) return;
            orow = g * (group - skip) + (i - skip);
        }
        h16* dst = o0 + (size_t)orow * D;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = sub + c * LPR;
            if (ch < nchunk) {
                float o[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 ww = *reinterpret_cast<const f32x4*>(w + ch * 8 + h * 4);
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(b + ch * 8 + h * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[h * 4 + e] = (v[c][h * 4 + e] - mean) * rstd * ww[e] + bb[e];
                }
                store8(dst + ch * 8, o);
            }
        }
    }
}

// partial[np, r, 2] = (sum, centred sum of squares) per 64 columns -> stat[r] = (mean, rstd); pairwise update in column order
// (Chan, Golub, LeVeque): no E[x^2] - mean^2 cancellation, fixed order.
// overflow (optional): set to 1 when a row's statistics are not finite - the fp16 planes of the split stream saturate at 65 504 from
// the token's own mean, and a saturated hi plane shows here as an infinite / NaN sum (every element of it is under these sums).
__global__ void __launch_bounds__(64) ln_stats_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stat, float eps, int rows, int np,
                                                               int* __restrict__ overflow) {
    const int r = blockIdx.x * 64 + threadIdx.x;          // one wave per workgroup: 685 workgroups for a ViT-L clip, every CU gets some
    if (r >= rows) return;
    const float2* p = reinterpret_cast<const float2*>(partial) + r;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    // the loads of a batch are issued together (a row's np partials are np independent 8-byte loads, a column block apart): the
    // plain loop waited for each before the next (8 us per launch for 5.6 MB, 47 launches per ViT-L clip)
    constexpr int BATCH = 8;
    for (int j0 = 0; j0 < np; j0 += BATCH) {
        float2 q[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) q[u] = p[(size_t)min(j0 + u, np - 1) * rows];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            if (j0 + u < np) {                   // same arithmetic, same order as before: the combination is a fixed sequence
                const float mb = q[u].x * (1.f / 64.f), delta = mb - mean, nn = n + 64.f;
                mean += delta * (64.f / nn);
                m2 += q[u].y + delta * delta * (n * 64.f / nn);
                n = nn;
            }
        }
    }
    *reinterpret_cast<float2*>(stat + 2 * (size_t)r) = float2{mean, rsqrtf(m2 / n + eps)};
    if (overflow != nullptr && !(__builtin_isfinite(mean) && __builtin_isfinite(m2))) *overflow = 1;      // (same value from every writer)
}

// Pack-time fold of LayerNorm's affine into the following Linear: one wave per output row.
__global__ void __launch_bounds__(64) fold_ln_weight_kernel(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ ln_w,
                                                            const float* __restrict__ ln_b, h16* __restrict__ Wf, float* __restrict__ c1,
                                                            float* __restrict__ c2, int N, int K) {
    const int n = blockIdx.x, lane = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float wv = W[(size_t)n * K + k];
        float t = wv * ln_w[k];
        asm volatile("" : "+v"(t));             // round to fp32, then to fp16 (no fused multiply-convert): the documented definition
        const h16 f = (h16)t;
        Wf[(size_t)n * K + k] = f;
        s1 += (float)f;
        s2 = fmaf(wv, ln_b[k], s2);
    }
    s1 = segment_sum<64>(s1, lane);
    s2 = segment_sum<64>(s2, lane);
    if (lane == 0) {
        c1[n] = s1;
        c2[n] = s2 + (bias ? bias[n] : 0.f);
    }
}

}  // namespace

template <typename OT, bool RES = false>
static int layernorm_launch(float* in, OT* out, const float* w, const float* b, float eps, int rows, int D, int group, int skip,
                            const float* pe, int pe_rows_per_step, int pe_steps, vda_stream_t stream, const h16* y = nullptr,
                            const float* gamma = nullptr) {
    VDA_REQUIRE(!RES || (y != nullptr && ((uintptr_t)y & 15) == 0 && ((uintptr_t)gamma & 15) == 0), "vda_layernorm_residual: y must be a 16-byte aligned pointer");
    VDA_REQUIRE(in && out && w && b, "vda_layernorm: null pointer");
    VDA_REQUIRE(rows > 0 && D > 0 && D % 8 == 0 && D <= 2048, "vda_layernorm: D=%d must be a multiple of 8 and <= 2048", D);
    VDA_REQUIRE(group == 0 || (group > 0 && skip >= 0 && skip < group && rows % group == 0), "vda_layernorm: bad group/skip");
    VDA_REQUIRE(pe == nullptr || (pe_rows_per_step > 0 && pe_steps > 0), "vda_layernorm: bad pe geometry");
    VDA_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)b & 15) == 0 &&
                    ((uintptr_t)pe & 15) == 0,
                "vda_layernorm: 16-byte alignment required");
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = D / 8;
#define VDA_LN_LAUNCH(LPR, NCH)                                                                                                     \
    hipLaunchKernelGGL((layernorm_kernel<LPR, NCH, OT, RES>), dim3((rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR))), dim3(256), 0, s, in, \
                       out, w, b, eps, rows, D, group, skip, pe, pe_rows_per_step, pe_steps, y, gamma)
    if (nchunk <= 8) VDA_LN_LAUNCH(8, 1);
    else if (nchunk <= 16) VDA_LN_LAUNCH(16, 1);
    else if (nchunk <= 32) VDA_LN_LAUNCH(32, 1);
    else if (nchunk <= 64) VDA_LN_LAUNCH(64, 1);
    else if (nchunk <= 128) VDA_LN_LAUNCH(64, 2);
    else VDA_LN_LAUNCH(64, 4);
#undef VDA_LN_LAUNCH
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_layernorm_f32_f16(const float* in, void* out, const float* w, const float* b, float eps, int rows, int D,
                                     int group, int skip, const float* pe, int pe_rows_per_step, int pe_steps,
                                     vda_stream_t stream) {
    return layernorm_launch<h16>(const_cast<float*>(in), (h16*)out, w, b, eps, rows, D, group, skip, pe, pe_rows_per_step, pe_steps, stream);
}

extern "C" int vda_layernorm_residual_f32_f16(float* x, const void* y, const float* gamma, void* out, const float* w, const float* b,
                                              float eps, int rows, int D, int group, int skip, vda_stream_t stream) {
    return layernorm_launch<h16, true>(x, (h16*)out, w, b, eps, rows, D, group, skip, nullptr, 0, 0, stream, (const h16*)y, gamma);
}

extern "C" int vda_layernorm_f32_f32(const float* in, float* out, const float* w, const float* b, float eps, int rows, int D,
                                     int group, int skip, const float* pe, int pe_rows_per_step, int pe_steps,
                                     vda_stream_t stream) {
    return layernorm_launch<float>(const_cast<float*>(in), out, w, b, eps, rows, D, group, skip, pe, pe_rows_per_step, pe_steps, stream);
}

template <int MODE>
static int ln_split_launch(const float* xin, const h16* hin, const h16* lin, h16* o0, h16* o1, float* stat, const float* w, const float* b, float eps,
                           int rows, int D, int group, int skip, vda_stream_t stream) {
    VDA_REQUIRE(rows > 0 && D > 0 && D % 8 == 0 && D <= 2048, "vda_layernorm_split / vda_split_stats: D=%d must be a multiple of 8 and <= 2048", D);
    VDA_REQUIRE(group == 0 || (group > 0 && skip >= 0 && skip < group && rows % group == 0), "vda_layernorm_split: bad group/skip");
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = D / 8;
#define VDA_LNS_LAUNCH(LPR, NCH)                                                                                                          \
    hipLaunchKernelGGL((ln_split_kernel<LPR, NCH, MODE>), dim3((rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR))), dim3(256), 0, s, xin, hin, lin, \
                       o0, o1, stat, w, b, eps, rows, D, group, skip)
    if (nchunk <= 8) VDA_LNS_LAUNCH(8, 1);
    else if (nchunk <= 16) VDA_LNS_LAUNCH(16, 1);
    else if (nchunk <= 32) VDA_LNS_LAUNCH(32, 1);
    else if (nchunk <= 64) VDA_LNS_LAUNCH(64, 1);
    else if (nchunk <= 128) VDA_LNS_LAUNCH(64, 2);
    else VDA_LNS_LAUNCH(64, 4);
#undef VDA_LNS_LAUNCH
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_split_stats_f32(const float* x, void* hi, void* lo, float* stat, float eps, int rows, int D, vda_stream_t stream) {
    VDA_REQUIRE(x && hi && lo && stat, "vda_split_stats_f32: null pointer");
    VDA_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)hi & 15) == 0 && ((uintptr_t)lo & 15) == 0 && ((uintptr_t)stat & 7) == 0,
                "vda_split_stats_f32: 16-byte alignment required");
    return ln_split_launch<0>(x, nullptr, nullptr, (h16*)hi, (h16*)lo, stat, nullptr, nullptr, eps, rows, D, 0, 0, stream);
}

extern "C" int vda_split_center_stats_f32(const float* x, void* hi, void* lo, float* stat, float eps, int rows, int D, vda_stream_t stream) {
    VDA_REQUIRE(x && hi && lo && stat, "vda_split_center_stats_f32: null pointer");
    VDA_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)hi & 15) == 0 && ((uintptr_t)lo & 15) == 0 && ((uintptr_t)stat & 7) == 0,
                "vda_split_center_stats_f32: 16-byte alignment required");
    return ln_split_launch<2>(x, nullptr, nullptr, (h16*)hi, (h16*)lo, stat, nullptr, nullptr, eps, rows, D, 0, 0, stream);
}

extern "C" int vda_layernorm_split_f16(const void* hi, const void* lo, void* out, const float* w, const float* b, float eps, int rows, int D,
                                       int group, int skip, vda_stream_t stream) {
    VDA_REQUIRE(hi && lo && out && w && b, "vda_layernorm_split_f16: null pointer");
    VDA_REQUIRE(((uintptr_t)hi & 15) == 0 && ((uintptr_t)lo & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)b & 15) == 0,
                "vda_layernorm_split_f16: 16-byte alignment required");
    return ln_split_launch<1>(nullptr, (const h16*)hi, (const h16*)lo, (h16*)out, nullptr, nullptr, w, b, eps, rows, D, group, skip, stream);
}

extern "C" int vda_ln_stats_finalize(const float* partial, float* stat, float eps, int rows, int np, int32_t* overflow, vda_stream_t stream) {
    VDA_REQUIRE(partial && stat && rows > 0 && np > 0, "vda_ln_stats_finalize: bad arguments");
    VDA_REQUIRE(((uintptr_t)partial & 7) == 0 && ((uintptr_t)stat & 7) == 0, "vda_ln_stats_finalize: 8-byte alignment required");
    hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, partial, stat, eps, rows, np, (int*)overflow);
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_fold_ln_weight(const float* W, const float* bias, const float* ln_w, const float* ln_b, void* Wf, float* c1, float* c2, int N,
                                  int K, vda_stream_t stream) {
    VDA_REQUIRE(W && ln_w && ln_b && Wf && c1 && c2 && N > 0 && K > 0, "vda_fold_ln_weight: bad arguments");
    hipLaunchKernelGGL(fold_ln_weight_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, W, bias, ln_w, ln_b, (h16*)Wf, c1, c2, N, K);
    VDA_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int groupnorm_launch(const T* in, T* out, const float* w, const float* b, float eps, int frames, int hw, int C, int groups,
                            float* partial, int chunks, vda_stream_t stream) {
    VDA_REQUIRE(in && out && w && b && partial, "vda_groupnorm: null pointer");
    VDA_REQUIRE(frames > 0 && hw > 0 && C > 0 && groups > 0 && groups <= 64 && C % groups == 0 && C % 8 == 0 && C <= 1024,
                "vda_groupnorm: bad geometry C=%d groups=%d", C, groups);
    VDA_REQUIRE(chunks > 0 && chunks <= hw, "vda_groupnorm: chunks=%d out of range", chunks);
    VDA_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, "vda_groupnorm: 16-byte alignment required");
    const int nv = C / 8, rows_par = 256 / nv;
    const size_t smem = ((size_t)rows_par * C * 2 + (size_t)C * 2) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((groupnorm_partial_kernel<T>), dim3(chunks, frames), dim3(256), smem, s, in, partial, hw, C, groups, chunks);
    VDA_LAUNCH_CHECK();
    const int rows_per_block = 32;
    hipLaunchKernelGGL((groupnorm_apply_kernel<T>), dim3((hw + rows_per_block - 1) / rows_per_block, frames), dim3(256), 0, s, in, out, w,
                       b, partial, eps, hw, C, groups, chunks, rows_per_block);
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_groupnorm_nhwc_f16(const void* in, void* out, const float* w, const float* b, float eps, int frames, int hw,
                                      int C, int groups, float* partial, int chunks, vda_stream_t stream) {
    return groupnorm_launch<h16>((const h16*)in, (h16*)out, w, b, eps, frames, hw, C, groups, partial, chunks, stream);
}

extern "C" int vda_groupnorm_nhwc_f32(const float* in, float* out, const float* w, const float* b, float eps, int frames, int hw,
                                      int C, int groups, float* partial, int chunks, vda_stream_t stream) {
    return groupnorm_launch<float>(in, out, w, b, eps, frames, hw, C, groups, partial, chunks, stream);
}
