// Spatial self-attention of the ViT encoder for gfx950: softmax(q k^T / 8) v, head dim 64,
// fp16 in / fp16 out, fp32 scores, softmax and accumulation; the N x N score matrix never
// leaves registers (online softmax over 64-key tiles).
//
// Work split: one workgroup = 128 queries of one (frame, head); 4 waves x 32 queries.
// Per 64-key tile and wave:
//   S^T[key][query] = K_tile . Q^T    v_mfma_f32_32x32x16_f16, K rows from LDS (A operand),
//                                     Q fragments held in registers for the whole kernel (B operand).
//     -> a lane owns ONE query (lane & 31) and 32 of the tile's 64 keys in registers; its partner
//        lane ^ 32 owns the other 32, so row max / row sum are in-lane plus one cross-lane exchange.
//   O^T[ch][query] += V_tile^T . P^T  the exponentiated accumulators, converted to fp16 in place, ARE
//                                     the B operand (k order permuted: element j of lane half h of
//                                     16-key step s is key 16s + 8(j>>2) + 4h + (j&3)); V^T comes from
//                                     the row-major V tile with ds_read_b64_tr_b16 in the same k order.
//     -> the output accumulator also has the query on the lane, so the online-softmax rescale is a
//        plain per-lane multiply.
// K and V tiles arrive by 16-byte global_load_lds into a double buffer; XOR swizzles are applied on
// the source address (the DMA image is lane-linear) and again on the LDS read.
#include "vda_common.h"

namespace {

constexpr int HD = 64;       // head dim (both ViT-S and ViT-L)
constexpr int BQ = 128;      // queries per workgroup
constexpr int BKV = 64;      // keys per tile
constexpr int TILE_BYTES = BKV * HD * 2;          // 8 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;       // K + V

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ int k_swz(int row) { return (row >> 1) & 7; }          // 16-row conflict-free for 32-row b128 fragments
__device__ __forceinline__ int v_swz(int row) { return ((row >> 1) & 1) << 2; }   // separates the 4 rows of a tr16 block

// MB ("max through the matrix pipe"): Q is pre-scaled by log2(e)/8 and the running reference point m of the online softmax is fed
// to the score MFMAs as a fifth k-step (K side: a constant [1, 1, 0, ...] row; Q side: [-m_hi, -m_lo, 0, ...], m = m_hi + m_lo in
// fp16 pairs): the accumulators come out as log2-domain scores MINUS the reference, and p = exp2(acc) needs no per-score fma
// (31 of the ~155 VALU issues per 64-key tile; the loop is VALU-bound). Any reference point gives the same softmax; m only has
// to stay within rounding of the running maximum, which its fp16 pair does (22 bits).
template <bool TR, bool PK, bool LSUM = false, bool MB = false, int ABL = 0>   // ABL: timing ablations (tools/attn_one.py), wrong results
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) attn_kernel(const h16* __restrict__ qkv, h16* __restrict__ out, int N, int H,
                                                   int nqb, int total_blocks) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware remap: blocks sharing an XCD (bid % 8) get whole (frame, head) groups so K/V stay in that L2.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, qd = total_blocks >> 3, rm = total_blocks & 7;
    const int t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int bh = t / nqb, qb = t - bh * nqb;
    const int b = bh / H, head = bh - b * H;

    const size_t rs = (size_t)3 * H * HD;                       // row stride of qkv in halves
    const unsigned rs32 = (unsigned)rs;                         // one frame's N * rs fits 32 bits (checked by the launcher)
    const h16* Qb = qkv + (size_t)b * N * rs + head * HD;
    const h16* Kb = Qb + (size_t)H * HD;
    const h16* Vb = Kb + (size_t)H * HD;

    // ---- Q fragments (B operand: lane holds Q[query r][16ks + 8h .. +7]), pre-scaled by 1/8 (exact)
    const int q_row = qb * BQ + wave * 32 + r;
    const bool wave_active = __builtin_amdgcn_readfirstlane((int)(qb * BQ + wave * 32 < N)) != 0;
    h16x8 qf[4];
    {
        const h16* qp = Qb + (size_t)min(q_row, N - 1) * rs + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const h16x8*>(qp + ks * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = MB ? (h16)((float)qf[ks][e] * (0.125f * 1.4426950408889634f)) : qf[ks][e] * (h16)0.125f;
        }
    }

    // ---- DMA sources: each wave moves 2 K pieces and 2 V pieces (8 rows x 128 B each) per tile
    const int lrow = lane >> 3, lpos = lane & 7;
    auto stage = [&](int kt, char* buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int piece = wave + 4 * j;
            const int row = piece * 8 + lrow;
            const unsigned key = (unsigned)min(kt * BKV + row, N - 1) * rs32;     // 32-bit: one multiply, no 64-bit carry chain
            glds16(Kb + (key + ((lpos ^ k_swz(row)) << 3)), buf + piece * 1024);
            glds16(Vb + (key + ((lpos ^ v_swz(row)) << 3)), buf + TILE_BYTES + piece * 1024);
        }
    };

    f32x16 acc_o[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_o[c][e] = 0.f;
    float m_run = MB ? 0.f : -1e30f, l_run = 0.f;
    // MB: the fifth k-step's operands. K side: k' = 0, 1 are one (lanes of half h = 0 hold k' = 0..7); Q side: -m as an fp16 pair.
    h16x8 kone, mneg;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        kone[e] = (h16)((h == 0 && e < 2) ? 1.f : 0.f);
        mneg[e] = (h16)0.f;
    }
    // LSUM: the softmax denominator comes out of the matrix pipe - one extra MFMA per 16-key step with an all-ones A operand
    // sums the fp16 P the numerator uses - instead of 32 v_add_f32 per tile on the (binding) VALU.
    f32x16 acc_l;
    h16x8 ones;
    if constexpr (LSUM) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_l[e] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (h16)1.f;
    }
    constexpr float LOG2E = 1.4426950408889634f;

    const int nt = (N + BKV - 1) / BKV;
    stage(0, smem);
    int cur = 0;
    for (int kt = 0; kt < nt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nt) stage(kt + 1, smem + (cur ^ 1) * STAGE_BYTES);
        const char* Kt = smem + cur * STAGE_BYTES;
        const char* Vt = Kt + TILE_BYTES;
        // A wave whose 32 queries all lie past N (the 4th wave of a frame's last query block: 1370 = 10 x 128 + 90) only helps
        // with the staging and the barriers: its SIMD is left to the other workgroups' waves.
        if (!wave_active) {
            cur ^= 1;
            continue;
        }

        // ---- S^T = K . Q^T for the tile's two 32-key halves
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[sub][e] = 0.f;
            const int row = sub * 32 + r;
            const char* kp = Kt + row * 128;
            const int sw = k_swz(row);
#pragma unroll
            for (int ks = 0; ks < (ABL == 5 ? 1 : 4); ++ks) {
                const h16x8 kf = *reinterpret_cast<const h16x8*>(kp + (((2 * ks + h) ^ sw) << 4));
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[sub], 0, 0, 0);
            }
            if constexpr (MB) s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kone, mneg, s[sub], 0, 0, 0);
        }
        // accumulator register e of half `sub` is key  kt*64 + sub*32 + (e&3) + 8*(e>>2) + 4h
        if (__builtin_amdgcn_readfirstlane((int)(kt == nt - 1 && (N % BKV) != 0))) {      // scalar branch, last tile only
            const int kbase = kt * BKV + 4 * h;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (kbase + sub * 32 + (e & 3) + 8 * (e >> 2) >= N) s[sub][e] = -1e30f;
        }

        // ---- online softmax (per query = per lane pair {lane, lane^32})
        float mx = s[0][0];
        if constexpr (ABL != 2) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[sub][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        }
        if constexpr (MB) {
            // the accumulators are scores minus the reference m_run (log2 domain): nothing to do while no score exceeds it
            if (__builtin_amdgcn_readfirstlane((int)(kt == 0 || __ballot(mx > 0.f) != 0ull))) {
                // new reference = the running maximum rounded to an fp16 pair (lanes whose maximum did not move keep theirs: delta = 0)
                const float m_want = kt == 0 ? mx : m_run + fmaxf(mx, 0.f);
                const h16 mh = (h16)m_want, ml = (h16)(m_want - (float)mh);
                const float m_new = (float)mh + (float)ml;
                const float delta = m_new - m_run;                     // exact: both are 22-bit values of similar magnitude
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                l_run *= alpha;
                if constexpr (LSUM) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc_l[e] *= alpha;
                }
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc_o[c][e] *= alpha;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[sub][e] -= delta;
                m_run = m_new;
                mneg[0] = h == 0 ? -mh : (h16)0.f;
                mneg[1] = h == 0 ? -ml : (h16)0.f;
            }
        } else {
        const float m_new = fmaxf(m_run, mx);
        // Rescale only when some query's running max moved (alpha == 1 exactly otherwise, so skipping is bit-exact).
        if (__builtin_amdgcn_readfirstlane((int)(__ballot(m_new > m_run) != 0ull))) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
            l_run *= alpha;
            if constexpr (LSUM) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc_l[e] *= alpha;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc_o[c][e] *= alpha;
            m_run = m_new;
        }
        }
        h16x8 pf[4];
        if constexpr (PK) {
            // p = exp2(s*log2e - m*log2e) on float pairs (v_pk_fma_f32 / v_pk_add_f32); raw v_exp_f32: the argument
            // is <= 0 and flushing tiny results to zero is harmless.
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 mb2 = {m_run * LOG2E, m_run * LOG2E};
            const f32x2 l2e = {LOG2E, LOG2E};
            f32x2 ps2 = {0.f, 0.f};
    #pragma unroll
            for (int sub = 0; sub < 2; ++sub)
    #pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    f32x2 a = {s[sub][e], s[sub][e + 1]};
                    a = a * l2e - mb2;
                    f32x2 pv = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
                    ps2 += pv;
                    pf[sub * 2 + (e >> 3)][e & 7] = (h16)pv[0];
                    pf[sub * 2 + (e >> 3)][(e & 7) + 1] = (h16)pv[1];
                }
            l_run += ps2[0] + ps2[1];
        } else {
            // scalar form of the same arithmetic (A/B: v_pk_*_f32 vs two scalar issues)
            const float mb = m_run * LOG2E;
            float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float arg = MB ? s[sub][e] : fmaf(s[sub][e], LOG2E, -mb);
                    const float pv = ABL == 1 ? arg : __builtin_amdgcn_exp2f(arg);
                    if constexpr (!LSUM && ABL != 3) ps[e & 3] += pv;
                    pf[sub * 2 + (e >> 3)][e & 7] = (h16)pv;
                }
            if constexpr (!LSUM) l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        }

        // ---- O^T += V^T . P^T  (4 steps of 16 keys, 2 halves of 32 channels)
#pragma unroll
        for (int kstep = 0; kstep < 4; ++kstep) {
            if constexpr (LSUM) acc_l = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf[kstep], acc_l, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                h16x8 vf;
                if constexpr (TR) {
                    // 16-lane group reads a 4-key x 16-channel block; lane 4q+p addresses key q, channels 4p..4p+3.
                    const int i = lane & 15, qq = i >> 2, pp = i & 3;
                    const int col = c * 32 + 16 * ((lane >> 4) & 1) + 4 * pp;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int key = kstep * 16 + half * 8 + 4 * h + qq;
                        const char* ap = Vt + key * 128 + ((((col >> 3) ^ v_swz(key))) << 4) + ((col & 7) << 1);
                        const fp16x4_t v4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((VDA_LDS_AS fp16x4_t*)ap);
#pragma unroll
                        for (int e = 0; e < 4; ++e) vf[half * 4 + e] = (h16)v4[e];
                    }
                } else {
                    const int ch = c * 32 + r;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int key = kstep * 16 + 8 * (j >> 2) + 4 * h + (j & 3);
                        vf[j] = *reinterpret_cast<const h16*>(Vt + key * 128 + ((((ch >> 3) ^ v_swz(key))) << 4) + ((ch & 7) << 1));
                    }
                }
                if constexpr (ABL == 4) acc_o[c][kstep] += (float)vf[0] * (float)pf[kstep][c];
                else acc_o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kstep], acc_o[c], 0, 0, 0);
            }
        }
        cur ^= 1;
    }

    // ---- normalise and store: lane holds query r, channels c*32 + (e&3) + 8*(e>>2) + 4h
    const float l_tot = LSUM ? acc_l[0] : l_run + __shfl_xor(l_run, 32, 64);   // every row of acc_l holds the query's full sum
    const float inv = 1.0f / l_tot;
    if (q_row < N) {
        h16* op = out + ((size_t)b * N + q_row) * ((size_t)H * HD) + head * HD + 4 * h;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h16x4 o = {(h16)(acc_o[c][4 * g + 0] * inv), (h16)(acc_o[c][4 * g + 1] * inv),
                           (h16)(acc_o[c][4 * g + 2] * inv), (h16)(acc_o[c][4 * g + 3] * inv)};
                *reinterpret_cast<h16x4*>(op + c * 32 + 8 * g) = o;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Software-pipelined form (variant 7): the score MFMAs of key tile t+1 are issued inside the softmax of tile t, so a wave has
// matrix work and VALU work in the same stretch of its instruction stream instead of alternating between the two (the
// ablations in DESIGN.md: the loop runs as MFMA time PLUS softmax time). Costs a second set of score accumulators (32 VGPRs:
// three waves per SIMD instead of four) and a third K slot (K two tiles ahead, V one): 40 KiB of LDS per workgroup.
// Same instructions on the same operands in the same order per query as attn_kernel<true, false>: bit-identical results.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) attn_pipe_kernel(const h16* __restrict__ qkv, h16* __restrict__ out, int N,
                                                                                              int H, int nqb, int total_blocks) {
    __shared__ __attribute__((aligned(16))) char smem[5 * TILE_BYTES];          // K0 K1 K2 | V0 V1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, qd = total_blocks >> 3, rm = total_blocks & 7;
    const int t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int bh = t / nqb, qb = t - bh * nqb;
    const int b = bh / H, head = bh - b * H;
    const size_t rs = (size_t)3 * H * HD;
    const unsigned rs32 = (unsigned)rs;
    const h16* Qb = qkv + (size_t)b * N * rs + head * HD;
    const h16* Kb = Qb + (size_t)H * HD;
    const h16* Vb = Kb + (size_t)H * HD;
    const int q_row = qb * BQ + wave * 32 + r;
    const bool wave_active = __builtin_amdgcn_readfirstlane((int)(qb * BQ + wave * 32 < N)) != 0;
    h16x8 qf[4];
    {
        const h16* qp = Qb + (size_t)min(q_row, N - 1) * rs + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const h16x8*>(qp + ks * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = qf[ks][e] * (h16)0.125f;
        }
    }
    const int lrow = lane >> 3, lpos = lane & 7;
    auto stage_k = [&](int kt, char* buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int piece = wave + 4 * j, row = piece * 8 + lrow;
            const unsigned key = (unsigned)min(kt * BKV + row, N - 1) * rs32;
            glds16(Kb + (key + ((lpos ^ k_swz(row)) << 3)), buf + piece * 1024);
        }
    };
    auto stage_v = [&](int kt, char* buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int piece = wave + 4 * j, row = piece * 8 + lrow;
            const unsigned key = (unsigned)min(kt * BKV + row, N - 1) * rs32;
            glds16(Vb + (key + ((lpos ^ v_swz(row)) << 3)), buf + piece * 1024);
        }
    };
    auto qk = [&](const char* Kt, f32x16 (&s)[2]) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[sub][e] = 0.f;
            const int row = sub * 32 + r;
            const char* kp = Kt + row * 128;
            const int sw = k_swz(row);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h16x8 kf = *reinterpret_cast<const h16x8*>(kp + (((2 * ks + h) ^ sw) << 4));
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[sub], 0, 0, 0);
            }
        }
    };

    f32x16 acc_o[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_o[c][e] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    constexpr float LOG2E = 1.4426950408889634f;
    const int nt = (N + BKV - 1) / BKV;
    char* const Ks = smem;
    char* const Vs = smem + 3 * TILE_BYTES;
    stage_k(0, Ks);
    stage_v(0, Vs);
    if (nt > 1) stage_k(1, Ks + TILE_BYTES);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 sc[2];
    if (wave_active) qk(Ks, sc);
    else {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[sub][e] = 0.f;
    }
    int kn = 1;                                             // K slot of tile kt + 1
    for (int kt = 0; kt < nt; ++kt) {
        if (kt > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // K(kt+1), V(kt) landed (issued an iteration ago)
            __syncthreads();                                      // ... for every wave; and everyone is done with tile kt-1's slots
        }
        const int k2 = kn == 2 ? 0 : kn + 1;                       // slot of K(kt+2) = the one K(kt-1) had
        if (kt + 2 < nt) stage_k(kt + 2, Ks + k2 * TILE_BYTES);
        if (kt + 1 < nt) stage_v(kt + 1, Vs + ((kt + 1) & 1) * TILE_BYTES);
        const char* Kn = Ks + kn * TILE_BYTES;
        const char* Vt = Vs + (kt & 1) * TILE_BYTES;
        kn = k2;
        if (!wave_active) continue;
        if (__builtin_amdgcn_readfirstlane((int)(kt == nt - 1 && (N % BKV) != 0))) {
            const int kbase = kt * BKV + 4 * h;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (kbase + sub * 32 + (e & 3) + 8 * (e >> 2) >= N) sc[sub][e] = -1e30f;
        }
        float mx = sc[0][0];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sc[sub][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        if (__builtin_amdgcn_readfirstlane((int)(__ballot(m_new > m_run) != 0ull))) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
            l_run *= alpha;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc_o[c][e] *= alpha;
            m_run = m_new;
        }
        // ---- one stretch: scores of tile kt+1 (matrix pipe) | exp / sums / conversion of tile kt (VALU) | P.V of tile kt
        // (after the last tile the next-tile MFMAs run on a stale K slot and their result is dropped: no branch, one basic block)
        f32x16 sn[2];
        qk(Kn, sn);
        h16x8 pf[4];
        const float mb = m_run * LOG2E;
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sc[sub][e], LOG2E, -mb));
                ps[e & 3] += pv;
                pf[sub * 2 + (e >> 3)][e & 7] = (h16)pv;
            }
        l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
        for (int kstep = 0; kstep < 4; ++kstep) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                h16x8 vf;
                const int i = lane & 15, qq = i >> 2, pp = i & 3;
                const int col = c * 32 + 16 * ((lane >> 4) & 1) + 4 * pp;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int key = kstep * 16 + half * 8 + 4 * h + qq;
                    const char* ap = Vt + key * 128 + ((((col >> 3) ^ v_swz(key))) << 4) + ((col & 7) << 1);
                    const fp16x4_t v4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((VDA_LDS_AS fp16x4_t*)ap);
#pragma unroll
                    for (int e = 0; e < 4; ++e) vf[half * 4 + e] = (h16)v4[e];
                }
                acc_o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kstep], acc_o[c], 0, 0, 0);
            }
        }
        // issue order of the stretch: every matrix instruction followed by a slice of the softmax's VALU / transcendental work
        // (16 MFMAs, ~150 VALU): the 8 score MFMAs of the next tile first (their operands are ready), the 8 P.V ones as their P slices
        // complete
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x402, 9, 0);       // nine VALU / transcendental
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // two LDS reads
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) sc[sub] = sn[sub];
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < N) {
        h16* op = out + ((size_t)b * N + q_row) * ((size_t)H * HD) + head * HD + 4 * h;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h16x4 o = {(h16)(acc_o[c][4 * g + 0] * inv), (h16)(acc_o[c][4 * g + 1] * inv),
                           (h16)(acc_o[c][4 * g + 2] * inv), (h16)(acc_o[c][4 * g + 3] * inv)};
                *reinterpret_cast<h16x4*>(op + c * 32 + 8 * g) = o;
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 3 (variants 8 / 9, the default): the online softmax with the reference point in the MATRIX pipe and a lazy rescale.
//
// attn_kernel spends ~185 VALU issues per 64-key tile and wave (32 exp, 31 fma for "score * log2e - max * log2e", 37 adds,
// 16 max3, 16 cvt_pk, and - on 85 % of the tiles, because SOME query of the wave's 32 sees a new maximum - 32 multiplies of the
// output accumulators) against 512 cycles of MFMA: it is VALU-bound. Two changes remove ~80 of them:
//   * Q is pre-scaled by log2(e) / 8 and the score MFMA chains START from a register block holding -m (the query's reference
//     point, log2 domain; a lane owns ONE query, so the block is the same value in all 16 registers): D = K.Q^T + (-m) comes
//     out of the matrix pipe ready for v_exp_f32. No fma per score, no extra MFMA (variants 4 / 5 paid two MFMAs per tile for
//     the same effect and lost).
//   * The reference point only has to bound the scores, not equal their maximum: it is moved - per lane, by the amount that
//     lane's maximum exceeds it - only when a score exceeds it by more than LAZY (log2 units), so p <= 2^LAZY (exact in fp16:
//     the format is floating, fp32 sums). After the first tiles no wave rescales any more; a lane that is not moved multiplies
//     by exp2(0) = 1 exactly, so a query's result does not depend on its neighbours in the wave.
// Same tile loop, LDS image, fragment shapes and PV product as attn_kernel.
template <int LAZY>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) attn_cm_kernel(const h16* __restrict__ qkv, h16* __restrict__ out, int N,
                                                                                            int H, int nqb, int total_blocks) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, qd = total_blocks >> 3, rm = total_blocks & 7;
    const int t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int bh = t / nqb, qb = t - bh * nqb;
    const int b = bh / H, head = bh - b * H;
    const size_t rs = (size_t)3 * H * HD;
    const unsigned rs32 = (unsigned)rs;
    const h16* Qb = qkv + (size_t)b * N * rs + head * HD;
    const h16* Kb = Qb + (size_t)H * HD;
    const h16* Vb = Kb + (size_t)H * HD;

    // Q fragments, pre-scaled by log2(e) / 8: the scores leave the MFMAs in the log2 domain
    const int q_row = qb * BQ + wave * 32 + r;
    const bool wave_active = __builtin_amdgcn_readfirstlane((int)(qb * BQ + wave * 32 < N)) != 0;
    h16x8 qf[4];
    {
        const h16* qp = Qb + (size_t)min(q_row, N - 1) * rs + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const h16x8*>(qp + ks * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = (h16)((float)qf[ks][e] * (0.125f * 1.4426950408889634f));
        }
    }
    const int lrow = lane >> 3, lpos = lane & 7;
    auto stage = [&](int kt, char* buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int piece = wave + 4 * j;
            const int row = piece * 8 + lrow;
            const unsigned key = (unsigned)min(kt * BKV + row, N - 1) * rs32;
            glds16(Kb + (key + ((lpos ^ k_swz(row)) << 3)), buf + piece * 1024);
            glds16(Vb + (key + ((lpos ^ v_swz(row)) << 3)), buf + TILE_BYTES + piece * 1024);
        }
    };

    f32x16 acc_o[2], cneg;                      // cneg: -m in every register = the C operand of each score chain's first MFMA
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        acc_o[0][e] = 0.f;
        acc_o[1][e] = 0.f;
        cneg[e] = 0.f;
    }
    float l_run = 0.f;
    const int nt = (N + BKV - 1) / BKV;
    stage(0, smem);
    int cur = 0;
    for (int kt = 0; kt < nt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nt) stage(kt + 1, smem + (cur ^ 1) * STAGE_BYTES);
        const char* Kt = smem + cur * STAGE_BYTES;
        const char* Vt = Kt + TILE_BYTES;
        cur ^= 1;
        if (!wave_active) continue;             // (a wave whose 32 queries lie past N only stages and synchronises)

        // ---- S^T - m = K . Q^T + (-m): two 32-key halves, four 16-channel k-steps each
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int row = sub * 32 + r;
            const char* kp = Kt + row * 128;
            const int sw = k_swz(row);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h16x8 kf = *reinterpret_cast<const h16x8*>(kp + (((2 * ks + h) ^ sw) << 4));
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? cneg : s[sub], 0, 0, 0);
            }
        }
        if (__builtin_amdgcn_readfirstlane((int)(kt == nt - 1 && (N % BKV) != 0))) {      // scalar branch, last tile only
            const int kbase = kt * BKV + 4 * h;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (kbase + sub * 32 + (e & 3) + 8 * (e >> 2) >= N) s[sub][e] = -1e30f;
        }
        // ---- the tile's maximum per query (lane pair {lane, lane ^ 32}), relative to the reference point
        float mx = s[0][0];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[sub][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const bool move = kt == 0 || mx > (float)LAZY;             // first tile: the reference point becomes the tile's maximum
        if (__builtin_amdgcn_readfirstlane((int)(__ballot(move) != 0ull))) {
            const float delta = move ? mx : 0.f;                    // lanes that stay: alpha = exp2(-0) = 1 exactly, all no-ops
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            l_run *= alpha;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc_o[c][e] *= alpha;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[sub][e] -= delta;
#pragma unroll
            for (int e = 0; e < 16; ++e) cneg[e] -= delta;
        }
        // ---- p = exp2(s - m) straight from the accumulators, row sums, fp16 fragments of P^T
        h16x8 pf[4];
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pv = __builtin_amdgcn_exp2f(s[sub][e]);
                ps[e & 3] += pv;
                pf[sub * 2 + (e >> 3)][e & 7] = (h16)pv;
            }
        l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);

        // ---- O^T += V^T . P^T  (4 steps of 16 keys, 2 halves of 32 channels)
#pragma unroll
        for (int kstep = 0; kstep < 4; ++kstep) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                h16x8 vf;
                const int i = lane & 15, qq = i >> 2, pp = i & 3;
                const int col = c * 32 + 16 * ((lane >> 4) & 1) + 4 * pp;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int key = kstep * 16 + half * 8 + 4 * h + qq;
                    const char* ap = Vt + key * 128 + ((((col >> 3) ^ v_swz(key))) << 4) + ((col & 7) << 1);
                    const fp16x4_t v4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((VDA_LDS_AS fp16x4_t*)ap);
#pragma unroll
                    for (int e = 0; e < 4; ++e) vf[half * 4 + e] = (h16)v4[e];
                }
                acc_o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kstep], acc_o[c], 0, 0, 0);
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < N) {
        h16* op = out + ((size_t)b * N + q_row) * ((size_t)H * HD) + head * HD + 4 * h;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h16x4 o = {(h16)(acc_o[c][4 * g + 0] * inv), (h16)(acc_o[c][4 * g + 1] * inv),
                           (h16)(acc_o[c][4 * g + 2] * inv), (h16)(acc_o[c][4 * g + 3] * inv)};
                *reinterpret_cast<h16x4*>(op + c * 32 + 8 * g) = o;
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Variant 10: attn_cm_kernel with two more groups of VALU instructions taken out of the tile loop (PMC on variant 9: VALU issue
// 82 % of SIMD time, MFMA busy 44 %, 10.8 VALU per MFMA - the loop is bound by VALU ISSUE, not by dependency stalls):
//   * K / V staging by bounds-checked `buffer_load ... lds` with per-lane constant offsets and ONE scalar per tile (the tile's
//     row offset in soffset) instead of four 64-bit global addresses per lane and tile (~25 VALU); rows past N are out of the
//     descriptor's range and arrive as zeros (their scores are masked on the last tile, their V rows meet p = 0).
//   * no running maximum in the hot path (18 max3 + exchange + compare): whether the reference point must move is read off the
//     row sum the loop computes anyway - p >= 0, so max p <= sum p, and sum p <= 2^11 proves that no p came near fp16's range.
//     A wave whose sums say otherwise (spiked scores; always the first tile, whose reference point is 0) takes the slow path:
//     the scores are formed again from the K tile still in LDS, their maximum moves the reference point exactly as in
//     attn_cm_kernel, and the tile's p is recomputed. Decided per lane pair (one query): neighbours never change a row's result.
//
// FLAGS (variant 11, A/B): the per-tile workgroup barrier replaced by two pairs of LDS counters. A wave counts itself in on
// full[b] when its own pieces of the tile in buffer b have landed and on empty[b] when it has finished reading that buffer; it
// waits on full[b] before reading and on empty[b ^ 1] before re-staging. The four waves of a workgroup sit on four different
// SIMDs, each shared with three other workgroups' waves, so they drift apart: with s_barrier every tile ends at the slowest wave
// (PMC: 29 % of wave-cycles parked), with counters a wave may run up to a tile ahead of the slowest.
template <bool FLAGS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) attn_cs_kernel(const h16* __restrict__ qkv, h16* __restrict__ out, int N, int H,
                                                                                            int nqb, int total_blocks) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES + (FLAGS ? 16 : 0)];
    volatile VDA_LDS_AS int* const flags = (volatile VDA_LDS_AS int*)(smem + 2 * STAGE_BYTES);      // full[0], full[1], empty[0], empty[1]
    if constexpr (FLAGS) {
        if (threadIdx.x < 4) flags[threadIdx.x] = 0;
        __syncthreads();
    }
    auto count_in = [&](int idx) {                  // one lane per wave: this wave is through
        if constexpr (FLAGS) {
            if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add((VDA_LDS_AS int*)(flags + idx), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto wait_for = [&](int idx, int target) {      // all four waves are through (counters only grow)
        if constexpr (FLAGS) {
            while (__builtin_amdgcn_readfirstlane(flags[idx]) < target) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
    };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, qd = total_blocks >> 3, rm = total_blocks & 7;
    const int t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int bh = t / nqb, qb = t - bh * nqb;
    const int b = bh / H, head = bh - b * H;
    const size_t rs = (size_t)3 * H * HD;
    const unsigned rs32 = (unsigned)rs;
    const h16* Qb = qkv + (size_t)b * N * rs + head * HD;
    const h16* Kb = Qb + (size_t)H * HD;
    const h16* Vb = Kb + (size_t)H * HD;

    const int q_row = qb * BQ + wave * 32 + r;
    const bool wave_active = __builtin_amdgcn_readfirstlane((int)(qb * BQ + wave * 32 < N)) != 0;
    h16x8 qf[4];
    {
        const h16* qp = Qb + (size_t)min(q_row, N - 1) * rs + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const h16x8*>(qp + ks * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = (h16)((float)qf[ks][e] * (0.125f * 1.4426950408889634f));
        }
    }
    // staging: descriptor = the frame's K (V) rows of this head, N rows of 64 halves at stride rs; per-lane byte offsets of the
    // wave's two pieces (rows piece * 8 + lrow, swizzled 16-byte chunk), constant for the whole kernel
    const unsigned frame_bytes = ((unsigned)(N - 1) * rs32 + HD) * 2u;            // first byte past row N - 1's 64 halves
    const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(Kb), 0, frame_bytes, 0x00020000);
    const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(Vb), 0, frame_bytes, 0x00020000);
    const int lrow = lane >> 3, lpos = lane & 7;
    unsigned koff[2], voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave + 4 * j) * 8 + lrow;
        koff[j] = ((unsigned)row * rs32 + (unsigned)((lpos ^ k_swz(row)) << 3)) * 2u;
        voff[j] = ((unsigned)row * rs32 + (unsigned)((lpos ^ v_swz(row)) << 3)) * 2u;
    }
    const unsigned tile_stride = (unsigned)BKV * rs32 * 2u;                         // bytes between key tiles
    auto stage = [&](int kt, char* buf) {
        const unsigned so = (unsigned)kt * tile_stride;                             // scalar
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int piece = wave + 4 * j;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (VDA_LDS_AS void*)(buf + piece * 1024), 16, koff[j], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (VDA_LDS_AS void*)(buf + TILE_BYTES + piece * 1024), 16, voff[j], so, 0, 0);
        }
    };

    f32x16 acc_o[2], cneg;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        acc_o[0][e] = 0.f;
        acc_o[1][e] = 0.f;
        cneg[e] = 0.f;
    }
    float l_run = 0.f;
    constexpr float SUM_LIMIT = 2048.f;             // 2^11: every p of the tile is below it when the row sum is
    const int nt = (N + BKV - 1) / BKV;
    stage(0, smem);
    int cur = 0;
    for (int kt = 0; kt < nt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (FLAGS) {
            const int use = (kt >> 1) + 1;                               // how many tiles buffer `cur` has held, this one included
            count_in(cur);                                               // my pieces of tile kt are in buffer cur
            if (kt + 1 < nt) {
                if (kt >= 1) wait_for(2 + (cur ^ 1), 4 * ((kt - 1) / 2 + 1));    // everyone has finished reading tile kt - 1
                stage(kt + 1, smem + (cur ^ 1) * STAGE_BYTES);
            }
            wait_for(cur, 4 * use);                                      // everyone's pieces of tile kt have landed
        } else {
            __syncthreads();
            if (kt + 1 < nt) stage(kt + 1, smem + (cur ^ 1) * STAGE_BYTES);
        }
        const char* Kt = smem + cur * STAGE_BYTES;
        const char* Vt = Kt + TILE_BYTES;
        const int done_idx = 2 + cur;
        cur ^= 1;
        if (!wave_active) {
            count_in(done_idx);                                          // (reads nothing)
            continue;
        }

        const bool last_partial = __builtin_amdgcn_readfirstlane((int)(kt == nt - 1 && (N % BKV) != 0)) != 0;
        // scores relative to the reference point: S^T - m = K . Q^T + (-m); keys past N masked
        auto scores = [&](f32x16 (&s)[2]) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int row = sub * 32 + r;
                const char* kp = Kt + row * 128;
                const int sw = k_swz(row);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const h16x8 kf = *reinterpret_cast<const h16x8*>(kp + (((2 * ks + h) ^ sw) << 4));
                    s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? cneg : s[sub], 0, 0, 0);
                }
            }
            if (last_partial) {
                const int kbase = kt * BKV + 4 * h;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kbase + sub * 32 + (e & 3) + 8 * (e >> 2) >= N) s[sub][e] = -1e30f;
            }
        };
        // p = exp2(s), fp16 fragments of P^T, the lane's sum over its 32 keys
        h16x8 pf[4];
        auto softmax = [&](const f32x16 (&s)[2]) -> float {
            float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(s[sub][e]);
                    ps[e & 3] += pv;
                    pf[sub * 2 + (e >> 3)][e & 7] = (h16)pv;
                }
            return (ps[0] + ps[1]) + (ps[2] + ps[3]);
        };
        float tsum = 0.f;
        bool trig = true;                           // first tile: the reference point (0) is not one yet
        if (kt > 0) {
            f32x16 s[2];
            scores(s);
            tsum = softmax(s);
            const float qsum = tsum + __shfl_xor(tsum, 32, 64);          // the query's sum over the tile's 64 keys
            trig = !(qsum <= SUM_LIMIT);                                  // (also true for inf / nan)
        }
        if (__builtin_amdgcn_readfirstlane((int)(__ballot(trig) != 0ull))) {
            // slow path: scores again (the K tile is still in LDS), their maximum moves the reference point of the lanes that asked
            f32x16 s[2];
            scores(s);
            float mx = s[0][0];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[sub][e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float delta = trig ? mx : 0.f;                          // lanes that stay: exp2(-0) = 1 exactly, all no-ops
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            l_run *= alpha;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc_o[c][e] *= alpha;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[sub][e] -= delta;
#pragma unroll
            for (int e = 0; e < 16; ++e) cneg[e] -= delta;
            tsum = softmax(s);
        }
        l_run += tsum;

        // ---- O^T += V^T . P^T  (4 steps of 16 keys, 2 halves of 32 channels)
#pragma unroll
        for (int kstep = 0; kstep < 4; ++kstep) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                h16x8 vf;
                const int i = lane & 15, qq = i >> 2, pp = i & 3;
                const int col = c * 32 + 16 * ((lane >> 4) & 1) + 4 * pp;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int key = kstep * 16 + half * 8 + 4 * h + qq;
                    const char* ap = Vt + key * 128 + ((((col >> 3) ^ v_swz(key))) << 4) + ((col & 7) << 1);
                    const fp16x4_t v4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((VDA_LDS_AS fp16x4_t*)ap);
#pragma unroll
                    for (int e = 0; e < 4; ++e) vf[half * 4 + e] = (h16)v4[e];
                }
                acc_o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kstep], acc_o[c], 0, 0, 0);
            }
        }
        if constexpr (FLAGS) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // my reads of the buffer are done (LDS returns in order)
            count_in(done_idx);
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < N) {
        h16* op = out + ((size_t)b * N + q_row) * ((size_t)H * HD) + head * HD + 4 * h;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h16x4 o = {(h16)(acc_o[c][4 * g + 0] * inv), (h16)(acc_o[c][4 * g + 1] * inv),
                           (h16)(acc_o[c][4 * g + 2] * inv), (h16)(acc_o[c][4 * g + 3] * inv)};
                *reinterpret_cast<h16x4*>(op + c * 32 + 8 * g) = o;
            }
    }
}

}  // namespace

// -1 (default): the kernel picked below. 10: attn_cs_kernel (as 9, staging by scalar-offset buffer loads, no maximum in the hot path);
// 8 / 9: attn_cm_kernel (reference point through the MFMA C operand; 9 = lazy rescale, 2^6);
// 1: attn_kernel with ds_read_b64_tr_b16 V fragments + scalar softmax math; 2: the same with v_pk_*_f32 softmax math; 0: scalar LDS
// reads of V (debug cross-check); 3 / 4 / 5 / 7: the round-2 experiments (row sums / running max through the matrix pipe, the
// software-pipelined form); 11..15: timing ablations of attn_kernel (wrong results).
constexpr int VDA_ATTN_DEFAULT = 10;
static int g_attn_variant = -1;

extern "C" int vda_attention_set_variant(int v) {
    g_attn_variant = v;
    return 0;
}

extern "C" int vda_attention_f16(const void* qkv, void* out, int B, int N, int heads, vda_stream_t stream) {
    VDA_REQUIRE(qkv && out, "vda_attention_f16: null pointer");
    VDA_REQUIRE(B > 0 && N > 0 && heads > 0, "vda_attention_f16: empty problem B=%d N=%d heads=%d", B, N, heads);
    VDA_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "vda_attention_f16: 16-byte alignment required");
    const int nqb = (N + BQ - 1) / BQ;
    const long long total = (long long)nqb * B * heads;
    VDA_REQUIRE(total < (1ll << 31), "vda_attention_f16: grid too large");
    VDA_REQUIRE((long long)N * 3 * heads * HD < (1ll << 31), "vda_attention_f16: one frame's qkv exceeds 32-bit element offsets");
    hipStream_t s = (hipStream_t)stream;
    const int g_attn_variant = ::g_attn_variant < 0 ? VDA_ATTN_DEFAULT : ::g_attn_variant;      // (shadows the global inside this call)
#define VDA_ATTN_ABL(K)                                                                                                                        \
    if (g_attn_variant == 20 + K)                                                                                                              \
        hipLaunchKernelGGL((attn_kernel<true, false, false, false, K>), dim3((unsigned)total), dim3(256), 0, s, (const h16*)qkv, (h16*)out, N, heads, nqb, (int)total); \
    else
    VDA_ATTN_ABL(1) VDA_ATTN_ABL(2) VDA_ATTN_ABL(3) VDA_ATTN_ABL(4) VDA_ATTN_ABL(5)
#undef VDA_ATTN_ABL
    if (g_attn_variant == 11)
        hipLaunchKernelGGL((attn_cs_kernel<true>), dim3((unsigned)total), dim3(256), 0, s, (const h16*)qkv, (h16*)out, N, heads, nqb, (int)total);
    else if (g_attn_variant == 10)
        hipLaunchKernelGGL((attn_cs_kernel<false>), dim3((unsigned)total), dim3(256), 0, s, (const h16*)qkv, (h16*)out, N, heads, nqb, (int)total);
    else if (g_attn_variant == 8)
        hipLaunchKernelGGL((attn_cm_kernel<0>), dim3((unsigned)total), dim3(256), 0, s, (const h16*)qkv, (h16*)out, N, heads, nqb, (int)total);
    else if (g_attn_variant == 9)
        hipLaunchKernelGGL((attn_cm_kernel<6>), dim3((unsigned)total), dim3(256), 0, s, (const h16*)qkv, (h16*)out, N, heads, nqb, (int)total);
    else if (g_attn_variant == 7)
        hipLaunchKernelGGL(attn_pipe_kernel, dim3((unsigned)total), dim3(256), 0, s, (const h16*)qkv, (h16*)out, N, heads, nqb, (int)total);
    else if (g_attn_variant == 4)
        hipLaunchKernelGGL((attn_kernel<true, false, false, true>), dim3((unsigned)total), dim3(256), 0, s, (const h16*)qkv, (h16*)out, N, heads, nqb, (int)total);
    else if (g_attn_variant == 5)
        hipLaunchKernelGGL((attn_kernel<true, false, true, true>), dim3((unsigned)total), dim3(256), 0, s, (const h16*)qkv, (h16*)out, N, heads, nqb, (int)total);
    else if (g_attn_variant == 3)
        hipLaunchKernelGGL((attn_kernel<true, false, true>), dim3((unsigned)total), dim3(256), 0, s, (const h16*)qkv, (h16*)out, N, heads, nqb, (int)total);
    else if (g_attn_variant == 1)
        hipLaunchKernelGGL((attn_kernel<true, false>), dim3((unsigned)total), dim3(256), 0, s, (const h16*)qkv, (h16*)out, N, heads, nqb, (int)total);
    else if (g_attn_variant)
        hipLaunchKernelGGL((attn_kernel<true, true>), dim3((unsigned)total), dim3(256), 0, s, (const h16*)qkv, (h16*)out, N, heads, nqb, (int)total);
    else
        hipLaunchKernelGGL((attn_kernel<false, true>), dim3((unsigned)total), dim3(256), 0, s, (const h16*)qkv, (h16*)out, N, heads, nqb, (int)total);
    VDA_LAUNCH_CHECK();
    return 0;
}
