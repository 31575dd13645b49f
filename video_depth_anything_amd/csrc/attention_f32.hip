// fp32-operand spatial self-attention (the `fp32=True` path: dinov2_layers/attention.py:51-59 with autocast off).
// softmax(q k^T / 8) v, head dim 64, fp32 in / fp32 out, every product on v_mfma_f32_32x32x2_f32 (exact fp32); the N x N
// scores never leave registers (online softmax over 64-key tiles). Same orientation as the fp16 kernel (attention.hip):
//   S^T[key][query] = K_tile . Q^T  -> a lane owns ONE query (lane & 31) and 32 of the tile's 64 keys (its partner lane ^ 32
//                                      the other 32): in-lane softmax plus one cross-lane exchange;
//   O^T[ch][query] += V^T . P^T     -> accumulator register e of S^T (keys k and k + 4 on the two lane halves) IS the B
//                                      operand of one 32x32x2 MFMA; V^T comes from the row-major V tile by ds_read_b32
//                                      (32 consecutive channels of one key per lane half: conflict-free).
// MFMA-bound at 1/16 of the fp16 rate, so staging is plain: global -> registers (issued before the tile's compute, T14
// "issue early / write late") -> LDS rows of 64 + 4 floats (the pad spreads a b128 lane group's 16 rows over all 16 slots).
#include "vda_common.h"

namespace {

constexpr int HD = 64;
constexpr int BQ = 128;      // queries per workgroup (4 waves x 32)
constexpr int BKV = 64;      // keys per tile
constexpr int PITCH = HD + 4;                      // floats per LDS row
constexpr int TILE_FLOATS = BKV * PITCH;

__global__ void __launch_bounds__(256) attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int H, int nqb,
                                                       int total_blocks) {
    __shared__ __attribute__((aligned(16))) float lk[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float lv[TILE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware remap: blocks sharing an XCD (bid % 8) take whole (frame, head) groups so K/V stay in that L2
    const int bid = blockIdx.x;
    const int xcd = bid & 7, qd = total_blocks >> 3, rm = total_blocks & 7;
    const int t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int bh = t / nqb, qb = t - bh * nqb;
    const int b = bh / H, head = bh - b * H;

    const size_t rs = (size_t)3 * H * HD;                       // row stride of qkv in floats
    const float* Qb = qkv + (size_t)b * N * rs + head * HD;
    const float* Kb = Qb + (size_t)H * HD;
    const float* Vb = Kb + (size_t)H * HD;

    // Q fragments: lane holds Q[query r][8j + 4h .. +3], j = 0..7, pre-scaled by 1/8 (exact)
    const int q_row = qb * BQ + wave * 32 + r;
    f32x4 qf[8];
    {
        const float* qp = Qb + (size_t)min(q_row, N - 1) * rs + 4 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[j] = *reinterpret_cast<const f32x4*>(qp + 8 * j) * 0.125f;
    }

    // staging: thread -> (row = tid >> 2 of the tile, 16-float segment tid & 3): 4 x 16-byte loads per tensor
    const int srow = tid >> 2, sseg = (tid & 3) * 16;
    f32x4 kreg[4], vreg[4];
    auto fetch = [&](int kt) {
        const size_t key = (size_t)min(kt * BKV + srow, N - 1);
        const float* kp = Kb + key * rs + sseg;
        const float* vp = Vb + key * rs + sseg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            kreg[i] = *reinterpret_cast<const f32x4*>(kp + 4 * i);
            vreg[i] = *reinterpret_cast<const f32x4*>(vp + 4 * i);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(lk + srow * PITCH + sseg + 4 * i) = kreg[i];
            *reinterpret_cast<f32x4*>(lv + srow * PITCH + sseg + 4 * i) = vreg[i];
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
    fetch(0);
    commit();
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        if (kt + 1 < nt) fetch(kt + 1);                       // in flight under this tile's MFMAs

        // S^T = K . Q^T for the tile's two 32-key halves
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[sub][e] = 0.f;
            const float* kp = lk + (sub * 32 + r) * PITCH + 4 * h;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + 8 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[j][e], s[sub], 0, 0, 0);
            }
        }
        // register e of half `sub` is key kt*64 + sub*32 + (e&3) + 8*(e>>2) + 4h
        if (kt == nt - 1 && (N % BKV) != 0) {
            const int kbase = kt * BKV + 4 * h;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (kbase + sub * 32 + (e & 3) + 8 * (e >> 2) >= N) s[sub][e] = -1e30f;
        }

        // online softmax (per query = per lane pair {lane, lane ^ 32})
        float mx = s[0][0];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[sub][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f((m_run - m_new) * LOG2E);
        l_run *= alpha;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc_o[c][e] *= alpha;
        m_run = m_new;
        const float mb = m_run * LOG2E;
        float ps = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pv = exp2f(fmaf(s[sub][e], LOG2E, -mb));
                s[sub][e] = pv;
                ps += pv;
            }
        l_run += ps;

        // O^T += V^T . P^T: one MFMA per accumulator register of S^T and 32-channel half
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float vf = lv[key * PITCH + c * 32 + r];
                    acc_o[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[sub][e], acc_o[c], 0, 0, 0);
                }
            }
        __syncthreads();                                      // everyone is done reading this tile
        if (kt + 1 < nt) {
            commit();
            __syncthreads();
        }
    }

    // normalise and store: lane holds query r, channels c*32 + (e&3) + 8*(e>>2) + 4h
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
    if (q_row < N) {
        float* op = out + ((size_t)b * N + q_row) * ((size_t)H * HD) + head * HD + 4 * h;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 o = {acc_o[c][4 * g] * inv, acc_o[c][4 * g + 1] * inv, acc_o[c][4 * g + 2] * inv, acc_o[c][4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(op + c * 32 + 8 * g) = o;
            }
    }
}

}  // namespace

extern "C" int vda_attention_f32(const float* qkv, float* out, int B, int N, int heads, vda_stream_t stream) {
    VDA_REQUIRE(qkv && out, "vda_attention_f32: null pointer");
    VDA_REQUIRE(B > 0 && N > 0 && heads > 0, "vda_attention_f32: empty problem B=%d N=%d heads=%d", B, N, heads);
    VDA_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "vda_attention_f32: 16-byte alignment required");
    const int nqb = (N + BQ - 1) / BQ;
    const long long total = (long long)nqb * B * heads;
    VDA_REQUIRE(total < (1ll << 31), "vda_attention_f32: grid too large");
    hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, qkv, out, N, heads, nqb, (int)total);
    VDA_LAUNCH_CHECK();
    return 0;
}
