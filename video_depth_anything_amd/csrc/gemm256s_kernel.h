// Large-tile fp16 MFMA GEMM / implicit-GEMM conv for gfx950: 256 x BN x 64 tiles, 8 waves, one
// workgroup per CU. This is the kernel the encoder's qkv/proj/fc1/fc2 GEMMs and the head's wide convs run on.
//
// Why this shape: with a 128x128 tile every 32 MFMAs per wave need 32 KB staged from L2 into LDS; at MFMA
// peak that is more than the L2 can deliver, so the small tile is staging-bound. 256x256 halves the staged
// bytes per FLOP (64 KB per 64 MFMAs per wave) and the 128x64 wave tile cuts LDS fragment reads to
// 0.75 ds_read_b128 per v_mfma_f32_32x32x16_f16.
//
// Pipeline (one __syncthreads per 64-deep K tile):
//   LDS holds two K tiles (2 x 64 KB at BN=256). Fragments are register double-buffered at 16-deep k-step
//   granularity: while the 8 MFMAs of step t issue, the 6 ds_read_b128 of step t+1 are in flight. The last
//   step of tile k waits for tile k+1 (issued one whole tile earlier), barriers once, issues the 8
//   global_load_lds of tile k+2 into the buffer tile k just vacated, and prefetches tile k+1's first
//   fragments while its own MFMAs run, so the barrier never exposes an LDS or HBM round trip.
//
// LDS image: [rows][8 x 16 B], chunk ^= (row >> 1) & 7 (conflict-free for the 32-row ds_read_b128 fragment),
// applied on the DMA SOURCE address and on the read address. MFMA operands are swapped (W rows as "A",
// activation rows as "B") so a lane's accumulator registers are runs of 4 consecutive output columns.
// 16x16x32-MFMA variant of gemm256_kernel.h (same tile, DMA, LDS image, persistence and epilogue; the K loop
// is pipelined in units of 16 MFMAs: one 32-deep k step x half of the wave's rows).
#pragma once
#include "gemm_epilogue.h"

namespace vda_gemm256s {

constexpr int BK = 64;
constexpr int ROW_BYTES = BK * 2;
// BM = 256 on 8 waves, or (BN = 128 only) BM = 192 on 6 waves (3 x 2, the same 64 x 64 wave tile): the tile for problems whose
// 256-row tile count sits just above a multiple of the CU count - ViT-S's proj / fc2: 172 x 3 = 516 tiles = 2.016 rounds run as
// 3; 229 x 3 = 687 tiles of 3/4 the size = 2.68 rounds run as 3 x 0.75 = 2.25.
// [r3] BN = 384 (BM = 192, twelve waves as 2 x 6, the same 96 x 64 wave tile as the 192 x 256 8-phase kernel): ONE tile spans
// ViT-S's whole embedding width, so proj / fc2 read their A panel and their residual rows once, and 43 840 rows are 229 tiles = one
// round of the chip (687 tiles of 192 x 128 ran as 3). qkv (N = 1152) is three such column tiles. 144 KiB of pipeline buffers:
// the epilogue stages through BOTH of them (12 x 8 KiB), so this variant issues the next tile's first K tile after its epilogue.
// [r3] PER_CU = 2 (BM = 192, BN = 128 only): TWO such workgroups per CU, 80 KiB of LDS each (the epilogue stages through the pipeline
// buffers and the next tile's first K tile is issued after it), twelve waves per CU = three per SIMD (<= 168 VGPRs). The two
// workgroups of a CU drift out of phase, so one's epilogue runs under the other's K loop - the experiment VERDICT r2 item 2 asked for
// (vda_gemm_set_variant(11)); measured against the 8-phase 256 x 256 kernel in DESIGN.md section 4.
template <int BN, int AMODE, int EPI, int BM = 256, int PER_CU = 1>
__global__ void __launch_bounds__(BN == 384 ? 768 : BM / 32 * 64) gemm256s_kernel(const vda_gemm_args p) {
    static_assert(PER_CU == 1 || (PER_CU == 2 && BM == 192 && BN == 128), "two workgroups per CU: the 192 x 128 tile only");
    static_assert(BM == 256 || (BM == 192 && (BN == 128 || BN == 384)), "tile heights: 256, or 192 with BN = 128 / 384");
    static_assert(BN != 384 || (BM == 192 && AMODE == VDA_A_DENSE), "384 columns: 192 rows, dense A");
    constexpr int NW = BN == 384 ? 12 : BM / 32;   // waves: 8, or 6, or 12
    constexpr int WN = BN == 384 ? 6 : BN == 256 ? 4 : 2;          // waves along N
    constexpr int WM = NW / WN;                    // waves along M
    constexpr int WTM = BM / WM, WTN = BN / WN;    // wave tile: 128x64 (BN=256) or 64x64 (BN=128)
    constexpr int MI = WTM / 16, NJ = WTN / 16;    // 16x16 subtiles per wave
    constexpr int MH = MI / 2;                       // subtiles per half of the wave's rows (pipeline unit)
    constexpr int AJ = BM / 8 / NW, WP = BN / 8, WJ = (WP + NW - 1) / NW;   // 1-KiB DMA pieces per wave (W: dealt round robin, guarded when NW does not divide them)
    static_assert((BM / 8) % NW == 0 && NW % 2 == 0, "A pieces per wave; the swizzle constant needs an even wave count");
    constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES, STAGE = A_BYTES + W_BYTES;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // Persistent workgroups: one per CU, each walks tiles round by round. In round r the 32 workgroups of an
    // XCD (equal bid % 8) take 32 CONSECUTIVE tiles (N fastest), so concurrently running tiles share A / W panels
    // in that XCD's L2.
    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    const int ntiles = nbm * nbn;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int per_xcd = nwg >> 3;                                    // launch guarantees nwg % 8 == 0
    // last partial round dealt evenly to the XCDs in row-panel groups, as in gemm8p_kernel.h
    const int full_rounds = ntiles / nwg, rem = ntiles - full_rounds * nwg;
    const bool deal_last = rem > 0 && full_rounds > 0 && per_xcd % nbn == 0;
    auto tile_of = [&](int round) {
        if (deal_last && round == full_rounds) {
            const int slot = bid >> 3, j = ((slot / nbn) * 8 + (bid & 7)) * nbn + slot % nbn;
            return j < rem ? round * nwg + j : ntiles;
        }
        return round * nwg + (bid & 7) * per_xcd + (bid >> 3);
    };

    // ---- per-lane DMA sources. A piece is 8 rows x 128 B; lane -> (row lrow of the piece, LDS chunk lane & 7).
    // The swizzled source chunk ((lane&7) ^ ((row>>1)&7)) does not depend on the piece index (pieces are 8 rows
    // apart, the swizzle has period 16 rows and the wave stride is 64 rows), so it is one lane constant.
    // Dense / W offsets are recomputed per K tile from the tile origin (3 VALU ops per piece) instead of being
    // kept in registers; only the conv row decomposition (two divisions per piece) is cached.
    const int lrow = lane >> 3;
    const int src_chk = ((lane & 7) ^ ((((wave & 1) << 2) + (lrow >> 1)) & 7)) * 8;     // halves
    int a_pix[AJ], a_yx[AJ];          // conv: b*H*W, and (oy*stride-1) | (ox*stride-1) << 16
    int tm0 = 0, tn0 = 0;             // origin of the tile being staged
    auto set_sources = [&](int t) {
        const int bm = t / nbn, bn = t - bm * nbn;
        tm0 = bm * BM;
        tn0 = bn * BN;
        if constexpr (AMODE == VDA_A_CONV3X3) {
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                int m = tm0 + (wave + NW * j) * 8 + lrow;
                const bool ok = m < p.M;
                m = min(m, p.M - 1);
                const int hw = p.cHo * p.cWo;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / p.cWo, ox = rem - oy * p.cWo;
                a_pix[j] = b * p.cH * p.cW;
                const int iy0 = ok ? oy * p.cStride - 1 : -20000, ix0 = ox * p.cStride - 1;
                a_yx[j] = (iy0 & 0xffff) | (ix0 << 16);
            }
        }
    };

    // conv: (3x3 tap, 64-channel slice) of the K tile being staged. Set by tap_of() at the tile seams (one integer division by
    // the runtime channel count), advanced incrementally inside the K loop: the division is a ~40-instruction dependent chain
    // that would sit between the barrier and the DMA burst of every K tile.
    int cv_tap = 0, cv_ci0 = 0;
    auto tap_of = [&](int kt) {
        if constexpr (AMODE == VDA_A_CONV3X3) {
            const int k0 = kt * BK;
            cv_tap = k0 / p.cCin;
            cv_ci0 = k0 - cv_tap * p.cCin;
        }
    };
    auto tap_next = [&]() {
        if constexpr (AMODE == VDA_A_CONV3X3) {
            cv_ci0 += BK;
            if (cv_ci0 >= p.cCin) {
                cv_ci0 = 0;
                ++cv_tap;
            }
        }
    };
    auto stage = [&](int kt, char* buf) {
        const int k0 = kt * BK;
        if constexpr (AMODE == VDA_A_DENSE) {
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const int m = min(tm0 + (wave + NW * j) * 8 + lrow, p.M - 1);
                glds16((const h16*)p.A + (size_t)(unsigned)(m * p.lda + src_chk + k0), buf + (wave + NW * j) * 1024);
            }
        } else {
            const int tap = cv_tap, ci0 = cv_ci0;
            const int ky = (tap * 11) >> 5, kx = tap - ky * 3;   // tap / 3 for tap = 0..8
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const int iy = (int)(short)(a_yx[j] & 0xffff) + ky, ix = (a_yx[j] >> 16) + kx;
                const bool ok = (unsigned)iy < (unsigned)p.cH && (unsigned)ix < (unsigned)p.cW;
                const h16* src = ok ? (const h16*)p.A + ((size_t)(a_pix[j] + iy * p.cW + ix) * p.cCin + ci0 + src_chk)
                                    : (const h16*)p.zero_page + src_chk;
                glds16(src, buf + (wave + NW * j) * 1024);
            }
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            if (WP % NW != 0 && wave + NW * j >= WP) break;                  // wave-uniform
            const int n = min(tn0 + (wave + NW * j) * 8 + lrow, p.N - 1);
            glds16((const h16*)p.W + (size_t)(unsigned)(n * p.K + src_chk + k0), buf + A_BYTES + (wave + NW * j) * 1024);
        }
    };

    // fragment addressing: row = lane & 15 inside a 16-row subtile, k-chunk = 4*ks + (lane >> 4)
    const int frow = lane & 15, fh = lane >> 4, fsw = (lane >> 1) & 7;     // ((row >> 1) & 7) with row = subtile*16 + frow
    const int a_off = (wm * WTM + frow) * ROW_BYTES, w_off = A_BYTES + (wn * WTN + frow) * ROW_BYTES;
    // relu on the activation operand (conv only), branch-free: max(x, 0) or max(x, -inf)
    const h16 relu_floor = (p.relu_in & 1) ? (h16)0.f : (h16)(-65504.f);
    h16x8 relu_thr;
#pragma unroll
    for (int e = 0; e < 8; ++e) relu_thr[e] = relu_floor;

    struct AF {
        h16x8 a[MH];
    };
    struct WF {
        h16x8 w[NJ];
    };
    auto read_a = [&](const char* buf, int ks, int half, AF& f) {
        const int coff = ((4 * ks + fh) ^ fsw) << 4;
#pragma unroll
        for (int i = 0; i < MH; ++i) f.a[i] = *reinterpret_cast<const h16x8*>(buf + a_off + (half * MH + i) * 16 * ROW_BYTES + coff);
    };
    auto read_w = [&](const char* buf, int ks, WF& f) {
        const int coff = ((4 * ks + fh) ^ fsw) << 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) f.w[j] = *reinterpret_cast<const h16x8*>(buf + w_off + j * 16 * ROW_BYTES + coff);
    };

    const int nt = p.K / BK;
    static_assert(WTN == 64, "epilogue staging assumes a 64-column wave tile");
    // epilogue staging (NW waves x 8 KiB): inside pipeline buffer 1 when that is 64 KiB (BN=256), else after the buffers
    constexpr bool LATE_NEXT = PER_CU == 2 || (NW * 8192 > STAGE && 2 * STAGE + NW * 8192 > 160 * 1024);   // staging needs both buffers (BN = 384)
    constexpr int STG_OFF = LATE_NEXT ? 0 : (STAGE >= 8 * 8192) ? STAGE : 2 * STAGE;
    char* stg = smem + STG_OFF + wave * 8192;

    int tile = tile_of(0);
    if (tile >= ntiles) return;                        // uniform per workgroup
    // stagger of the workgroups that sit out the last round (gemm8p_kernel.h): up to 3/4 of a tile time, four phases, free
    if (!((p.relu_in >> 8) & 16) && rem > 0 && full_rounds > 0 && tile_of(full_rounds) >= ntiles) {       // VDA_GEMM_STAGGER=0 switches it off
        // phase by slot: the nbn workgroups of an XCD that share an A row panel land in DIFFERENT phases (keeping them in phase -
        // VDA_GEMM_STAGGER=2 - is slower than no stagger at all: it is those neighbours' epilogues that collide)
        const int q = ((p.relu_in >> 8) & 32) ? ((bid >> 3) / (nbn <= 8 ? nbn : 8)) & 3 : (bid >> 3) & 3;
        const int units = (nt * 60 + 260) * q / 4;
        for (int i = 0; i < units; i += 120) __builtin_amdgcn_s_sleep(120);
    }
    set_sources(tile);
    tap_of(0);
    stage(0, smem);
    for (int round = 0; tile < ntiles; ++round) {
        const int bm = tile / nbn, bn = tile - bm * nbn;
        const int m0 = bm * BM, n0 = bn * BN;

        f32x4 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto mma = [&](int half, AF& fa, const WF& fw) {
            if constexpr (AMODE == VDA_A_CONV3X3) {
#pragma unroll
                for (int i = 0; i < MH; ++i) fa.a[i] = __builtin_elementwise_max(fa.a[i], relu_thr);
            }
#pragma unroll
            for (int i = 0; i < MH; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (half == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw.w[j], fa.a[i], acc[i][j], 0, 0, 0);
                    else acc[MH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw.w[j], fa.a[i], acc[MH + i][j], 0, 0, 0);
                }
        };

        // K tile 0 of this tile was issued before the previous tile's epilogue (or above, for the first tile).
        AF a0, a1;
        WF w0, w1;
        if (round > 0) set_sources(tile);       // recomputed rather than kept live across the epilogue
        // Full barrier (with its fences, so no LDS read can be scheduled above it): K tile 0 has landed everywhere.
        // K tile 1 is issued after it and has the whole of K tile 0's MFMA work to land, as in the steady state.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nt > 1) {
            tap_of(1);
            stage(1, smem + STAGE);
        }
        tap_of(2);
        read_w(smem, 0, w0);
        read_a(smem, 0, 0, a0);
        for (int kt = 0; kt < nt; ++kt) {
            char* cb = smem + (kt & 1) * STAGE;
            char* nb = smem + ((kt + 1) & 1) * STAGE;
            read_a(cb, 0, 1, a1);
            read_w(cb, 1, w1);
            mma(0, a0, w0);
            read_a(cb, 1, 0, a0);
            mma(1, a1, w0);
            read_a(cb, 1, 1, a1);
            mma(0, a0, w1);
            // Tile kt+1 must have landed everywhere and every wave must be done READING tile kt (its last fragments
            // are already in a1 / w1 once lgkmcnt drains): one full barrier per K tile, explicit DMA drain first.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 2 < nt) {
                stage(kt + 2, cb);
                tap_next();
            }
            if (kt + 1 < nt) {
                read_w(nb, 0, w0);
                read_a(nb, 0, 0, a0);
            }
            mma(1, a1, w1);
        }
        // After the loop's last barrier nobody reads the pipeline buffers any more: start the NEXT tile's first
        // K tile now, so its HBM/L2 latency is covered by this tile's epilogue.
        const int next = tile_of(round + 1);
        if (!LATE_NEXT && next < ntiles) {
            set_sources(next);
            tap_of(0);
            stage(0, smem);
        }

        // ---- epilogue. Accumulator register e of subtile (i,j) is row m = lane & 31, column 8*(e>>2) + 4*(lane>>5) + (e&3):
        // stored straight from this layout a wave instruction would touch 32 rows x 16 B (32 partial lines). Instead each
        // wave transposes one 32-row x 64-column fp32 block at a time through its own 8 KiB of (idle) pipeline buffer 1
        // (16-byte chunks XOR-swizzled by row: conflict-free both ways) and then owns whole row segments: bias / LayerScale /
        // residual reads and the output stores become contiguous 16-byte accesses covering full 128-byte lines.
        const int bm0 = m0 + wm * WTM, bn0 = n0 + wn * WTN;
        const bool interior = bm0 + WTM <= p.M && bn0 + WTN <= p.N;       // wave-uniform
        {
            using RT = vda_gemm::RowTraits<EPI>;
            constexpr int NC = RT::NC, RR = RT::f32_out ? 8 : 4;            // columns per lane, row groups per 32-row block
            // lane -> (row inside a group, column block): fp32 out: 4 rows x 16 lanes x 4 cols; fp16 out: 8 rows x 8 lanes x 8 cols
            // (twelve-wave tile, 168 VGPRs: the epilogue's lane-derived offsets are recomputed here from an opaque copy of the lane id,
            // so that hipcc cannot hoist them - and the 64-bit addresses built on them - above the tile loop and keep them live, i.e.
            // spilled, across the K loop: the split-residual instantiation reloaded two of its K-loop registers from scratch in EVERY K
            // tile, and a scratch reload is a vector-memory load whose in-order wait also drains the tile's LDS-DMA)
            int lane_e = lane;
            if constexpr (BN == 384) asm volatile("" : "+v"(lane_e));
            const int lrow_e = RT::f32_out ? (lane_e >> 4) : (lane_e >> 3);
            const int c0 = RT::f32_out ? (lane_e & 15) : 2 * (lane_e & 7);      // first 16-byte (4-column) chunk of the lane
            const int en = bn0 + c0 * 4;
            const bool geglu_idle = (EPI == VDA_EPI_GEGLU_F16) && (c0 & 7) >= 4;   // gate lanes only feed their value lanes
            vda_gemm::ColConst<NC> cc;
            vda_gemm::load_col_const<EPI, NC>(p, en, cc);
            // rows per row group (twelve-wave tile: half as many - three RowAux sets instead of six are live at the first block, where
            // all 96 accumulators still are; twelve waves per CU keep as many row loads in flight as eight did)
            constexpr int RG = RT::f32_out ? (BN == 384 ? 2 : 4) : (BN == 384 ? 1 : 2);
            vda_gemm::RowAux carry[RG];                                      // the next block's first group, loaded a group early
#pragma unroll
            for (int i = 0; i < MI / 2; ++i) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        // accumulator (mi = 2i+h2, nj = j): row = h2*16 + (lane & 15) of the 32-row block, columns 16j + 4*(lane>>4) + e
                        const int row = h2 * 16 + (lane_e & 15), c = j * 4 + (lane_e >> 4);
                        *reinterpret_cast<f32x4*>(stg + row * 256 + ((c ^ (row & 15)) << 4)) = acc[2 * i + h2][j];
                    }
                // The transposition is a cross-LANE exchange inside one wave: the hardware runs a wave's LDS ops in order,
                // but the compiler must not reorder the reads below across the writes above (it only reasons per thread).
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // Row groups are processed RG at a time: phase 1 issues every row-dependent global load of the group,
                // phase 2 reads the transposed accumulators back, finishes and stores.
                // Row-dependent loads (residuals, pos-embed) run ONE ROW GROUP AHEAD of the stores: vmcnt retires in order and
                // counts stores, so a load issued after a group's stores cannot complete before them - issued before, it can.
                auto row_groups = [&](auto guard) {
                    constexpr bool GUARD = decltype(guard)::value;
                    constexpr int NG = RR / RG;                                  // row groups per 32-row block
                    auto load_group = [&](int blk, int r0, vda_gemm::RowAux (&ax)[RG]) {
                        if (!geglu_idle) {
#pragma unroll
                            for (int q = 0; q < RG; ++q)
                                vda_gemm::load_row_aux<EPI, GUARD>(p, bm0 + blk * 32 + (r0 + q) * (32 / RR) + lrow_e, en, ax[q]);
                        }
                    };
                    vda_gemm::RowAux aux[2][RG];
                    if (i == 0) load_group(0, 0, aux[0]);
                    else {
#pragma unroll
                        for (int q = 0; q < RG; ++q) aux[0][q] = carry[q];
                    }
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const int r0 = g * RG;
                        if (g + 1 < NG) load_group(i, r0 + RG, aux[(g + 1) & 1]);
                        else if (i + 1 < MI / 2) load_group(i + 1, 0, carry);     // first group of the next block
#pragma unroll
                        for (int q = 0; q < RG; ++q) {
                            const int row = (r0 + q) * (32 / RR) + lrow_e;
                            const char* rp = stg + row * 256;
                            const f32x4 a = *reinterpret_cast<const f32x4*>(rp + ((c0 ^ (row & 15)) << 4));
                            if constexpr (RT::f32_out) {
                                vda_gemm::finish_row4<EPI, GUARD>(p, bm0 + i * 32 + row, en, a, cc, aux[g & 1][q]);
                            } else {
                                const f32x4 b = *reinterpret_cast<const f32x4*>(rp + (((c0 + 1) ^ (row & 15)) << 4));
                                float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                                float gt[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                if constexpr (EPI == VDA_EPI_GEGLU_F16) {
                                    const f32x4 ga = *reinterpret_cast<const f32x4*>(rp + (((c0 + 4) & 15) ^ (row & 15)) * 16);
                                    const f32x4 gb = *reinterpret_cast<const f32x4*>(rp + (((c0 + 5) & 15) ^ (row & 15)) * 16);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        gt[e] = ga[e];
                                        gt[4 + e] = gb[e];
                                    }
                                }
                                if (!geglu_idle) vda_gemm::finish_row8<EPI, GUARD>(p, bm0 + i * 32 + row, en, v, gt, cc, aux[g & 1][q]);
                                if constexpr (EPI == VDA_EPI_SCALE_RES_SPLIT) {      // all 8 lanes of the row store the same pair: no divergence
                                    if (!GUARD || (bm0 + i * 32 + row < p.M && en < p.N)) vda_gemm::store_split_stats(p, bm0 + i * 32 + row, en, aux[g & 1][q]);
                                }
                            }
                        }
                    }
                };
                // interior wave tiles (all but the matrix's last row / column of tiles) take the branch-free form
                if (interior) row_groups(std::false_type{});
                else row_groups(std::true_type{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads of this block done before the next block's writes
            }
        }
        // Every wave is done with its staging slice before the next tile's stage(1) overwrites buffer 1.
        __syncthreads();
        if (LATE_NEXT && next < ntiles) {       // (staging used both buffers: the next tile's first K tile starts here)
            set_sources(next);
            tap_of(0);
            stage(0, smem);
        }
        tile = next;
    }
}

template <int BN, int AMODE, int EPI, int BM = 256, int PER_CU = 1>
int launch256(const vda_gemm_args& a, hipStream_t s) {
    constexpr int stage_bytes = (BM + BN) * ROW_BYTES;
    constexpr int nthreads = BN == 384 ? 768 : BM / 32 * 64;
    constexpr int smem = (PER_CU == 2 || BN == 384 || stage_bytes >= 8 * 8192) ? 2 * stage_bytes : 2 * stage_bytes + BM / 32 * 8192;
    static_assert(smem * PER_CU <= 160 * 1024, "LDS budget");
    static VdaKernelDeviceState dev_state;
    const int num_cu = vda_prepare_kernel(reinterpret_cast<const void*>(&gemm256s_kernel<BN, AMODE, EPI, BM, PER_CU>), smem, dev_state);
    if (num_cu < 0) return 2;
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    const int ntiles = nbm * nbn;
    const int grid = ntiles < PER_CU * num_cu ? (ntiles + 7) / 8 * 8 : PER_CU * num_cu;     // PER_CU persistent workgroups per CU
    if (PER_CU == 2 && getenv("VDA_DEBUG_OCC")) {           // what the runtime thinks fits on a CU (A/B aid)
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm256s_kernel<BN, AMODE, EPI, BM, PER_CU>, nthreads, smem);
        fprintf(stderr, "gemm256s<%d,%d,%d,%d,2>: %d workgroups of %d threads, %d B LDS per CU (runtime occupancy query)\n", BN, AMODE, EPI, BM, nb, nthreads, smem);
    }
    hipLaunchKernelGGL((gemm256s_kernel<BN, AMODE, EPI, BM, PER_CU>), dim3(grid), dim3(nthreads), smem, s, a);
    VDA_LAUNCH_CHECK();
    return 0;
}

// Dense A: every epilogue. Conv A: the three the head uses.
template <int BN>
int launch_dense(const vda_gemm_args& a, hipStream_t s) {
    switch (a.epilogue) {
        case VDA_EPI_BIAS_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_BIAS_F16>(a, s);
        case VDA_EPI_BIAS_GELU_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_BIAS_GELU_F16>(a, s);
        case VDA_EPI_BIAS_RELU_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_BIAS_RELU_F16>(a, s);
        case VDA_EPI_SCALE_RES_F32: return launch256<BN, VDA_A_DENSE, VDA_EPI_SCALE_RES_F32>(a, s);
        case VDA_EPI_RES_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_RES_F16>(a, s);
        case VDA_EPI_GEGLU_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_GEGLU_F16>(a, s);
        case VDA_EPI_PATCH_F32: return launch256<BN, VDA_A_DENSE, VDA_EPI_PATCH_F32>(a, s);
        case VDA_EPI_CONVT_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_CONVT_F16>(a, s);
        case VDA_EPI_BIAS_F32: return launch256<BN, VDA_A_DENSE, VDA_EPI_BIAS_F32>(a, s);
        case VDA_EPI_SCALE_RES_F32_H: return launch256<BN, VDA_A_DENSE, VDA_EPI_SCALE_RES_F32_H>(a, s);
        case VDA_EPI_SCALE_RES_SPLIT: return launch256<BN, VDA_A_DENSE, VDA_EPI_SCALE_RES_SPLIT>(a, s);
        case VDA_EPI_LN_BIAS_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_LN_BIAS_F16>(a, s);
        case VDA_EPI_LN_GELU_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_LN_GELU_F16>(a, s);
        default: break;
    }
    return -1;
}

// two 192 x 128 workgroups per CU: the epilogues whose kernels fit 168 VGPRs (three waves per SIMD)
template <int BN>
int launch_dense_bm192_x2(const vda_gemm_args& a, hipStream_t s) {
    switch (a.epilogue) {
        case VDA_EPI_BIAS_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_BIAS_F16, 192, 2>(a, s);
        case VDA_EPI_BIAS_GELU_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_BIAS_GELU_F16, 192, 2>(a, s);
        case VDA_EPI_LN_BIAS_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_LN_BIAS_F16, 192, 2>(a, s);
        case VDA_EPI_LN_GELU_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_LN_GELU_F16, 192, 2>(a, s);
        default: break;
    }
    return -1;
}

// 192-row tiles (BN = 128, dense A): the epilogues the encoder uses
template <int BN>        // (a template only so that the kernels are instantiated in the one translation unit that calls it)
int launch_dense_bm192(const vda_gemm_args& a, hipStream_t s) {
    switch (a.epilogue) {
        case VDA_EPI_BIAS_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_BIAS_F16, 192>(a, s);
        case VDA_EPI_BIAS_GELU_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_BIAS_GELU_F16, 192>(a, s);
        case VDA_EPI_SCALE_RES_F32: return launch256<BN, VDA_A_DENSE, VDA_EPI_SCALE_RES_F32, 192>(a, s);
        case VDA_EPI_SCALE_RES_SPLIT: return launch256<BN, VDA_A_DENSE, VDA_EPI_SCALE_RES_SPLIT, 192>(a, s);
        case VDA_EPI_LN_BIAS_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_LN_BIAS_F16, 192>(a, s);
        case VDA_EPI_LN_GELU_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_LN_GELU_F16, 192>(a, s);
        default: break;
    }
    return -1;
}

// 192 x 384 tiles on twelve waves (dense A): ViT-S's embedding width in one tile
template <int BN>
int launch_dense_bn384(const vda_gemm_args& a, hipStream_t s) {
    switch (a.epilogue) {
        case VDA_EPI_BIAS_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_BIAS_F16, 192>(a, s);
        case VDA_EPI_SCALE_RES_F32: return launch256<BN, VDA_A_DENSE, VDA_EPI_SCALE_RES_F32, 192>(a, s);
        case VDA_EPI_SCALE_RES_SPLIT: return launch256<BN, VDA_A_DENSE, VDA_EPI_SCALE_RES_SPLIT, 192>(a, s);
        case VDA_EPI_LN_BIAS_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_LN_BIAS_F16, 192>(a, s);
        case VDA_EPI_LN_GELU_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_LN_GELU_F16, 192>(a, s);
        default: break;
    }
    return -1;
}

template <int BN>
int launch_conv(const vda_gemm_args& a, hipStream_t s) {
    switch (a.epilogue) {
        case VDA_EPI_BIAS_F16: return launch256<BN, VDA_A_CONV3X3, VDA_EPI_BIAS_F16>(a, s);
        case VDA_EPI_BIAS_RELU_F16: return launch256<BN, VDA_A_CONV3X3, VDA_EPI_BIAS_RELU_F16>(a, s);
        case VDA_EPI_RES_F16: return launch256<BN, VDA_A_CONV3X3, VDA_EPI_RES_F16>(a, s);
        default: break;
    }
    return -1;                                  // caller falls back to the 128-row kernel
}

}  // namespace vda_gemm256s
