// 8-phase, two-group (ping-pong) schedule of the 256 x 256 x 64 fp16 MFMA GEMM / implicit-GEMM conv for gfx950.
// Same tile, LDS image (rows of 128 B, 16-byte chunks XOR-swizzled by (row >> 1) & 7 on the DMA SOURCE address and on the
// read), fragment shapes (v_mfma_f32_16x16x32_f16, swapped operands), persistence and epilogue as gemm8p_kernel.h;
// what changes is WHEN things are issued:
//
//   * A K tile is four phases of 16 MFMAs: (k-step 0, rows lo), (k0, rows hi), (k1, rows lo), (k1, rows hi) of the wave's
//     128 x 64 tile. A phase = { ds_read its fragments (8 / 4 / 8 / 4 x b128) ; issue a slice of the LDS-DMA of K tile
//     kt+2 ; lgkmcnt(0) ; s_barrier ; 16 MFMAs at s_setprio 1 ; s_barrier }.
//   * Waves 4..7 (the SIMD partners of waves 0..3) run ONE BARRIER BEHIND waves 0..3: while one wave of a SIMD is in its
//     MFMA section its partner is in its load section, so the DMA issue cost (60..185 cycles per 1-KiB piece, measured in
//     MI355X_MICROARCH.md) and the LDS reads never hold the matrix pipe. In the one-barrier-per-K-tile kernel both waves of
//     a SIMD issue their 8-piece DMA burst at the same moment (skipping the DMA there saves 24 %).
//   * LDS-DMA stays in flight across barriers: raw s_barrier, counted s_waitcnt vmcnt(6) once per K tile (phase 4) instead
//     of vmcnt(0) + __syncthreads. A is triple-buffered (3 x 32 KiB), W double-buffered (2 x 32 KiB) = 160 KiB: A(kt+2)
//     is issued in phases 2-3 of K tile kt, W(kt+2) half in phase 4 (its slot is K tile kt's own, last read in phase 3) and
//     half in phase 1 of K tile kt+1: two pieces per wave and phase, every piece has at least 3 phases of MFMA time to land
//     (SCHED 0 keeps the 2/2/0/4 form with vmcnt(8) and 4 phases of cover for A/B).
//
// Ordering rules followed (cdna_hip_programming.md, "The 256^2 8-phase template"):
//   RAW: a wave's counted vmcnt precedes its phase-4 barrier; K tile kt+1 is first read in phase 1 of the next K tile, i.e.
//        after a barrier every wave passed AFTER its own wait (group 1's phase-4 barrier 1 is group 0's barrier 2).
//   WAR: every reading phase drains lgkmcnt(0) BEFORE its first barrier; a slot is re-staged no earlier than the phase
//        after its last read (A: last read phase 4 of K tile kt-1, re-staged phase 1 of kt; W: phase 3 -> phase 4).
#pragma once
#include "gemm_epilogue.h"

namespace vda_gemm8p {

constexpr int BK = 64;
constexpr int ROW_BYTES = BK * 2;
constexpr int BM_MAX = 256;           // the LDS slots are laid out for the 256-row tile (a 192-row tile uses the first 192 rows of each)
constexpr int NW = 8;                 // waves
constexpr int NT = NW * 64;

// counted wait: everything but the wave's KEEP youngest vector-memory operations (LDS-DMA pieces) has landed
template <int KEEP>
__device__ __forceinline__ void vm_wait_keep() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
}

// BM = 192 (BN = 256 only): the same kernel on 192 x 256 tiles (wave tile 96 x 64, three A pieces per wave and K tile, 12 MFMAs per
// phase). A row's arithmetic is the same in either tile (same K order, same epilogue), so a GEMM may be cut by rows into a part
// that fills whole rounds of 256-row tiles and a remainder on 192-row tiles whose single round costs ~0.8 of a 256-row round:
// M = 43 840 leaves 2.69 / 8.06 rounds (proj and fc2 / qkv) that ran as 3 / 9 (vda_gemm_plan_split, gemm.hip).
// DYN = the dynamic tile draw (vda_gemm_args.sched) is compiled in. A TEMPLATE parameter, not a runtime test of p.sched: the draw is
// a returning atomic, and with it in the code - executed or not - hipcc's waitcnt pass parks `s_waitcnt vmcnt(0)` at the top of every
// tile (the atomic's destination VGPR is re-initialised there and may still be in flight as far as the pass can tell). That wait
// drains the previous epilogue's whole store burst and the prefetched K tile 0: ~4 k cycles per tile in EVERY launch, found in round 4
// (tools/gemm_ramp.py + the ISA). The static-schedule kernel (what a single-stream forward runs) now has no atomic in it at all.
template <int BN, int AMODE, int EPI, int SCHED = 1, int BM = 256, bool DYN = false>
__global__ void __launch_bounds__(NT) gemm8p_kernel(const vda_gemm_args p) {
    static_assert(BM == 256 || (BM == 192 && BN == 256 && AMODE == VDA_A_DENSE), "192-row tiles: dense A, 256 columns");
    constexpr int WN = BN == 256 ? 4 : 2;          // waves along N
    constexpr int WM = NW / WN;                    // waves along M
    constexpr int WTM = BM / WM, WTN = BN / WN;    // wave tile: 128x64 (BN=256) or 64x64 (BN=128)
    constexpr int MI = WTM / 16, NJ = WTN / 16;    // 16x16 subtiles per wave
    constexpr int MH = MI / 2;                       // subtiles per half of the wave's rows (pipeline unit)
    constexpr int AJ = BM / 8 / NW, WJ = BN / 8 / NW;   // 1-KiB DMA pieces per wave
    constexpr int A_BYTES = BM_MAX * ROW_BYTES, W_BYTES = BN * ROW_BYTES;     // (A_BYTES: slot stride)
    static_assert(BN == 256 || BN == 128, "the 8-phase schedule is built for 256 x 256 and 256 x 128 tiles");
    // 256 x 128 (8 waves as 4 x 2, wave tile 64 x 64): the same phases with half the W pieces (1 / 2 / 2 / 1 per phase instead of
    // 2 / 2 / 2 / 2); the counted wait of phase 4 leaves A(kt+2) and the first half of W(kt+2) in flight: AJ + WJ/2 pieces.
    constexpr int WH = WJ / 2;                         // W pieces per half-tile of W rows
    constexpr int A_SLOTS = 3, W_BASE = A_SLOTS * A_BYTES;          // [A0 | A1 | A2 | W0 | W1] = 160 KiB

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int grp = wave >> 2;                         // wave group: 0 = waves 0..3, 1 = their SIMD partners (rows 128.. of the tile)

    // Persistent workgroups: one per CU, each walks tiles round by round. In round r the 32 workgroups of an
    // XCD (equal bid % 8) take 32 CONSECUTIVE tiles (N fastest), so concurrently running tiles share A / W panels
    // in that XCD's L2.
    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    const int ntiles = nbm * nbn;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int per_xcd = nwg >> 3;                                    // launch guarantees nwg % 8 == 0
    // The LAST, partial round (ntiles % nwg tiles) is dealt to the XCDs in groups of nbn consecutive tiles (one row panel), round
    // robin, so that every XCD keeps the same share of busy CUs instead of whole XCDs going idle (needs nbn | per_xcd; otherwise
    // the plain mapping stays). A workgroup without a tile in that round has one tile time of slack: see the stagger below.
    const int full_rounds = ntiles / nwg, rem = ntiles - full_rounds * nwg;
    const bool deal_last = rem > 0 && full_rounds > 0 && per_xcd % nbn == 0;
    // L2-BLOCKED tile order: an XCD keeps ONE group of CG column panels for the whole launch (its W slice - 4 x 256 x K x 2 B = 2 MB at
    // K = 1024 - stays in its 4 MB L2) and walks down the row panels, interleaved with the other XCDs of its column group, instead of
    // 32 consecutive tiles per round (N fastest), which at nbn = 16 re-fetches all 8 MB of W past L2 every round (fc1: 832 MB fetched
    // per launch against 98 MB algorithmic, profiles/r03; 546 MB with it, profiles/r04). Needs 32 workgroups per XCD and at least one
    // full round; any other launch keeps the default walk. vda_gemm_set_debug: bit 0 = clock stamps (below), bit 1 = blocked also for
    // nbn = 4, bit 4 = never blocked.
    const int dbg2 = __builtin_amdgcn_readfirstlane((p.relu_in >> 16) & 0xff);
    // Column group width CG (panels) and the number of groups G = nbn / CG (2, 4 or 8: 8 / G XCDs share a group and interleave its
    // row panels): nbn = 8, 16, 32 -> CG 4; on request (debug bit 1) also nbn = 12 (qkv), 24 -> CG 6 and nbn = 4 -> one group: qkv
    // measured 275.0 against 272.7 us with it on one box - no gain; the sign of the fc1 gain itself flips between boxes (-1.5 % / +0.7 %).
    // DEFAULT for those widths (fc1: the dominant kernel fetched 8.5 x its algorithmic bytes without it; -1.5 .. -3 % on that launch
    // alone, +0.3 % on the power-capped forward); nbn = 4 (proj / fc2: the default walk already shares all of W) only when debug bit 1
    // asks for it (measured +1 % slower there); debug bit 4 switches it off everywhere (A/B).
    const int cgw = (nbn % 4 == 0 && (nbn == 8 || nbn == 16 || nbn == 32 || (nbn == 4 && (dbg2 & 2)))) ? 4 : (((nbn == 12 || nbn == 24) && (dbg2 & 2)) ? 6 : 0);
    const int cgroups = cgw ? nbn / cgw : 1;
    const bool blocked = cgw != 0 && per_xcd == 32 && !DYN && !(dbg2 & 16) && full_rounds > 0;
    auto tile_of = [&](int round) {
        if (blocked) {
            // the group's tiles in row-major order (CG wide), dealt 32 at a time to its XCDs: XCD i of the group takes chunk r * xpg + i
            const int x = bid & 7, slot = bid >> 3, xpg = 8 / cgroups;
            const int idx = ((round * xpg + x / cgroups) << 5) + slot;
            const int rp = idx / cgw, cp = (x % cgroups) * cgw + (idx - rp * cgw);
            return rp < nbm ? rp * nbn + cp : ntiles;
        }
        if (deal_last && round == full_rounds) {
            const int slot = bid >> 3, j = ((slot / nbn) * 8 + (bid & 7)) * nbn + slot % nbn;
            return j < rem ? round * nwg + j : ntiles;
        }
        return round * nwg + (bid & 7) * per_xcd + (bid >> 3);
    };

    // DYNAMIC tile schedule (p.sched != NULL: eight zeroed int32 counters, one per XCD). XCD x owns the contiguous tile range
    // [xb, xb + xc); a workgroup starts on tile xb + (its slot) and draws every further tile from the XCD's counter, so a workgroup
    // that starts late - because another kernel (a second stream's GEMM, a communication kernel) held its CU - simply takes fewer
    // tiles instead of running its fixed share one shift late: with the static walk a foreign kernel costs ~35 % of its active
    // time (tools/contention.py). The draw for tile i+2 is issued by lane 0 of wave 0 at the top of tile i+1 ... i.e. one tile
    // ahead of its use, and handed to the other waves through LDS at the K loop's last barrier.
    constexpr bool dyn = DYN;                  // (the launcher picks the instantiation by p.sched != NULL)
    const int xq = ntiles >> 3, xr = ntiles & 7;
    const int xb = (bid & 7) * xq + min(bid & 7, xr), xc = xq + ((bid & 7) < xr ? 1 : 0);
    const bool drawer = dyn && wave == 0;                          // wave-uniform (constant false without DYN)
    volatile VDA_LDS_AS int* const sched_slot =
        (volatile VDA_LDS_AS int*)(smem + 3 * A_BYTES + BN * ROW_BYTES + NW * 1024);      // W slot 1, past the statistics slices

    // ---- per-lane DMA sources. A piece is 8 rows x 128 B; lane -> (row lrow of the piece, LDS chunk lane & 7).
    // The swizzled source chunk ((lane&7) ^ ((row>>1)&7)) does not depend on the piece index (pieces are 8 rows
    // apart, the swizzle has period 16 rows and the wave stride is 64 rows), so it is one lane constant.
    // Dense / W offsets are recomputed per K tile from the tile origin (3 VALU ops per piece) instead of being
    // kept in registers; only the conv row decomposition (two divisions per piece) is cached.
    const int lrow = lane >> 3;
    const int src_chk = ((lane & 7) ^ ((((wave & 1) << 2) + (lrow >> 1)) & 7)) * 8;     // halves
    // conv: per A piece the element offset of the window's top-left tap (+ the lane's swizzled chunk) and a 9-bit mask of the taps
    // that fall inside the image. The gather itself is a bounds-checked buffer load: a lane whose tap is padding asks for an
    // offset past the descriptor's range and the hardware writes zeros into LDS - no zero page, no 64-bit pointer select, and
    // per K tile the address is one scalar (tap offset + channel slice) added to a lane constant, like the dense operand's.
    int a_base[AJ], a_mask[AJ];
    int tm0 = 0, tn0 = 0;             // origin of the tile being staged
    [[maybe_unused]] const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.A), 0, AMODE == VDA_A_CONV3X3 ? (unsigned)(p.cB * p.cH * p.cW * p.cCin) * 2u : 0u, 0x00020000);
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
                const int iy0 = oy * p.cStride - 1, ix0 = ox * p.cStride - 1;
                a_base[j] = ((b * p.cH + iy0) * p.cW + ix0) * p.cCin + src_chk;         // only used for taps the mask admits
                int vy = 0, vx = 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    vy |= ((unsigned)(iy0 + t) < (unsigned)p.cH ? 1 : 0) << (3 * t);     // bit 3*ky: row iy0 + ky is inside
                    vx |= ((unsigned)(ix0 + t) < (unsigned)p.cW ? 1 : 0) << t;           // bit kx
                }
                a_mask[j] = ok ? vx * vy : 0;                                            // bit 3*ky + kx
            }
        }
    };

    // A pieces j = 2h, 2h+1 of a wave lie in rows [128h, 128h + 128): the half read by wave group h.
    // conv: which 3x3 tap and which 64-channel slice K tile kt is. A division by the runtime channel count: used at the tile
    // seams only - inside the K loop the pair is advanced incrementally (a ~40-instruction dependent chain in a load section
    // outlasts the partner's MFMA section: the 8-phase conv was 16 % slower than the one-barrier kernel because of it).
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
    auto stage_a = [&](int kt, int j0, char* abuf) {
        const int k0 = kt * BK;
        if constexpr (AMODE == VDA_A_DENSE) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = j0 + jj;
                if (j < AJ) {                    // (192-row tile: three pieces, issued 2 + 1)
                    const int m = min(tm0 + (wave + NW * j) * 8 + lrow, p.M - 1);
                    glds16((const h16*)p.A + (size_t)(unsigned)(m * p.lda + src_chk + k0), abuf + (wave + NW * j) * 1024);
                }
            }
        } else {
            const int tap = cv_tap, ci0 = cv_ci0;            // set by tap_of(kt) / tap_next(): wave-uniform (scalar registers)
            const int ky = (tap * 11) >> 5, kx = tap - ky * 3;   // tap / 3 for tap = 0..8
            const int toff = (ky * p.cW + kx) * p.cCin + ci0, bit = 1 << tap;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = j0 + jj;
                const unsigned voff = (a_mask[j] & bit) ? (unsigned)(a_base[j] + toff) * 2u : 0xFFFFFFF0u;   // out of range = zeros
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (VDA_LDS_AS void*)(abuf + (wave + NW * j) * 1024), 16, voff, 0, 0, 0);
            }
        }
    };
    auto stage_w = [&](int kt, char* wbuf, int j0 = 0, int j1 = BN / 8 / NW) {
        const int k0 = kt * BK;
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            const int n = min(tn0 + (wave + NW * j) * 8 + lrow, p.N - 1);
            glds16((const h16*)p.W + (size_t)(unsigned)(n * p.K + src_chk + k0), wbuf + (wave + NW * j) * 1024);
        }
    };

    // fragment addressing: row = lane & 15 inside a 16-row subtile, k-chunk = 4*ks + (lane >> 4)
    const int frow = lane & 15, fh = lane >> 4, fsw = (lane >> 1) & 7;     // ((row >> 1) & 7) with row = subtile*16 + frow
    const int a_off = (wm * WTM + frow) * ROW_BYTES, w_off = (wn * WTN + frow) * ROW_BYTES;
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
    // epilogue staging (8 waves x 8 KiB) = A slots 1 and 2: free once the K loop's reads are done, while A slot 0 / W slot 0
    // already receive the next tile's first K tile
    char* stg = smem + A_BYTES + wave * 8192;

    auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

    // raw barrier: no vmcnt(0) (LDS-DMA stays in flight across it); the empty asm statements keep LDS accesses on their
    // side of it at IR level (the intrinsic itself is not a memory operation), sched_barrier pins the machine schedule
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };

    // The accumulators START at the bias (MFMA C layout: register e of subtile (i, j) is column 16j + 4(lane >> 4) + e of
    // the wave's 64): the epilogue then has no bias add and no column-constant load in front of it. The fragment of the
    // NEXT tile is fetched with that tile's first K tile, inside the current epilogue.
    f32x4 nbias[NJ];
    auto load_bias = [&](int t) {
        const int bnn = t % nbn;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = bnn * BN + wn * WTN + j * 16 + (lane >> 4) * 4;
            // (LayerNorm-folded epilogues add their constant AFTER the row scaling: the accumulators start at zero)
            nbias[j] = (!vda_gemm::is_ln_epi<EPI> && p.bias != nullptr && n + 3 < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    // LayerNorm-folded epilogues: the (mean, rstd) pairs of the wave's WTM rows (lane -> rows lane, lane + 64) are fetched with
    // the tile's first K tile, ride through the K loop in 4 VGPRs and are parked in the wave's slice of W slot 1 (idle during the
    // epilogue) when it starts: the epilogue's row groups then read them from LDS instead of waiting on a global load each
    // (qkv: +11 us per launch with per-group global loads). The split-residual epilogue uses the same slice the other way:
    // its partial row statistics are collected there and leave as two 512-byte stores per wave tile.
    constexpr bool LN_EPI = vda_gemm::is_ln_epi<EPI>;
    constexpr bool RT_F32OUT = vda_gemm::RowTraits<EPI>::f32_out || EPI == VDA_EPI_SCALE_RES_SPLIT;      // epilogues that do not go through store8h
    constexpr bool STAT_LDS = LN_EPI || EPI == VDA_EPI_SCALE_RES_SPLIT;
    static_assert(!STAT_LDS || (WTM <= 128 && W_BYTES >= NW * 1024), "statistics slice: 1 KiB per wave in W slot 1");
    char* const sst = smem + W_BASE + W_BYTES + wave * 1024;
    float2 nstat[2] = {float2{0.f, 0.f}, float2{0.f, 0.f}};
    auto load_stats = [&](int t) {
        if constexpr (LN_EPI) {
            const int r0 = (t / nbn) * BM + wm * WTM + lane;
            nstat[0] = *reinterpret_cast<const float2*>(p.stats + 2 * (size_t)min(r0, p.M - 1));
            if constexpr (WTM > 64) nstat[1] = *reinterpret_cast<const float2*>(p.stats + 2 * (size_t)min(r0 + 64, p.M - 1));
        }
    };

    int tile = dyn ? ((bid >> 3) < xc ? xb + (bid >> 3) : ntiles) : tile_of(0);
    if (tile >= ntiles) return;                        // uniform per workgroup
    // Start stagger (VDA_GEMM_STAGGER=1, OFF by default). All workgroups run the same tile time, so their epilogues hit HBM
    // together (a plain fp16 epilogue stores at 7.8 TB/s aggregate: it is bandwidth-bound only because it is synchronised).
    // Workgroups that sit out the last round can start up to 3/4 of a tile time late, in four phases, for free - the launch ends
    // with the others' last tile - and take their epilogues out of phase: -4..-6 us per qkv / fc1 / fc2 launch on some boxes,
    // nothing on others, and it costs L2 sharing (PMC: +40 % fetched bytes, the workgroups of an XCD no longer read the same W
    // and A tiles at the same moment) and 2 % with two clips in flight (a sleeping workgroup holds a CU the other stream's
    // kernel could use). Tile time estimate: 1.6 us per K tile + 8 us, at ~2.1 GHz, in 64-cycle sleep units.
    if (!dyn && !((p.relu_in >> 8) & 16) && rem > 0 && full_rounds > 0 && tile_of(full_rounds) >= ntiles) {       // switch off: variant 5 + 16 * 16
        // phase by slot: the nbn workgroups of an XCD that share an A row panel land in DIFFERENT phases (keeping them in phase -
        // VDA_GEMM_STAGGER=2 - is slower than no stagger at all: it is those neighbours' epilogues that collide)
        const int q = ((p.relu_in >> 8) & 32) ? ((bid >> 3) / (nbn <= 8 ? nbn : 8)) & 3 : (bid >> 3) & 3;
        const int units = (nt * 52 + 260) * q / 4;
        for (int i = 0; i < units; i += 120) __builtin_amdgcn_s_sleep(120);
    }
    load_bias(tile);
    load_stats(tile);
    set_sources(tile);
    tap_of(0);
    stage_a(0, 0, smem);
    stage_a(0, 2, smem);
    stage_w(0, smem + W_BASE);
    // In-kernel stamps (tools/gemm_stamps.py; vda_gemm_set_variant(5 + 16 * 2)): thread 0 of every workgroup writes the shader clock at
    // tile start / K-loop start / K-loop end / tile end into the (otherwise unused) pos operand.
    const bool stamp = EPI != VDA_EPI_PATCH_F32 && ((p.relu_in >> 9) & 1) && p.pos != nullptr && tid == 0;
    long long* stamps = (long long*)p.pos;
    // Clock stamps (vda_gemm_set_debug bit 0; MI355X_MICROARCH.md "DVFS give-back" item 6): thread 0 of every workgroup writes
    // (s_memtime = shader clock, s_memrealtime = 100 MHz) at tile start / K-loop start / K-loop end / tile end of its first 16 tiles
    // into pos, [workgroup][tile][4][2] int64. Diagnostic only: the buffer is read by nothing, no output depends on it, and a build
    // without the flag executes none of it (the flag is a wave-uniform scalar test).
    const bool stamp2 = EPI != VDA_EPI_PATCH_F32 && EPI != VDA_EPI_SCALE_RES_SPLIT && (dbg2 & 1) && p.pos != nullptr && tid == 0;
    auto clk = [&](int round, int which) {
        if (stamp2 && round < 16) {
            long long* q = (long long*)p.pos + ((size_t)(bid * 16 + round) * 4 + which) * 2;
            q[0] = (long long)__builtin_amdgcn_s_memtime();
            q[1] = (long long)__builtin_amdgcn_s_memrealtime();
        }
    };
    for (int round = 0; tile < ntiles; ++round) {
        // The draw's result lives INSIDE one iteration (issued at the tile's top, consumed behind its K loop). Declared outside the
        // loop it was a loop-carried VGPR defined by a returning atomic, and hipcc then parks an `s_waitcnt vmcnt(0)` on the loop's
        // back edge - in every build, dynamic schedule or not: each tile ended by draining its whole store burst AND the next
        // tile's prefetched K tile 0 (~4 k cycles per tile; found with tools/gemm_ramp.py + the ISA, round 4).
        int fetched = 0;
        // the draw: index (inside the XCD's range) of the tile after next; the youngest vector-memory operation of wave 0 when issued
        auto draw = [&]() {
            // (built with -mllvm -amdgpu-atomic-optimizer-strategy=None, build.py: the optimizer would broadcast the uniform result at
            // once, behind an s_waitcnt vmcnt(0) that also drains the LDS-DMA just issued; as a plain returning atomic the compiler waits
            // for it where `fetched` is first read - after the K loop - and keeps every copy of it behind that wait)
            if constexpr (DYN) {
                if (drawer && lane == 0) fetched = __hip_atomic_fetch_add(p.sched + (bid & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        if (stamp) stamps[(bid * 16 + (round & 15)) * 4 + 0] = __builtin_readcyclecounter();
        clk(round, 0);
        const int bm = tile / nbn, bn = tile - bm * nbn;
        const int m0 = bm * BM, n0 = bn * BN;

        f32x4 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = nbias[j];
        // ReLU on the conv activation operand: in the LOAD section (after the fragments landed, before the barrier), so it
        // runs under the partner wave's MFMA section instead of delaying this wave's
        auto relu_a = [&](AF&) {};               // (the ReLU now rides inside the MFMA cluster, see mma)
        auto mma = [&](int half, AF& fa, const WF& fw) {
            if constexpr (SCHED != 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MH; ++i) {
                // conv ReLU on the activation fragment just before its 4 MFMAs: 4 packed max per fragment ride in the MFMA gaps
                if constexpr (AMODE == VDA_A_CONV3X3) fa.a[i] = __builtin_elementwise_max(fa.a[i], relu_thr);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (half == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw.w[j], fa.a[i], acc[i][j], 0, 0, 0);
                    else acc[MH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw.w[j], fa.a[i], acc[MH + i][j], 0, 0, 0);
                }
            }
            if constexpr (SCHED != 2) __builtin_amdgcn_s_setprio(0);
        };

        // K tile 0 of this tile (A slot 0, W slot 0) was issued before the previous tile's epilogue (or above).
        AF fa;
        WF fw;
        if (round > 0) set_sources(tile);       // recomputed rather than kept live across the epilogue
        if (nt > 1) {                           // K tile 1 -> A slot 1 (the epilogue staging area: released by the tile-end barrier)
            tap_of(1);
            stage_a(1, 0, smem + A_BYTES);
            stage_a(1, 2, smem + A_BYTES);
            if constexpr (SCHED == 1) {
                stage_w(1, smem + W_BASE + W_BYTES, 0, WH);         // the second half of W's rows of K tile 1 follows in phase 1
                draw();
                if (drawer) vm_wait_keep<AJ + WH + 1>();            // (the draw stays in flight with K tile 1)
                // (This wait also drains the previous epilogue's stores - vmcnt retires in order and counts them. Round 4 tried a
                // count that leaves the stores issued after the K tile 0 prefetch in flight, vmcnt(AJ + WH + 12): the prologue
                // fell from 3.9 k to 0.8 k cycles and the first K tiles slowed by as much - the store burst has to drain through
                // HBM either way; fc1 414.8 us against 404.0 with the full drain, in one process. Not kept.)
                else vm_wait_keep<AJ + WH>();
            } else {
                stage_w(1, smem + W_BASE + W_BYTES);
                draw();
                if (drawer) vm_wait_keep<AJ + WJ + 1>();
                else vm_wait_keep<AJ + WJ>();                             // K tile 0 landed (mine); K tile 1 stays in flight
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            draw();
        }
        bar();                                  // ... and everyone's
        if (stamp) stamps[(bid * 16 + (round & 15)) * 4 + 1] = __builtin_readcyclecounter();
        clk(round, 1);
        if (grp == 1) bar();                    // waves 4..7 run one barrier behind from here to the end of the K loop

        int sa = 0;                             // A slot of K tile kt (kt % 3)
        tap_of(2);                              // conv: (tap, channel slice) of the next K tile to stage, advanced per K tile
        for (int kt = 0; kt < nt; ++kt) {
            // (diagnostic, vda_gemm_set_debug bit 2: the shader clock at the top of each of the first 32 K tiles of the workgroup's
            // SECOND tile, behind the tile stamps: where the K loop's ramp-up goes)
            if (stamp2 && (dbg2 & 4) && round == 1 && kt < 32) ((long long*)p.pos)[(size_t)nwg * 16 * 4 * 2 + bid * 32 + kt] = (long long)__builtin_amdgcn_s_memtime();
            const char* ab = smem + sa * A_BYTES;
            const char* wb = smem + W_BASE + (kt & 1) * W_BYTES;
            const int sa2 = sa == 0 ? 2 : sa - 1;                   // (kt + 2) % 3
            char* ab2 = smem + sa2 * A_BYTES;
            const bool more = kt + 2 < nt;
            if constexpr (SCHED == 1) {
                // DEFAULT: 2 DMA pieces per phase - W(kt+1) rows 128.. | A(kt+2) rows ..127 | A(kt+2) rows 128.. | W(kt+2) rows ..127 -
                // and vmcnt(6): even issue load, 3 phases of minimum cover. In-process A/B vs the 2/2/0/4 + vmcnt(8) form below
                // (SCHED 0, 4 phases of cover): -0.6..-3 % on every encoder shape.
                char* wb1 = smem + W_BASE + ((kt + 1) & 1) * W_BYTES;
                read_w(wb, 0, fw);
                read_a(ab, 0, 0, fa);
                if (kt + 1 < nt) stage_w(kt + 1, wb1, WH, WJ);
                lgkm0();
                relu_a(fa);
                bar();
                mma(0, fa, fw);
                bar();
                read_a(ab, 0, 1, fa);
                if (more) stage_a(kt + 2, 0, ab2);
                lgkm0();
                relu_a(fa);
                bar();
                mma(1, fa, fw);
                bar();
                read_w(wb, 1, fw);
                read_a(ab, 1, 0, fa);
                if (more) {
                    stage_a(kt + 2, 2, ab2);
                    tap_next();
                }
                lgkm0();
                relu_a(fa);
                bar();
                mma(0, fa, fw);
                bar();
                read_a(ab, 1, 1, fa);
                if (more) {
                    stage_w(kt + 2, const_cast<char*>(wb), 0, WH);
                    vm_wait_keep<AJ + WH>();
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                lgkm0();
                relu_a(fa);
                bar();
                mma(1, fa, fw);
                bar();
            } else {
                // phase 1: k-step 0, rows lo
                read_w(wb, 0, fw);
                read_a(ab, 0, 0, fa);
                if (more) stage_a(kt + 2, 0, ab2);
                lgkm0();
                relu_a(fa);
                bar();
                mma(0, fa, fw);
                bar();
                // phase 2: k-step 0, rows hi
                read_a(ab, 0, 1, fa);
                if (more) {
                    stage_a(kt + 2, 2, ab2);
                    tap_next();
                }
                lgkm0();
                relu_a(fa);
                bar();
                mma(1, fa, fw);
                bar();
                // phase 3: k-step 1, rows lo (last read of this K tile's W slot)
                read_w(wb, 1, fw);
                read_a(ab, 1, 0, fa);
                lgkm0();
                relu_a(fa);
                bar();
                mma(0, fa, fw);
                bar();
                // phase 4: k-step 1, rows hi (last read of this K tile's A slot); W(kt+2) into the W slot just released
                read_a(ab, 1, 1, fa);
                if (more) {
                    stage_w(kt + 2, const_cast<char*>(wb));
                    vm_wait_keep<AJ + WJ>();                                  // K tile kt+1 landed; A(kt+2), W(kt+2) stay in flight
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                lgkm0();
                relu_a(fa);
                bar();
                mma(1, fa, fw);
                bar();
            }
            sa = sa == 2 ? 0 : sa + 1;
        }
        if constexpr (DYN) {
            if (drawer) {                       // hand the drawn index to the other waves (W slot 1 is idle: every read of the K loop is done)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the draw has landed (nothing else is in flight here)
                if (lane == 0) *sched_slot = fetched + per_xcd;
                lgkm0();
            }
        }
        if (grp == 0) bar();                    // re-align the two groups: every wave is past its last MFMA section's reads
        if (stamp) stamps[(bid * 16 + (round & 15)) * 4 + 2] = __builtin_readcyclecounter();
        clk(round, 2);
        // Nobody reads the pipeline buffers any more. The NEXT tile's first K tile is issued after the first 32-row block of
        // the epilogue (below): early enough that the rest of the epilogue covers its HBM/L2 latency, late enough that the
        // epilogue's own first loads (column constants, residual rows) do not queue behind it on the in-order vmcnt.
        int next = tile_of(round + 1);
        if constexpr (DYN) {
            const int idx = __builtin_amdgcn_readfirstlane(*sched_slot);
            next = idx < xc ? xb + idx : ntiles;
        }
        const int dbg = __builtin_amdgcn_readfirstlane((p.relu_in >> 8) & 0xff);      // A/B switches (vda_gemm_set_variant(5 + 16 * flags))
        if ((dbg & 1) && next < ntiles) {
            load_bias(next);
            load_stats(next);
            set_sources(next);
            tap_of(0);
            stage_a(0, 0, smem);
            stage_a(0, 2, smem);
            stage_w(0, smem + W_BASE);
        }

        // ---- epilogue. Accumulator register e of subtile (i,j) is row m = lane & 31, column 8*(e>>2) + 4*(lane>>5) + (e&3):
        // stored straight from this layout a wave instruction would touch 32 rows x 16 B (32 partial lines). Instead each
        // wave transposes one 32-row x 64-column fp32 block at a time through its own 8 KiB of (idle) pipeline buffer 1
        // (16-byte chunks XOR-swizzled by row: conflict-free both ways) and then owns whole row segments: bias / LayerScale /
        // residual reads and the output stores become contiguous 16-byte accesses covering full 128-byte lines.
        const int bm0 = m0 + wm * WTM, bn0 = n0 + wn * WTN;
        const bool interior = bm0 + WTM <= p.M && bn0 + WTN <= p.N;       // wave-uniform
        const bool nt_out = !RT_F32OUT && __builtin_amdgcn_readfirstlane((p.relu_in >> 24) & 1) != 0;      // set by vda_gemm_f16 for outputs past the caches
        if constexpr (LN_EPI) {
            *reinterpret_cast<float2*>(sst + lane * 8) = nstat[0];
            if constexpr (WTM > 64) *reinterpret_cast<float2*>(sst + (64 + lane) * 8) = nstat[1];
        }
        {
            using RT = vda_gemm::RowTraits<EPI>;
            constexpr int NC = RT::NC, RR = RT::f32_out ? 8 : 4;            // columns per lane, row groups per 32-row block
            // lane -> (row inside a group, column block): fp32 out: 4 rows x 16 lanes x 4 cols; fp16 out: 8 rows x 8 lanes x 8 cols
            const int lrow_e = RT::f32_out ? (lane >> 4) : (lane >> 3);
            const int c0 = RT::f32_out ? (lane & 15) : 2 * (lane & 7);      // first 16-byte (4-column) chunk of the lane
            const int en = bn0 + c0 * 4;
            const bool geglu_idle = (EPI == VDA_EPI_GEGLU_F16) && (c0 & 7) >= 4;   // gate lanes only feed their value lanes
            vda_gemm::ColConst<NC> cc;
            vda_gemm::load_col_const<EPI, NC, false>(p, en, cc);
            // The column constants are loaded under lane / pointer conditions, so hipcc's waitcnt pass sees them as "possibly still in
            // flight" on some path into EVERY later join (each 32-row block is an interior | guarded diamond) and parks an
            // `s_waitcnt vmcnt(0)` at the first use in every block and at the top of the next tile: each one drained the whole store
            // burst issued so far, and the first also the prefetched K tile 0 (33 vmcnt waits in the fc1 kernel's ISA, 7 with this).
            // Consuming them HERE, on the one path every lane takes, makes the pass wait once - where only these loads are in flight.
#pragma unroll
            for (int i2 = 0; i2 < NC; ++i2) asm volatile("" ::"v"(cc.bias[i2]), "v"(cc.gamma[i2]), "v"(cc.gbias[i2]));
            constexpr int RG = RT::f32_out ? 4 : 2;                          // rows per row group
            vda_gemm::RowAux carry[RG];                                      // the next block's first group, loaded a group early
            // LayerNorm-folded epilogues: every (mean, rstd) pair this lane will need (one per row step: MI/2 blocks x RR steps) is
            // read from the wave's statistics slice NOW, before the next tile's K tile 0 is put in flight. A read of that slice with an
            // LDS-DMA pending gets an `s_waitcnt vmcnt(0)` from hipcc (it cannot tell the slice from the DMA's destination; the
            // staging reads, at constant offsets from another base, do not): one per 32-row block, each draining every store issued so
            // far and, in block 1, the K tile 0 prefetch itself (ISA, round 4). 32 VGPRs, inside what the dying accumulators free.
            float2 lnst[LN_EPI ? MI / 2 : 1][LN_EPI ? RR : 1];
            if constexpr (LN_EPI) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (this lane's slice writes above: same wave, program order)
#pragma unroll
                for (int b = 0; b < MI / 2; ++b)
#pragma unroll
                    for (int r = 0; r < RR; ++r) lnst[b][r] = *reinterpret_cast<const float2*>(sst + (b * 32 + r * (32 / RR) + lrow_e) * 8);
            }
#pragma unroll
            for (int i = 0; i < MI / 2; ++i) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        // accumulator (mi = 2i+h2, nj = j): row = h2*16 + (lane & 15) of the 32-row block, columns 16j + 4*(lane>>4) + e
                        const int row = h2 * 16 + (lane & 15), c = j * 4 + (lane >> 4);
                        *reinterpret_cast<f32x4*>(stg + row * 256 + ((c ^ (row & 15)) << 4)) = acc[2 * i + h2][j];
                    }
                // The transposition is a cross-LANE exchange inside one wave: the hardware runs a wave's LDS ops in order,
                // but the compiler must not reorder the reads below across the writes above (it only reasons per thread).
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // Row groups are processed RG at a time: phase 1 issues every row-dependent global load of the group,
                // phase 2 reads the transposed accumulators back, finishes and stores.
                // Row-dependent loads (residuals, pos-embed) run ONE ROW GROUP AHEAD of the stores: vmcnt retires in order and
                // counts stores, so a load issued after a group's stores cannot complete before them - issued before, it can.
                auto row_groups = [&](auto guard, auto nt_tag) {
                    constexpr bool GUARD = decltype(guard)::value;
                    constexpr bool NTS = decltype(nt_tag)::value;              // non-temporal output stores (gemm_epilogue.h, store8h)
                    constexpr int NG = RR / RG;                                  // row groups per 32-row block
                    auto load_group = [&](int blk, int r0, vda_gemm::RowAux (&ax)[RG]) {
                        if constexpr (LN_EPI) {
#pragma unroll
                            for (int q = 0; q < RG; ++q) {
                                ax[q].s0 = lnst[blk][r0 + q].x;
                                ax[q].s1 = lnst[blk][r0 + q].y;
                            }
                        } else if (!geglu_idle) {
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
                                vda_gemm::finish_row4<EPI, GUARD, false>(p, bm0 + i * 32 + row, en, a, cc, aux[g & 1][q]);
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
                                if (!geglu_idle) vda_gemm::finish_row8<EPI, GUARD, false, NTS>(p, bm0 + i * 32 + row, en, v, gt, cc, aux[g & 1][q]);
                                if constexpr (EPI == VDA_EPI_SCALE_RES_SPLIT)        // the row's 8 lanes hold the same pair (rows past M: never stored)
                                    *reinterpret_cast<float2*>(sst + (i * 32 + row) * 8) = float2{aux[g & 1][q].o0, aux[g & 1][q].o1};
                            }
                        }
                    }
                };
                // interior wave tiles (all but the matrix's last row / column of tiles) take the branch-free form
                // (three copies of the block: interior with ordinary stores, interior with non-temporal ones - a wave-uniform choice made
                // once per launch by the dispatcher from the output's size - and the guarded form for the matrix's edge tiles)
                if (interior) {
                    if (nt_out) row_groups(std::false_type{}, std::true_type{});
                    else row_groups(std::false_type{}, std::false_type{});
                } else {
                    row_groups(std::true_type{}, std::false_type{});
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads of this block done before the next block's writes
                // Epilogues with row-dependent loads (residual, pos-embed) prefetch after their LAST block instead: with an LDS-DMA
                // in flight hipcc waits vmcnt(0) - every store included - at each of those loads (dbg bit 2 switches this off)
                // (the LayerNorm-folded epilogues' 8-byte row statistics do not count: measured -20 us on fc1 with the early prefetch)
                constexpr bool ROW_AUX = EPI == VDA_EPI_SCALE_RES_F32 || EPI == VDA_EPI_SCALE_RES_F32_H || EPI == VDA_EPI_RES_F16 || EPI == VDA_EPI_PATCH_F32 ||
                                         EPI == VDA_EPI_SCALE_RES_SPLIT;
                const int pf_block = ((ROW_AUX && !(dbg & 4)) || (dbg & 8)) ? MI / 2 - 1 : 0;          // dbg 8: A/B, late for every epilogue
                if (i == pf_block && next < ntiles && !(dbg & 1)) {
                    load_bias(next);
                    load_stats(next);
                    set_sources(next);
                    tap_of(0);
                    stage_a(0, 0, smem);
                    stage_a(0, 2, smem);
                    stage_w(0, smem + W_BASE);
                }
            }
        }
        if constexpr (EPI == VDA_EPI_SCALE_RES_SPLIT) {
            // partial statistics of the wave tile: [column block][row], rows bm0 + lane and bm0 + 64 + lane: two 512-byte runs
            lgkm0();
            float* dst = p.stats + ((size_t)(bn0 >> 6) * (p.stats_ld ? p.stats_ld : p.M) + bm0) * 2;
#pragma unroll
            for (int hh = 0; hh < (WTM + 63) / 64; ++hh)
                if (hh * 64 + lane < WTM && bm0 + hh * 64 + lane < p.M && bn0 < p.N)
                    *reinterpret_cast<float2*>(dst + (hh * 64 + lane) * 2) = *reinterpret_cast<const float2*>(sst + (hh * 64 + lane) * 8);      // (wave tiles past N hold nothing)
        }
        // Every wave is done READING its staging slice before the next tile's K tile 1 lands in it: LDS ordering only, so a raw
        // barrier (a __syncthreads here would also drain the stores and the prefetched K tile 0).
        lgkm0();
        bar();
        if (stamp) stamps[(bid * 16 + (round & 15)) * 4 + 3] = __builtin_readcyclecounter();
        clk(round, 3);
        tile = next;
    }
}

template <int BN, int AMODE, int EPI, int SCHED = 1, int BM = 256>
int launch256(const vda_gemm_args& a, hipStream_t s) {
    constexpr int smem = 3 * BM_MAX * ROW_BYTES + 2 * BN * ROW_BYTES;
    static_assert(smem <= 160 * 1024, "LDS budget");
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    const int ntiles = nbm * nbn;
    if constexpr (SCHED == 1) {
        if (a.sched != nullptr) {                 // dynamic tile draw: its own instantiation (see DYN above)
            static VdaKernelDeviceState dyn_state;
            const int ncu = vda_prepare_kernel(reinterpret_cast<const void*>(&gemm8p_kernel<BN, AMODE, EPI, SCHED, BM, true>), smem, dyn_state);
            if (ncu < 0) return 2;
            hipLaunchKernelGGL((gemm8p_kernel<BN, AMODE, EPI, SCHED, BM, true>), dim3(ntiles < ncu ? (ntiles + 7) / 8 * 8 : ncu), dim3(NT), smem, s, a);
            VDA_LAUNCH_CHECK();
            return 0;
        }
    }
    static VdaKernelDeviceState dev_state;
    const int num_cu = vda_prepare_kernel(reinterpret_cast<const void*>(&gemm8p_kernel<BN, AMODE, EPI, SCHED, BM, false>), smem, dev_state);
    if (num_cu < 0) return 2;
    const int grid = ntiles < num_cu ? (ntiles + 7) / 8 * 8 : num_cu;     // one persistent workgroup per CU
    hipLaunchKernelGGL((gemm8p_kernel<BN, AMODE, EPI, SCHED, BM, false>), dim3(grid), dim3(NT), smem, s, a);
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

// 192-row tiles (dense, 256 columns): the epilogues of the encoder GEMMs a row split applies to
template <int BN>        // (a template only so that the kernels are instantiated in the one translation unit that calls it)
int launch_dense_bm192(const vda_gemm_args& a, hipStream_t s) {
    switch (a.epilogue) {
        case VDA_EPI_BIAS_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_BIAS_F16, 1, 192>(a, s);
        case VDA_EPI_SCALE_RES_F32: return launch256<BN, VDA_A_DENSE, VDA_EPI_SCALE_RES_F32, 1, 192>(a, s);
        case VDA_EPI_SCALE_RES_SPLIT: return launch256<BN, VDA_A_DENSE, VDA_EPI_SCALE_RES_SPLIT, 1, 192>(a, s);
        case VDA_EPI_LN_BIAS_F16: return launch256<BN, VDA_A_DENSE, VDA_EPI_LN_BIAS_F16, 1, 192>(a, s);
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

}  // namespace vda_gemm8p
