// 8-phase MFMA GEMM for the large-batch encoder GEMMs:  C[m][n] = sum_k A[m][k] * W[n][k]
//
// One 512-thread workgroup per CU computes 256(m) x BN(n) tiles (BN = 256 or 192) in K-tiles of 64.  The structure
// follows the "8-phase, counted vmcnt" schedule of the CDNA4 guide, rebuilt for this model's shapes:
//
//  * operand halves.  A K-tile in LDS = four independently staged slots: X0, X1 (activation rows 0-127 / 128-255 of the
//    tile) and W0, W1 (weight rows: 128 + (BN - 128)).  Two K-tiles are resident (ring of 2 x 4 slots, 128 / 112 KiB).
//    Every wave owns a 128(m) x BN/4(n) output block made of 64 rows from EACH X half and BN/8 (BN = 256) rows from each W
//    half, so its accumulators split into four quadrants q(hm, hn) = X-half hm x W-half hn, and a quadrant needs exactly
//    one X slot and one W slot.
//  * 2 segments per K-tile (round 3; round 2 ran four phases of one quadrant each -- same speed to 1 %, 8 barriers and 4 load sections
//    per K-tile, 15-25 registers more, a first-K-tile special case; profiles/gemm8_sched_r3.txt):
//        LA: read X0, W0, W1 fragments (16 / 14 ds_read_b128) | DMA X1 <- K-tile t+1 (other buffer) | counted vmcnt | lgkmcnt(0) | barrier
//        MA: MFMA q00, q01 (32 / 24)                                                                                              | barrier
//        LB: read X1 fragments (8)      | DMA X0, W0, W1 <- K-tile t+2 (this buffer)                | counted vmcnt | lgkmcnt(0) | barrier
//        MB: MFMA q10, q11 (32 / 24)                                                                                              | barrier
//    Slots are restaged by global_load_lds (16 B per lane); every load section ends with ONE counted wait that leaves exactly the
//    pieces issued since the previous section's wait in flight (6 + NW1), so a slot has 1 - 1.5 K-tiles of MFMAs to land.
//  * two wave groups (waves 0-3 / 4-7 = the two waves of every SIMD) run the same stream ONE BARRIER APART: while one wave of a
//    SIMD streams its MFMAs, its SIMD partner reads fragments and issues the DMA of its segment.
//  * ordering rules the schedule is built on (MI355X_MICROARCH.md, "Two waves per SIMD", item 7):
//      WAR: X0 / W0 / W1 of buffer B are read in LA(t) by group 0 and one segment later by group 1; both are past them at the barrier
//           in front of group 0's LB(t), where their restage starts (group 1 restages in ITS LB(t), later still).  X1 of buffer B is
//           read in LB(t) / one segment later and restaged in LA(t+1), behind the barrier that ends group 1's LB(t).  Every load
//           section ends with lgkmcnt(0) before its barrier.
//      RAW: LA(t) retires X1(t) (issued in LA(t-1), read in LB(t)); LB(t) retires X0 / W0 / W1 of K-tile t+1 (issued in LB(t-1), read
//           in LA(t+1)): the reading section starts two barriers after group 0's wait and one after group 1's.
//      tile boundary: the last LB of a tile waits one group deeper (it also retires X1 of the next tile's first K-tile), so the first
//           LA of a tile needs no wait -- a counted wait right behind the epilogue would wait for the epilogue's stores.
//  * the LDS image of a slot is lane-linear (global_load_lds writes wave base + lane * 16): the XOR swizzle of the
//    16-byte k-slots is applied to the per-lane SOURCE address and again on the ds_read_b128 fragment reads.
//  * persistent workgroups: the ring runs on across tile boundaries (the next tile's first two K-tiles stream in during
//    the last phases and the epilogue of the current one).  At a tile's end the two wave groups re-align for the epilogue (group 0 waits
//    one barrier) and fall one barrier apart again after it: run one after the other, each wave alone on its SIMD, the two groups'
//    epilogues cost twice their store latency.
//  * W rows are PERMUTED on their way into LDS (free: the source address of a DMA lane is arbitrary) so that the 4 * TI
//    accumulator values a lane holds for one output row are CONSECUTIVE columns: the 16-bit epilogues (qkv, fc1) store
//    straight from registers as 16-byte pieces that complete 128-byte lines -- no LDS round trip, no epilogue barrier.
//  * the residual epilogue (attn.proj / mlp.fc2: + bias + residual planes, two-plane output, LayerNorm row statistics):
//    256 x 192 tiles stage the tile through LDS exactly like gemm.hip's producer epilogue; 256 x 256 tiles take it straight
//    from registers (a row's four lanes are one 64-column statistics granule).  Both replicate gemm.hip's summation tree:
//    bit-identical statistics.
//
//  * 192(m) x 256(n) tiles (G8<256, 192>): X halves of 96 rows (12 DMA pieces: see gemm8_common.h for who issues which), three m-fragments
//    per wave and half.  M of this model is always a multiple of 192 (a crop is 192 tokens) and N = 768 = 3 x 256: mlp.fc2 gets the
//    register-direct residual epilogue and an operand ring that never stops (256 x 192 tiles drain and restart it for the LDS-staged one).
//
// Accumulation order per output element is the same as in gemm.hip (k ascending in steps of 32), so results are
// bit-identical to the 2-phase kernels: that identity is the race screen (tools/gemm8_check.py, tests).
#include <cstdio>

#include <cstdlib>
#include "gemm8_common.h"
#ifndef VP_G8_RESD
#define VP_G8_RESD 1
#endif
#ifndef VP_G8_RESD192
#define VP_G8_RESD192 1
#endif



namespace vp {

namespace {
__device__ __forceinline__ u32x2 lds_tr16(const char* p) {   // ds_read_b64_tr_b16: column i of a [4 keys][16 d] block to lane i of a 16-lane group
    typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h4;
    typedef __attribute__((address_space(3))) h4* lds_h4;
    const h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(p));
    return __builtin_bit_cast(u32x2, v);
}
}  // namespace

// EPI: EPI_BIAS / EPI_BIAS_GELU (16-bit output straight from registers, optional LayerNorm-consumer fold, optional
// 64x64-blocked output) or EPI_BIAS_RESID_LN (two-plane residual stream + row statistics, staged through LDS).
template <class T, int EPI, class C>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(GemmArgs g) {
    constexpr bool RESID = (EPI == EPI_BIAS_RESID_LN);
    // residual epilogue: 256-wide tiles take it straight from registers (a lane's 16 accumulator columns of a row are 16 consecutive
    // output columns and the four lanes of a row are exactly one 64-column statistics granule); 192-wide tiles stage through LDS
    constexpr bool RESID_LDS = RESID && C::BN != 256;
    constexpr bool QKVA = (EPI == EPI_QKV_ATTN);      // attn.qkv of one crop x one head (head dim 80) + the attention core: needs the whole LDS behind the K-loop
    constexpr bool DRAIN = RESID_LDS || QKVA;         // epilogues that drain the operand ring and restart it on the next tile
    static_assert(!QKVA || (C::BM == 192 && C::BN == 256), "the fused qkv + attention epilogue is written for the 192 x 256 tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;          // wave group (stagger) / column block
    const int K = g.K, nk = K >> 6;
    TileWalk tw;
    tw.init(g, C::BM, C::BN);
    if (tw.j0 >= tw.cnt) return;

    // ---- staging addresses: wave-uniform bases + one per-lane byte offset per operand ----
    // piece p (8 rows x 128 B = one DMA wave-instruction) of a slot = LDS rows 8p .. 8p+7; wave w issues p = w (and w + 8)
    const int rip = lane >> 3, pslot = lane & 7;
    const int slog = pslot ^ (((wave & 1) << 2) | (rip >> 1));   // logical k-slot this lane fetches (swizzled image)
    const uint32_t voff_x = g.a_blocked ? (uint32_t)(rip * 128 + slog * 16) : (uint32_t)(rip * K + slog * 8) * 2u;
    // W row permutation: LDS row r of half h holds tile column  wc(r) 16 TI + (rho >> 2) 4 TI + f 4 + (rho & 3),
    // rho = r & 15, f = fragment index of the wave (half 0: (r >> 4) & 1, half 1: 2 + ...), wc(r) = owner wave column: a lane's 4 TI
    // accumulator values of a row are 4 TI consecutive columns (residual epilogues).
    // 16-bit epilogues (SPLIT): column  wc(r) 64 + (f >> 1) 32 + (rho >> 2) 8 + (f & 1) 4 + (rho & 3) -- a lane's 16 values are TWO runs of 8
    // columns, 32 columns apart, so each of its two 16-byte stores of a row lands next to the other three lanes' pieces: a store
    // instruction writes 16 rows x 64 CONTIGUOUS bytes instead of 16 rows x four 16-byte pieces 32 bytes apart (55-63 instead of 75
    // cycles per instruction and CU: tools/store_probe.py, profiles/store_probe_r3.txt).
    constexpr bool SPLIT = !RESID && C::TI == 4;
    const int rho = ((wave & 1) << 3) | rip;
    const uint32_t voff_w = (uint32_t)(((rho >> 2) * (SPLIT ? 8 : 4 * C::TI) + (rho & 3)) * K + slog * 8) * 2u;
    const int wu0 = (wave >> 2) * 16 * C::TI + ((wave >> 1) & 1) * 4;                       // half 0, piece w   (piece w + 8: + 32 TI)
    const int wu1 = SPLIT ? wu0 + 32 : (C::BN == 256) ? wu0 + 8 : (wave >> 1) * 16 * C::TI + 8;   // half 1, piece w   (BN = 256: piece w + 8: + 32 TI)
    const size_t xrow_bytes = g.a_blocked ? 128 : (size_t)K * 2;                            // bytes between consecutive rows of a piece
    const size_t x64 = g.a_blocked ? (size_t)(K >> 6) * 8192 : (size_t)64 * K * 2;          // + 64 rows
    const size_t xkt = g.a_blocked ? 8192 : 128;                                            // + one K-tile
    const char* xb = nullptr;   // current issue tile: X rows m0 + 8 w
    const char* wb = nullptr;   // W rows n0 (column permutation applied by wu0 / wu1 / voff_w)
    auto set_tile = [&](int m0, int n0) {
        const int w8 = C::BM == 256 ? wave * 8 : 0;   // 192-row tiles form the row offset of every piece in issue()
        xb = g.a_blocked ? (const char*)(g.A + ((size_t)(m0 >> 6) * (K >> 6) << 12)) + (size_t)w8 * 128
                         : (const char*)(g.A + (size_t)(m0 + w8) * K);
        wb = (const char*)(g.W + (size_t)n0 * K);
    };
    // DMA of one slot of K-tile kt (of the issue tile) into ring buffer B
    auto issue = [&](int which, int B, int kt, bool force = false) {
        char* dst = smem + B * C::BUF + wave * 1024;
        if ((VP_ABLATE(g) & 1) && !force) return;
        // the per-lane offsets pass through an empty asm: every DMA's 64-bit source address is then formed right here, at its use,
        // instead of eight loop-invariant per-lane pointers being kept live across the phases (which costs 4-11 spilled VGPRs)
        uint32_t vx = voff_x, vw = voff_w;
        asm volatile("" : "+v"(vx), "+v"(vw));
        if (which < 2) {   // X half `which`
            if constexpr (C::BM == 256) {
                const char* src = xb + (size_t)(which * 2) * x64 + (size_t)kt * xkt + vx;
                glds16(src, dst + which * C::XH);
                glds16(src + x64, dst + which * C::XH + 8192);
            } else {   // 96 rows = pieces 0-11: wave w issues piece w; pieces 8-11 go to waves 0-3 (half 0: piece w + 8) / waves 4-7 (half 1: piece w + 4)
                // tile row R (a multiple of 8: a piece never straddles a 64-row block) -> byte offset from the tile's first row (xb carries no wave term here)
                auto xrow = [&](int R) { return g.a_blocked ? (size_t)(R >> 6) * x64 + (size_t)(R & 63) * 128 : (size_t)R * xrow_bytes; };
                const char* src = xb + (size_t)kt * xkt + vx;
                glds16(src + xrow(which * C::XR + wave * 8), dst + which * C::XH);
                if (which == 0) { if (!wr) glds16(src + xrow(64 + wave * 8), dst + 8192); }
                else { if (wr) glds16(src + xrow(C::XR + (wave + 4) * 8), dst + C::XH + 4096); }
            }
        } else if (which == 2) {
            const char* src = wb + ((size_t)wu0 * K + (size_t)kt * 64) * 2 + vw;
            glds16(src, dst + C::OFF_W0);
            glds16(src + (size_t)32 * C::TI * K * 2, dst + C::OFF_W0 + 8192);
        } else {
            const char* src = wb + ((size_t)wu1 * K + (size_t)kt * 64) * 2 + vw;
            glds16(src, dst + C::OFF_W1);
            if (C::NW1 == 2) glds16(src + (size_t)32 * C::TI * K * 2, dst + C::OFF_W1 + 8192);
        }
    };

    // ---- fragment read offsets (bytes inside a slot) ----
    const int frow = lane & 15, fg = lane >> 4;
    const int foff = frow * 128 + ((fg ^ ((frow >> 1) & 7)) << 4);
    const int xoff = wr * (C::XR / 2) * 128 + foff;   // X half h: + j * 2048, kk: ^ 64
    const int w0off = wc * 32 * 128 + foff;           // W half 0: + phi * 2048
    const int w1off = wc * 16 * C::NF1 * 128 + foff;  // W half 1

    constexpr int MJ = C::MJ, TJ = C::TJ;             // m-fragments per X half / per wave
    f32x4 acc[C::TI][TJ];
    // fragment registers: the current X half (both k-halves), W0, W1
    u32x4 xs[MJ][2], fa[2][2], fb[2][2];
    auto rowJ = [](int J) { return (J / MJ) * C::XR + (J % MJ) * 16; };   // tile row of m-fragment J of a wave (+ wr XR/2 + lane row)

    auto zero_acc = [&]() {
#pragma unroll
        for (int f = 0; f < C::TI; ++f)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

#ifndef VP_G8_ABL
#define VP_G8_ABL 0   // diagnosis builds (tools/gemm8_ablate2.py): 1 = no fragment reads after a tile's first K-tile, 2 = no barriers in the
                      // main loop, 4 = no MFMAs
#endif
#define KBAR() do { if (!(VP_G8_ABL & 2)) bar(); } while (0)
#if (VP_G8_ABL & 16)   // cycle stamps around every section of ONE K-tile (K-tile 4 of each workgroup's second tile): tools/gemm8_timeline.py
    unsigned long long sec[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) sec[i] = 0;
    bool stamp_now = false;
#define SEC(i) do { if (stamp_now) sec[i] = __builtin_readcyclecounter(); } while (0)
#else
#define SEC(i) do { } while (0)
#endif
    // one K-tile t in ring buffer B (two segments, see the top of the file): kA / kB = K-tile indices (relative to the ISSUE tile
    // pointers) of the K-tiles whose X1 / X0, W0, W1 are restaged here; swB = switch the issue pointers to the next tile before LB's
    // restage.  MODE 1 = first K-tile of a tile (no wait in LA), 2 = last K-tile of a tile (deeper wait in LB).
    constexpr int NKEEP = C::NKEEP;   // pieces of one LA + one LB issue (BM = 256: 2 + 4 + NW1): what a counted wait leaves in flight
    // the wait that leaves exactly one LB group in flight (last K-tile of a tile, ring start)
    auto wait_lb = [&]() {
        if constexpr (C::BM == 256) wait_vm<C::INFLIGHT>();
        else { if (wr) wait_vm<3 + C::NW1>(); else wait_vm<4 + C::NW1>(); }
    };
    auto ktile = [&](auto Bc, auto Mc, int kA, int kB, bool swB, int nm0, int nn0) {
        constexpr int B = decltype(Bc)::value;
        constexpr int MODE = decltype(Mc)::value;
        const char* sb = smem + B * C::BUF;
        SEC(0);
        // ---------------- LA: X0, W0, W1 | DMA X1(t+1)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) if (MODE == 1 || !(VP_G8_ABL & 1)) fa[p][kk] = *(const u32x4*)(sb + C::OFF_W0 + ((w0off + p * 2048) ^ (kk << 6)));
#pragma unroll
        for (int p = 0; p < C::NF1; ++p)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) if (MODE == 1 || !(VP_G8_ABL & 1)) fb[p][kk] = *(const u32x4*)(sb + C::OFF_W1 + ((w1off + p * 2048) ^ (kk << 6)));
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) if (MODE == 1 || !(VP_G8_ABL & 1)) xs[j][kk] = *(const u32x4*)(sb + C::OFF_X0 + ((xoff + j * 2048) ^ (kk << 6)));
        __builtin_amdgcn_sched_barrier(0);
        SEC(1);
        if constexpr (MODE != 4) issue(1, B ^ 1, kA);
        SEC(2);
        if constexpr (MODE == 4) wait_vm<0>();   // last K-tile of a draining tile: nothing left to fetch, X1 of this K-tile is all that is in flight
        else if constexpr (MODE != 1) wait_vm<NKEEP>();
        wait_lgkm<0>();
        SEC(3);
        KBAR();
        SEC(4);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < MJ; ++j) if (!(VP_G8_ABL & 4)) acc[p][j] = mfma16<T>(fa[p][kk], xs[j][kk], acc[p][j]);
#pragma unroll
            for (int p = 0; p < C::NF1; ++p)
#pragma unroll
                for (int j = 0; j < MJ; ++j) if (!(VP_G8_ABL & 4)) acc[2 + p][j] = mfma16<T>(fb[p][kk], xs[j][kk], acc[2 + p][j]);
        }
        __builtin_amdgcn_s_setprio(0);
        SEC(5);
        KBAR();
        SEC(6);
        // ---------------- LB: X1 | DMA X0, W0, W1 (t+2)
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) if (!(VP_G8_ABL & 1)) xs[j][kk] = *(const u32x4*)(sb + C::OFF_X1 + ((xoff + j * 2048) ^ (kk << 6)));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE < 3) {
            if (swB) set_tile(nm0, nn0);
            issue(2, B, kB);
            issue(0, B, kB);
            issue(3, B, kB);
        }
        SEC(7);
        if constexpr (MODE == 2) wait_lb();
        else if constexpr (MODE == 3) { if (wr) wait_vm<2>(); else wait_vm<1>(); }   // second-to-last K-tile of a draining tile (192-row geometry): LB fetches
                                                                                       // nothing; X0 / W0 / W1 of the last K-tile landed, this LA's X1 pieces (1 / 2 per wave) stay in flight
        else if constexpr (MODE < 3) wait_vm<NKEEP>();
        wait_lgkm<0>();
        SEC(8);
        KBAR();
        SEC(9);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < MJ; ++j) if (!(VP_G8_ABL & 4)) acc[p][MJ + j] = mfma16<T>(fa[p][kk], xs[j][kk], acc[p][MJ + j]);
#pragma unroll
            for (int p = 0; p < C::NF1; ++p)
#pragma unroll
                for (int j = 0; j < MJ; ++j) if (!(VP_G8_ABL & 4)) acc[2 + p][MJ + j] = mfma16<T>(fb[p][kk], xs[j][kk], acc[2 + p][MJ + j]);
        }
        __builtin_amdgcn_s_setprio(0);
        SEC(10);
        KBAR();
        SEC(11);
    };
#undef KBAR
#undef SEC
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    using M3 = std::integral_constant<int, 3>;
    using M4 = std::integral_constant<int, 4>;
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    // (re)start of the ring on the issue tile: K-tile 0 and X0 / W0 / W1 of K-tile 1 issued, K-tile 0 landed and visible
    auto ring_start = [&]() {
        issue(2, 0, 0, true); issue(0, 0, 0, true); issue(3, 0, 0, true); issue(1, 0, 0, true);
        issue(2, 1, 1, true); issue(0, 1, 1, true); issue(3, 1, 1, true);   // X1 of K-tile 1 is issued by the first LA
        wait_lb();                    // K-tile 0 landed; X0 / W0 / W1 of K-tile 1 stay in flight
        bar();
        if (wr) bar();   // stagger: waves 4-7 run one barrier behind waves 0-3
    };

    if (VP_STAGGER(g) > 0) {   // experiment (VP_G8_STAGGER): XCD x -- or, from 100 on, workgroup j of every XCD -- starts n * 1024 cycles late so
                           // that the epilogues no longer coincide.  Measured at every step: the launch gets slower by exactly the delay
                           // (the epilogue's cost is per CU, not a shared-bandwidth burst: profiles/gemm8_sections_r2.txt)
        const bool short_list = (tw.cnt - tw.j0 + tw.nloc - 1) / tw.nloc < (tw.cnt + tw.nloc - 1) / tw.nloc;   // one tile less than the longest list
        const int n = (VP_STAGGER(g) >= 300) ? (short_list ? VP_STAGGER(g) - 300 : 0)   // >= 300: only the workgroups with a tile of slack
                    : (VP_STAGGER(g) >= 200) ? ((blockIdx.x >> 3) & 1) * (VP_STAGGER(g) - 200)   // >= 200: every second workgroup of an XCD, all by the same delay (two phases)
                    : (VP_STAGGER(g) >= 100) ? (blockIdx.x >> 3) * (VP_STAGGER(g) - 100) : (blockIdx.x & 7) * VP_STAGGER(g);   // >= 100: per workgroup inside its XCD
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
    }
    // ---- prologue of the first tile: K-tile 0 complete, K-tile 1 in flight ----
    int t = tw.j0, m0, n0;
    tw.origin(t, g.reverse, C::BM, C::BN, m0, n0);
    set_tile(m0, n0);
    ring_start();

    for (;;) {
        zero_acc();
        const bool has_next = t + tw.nloc < tw.cnt;
        int nm0 = m0, nn0 = n0;   // no next tile: the ring keeps fetching (valid, unused) K-tiles 0 / 1 of this tile
        if (has_next) tw.origin(t + tw.nloc, g.reverse, C::BM, C::BN, nm0, nn0);
        const bool tl = (VP_ABLATE(g) & 32) != 0;
        unsigned long long ts0 = 0, ts1 = 0;
        if (tl) ts0 = __builtin_readcyclecounter();
        ktile(B0{}, M1{}, 1, 2, false, 0, 0);
        ktile(B1{}, M0{}, 2, 3, false, 0, 0);
        for (int kt = 2; kt < nk - 2; kt += 2) {
#if (VP_G8_ABL & 16)
            stamp_now = tl && kt == 4 && (t - tw.j0) / tw.nloc == 1;
#endif
            ktile(B0{}, M0{}, kt + 1, kt + 2, false, 0, 0);
#if (VP_G8_ABL & 16)
            stamp_now = false;
#endif
            ktile(B1{}, M0{}, kt + 2, kt + 3, false, 0, 0);
        }
        // last two K-tiles: LA(nk-2) still restages this tile's last X1; from LB(nk-2) on everything belongs to the next tile
        if constexpr (QKVA) {   // the fused qkv + attention epilogue needs the whole LDS: nothing of the next tile is fetched here (it streams in under the attention phase)
            ktile(B0{}, M3{}, nk - 1, 0, false, 0, 0);
            ktile(B1{}, M4{}, 0, 0, false, 0, 0);
        } else {
            ktile(B0{}, M0{}, nk - 1, 0, true, nm0, nn0);
            ktile(B1{}, M2{}, 0, 1, false, 0, 0);
        }
        if (tl) ts1 = __builtin_readcyclecounter();

        // ---------------- epilogue of tile (m0, n0) ----------------
        // (the same for the epilogue's per-lane offsets: rebuilt per tile from opaque copies of the lane coordinates)
        int frow_e = frow, fg_e = fg;
        asm volatile("" : "+v"(frow_e), "+v"(fg_e));
        // Both wave groups run their epilogues TOGETHER: group 0 waits one barrier for group 1 here, group 1 falls one barrier behind again
        // at the end.  One barrier apart and with no barrier inside it, the epilogues of the two groups ran one AFTER the other -- each
        // wave alone on its SIMD, held ~350 cycles by every 1 KiB store instruction (16 per wave) with nobody to issue beside it.
        // (round 3: fc1 -4.6 %, qkv -2.5 %, bit-identical; profiles/gemm8_sched_r3.txt)
        if constexpr (!DRAIN) { if (!wr) bar(); }
        if constexpr (QKVA) {
            // ---- attn.qkv epilogue + attention core for ONE crop x ONE head of head dim 80 (ViTPose-H) ----
            // Tile columns (the head-major weight copy, qkvattn.hip::qkv_head_major80_kernel): [0, 80) = q, [80, 160) = k, [160, 240) = v, [240, 256) = zero
            // rows.  Lane (fg_e, frow_e) holds for fragment f and row group J the four columns  c0 .. c0 + 3,  c0 = wc 64 + (f >> 1) 32 + fg_e 8 + (f & 1) 4
            // (the SPLIT placement of the 16-bit epilogues), of token row  wr 48 + rowJ(J) + frow_e  of the crop.
            //   1. LayerNorm fold + bias (common.h::ln_fold: the qkv epilogue's arithmetic), rounded to 16 bit, written to LDS in the attention layouts:
            //      Q and K as [192][160 B] rows (stride 160 B = 40 dwords: the four 16-lane service groups of a ds_read_b128 fragment read are conflict-free
            //      without padding or swizzle), V as five [192 keys][16 d] sub-tiles (attention.hip's layout for an odd number of sub-tiles) = 90 KiB of
            //      the 112 KiB ring, which is drained first (the K-loop's run-ahead fetches of the next tile are dropped and re-issued by ring_start below,
            //      as in the LDS-staged residual epilogue).
            //   2. the attention core of attention.hip<80>, instruction for instruction per query tile (S^T = K Q^T over three k-steps with the d >= 80
            //      lanes zeroed on both operands, fp32 softmax, O^T = V^T P^T through ds_read_b64_tr_b16): 12 query tiles on 8 waves -- tiles 0-7 one per
            //      wave, tiles 8-11 on waves 0-3 (every SIMD = waves w, w + 4 gets three tiles).  y is BIT-IDENTICAL to gemm (EPI_BIAS) + attention_launch.
            //   3. LDS: q / k / v live in the LAST 90 KiB of the CU's 160 KiB ([70 K, 160 K): Q, K, V), so ring buffer 0 and the X0 slot of buffer 1
            //      ([0, 68 K)) are free from the barrier that ends the K-loop on: the next tile's first K-tile and X0 of its second stream in under the
            //      fold, the LDS hand-over and the whole attention phase; W0 / W1 of K-tile 1 (which overlap Q) follow behind the attention phase, and the
            //      first K-tile starts as after ring_start -- without a ring restart latency and without the two K-tiles of run-ahead fetches the draining
            //      residual epilogue throws away.
            constexpr int HD = 80, QSTR = HD * 2, VSUB = 192 * 32, DT = HD / 16, KS = 3;
            constexpr int Q_OFF = 160 * 1024 - 2 * 192 * QSTR - DT * VSUB, K_OFF = Q_OFF + 192 * QSTR, V_OFF = K_OFF + 192 * QSTR;
            static_assert(Q_OFF >= C::BUF + C::XH && Q_OFF % 16 == 0, "q / k / v behind ring buffer 0 and the X0 slot of buffer 1");
            const int nbq = n0 + wc * 64 + fg_e * 8;
            const int mrow = m0 + wr * (C::XR / 2) + frow_e;
            f32x4 b4[4], s4[4];
            float2 st[TJ];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                b4[f] = *(const f32x4*)(g.bias + nbq + (f >> 1) * 32 + (f & 1) * 4);
                s4[f] = *(const f32x4*)(g.ln_s + nbq + (f >> 1) * 32 + (f & 1) * 4);
            }
#pragma unroll
            for (int J = 0; J < TJ; ++J) st[J] = *(const float2*)(g.rowstat + 2 * (size_t)(mrow + rowJ(J)));
            wait_vm<0>();
            if (!wr) bar();     // undo the stagger: both groups meet here
            __syncthreads();    // every wave is done with the ring
            if (has_next) {     // the next tile's K-tile 0 (ring buffer 0) and X0 of its K-tile 1: on their way from here on
                set_tile(nm0, nn0);
                issue(2, 0, 0, true); issue(0, 0, 0, true); issue(3, 0, 0, true); issue(1, 0, 0, true);
                issue(0, 1, 1, true);
            }
            if (!(VP_ABLATE(g) & 2)) {   // (tools: 2 = no fold / hand-over, 16 = no attention phase, 8 = no output stores)
                const int lrow = wr * (C::XR / 2) + frow_e;          // token row of the crop, + rowJ(J)
                // fragments 2 fp, 2 fp + 1 of a lane are EIGHT consecutive columns c0 .. c0 + 7 (never across a q / k / v boundary: 80 is a multiple of 8;
                // inside one 16-d V sub-tile row): one 16-byte LDS store per row group instead of two 8-byte ones (the 16 token rows of a store
                // instruction are 160 / 32 bytes apart: 2-way instead of 4-way bank conflicts -- the hand-over was LDS-store-bound)
#pragma unroll
                for (int fp = 0; fp < 2; ++fp) {
                    const int c0 = wc * 64 + fp * 32 + fg_e * 8;
                    const int seg = (c0 >= HD) + (c0 >= 2 * HD) + (c0 >= 3 * HD);   // 0 q, 1 k, 2 v, 3 padding (never stored)
                    const int d0 = c0 - seg * HD;
                    const int rmul = seg < 2 ? QSTR : 32;
                    const int base = seg < 2 ? (seg ? K_OFF : Q_OFF) + d0 * 2 : V_OFF + (d0 >> 4) * VSUB + (d0 & 15) * 2;
#pragma unroll
                    for (int J = 0; J < TJ; ++J) {
                        u32x4 o;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int f = 2 * fp + h;
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = ln_fold(acc[f][J][e], st[J].x, s4[f][e], st[J].y, b4[f][e]);
                            o[2 * h] = pack2<T>(v[0], v[1]);
                            o[2 * h + 1] = pack2<T>(v[2], v[3]);
                        }
                        if (seg < 3) *(u32x4*)(smem + base + (lrow + rowJ(J)) * rmul) = o;
                    }
                }
            }
            wait_lgkm<0>();
            bar();              // q / k / v visible (a raw barrier: __syncthreads() would wait for the DMA pieces issued above)
            if (!(VP_ABLATE(g) & 16)) {
                const char* Qs = smem + Q_OFF;
                const char* Ks = smem + K_OFF;
                const char* Vs = smem + V_OFF;
                int fr = frow, fgq = fg;
                asm volatile("" : "+v"(fr), "+v"(fgq));
                // 16-byte slot of k-step kk: kk 4 + fg; the d >= 80 lanes of k-step 2 (fg 2, 3) read the slot of lane fg - 2 (a broadcast: no bank conflict) and are zeroed
                const bool live2 = fgq < 2;
                const int slot2 = (live2 ? 8 + fgq : 6 + fgq) * 16;
                const char* vfrag = Vs + (fgq * 4 + (fr >> 2)) * 32 + (fr & 3) * 8;
                uint16_t* ybase = (uint16_t*)g.out + (size_t)m0 * K + (n0 >> 8) * HD;
                const float scale_log2e = g.attn_scale_log2e;
                for (int pass = 0; pass < 2; ++pass) {
                    const int qt = pass * 8 + wave;
                    if (qt >= 12) break;
                    u32x4 qf[KS];
                    {
                        const char* qrow = Qs + (qt * 16 + fr) * QSTR;
                        qf[0] = *(const u32x4*)(qrow + fgq * 16);
                        qf[1] = *(const u32x4*)(qrow + (4 + fgq) * 16);
                        qf[2] = *(const u32x4*)(qrow + slot2);
                        if (!live2) qf[2] = u32x4{0u, 0u, 0u, 0u};
                    }
                    f32x4 sc[12];
#pragma unroll
                    for (int kt = 0; kt < 12; ++kt) sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kt = 0; kt < 12; ++kt) {
                        const char* krow = Ks + (kt * 16 + fr) * QSTR;
                        const u32x4 k0 = *(const u32x4*)(krow + fgq * 16);
                        const u32x4 k1 = *(const u32x4*)(krow + (4 + fgq) * 16);
                        const u32x4 k2 = *(const u32x4*)(krow + slot2);   // d >= 80 lanes: finite k values of lane fg - 2 against q = 0: products are exact zeros,
                                                                          // as with attention.hip's zero-padded K rows (16-bit q / k / v are saturated: never inf)
                        sc[kt] = mfma16<T>(k0, qf[0], sc[kt]);
                        sc[kt] = mfma16<T>(k1, qf[1], sc[kt]);
                        sc[kt] = mfma16<T>(k2, qf[2], sc[kt]);
                    }
                    float mx = -3.0e38f;
#pragma unroll
                    for (int kt = 0; kt < 12; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[kt][r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    float l = 0.f;
                    const float mb = mx * scale_log2e;
#pragma unroll
                    for (int kt = 0; kt < 12; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = softmax_p(sc[kt][r], scale_log2e, mb);
                            sc[kt][r] = pv;
                            l += pv;
                        }
                    l += __shfl_xor(l, 16, 64);
                    l += __shfl_xor(l, 32, 64);
                    const float inv_l = 1.0f / l;
                    u32x4 pf[6];
#pragma unroll
                    for (int kb = 0; kb < 6; ++kb) {
                        pf[kb][0] = pack2_nosat<T>(sc[2 * kb][0], sc[2 * kb][1]);
                        pf[kb][1] = pack2_nosat<T>(sc[2 * kb][2], sc[2 * kb][3]);
                        pf[kb][2] = pack2_nosat<T>(sc[2 * kb + 1][0], sc[2 * kb + 1][1]);
                        pf[kb][3] = pack2_nosat<T>(sc[2 * kb + 1][2], sc[2 * kb + 1][3]);
                    }
                    uint16_t* dst = ybase + (size_t)(qt * 16 + fr) * K;
#pragma unroll
                    for (int dp = 0; dp < DT; ++dp) {
                        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kb = 0; kb < 6; ++kb) {
                            const char* vp_ = vfrag + dp * VSUB + kb * 1024;
                            const u32x2 lo = lds_tr16(vp_);          // keys 32 kb + 4 g + 0..3
                            const u32x2 hi = lds_tr16(vp_ + 512);    // keys 32 kb + 16 + 4 g + 0..3
                            o = mfma16<T>(u32x4{lo[0], lo[1], hi[0], hi[1]}, pf[kb], o);
                        }
                        u32x2 w;
                        w[0] = pack2_nosat<T>(o[0] * inv_l, o[1] * inv_l);
                        w[1] = pack2_nosat<T>(o[2] * inv_l, o[3] * inv_l);
                        if (!(VP_ABLATE(g) & 8)) *(u32x2*)(dst + dp * 16 + fgq * 4) = w;
                    }
                }
            }
            wait_lgkm<0>();
            bar();              // every wave is done reading q / k / v before the rest of the ring refills the LDS (raw: the output stores stay in flight)
            if (has_next) {
                issue(2, 1, 1, true); issue(3, 1, 1, true);   // W0, W1 of K-tile 1: with X0 (issued above) ring_start's state; X1 of K-tile 1 is issued by the first LA
                // K-tile 0 has landed once only the operations issued behind its last piece are outstanding: X0 of K-tile 1 (2 / 1 pieces), this wave's
                // output stores (waves 0-3: two query tiles = 10, waves 4-7: 5) and the four pieces above.  The stores themselves are NOT waited for here:
                // the first counted wait of the K-loop (LB of K-tile 0) retires them, half a K-tile later.
                if (VP_ABLATE(g) & (8 | 16)) { if (wr) wait_vm<5>(); else wait_vm<6>(); }
                else { if (wr) wait_vm<10>(); else wait_vm<16>(); }
                bar();
                if (wr) bar();   // stagger: waves 4-7 run one barrier behind waves 0-3
            }
        } else if constexpr (RESID && !RESID_LDS) {
            // ---- residual epilogue straight from registers (EPI_BIAS_RESID_LN on 256 x 256 tiles) ----
            // lane (fg_e, frow_e): rows m0 + (J >> 2) 128 + wr 64 + (J & 3) 16 + frow_e, columns nb .. nb + 15 (W rows are permuted on their
            // way into LDS, see above).  v = acc + bias + (hi + lo) of the residual stream, written back as two 16-bit planes;
            // LayerNorm partial statistics per (row, 64-column granule): the granule is the four lanes fg_e = 0..3 of a row, and the
            // summation tree is gemm.hip's -- per 8-column chunk ((v0+v1)+(v2+v3))+((v4+v5)+(v6+v7)), then chunk pairs, then
            // pairs of pairs (there: three DPP steps over 8 lanes; here: one add in the lane and two cross-lane adds) -- so the
            // statistics and everything downstream stay bit-identical to the LDS-staged epilogues.  No LDS, no barrier: the operand
            // ring runs on across the tile boundary exactly as for the 16-bit epilogues.
            const int nb = n0 + wc * 64 + fg_e * 16;
            const int mrow = m0 + wr * (C::XR / 2) + frow_e;
            uint16_t* out_hi = (uint16_t*)g.out;
            uint16_t* out_lo = out_hi + g.plane;
            const uint16_t* aux_hi = (const uint16_t*)g.aux;
            const uint16_t* aux_lo = aux_hi + g.plane;
            f32x4 bias4[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) bias4[f] = *(const f32x4*)(g.bias + nb + f * 4);
            const bool store = !(VP_ABLATE(g) & 8);
            const int gran = g.N >> 6;
            constexpr int RD = C::BM == 192 ? VP_G8_RESD192 : VP_G8_RESD;   // residual of row group J: hi cols 0-7, hi 8-15, lo 0-7, lo 8-15; fetched RD row groups ahead
            u32x4 res[RD + 1][4];
            auto load_res = [&](int J, u32x4(&r)[4]) {
                const size_t o = (size_t)(mrow + rowJ(J)) * g.ldo + nb;
                r[0] = *(const u32x4*)(aux_hi + o);
                r[1] = *(const u32x4*)(aux_hi + o + 8);
                r[2] = *(const u32x4*)(aux_lo + o);
                r[3] = *(const u32x4*)(aux_lo + o + 8);
            };
#pragma unroll
            for (int J = 0; J < RD; ++J) load_res(J, res[J]);
#pragma unroll
            for (int J = 0; J < TJ; ++J) {
                if (J + RD < TJ) load_res(J + RD, res[(J + RD) % (RD + 1)]);
                const u32x4(&r)[4] = res[J % (RD + 1)];
                const int m = mrow + rowJ(J);
                const size_t o = (size_t)m * g.ldo + nb;
                float v[16];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const f32x4 st = acc[f][J] + bias4[f];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = f * 4 + e;                       // column nb + c
                        const uint32_t wh = r[c >> 3][(c & 7) >> 1], wl = r[2 + (c >> 3)][(c & 7) >> 1];
                        const int sh = (c & 1) * 16;
                        v[c] = st[e] + (from_bits<T>((uint16_t)(wh >> sh)) + from_bits<T>((uint16_t)(wl >> sh)));
                    }
                }
                u32x4 oh[2], ol[2];
#pragma unroll
                for (int c = 0; c < 16; c += 2) {   // clamps v to the 16-bit range (the statistics below see the stored value)
                    uint32_t h_, l_;
                    split_planes2<T>(v[c], v[c + 1], h_, l_);
                    oh[c >> 3][(c & 7) >> 1] = h_;
                    ol[c >> 3][(c & 7) >> 1] = l_;
                }
                if (store) {
                    *(u32x4*)(out_hi + o) = oh[0];
                    *(u32x4*)(out_hi + o + 8) = oh[1];
                    *(u32x4*)(out_lo + o) = ol[0];
                    *(u32x4*)(out_lo + o + 8) = ol[1];
                }
                float sa = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                float sb = ((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15]));
                float s1 = sa + sb;
                s1 += __shfl_xor(s1, 16, 64);
                s1 += __shfl_xor(s1, 32, 64);
                const float mg = s1 * (1.0f / 64.0f);
                float qa = 0.f, qb = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float da = v[e] - mg, db = v[8 + e] - mg;
                    qa = fmaf(da, da, qa);
                    qb = fmaf(db, db, qb);
                }
                float s2 = qa + qb;
                s2 += __shfl_xor(s2, 16, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (fg_e == 0 && store) *(float2*)(g.stats_out + ((size_t)m * gran + (nb >> 6)) * 2) = float2{s1, s2};
            }
        } else if constexpr (!RESID) {
            // lane (fg_e, frow_e): rows m0 + (J >> 2) 128 + wr 64 + (J & 3) 16 + frow_e; columns (SPLIT, TI = 4) n0 + wc 64 + fg_e 8 + [0, 8) from
            // fragments 0, 1 and + 32 + [0, 8) from fragments 2, 3; otherwise n0 + wc 16 TI + fg_e 4 TI + [0, 4 TI).
            // Every operand of the epilogue is loaded up front (one latency, not one per row group); the LayerNorm-consumer
            // variant is chosen by ONE wave-uniform branch around the whole block.
            const int nb = SPLIT ? n0 + wc * 64 + fg_e * 8 : n0 + wc * 16 * C::TI + fg_e * 4 * C::TI;
            auto fcol = [&](int f) { return SPLIT ? (f >> 1) * 32 + (f & 1) * 4 : f * 4; };   // first column of fragment f relative to nb
            const int mrow = m0 + wr * (C::XR / 2) + frow_e;
            auto epilogue = [&](auto LNc) {
                constexpr bool LN = decltype(LNc)::value;
                f32x4 bias4[C::TI], s4[LN ? C::TI : 1];
                float2 st[LN ? TJ : 1];
#pragma unroll
                for (int f = 0; f < C::TI; ++f) bias4[f] = *(const f32x4*)(g.bias + nb + fcol(f));
                if constexpr (LN) {
#pragma unroll
                    for (int f = 0; f < C::TI; ++f) s4[f] = *(const f32x4*)(g.ln_s + nb + fcol(f));
#pragma unroll
                    for (int J = 0; J < TJ; ++J) st[J] = *(const float2*)(g.rowstat + 2 * (size_t)(mrow + rowJ(J)));
                }
                uint16_t* obase = g.out_blocked
                    ? (uint16_t*)g.out + (((size_t)(mrow >> 6) * (g.ldo >> 6) + (nb >> 6)) << 12) + ((mrow & 63) << 6) + (nb & 63)
                    : (uint16_t*)g.out + (size_t)mrow * g.ldo + nb;
                // + 16 rows: blocked 16 * 64 elements (mrow & 63 = wr-independent multiple: rows stay inside one 64-row block
                // for (J & 3) 16 + frow_e < 64), + 128 rows: two block rows
                const size_t step16 = g.out_blocked ? (size_t)16 * 64 : (size_t)16 * g.ldo;
                const size_t step128 = g.out_blocked ? ((size_t)2 * (g.ldo >> 6) << 12) : (size_t)C::XR * g.ldo;   // + one X half (blocked output: 256-row tiles only)
                const bool store = !(VP_ABLATE(g) & 8);
#pragma unroll
                for (int J = 0; J < TJ; ++J) {
                    uint32_t o[2 * C::TI];
#pragma unroll
                    for (int f = 0; f < C::TI; ++f) {
                        f32x4 v = acc[f][J];
                        if constexpr (LN) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = ln_fold(v[r], st[J].x, s4[f][r], st[J].y, bias4[f][r]);
                        } else {
                            v += bias4[f];
                        }
                        if (EPI == EPI_BIAS_GELU) {   // saturated by gelu_sat: no clamp in the conversion (one VALU instruction per value less: the epilogue is VALU-issue-bound)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = gelu_sat<T>(v[r]);
                            o[2 * f] = pack2_nosat<T>(v[0], v[1]);
                            o[2 * f + 1] = pack2_nosat<T>(v[2], v[3]);
                        } else {
                            o[2 * f] = pack2<T>(v[0], v[1]);
                            o[2 * f + 1] = pack2<T>(v[2], v[3]);
                        }
                    }
                    uint16_t* dst = obase + (size_t)(J / MJ) * step128 + (size_t)(J % MJ) * step16;
                    if constexpr (C::BM != 256) {   // 192-row tiles: a wave's 48 rows of an X half straddle 64-row blocks -- blocked output addressed row by row
                        if (g.out_blocked) {
                            const int row = mrow + rowJ(J);
                            dst = (uint16_t*)g.out + (((size_t)(row >> 6) * (g.ldo >> 6) + (nb >> 6)) << 12) + ((row & 63) << 6) + (nb & 63);
                        }
                    }
                    if (store) {
                        if constexpr (C::TI == 4) {
                            if (VP_ABLATE(g) & 64) {   // experiment: streaming (non-temporal) stores
                                __builtin_nontemporal_store(u32x4{o[0], o[1], o[2], o[3]}, (u32x4*)dst);
                                __builtin_nontemporal_store(u32x4{o[4], o[5], o[6], o[7]}, (u32x4*)(dst + 32));
                            } else {
                                *(u32x4*)dst = u32x4{o[0], o[1], o[2], o[3]};
                                *(u32x4*)(dst + 32) = u32x4{o[4], o[5], o[6], o[7]};   // SPLIT: the second run of 8 columns
                            }
                        } else {   // 12 columns = 24 bytes, 8-byte aligned
                            *(u32x2*)dst = u32x2{o[0], o[1]};
                            *(u32x2*)(dst + 4) = u32x2{o[2], o[3]};
                            *(u32x2*)(dst + 8) = u32x2{o[4], o[5]};
                        }
                    }
                }
            };
            // (a static issue priority for either wave group inside this epilogue -- s_setprio 2 for waves 4-7 or for waves 0-3 -- measured no change:
            // fc1 2.745 / 2.772 / 2.741 ms per step, profiles/gemm8_timeline_r5.txt)
            if (g.rowstat != nullptr) epilogue(std::true_type{});
            else epilogue(std::false_type{});
            if (tl && lane == 0 && (wave & 3) == 0) {   // stamps of waves 0 and 4 (one per stagger group): [wg][group][tile < 16][8]
                const int ti = (t - tw.j0) / tw.nloc;
                if (ti < 16) {
                    unsigned long long* sp = (unsigned long long*)g.stats_out + (((size_t)blockIdx.x * 2 + wr) * 16 + ti) * 8;
                    sp[0] = ts0; sp[1] = ts1; sp[2] = __builtin_readcyclecounter();   // main loop begin / end, epilogue end
#if (VP_G8_ABL & 16)
                    if (ti == 1) {
                        unsigned long long* sq = (unsigned long long*)g.stats_out + (((size_t)blockIdx.x * 2 + wr) * 16 + 8) * 8;
#pragma unroll
                        for (int i = 0; i < 12; ++i) sq[i] = sec[i];
                    }
#endif
                }
            }
        } else {
            // Residual epilogue through LDS (the ring is drained first and restarted afterwards: fc2's 48 K-tiles make
            // the tile boundary cheap).  Arithmetic and statistics order = gemm.hip's fused-LayerNorm producer.
            static_assert(C::BM == 256, "the LDS-staged residual epilogue is written for 256-row tiles");
            constexpr int ROWBYTES = C::BN * 4 + 16;
            constexpr int JPP = (C::BN == 256) ? 2 : 4;  // m-fragments (per wave and X half) staged per pass
            constexpr int CR = 32 * JPP;                 // rows per pass
            constexpr int NPASS = 256 / CR;
            constexpr int CPR = C::BN / 8;               // 8-element chunks per row
            constexpr int NCH = CR * CPR / C::NT;        // chunks per thread per pass
            constexpr int GR = C::BN / 64;
            static_assert(NCH * C::NT == CR * CPR, "chunks must split evenly over threads");
            static_assert(CR * ROWBYTES + C::BM * GR * 8 <= 160 * 1024, "LDS");
            float* statbuf = (float*)(smem + CR * ROWBYTES);
            uint16_t* out_hi = (uint16_t*)g.out;
            uint16_t* out_lo = out_hi + g.plane;
            const uint16_t* aux_hi = (const uint16_t*)g.aux;
            const uint16_t* aux_lo = aux_hi + g.plane;
            const int nl = wc * 16 * C::TI + fg_e * 4 * C::TI;   // first tile column of this lane
            f32x4 bias4[C::TI];
#pragma unroll
            for (int f = 0; f < C::TI; ++f) bias4[f] = *(const f32x4*)(g.bias + n0 + nl + f * 4);
            unsigned long long rs[6] = {ts0, ts1, 0, 0, 0, 0};   // tools/gemm8_timeline.py --resid: drain, passes, statistics, restart
            // staged row lr of pass p  <->  tile row (p / (4/JPP)) 128 + (lr / (16 JPP)) 64 + ((p % (4/JPP)) JPP + (lr / 16) % JPP) 16 + lr % 16
            auto tile_row = [&](int p, int lr) {
                return (p / (4 / JPP)) * 128 + (lr / (16 * JPP)) * 64 + ((p % (4 / JPP)) * JPP + (lr / 16) % JPP) * 16 + (lr & 15);
            };
            wait_vm<0>();
            if (!wr) bar();     // undo the stagger: both groups meet here
            __syncthreads();    // every wave is done with the ring
            if (tl) rs[2] = __builtin_readcyclecounter();
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                size_t orow_q[NCH];
                u32x4 ra[NCH], rb[NCH];
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    const int c = tid + q * C::NT;
                    const int lr = c / CPR, ch = c - lr * CPR;
                    const int m = m0 + tile_row(p, lr);
                    orow_q[q] = (size_t)m * g.ldo;
                    ra[q] = *(const u32x4*)(aux_hi + orow_q[q] + n0 + ch * 8);
                    rb[q] = *(const u32x4*)(aux_lo + orow_q[q] + n0 + ch * 8);
                }
#pragma unroll
                for (int jj = 0; jj < JPP; ++jj) {
                    char* lrow = smem + (wr * 16 * JPP + jj * 16 + frow_e) * ROWBYTES + nl * 4;
                    const int J = (p / (4 / JPP)) * 4 + (p % (4 / JPP)) * JPP + jj;
#pragma unroll
                    for (int f = 0; f < C::TI; ++f) *(f32x4*)(lrow + f * 16) = acc[f][J] + bias4[f];
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    const int c = tid + q * C::NT;
                    const int lr = c / CPR, ch = c - lr * CPR;
                    float v[8];
                    const f32x4 s0 = *(const f32x4*)(smem + lr * ROWBYTES + ch * 32);
                    const f32x4 s1 = *(const f32x4*)(smem + lr * ROWBYTES + ch * 32 + 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float st = e < 4 ? s0[e] : s1[e - 4];
                        const int sh = (e & 1) * 16;
                        const float r = from_bits<T>((uint16_t)(ra[q][e >> 1] >> sh)) + from_bits<T>((uint16_t)(rb[q][e >> 1] >> sh));
                        v[e] = st + r;
                    }
                    u32x4 oh, ol;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) { uint32_t h_, l_; split_planes2<T>(v[e], v[e + 1], h_, l_); oh[e >> 1] = h_; ol[e >> 1] = l_; }   // clamps v to the 16-bit range (statistics below see the stored value)
                    if (!(VP_ABLATE(g) & 8)) {
                        size_t so = orow_q[q] + n0 + ch * 8;
                        if (VP_ABLATE(g) & 128) so &= (size_t)0xFFFF8;   // tools/resid_store_probe.py: every store lands in the first 2 MB of the planes (stays in L2)
                        *(u32x4*)(out_hi + so) = oh;
                        *(u32x4*)(out_lo + so) = ol;
                    }
                    float s1s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                    s1s = row8_sum8(s1s);
                    const float mg = s1s * (1.0f / 64.0f);
                    float s2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = v[e] - mg;
                        s2 = fmaf(d, d, s2);
                    }
                    s2 = row8_sum8(s2);
                    if ((ch & 7) == 0) *(float2*)(statbuf + (tile_row(p, lr) * GR + (ch >> 3)) * 2) = float2{s1s, s2};
                }
                __syncthreads();
            }
            if (tl) rs[3] = __builtin_readcyclecounter();
            for (int i = tid; i < C::BM * GR; i += C::NT) {
                const int trow = i / GR, gi = i - trow * GR;
                if (!(VP_ABLATE(g) & 8))
                    *(float2*)(g.stats_out + ((size_t)(m0 + trow) * (g.N / 64) + ((n0 >> 6) + gi)) * 2) = *(const float2*)(statbuf + i * 2);
            }
            __syncthreads();
            if (tl) rs[4] = __builtin_readcyclecounter();
            if (has_next) ring_start();   // restart the ring on the next tile (the issue pointers already point at it)
            if (tl && g.ln_part && lane == 0 && (wave & 3) == 0) {
                const int ti = (t - tw.j0) / tw.nloc;
                if (ti < 16) {
                    unsigned long long* sp = (unsigned long long*)g.ln_part + (((size_t)blockIdx.x * 2 + wr) * 16 + ti) * 8;
                    rs[5] = __builtin_readcyclecounter();
#pragma unroll
                    for (int i = 0; i < 6; ++i) sp[i] = rs[i];
                }
            }
        }
        if constexpr (!DRAIN) { if (wr) bar(); }   // stagger again: waves 4-7 one barrier behind waves 0-3
        if (!has_next) break;
        t += tw.nloc;
        m0 = nm0;
        n0 = nn0;
    }
    wait_vm<0>();   // the ring's run-ahead DMAs must have landed before the LDS is released
    if constexpr (!DRAIN) {
        if (!wr) bar();   // pair the extra barrier of the staggered group
    }
}

template <class T, int EPI, class C>
static hipError_t launch8(const GemmArgs& a, hipStream_t s) {
    auto kern = gemm8_kernel<T, EPI, C>;
    constexpr int LDS = (EPI == EPI_QKV_ATTN) ? 160 * 1024 : (EPI == EPI_BIAS_RESID_LN && C::BN != 256) ? ((128 * (C::BN * 4 + 16) + C::BM * (C::BN / 64) * 8) > C::RING ? (128 * (C::BN * 4 + 16) + C::BM * (C::BN / 64) * 8) : C::RING) : C::RING;
    static bool attr_done[64] = {};   // per device: the LDS opt-in is a per-device function attribute
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    const int tiles = (a.M / C::BM) * (a.N / C::BN);
    int grid = tiles < 256 ? tiles : 256;
#ifdef VP_TOOLS   // experiment (tools/two_lane_probe.py): persistent grids of fewer workgroups, so that two handles' launches can share the chip
    static const int max_wgs = [] { const char* e = getenv("VP_G8_WGS"); return e ? atoi(e) : 256; }();
    if (grid > max_wgs) grid = max_wgs;
#endif
    if (grid >= 256) grid &= ~7;   // (only the tools override can make it a smaller non-multiple; below 256 tiles: one workgroup per tile)
    if (grid < 8) return hipErrorInvalidValue;
    // (the name rocprofv3 prints for this instantiation: G8<BN, BM>)
    if (a.desc) snprintf(a.desc, a.desc_cap, "gemm8_kernel<%s, %d, G8<%d, %d>>", std::is_same<T, F16>::value ? "F16" : "BF16", EPI, C::BN, C::BM);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), LDS, s, a);
    return hipGetLastError();
}

bool gemm8_supported(int epi, const GemmArgs& a, int bn, int bm) {
    if (epi == EPI_QKV_ATTN)   // one crop x one head (head dim 80) per 192 x 256 tile: M = crops * 192, N = heads * 256 (head-major weights), y [M, K]
        return bn == 256 && bm == 192 && a.M % 192 == 0 && a.N % 256 == 0 && (a.N >> 8) * 80 == a.K && a.K % 128 == 0 && a.K >= 256 && a.ldo == a.K &&
               (size_t)a.M * a.K * 2 < (1ull << 32) && (size_t)a.w_rows * a.K * 2 < (1ull << 32) && (a.M / 192) * (a.N / 256) >= 8 && a.rowstat && a.ln_s &&
               !a.a_blocked && !a.out_blocked && !a.reverse;
    if (epi != EPI_BIAS && epi != EPI_BIAS_GELU && epi != EPI_BIAS_RESID_LN) return false;
    if (bn != 256 && bn != 192) return false;
    if (bm != 256 && !(bm == 192 && bn == 256)) return false;   // 192-row tiles: 256 columns wide
    if (a.M % bm || a.N % bn || a.K % 128 || a.K < 256) return false;
    if ((size_t)a.M * a.K * 2 >= (1ull << 32) || (size_t)a.w_rows * a.K * 2 >= (1ull << 32)) return false;   // 32-bit per-lane offsets
    if ((a.M / bm) * (a.N / bn) < 8) return false;
    if (epi == EPI_BIAS_RESID_LN) return a.ldo == a.N && a.plane && a.stats_out && !a.out_blocked;
    if ((size_t)a.M * a.N >= (1ull << 31)) return false;
    return bn == 256 && a.ldo == a.N;   // wide 16-bit-output GEMMs: 256 x 256 tiles only
}

hipError_t gemm8_launch(int dtype, int epi, const GemmArgs& a, int bn, hipStream_t s, int bm) {
    if (!gemm8_supported(epi, a, bn, bm)) return hipErrorInvalidValue;
#define VP_G8(TY)                                                                                                       \
    do {                                                                                                                \
        if (epi == EPI_QKV_ATTN) return launch8<TY, EPI_QKV_ATTN, G8<256, 192>>(a, s);                                  \
        if (epi == EPI_BIAS) return bm == 192 ? launch8<TY, EPI_BIAS, G8<256, 192>>(a, s) : launch8<TY, EPI_BIAS, G8<256>>(a, s);                \
        if (epi == EPI_BIAS_GELU) return bm == 192 ? launch8<TY, EPI_BIAS_GELU, G8<256, 192>>(a, s) : launch8<TY, EPI_BIAS_GELU, G8<256>>(a, s);  \
        if (bm == 192) return launch8<TY, EPI_BIAS_RESID_LN, G8<256, 192>>(a, s);                                       \
        if (bn == 256) return launch8<TY, EPI_BIAS_RESID_LN, G8<256>>(a, s);                                            \
        return launch8<TY, EPI_BIAS_RESID_LN, G8<192>>(a, s);                                                           \
    } while (0)
    if (dtype == DT_F16) VP_G8(F16);
    VP_G8(BF16);
#undef VP_G8
}

}  // namespace vp
