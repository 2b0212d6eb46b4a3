// MFMA GEMM for the ViT encoder / head:  C[m][n] = sum_k A[m][k] * W[n][k]
//
// gfx950 design (not a warp-tiled CUDA kernel recompiled):
//  * block tile BM(m) x BN(n) x BK(k), waves arranged NWM x NWN, each wave a WM x WN
//    output tile of MFMA 16x16x32 accumulators (fp32).  Tile shapes are template
//    configurations (TileCfg); gemm_launch picks one per problem.
//  * operands are swapped into the MFMA: the W tile is the MFMA "A" operand (rows =
//    n), the activation tile is the "B" operand (cols = m).  A lane's 4 accumulator
//    registers are then 4 CONSECUTIVE n of one m: the epilogue reads bias/residual
//    and writes its result as one 8-byte (16-bit out) or 16-byte (fp32 out) access.
//  * both operand tiles go HBM/L2 -> LDS with global_load_lds (16 B per lane, no VGPR
//    round trip) through a STAGES-deep ring, one barrier per k-step, counted
//    s_waitcnt vmcnt(N) so that STAGES-2 tiles stay in flight across the barrier
//    (raw s_barrier: __syncthreads() would drain the LDS-DMA queue).
//  * the LDS image of a tile is lane-linear (global_load_lds writes base + lane*16),
//    so the bank-conflict XOR swizzle of the 16-B slots is applied to the per-lane
//    SOURCE address and again on the ds_read_b128 fragment reads:
//       BK=64 (128-B rows): slot ^= (row>>1)&7      BK=32 (64-B rows): slot ^= 3*((row>>3)&1)
//    both are conflict-free for the four 16-lane groups ds_read_b128 is serviced in.
//  * blockIdx -> tile mapping: XCD-aware (each of the 8 XCDs walks a contiguous range
//    of tiles) and grouped (GROUP_M m-tiles x all n-tiles at a time), so the blocks
//    resident on one XCD share a small set of A and W panels in its private 4 MiB L2.
//  * the deconv layers are implicit GEMMs: ConvTranspose2d(k=4,s=2,p=1) splits into 4
//    output-parity classes, each a GEMM with K = 4*Cin whose A rows are gathered
//    (one row chunk per k-step, zero row at the border) straight by the
//    global_load_lds source addresses -- no im2col buffer in HBM.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace vp {

template <int BM_, int BN_, int BK_, int WM_, int WN_, int STAGES_, int PIPE_ = 0, int DIRECT_ = 0>
struct TileCfg {
    static constexpr int DIRECT = DIRECT_;   // 1: fragment-shaped epilogue stores straight from registers (A/B reference)
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, STAGES = STAGES_;
    static constexpr int PIPE = PIPE_;   // 1: issue the fragment reads of both k-halves before the first MFMA block
    static constexpr int NWM = BM / WM, NWN = BN / WN, NWAVES = NWM * NWN, NT = NWAVES * 64;
    static constexpr int ROWB = BK * 2;            // bytes per tile row
    static constexpr int SLOTS = ROWB / 16;        // 16-B slots per row (8 or 4)
    static constexpr int RPG = 1024 / ROWB;        // rows per global_load_lds wave-instruction (8 or 16)
    static constexpr int W_BYTES = BN * ROWB, A_BYTES = BM * ROWB, STAGE_BYTES = W_BYTES + A_BYTES;
    static constexpr int LDS = STAGES * STAGE_BYTES;
    static constexpr int WP = BN / RPG / NWAVES, AP = BM / RPG / NWAVES;   // pieces per wave per stage
    static constexpr int G = WP + AP;
    static constexpr int KK = BK / 32, TI = WN / 16, TJ = WM / 16;
    static_assert(BK == 64 || BK == 32, "BK");
    static_assert(WP * RPG * NWAVES == BN && AP * RPG * NWAVES == BM, "tile rows must split evenly over waves");
    static_assert(LDS <= 160 * 1024, "LDS");
};

template <int BK> __device__ __forceinline__ int swz(int row, int slot) {
    return BK == 64 ? (slot ^ ((row >> 1) & 7)) : (slot ^ (((row >> 3) & 1) * 3));
}

// nn.GELU(approximate='none') (vit.py:127): common.h::gelu_erf (degree-4 fit of erfc on a <= 6.4, |error of erfc| <= 6e-7, monotone
// beyond, so the tail underflows to 0; Abramowitz-Stegun 7.1.26 needs v_rcp + v_exp + 12 VALU and was 48 us of the 318 us fc1 launch,
// ocml's erff ~3x more again).

// largest number of 16-row tiles per wave (dividing TJ) whose staged C rows fit in the tile ring's LDS
template <class C, int ES> constexpr int epi_rows_per_pass() {
    int jp = C::TJ;
    while (jp > 1 && C::NWM * jp * 16 * (C::BN * ES + 16) > C::LDS) jp /= 2;
    return jp;
}

// sum over the 16 lanes of a DPP row (every lane receives it): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
    return x;
}

// sum over the 8 lanes of a DPP half-row (every lane receives it): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
__device__ __forceinline__ float row8_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
    return x;
}

// fused-LayerNorm producer: as epi_rows_per_pass<C, 4>, leaving room for the tile's row statistics behind the staged rows
template <class C> constexpr int epi_rows_per_pass_ln() {
    int jp = C::TJ;
    while (jp > 1 && C::NWM * jp * 16 * (C::BN * 4 + 16) + C::BM * (C::BN / 64) * 8 > C::LDS) jp /= 2;
    return jp;
}

// ---- hand-scheduled fragment pipeline (PIPE 5): ds_read_b128 issued by inline asm so that hipcc does not
// track them (it otherwise waits lgkmcnt(0) for a whole burst); completion is waited for with COUNTED
// s_waitcnt lgkmcnt(N) placed by hand, each followed by sched_barrier(0) so no MFMA is hoisted above its wait.
template <int OFF> __device__ __forceinline__ void lds_read_b128(u32x4& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
template <int ROWB, int... Is>
__device__ __forceinline__ void lds_read_frags(u32x4* d, uint32_t addr, std::integer_sequence<int, Is...>) {
    (lds_read_b128<Is * 16 * ROWB>(d[Is], addr), ...);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// read number RI (0 .. R-1) of k-half 1: weight fragment RI, or activation fragment RI - TI
template <class C, int RI>
__device__ __forceinline__ void pipe5_issue(u32x4* f1, uint32_t wa1, uint32_t aa1) {
    if constexpr (RI < C::TI) lds_read_b128<RI * 16 * C::ROWB>(f1[RI], wa1);
    else if constexpr (RI < C::TI + C::TJ) lds_read_b128<(RI - C::TI) * 16 * C::ROWB>(f1[RI], aa1);
}

// k-half 0, group J: wait for activation fragment J, TI MFMAs, then issue reads 2J and 2J+1 of k-half 1
template <class T, class C, int J>
__device__ __forceinline__ void pipe5_half0(f32x4 (&acc)[C::TI][C::TJ], u32x4* f0, u32x4* f1, uint32_t wa1, uint32_t aa1) {
    if constexpr (J < C::TJ) {
        constexpr int R = C::TI + C::TJ;
        constexpr int issued = R + (2 * J < R ? 2 * J : R);
        wait_lgkm<issued - (C::TI + J + 1)>();
#pragma unroll
        for (int i = 0; i < C::TI; ++i) acc[i][J] = mfma16<T>(f0[i], f0[C::TI + J], acc[i][J]);
        __builtin_amdgcn_sched_barrier(0);
        pipe5_issue<C, 2 * J>(f1, wa1, aa1);
        pipe5_issue<C, 2 * J + 1>(f1, wa1, aa1);
        __builtin_amdgcn_sched_barrier(0);
        pipe5_half0<T, C, J + 1>(acc, f0, f1, wa1, aa1);
    }
}

// k-half 1, group J: every read is issued; fragment TI + J is complete once TJ - J - 1 younger reads remain
template <class T, class C, int J>
__device__ __forceinline__ void pipe5_half1(f32x4 (&acc)[C::TI][C::TJ], u32x4* f1) {
    if constexpr (J < C::TJ) {
        wait_lgkm<C::TJ - J - 1>();
#pragma unroll
        for (int i = 0; i < C::TI; ++i) acc[i][J] = mfma16<T>(f1[i], f1[C::TI + J], acc[i][J]);
        __builtin_amdgcn_sched_barrier(0);
        pipe5_half1<T, C, J + 1>(acc, f1);
    }
}

template <class T, int EPI_, int AMODE, class C>
__global__ __launch_bounds__(C::NT, 2) void gemm_kernel(GemmArgs g) {
    constexpr bool LN_PROD = (EPI_ == EPI_BIAS_RESID_LN || EPI_ == EPI_POS_LN);   // fused-LayerNorm producer variant
    constexpr int EPI = LN_PROD ? EPI_ - 4 : EPI_;                                // arithmetic of the base epilogue
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (g.N + C::BN - 1) / C::BN;
    const int tiles_m = (g.M + C::BM - 1) / C::BM;
    // deconv: the four output parities of a tile are four CONSECUTIVE logical blocks (they read the same input rows, shifted by a tap): with
    // the XCD remap they run side by side on one XCD and share those rows in its L2 (parity on blockIdx.y ran all tiles of parity 0
    // first: every parity fetched the activations again)
    int bid, parity = 0, split = 0;
    if (AMODE == A_DECONV && g.parity_fast) {
        const int lb = xcd_remap(blockIdx.x, tiles_m * tiles_n * 4);
        parity = lb & 3;
        bid = lb >> 2;
    } else if (EPI_ == EPI_PARTIAL) {
        // split-K: logical block = split * tiles + tile, so that the contiguous range an XCD owns is (mostly) ONE k range of neighbouring tiles -- they share
        // operand panels in its L2; the S workgroups of a tile share nothing but the output region
        const int lb = xcd_remap(blockIdx.x, tiles_m * tiles_n * g.splitk);
        split = lb / (tiles_m * tiles_n);
        bid = lb - split * (tiles_m * tiles_n);
    } else {
        bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
        if (AMODE == A_DECONV) parity = blockIdx.y;
    }
    if (g.reverse) bid = tiles_m * tiles_n - 1 - bid;   // walk the tiles last-to-first: start with what the producer wrote last
    int tm, tn;
    if (g.group_m > 1) {   // grouped order: GROUP_M m-tiles x all n-tiles, m fastest inside the group
        const int per_group = g.group_m * tiles_n;
        const int grp = bid / per_group, first_m = grp * g.group_m;
        const int gsz = min(tiles_m - first_m, g.group_m);
        const int r = bid - grp * per_group;
        tm = first_m + r % gsz;
        tn = r / gsz;
    } else {
        tm = bid / tiles_n;
        tn = bid - tm * tiles_n;
    }
    const int n0 = tn * C::BN, m0 = tm * C::BM;
    const int K = g.K;
    const int klen = (EPI_ == EPI_PARTIAL) ? K / g.splitk : K;   // this workgroup's k range: [kbase, kbase + klen)
    const int kbase = (EPI_ == EPI_PARTIAL) ? split * klen : 0;
    if constexpr (LN_PROD) {
        // experiment (tools build, VP_PROJ_STAGGER = 1000 mode + n): of the first round of workgroups (two per CU, all started together) the SECOND
        // one of every CU starts n x 1024 cycles late, so that one workgroup of a CU is in its HBM-bound epilogue while the other runs its K-loop.
        // mode 0: second = wave slots 2 / 3 of the SIMD (HW_REG_HW_ID), 1: blockIdx >= 256, 2: odd (blockIdx >> 3); whole CUs late (both workgroups):
        // 3: odd CU id, 4: odd XCD, 5: odd shader engine
        if (VP_STAGGER(g) > 0 && blockIdx.x < 512) {
            const int mode = VP_STAGGER(g) / 1000, n = VP_STAGGER(g) % 1000;
            const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
            bool late;
            if (mode == 0) late = (hw >> 1) & 1;
            else if (mode == 1) late = blockIdx.x >= 256;
            else if (mode == 3) late = (hw >> 8) & 1;
            else if (mode == 4) late = blockIdx.x & 1;
            else if (mode == 5) late = (hw >> 13) & 1;
            else late = (blockIdx.x >> 3) & 1;
            if (late) for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
        }
    }

    // ---- staging addresses: piece p of an operand = rows [(p*NWAVES + wave)*RPG, +RPG) ----
    const int rip = lane / C::SLOTS, pslot = lane % C::SLOTS;
    const uint16_t* wsrc[C::WP];
    const uint16_t* asrc[C::AP];
    int ai[C::AP], aj[C::AP];
    const uint16_t* W = g.W + (AMODE == A_DECONV ? (size_t)parity * g.w_parity_stride : 0);
#pragma unroll
    for (int p = 0; p < C::WP; ++p) {
        const int r = (p * C::NWAVES + wave) * C::RPG + rip;
        wsrc[p] = W + (size_t)(n0 + r) * K + swz<C::BK>(r, pslot) * 8;
    }
#pragma unroll
    for (int p = 0; p < C::AP; ++p) {
        const int r = (p * C::NWAVES + wave) * C::RPG + rip;
        int m = ((VP_ABLATE(g) & 2) ? 0 : m0) + r;   // ablate 2 (tools): every tile loads the A rows of m-tile 0
        if (m > g.M - 1) m = g.M - 1;
        const int sl = swz<C::BK>(r, pslot) * 8;
        if (AMODE == A_DENSE) {
            asrc[p] = g.a_blocked ? g.A + ((size_t)(m >> 6) * (K >> 6) << 12) + ((m & 63) << 6) + sl : g.A + (size_t)m * K + sl;
        } else {
            const int t = m / g.Win;
            ai[p] = t % g.Hin;
            aj[p] = m - t * g.Win;
            asrc[p] = g.A + (size_t)m * g.Cin + sl;
        }
    }
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * C::STAGE_BYTES + wave * 1024;
        const int k0 = kbase + kt * C::BK;
#pragma unroll
        for (int p = 0; p < C::WP; ++p) glds16(wsrc[p] + k0, base + p * (C::NWAVES * 1024));
        if (AMODE == A_DENSE) {
            const int ka = g.a_blocked ? ((k0 >> 6) << 12) + (k0 & 63) : k0;   // 64x64-blocked A: next k block = +4096 elements
#pragma unroll
            for (int p = 0; p < C::AP; ++p) glds16(asrc[p] + ka, base + C::W_BYTES + p * (C::NWAVES * 1024));
        } else {
            const int tap = k0 / g.Cin, c0 = k0 - tap * g.Cin;
            const int ti = tap >> 1, tj = tap & 1;
            // parity a (rows): a=0 -> taps ky=1 (di=0), ky=3 (di=-1); a=1 -> ky=0 (di=+1), ky=2 (di=0)
            const int pa = parity >> 1, pb = parity & 1;
            const int di = pa ? (ti ? 0 : 1) : (ti ? -1 : 0);
            const int dj = pb ? (tj ? 0 : 1) : (tj ? -1 : 0);
#pragma unroll
            for (int p = 0; p < C::AP; ++p) {
                const bool ok = (unsigned)(ai[p] + di) < (unsigned)g.Hin && (unsigned)(aj[p] + dj) < (unsigned)g.Win;
                const uint16_t* src = ok ? asrc[p] + (ptrdiff_t)(di * g.Win + dj) * g.Cin + c0 : g.zero + pslot * 8;
                glds16(src, base + C::W_BYTES + p * (C::NWAVES * 1024));
            }
        }
    };

    // ---- fragment read offsets (bytes inside a stage) ----
    const int wn = wave / C::NWM, wm = wave % C::NWM;
    const int frow = lane & 15, fg = lane >> 4;
    const int foff = frow * C::ROWB + (swz<C::BK>(frow, fg) << 4);   // kk = 1 (BK=64): foff ^ 64
    const int woff = wn * C::WN * C::ROWB + foff;
    const int aoff = C::W_BYTES + wm * C::WM * C::ROWB + foff;

    f32x4 acc[C::TI][C::TJ];
#pragma unroll
    for (int i = 0; i < C::TI; ++i)
#pragma unroll
        for (int j = 0; j < C::TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = klen / C::BK;
    if constexpr (C::PIPE == 2) {
        // Register double-buffered fragments over a STAGES-deep LDS ring (BK = 32, one 32-deep MFMA step
        // per barrier): while the MFMAs of tile kt run, the ds_reads of tile kt+1 (already landed and made
        // visible by this step's barrier) fill the other fragment set and tile kt+STAGES-1 streams into the
        // slot freed two barriers ago -- the matrix pipe never waits for LDS, only at the barrier itself.
        static_assert(C::KK == 1 && C::STAGES >= 3, "PIPE 2 needs BK = 32 and >= 3 stages");
        u32x4 wfA[C::TI], afA[C::TJ], wfB[C::TI], afB[C::TJ];
        auto ldfrag = [&](u32x4(&wf)[C::TI], u32x4(&af)[C::TJ], int b) {
            const char* sb = smem + b * C::STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < C::TI; ++i) wf[i] = *(const u32x4*)(sb + (woff + i * 16 * C::ROWB));
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) af[j] = *(const u32x4*)(sb + (aoff + j * 16 * C::ROWB));
        };
        auto settle = [&](u32x4(&wf)[C::TI], u32x4(&af)[C::TJ]) {   // make the compiler wait for a fragment set HERE
#pragma unroll
            for (int i = 0; i < C::TI; ++i) asm volatile("" ::"v"(wf[i]));
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) asm volatile("" ::"v"(af[j]));
        };
        auto mma = [&](u32x4(&wf)[C::TI], u32x4(&af)[C::TJ]) {
#pragma unroll
            for (int i = 0; i < C::TI; ++i)
#pragma unroll
                for (int j = 0; j < C::TJ; ++j) acc[i][j] = mfma16<T>(wf[i], af[j], acc[i][j]);
        };
#pragma unroll
        for (int s = 0; s < C::STAGES - 1; ++s)
            if (s < nk) stage(s, s);
        if (nk >= C::STAGES - 1) wait_vmcnt<C::G * (C::STAGES - 2)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ldfrag(wfA, afA, 0);
        int nbuf = 1 % C::STAGES, pbuf = C::STAGES - 1;   // buffer of tile kt+1, buffer tile kt+STAGES-1 goes to
        auto step = [&](int kt, u32x4(&cwf)[C::TI], u32x4(&caf)[C::TJ], u32x4(&nwf)[C::TI], u32x4(&naf)[C::TJ]) {
            if (kt + C::STAGES - 2 < nk) wait_vmcnt<C::G * (C::STAGES - 3)>();   // tile kt+1 landed (own share)
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();   // tile kt+1 visible; every wave is done with the slot of tile kt-1
            asm volatile("" ::: "memory");
            settle(cwf, caf);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + C::STAGES - 1 < nk) stage(kt + C::STAGES - 1, pbuf);
            if (kt + 1 < nk) ldfrag(nwf, naf, nbuf);
            __builtin_amdgcn_sched_barrier(0);
            mma(cwf, caf);
            nbuf = (nbuf + 1 == C::STAGES) ? 0 : nbuf + 1;
            pbuf = (pbuf + 1 == C::STAGES) ? 0 : pbuf + 1;
        };
        for (int kt = 0; kt < nk; kt += 2) {   // nk is even (K % 64 == 0, BK = 32)
            step(kt, wfA, afA, wfB, afB);
            step(kt + 1, wfB, afB, wfA, afA);
        }
    } else if constexpr (C::PIPE == 3) {
        // Staggered two-group schedule (8 waves, BK = 32, 4-stage LDS ring).  Waves 0-3 (group A) and 4-7
        // (group B) run the same LOAD -> barrier -> MFMA -> barrier sequence one barrier apart, so in every
        // barrier interval one wave of each SIMD streams its 32 MFMAs while its SIMD partner reads the next
        // fragments from LDS and issues the global_load_lds of tile k+3: the matrix pipe and the LDS/TA
        // pipes are busy simultaneously instead of alternately.
        //   A: LOAD(k) between barriers 2k..2k+1, MFMA(k) between 2k+1..2k+2;  B: one barrier later.
        //   tile j: issued in LOAD(j-3) into slot j%4 (last read in LOAD(j-4), >= 1 barrier earlier),
        //   landed before barrier 2j because every LOAD phase ends with vmcnt(2G) (two younger tiles in flight).
        static_assert(C::KK == 1 && C::STAGES == 4 && C::NWAVES == 8, "PIPE 3: BK = 32, 4 stages, 8 waves");
        const bool grp_b = __builtin_amdgcn_readfirstlane(tid) >= 256;
        u32x4 wf[C::TI], af[C::TJ];
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (s < nk) stage(s, s);
        if (nk >= 3) wait_vmcnt<2 * C::G>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (grp_b) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        int rbuf = 0, pbuf = 3;
        for (int k = 0; k < nk; ++k) {
            // ---- LOAD(k)
            const char* sb = smem + rbuf * C::STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < C::TI; ++i) wf[i] = *(const u32x4*)(sb + (woff + i * 16 * C::ROWB));
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) af[j] = *(const u32x4*)(sb + (aoff + j * 16 * C::ROWB));
            if (k + 3 < nk) {
                if (!(VP_ABLATE(g) & 1)) stage(k + 3, pbuf);
                wait_vmcnt<2 * C::G>();
            } else if (k + 2 < nk) {
                wait_vmcnt<C::G>();
            } else {
                wait_vmcnt<0>();
            }
#pragma unroll
            for (int i = 0; i < C::TI; ++i) asm volatile("" ::"v"(wf[i]));   // fragments complete before the barrier
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) asm volatile("" ::"v"(af[j]));
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // ---- MFMA(k)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < C::TI; ++i)
#pragma unroll
                for (int j = 0; j < C::TJ; ++j) acc[i][j] = mfma16<T>(wf[i], af[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            rbuf = (rbuf + 1) & 3;
            pbuf = (pbuf + 1) & 3;
        }
        if (!grp_b) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    } else {
    // PIPE 6 (round 5, small batches): TWO k-blocks per barrier.  A 64 x 64 tile spends ~500 of the ~760 cycles of a k-step on the wait, the barrier and the
    // LDS round trip in front of its 8 MFMAs per wave, and at a few crops no second workgroup shares the SIMD to hide them; with two k-blocks per barrier that
    // overhead is paid once per 128 k.  Same k order, same MFMAs: bit-identical.  Ring: tile j in slot j % STAGES; the prologue issues STAGES - KS tiles, an
    // iteration issues the KS tiles that go into the slots the previous iteration read.
    constexpr int KS = (C::PIPE == 6) ? 2 : 1;
    static_assert(C::PIPE != 6 || (C::STAGES >= 4 && C::KK == 2), "PIPE 6: >= 4 stages, BK = 64");
#pragma unroll
    for (int s = 0; s < C::STAGES - KS; ++s)
        if (s < nk) stage(s, s);
    if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
        // Fused LayerNorm at small batch (GemmArgs::ln_part): the tile's BM rows are merged ONCE, one row per thread, and shared through LDS
        // behind the operand ring -- not once per lane and fragment row in the epilogue (16 x redundant, round 2: slower than the separate
        // ln_finalize launch it replaced).  Round 6: a thread fetches its row's granule statistics straight from L2 into registers -- ln_tiles / 2
        // loads of 16 bytes, issued beside the first k-blocks -- and runs ln_merge on the register copy: the code of ln_finalize_kernel_t, the same
        // operations in the same order, hence the same bits.  (Rounds 3-5 DMA'd the tile's block of partials into a free ring slot and merged from
        // LDS: rows of ln_tiles * 8 = 96 / 128 / 160 bytes put every thread's reads on the same few banks -- up to 32-way conflicts at D = 1024 --
        // and the merge needed its own vmcnt(0) + barrier in front of the K-loop: with the 192-row one-round tiles the two ln_finalize launches
        // were cheaper than that; profiles/small_batch_r6.txt calls 7-8 and 11.)  (mean, rstd) land behind the ring; the first reader is the
        // epilogue, behind its own barrier.
        if (g.ln_part) {
            float2* st = (float2*)(smem + C::LDS);
            for (int r = tid; r < C::BM; r += C::NT) {
                int m = m0 + r;
                if (m > g.M - 1) m = g.M - 1;                                  // rows past M are never stored
                const float* src = g.ln_part + (size_t)m * g.ln_tiles * 2;
                float mean = 0.f, rstd = 1.f;
                switch (g.ln_tiles) {
                    case 6: ln_merge_row<6>(src, g.ln_inv_d, mean, rstd); break;
                    case 12: ln_merge_row<12>(src, g.ln_inv_d, mean, rstd); break;
                    case 16: ln_merge_row<16>(src, g.ln_inv_d, mean, rstd); break;
                    case 20: ln_merge_row<20>(src, g.ln_inv_d, mean, rstd); break;
                    default: ln_merge(src, g.ln_tiles, g.ln_inv_d, mean, rstd);
                }
                st[r] = float2{mean, rstd};
            }
        }
    }
    int buf = 0, pbuf = C::STAGES - KS;
    for (int kt = 0; kt < nk; kt += KS) {   // PIPE 6: nk is even (launch() checks K % 128 == 0)
        // tiles kt .. kt+KS-1 have landed once at most STAGES-2 KS younger tiles are still in flight (all of them issued: kt + STAGES - KS <= nk)
        if (C::STAGES > 2 * KS && kt + C::STAGES - KS <= nk) wait_vmcnt<C::G * (C::STAGES - 2 * KS)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // every wave's share of tiles kt .. kt+KS-1 landed; everyone finished tiles kt-KS .. kt-1
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int pb = pbuf + ks >= C::STAGES ? pbuf + ks - C::STAGES : pbuf + ks;
            if (kt + C::STAGES - KS + ks < nk && !(VP_ABLATE(g) & 1)) stage(kt + C::STAGES - KS + ks, pb);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
        const int cbuf = buf + ks >= C::STAGES ? buf + ks - C::STAGES : buf + ks;
        const char* sb = smem + cbuf * C::STAGE_BYTES;
        if constexpr (C::PIPE == 5) {
            // reads: per k-half wf[0..TI-1] then af[0..TJ-1] (R = TI + TJ); k-half 0 is issued up front, two
            // reads of k-half 1 after each MFMA group of k-half 0.  Group (kk, j) = TI MFMAs wf[kk][*] x af[kk][j];
            // it needs read TI + j of its half, so before it the allowed outstanding count is
            //   half 0: R + min(R, 2j) - (TI + j + 1)        half 1: TJ - j - 1
            static_assert(C::KK == 2 && C::TI + C::TJ <= 12 && 2 * C::TJ >= C::TI + C::TJ, "PIPE 5 schedule");
            constexpr int R = C::TI + C::TJ;
            u32x4 f0[R], f1[R];                       // [0, TI): weight fragments, [TI, R): activation fragments
            const uint32_t lb = (uint32_t)(size_t)(lds_ptr_t)(smem) + cbuf * C::STAGE_BYTES;
            const uint32_t wa0 = lb + woff, wa1 = lb + (woff ^ 64), aa0 = lb + aoff, aa1 = lb + (aoff ^ 64);
            lds_read_frags<C::ROWB>(f0, wa0, std::make_integer_sequence<int, C::TI>{});
            lds_read_frags<C::ROWB>(f0 + C::TI, aa0, std::make_integer_sequence<int, C::TJ>{});
            __builtin_amdgcn_s_setprio(1);
            __builtin_amdgcn_sched_barrier(0);
            pipe5_half0<T, C, 0>(acc, f0, f1, wa1, aa1);
            pipe5_half1<T, C, 0>(acc, f1);
            __builtin_amdgcn_s_setprio(0);
        } else if (C::PIPE == 1 && C::KK == 2) {
            // software pipelined fragment reads: the ds_reads of k-half 1 are issued between the two MFMA
            // blocks of k-half 0 and complete in their shadow (lgkmcnt is only 4 bits wide, so no more than
            // one half's reads are outstanding at a wait); sched_barriers pin this order for the compiler.
            u32x4 wf0[C::TI], af0[C::TJ], wf1[C::TI], af1[C::TJ];
#pragma unroll
            for (int i = 0; i < C::TI; ++i) wf0[i] = *(const u32x4*)(sb + (woff + i * 16 * C::ROWB));
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) af0[j] = *(const u32x4*)(sb + (aoff + j * 16 * C::ROWB));
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < C::TI / 2; ++i)
#pragma unroll
                for (int j = 0; j < C::TJ; ++j) acc[i][j] = mfma16<T>(wf0[i], af0[j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < C::TI; ++i) wf1[i] = *(const u32x4*)(sb + ((woff + i * 16 * C::ROWB) ^ 64));
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) af1[j] = *(const u32x4*)(sb + ((aoff + j * 16 * C::ROWB) ^ 64));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = C::TI / 2; i < C::TI; ++i)
#pragma unroll
                for (int j = 0; j < C::TJ; ++j) acc[i][j] = mfma16<T>(wf0[i], af0[j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < C::TI; ++i)
#pragma unroll
                for (int j = 0; j < C::TJ; ++j) acc[i][j] = mfma16<T>(wf1[i], af1[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) {
                u32x4 wf[C::TI], af[C::TJ];
#pragma unroll
                for (int i = 0; i < C::TI; ++i) wf[i] = *(const u32x4*)(sb + ((woff + i * 16 * C::ROWB) ^ (kk << 6)));
#pragma unroll
                for (int j = 0; j < C::TJ; ++j) af[j] = *(const u32x4*)(sb + ((aoff + j * 16 * C::ROWB) ^ (kk << 6)));
#pragma unroll
                for (int i = 0; i < C::TI; ++i)
#pragma unroll
                    for (int j = 0; j < C::TJ; ++j) acc[i][j] = mfma16<T>(wf[i], af[j], acc[i][j]);
            }
        }
        }
        buf = (buf + KS >= C::STAGES) ? buf + KS - C::STAGES : buf + KS;
        pbuf = (pbuf + KS >= C::STAGES) ? pbuf + KS - C::STAGES : pbuf + KS;
    }

    }

    // ---- epilogue ----
    // The accumulator fragment (4 consecutive n of one m per lane) would store 32-byte pieces at a row
    // stride: measured 1.2-1.7 TB/s.  Instead the C tile goes through the (now idle) LDS ring in passes
    // of JP row-tiles per wave and leaves as whole-row 16-byte-per-lane accesses (a wave instruction =
    // 1 KiB of consecutive output bytes); the fp32 residual / pos operand is read the same way.
    if constexpr (EPI == EPI_DECONV_FINAL) {
        // deconv2 + folded BN + ReLU with the final 1x1 conv fused behind it (topdown_heatmap_simple_head.py:188-193): the tile
        // holds ALL 256 channels of its 256 output pixels, so the 16-bit activations go to LDS instead of HBM (the [B,64,48,256]
        // tensor, 402 MB at batch 256, is never written or read) and a second small MFMA product with the hi + lo final-layer
        // weights (weights.hip upload_final: [16 hi rows][16 lo rows] groups) gives the heatmaps.  Arithmetic and
        // accumulation order are those of EPI_DECONV followed by EPI_HEATMAP (k ascending in steps of 32, hi and lo products
        // in separate accumulators, hi + lo, + bias): bit-identical heatmaps -- tests/test_gpu_gemm_cfgs.py, test_gpu_api.py.
        static_assert(C::BN == 256 && C::BM % (C::NWAVES * 16) == 0, "the fused head needs all 256 channels in one tile");
        constexpr int RB = C::BN * 2 + 16;                 // staged row: 256 channels x 16 bit + 16 B (conflict-free fragment reads)
        constexpr int RPW = C::BM / C::NWAVES, JW = RPW / 16;   // rows / 16-row fragments of the second product per wave
        f32x4 bias4[C::TI];
#pragma unroll
        for (int i = 0; i < C::TI; ++i) bias4[i] = *(const f32x4*)(g.bias + wn * C::WN + i * 16 + fg * 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // every wave is done with the operand ring
#pragma unroll
        for (int j = 0; j < C::TJ; ++j) {
            char* lrow = smem + (wm * C::WM + j * 16 + frow) * RB + (wn * C::WN + fg * 4) * 2;
#pragma unroll
            for (int i = 0; i < C::TI; ++i) {
                f32x4 v = acc[i][j] + bias4[i];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                u32x2 o;
                o[0] = pack2<T>(v[0], v[1]);
                o[1] = pack2<T>(v[2], v[3]);
                *(u32x2*)(lrow + i * 32) = o;
            }
        }
        // second product: wave w owns tile rows [w RPW, (w + 1) RPW); B fragments (pixels x 32 channels) from LDS, A fragments
        // (16 hi or lo weight rows x 32 channels) straight from L2 (the whole operand is 32 * ceil(Kp / 16) * 512 bytes).  The
        // weight fragments of a 16-joint group are fetched one group ahead (the first before the barrier that publishes the
        // staged tile: the accumulators are dead by then), so only one L2 round trip is exposed per tile.
        const int groups = (g.Kp + 15) >> 4;
        // the lane coordinates pass through an empty asm here and at every use below: the per-lane 64-bit addresses of the weight loads and
        // of the heatmap stores are then formed where they are used instead of being hoisted out of the group loop and kept live beside
        // 2 x 64 weight + 64 activation fragment registers (this kernel sat at 256 VGPRs with 8 spilled: VERDICT r3 weak 6)
        auto ldw = [&](u32x4(&w)[16], int u) {
            int frow_w = frow, fg_w = fg;
            asm volatile("" : "+v"(frow_w), "+v"(fg_w));
            const uint16_t* wrow = g.W2 + (size_t)(u * 32 + frow_w) * C::BN + fg_w * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                w[2 * ks] = *(const u32x4*)(wrow + ks * 32);
                w[2 * ks + 1] = *(const u32x4*)(wrow + 16 * C::BN + ks * 32);
            }
        };
        u32x4 wA[16], wB[16];
        ldw(wA, 0);
        __syncthreads();
        u32x4 bf[JW][8];
#pragma unroll
        for (int jj = 0; jj < JW; ++jj)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) bf[jj][ks] = *(const u32x4*)(smem + (wave * RPW + jj * 16 + frow) * RB + (ks * 32 + fg * 8) * 2);
        // output pixel of fragment row jj as (image, pixel inside the image): two 32-bit values per row instead of a 64-bit element offset
        int oimg[JW], opx[JW];
        const int opix = 4 * g.Hin * g.Win;
#pragma unroll
        for (int jj = 0; jj < JW; ++jj) {
            const int m = m0 + wave * RPW + jj * 16 + frow;
            const int t = m / g.Win, jx = m - t * g.Win, ii = t % g.Hin, img = t / g.Hin;
            oimg[jj] = (m < g.M && !(VP_ABLATE(g) & 8)) ? img : -1;
            opx[jj] = (2 * ii + (parity >> 1)) * (2 * g.Win) + 2 * jx + (parity & 1);
        }
        auto group = [&](const u32x4(&w)[16], int u) {
            f32x4 ah[JW], al[JW];
            int fg_g = fg;
            asm volatile("" : "+v"(fg_g));
            const int nb = u * 16 + fg_g * 4;
            const f32x4 b2 = *(const f32x4*)(g.bias2 + nb);   // padded to a multiple of 256 floats at upload
#pragma unroll
            for (int jj = 0; jj < JW; ++jj) { ah[jj] = f32x4{0.f, 0.f, 0.f, 0.f}; al[jj] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int jj = 0; jj < JW; ++jj) {
                    ah[jj] = mfma16<T>(w[2 * ks], bf[jj][ks], ah[jj]);
                    al[jj] = mfma16<T>(w[2 * ks + 1], bf[jj][ks], al[jj]);
                }
#pragma unroll
            for (int jj = 0; jj < JW; ++jj) {
                const f32x4 v = ah[jj] + al[jj];
                int im = oimg[jj], px = opx[jj];
                asm volatile("" : "+v"(im), "+v"(px));
                float* orow = g.out2 + ((size_t)im * g.Kp + nb) * opix + px;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (im >= 0 && nb + r < g.Kp) orow[(size_t)r * opix] = v[r] + b2[r];
            }
        };
        for (int u = 0; u < groups; u += 2) {
            if (u + 1 < groups) ldw(wB, u + 1);
            group(wA, u);
            if (u + 1 < groups) {
                if (u + 2 < groups) ldw(wA, u + 2);
                group(wB, u + 1);
            }
        }
    } else if constexpr (LN_PROD && !C::DIRECT) {
        // Fused-LayerNorm producer (patch embed, attn.proj, mlp.fc2).  The residual stream lives in HBM as two
        // 16-bit planes, x = hi + lo with hi = round16(x), lo = round16(x - hi): the same 4 bytes per element as
        // fp32 (>= 22 significant bits), but the hi plane IS the un-normalised 16-bit operand the next qkv / fc1
        // GEMM reads, so the fusion needs no extra copy.  Rows leave the LDS staging as 8-element chunks: two
        // 16-byte plane loads (residual) and two 16-byte plane stores per lane -- the same instruction count as
        // the fp32 epilogue.  Partial row statistics (sum, centred sum of squares) are taken per 64-column granule = 8
        // aligned lanes (DPP adds, fixed order: independent of the tile shape, hence of the batch size), parked in
        // LDS and written once per tile.
        constexpr int ROWBYTES = C::BN * 4 + 16;
        constexpr int JP = epi_rows_per_pass_ln<C>();
        constexpr int CR = C::NWM * JP * 16;          // rows staged per pass
        constexpr int CPR = C::BN / 8;                // 8-element chunks per row
        constexpr int NCH = CR * CPR / C::NT;         // chunks per thread per pass
        constexpr int GR = C::BN / 64;                // statistic granules per tile row
        static_assert(NCH * C::NT == CR * CPR && CPR % 8 == 0 && C::NT % 8 == 0, "chunks must split evenly over threads");
        static_assert(CR * ROWBYTES + C::BM * GR * 8 <= C::LDS, "row statistics must fit behind the staged rows");
        float* statbuf = (float*)(smem + CR * ROWBYTES);
        uint16_t* out_hi = (uint16_t*)g.out;
        uint16_t* out_lo = out_hi + g.plane;
        const uint16_t* aux_hi = (const uint16_t*)g.aux;
        const uint16_t* aux_lo = aux_hi + g.plane;
        f32x4 bias4[C::TI];
#pragma unroll
        for (int i = 0; i < C::TI; ++i)
            bias4[i] = (EPI == EPI_POS) ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(g.bias + n0 + wn * C::WN + i * 16 + fg * 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // every wave is done with the operand ring
#pragma unroll
        for (int p = 0; p < C::TJ / JP; ++p) {
            size_t orow_q[NCH];
            u32x4 ra[NCH], rb[NCH];   // residual: (hi, lo) planes of 8 elements, or the 8 fp32 pos values
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int c = tid + q * C::NT;
                const int lr = c / CPR, ch = c - lr * CPR;
                const int wmr = lr / (JP * 16), rr = lr - wmr * (JP * 16);
                const int m = m0 + wmr * C::WM + p * JP * 16 + rr;
                const int n = n0 + ch * 8;
                orow_q[q] = (size_t)-1;
                if (m >= g.M || n >= g.N) continue;
                orow_q[q] = (size_t)m * g.ldo;
                if (EPI == EPI_POS) {
                    const float* pr = g.aux + (size_t)(m % 192) * g.ldo + n;
                    ra[q] = *(const u32x4*)pr;
                    rb[q] = *(const u32x4*)(pr + 4);
                } else {
                    ra[q] = *(const u32x4*)(aux_hi + orow_q[q] + n);
                    rb[q] = *(const u32x4*)(aux_lo + orow_q[q] + n);
                }
            }
#pragma unroll
            for (int jj = 0; jj < JP; ++jj) {
                char* lrow = smem + ((wm * JP + jj) * 16 + frow) * ROWBYTES + (wn * C::WN + fg * 4) * 4;
#pragma unroll
                for (int i = 0; i < C::TI; ++i) *(f32x4*)(lrow + i * 64) = acc[i][p * JP + jj] + bias4[i];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int c = tid + q * C::NT;
                const int lr = c / CPR, ch = c - lr * CPR;
                const bool ok = orow_q[q] != (size_t)-1;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
                if (ok) {
                    const f32x4 s0 = *(const f32x4*)(smem + lr * ROWBYTES + ch * 32);
                    const f32x4 s1 = *(const f32x4*)(smem + lr * ROWBYTES + ch * 32 + 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float st = e < 4 ? s0[e] : s1[e - 4];
                        float r;
                        if (EPI == EPI_POS) {
                            r = __builtin_bit_cast(float, e < 4 ? ra[q][e] : rb[q][e - 4]);
                        } else {
                            const int sh = (e & 1) * 16;
                            r = from_bits<T>((uint16_t)(ra[q][e >> 1] >> sh)) + from_bits<T>((uint16_t)(rb[q][e >> 1] >> sh));
                        }
                        v[e] = st + r;
                    }
                    u32x4 oh, ol;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) { uint32_t h_, l_; split_planes2<T>(v[e], v[e + 1], h_, l_); oh[e >> 1] = h_; ol[e >> 1] = l_; }   // clamps v to the 16-bit range (statistics below see the stored value)
                    if (!(VP_ABLATE(g) & 8)) {
                        *(u32x4*)(out_hi + orow_q[q] + n0 + ch * 8) = oh;
                        *(u32x4*)(out_lo + orow_q[q] + n0 + ch * 8) = ol;
                    }
                }
                // granule statistics as (sum, M2 about the granule's own mean): merged exactly by ln_finalize (Chan et
                // al.), so the fused path has the two-pass LayerNorm's robustness to rows with a large common offset
                float s1 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                s1 = row8_sum(s1);
                const float mg = s1 * (1.0f / 64.0f);
                float s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[e] - mg;
                    s2 = fmaf(d, d, s2);
                }
                s2 = row8_sum(s2);
                if ((ch & 7) == 0) {
                    const int wmr = lr / (JP * 16), rr = lr - wmr * (JP * 16);
                    const int trow = wmr * C::WM + p * JP * 16 + rr;
                    *(float2*)(statbuf + (trow * GR + (ch >> 3)) * 2) = float2{s1, s2};
                }
            }
            if (p + 1 < C::TJ / JP) __syncthreads();
        }
        __syncthreads();
        for (int t = tid; t < C::BM * GR; t += C::NT) {
            const int trow = t / GR, gi = t - trow * GR;
            const int m = m0 + trow, n = n0 + gi * 64;
            if (m < g.M && n < g.N && !(VP_ABLATE(g) & 8))
                *(float2*)(g.stats_out + ((size_t)m * (g.N / 64) + (n >> 6)) * 2) = *(const float2*)(statbuf + t * 2);
        }
    } else if constexpr (EPI != EPI_HEATMAP && !C::DIRECT) {
        constexpr int ES = (EPI == EPI_BIAS_RESID || EPI == EPI_POS || EPI == EPI_PARTIAL) ? 4 : 2;   // bytes per staged element
        constexpr int ROWBYTES = C::BN * ES + 16;                               // +16 B: conflict-free fragment writes
        constexpr int JP = epi_rows_per_pass<C, ES>();
        constexpr int CR = C::NWM * JP * 16;                                    // rows staged per pass
        constexpr int CPR = C::BN * ES / 16;                                    // 16-B chunks per row
        constexpr int EPC = 16 / ES;                                            // elements per chunk
        constexpr int NCH = CR * CPR / C::NT;                                   // chunks per thread per pass
        static_assert(NCH * C::NT == CR * CPR, "chunks must split evenly over threads");
        f32x4 bias4[C::TI];
#pragma unroll
        for (int i = 0; i < C::TI; ++i)
            bias4[i] = (EPI == EPI_POS || EPI == EPI_PARTIAL) ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(g.bias + n0 + wn * C::WN + i * 16 + fg * 4);
        // fused LayerNorm, consumer side: per-row (mean, rstd) of this lane's TJ fragment rows
        constexpr bool LN_CONSUMER = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU);
        const bool ln_in = LN_CONSUMER && (g.rowstat != nullptr || g.ln_part != nullptr);
        // neutral defaults (mean 0, rstd 1: ln_fold(acc, 0, s, 1, b) == acc + b exactly), so that the arithmetic below is
        // unconditional straight-line code; only the LOADS sit behind the wave-uniform `ln_in` branch.  (A per-element
        // `if (ln_in)` around the fold compiled to a chain of scalar branches between the LDS staging writes, and that build
        // was not run-to-run deterministic once several workgroups shared a CU: tools/gemm_selfcheck.py.)
        float ln_mean[LN_CONSUMER ? C::TJ : 1], ln_rstd[LN_CONSUMER ? C::TJ : 1];
        f32x4 ln_s4[LN_CONSUMER ? C::TI : 1];
        if (LN_CONSUMER) {
#pragma unroll
            for (int i = 0; i < C::TI; ++i) ln_s4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) { ln_mean[j] = 0.f; ln_rstd[j] = 1.f; }
        }
        if (LN_CONSUMER && ln_in) {
#pragma unroll
            for (int i = 0; i < C::TI; ++i) ln_s4[i] = *(const f32x4*)(g.ln_s + n0 + wn * C::WN + i * 16 + fg * 4);
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) {
                int m = m0 + wm * C::WM + j * 16 + frow;
                if (m > g.M - 1) m = g.M - 1;
                if (g.ln_part) {   // merged once per tile row in the prologue (see there)
                    const float2 s2 = ((const float2*)(smem + C::LDS))[wm * C::WM + j * 16 + frow];
                    ln_mean[j] = s2.x;
                    ln_rstd[j] = s2.y;
                } else {
                    ln_mean[j] = g.rowstat[2 * (size_t)m];
                    ln_rstd[j] = g.rowstat[2 * (size_t)m + 1];
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // every wave is done with the operand ring
#pragma unroll
        for (int p = 0; p < C::TJ / JP; ++p) {
            // this thread's output chunks of the pass: row offsets, and the fp32 residual / pos operand is
            // fetched NOW (coalesced 16 B per lane) so its latency overlaps the LDS staging below
            size_t orow_q[NCH];
            f32x4 res[(ES == 4 && EPI != EPI_PARTIAL) ? NCH : 1];
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int c = tid + q * C::NT;
                const int lr = c / CPR, ch = c - lr * CPR;
                const int wmr = lr / (JP * 16), rr = lr - wmr * (JP * 16);
                const int m = m0 + wmr * C::WM + p * JP * 16 + rr;
                const int n = n0 + ch * EPC;
                orow_q[q] = (size_t)-1;
                if (m >= g.M || n >= g.N) continue;
                if (EPI == EPI_DECONV) {
                    const int t = m / g.Win, jx = m - t * g.Win, ii = t % g.Hin, img = t / g.Hin;
                    orow_q[q] = ((size_t)(img * 2 * g.Hin + 2 * ii + (parity >> 1)) * (2 * g.Win) + 2 * jx + (parity & 1)) * (size_t)g.ldo;
                } else {
                    orow_q[q] = (size_t)m * g.ldo;
                }
                if (EPI == EPI_PARTIAL) orow_q[q] += (size_t)split * g.M * g.ldo;   // partial slab of this k range
                if (ES == 4 && EPI != EPI_PARTIAL) {
                    const size_t arow = (EPI == EPI_POS) ? (size_t)(m % 192) * g.ldo : orow_q[q];
                    res[q] = *(const f32x4*)(g.aux + arow + n);
                }
            }
#pragma unroll
            for (int jj = 0; jj < JP; ++jj) {
                char* lrow = smem + ((wm * JP + jj) * 16 + frow) * ROWBYTES + (wn * C::WN + fg * 4) * ES;
#pragma unroll
                for (int i = 0; i < C::TI; ++i) {
                    f32x4 v = acc[i][p * JP + jj];
                    if (LN_CONSUMER) {   // LayerNorm folded into this GEMM: rstd * (x.W' - mean * sum_k W') + c
                        const float mu = ln_mean[p * JP + jj], rs = ln_rstd[p * JP + jj];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = ln_fold(v[r], mu, ln_s4[i][r], rs, bias4[i][r]);
                    } else {
                        v += bias4[i];
                    }
                    if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_sat<T>(v[r]);
                    }
                    if (EPI == EPI_DECONV) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    if (ES == 2) {
                        u32x2 o;
                        if (EPI == EPI_BIAS_GELU) {   // already saturated (common.h::gelu_sat)
                            o[0] = pack2_nosat<T>(v[0], v[1]);
                            o[1] = pack2_nosat<T>(v[2], v[3]);
                        } else {
                            o[0] = pack2<T>(v[0], v[1]);
                            o[1] = pack2<T>(v[2], v[3]);
                        }
                        *(u32x2*)(lrow + i * 16 * ES) = o;
                    } else {
                        *(f32x4*)(lrow + i * 16 * ES) = v;
                    }
                }
            }
            __syncthreads();
            if (!(VP_ABLATE(g) & 8)) {
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    const int c = tid + q * C::NT;
                    const int lr = c / CPR, ch = c - lr * CPR;
                    const int n = n0 + ch * EPC;
                    const char* src = smem + lr * ROWBYTES + ch * 16;
                    if (orow_q[q] == (size_t)-1) continue;
                    if (ES == 2) {
                        const u32x4 v = *(const u32x4*)src;
                        uint16_t* dst = (uint16_t*)g.out + orow_q[q] + n;
                        if (EPI != EPI_DECONV && g.out_blocked) {   // [M/64][N/64][64][64] blocks (the next GEMM's A tiles)
                            const int wmr = lr / (JP * 16), rr = lr - wmr * (JP * 16);
                            const int m = m0 + wmr * C::WM + p * JP * 16 + rr;
                            dst = (uint16_t*)g.out + (((size_t)(m >> 6) * (g.ldo >> 6) + (n >> 6)) << 12) + ((m & 63) << 6) + (n & 63);
                        }
                        if (n + 8 <= g.N) *(u32x4*)dst = v;
                        else *(u32x2*)dst = u32x2{v[0], v[1]};   // N % 8 == 4 tail (N % 4 == 0 is required)
                    } else if (EPI == EPI_PARTIAL) {
                        *(f32x4*)((float*)g.out + orow_q[q] + n) = *(const f32x4*)src;
                    } else {
                        *(f32x4*)((float*)g.out + orow_q[q] + n) = *(const f32x4*)src + res[q];
                    }
                }
            }
            if (p + 1 < C::TJ / JP) __syncthreads();
        }
    } else {
        // direct fragment-shaped stores: EPI_HEATMAP (final 1x1 conv, few n, transposed out[(img*Kp + n)*3072 + p])
        // and the DIRECT A/B reference configurations
#pragma unroll
        for (int j = 0; j < C::TJ; ++j) {
            const int m = m0 + wm * C::WM + j * 16 + frow;
            if (m >= g.M || (VP_ABLATE(g) & 8)) continue;
            size_t orow;
            if (EPI == EPI_DECONV) {
                const int t = m / g.Win, jx = m - t * g.Win, ii = t % g.Hin, img = t / g.Hin;
                orow = ((size_t)(img * 2 * g.Hin + 2 * ii + (parity >> 1)) * (2 * g.Win) + 2 * jx + (parity & 1)) * (size_t)g.ldo;
            } else if (EPI == EPI_HEATMAP) {
                const int img = m / 3072, pix = m - img * 3072;
                orow = (size_t)img * g.Kp * 3072 + pix;
            } else {
                orow = (size_t)m * g.ldo;
            }
            if constexpr (EPI == EPI_HEATMAP) {
                // weight rows come as [16 hi][16 lo] groups (weights.hip upload_final): fragments 2u and 2u + 1
                // are the hi and lo products of output columns (n0 + wn WN) / 2 + 16 u + 4 fg .. + 3
#pragma unroll
                for (int u = 0; u < C::TI / 2; ++u) {
                    const int nb = (n0 + wn * C::WN) / 2 + u * 16 + fg * 4;
                    if (nb >= g.Kp) continue;
                    const f32x4 v = acc[2 * u][j] + acc[2 * u + 1][j];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb + r < g.Kp) ((float*)g.out)[orow + (size_t)(nb + r) * 3072] = v[r] + g.bias[nb + r];
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < C::TI; ++i) {
                const int nb = n0 + wn * C::WN + i * 16 + fg * 4;
                if (nb >= g.N) continue;
                f32x4 v = acc[i][j];
                if (EPI != EPI_POS) v += *(const f32x4*)(g.bias + nb);
                if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_DECONV) {
                    if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                    }
                    if (EPI == EPI_DECONV) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    u32x2 o;
                    o[0] = pack2<T>(v[0], v[1]);
                    o[1] = pack2<T>(v[2], v[3]);
                    *(u32x2*)((uint16_t*)g.out + orow + nb) = o;
                } else if (EPI == EPI_BIAS_RESID) {
                    *(f32x4*)((float*)g.out + orow + nb) = v + *(const f32x4*)(g.aux + orow + nb);
                } else {
                    *(f32x4*)((float*)g.out + orow + nb) = v + *(const f32x4*)(g.aux + (size_t)(m % 192) * g.ldo + nb);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Persistent variant for the wide GEMMs with a 16-bit output (qkv: EPI_BIAS, fc1: EPI_BIAS_GELU; dense A, BK = 64,
// two-stage ring, K a multiple of 128).  A workgroup walks a strided list of tiles and the operand ring simply
// runs on across the tile boundary:
//   * in the LAST k-step of a tile the staging addresses switch to the next tile and its k-block 0 streams into ring
//     buffer 0, so the prologue latency (~1.5 us of a ~10 us tile) is covered by that k-step and the epilogue;
//   * the epilogue stages the C tile in ring buffer 1 only, waits for the prefetched k-block BEFORE it issues its
//     global stores, and the next tile's first k-step needs no vmcnt wait: the store acknowledgements drain behind
//     a whole k-step of MFMAs instead of in front of a workgroup exit + relaunch.
// Ordering argument (everything below follows from it): a global_load_lds into buffer b is only issued after a
// barrier that every wave passes after its last read of b; a read of b only happens after the issuing waves'
// vmcnt(0) AND a later barrier.  Buffer 1 <- staged C rows after the barrier that ends the main loop; the next
// tile's k-block 1 goes into buffer 1 after that tile's first barrier, which every wave reaches after its last
// read of the staged rows.
// Tile schedule: XCD x (= blockIdx.x & 7) owns a contiguous range of the (grouped) tile order, its workgroups
// take it round-robin -- at any time the resident workgroups of an XCD work on consecutive tiles, as in the
// one-tile-per-workgroup launch.
template <class T, int EPI, class C>
__global__ __launch_bounds__(C::NT, 2) void gemm_persist_kernel(GemmArgs g) {
    static_assert(C::BK == 64 && C::STAGES == 2 && C::PIPE == 1 && (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU), "persistent variant");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (g.N + C::BN - 1) / C::BN;
    const int tiles_m = (g.M + C::BM - 1) / C::BM;
    const int ntiles = tiles_m * tiles_n;
    const int K = g.K, nk = K / C::BK;
    // this workgroup's tiles: base + j, base + j + nloc, ...
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int q = ntiles >> 3, r8 = ntiles & 7;
    const int base = (xcd < r8) ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    const int cnt = q + (xcd < r8 ? 1 : 0);
    if (j0 >= cnt) return;
    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int bid = base + t;
        int tm, tn;
        if (g.group_m > 1) {
            const int per_group = g.group_m * tiles_n;
            const int grp = bid / per_group, first_m = grp * g.group_m;
            const int gsz = min(tiles_m - first_m, g.group_m);
            const int r = bid - grp * per_group;
            tm = first_m + r % gsz;
            tn = r / gsz;
        } else {
            tm = bid / tiles_n;
            tn = bid - tm * tiles_n;
        }
        m0 = tm * C::BM;
        n0 = tn * C::BN;
    };
    // Staging addresses = wave-uniform base (SGPRs: tile, piece, k-step) + ONE per-lane byte offset that never
    // changes: row (p NWAVES + wave) 8 + rip of the tile, 16-byte slot pslot ^ ((row >> 1) & 7) -- and (row >> 1) & 7
    // = 4 (wave & 1) + (rip >> 1) for every piece p (NWAVES is even).  Keeps the tile loop out of the VGPR budget.
    static_assert(C::RPG == 8 && C::NWAVES % 2 == 0, "address split assumes 8 rows per piece and an even wave count");
    const int rip = lane >> 3, pslot = lane & 7;
    const int sw = pslot ^ (((wave & 1) << 2) | (rip >> 1));
    const uint32_t voff_w = (uint32_t)(rip * K + sw * 8) * 2u;
    const uint32_t voff_a = g.a_blocked ? (uint32_t)(rip * 64 + sw * 8) * 2u : voff_w;
    const char* wbase = nullptr;   // W + (n0 + wave 8) K        (wave-uniform)
    const char* abase = nullptr;   // A + (m0 + wave 8) K, or the 64x64 block of row m0 + wave 8
    auto set_addr = [&](int m0, int n0) {
        wbase = (const char*)(g.W + (size_t)(n0 + wave * 8) * K);
        abase = g.a_blocked ? (const char*)(g.A + ((size_t)(m0 >> 6) * (K >> 6) << 12))
                            : (const char*)(g.A + (size_t)(m0 + wave * 8) * K);
    };
    auto stage = [&](int kt, int buf) {
        char* sbase = smem + buf * C::STAGE_BYTES + wave * 1024;
        const int k0 = kt * C::BK;
#pragma unroll
        for (int p = 0; p < C::WP; ++p)
            glds16(wbase + ((size_t)(p * C::NWAVES * 8) * K + k0) * 2 + voff_w, sbase + p * (C::NWAVES * 1024));
#pragma unroll
        for (int p = 0; p < C::AP; ++p) {
            const char* pa;
            if (g.a_blocked) {   // piece rows (p NWAVES + wave) 8 .. + 7 of the tile: block row (..) >> 3, row (..) & 7 inside it
                const int pr = p * C::NWAVES + wave;
                pa = abase + ((((size_t)(pr >> 3) * (K >> 6) + (k0 >> 6)) << 12) + ((pr & 7) << 9)) * 2;
            } else {
                pa = abase + ((size_t)(p * C::NWAVES * 8) * K + k0) * 2;
            }
            glds16(pa + voff_a, sbase + C::W_BYTES + p * (C::NWAVES * 1024));
        }
    };
    const int wn = wave / C::NWM, wm = wave % C::NWM;
    const int frow = lane & 15, fg = lane >> 4;
    const int foff = frow * C::ROWB + (swz<C::BK>(frow, fg) << 4);
    const int woff = wn * C::WN * C::ROWB + foff;
    const int aoff = C::W_BYTES + wm * C::WM * C::ROWB + foff;

    // epilogue geometry: C rows staged in ring buffer 1
    constexpr int ROWBYTES = C::BN * 2 + 16;
    constexpr int JP = [] { int jp = C::TJ; while (jp > 1 && (C::TJ % jp || C::NWM * jp * 16 * ROWBYTES > C::STAGE_BYTES)) --jp; return jp; }();
    constexpr int CR = C::NWM * JP * 16, CPR = C::BN / 8, NCH = CR * CPR / C::NT;
    static_assert(C::TJ % JP == 0 && NCH * C::NT == CR * CPR && CR * ROWBYTES <= C::STAGE_BYTES, "epilogue staging must fit one ring buffer");
    char* cst = smem + C::STAGE_BYTES;
    const bool ln_in = g.rowstat != nullptr;

    int t = j0, m0, n0;
    tile_origin(t, m0, n0);
    set_addr(m0, n0);
    stage(0, 0);
    bool landed = false;   // k-block 0 of the current tile already waited for (by the previous tile's epilogue)
    for (;;) {
        f32x4 acc[C::TI][C::TJ];
#pragma unroll
        for (int i = 0; i < C::TI; ++i)
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int nm0 = 0, nn0 = 0;
        bool has_next = false;
        unsigned long long ts0 = 0, ts1 = 0, ks[5] = {0, 0, 0, 0, 0};
        if (VP_ABLATE(g) & 32) ts0 = __builtin_readcyclecounter();   // tools/gemm_timeline.py: per-tile phase stamps of wave 0
        for (int kt = 0; kt < nk; ++kt) {
            const bool stamp = (VP_ABLATE(g) & 32) && kt == 5;        // ... and the phases inside k-step 5
            if (stamp) ks[0] = __builtin_readcyclecounter();
            if (kt > 0 || !landed) wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (stamp) ks[1] = __builtin_readcyclecounter();
            if (kt + 1 < nk) {
                stage(kt + 1, (kt + 1) & 1);
            } else {
                has_next = t + nloc < cnt;
                if (has_next) {
                    tile_origin(t + nloc, nm0, nn0);
                    set_addr(nm0, nn0);
                    stage(0, 0);
                }
            }
            if (stamp) ks[2] = __builtin_readcyclecounter();
            const char* sb = smem + (kt & 1) * C::STAGE_BYTES;
            u32x4 wf0[C::TI], af0[C::TJ], wf1[C::TI], af1[C::TJ];
#pragma unroll
            for (int i = 0; i < C::TI; ++i) wf0[i] = *(const u32x4*)(sb + (woff + i * 16 * C::ROWB));
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) af0[j] = *(const u32x4*)(sb + (aoff + j * 16 * C::ROWB));
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < C::TI / 2; ++i)
#pragma unroll
                for (int j = 0; j < C::TJ; ++j) acc[i][j] = mfma16<T>(wf0[i], af0[j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (stamp) ks[3] = __builtin_readcyclecounter();
#pragma unroll
            for (int i = 0; i < C::TI; ++i) wf1[i] = *(const u32x4*)(sb + ((woff + i * 16 * C::ROWB) ^ 64));
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) af1[j] = *(const u32x4*)(sb + ((aoff + j * 16 * C::ROWB) ^ 64));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = C::TI / 2; i < C::TI; ++i)
#pragma unroll
                for (int j = 0; j < C::TJ; ++j) acc[i][j] = mfma16<T>(wf0[i], af0[j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < C::TI; ++i)
#pragma unroll
                for (int j = 0; j < C::TJ; ++j) acc[i][j] = mfma16<T>(wf1[i], af1[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            if (stamp) ks[4] = __builtin_readcyclecounter();
        }

        if (VP_ABLATE(g) & 32) ts1 = __builtin_readcyclecounter();
        // ---- epilogue of tile (m0, n0): bias / LayerNorm consumer / GELU, staged through ring buffer 1 ----
        f32x4 bias4[C::TI], ln_s4[C::TI];
        float ln_mean[C::TJ], ln_rstd[C::TJ];
#pragma unroll
        for (int i = 0; i < C::TI; ++i) bias4[i] = (VP_ABLATE(g) & 64) ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(g.bias + n0 + wn * C::WN + i * 16 + fg * 4);
#pragma unroll
        for (int i = 0; i < C::TI; ++i) ln_s4[i] = f32x4{0.f, 0.f, 0.f, 0.f};   // neutral fold, see gemm_kernel
#pragma unroll
        for (int j = 0; j < C::TJ; ++j) { ln_mean[j] = 0.f; ln_rstd[j] = 1.f; }
        if (ln_in) {
#pragma unroll
            for (int i = 0; i < C::TI; ++i) ln_s4[i] = *(const f32x4*)(g.ln_s + n0 + wn * C::WN + i * 16 + fg * 4);
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) {
                int m = m0 + wm * C::WM + j * 16 + frow;
                if (m > g.M - 1) m = g.M - 1;
                ln_mean[j] = g.rowstat[2 * (size_t)m];
                ln_rstd[j] = g.rowstat[2 * (size_t)m + 1];
            }
        }
        __syncthreads();   // every wave is done reading ring buffer 1 (the last k-block)
#pragma unroll
        for (int p = 0; p < C::TJ / JP; ++p) {
#pragma unroll
            for (int jj = 0; jj < JP; ++jj) {
                char* lrow = cst + ((wm * JP + jj) * 16 + frow) * ROWBYTES + (wn * C::WN + fg * 4) * 2;
#pragma unroll
                for (int i = 0; i < C::TI; ++i) {
                    f32x4 v = acc[i][p * JP + jj];
                    {
                        const float mu = ln_mean[p * JP + jj], rs = ln_rstd[p * JP + jj];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = ln_fold(v[r], mu, ln_s4[i][r], rs, bias4[i][r]);
                    }
                    u32x2 o;
                    if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_sat<T>(v[r]);
                        o[0] = pack2_nosat<T>(v[0], v[1]);
                        o[1] = pack2_nosat<T>(v[2], v[3]);
                    } else {
                        o[0] = pack2<T>(v[0], v[1]);
                        o[1] = pack2<T>(v[2], v[3]);
                    }
                    *(u32x2*)(lrow + i * 32) = o;
                }
            }
            if (p == 0) wait_vmcnt<0>();   // the next tile's k-block 0 has landed (own share) -- BEFORE any store is in flight
            __syncthreads();
#pragma unroll
            for (int qq = 0; qq < NCH; ++qq) {
                const int c = tid + qq * C::NT;
                const int lr = c / CPR, ch = c - lr * CPR;
                const int wmr = lr / (JP * 16), rr = lr - wmr * (JP * 16);
                const int m = m0 + wmr * C::WM + p * JP * 16 + rr;
                const int n = n0 + ch * 8;
                if (m >= g.M || n >= g.N) continue;
                const u32x4 v = *(const u32x4*)(cst + lr * ROWBYTES + ch * 16);
                uint16_t* dst = g.out_blocked
                    ? (uint16_t*)g.out + (((size_t)(m >> 6) * (g.ldo >> 6) + (n >> 6)) << 12) + ((m & 63) << 6) + (n & 63)
                    : (uint16_t*)g.out + (size_t)m * g.ldo + n;
                *(u32x4*)dst = v;
            }
            if (p + 1 < C::TJ / JP) __syncthreads();
        }
        if ((VP_ABLATE(g) & 32) && tid == 0) {
            unsigned long long* st = (unsigned long long*)g.stats_out + ((size_t)blockIdx.x * 32 + (t - j0) / nloc) * 8;
            st[0] = ts0; st[1] = ts1; st[2] = __builtin_readcyclecounter();
#pragma unroll
            for (int e = 0; e < 5; ++e) st[3 + e] = ks[e];
        }
        if (!has_next) break;
        t += nloc;
        m0 = nm0;
        n0 = nn0;
        landed = true;
    }
}

template <class T, int EPI, class C>
static hipError_t launch_persist(const GemmArgs& a, hipStream_t s) {
    auto kern = gemm_persist_kernel<T, EPI, C>;
    static bool attr_done[64] = {};   // the > 64 KiB LDS opt-in is a per-device function attribute
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    const int tiles = ((a.M + C::BM - 1) / C::BM) * ((a.N + C::BN - 1) / C::BN);
    const int resident = 256 * (160 * 1024 / C::LDS);           // workgroups the chip holds at once
    int grid = tiles < resident ? tiles : resident;
    grid &= ~7;
    if (grid < 8) return hipErrorInvalidValue;
    if (a.desc)
        snprintf(a.desc, a.desc_cap, "gemm_persist_kernel<%s, %d, TileCfg<%d, %d, %d, %d, %d, %d, %d, %d>>", std::is_same<T, F16>::value ? "F16" : "BF16", EPI,
                 C::BM, C::BN, C::BK, C::WM, C::WN, C::STAGES, C::PIPE, C::DIRECT);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS, s, a);
    return hipGetLastError();
}

// Tile configurations (id = GemmArgs::variant).  Measured on MI355X at M = 49152 (tools/gemm_tune.py,
// profiles/gemm_tune_r1.txt).  Default = Cfg8: one 192-token crop per m-tile, so the tile count divides
// evenly over 256 CUs x 2 resident blocks for every encoder GEMM (no tail wave), and the epilogue of one
// block overlaps the main loop of its CU partner.  The others are kept as measured alternatives:
// 256x256 halves the L2->LDS operand traffic but runs 1 block / CU (epilogue exposed, tail wave at N = D);
// Cfg6 is the staggered two-group (anti-phase wave pairs) schedule; Cfg5 the fragment-store A/B reference;
// Cfg10 = Cfg8 with hand-scheduled inline-asm ds_reads and counted lgkmcnt (bare loop +7 %, end to end +0-2 %:
// the operand stream, not the intra-wave schedule, is what bounds these GEMMs).  Cfg11 = the Cfg8 block tile cut
// into 8 wave tiles of 48x64: same main-loop rate (LDS reads are at 18 % utilisation, so the smaller wave tile
// costs nothing), but twice the threads for the VALU-heavy residual epilogue (plane split + LayerNorm statistics):
// proj+fc2 4.54 -> 4.30 ms per step; no gain for qkv / fc1, slower for the deconvs (96x32 wave tiles: -25 %).
//                    BM   BN  BK   WM  WN  STAGES PIPE DIRECT      LDS   waves
using Cfg0 = TileCfg<128, 128, 64, 64, 64, 2, 0, 0>;    //  64 KiB   4   (2 blocks / CU)
using Cfg1 = TileCfg<128, 128, 64, 64, 64, 2, 1, 0>;    //  same + pipelined fragment reads
using Cfg2 = TileCfg<256, 256, 64, 128, 64, 2, 0, 0>;   // 128 KiB   8   (1 block / CU)
using Cfg3 = TileCfg<256, 256, 64, 128, 64, 2, 1, 0>;   //  same + pipelined fragment reads
using Cfg4 = TileCfg<256, 256, 32, 128, 64, 4, 2, 0>;   // 128 KiB   8   4-stage ring, register double-buffered fragments
using Cfg5 = TileCfg<128, 128, 64, 64, 64, 2, 0, 1>;    //  Cfg0 with fragment-shaped epilogue stores (A/B reference)
using Cfg6 = TileCfg<256, 256, 32, 128, 64, 4, 3, 0>;   // 128 KiB   8   staggered two-group schedule
using Cfg7 = TileCfg<192, 256, 64, 96, 64, 2, 1, 0>;    // 112 KiB   8   (1 block / CU)
using Cfg8 = TileCfg<192, 128, 64, 96, 64, 2, 1, 0>;    //  80 KiB   4   (2 blocks / CU)  <- default
using Cfg9 = TileCfg<64, 64, 64, 32, 32, 2, 0, 0>;      //  32 KiB   4   (5 blocks / CU)  small batches: enough tiles to fill 256 CUs
using Cfg10 = TileCfg<192, 128, 64, 96, 64, 2, 5, 0>;   //  Cfg8 with the hand-scheduled (inline-asm ds_read, counted lgkmcnt) fragment pipeline
using Cfg11 = TileCfg<192, 128, 64, 48, 64, 2, 1, 0>;  //  80 KiB   8   Cfg8 tile as 8 waves of 48x64 (4 waves / SIMD, 122 VGPRs)  <- default for the residual GEMMs
// small batches (a few crops per GPU): the 2-stage ring waits for every k-block's full L2 latency; deeper rings keep 2-3 blocks in flight
using Cfg12 = TileCfg<64, 64, 64, 32, 32, 4, 0, 0>;     //  64 KiB   4   (2 blocks / CU)  Cfg9 with a 4-stage ring
using Cfg13 = TileCfg<128, 128, 64, 64, 64, 3, 1, 0>;   //  96 KiB   4   (1 block / CU)   Cfg1 with a 3-stage ring
using Cfg14 = TileCfg<64, 64, 64, 32, 32, 3, 0, 0>;     //  48 KiB   4   (3 blocks / CU)  Cfg9 with a 3-stage ring
using Cfg15 = TileCfg<128, 64, 64, 64, 32, 3, 0, 0>;    //  72 KiB   4   (2 blocks / CU)  128(m) x 64(n), 3-stage ring
static constexpr int NUM_TILE_CFGS = 16;
// (16-18 = the 8-phase kernel of gemm8.hip.)  Round 5, small batches IN SITU: every layer's weights are first touched from HBM (ViTPose-L: 25 MB per layer,
// 600 MB per forward -- more than L2 + the memory-side cache hold), so a k-block costs an HBM round trip, not the L2 hit the isolated sweeps of rounds 2-3 saw:
// a workgroup retires STAGES - 1 k-blocks per round trip whatever its tile, and one full round of workgroups with a deep ring beats more, smaller tiles.
using Cfg19 = TileCfg<192, 128, 64, 96, 64, 3, 1, 0>;   // 120 KiB   4   (1 block / CU)   Cfg8 with a 3-stage ring
using Cfg20 = TileCfg<192, 128, 64, 48, 64, 3, 1, 0>;   // 120 KiB   8   (1 block / CU)   Cfg11 with a 3-stage ring
using Cfg21 = TileCfg<64, 64, 64, 32, 32, 5, 0, 0>;     //  80 KiB   4   (2 blocks / CU)  Cfg9 with a 5-stage ring
using Cfg22 = TileCfg<128, 64, 64, 64, 32, 6, 0, 0>;    // 144 KiB   4   (1 block / CU)   128(m) x 64(n), 6-stage ring
using Cfg23 = TileCfg<64, 64, 64, 32, 32, 8, 0, 0>;     // 128 KiB   4   (1 block / CU)   Cfg9 with an 8-stage ring
using Cfg24 = TileCfg<128, 128, 64, 64, 64, 4, 1, 0>;   // 128 KiB   4   (1 block / CU)   Cfg1 with a 4-stage ring
using Cfg25 = TileCfg<128, 128, 32, 64, 64, 4, 0, 0>;   //  64 KiB   4   (2 blocks / CU)  128 x 128 with k-blocks of 32: 4-stage ring in Cfg1's LDS
using Cfg26 = TileCfg<128, 128, 32, 64, 64, 5, 0, 0>;   //  80 KiB   4   (2 blocks / CU)  ... 5-stage
using Cfg27 = TileCfg<64, 64, 64, 32, 32, 4, 1, 0>;     //  64 KiB   4   (2 blocks / CU)  Cfg12 with pipelined fragment reads
using Cfg28 = TileCfg<64, 64, 64, 32, 32, 5, 6, 0>;     //  80 KiB   4   (2 blocks / CU)  64 x 64, two k-blocks per barrier, 5-stage ring (3 k-blocks in flight)
using Cfg29 = TileCfg<64, 64, 64, 32, 32, 4, 6, 0>;     //  64 KiB   4   (2 blocks / CU)  ... 4-stage ring (2 in flight)
using Cfg30 = TileCfg<64, 64, 64, 32, 32, 6, 6, 0>;     //  96 KiB   4   (1 block / CU)   ... 6-stage ring (4 in flight)
using Cfg31 = TileCfg<32, 64, 64, 16, 32, 6, 6, 0>;     //  72 KiB   4   (2 blocks / CU)  32(m) x 64(n): twice the workgroups of a 1-2 crop GEMM, half the MFMAs per wave and k-block
using Cfg32 = TileCfg<32, 64, 64, 16, 32, 8, 6, 0>;     //  96 KiB   4   (1 block / CU)   ... 8-stage ring (6 in flight)
using Cfg41 = TileCfg<96, 64, 64, 48, 32, 4, 0, 0>;     //  80 KiB   4   (2 blocks / CU)  96(m) x 64(n), 4-stage ring: the residual GEMMs between the 64 x 64 and 128 x 64 regimes (round 6: <= 448 tiles;
                                                        //                                 ViTPose-L 11-14 crops, -B 15-18, -H 9-11: profiles/small_batch_r6.txt call 18)
// (round 6: the staggered two-group schedule -- PIPE 3, waves 0-3 / 4-7 one barrier apart -- on 256 x 128 / 128 x 128 / 128 x 256 tiles with k-blocks of 32 was measured for the
// 8-crop wide GEMMs and lost everywhere, +8 ... +25 % per step, as did register double-buffered fragments -- PIPE 2 -- on 192 x 128 / 128 x 128 / 128 x 64 / 64 x 64 tiles with k-blocks of 32,
// +3 ... +20 %: profiles/small_batch_r6.txt calls 9-10; the configurations are not kept.  Also measured and not kept (calls 13, 15): a 4-stage ring on the one-round
// 192 x 128 tile (+1 %), the 96 x 64 tile with a 6-stage ring / two k-blocks per barrier (loses wherever the 4-stage one wins).)

template <class T, int EPI, int AMODE, class C>
static hipError_t launch(const GemmArgs& a, hipStream_t s) {
    if constexpr ((EPI == EPI_BIAS_RESID_LN || EPI == EPI_POS_LN) && C::DIRECT) return hipErrorInvalidValue;
    auto kern = gemm_kernel<T, EPI, AMODE, C>;
    // fused head: the staged 16-bit tile [BM][BN + 8] may be larger than the operand ring
    constexpr int LDS_BYTES = (EPI == EPI_DECONV_FINAL && C::BM * (C::BN * 2 + 16) > C::LDS) ? C::BM * (C::BN * 2 + 16) : C::LDS;
    constexpr int LN_STAT_BYTES = C::BM * 8;   // (mean, rstd) per tile row behind the ring: used when GemmArgs::ln_part is set
    static_assert(LDS_BYTES + LN_STAT_BYTES <= 160 * 1024, "LDS");
    if (a.ln_part && (C::PIPE == 2 || C::PIPE == 3 || C::DIRECT)) return hipErrorInvalidValue;   // the prologue merge lives in the generic loop
    if (C::PIPE == 6 && a.K % 128 != 0) return hipErrorInvalidValue;                                      // two k-blocks per barrier
    if (a.ln_part && (a.ln_tiles & 1)) return hipErrorInvalidValue;   // a row's partial statistics are fetched as 16-byte pieces: ln_tiles * 8 bytes must be a multiple of 16
    static bool attr_done[64] = {};   // the > 64 KiB LDS opt-in is a per-device function attribute
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + LN_STAT_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    GemmArgs g = a;
    const int tiles_n = (a.N + C::BN - 1) / C::BN;
    g.w_parity_stride = (size_t)a.w_rows * a.K;
    if ((size_t)tiles_n * C::BN > (size_t)a.w_rows) return hipErrorInvalidValue;   // weight rows are padded at upload
    const int tiles = ((a.M + C::BM - 1) / C::BM) * tiles_n;
    if constexpr (EPI == EPI_PARTIAL) {
        if (a.splitk < 1 || a.K % (a.splitk * C::BK * (C::PIPE == 6 ? 2 : 1)) != 0 || a.ldo != a.N || a.N % 4 != 0 || C::DIRECT || C::PIPE == 2 || C::PIPE == 3) return hipErrorInvalidValue;
    }
    dim3 grid(AMODE == A_DECONV && a.parity_fast ? tiles * 4 : (EPI == EPI_PARTIAL ? tiles * a.splitk : tiles), AMODE == A_DECONV && !a.parity_fast ? 4 : 1);
    if (a.desc)
        snprintf(a.desc, a.desc_cap, "gemm_kernel<%s, %d, %d, TileCfg<%d, %d, %d, %d, %d, %d, %d, %d>>", std::is_same<T, F16>::value ? "F16" : "BF16", EPI,
                 AMODE, C::BM, C::BN, C::BK, C::WM, C::WN, C::STAGES, C::PIPE, C::DIRECT);
    hipLaunchKernelGGL(kern, grid, dim3(C::NT), LDS_BYTES + (a.ln_part ? LN_STAT_BYTES : 0), s, g);
    return hipGetLastError();
}

// the product library carries the configurations the selection rule of tile_rules.hip can pick (1, 3, 8, 9, 11, 12, 15, 20, 30, 31, 41);
// the measured alternatives are instantiated in the VP_TOOLS build only
#ifdef VP_TOOLS
#define VP_TOOLS_CASE(v) case v: return launch<T, EPI, AMODE, Cfg##v>(a, s);
#else
#define VP_TOOLS_CASE(v)
#endif
template <class T, int EPI, int AMODE>
static hipError_t by_variant(const GemmArgs& a, hipStream_t s) {
    switch (a.variant) {
        VP_TOOLS_CASE(0)
        case 1: return launch<T, EPI, AMODE, Cfg1>(a, s);
        VP_TOOLS_CASE(2)
        case 3: return launch<T, EPI, AMODE, Cfg3>(a, s);
        VP_TOOLS_CASE(4)
        VP_TOOLS_CASE(5)
        VP_TOOLS_CASE(6)
        VP_TOOLS_CASE(7)
        case 8: return launch<T, EPI, AMODE, Cfg8>(a, s);
        case 9: return launch<T, EPI, AMODE, Cfg9>(a, s);
        VP_TOOLS_CASE(10)
        case 11: return launch<T, EPI, AMODE, Cfg11>(a, s);
        case 12: return launch<T, EPI, AMODE, Cfg12>(a, s);
        VP_TOOLS_CASE(13)
        VP_TOOLS_CASE(14)
        case 15: return launch<T, EPI, AMODE, Cfg15>(a, s);
        VP_TOOLS_CASE(19)
        case 20: return launch<T, EPI, AMODE, Cfg20>(a, s);
        VP_TOOLS_CASE(21)
        VP_TOOLS_CASE(22)
        VP_TOOLS_CASE(23)
        VP_TOOLS_CASE(24)
        VP_TOOLS_CASE(25)
        VP_TOOLS_CASE(26)
        VP_TOOLS_CASE(27)
        VP_TOOLS_CASE(28)
        VP_TOOLS_CASE(29)
        case 30: return launch<T, EPI, AMODE, Cfg30>(a, s);
        case 31: return launch<T, EPI, AMODE, Cfg31>(a, s);
        VP_TOOLS_CASE(32)
        case 41: return launch<T, EPI, AMODE, Cfg41>(a, s);
    }
    return hipErrorInvalidValue;
}

// split-K partial products (EPI_PARTIAL): the tiles the small-batch rule of tile_rules.hip may pick for them
template <class T>
static hipError_t by_variant_partial(const GemmArgs& a, hipStream_t s) {
    switch (a.variant) {
        case 1: return launch<T, EPI_PARTIAL, A_DENSE, Cfg1>(a, s);
        case 11: return launch<T, EPI_PARTIAL, A_DENSE, Cfg11>(a, s);
        case 12: return launch<T, EPI_PARTIAL, A_DENSE, Cfg12>(a, s);
        case 15: return launch<T, EPI_PARTIAL, A_DENSE, Cfg15>(a, s);
        case 20: return launch<T, EPI_PARTIAL, A_DENSE, Cfg20>(a, s);
        case 30: return launch<T, EPI_PARTIAL, A_DENSE, Cfg30>(a, s);
        case 31: return launch<T, EPI_PARTIAL, A_DENSE, Cfg31>(a, s);
    }
    return hipErrorInvalidValue;
}

template <class T>
static hipError_t dispatch(int epi, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS: return by_variant<T, EPI_BIAS, A_DENSE>(a, s);
        case EPI_BIAS_GELU: return by_variant<T, EPI_BIAS_GELU, A_DENSE>(a, s);
        case EPI_BIAS_RESID: return by_variant<T, EPI_BIAS_RESID, A_DENSE>(a, s);
        case EPI_POS: return by_variant<T, EPI_POS, A_DENSE>(a, s);
        case EPI_DECONV: return by_variant<T, EPI_DECONV, A_DECONV>(a, s);
        case EPI_HEATMAP: return by_variant<T, EPI_HEATMAP, A_DENSE>(a, s);
        case EPI_BIAS_RESID_LN: return by_variant<T, EPI_BIAS_RESID_LN, A_DENSE>(a, s);
        case EPI_POS_LN: return by_variant<T, EPI_POS_LN, A_DENSE>(a, s);
        case EPI_PARTIAL: return by_variant_partial<T>(a, s);
        case EPI_DECONV_FINAL:   // one tile configuration: 256 x 256 (all channels of a pixel in one tile)
            if (a.variant != 3 || a.N != Cfg3::BN || !a.W2 || !a.bias2 || !a.out2 || a.Kp <= 0) return hipErrorInvalidValue;
            return launch<T, EPI_DECONV_FINAL, A_DECONV, Cfg3>(a, s);
    }
    return hipErrorInvalidValue;
}

int gemm_tile_bn(int variant) {
    static const int bn[NUM_TILE_CFGS] = {Cfg0::BN, Cfg1::BN, Cfg2::BN, Cfg3::BN, Cfg4::BN, Cfg5::BN, Cfg6::BN, Cfg7::BN, Cfg8::BN, Cfg9::BN, Cfg10::BN, Cfg11::BN,
                                          Cfg12::BN, Cfg13::BN, Cfg14::BN, Cfg15::BN};
    if (variant == 16 || variant == 18) return 256;
    if (variant == 17) return 192;
    if (variant == 19 || variant == 20 || (variant >= 24 && variant <= 26)) return 128;
    if ((variant >= 21 && variant <= 23) || (variant >= 27 && variant <= 32) || variant == 41) return 64;
    return (variant >= 0 && variant < NUM_TILE_CFGS) ? bn[variant] : 0;
}

hipError_t gemm_launch(int dtype, int epi, const GemmArgs& a, hipStream_t s) {
    if (a.K % 64 != 0 || a.M <= 0 || a.N <= 0) return hipErrorInvalidValue;
    if (a.variant >= 16 && a.variant <= 18) {   // 16: 256 x 256, 17: 256 x 192, 18: 192 x 256
#ifdef VP_TOOLS
        static const int stagger_env = [] { const char* e = getenv("VP_G8_STAGGER"); return e ? atoi(e) : -1; }();
        if (stagger_env >= 0) {
            GemmArgs b = a;   // experiments: override the start stagger
            b.stagger = stagger_env;
            return gemm8_launch(dtype, epi, b, a.variant == 17 ? 192 : 256, s, a.variant == 18 ? 192 : 256);
        }
#endif
        return gemm8_launch(dtype, epi, a, a.variant == 17 ? 192 : 256, s, a.variant == 18 ? 192 : 256);
    }
#ifdef VP_TOOLS
    static const int proj_stagger_env = [] { const char* e = getenv("VP_PROJ_STAGGER"); return e ? atoi(e) : 0; }();
    if (proj_stagger_env > 0 && (epi == EPI_BIAS_RESID_LN) && a.K <= a.N) {
        GemmArgs b = a;
        b.stagger = proj_stagger_env;
        return dtype == DT_F16 ? dispatch<F16>(epi, b, s) : dispatch<BF16>(epi, b, s);
    }
#endif
    if (a.persist) {   // persistent variant: wide 16-bit-output GEMMs on the default tile (a 256x256 instantiation spilled and was slower)
        if ((epi != EPI_BIAS && epi != EPI_BIAS_GELU) || a.variant != 8 || a.K % 128 || a.N % 8 || a.ldo != a.N || a.reverse ||
            a.M % Cfg8::BM || (size_t)a.M * a.K * 2 >= (1ull << 32) ||
            (size_t)((a.N + Cfg8::BN - 1) / Cfg8::BN) * Cfg8::BN > (size_t)a.w_rows)
            return hipErrorInvalidValue;
        if (dtype == DT_F16)
            return epi == EPI_BIAS ? launch_persist<F16, EPI_BIAS, Cfg8>(a, s) : launch_persist<F16, EPI_BIAS_GELU, Cfg8>(a, s);
        return epi == EPI_BIAS ? launch_persist<BF16, EPI_BIAS, Cfg8>(a, s) : launch_persist<BF16, EPI_BIAS_GELU, Cfg8>(a, s);
    }
    if (a.a_blocked && (epi == EPI_DECONV || epi == EPI_DECONV_FINAL || (a.M & 63))) return hipErrorInvalidValue;
    if (a.out_blocked && ((epi != EPI_BIAS && epi != EPI_BIAS_GELU) || (a.M & 63) || (a.N & 63) || a.ldo != a.N)) return hipErrorInvalidValue;
    if ((epi == EPI_BIAS_RESID_LN || epi == EPI_POS_LN) && (a.N % 64 != 0 || !a.plane || !a.stats_out)) return hipErrorInvalidValue;
    return dtype == DT_F16 ? dispatch<F16>(epi, a, s) : dispatch<BF16>(epi, a, s);
}

}  // namespace vp
