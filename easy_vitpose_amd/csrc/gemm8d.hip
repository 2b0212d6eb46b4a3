// Deferred-epilogue variant of the 8-phase GEMM (experiment, GemmArgs::variant 19; own translation unit so that it does not
// perturb the register allocation of gemm8.hip's kernels).  MEASURED SLOWER than the end-of-tile epilogue of gemm8.hip
// (qkv 183-191 vs 183-188 us, fc1 280-287 vs 264-270 us at M = 49152): kept as the record of that experiment, see DESIGN.md.
#include "gemm8_common.h"

namespace vp {

// ---------------------------------------------------------------------------------------------------------------------
// Wide GEMMs with a 16-bit output (qkv: EPI_BIAS, fc1: EPI_BIAS_GELU), 256 x 256 tiles, DEFERRED EPILOGUE.
//
// With one workgroup per CU nothing overlaps an end-of-tile epilogue: the timeline (tools/gemm8_timeline.py) showed 18 %
// (qkv) / 21 % (fc1) of a tile's cycles in it -- all 256 CUs reach it together, their 32 MB of stores drain at the HBM
// write rate and the matrix pipes idle meanwhile.  Here the epilogue of a tile is cut into its four accumulator quadrants
// and folded into the main loop around the tile boundary, in the LOAD intervals (while the SIMD partner wave streams
// MFMAs), so both the VALU work (LayerNorm fold, bias, GELU, packing) and the stores overlap matrix work:
//     quadrant q00 is final after P1 of the tile's LAST K-tile -> converted + stored in that K-tile's P2 interval
//     q01 final after P2 -> P3 interval;   q11 final after P3 -> converted in P4, stored in P1 of the NEXT tile's first K-tile
//     q10 final after P4 -> converted + stored in P2 of the next tile's first K-tile
// and each quadrant's accumulators are free again exactly when the next tile needs them (q00 in its P1, ... q10 in its
// P4: the first K-tile of a tile multiplies into a ZERO accumulator operand instead of clearing registers).  No extra
// accumulator set: the only carried state is q11's 16 packed registers across one barrier.
// The epilogue operands (bias, LayerNorm row sums s[n], per-row (mean, rstd)) must not be ordinary global loads inside the
// loop (hipcc would wait vmcnt(0) for them and drain the DMA ring): they are staged by LDS-DMA into a small double-buffered
// LDS area in the tile's first K-tile (issued BEFORE that phase's operand DMA, so the counted P4 wait covers them without
// changing its count) and read with ds_read.
// Column permutation of this kernel: lane (fg, row) holds columns wc 64 + hn 32 + fg 8 + [0, 8) of quadrant (hm, hn), so
// a quadrant's store instruction writes 64 contiguous bytes per row.
// Stores count on vmcnt and may complete out of order with loads; the counted wait stays safe (loads return in order: if the
// K-tile it protects had not landed, the three younger slots' six DMAs would be outstanding too) and merely has to see the
// stores of the last 1-3 intervals acknowledged, which spread-out stores are.
template <class T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm8_wide_kernel(GemmArgs g) {
    using C = G8<256>;
    constexpr int EPI_LDS = C::RING;          // two 4 KiB operand areas behind the ring: [bias 1 KiB | s 1 KiB | (mean, rstd) 2 KiB]
    constexpr int SCR_LDS = C::RING + 8192;   // 1 KiB of transpose scratch per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int K = g.K, nk = K >> 6;
    TileWalk tw;
    tw.init(g, C::BM, C::BN);
    if (tw.j0 >= tw.cnt) return;
    const bool ln_in = g.rowstat != nullptr;

    const int rip = lane >> 3, pslot = lane & 7;
    const int slog = pslot ^ (((wave & 1) << 2) | (rip >> 1));
    const uint32_t voff_x = g.a_blocked ? (uint32_t)(rip * 128 + slog * 16) : (uint32_t)(rip * K + slog * 8) * 2u;
    // W row permutation: LDS row r of half h (owner column block wc = r >> 5, fragment phi = (r >> 4) & 1, rho = r & 15) holds
    // tile column  wc 64 + h 32 + (rho >> 2) 8 + phi 4 + (rho & 3)
    const int rho = ((wave & 1) << 3) | rip;
    const uint32_t voff_w = (uint32_t)(((rho >> 2) * 8 + (rho & 3)) * K + slog * 8) * 2u;
    const int wu0 = (wave >> 2) * 64 + ((wave >> 1) & 1) * 4;   // half 0, piece w (piece w + 8: + 128 columns; half 1: + 32)
    const size_t x64 = g.a_blocked ? (size_t)(K >> 6) * 8192 : (size_t)64 * K * 2;
    const size_t xkt = g.a_blocked ? 8192 : 128;
    const char* xb = nullptr;
    const char* wb = nullptr;
    auto set_tile = [&](int m0, int n0) {
        xb = g.a_blocked ? (const char*)(g.A + ((size_t)(m0 >> 6) * (K >> 6) << 12)) + (size_t)wave * 8 * 128
                         : (const char*)(g.A + (size_t)(m0 + wave * 8) * K);
        wb = (const char*)(g.W + (size_t)n0 * K);
    };
    auto issue = [&](int which, int B, int kt, bool force = false) {
        char* dst = smem + B * C::BUF + wave * 1024;
        if ((VP_ABLATE(g) & 1) && !force) return;
        if (which < 2) {
            const char* src = xb + (size_t)(which * 2) * x64 + (size_t)kt * xkt + voff_x;
            glds16(src, dst + which * C::HALF);
            glds16(src + x64, dst + which * C::HALF + 8192);
        } else {
            const char* src = wb + ((size_t)(wu0 + (which - 2) * 32) * K + (size_t)kt * 64) * 2 + voff_w;
            glds16(src, dst + (which == 2 ? C::OFF_W0 : C::OFF_W1));
            glds16(src + (size_t)128 * K * 2, dst + (which == 2 ? C::OFF_W0 : C::OFF_W1) + 8192);
        }
    };
    // epilogue operands of tile (m0, n0) -> LDS area e (waves 0-3, one 1 KiB piece each)
    auto stage_epi = [&](int m0, int n0, int e) {
        if (wave >= (ln_in ? 4 : 1)) return;   // without the LayerNorm fold only the bias is staged (neutral s / (mean, rstd) prefilled)
        char* dst = smem + EPI_LDS + e * 4096 + wave * 1024;
        const char* src;
        if (wave == 0) src = (const char*)(g.bias + n0);
        else if (wave == 1) src = (const char*)(g.ln_s + n0);
        else src = (const char*)(g.rowstat + 2 * (size_t)m0) + (wave - 2) * 1024;
        glds16(src + lane * 16, dst);
    };

    const int frow = lane & 15, fg = lane >> 4;
    const int foff = frow * 128 + ((fg ^ ((frow >> 1) & 7)) << 4);
    const int xoff = wr * 64 * 128 + foff;
    const int woff = wc * 32 * 128 + foff;

    f32x4 acc[4][8];
    u32x4 xs[4][2], w0[2][2], w1[2][2];
    u32x4 pk11[4];                       // q11 of the previous tile, converted, waiting for its store slot
#pragma unroll
    for (int j = 0; j < 4; ++j) pk11[j] = u32x4{0, 0, 0, 0};
    const bool store = !(VP_ABLATE(g) & 8);
    // 32-bit element offsets into the output (M N < 2^31 is checked by gemm8_supported)
    const uint32_t step16 = g.out_blocked ? 16u * 64u : 16u * (uint32_t)g.ldo;
    const uint32_t step128 = g.out_blocked ? ((2u * (uint32_t)(g.ldo >> 6)) << 12) : 128u * (uint32_t)g.ldo;
    const uint32_t lane_off = g.out_blocked ? (uint32_t)((wr * (g.ldo >> 6) + wc) << 12) + (uint32_t)(frow << 6) + (uint32_t)(fg * 8)
                                            : (uint32_t)(wr * 64 + frow) * (uint32_t)g.ldo + (uint32_t)(wc * 64 + fg * 8);
    // the same for the transposed (immediate) stores: lane -> row lane >> 2, 16-byte chunk lane & 3 of the quadrant's 64 bytes
    const uint32_t lane_off_t = g.out_blocked ? (uint32_t)((wr * (g.ldo >> 6) + wc) << 12) + (uint32_t)((lane >> 2) << 6) + (uint32_t)((lane & 3) * 8)
                                              : (uint32_t)(wr * 64 + (lane >> 2)) * (uint32_t)g.ldo + (uint32_t)(wc * 64 + (lane & 3) * 8);
    // output element offset of row group 0 of quadrant (hm, hn) of tile (m0, n0) for this lane
    auto out_off = [&](int m0, int n0, int hm, int hn, bool transposed = true) -> uint32_t {
        if (VP_ABLATE(g) & 128) { m0 = (blockIdx.x & 127) * 256; n0 = 0; }   // experiment: every tile of a workgroup overwrites the same 128 KiB (L2-resident stores)
        const uint32_t tile_off = g.out_blocked ? (uint32_t)(((m0 >> 6) * (g.ldo >> 6) + (n0 >> 6)) << 12) : (uint32_t)m0 * (uint32_t)g.ldo + (uint32_t)n0;
        return tile_off + (transposed ? lane_off_t : lane_off) + (uint32_t)hm * step128 + (uint32_t)hn * 32u;
    };
    // convert quadrant (HM, HN): LayerNorm fold / bias / GELU / pack -> four 16-byte row pieces
    auto convert = [&](auto HMc, auto HNc, int e, u32x4 (&pk)[4], bool now = false, uint32_t ooff = 0) {
        constexpr int HM = decltype(HMc)::value, HN = decltype(HNc)::value;
        const char* ep = smem + EPI_LDS + e * 4096;
        const int cb = (wc * 64 + HN * 32 + fg * 8) * 4;
        f32x4 b[2], sv[2];
        b[0] = *(const f32x4*)(ep + cb); b[1] = *(const f32x4*)(ep + cb + 16);
        sv[0] = *(const f32x4*)(ep + 1024 + cb); sv[1] = *(const f32x4*)(ep + 1024 + cb + 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // unconditional read (a conditional one makes hipcc drain the DMA ring with vmcnt(0) at the block entry); without
            // the LayerNorm fold the area holds the neutral (mean 0, rstd 1), s = 0: acc + bias exactly
            // (read as the aligned 16-byte pair of rows and selected: 8-byte LDS reads get merged into ds_read2_b64, in front of
            // which hipcc waits vmcnt(0) for the LDS-DMA in flight)
            const f32x4 st2 = *(const f32x4*)(ep + 2048 + (HM * 128 + wr * 64 + j * 16 + (frow & ~1)) * 8);
            const float mu = (frow & 1) ? st2[2] : st2[0], rs = (frow & 1) ? st2[3] : st2[1];
            uint32_t o[4];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                f32x4 v = acc[2 * HN + p][4 * HM + j];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = ln_fold(v[r], mu, sv[p][r], rs, b[p][r]);
                if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                }
                o[2 * p] = pack2<T>(v[0], v[1]);
                o[2 * p + 1] = pack2<T>(v[2], v[3]);
            }
            if (now) {
                // wave-private 16 x 64 B transpose through LDS: the fragment layout has the 16 lanes of a quarter-wave on 16
                // DIFFERENT rows (64 separate 16-byte write transactions per store instruction, ~350 cycles of issue each);
                // after the transpose four consecutive lanes hold one row's 64 bytes (16 transactions)
                char* sc = smem + SCR_LDS + wave * 1024;
                *(u32x4*)(sc + frow * 64 + ((fg ^ (frow >> 1)) & 3) * 16) = u32x4{o[0], o[1], o[2], o[3]};
                const u32x4 tr = *(const u32x4*)(sc + (lane >> 2) * 64 + (((lane & 3) ^ (lane >> 3)) & 3) * 16);
                if (store) *(u32x4*)((uint16_t*)g.out + (ooff + (uint32_t)j * step16)) = tr;
            } else {
                pk[j] = u32x4{o[0], o[1], o[2], o[3]};
            }
            __builtin_amdgcn_sched_barrier(0);   // one row group at a time: keeps the conversion's temporaries out of the 256-VGPR budget
        }
    };
    auto put = [&](uint32_t ooff, const u32x4 (&pk)[4]) {
        if (!store) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) *(u32x4*)((uint16_t*)g.out + (ooff + (uint32_t)j * step16)) = pk[j];
    };

    // one K-tile.  MODE 0: steady state; 1: first K-tile of a tile (zero accumulator operand, epilogue operands staged, and --
    // if a previous tile exists -- its q11 stores in P1 and its q10 epilogue in P2); 2: last K-tile (q00 in P2, q01 in P3, q11
    // converted in P4).  (pm0, pn0, pe): previous tile and its operand area; (m0, n0, e): this tile.
    auto ktile = [&](auto Bc, auto Mc, int kn1, int kn2, bool sw, int nm0, int nn0, int m0, int n0, int e, bool have_prev, int pm0, int pn0) {
        constexpr int B = decltype(Bc)::value, MODE = decltype(Mc)::value;
        const char* sb = smem + B * C::BUF;
        const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---------------- P1
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) w0[p][kk] = *(const u32x4*)(sb + C::OFF_W0 + ((woff + p * 2048) ^ (kk << 6)));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xs[j][kk] = *(const u32x4*)(sb + C::OFF_X0 + ((xoff + j * 2048) ^ (kk << 6)));
        __builtin_amdgcn_sched_barrier(0);
        issue(1, B ^ 1, kn1);
        if (sw) set_tile(nm0, nn0);
        if constexpr (MODE == 1) {
            if (have_prev) put(out_off(pm0, pn0, 1, 1, false), pk11);
        }
        wait_lgkm<8>();
        bar();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[p][j] = mfma16<T>(w0[p][kk], xs[j][kk], (MODE == 1 && kk == 0) ? zero4 : acc[p][j]);
        __builtin_amdgcn_s_setprio(0);
        bar();
        // ---------------- P2
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) w1[p][kk] = *(const u32x4*)(sb + C::OFF_W1 + ((woff + p * 2048) ^ (kk << 6)));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 1) stage_epi(m0, n0, e);   // BEFORE this phase's operand DMA: covered by the P4 wait as it is
        issue(2, B, kn2);
        if constexpr (MODE == 1) {
            if (have_prev) convert(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, e ^ 1, pk11, true, out_off(pm0, pn0, 1, 0));
        }
        if constexpr (MODE == 2) convert(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, e, pk11, true, out_off(m0, n0, 0, 0));
        bar();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[2 + p][j] = mfma16<T>(w1[p][kk], xs[j][kk], (MODE == 1 && kk == 0) ? zero4 : acc[2 + p][j]);
        __builtin_amdgcn_s_setprio(0);
        bar();
        // ---------------- P3
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xs[j][kk] = *(const u32x4*)(sb + C::OFF_X1 + ((xoff + j * 2048) ^ (kk << 6)));
        __builtin_amdgcn_sched_barrier(0);
        issue(0, B, kn2);
        if constexpr (MODE == 2) convert(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, e, pk11, true, out_off(m0, n0, 0, 1));
        bar();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[2 + p][4 + j] = mfma16<T>(w1[p][kk], xs[j][kk], (MODE == 1 && kk == 0) ? zero4 : acc[2 + p][4 + j]);
        __builtin_amdgcn_s_setprio(0);
        bar();
        // ---------------- P4
        issue(3, B, kn2);
        if constexpr (MODE == 2) convert(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, e, pk11);
        wait_vm<C::INFLIGHT>();
        bar();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[p][4 + j] = mfma16<T>(w0[p][kk], xs[j][kk], (MODE == 1 && kk == 0) ? zero4 : acc[p][4 + j]);
        __builtin_amdgcn_s_setprio(0);
        bar();
    };

    if (VP_STAGGER(g) > 0) {   // workgroup j of an XCD starts j * stagger * 64 cycles late: the XCD's store stream is spread over the tile time
        const int n = (blockIdx.x >> 3) * VP_STAGGER(g);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
    }
    if (!ln_in) {   // neutral LayerNorm operands in both areas: s = 0, (mean, rstd) = (0, 1)
        for (int i = tid; i < 2 * 768; i += 512) {
            const int e2 = i / 768, o = i - e2 * 768;   // 256 s values + 512 (mean, rstd) floats per area
            *(float*)(smem + EPI_LDS + e2 * 4096 + 1024 + o * 4) = (o >= 256 && (o & 1)) ? 1.f : 0.f;
        }
    }
    int t = tw.j0, m0, n0;
    tw.origin(t, g.reverse, C::BM, C::BN, m0, n0);
    set_tile(m0, n0);
    issue(2, 0, 0, true); issue(0, 0, 0, true); issue(3, 0, 0, true); issue(1, 0, 0, true);
    issue(2, 1, 1, true); issue(0, 1, 1, true); issue(3, 1, 1, true);
    wait_vm<C::INFLIGHT>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();
    if (wr) bar();

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    bool have_prev = false;
    int pm0 = 0, pn0 = 0, e = 0;
    for (;;) {
        const bool has_next = t + tw.nloc < tw.cnt;
        int nm0 = m0, nn0 = n0;
        if (has_next) tw.origin(t + tw.nloc, g.reverse, C::BM, C::BN, nm0, nn0);
        ktile(I0{}, I1{}, 1, 2, false, 0, 0, m0, n0, e, have_prev, pm0, pn0);
        ktile(I1{}, I0{}, 2, 3, false, 0, 0, m0, n0, e, false, 0, 0);
        for (int kt = 2; kt < nk - 2; kt += 2) {
            ktile(I0{}, I0{}, kt + 1, kt + 2, false, 0, 0, m0, n0, e, false, 0, 0);
            ktile(I1{}, I0{}, kt + 2, kt + 3, false, 0, 0, m0, n0, e, false, 0, 0);
        }
        ktile(I0{}, I0{}, nk - 1, 0, true, nm0, nn0, m0, n0, e, false, 0, 0);
        ktile(I1{}, I2{}, 0, 1, false, 0, 0, m0, n0, e, false, 0, 0);
        have_prev = true;
        pm0 = m0; pn0 = n0;
        e ^= 1;
        if (!has_next) break;
        t += tw.nloc;
        m0 = nm0;
        n0 = nn0;
    }
    // the last tile's q11 stores and q10 epilogue (its operand area is e ^ 1)
    put(out_off(pm0, pn0, 1, 1, false), pk11);
    convert(I1{}, I0{}, e ^ 1, pk11, true, out_off(pm0, pn0, 1, 0));
    wait_vm<0>();
    if (!wr) bar();
}

template <class T, int EPI>
static hipError_t launch8_wide(const GemmArgs& a, hipStream_t s) {
    auto kern = gemm8_wide_kernel<T, EPI>;
    constexpr int LDS = G8<256>::RING + 2 * 4096 + 8 * 1024;
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    const int tiles = (a.M / 256) * (a.N / 256);
    int grid = tiles < 256 ? tiles : 256;
    grid &= ~7;
    if (grid < 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, s, a);
    return hipGetLastError();
}

hipError_t gemm8_deferred_launch(int dtype, int epi, const GemmArgs& a, hipStream_t s) {
    if (epi != EPI_BIAS && epi != EPI_BIAS_GELU) return hipErrorInvalidValue;
    if (dtype == DT_F16) return epi == EPI_BIAS ? launch8_wide<F16, EPI_BIAS>(a, s) : launch8_wide<F16, EPI_BIAS_GELU>(a, s);
    return epi == EPI_BIAS ? launch8_wide<BF16, EPI_BIAS>(a, s) : launch8_wide<BF16, EPI_BIAS_GELU>(a, s);
}

}  // namespace vp
