// 8-phase GEMM on MXFP8 operands -- the encoder GEMMs of the opt-in fp8 mode (vp_config.dtype = VP_DTYPE_FP8, BASELINE configs[4]):
//   C[m][n] = w_scale[n] * sum_k dequant(A8)[m][k] * W8[n][k]
// A8 = activations as OCP e4m3 codes with one E8M0 scale per 32 consecutive k (csrc/mx8.h: codes in 64 x 128 blocks, scales packed four
// 16-row fragments to a dword), W8 = weights as e4m3 codes with one fp32 scale per output channel.  The matrix instruction is the
// block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (the ~5 PFLOP/s pipe): the activation scale is an operand of the instruction, so the
// K-loop multiplies codes and never de-quantises.
//
// Structure = gemm8.hip's (read its header first): 512-thread persistent workgroups, 256 x BN tiles, ring of two K-tiles of four slots,
// two load / two MFMA sections per K-tile, two wave groups one barrier apart, counted vmcnt, W rows permuted on their way into LDS.
// What changes with one-byte operands:
//   * a K-tile is 128 k (the same 128-byte LDS rows, the same slot geometry and swizzle): K = 768 is 6 K-tiles instead of 12, and a
//     fragment pair (the two 16-byte reads of a row's k-halves, slots g and g + 4) is ONE MFMA operand: the instruction wants
//     k = 16 g + [0, 16) in registers 0-3 and k = 64 + 16 g + [0, 16) in registers 4-7 (measured: tools/mx_probe_diag.py) -- exactly
//     what the 16-bit kernel's two k-half reads fetch.  Half the MFMAs per K-tile, each twice as long: the same matrix-pipe time per
//     K-tile, half the K-tiles.
//   * the activation scales: per X half and K-tile ONE dword per lane (rows frow + 16 j, j = 0..3 -> op_sel j; k block = lane group),
//     fetched by an ordinary global load that rides in the counted-vmcnt stream right behind the DMA pieces of the slot it belongs
//     to (issued from inline asm, so hipcc neither tracks it nor waits for it: the section's counted wait retires it together with
//     its slot).  Two registers per X half alternate with the ring buffer.
//   * epilogues: acc * w_scale[n] + bias instead of acc + bias; the LayerNorm is NOT folded (the A operand is the normalised row,
//     quantised by ln_quant_kernel), and mlp.fc1 writes its GELU output as MXFP8 itself -- a 32-column block is two lanes of a wave --
//     so `hid` costs 1 byte per element on the way out and on the way into mlp.fc2.
// Accuracy: e4m3 operands carry 3 mantissa bits; this mode does NOT meet the north_star's 1e-3 on confidences (DESIGN.md section 6,
// tests/test_gpu_fp8.py assert the measured bounds).  It is never the default.
#include <cstdio>
#include <cstdlib>

#include "gemm8_common.h"
#include "mx8.h"

namespace vp {

namespace {
__device__ __forceinline__ i32x8 cat8(u32x4 lo, u32x4 hi) {
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}
// D += A * B in place, as ONE instruction with the accumulator tied (inline asm): through the builtin hipcc emits the untied form (vdst != srcC,
// early-clobber) for most of the 128 accumulator registers, which costs a second accumulator tuple per instruction in flight and ended in
// 120-260 spilled VGPRs (round 4).  Hazards the compiler no longer sees, and why they cannot occur: the operand registers are written by
// ds_read (explicit lgkmcnt(0) + barrier before every MFMA section) and by the scale loads (counted vmcnt + barrier); an accumulator is
// read again only one K-tile section later (>= 2 barriers) or in the epilogue.  J = which byte of the activation scale dword (op_sel).
template <int J>
__device__ __forceinline__ void mfma_mx_acc(f32x4& acc, const i32x8& w, const i32x8& x, int sw, int sx) {
    if constexpr (J == 0) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(w), "v"(x), "v"(sw), "v"(sx));
    if constexpr (J == 1) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,1,0] op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(w), "v"(x), "v"(sw), "v"(sx));
    if constexpr (J == 2) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(w), "v"(x), "v"(sw), "v"(sx));
    if constexpr (J == 3) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(w), "v"(x), "v"(sw), "v"(sx));
}
// untracked global load of one dword: address = wave-uniform base + per-lane byte offset.  The compiler believes `d` is ready at once;
// the schedule's counted s_waitcnt retires the load before the register is read (see ktile below).
__device__ __forceinline__ void load_scale(int& d, uint32_t voff, const uint8_t* sbase) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
}  // namespace

template <int EPI, class C>
__global__ __launch_bounds__(512, 2) void gemm8f_kernel(GemmArgs g) {
    using T = F16;   // type of the 16-bit outputs / residual planes
    constexpr bool RESID = (EPI == EPI_BIAS_RESID_LN);
    constexpr bool RESID_LDS = RESID && C::BN != 256;
    constexpr bool MXOUT = (EPI == EPI_BIAS_GELU);       // fc1: output quantised to MXFP8 in the epilogue
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int K = g.K, nk = K >> 7;                       // K in elements = bytes
    TileWalk tw;
    tw.init(g, C::BM, C::BN);
    if (tw.j0 >= tw.cnt) return;
    const uint8_t* A8 = (const uint8_t*)g.A;
    const uint8_t* W8 = (const uint8_t*)g.W;

    // ---- staging addresses (gemm8.hip): piece p = LDS rows 8p .. 8p+7 of a slot; wave w issues pieces w and w + 8 ----
    const int rip = lane >> 3, pslot = lane & 7;
    const int slog = pslot ^ (((wave & 1) << 2) | (rip >> 1));
    const uint32_t voff_x = (uint32_t)(rip * 128 + slog * 16);
    constexpr bool SPLIT = (EPI == EPI_BIAS) && C::TI == 4;
    const int rho = ((wave & 1) << 3) | rip;
    const uint32_t voff_w = (uint32_t)(((rho >> 2) * (SPLIT ? 8 : 4 * C::TI) + (rho & 3)) * K + slog * 16);
    const int wu0 = (wave >> 2) * 16 * C::TI + ((wave >> 1) & 1) * 4;
    const int wu1 = SPLIT ? wu0 + 32 : (C::BN == 256) ? wu0 + 8 : (wave >> 1) * 16 * C::TI + 8;
    const size_t x64 = (size_t)(K >> 7) * 8192;           // + 64 rows (one block row of the 64 x 128 code blocks)
    const char* xb = nullptr;
    const char* wb = nullptr;
    const uint8_t* sb_ = nullptr;                         // scale dwords of this wave's 64-row group of X half 0 (half 1: + 2 block rows)
    const size_t s64 = (size_t)(K >> 5) * 64;             // scale bytes of one 64-row group
    const int frow = lane & 15, fg = lane >> 4;
    const uint32_t voff_s = (uint32_t)(fg * 64 + frow * 4);
    auto set_tile = [&](int m0, int n0) {
        xb = (const char*)(A8 + (size_t)(m0 >> 6) * x64) + (size_t)wave * 8 * 128;
        wb = (const char*)(W8 + (size_t)n0 * K);
        sb_ = g.a_scales + ((size_t)(m0 >> 6) + wr) * s64;
    };
    // DMA of one slot of K-tile kt (of the issue tile) into ring buffer B; the X slots are followed by their scale dword
    int sx0[2], sx1[2];                                   // activation scales: X half 0 / 1, alternating with the ring buffer
    auto issue = [&](int which, int B, int kt, int& sreg) {
        char* dst = smem + B * C::BUF + wave * 1024;
        uint32_t vx = voff_x, vw = voff_w, vs = voff_s;
        asm volatile("" : "+v"(vx), "+v"(vw), "+v"(vs));
        if (which < 2) {
            const char* src = xb + (size_t)(which * 2) * x64 + (size_t)kt * 8192 + vx;
            glds16(src, dst + which * C::HALF);
            glds16(src + x64, dst + which * C::HALF + 8192);
            load_scale(sreg, vs, sb_ + (size_t)(which * 2) * s64 + (size_t)kt * 256);
        } else if (which == 2) {
            const char* src = wb + ((size_t)wu0 * K + (size_t)kt * 128) + vw;
            glds16(src, dst + C::OFF_W0);
            glds16(src + (size_t)32 * C::TI * K, dst + C::OFF_W0 + 8192);
        } else {
            const char* src = wb + ((size_t)wu1 * K + (size_t)kt * 128) + vw;
            glds16(src, dst + C::OFF_W1);
            if (C::NW1 == 2) glds16(src + (size_t)32 * C::TI * K, dst + C::OFF_W1 + 8192);
        }
    };

    // ---- fragment read offsets (bytes inside a slot): the two k-half reads of a row = one operand of the 128-deep instruction ----
    const int foff = frow * 128 + ((fg ^ ((frow >> 1) & 7)) << 4);
    const int xoff = wr * 64 * 128 + foff;
    const int w0off = wc * 32 * 128 + foff;
    const int w1off = wc * 16 * C::NF1 * 128 + foff;

    f32x4 acc[C::TI][8];
    // fragment registers: one 8-register operand per 16-row fragment = the row's two k-half reads (slots g and g + 4)
    i32x8 xs[4], fa[2], fb[2];
    auto rd8 = [&](const char* slot, int off) {   // off = byte offset of the k-half 0 piece inside the slot; k-half 1 = off ^ 64
        const u32x4 lo = *(const u32x4*)(slot + off);
        const u32x4 hi = *(const u32x4*)(slot + (off ^ 64));
        return cat8(lo, hi);
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int f = 0; f < C::TI; ++f)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    const int sone = 0x7f7f7f7f;                          // weight block scale 1.0 (E8M0 byte 127)
    auto mma_half = [&](int sdw, int jbase) {
        // one X half against W0 and W1: TI x 4 instructions; op_sel j picks the scale byte of fragment j out of the lane's dword
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            mfma_mx_acc<0>(acc[p][jbase + 0], fa[p], xs[0], sone, sdw);
            mfma_mx_acc<1>(acc[p][jbase + 1], fa[p], xs[1], sone, sdw);
            mfma_mx_acc<2>(acc[p][jbase + 2], fa[p], xs[2], sone, sdw);
            mfma_mx_acc<3>(acc[p][jbase + 3], fa[p], xs[3], sone, sdw);
        }
#pragma unroll
        for (int p = 0; p < C::NF1; ++p) {
            mfma_mx_acc<0>(acc[2 + p][jbase + 0], fb[p], xs[0], sone, sdw);
            mfma_mx_acc<1>(acc[2 + p][jbase + 1], fb[p], xs[1], sone, sdw);
            mfma_mx_acc<2>(acc[2 + p][jbase + 2], fb[p], xs[2], sone, sdw);
            mfma_mx_acc<3>(acc[2 + p][jbase + 3], fb[p], xs[3], sone, sdw);
        }
    };

    // one K-tile t in ring buffer B.  Issue order and what each counted wait retires (loads return in order):
    //   LA(t): X1(t+1) [2 DMA + 1 scale -> sx1[B^1]]          wait leaves LB(t-1) + LA(t) in flight: retires X1(t) and sx1[B]
    //   LB(t): W0, X0, W1 (t+2) [4 + NW1 DMA + 1 scale -> sx0[B]]   wait leaves LA(t) + LB(t): retires X0/W0/W1(t+1) and sx0[B^1]
    // Scale registers: MA(t) reads sx0[B] (loaded in LB(t-2), retired by LB(t-1)'s wait) and LB(t) -- after MA(t) -- reloads it for
    // K-tile t+2; MB(t) reads sx1[B] (loaded in LA(t-1), retired by LA(t)'s wait), LA(t+1) reloads it for K-tile t+2.
    constexpr int NLA = 3, NLB = 5 + C::NW1;             // vector-memory operations a load section issues
    constexpr int NKEEP = NLA + NLB;
    auto ktile = [&](auto Bc, auto Mc, int kA, int kB, bool swB, int nm0, int nn0) {
        constexpr int B = decltype(Bc)::value;
        constexpr int MODE = decltype(Mc)::value;        // 1 = first K-tile of a tile (no wait in LA), 2 = last (deeper wait in LB)
        const char* sb = smem + B * C::BUF;
        // ---------------- LA: X0, W0, W1 | DMA X1(t+1)
#pragma unroll
        for (int p = 0; p < 2; ++p) fa[p] = rd8(sb + C::OFF_W0, w0off + p * 2048);
#pragma unroll
        for (int p = 0; p < C::NF1; ++p) fb[p] = rd8(sb + C::OFF_W1, w1off + p * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[j] = rd8(sb + C::OFF_X0, xoff + j * 2048);
        __builtin_amdgcn_sched_barrier(0);
        issue(1, B ^ 1, kA, sx1[B ^ 1]);
        if constexpr (MODE != 1) wait_vm<NKEEP>();
        asm volatile("" : "+v"(sx1[B]));     // retired by the wait above: re-defined HERE for the compiler, so that no copy or use of it can move in front of the wait
        wait_lgkm<0>();
        bar();
        __builtin_amdgcn_s_setprio(1);
        mma_half(sx0[B], 0);
        __builtin_amdgcn_s_setprio(0);
        bar();
        // ---------------- LB: X1 | DMA X0, W0, W1 (t+2)
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[j] = rd8(sb + C::OFF_X1, xoff + j * 2048);
        __builtin_amdgcn_sched_barrier(0);
        if (swB) set_tile(nm0, nn0);
        int dummy;
        issue(2, B, kB, dummy);
        issue(0, B, kB, sx0[B]);
        issue(3, B, kB, dummy);
        if constexpr (MODE == 2) wait_vm<NLB>(); else wait_vm<NKEEP>();
        asm volatile("" : "+v"(sx0[B ^ 1]));   // retired by the wait above (read by the next K-tile's first MFMA section)
        wait_lgkm<0>();
        bar();
        __builtin_amdgcn_s_setprio(1);
        mma_half(sx1[B], 4);
        __builtin_amdgcn_s_setprio(0);
        bar();
    };
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    // (re)start of the ring on the issue tile: K-tile 0 complete (incl. both scale dwords), X0 / W0 / W1 + scale of K-tile 1 in flight
    auto ring_start = [&]() {
        int dummy;
        issue(2, 0, 0, dummy); issue(0, 0, 0, sx0[0]); issue(3, 0, 0, dummy); issue(1, 0, 0, sx1[0]);
        issue(2, 1, 1, dummy); issue(0, 1, 1, sx0[1]); issue(3, 1, 1, dummy);
        wait_vm<NLB>();
        bar();
        if (wr) bar();
    };

    int t = tw.j0, m0, n0;
    tw.origin(t, g.reverse, C::BM, C::BN, m0, n0);
    set_tile(m0, n0);
    ring_start();

    for (;;) {
        zero_acc();
        const bool has_next = t + tw.nloc < tw.cnt;
        int nm0 = m0, nn0 = n0;
        if (has_next) tw.origin(t + tw.nloc, g.reverse, C::BM, C::BN, nm0, nn0);
        ktile(B0{}, M1{}, 1, 2, false, 0, 0);
        ktile(B1{}, M0{}, 2, 3, false, 0, 0);
        for (int kt = 2; kt < nk - 2; kt += 2) {
            ktile(B0{}, M0{}, kt + 1, kt + 2, false, 0, 0);
            ktile(B1{}, M0{}, kt + 2, kt + 3, false, 0, 0);
        }
        ktile(B0{}, M0{}, nk - 1, 0, true, nm0, nn0);
        ktile(B1{}, M2{}, 0, 1, false, 0, 0);

        // ---------------- epilogue of tile (m0, n0) ----------------
        // the MFMAs are inline asm: hipcc's hazard recognizer does not know that the accumulators were written by the matrix pipe.  The last MFMA section is
        // followed by a barrier and the epilogue's own address arithmetic, but the wait states an MFMA result needs before a VALU read (<= 18) are spent
        // HERE explicitly, once per tile, so that no compiler version can schedule the first read too early
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        int frow_e = frow, fg_e = fg;
        asm volatile("" : "+v"(frow_e), "+v"(fg_e));
        if constexpr (!RESID_LDS) { if (!wr) bar(); }
        if constexpr (RESID && !RESID_LDS) {
            // residual epilogue straight from registers (256 x 256 tiles; gemm8.hip): v = acc * w_scale + bias + (hi + lo)
            const int nb = n0 + wc * 64 + fg_e * 16;
            const int mrow = m0 + wr * 64 + frow_e;
            uint16_t* out_hi = (uint16_t*)g.out;
            uint16_t* out_lo = out_hi + g.plane;
            const uint16_t* aux_hi = (const uint16_t*)g.aux;
            const uint16_t* aux_lo = aux_hi + g.plane;
            f32x4 bias4[4], ws4[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) { bias4[f] = *(const f32x4*)(g.bias + nb + f * 4); ws4[f] = *(const f32x4*)(g.w_scale + nb + f * 4); }
            const int gran = g.N >> 6;
#pragma unroll
            for (int J = 0; J < 8; ++J) {
                const int m = mrow + (J >> 2) * 128 + (J & 3) * 16;
                const size_t o = (size_t)m * g.ldo + nb;
                u32x4 r[4];
                r[0] = *(const u32x4*)(aux_hi + o);
                r[1] = *(const u32x4*)(aux_hi + o + 8);
                r[2] = *(const u32x4*)(aux_lo + o);
                r[3] = *(const u32x4*)(aux_lo + o + 8);
                float v[16];
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = f * 4 + e;
                        const uint32_t wh = r[c >> 3][(c & 7) >> 1], wl = r[2 + (c >> 3)][(c & 7) >> 1];
                        const int sh = (c & 1) * 16;
                        v[c] = __builtin_fmaf(acc[f][J][e], ws4[f][e], bias4[f][e]) + (from_bits<T>((uint16_t)(wh >> sh)) + from_bits<T>((uint16_t)(wl >> sh)));
                    }
                u32x4 oh[2], ol[2];
#pragma unroll
                for (int c = 0; c < 16; c += 2) {
                    uint32_t h_, l_;
                    split_planes2<T>(v[c], v[c + 1], h_, l_);
                    oh[c >> 3][(c & 7) >> 1] = h_;
                    ol[c >> 3][(c & 7) >> 1] = l_;
                }
                *(u32x4*)(out_hi + o) = oh[0];
                *(u32x4*)(out_hi + o + 8) = oh[1];
                *(u32x4*)(out_lo + o) = ol[0];
                *(u32x4*)(out_lo + o + 8) = ol[1];
                float sa = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                float sb2 = ((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15]));
                float s1 = sa + sb2;
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
                if (fg_e == 0) *(float2*)(g.stats_out + ((size_t)m * gran + (nb >> 6)) * 2) = float2{s1, s2};
            }
        } else if constexpr (MXOUT) {
            // mlp.fc1: v = gelu(acc * w_scale + bias), written as MXFP8 (the A operand of mlp.fc2).  Lane (fg_e, frow_e) holds, for each
            // of its 8 rows, 16 CONSECUTIVE columns n0 + wc 64 + fg_e 16 + [0, 16): a 32-column block is this lane and lane ^ 16.
            const int nb = n0 + wc * 64 + fg_e * 16;
            const int mrow = m0 + wr * 64 + frow_e;
            f32x4 bias4[4], ws4[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) { bias4[f] = *(const f32x4*)(g.bias + nb + f * 4); ws4[f] = *(const f32x4*)(g.w_scale + nb + f * 4); }
            uint8_t* codes = (uint8_t*)g.out;
            const size_t NO = (size_t)g.ldo;                              // width of the output = K of the consuming GEMM
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t sdw = 0;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int J = h * 4 + jj;
                    float v[16], amax = 0.f;
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = gelu_erf(__builtin_fmaf(acc[f][J][e], ws4[f][e], bias4[f][e]));
                            v[f * 4 + e] = x;
                            amax = fmaxf(amax, fabsf(x));
                        }
                    amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
                    const uint32_t E = mx_scale_byte(amax);
                    const float inv = mx_inv_scale(E);
                    u32x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = mx_pack4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3], inv);
                    const int m = mrow + h * 128 + jj * 16;
                    *(u32x4*)(codes + mx_code_off((size_t)m, (size_t)nb, NO)) = o;
                    sdw |= E << (8 * jj);
                }
                // rows frow_e + 16 jj of row group (m0 >> 6) + 2 h + wr: one dword per (block, frow_e); written by the even lane of the pair
                if (!(fg_e & 1))
                    *(uint32_t*)(g.out_scales + ((((size_t)((m0 >> 6) + 2 * h + wr)) * (NO >> 5) + (size_t)(nb >> 5)) << 6) + frow_e * 4) = sdw;
            }
        } else if constexpr (!RESID) {
            // attn.qkv: v = acc * w_scale + bias -> 16-bit, straight from registers (SPLIT column layout, optional 64x64-blocked output)
            const int nb = SPLIT ? n0 + wc * 64 + fg_e * 8 : n0 + wc * 16 * C::TI + fg_e * 4 * C::TI;
            auto fcol = [&](int f) { return SPLIT ? (f >> 1) * 32 + (f & 1) * 4 : f * 4; };
            const int mrow = m0 + wr * 64 + frow_e;
            f32x4 bias4[C::TI], ws4[C::TI];
#pragma unroll
            for (int f = 0; f < C::TI; ++f) { bias4[f] = *(const f32x4*)(g.bias + nb + fcol(f)); ws4[f] = *(const f32x4*)(g.w_scale + nb + fcol(f)); }
            uint16_t* obase = g.out_blocked
                ? (uint16_t*)g.out + (((size_t)(mrow >> 6) * (g.ldo >> 6) + (nb >> 6)) << 12) + ((mrow & 63) << 6) + (nb & 63)
                : (uint16_t*)g.out + (size_t)mrow * g.ldo + nb;
            const size_t step16 = g.out_blocked ? (size_t)16 * 64 : (size_t)16 * g.ldo;
            const size_t step128 = g.out_blocked ? ((size_t)2 * (g.ldo >> 6) << 12) : (size_t)128 * g.ldo;
#pragma unroll
            for (int J = 0; J < 8; ++J) {
                uint32_t o[2 * C::TI];
#pragma unroll
                for (int f = 0; f < C::TI; ++f) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[f][J][r], ws4[f][r], bias4[f][r]);
                    o[2 * f] = pack2<T>(v[0], v[1]);
                    o[2 * f + 1] = pack2<T>(v[2], v[3]);
                }
                uint16_t* dst = obase + (size_t)(J >> 2) * step128 + (size_t)(J & 3) * step16;
                if constexpr (C::TI == 4) {
                    *(u32x4*)dst = u32x4{o[0], o[1], o[2], o[3]};
                    *(u32x4*)(dst + 32) = u32x4{o[4], o[5], o[6], o[7]};
                } else {
                    *(u32x2*)dst = u32x2{o[0], o[1]};
                    *(u32x2*)(dst + 4) = u32x2{o[2], o[3]};
                    *(u32x2*)(dst + 8) = u32x2{o[4], o[5]};
                }
            }
        } else {
            // residual epilogue through LDS (256 x 192 tiles; gemm8.hip): the ring is drained first and restarted afterwards
            constexpr int ROWBYTES = C::BN * 4 + 16;
            constexpr int JPP = (C::BN == 256) ? 2 : 4;
            constexpr int CR = 32 * JPP;
            constexpr int NPASS = 256 / CR;
            constexpr int CPR = C::BN / 8;
            constexpr int NCH = CR * CPR / C::NT;
            constexpr int GR = C::BN / 64;
            static_assert(NCH * C::NT == CR * CPR, "chunks must split evenly over threads");
            static_assert(CR * ROWBYTES + C::BM * GR * 8 <= 160 * 1024, "LDS");
            float* statbuf = (float*)(smem + CR * ROWBYTES);
            uint16_t* out_hi = (uint16_t*)g.out;
            uint16_t* out_lo = out_hi + g.plane;
            const uint16_t* aux_hi = (const uint16_t*)g.aux;
            const uint16_t* aux_lo = aux_hi + g.plane;
            const int nl = wc * 16 * C::TI + fg_e * 4 * C::TI;
            f32x4 bias4[C::TI], ws4[C::TI];
#pragma unroll
            for (int f = 0; f < C::TI; ++f) { bias4[f] = *(const f32x4*)(g.bias + n0 + nl + f * 4); ws4[f] = *(const f32x4*)(g.w_scale + n0 + nl + f * 4); }
            auto tile_row = [&](int p, int lr) {
                return (p / (4 / JPP)) * 128 + (lr / (16 * JPP)) * 64 + ((p % (4 / JPP)) * JPP + (lr / 16) % JPP) * 16 + (lr & 15);
            };
            wait_vm<0>();
            // The scale dwords the tile-boundary run-ahead fetched are dead in this variant (ring_start below fetches them again), and a
            // register hipcc considers dead is reused at once -- while the untracked load into it is still in flight (this was a memory
            // fault: a landed scale overwrote an epilogue address).  Keep the four registers allocated until the loads have retired.
            asm volatile("" ::"v"(sx0[0]), "v"(sx0[1]), "v"(sx1[0]), "v"(sx1[1]));
            if (!wr) bar();
            __syncthreads();
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
                    for (int f = 0; f < C::TI; ++f) {
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[f][J][r], ws4[f][r], bias4[f][r]);
                        *(f32x4*)(lrow + f * 16) = v;
                    }
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
                    for (int e = 0; e < 8; e += 2) { uint32_t h_, l_; split_planes2<T>(v[e], v[e + 1], h_, l_); oh[e >> 1] = h_; ol[e >> 1] = l_; }
                    const size_t so = orow_q[q] + n0 + ch * 8;
                    *(u32x4*)(out_hi + so) = oh;
                    *(u32x4*)(out_lo + so) = ol;
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
            for (int i = tid; i < C::BM * GR; i += C::NT) {
                const int trow = i / GR, gi = i - trow * GR;
                *(float2*)(g.stats_out + ((size_t)(m0 + trow) * (g.N / 64) + ((n0 >> 6) + gi)) * 2) = *(const float2*)(statbuf + i * 2);
            }
            __syncthreads();
            if (has_next) ring_start();
        }
        if constexpr (!RESID_LDS) { if (wr) bar(); }
        if (!has_next) break;
        t += tw.nloc;
        m0 = nm0;
        n0 = nn0;
    }
    wait_vm<0>();
    asm volatile("" ::"v"(sx0[0]), "v"(sx0[1]), "v"(sx1[0]), "v"(sx1[1]));   // see the residual epilogue: live until the run-ahead loads have landed
    if constexpr (!RESID_LDS) {
        if (!wr) bar();
    }
}

template <int EPI, class C>
static hipError_t launch8f(const GemmArgs& a, hipStream_t s) {
    auto kern = gemm8f_kernel<EPI, C>;
    constexpr int LDS = (EPI == EPI_BIAS_RESID_LN && C::BN != 256) ? ((128 * (C::BN * 4 + 16) + C::BM * (C::BN / 64) * 8) > C::RING ? (128 * (C::BN * 4 + 16) + C::BM * (C::BN / 64) * 8) : C::RING) : C::RING;
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    const int tiles = (a.M / C::BM) * (a.N / C::BN);
    const int grid = tiles < 256 ? tiles : 256;   // below 256 tiles: one workgroup per tile (TileWalk handles any count)
    if (grid < 8) return hipErrorInvalidValue;
    if (a.desc) snprintf(a.desc, a.desc_cap, "gemm8f_kernel<%d, G8<%d, %d>>", EPI, C::BN, C::BM);   // as rocprofv3 prints it
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), LDS, s, a);
    return hipGetLastError();
}

bool gemm8f_supported(int epi, const GemmArgs& a, int bn) {
    if (epi != EPI_BIAS && epi != EPI_BIAS_GELU && epi != EPI_BIAS_RESID_LN) return false;
    if (bn != 256 && bn != 192) return false;
    if (a.M % 256 || a.N % bn || a.K % 256 || a.K < 512) return false;            // K-tiles of 128, an even number of them, >= 4
    if (!a.a_scales || !a.w_scale) return false;
    if ((size_t)a.w_rows * a.K >= (1ull << 32)) return false;                     // 32-bit per-lane offsets
    if ((a.M / 256) * (a.N / bn) < 8) return false;
    if (epi == EPI_BIAS_RESID_LN) return a.ldo == a.N && a.plane && a.stats_out && !a.out_blocked;
    if (epi == EPI_BIAS_GELU) return bn == 256 && a.ldo == a.N && a.out_scales && a.N % 128 == 0;
    if ((size_t)a.M * a.N >= (1ull << 31)) return false;
    return bn == 256 && a.ldo == a.N;
}

hipError_t gemm8f_launch(int epi, const GemmArgs& a, int bn, hipStream_t s) {
    if (!gemm8f_supported(epi, a, bn)) return hipErrorInvalidValue;
    if (epi == EPI_BIAS) return launch8f<EPI_BIAS, G8<256>>(a, s);
    if (epi == EPI_BIAS_GELU) return launch8f<EPI_BIAS_GELU, G8<256>>(a, s);
    if (bn == 256) return launch8f<EPI_BIAS_RESID_LN, G8<256>>(a, s);
    return launch8f<EPI_BIAS_RESID_LN, G8<192>>(a, s);
}

}  // namespace vp
