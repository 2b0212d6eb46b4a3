// attn.qkv + the attention core in ONE kernel (head dim 64): the [M, 3D] qkv tensor never exists in HBM.
//
//   tile = (crop pair p, head h):  [384 token rows of crops 2p, 2p+1] x [q_h | k_h | v_h] (192 weight rows), K = D
//   1. GEMM  C[384 x 192] = X_hi[384 x K] . Wh_h[192 x K]^T on the 8-phase schedule of gemm8.hip (two load + two MFMA sections per K-tile, two
//      wave groups one barrier apart, counted vmcnt, LDS-DMA ring of two K-tiles) with a 384 x 192 geometry: the X halves are the two CROPS
//      (192 rows = 24 DMA pieces = 3 per wave), W = W0 (128 rows: q and k of the head) + W1 (64 rows: v).  Wave (wr, wc) owns rows
//      wr 96 + [0, 96) of EACH crop and 16 columns of each of q, k, v: 36 accumulator fragments (144 registers).
//   2. epilogue: LayerNorm fold + bias (common.h::ln_fold, the qkv epilogue's arithmetic), rounded to 16 bit and written to LDS in the
//      attention kernel's layouts -- per crop Q and K as [192][128 B] rows with the XOR swizzle, V as [192 keys][16 d] sub-tile pairs --
//      144 KiB for the two crops = exactly the operand ring, which is dead by then.
//   3. the attention core of attention.hip (S^T = K Q^T in registers, fp32 softmax, O^T = V^T P^T through ds_read_b64_tr_b16), waves 0-3 on
//      crop 2p, waves 4-7 on crop 2p+1, Q fragments from LDS instead of HBM; output rows straight to `y`.
//   The ring is drained before the epilogue and restarted on the next tile after the attention phase (both need the whole LDS).
// Same accumulation order, same fold, same roundings, same attention arithmetic as gemm8 / gemm.hip + attention.hip: `y` is BIT-IDENTICAL to
// the unfused path (tests/test_gpu_api.py flips VP_FUSE_QKV_ATTN).  What it removes per layer at 256 crops: 226 MB written + 226 MB read,
// qkv's write-back phase and one launch.
#include <cstdio>
#include <cstdlib>

#include "gemm8_common.h"

namespace vp {

namespace {
struct QA {
    static constexpr int NT = 512;
    static constexpr int XS = 192 * 128;                 // one crop's 192 rows x 64 k (16-bit) = 24 KiB; also one of Q / K / V in the attention phase
    static constexpr int W0S = 128 * 128, W1S = 64 * 128;
    static constexpr int OFF_X0 = 0, OFF_X1 = XS, OFF_W0 = 2 * XS, OFF_W1 = 2 * XS + W0S;
    static constexpr int BUF = 2 * XS + W0S + W1S;       // 72 KiB
    static constexpr int RING = 2 * BUF;                 // 144 KiB
    // attention-phase layout: [Q0 | K0 | Q1 | K1 | V0 | V1] (crop 0 / 1 of the pair).  Q and K are dead once every wave has its S^T = K Q^T
    // (all three query tiles of a wave are multiplied up front), and ring buffer 0 (X0, X1, W0, W1 of a K-tile = the first 72 KiB) lies inside
    // Q0 | K0 | Q1, the X0 slot of buffer 1 is K1: the next tile's first K-tile streams in under the softmax and the P V products
    static constexpr int VSUB = 192 * 32;                // one [192 keys][16 d] V sub-tile
    static constexpr int V_BASE = 4 * XS;
    static_assert(6 * XS == RING && BUF <= 3 * XS && BUF + XS <= 4 * XS, "the attention phase reuses exactly the ring; buffer 0 + X0 of buffer 1 inside Q0 K0 Q1 K1");
};

__device__ __forceinline__ u32x2 lds_read_tr16(const char* p) {
    typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h4;
    typedef __attribute__((address_space(3))) h4* lds_h4;
    const h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(p));
    return __builtin_bit_cast(u32x2, v);
}
}  // namespace

template <class T>
__global__ __launch_bounds__(512, 2) void qkvattn_kernel(QkvAttnArgs g) {
    constexpr int QT = 3;   // query tiles of a wave processed together in the attention phase (all of them: see QA)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int K = g.D, nk = K >> 6;
    // tile walk: XCD x (= blockIdx & 7) owns a contiguous range of (pair, head) tiles, head fastest: the workgroups resident on one XCD
    // share a few X panels and the whole weight matrix in its L2
    const int ntiles = g.npairs * g.heads;
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3;
    const int nloc = (gridDim.x >> 3) + (xcd < (int)(gridDim.x & 7) ? 1 : 0);   // workgroups on this XCD (fewer than 256 tiles: one workgroup per tile, any count)
    const int tq = ntiles >> 3, tr8 = ntiles & 7;
    const int tbase = (xcd < tr8) ? xcd * (tq + 1) : tr8 * (tq + 1) + (xcd - tr8) * tq;
    const int tcnt = tq + (xcd < tr8 ? 1 : 0);
    if (j0 >= tcnt) return;

    // ---- staging: piece p = LDS rows 8p .. 8p+7 of a slot (one DMA wave-instruction); wave w issues pieces w, w + 8 (, w + 16) ----
    const int rip = lane >> 3, pslot = lane & 7;
    const int slog = pslot ^ (((wave & 1) << 2) | (rip >> 1));
    const uint32_t voff = (uint32_t)(rip * K + slog * 8) * 2u;
    const size_t r64 = (size_t)64 * K * 2;               // + 64 rows
    const char* xb = nullptr;
    const char* wb = nullptr;
    size_t xb1off = 0;                                    // byte offset of the pair's second crop from its first (0 when an odd batch's last crop stands in for it)
    int pair = 0, head = 0;                               // of the tile whose accumulators / attention phase are live
    int ipair = 0, ihead = 0;                             // of the ISSUE tile (xb / wb): one tile ahead once the prefetch inside the attention phase has run
    auto set_tile = [&](int t) {
        const int tile = tbase + t;
        ipair = tile / g.heads;
        ihead = tile - ipair * g.heads;
        xb = (const char*)(g.x_hi + ((size_t)ipair * 384 + wave * 8) * K);
        // odd batch: the last pair's second half is the last crop once more (same inputs, same arithmetic, the same bytes stored twice)
        xb1off = (2 * ipair + 1 < g.ncrops) ? (size_t)192 * K * 2 : 0;
        wb = (const char*)(g.wh + ((size_t)ihead * 192 + wave * 8) * K);
    };
    auto issue = [&](int which, int B, int kt) {
        char* dst = smem + B * QA::BUF + wave * 1024;
        uint32_t v = voff;
        asm volatile("" : "+v"(v));                       // addresses formed at their use (gemm8.hip: no pointers kept live across the sections)
        if (which < 2) {   // X half = crop `which` of the pair
            const char* src = xb + (which ? xb1off : 0) + (size_t)kt * 128 + v;
            glds16(src, dst + which * QA::XS);
            glds16(src + r64, dst + which * QA::XS + 8192);
            glds16(src + 2 * r64, dst + which * QA::XS + 16384);
        } else if (which == 2) {
            const char* src = wb + (size_t)kt * 128 + v;
            glds16(src, dst + QA::OFF_W0);
            glds16(src + r64, dst + QA::OFF_W0 + 8192);
        } else {
            const char* src = wb + 2 * r64 + (size_t)kt * 128 + v;
            glds16(src, dst + QA::OFF_W1);
        }
    };

    // ---- fragment read offsets ----
    const int frow = lane & 15, fg = lane >> 4;
    const int foff = frow * 128 + ((fg ^ ((frow >> 1) & 7)) << 4);
    // X: + wr 96 rows (+ j * 2048, j < 6; (row >> 1) & 7 of row wr 96 + j 16 + frow is (frow >> 1) & 7); W0: + wc 32 rows (+ p * 2048: p = 0 -> q columns,
    // p = 1 -> k columns of this wave); W1: + wc 16 rows (v columns).  Only `foff` stays live across the K-loop: the per-slot / per-buffer / per-k-half
    // LDS addresses (16 combinations, beyond the 16-bit immediate of ds_read) are rebuilt in every load section from an opaque copy -- hoisted
    // out of the loop they cost 14 VGPRs and, beside 144 accumulator + 72 fragment registers, spills that were reloaded INSIDE the K-loop
    const int xadd = wr * 96 * 128, w0add = wc * 32 * 128, w1add = wc * 16 * 128;

    f32x4 acc[3][12];                                     // [q | k | v][crop 0: j 0-5, crop 1: j 6-11]
    u32x4 xs[6][2], fa[2][2], fb[2];

    constexpr int NKEEP = 9;                              // DMA pieces of one LA (3) + one LB (6)
    auto ktile = [&](auto Bc, auto Mc, int kA, int kB) {
        constexpr int B = decltype(Bc)::value;
        constexpr int MODE = decltype(Mc)::value;        // 1 = first K-tile after a ring start (no wait in LA); 2 / 3 = the tile's last two K-tiles:
                                                         // nothing of this tile is left to fetch -- 2: LB issues nothing and waits for all but this LA's
                                                         // X1 pieces; 3: no issue at all, LA drains the queue (X1 of the last K-tile)
        const char* sb = smem + B * QA::BUF;
        int fo = foff;
        asm volatile("" : "+v"(fo));
        const int xoff = xadd + fo, w0off = w0add + fo, w1off = w1add + fo;
        // ---------------- LA: X0, W0, W1 | DMA X1(t+1)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[p][kk] = *(const u32x4*)(sb + QA::OFF_W0 + ((w0off + p * 2048) ^ (kk << 6)));
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb[kk] = *(const u32x4*)(sb + QA::OFF_W1 + (w1off ^ (kk << 6)));
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xs[j][kk] = *(const u32x4*)(sb + QA::OFF_X0 + ((xoff + j * 2048) ^ (kk << 6)));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE != 3) issue(1, B ^ 1, kA);
        if constexpr (MODE == 3) wait_vm<0>();
        else if constexpr (MODE != 1) wait_vm<NKEEP>();
        wait_lgkm<0>();
        bar();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                acc[0][j] = mfma16<T>(fa[0][kk], xs[j][kk], acc[0][j]);
                acc[1][j] = mfma16<T>(fa[1][kk], xs[j][kk], acc[1][j]);
                acc[2][j] = mfma16<T>(fb[kk], xs[j][kk], acc[2][j]);
            }
        __builtin_amdgcn_s_setprio(0);
        bar();
        // ---------------- LB: X1 | DMA X0, W0, W1 (t+2)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xs[j][kk] = *(const u32x4*)(sb + QA::OFF_X1 + ((xoff + j * 2048) ^ (kk << 6)));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE < 2) {
            issue(2, B, kB);
            issue(0, B, kB);
            issue(3, B, kB);
            wait_vm<NKEEP>();
        } else if constexpr (MODE == 2) {
            wait_vm<3>();   // X0 / W0 / W1 of the last K-tile (issued one LB ago) have landed; this LA's three X1 pieces stay in flight
        }
        wait_lgkm<0>();
        bar();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                acc[0][6 + j] = mfma16<T>(fa[0][kk], xs[j][kk], acc[0][6 + j]);
                acc[1][6 + j] = mfma16<T>(fa[1][kk], xs[j][kk], acc[1][6 + j]);
                acc[2][6 + j] = mfma16<T>(fb[kk], xs[j][kk], acc[2][6 + j]);
            }
        __builtin_amdgcn_s_setprio(0);
        bar();
    };
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    using M3 = std::integral_constant<int, 3>;
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    // start of the ring on the current tile: K-tile 0 landed and visible, X0 / W0 / W1 of K-tile 1 in flight (its X1 is issued by the first LA)
    auto ring_start = [&]() {
        issue(2, 0, 0); issue(0, 0, 0); issue(3, 0, 0); issue(1, 0, 0);
        issue(2, 1, 1); issue(0, 1, 1); issue(3, 1, 1);
        wait_vm<6>();
        bar();
        if (wr) bar();   // stagger: waves 4-7 run one barrier behind waves 0-3
    };

    int t = j0;
    set_tile(t);
    ring_start();
    for (;;) {
        pair = ipair;
        head = ihead;
        const bool has_next = t + nloc < tcnt;
        bool prefetched = false;
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int j = 0; j < 12; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        ktile(B0{}, M1{}, 1, 2);
        ktile(B1{}, M0{}, 2, 3);
        for (int kt = 2; kt < nk - 2; kt += 2) {
            if (VP_ABLATE(g) & 4) break;
            ktile(B0{}, M0{}, kt + 1, kt + 2);
            ktile(B1{}, M0{}, kt + 2, kt + 3);
        }
        // the last two K-tiles fetch nothing beyond this tile (the next tile's first K-tile streams in later, under the attention phase): until
        // round 4's end they kept the issue pattern by fetching K-tiles 0 / 1 of this tile once more -- 144 KB of dead DMA per tile
        ktile(B0{}, M2{}, nk - 1, 0);
        ktile(B1{}, M3{}, 0, 0);
        // operands of the epilogue (bias, row sums, this lane's 12 row statistics) are requested HERE, in front of the drain of the ring: their
        // L2 latency then hides behind the wait for the run-ahead DMAs and the two barriers instead of in front of the first LDS store
        int frow_e = frow, fg_e = fg;
        asm volatile("" : "+v"(frow_e), "+v"(fg_e));
        // this lane's weight rows (tile columns): fragment f = 0 / 1: head * 192 + wc 32 + f 16 + fg 4 + e; f = 2: head * 192 + 128 + wc 16 + fg 4 + e
        const int cb = head * 192 + fg_e * 4;
        f32x4 b4[3], s4[3];
        float2 stat[12];
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const int c = cb + (f < 2 ? wc * 32 + f * 16 : 128 + wc * 16);
            b4[f] = *(const f32x4*)(g.bh + c);
            s4[f] = *(const f32x4*)(g.sh + c);
        }
        const bool twin = 2 * pair + 1 >= g.ncrops;                                   // odd batch, last pair: its second half IS its first crop
        const float* rs0 = g.rowstat + 2 * ((size_t)pair * 384 + wr * 96 + frow_e);   // crop 0 of the pair; crop 1: + 2 * 192 floats
#pragma unroll
        for (int j = 0; j < 6; ++j) stat[j] = *(const float2*)(rs0 + 32 * j);          // crop 1's six follow inside the epilogue, under crop 0's stores
        wait_vm<0>();
        if (!wr) bar();     // undo the stagger: both groups meet here
        __syncthreads();    // every wave is done with the ring

        // ---------------- qkv epilogue: LayerNorm fold + bias -> 16 bit -> LDS (attention layouts) ----------------
        if (!(VP_ABLATE(g) & 2)) {
            const int dcol = wc * 16 + fg_e * 4;                                   // head-dim column of this lane's four values
            const int qk_byte = (fg_e & 1) * 8;                                    // inside the 16-byte slot d >> 3 = wc 2 + (fg >> 1)
            const int qk_slot = wc * 2 + (fg_e >> 1);
            const int v_sub = 2 * (wc >> 1) + (fg_e & 1);                          // attention.hip: d 8 ch .. + 3 -> sub-tile 2 (ch / 4), + 4 .. + 7 -> 2 (ch / 4) + 1
            const int v_byte = ((wc & 1) * 2 + (fg_e >> 1)) * 8;
            (void)dcol;
#pragma unroll
            for (int j = 6; j < 12; ++j) stat[j] = *(const float2*)(rs0 + (twin ? 0 : 384) + 32 * (j - 6));
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int crop = j / 6;
                const int row = wr * 96 + (j % 6) * 16 + frow_e;
                const float2 st = stat[j];
                char* cb_ = smem + crop * 2 * QA::XS;                              // Q of this crop; K: + XS; V: V_BASE + crop * XS
                const int rsw = (row >> 1) & 7;
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ln_fold(acc[f][j][e], st.x, s4[f][e], st.y, b4[f][e]);
                    const u32x2 o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
                    if (f < 2) *(u32x2*)(cb_ + f * QA::XS + row * 128 + ((qk_slot ^ rsw) << 4) + qk_byte) = o;
                    else *(u32x2*)(smem + QA::V_BASE + crop * QA::XS + v_sub * QA::VSUB + row * 32 + v_byte) = o;
                }
            }
        }
        __syncthreads();

        // ---------------- attention core (attention.hip, one query tile live at a time): waves 0-3 crop 0, waves 4-7 crop 1 ----------------
        if (!(VP_ABLATE(g) & 1)) {
            const int crop = wave >> 2, lw = wave & 3;
            const char* Qs = smem + crop * 2 * QA::XS;
            const char* Ks = Qs + QA::XS;
            const char* Vs = smem + QA::V_BASE + crop * QA::XS;
            // lane coordinates through an empty asm: every address of this phase is then formed here, after the K-loop, instead of being
            // hoisted in front of the tile loop and kept live (= spilled) across 144 accumulator + 72 fragment registers
            int fr = frow, fg = lane >> 4;
            asm volatile("" : "+v"(fr), "+v"(fg));
            const char* kfrag = Ks + fr * 128;
            const int kswz = (fr >> 1) & 7;
            const char* vfrag = Vs + (fg * 4 + (fr >> 2)) * 32 + (fr & 3) * 8;
            const size_t b = (size_t)pair * 2 + ((2 * pair + 1 < g.ncrops) ? crop : 0);
            // QT query tiles of the wave at a time.  QT = 3 (the accumulators are dead: 256 registers for this phase): every K / V^T fragment read from
            // LDS feeds three MFMAs; QT = 1: attention.hip's shipped form.  Per query tile the arithmetic is attention.hip's, instruction for instruction.
#pragma unroll
            for (int t0 = 0; t0 < 3; t0 += QT) {
                u32x4 qf[QT][2];
#pragma unroll
                for (int tq_ = 0; tq_ < QT; ++tq_)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) qf[tq_][kk] = *(const u32x4*)(Qs + ((lw * 3 + t0 + tq_) * 16 + fr) * 128 + (((kk * 4 + fg) ^ kswz) << 4));
                f32x4 s[QT][12];
#pragma unroll
                for (int tq_ = 0; tq_ < QT; ++tq_)
#pragma unroll
                    for (int kt = 0; kt < 12; ++kt) s[tq_][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < 12; ++kt)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const u32x4 kf = *(const u32x4*)(kfrag + kt * 16 * 128 + (((kk * 4 + fg) ^ kswz) << 4));
#pragma unroll
                        for (int tq_ = 0; tq_ < QT; ++tq_) s[tq_][kt] = mfma16<T>(kf, qf[tq_][kk], s[tq_][kt]);
                    }
                // every wave has its scores: Q and K are dead.  The next tile's K-tile 0 (ring buffer 0) and the X0 slot of its K-tile 1 start
                // streaming into their LDS space now, under the softmax and the P V products of this tile
                __syncthreads();
                if (has_next) {
                    set_tile(t + nloc);
                    issue(2, 0, 0); issue(0, 0, 0); issue(3, 0, 0); issue(1, 0, 0);
                    issue(0, 1, 1);
                    prefetched = true;
                }
                u32x4 pf[QT][6];
                float inv_l[QT];
#pragma unroll
                for (int tq_ = 0; tq_ < QT; ++tq_) {
                    float mx = -3.0e38f;
#pragma unroll
                    for (int kt = 0; kt < 12; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[tq_][kt][r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    float l = 0.f;
                    const float mb = mx * g.scale_log2e;
#pragma unroll
                    for (int kt = 0; kt < 12; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = softmax_p(s[tq_][kt][r], g.scale_log2e, mb);
                            s[tq_][kt][r] = p;
                            l += p;
                        }
                    l += __shfl_xor(l, 16, 64);
                    l += __shfl_xor(l, 32, 64);
                    inv_l[tq_] = 1.0f / l;
#pragma unroll
                    for (int kb = 0; kb < 6; ++kb) {
                        pf[tq_][kb][0] = pack2_nosat<T>(s[tq_][2 * kb][0], s[tq_][2 * kb][1]);
                        pf[tq_][kb][1] = pack2_nosat<T>(s[tq_][2 * kb][2], s[tq_][2 * kb][3]);
                        pf[tq_][kb][2] = pack2_nosat<T>(s[tq_][2 * kb + 1][0], s[tq_][2 * kb + 1][1]);
                        pf[tq_][kb][3] = pack2_nosat<T>(s[tq_][2 * kb + 1][2], s[tq_][2 * kb + 1][3]);
                    }
                }
#pragma unroll
                for (int dp = 0; dp < 4; dp += 2) {
                    f32x4 o[2][QT];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int tq_ = 0; tq_ < QT; ++tq_) o[u][tq_] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 6; ++kb)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const char* vp_ = vfrag + (dp + u) * QA::VSUB + kb * 1024;
                            const u32x2 lo = lds_read_tr16(vp_);
                            const u32x2 hi = lds_read_tr16(vp_ + 512);
                            const u32x4 vf = u32x4{lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
                            for (int tq_ = 0; tq_ < QT; ++tq_) o[u][tq_] = mfma16<T>(vf, pf[tq_][kb], o[u][tq_]);
                        }
#pragma unroll
                    for (int tq_ = 0; tq_ < QT; ++tq_) {
                        const int q = (lw * 3 + t0 + tq_) * 16 + fr;
                        uint16_t* dst = g.y + (b * 192 + q) * g.D + head * 64;
                        u32x4 w;
                        w[0] = pack2_nosat<T>(o[0][tq_][0] * inv_l[tq_], o[0][tq_][1] * inv_l[tq_]);
                        w[1] = pack2_nosat<T>(o[0][tq_][2] * inv_l[tq_], o[0][tq_][3] * inv_l[tq_]);
                        w[2] = pack2_nosat<T>(o[1][tq_][0] * inv_l[tq_], o[1][tq_][1] * inv_l[tq_]);
                        w[3] = pack2_nosat<T>(o[1][tq_][2] * inv_l[tq_], o[1][tq_][3] * inv_l[tq_]);
                        *(u32x4*)(dst + dp * 16 + fg * 8) = w;
                    }
                }
            }
        }
        if (!has_next) break;
        t += nloc;
        __syncthreads();    // every wave is done reading V before the rest of the ring refills the LDS
        if (prefetched) {   // K-tile 0 and X0 of K-tile 1 are on their way since the middle of the attention phase: W0, W1 of K-tile 1 complete ring_start's state
            issue(2, 1, 1); issue(3, 1, 1);
            wait_vm<6>();   // leaves X0 / W0 / W1 of K-tile 1 (this tile's output stores are older than any load the count lets pass)
            bar();
            if (wr) bar();
        } else {
            set_tile(t);
            ring_start();
        }
    }
}

// head-major copy of the LayerNorm-folded qkv weights for the fused kernel: row h 192 + r of the copy = the weight row whose output the
// kernel's wave / fragment geometry puts at tile column r (r < 128: wc = r >> 5, q (p = 0) or k (p = 1) column h 64 + wc 16 + (r & 15);
// r >= 128: v column h 64 + (r - 128)); bias and row sums permuted alike
__global__ void qkv_head_major_kernel(const uint16_t* __restrict__ w, const float* __restrict__ b, const float* __restrict__ s, uint16_t* __restrict__ wh,
                                      float* __restrict__ bh, float* __restrict__ sh, int D, int K) {
    const int dst = blockIdx.x;                           // 0 .. 3 D - 1
    const int h = dst / 192, r = dst - h * 192;
    int src;
    if (r < 128) {
        const int wc = r >> 5, p = (r >> 4) & 1, rho = r & 15;
        src = p * D + h * 64 + wc * 16 + rho;
    } else {
        src = 2 * D + h * 64 + (r - 128);
    }
    for (int k = threadIdx.x; k < K; k += blockDim.x) wh[(size_t)dst * K + k] = w[(size_t)src * K + k];
    if (threadIdx.x == 0) { bh[dst] = b[src]; sh[dst] = s[src]; }
}

hipError_t qkv_head_major_launch(const uint16_t* w, const float* b, const float* s, uint16_t* wh, float* bh, float* sh, int D, int K, hipStream_t st) {
    if (D % 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(qkv_head_major_kernel, dim3(3 * D), dim3(256), 0, st, w, b, s, wh, bh, sh, D, K);
    return hipGetLastError();
}

// head dim 80: rows h 256 + r of the copy = [q_h (80) | k_h (80) | v_h (80) | 16 zero rows] in their natural order (gemm8.hip places tile column c of a
// 256-wide tile where its 16-bit epilogues expect it: the permutation is applied by the DMA source addresses there, not here)
__global__ void qkv_head_major80_kernel(const uint16_t* __restrict__ w, const float* __restrict__ b, const float* __restrict__ s, uint16_t* __restrict__ wh,
                                        float* __restrict__ bh, float* __restrict__ sh, int D, int K) {
    const int dst = blockIdx.x;                           // 0 .. heads * 256 - 1
    const int h = dst >> 8, r = dst & 255;
    const int src = r < 240 ? (r / 80) * D + h * 80 + r % 80 : -1;
    for (int k = threadIdx.x; k < K; k += blockDim.x) wh[(size_t)dst * K + k] = src >= 0 ? w[(size_t)src * K + k] : (uint16_t)0;
    if (threadIdx.x == 0) { bh[dst] = src >= 0 ? b[src] : 0.f; sh[dst] = src >= 0 ? s[src] : 0.f; }
}

hipError_t qkv_head_major80_launch(const uint16_t* w, const float* b, const float* s, uint16_t* wh, float* bh, float* sh, int D, int K, int heads, hipStream_t st) {
    if (heads * 80 != D) return hipErrorInvalidValue;
    hipLaunchKernelGGL(qkv_head_major80_kernel, dim3(heads * 256), dim3(256), 0, st, w, b, s, wh, bh, sh, D, K);
    return hipGetLastError();
}

bool qkvattn_supported(const QkvAttnArgs& a) {
    if (a.D % 128 || a.D < 256 || a.heads * 64 != a.D || a.npairs <= 0 || (a.ncrops != 2 * a.npairs && a.ncrops != 2 * a.npairs - 1)) return false;
    if ((size_t)a.npairs * 384 * a.D * 2 >= (1ull << 32)) return false;   // 32-bit per-lane offsets are relative to the tile base: only the row span matters; kept conservative
    return a.npairs * a.heads >= 8;
}

hipError_t qkvattn_launch(int dtype, const QkvAttnArgs& a, hipStream_t s, char* desc, int desc_cap) {
    if (!qkvattn_supported(a)) return hipErrorInvalidValue;
    const int tiles = a.npairs * a.heads;
    const int grid = tiles < 256 ? tiles : 256;
    static bool attr_done[2][64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int di = dtype == DT_F16 ? 0 : 1;
    auto kern = dtype == DT_F16 ? qkvattn_kernel<F16> : qkvattn_kernel<BF16>;
    if (dev < 0 || dev >= 64 || !attr_done[di][dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, QA::RING);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_done[di][dev] = true;
    }
    if (desc) snprintf(desc, desc_cap, "qkvattn_kernel<%s>", dtype == DT_F16 ? "F16" : "BF16");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(QA::NT), QA::RING, s, a);
    return hipGetLastError();
}

}  // namespace vp
