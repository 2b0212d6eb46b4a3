// Fused scaled-dot-product attention for the fixed 192-token ViTPose sequence
// (vit.py:164-176): one workgroup per (crop, head); the 192x192 score tile never
// leaves the CU.
//
//  S^T = K Q^T   (MFMA A = K rows from LDS, B = Q rows straight from HBM)
//  softmax over keys in fp32 registers (scale folded into the exponent)
//  O^T = V^T P^T (MFMA A = V^T via the gfx950 LDS transpose read, B = P^T = the S^T
//                 accumulators re-packed)
//
// Computing the TRANSPOSED products makes every lane own one query column
// (q = lane & 15): row max / row sum need only two cross-lane steps (xor 16, 32), and
// the S^T accumulator registers of two 16-key tiles are directly the 8-element
// B fragment of the PV MFMA (the MFMA k index is a permutation-invariant sum: k slot
// (g, e) = key 32 kb + 4 g + e for e < 4, 32 kb + 16 + 4 g + e - 4 otherwise).
//
// V stays ROW-MAJOR in LDS, cut into [192 keys][16 d] sub-tiles (32-byte rows, so the 8
// rows a 32-lane half touches are one 256-byte bank row: conflict-free without padding);
// `ds_read_b64_tr_b16` hands lane i of a 16-lane group column i of a [4 keys][16 d] block,
// i.e. the V^T fragment, with no transposed staging pass.  When HD is a multiple of 32 the
// d columns of sub-tile pairs are interleaved (column c of sub-tile dt = d 32(dt/2) +
// 8(c/4) + 4(dt&1) + c%4) so that a lane's two O^T accumulators are 8 consecutive head-dim
// values of one query = one 16-byte store.  K (HD = 64) is kept unpadded with the GEMM's
// XOR swizzle.  LDS = 48 KiB at HD = 64 -> 3 blocks per CU.
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "mx8.h"

namespace vp {

static constexpr int T = 192;
#ifndef VP_ATTN_QSPLIT_DEFAULT
#define VP_ATTN_QSPLIT_DEFAULT 128   // (crop, head) pairs: B x 1 0.521 -> 0.495 ms, L x 1 1.215 -> 1.152, B x 8 0.949 -> 0.906; neutral at 128 pairs, slower from 192 on
#endif

template <int HD> struct AttnCfg {
    static constexpr int HDP = (HD + 31) / 32 * 32;   // head dim padded to the MFMA k step
    static constexpr bool KSWZ = (HD == 64);          // 128-byte rows, 16-byte slots XOR-swizzled
    static constexpr int KSTR = KSWZ ? 128 : HDP * 2 + 16;   // K row stride in bytes
    static constexpr int K_BYTES = T * KSTR;
    static constexpr int DT = HD / 16;                // [T][16] V sub-tiles
    static constexpr int VSUB = T * 32;               // bytes per sub-tile
    static constexpr bool PAIR = (DT % 2 == 0);       // interleaved sub-tile pairs -> 16-byte output stores
    static constexpr int LDS = K_BYTES + DT * VSUB;
};

__device__ __forceinline__ u32x2 lds_read_tr16(const char* p) {
    typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h4;
    typedef __attribute__((address_space(3))) h4* lds_h4;
    const h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(p));
    return __builtin_bit_cast(u32x2, v);
}

// MX (fp8 mode, HD = 64 only): the output is written as MXFP8 -- e4m3 codes in the 64 x 128-blocked layout + one E8M0 scale per 32 columns
// (csrc/mx8.h) -- the A operand of the fp8 attn.proj GEMM; `out` then points at the codes and `out_scales` at the scale bytes.  A block of 32
// output columns (half a head) of one query is the four lanes fg = 0..3 of a d-pair: amax by two cross-lane steps.
// QS (round 6, small batches): query parts per (crop, head).  QS = 3: three workgroups share a (crop, head), each wave owns ONE of its three query tiles
// (the same tile rows, the same arithmetic: bit-identical output) and every workgroup stages the whole K and V -- a single crop of ViTPose-B is then 36
// workgroups with a third of the serial work each instead of 12.
template <class Ty, int HD, int QT, bool MX = false, int QS = 1>
__global__ __launch_bounds__(256, QT == 3 ? 2 : 3) void attention_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                            int D, int heads, float scale_log2e, int blocked, uint8_t* __restrict__ out_scales = nullptr) {
    using C = AttnCfg<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + C::K_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Block order.  The dispatcher puts block i on XCD i % 8, so the heads of one crop land on eight different L2s.  When a head slice is not a whole
    // number of 128-byte lines (HD = 32: 64 bytes, HD = 80: 160 bytes) neighbouring heads share lines.  Inside super-groups of 32 block ids every XCD
    // therefore takes FOUR consecutive (crop, head) ids: neighbours meet in one L2 and the eight XCDs still walk the same crops together
    // (profiles/attention_order_r3.txt: attention of ViTPose-S -15 %, of ViTPose-H -6 %; an XCD-contiguous order over the whole launch gains less,
    // and with whole-line slices -- HD = 64 -- it is 4-11 % SLOWER than the plain order, which stays there).
    constexpr bool REMAP = (HD * 2) % 128 != 0;
    static_assert(QS == 1 || (QS == 3 && QT == 1), "query split: one tile per wave");
    constexpr int NTW = 3 / QS;                       // query tiles per wave
    const int part = QS == 1 ? 0 : (int)(blockIdx.x % QS);
    int bid = blockIdx.x / QS;
    if constexpr (REMAP) {
        constexpr int P = 4, G = 8 * P;
        const int i = bid;
        if (i < (int)(gridDim.x / QS / G) * G) bid = (i & ~(G - 1)) | ((i & 7) * P) | ((i >> 3) & (P - 1));
    }
    const int b = bid / heads, h = bid % heads;
    const size_t ld = (size_t)3 * D;
    // row `key` of this (crop, head)'s q / k / v slab (sel = 0 / 1 / 2).  Row-major qkv [M][3 D]: 2 HD bytes of a 6 D-byte row.  `blocked` (HD = 64 only):
    // qkv as the GEMM's 64 x 64-blocked output [M / 64][3 D / 64][64][64] -- a slab is then three CONTIGUOUS 8 KiB blocks instead of 192 pieces of 128 bytes
    // 6 D bytes apart, for the qkv epilogue's stores as for the loads here
    auto rowp = [&](int sel, int key) -> const uint16_t* {
        if (HD == 64 && blocked)
            return qkv + ((((size_t)(b * 3 + (key >> 6)) * (ld >> 6)) + (size_t)sel * (D >> 6) + h) << 12) + ((key & 63) << 6);
        return qkv + ((size_t)b * T + key) * ld + (size_t)sel * D + (size_t)h * HD;
    };

    // ---- global loads first (K, Q, then V), LDS writes as the data arrives: K is written and made visible before
    //      V has to be there, so the first QK^T + softmax run in the shadow of the V loads ----
    constexpr int CH = HD / 8;          // 16-B chunks per row
    constexpr int CHP = C::HDP / 8;
    constexpr int NKL = (T * CHP + 255) / 256, NVL = (T * CH + 255) / 256;
    const int fr = lane & 15, fg = lane >> 4;
    constexpr int KS = C::HDP / 32;     // k steps of QK^T
    constexpr int DT = C::DT;
    u32x4 kreg[NKL], vreg[NVL], qf_all[NTW][KS];
#pragma unroll
    for (int i = 0; i < NKL; ++i) {
        const int c = tid + i * 256, key = c / CHP, ch = c % CHP;
        kreg[i] = u32x4{0, 0, 0, 0};
        if (c < T * CHP && ch < CH) kreg[i] = *(const u32x4*)(rowp(1, key) + ch * 8);
    }
    // Q fragments (B operand: lane holds Q[q = lane&15][d = kk*32 + (lane>>4)*8 .. +7]).  A wave owns 3 query
    // tiles, processed QT at a time: QT = 3 shares every K / V^T fragment read between the tiles (fewest LDS reads,
    // > 200 VGPRs -> 2 blocks/CU); QT = 1 keeps one tile live (~100 VGPRs -> 3 blocks/CU, more loads in flight
    // while other blocks compute: measured 8 % faster at B = 256).
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int q = (wave * 3 + part * NTW + t) * 16 + fr;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int d = kk * 32 + fg * 8;
            qf_all[t][kk] = (d < HD) ? *(const u32x4*)(rowp(0, q) + d) : u32x4{0, 0, 0, 0};
        }
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
        const int c = tid + i * 256, key = c / CH, ch = c % CH;
        if (c < T * CH) vreg[i] = *(const u32x4*)(rowp(2, key) + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < NKL; ++i) {
        const int c = tid + i * 256, key = c / CHP, ch = c % CHP;
        const int slot = C::KSWZ ? (ch ^ ((key >> 1) & 7)) : ch;
        if (c < T * CHP) *(u32x4*)(Ks + key * C::KSTR + slot * 16) = kreg[i];
    }
    __syncthreads();
    auto store_v = [&]() {
#pragma unroll
        for (int i = 0; i < NVL; ++i) {
            const int c = tid + i * 256, key = c / CH, ch = c % CH;
            if (c >= T * CH) continue;
            const u32x4 v = vreg[i];
            if constexpr (C::PAIR) {   // d = 8 ch .. +3 -> sub-tile 2(ch/4), d + 4 .. +7 -> sub-tile 2(ch/4) + 1, column chunk ch%4
                char* dst = Vs + (2 * (ch >> 2)) * C::VSUB + key * 32 + (ch & 3) * 8;
                *(u32x2*)(dst) = u32x2{v[0], v[1]};
                *(u32x2*)(dst + C::VSUB) = u32x2{v[2], v[3]};
            } else {
                *(u32x4*)(Vs + (ch >> 1) * C::VSUB + key * 32 + (ch & 1) * 16) = v;
            }
        }
        __syncthreads();
    };

    // per-lane LDS bases: K fragment row fr (+16 kt), slot kk*4+fg; V transpose read: lane (4 j + m) of a 16-lane
    // group supplies the address of (key 4 g + j, column chunk m) and receives column fr of keys 4 g .. 4 g + 3
    const char* kfrag = Ks + fr * C::KSTR;
    const int kswz = C::KSWZ ? ((fr >> 1) & 7) : 0;    // (row >> 1) & 7 with row = 16 kt + fr
    const char* vfrag = Vs + (fg * 4 + (fr >> 2)) * 32 + (fr & 3) * 8;

#pragma unroll
    for (int t0 = 0; t0 < NTW; t0 += QT) {
        // ---- S^T[key][q] for 12 key tiles x QT query tiles ----
        f32x4 s[QT][12];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int kt = 0; kt < 12; ++kt) s[t][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 12; ++kt) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const u32x4 kf = *(const u32x4*)(kfrag + kt * 16 * C::KSTR + (((kk * 4 + fg) ^ kswz) << 4));
#pragma unroll
                for (int t = 0; t < QT; ++t) s[t][kt] = mfma16<Ty>(kf, qf_all[t0 + t][kk], s[t][kt]);
            }
        }

        // ---- softmax over keys (per query column), fp32; P re-packed as PV B-fragments.  P is in [0, 1] and O a convex
        //      combination of 16-bit V values: neither can overflow the 16-bit range, so no saturating conversion ----
        u32x4 pf[QT][6];
        float inv_l[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < 12; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float l = 0.f;
            const float mb = mx * scale_log2e;
#pragma unroll
            for (int kt = 0; kt < 12; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = softmax_p(s[t][kt][r], scale_log2e, mb);
                    s[t][kt][r] = p;
                    l += p;
                }
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            inv_l[t] = 1.0f / l;
#pragma unroll
            for (int kb = 0; kb < 6; ++kb) {
                pf[t][kb][0] = pack2_nosat<Ty>(s[t][2 * kb][0], s[t][2 * kb][1]);
                pf[t][kb][1] = pack2_nosat<Ty>(s[t][2 * kb][2], s[t][2 * kb][3]);
                pf[t][kb][2] = pack2_nosat<Ty>(s[t][2 * kb + 1][0], s[t][2 * kb + 1][1]);
                pf[t][kb][3] = pack2_nosat<Ty>(s[t][2 * kb + 1][2], s[t][2 * kb + 1][3]);
            }
        }

        // ---- O^T[d][q] = sum_key V^T[d][key] P^T[key][q], sub-tile (pair) at a time ----
        if (t0 == 0) store_v();
        constexpr int G = C::PAIR ? 2 : 1;
#pragma unroll
        for (int dp = 0; dp < DT; dp += G) {
            f32x4 o[G][QT];
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
                for (int t = 0; t < QT; ++t) o[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 6; ++kb) {
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const char* vp_ = vfrag + (dp + u) * C::VSUB + kb * 1024;
                    const u32x2 lo = lds_read_tr16(vp_);          // keys 32 kb + 4 g + 0..3
                    const u32x2 hi = lds_read_tr16(vp_ + 512);    // keys 32 kb + 16 + 4 g + 0..3
                    const u32x4 vf = u32x4{lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
                    for (int t = 0; t < QT; ++t) o[u][t] = mfma16<Ty>(vf, pf[t][kb], o[u][t]);
                }
            }
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const int q = (wave * 3 + part * NTW + t0 + t) * 16 + fr;
                uint16_t* dst = out + ((size_t)b * T + q) * D + h * HD;
                if constexpr (MX && C::PAIR) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = o[0][t][e] * inv_l[t]; v[4 + e] = o[1][t][e] * inv_l[t]; }
                    float amax = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
                    amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
                    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
                    const uint32_t E = mx_scale_byte(amax);
                    const float inv = mx_inv_scale(E);
                    const size_t m = (size_t)b * T + q;
                    const int col = h * HD + dp * 16;                       // first column of this 32-column block
                    *(u32x2*)((uint8_t*)out + mx_code_off(m, (size_t)col + fg * 8, (size_t)D)) =
                        u32x2{mx_pack4(v[0], v[1], v[2], v[3], inv), mx_pack4(v[4], v[5], v[6], v[7], inv)};
                    if (fg == 0) out_scales[mx_scale_off(m, (size_t)(col >> 5), (size_t)D)] = (uint8_t)E;
                } else if constexpr (C::PAIR) {   // accumulator rows of the pair = d 16 dp + 8 fg + {0..3} and + {4..7}
                    u32x4 w;
                    w[0] = pack2_nosat<Ty>(o[0][t][0] * inv_l[t], o[0][t][1] * inv_l[t]);
                    w[1] = pack2_nosat<Ty>(o[0][t][2] * inv_l[t], o[0][t][3] * inv_l[t]);
                    w[2] = pack2_nosat<Ty>(o[1][t][0] * inv_l[t], o[1][t][1] * inv_l[t]);
                    w[3] = pack2_nosat<Ty>(o[1][t][2] * inv_l[t], o[1][t][3] * inv_l[t]);
                    *(u32x4*)(dst + dp * 16 + fg * 8) = w;
                } else {
                    u32x2 w;
                    w[0] = pack2_nosat<Ty>(o[0][t][0] * inv_l[t], o[0][t][1] * inv_l[t]);
                    w[1] = pack2_nosat<Ty>(o[0][t][2] * inv_l[t], o[0][t][3] * inv_l[t]);
                    *(u32x2*)(dst + dp * 16 + fg * 4) = w;
                }
            }
        }
    }
}

template <class Ty, int HD>
static hipError_t launch(const uint16_t* qkv, uint16_t* out, int B, int D, int heads, hipStream_t s, int blocked) {
#ifdef VP_TOOLS   // measurement build: VP_ATTN_QT=3 runs the three-query-tiles-at-a-time variant (fewest LDS reads, 2 blocks per CU: measured slower)
    static const int qt = [] { const char* e = getenv("VP_ATTN_QT"); return e ? atoi(e) : 1; }();
    auto kern = qt == 1 ? attention_kernel<Ty, HD, 1> : attention_kernel<Ty, HD, 3>;
#else
    auto kern = attention_kernel<Ty, HD, 1>;
#endif
    const float scale = 1.0f / sqrtf((float)HD);   // head_dim ** -0.5, vit.py:156
    if (blocked && HD != 64) return hipErrorInvalidValue;
    // small batches: three workgroups per (crop, head), one query tile per wave (QS = 3) while that still leaves CUs idle otherwise -- VP_ATTN_QSPLIT = the largest
    // number of (crop, head) pairs that takes it (0: never; profiles/small_batch_r6.txt)
    const char* qs_env = getenv("VP_ATTN_QSPLIT");   // read per launch (a parity test flips it inside one process; the hipGraph of a chunk keeps what it captured)
    const int qsplit_max = qs_env ? atoi(qs_env) : VP_ATTN_QSPLIT_DEFAULT;
    if (B * heads <= qsplit_max) {
        hipLaunchKernelGGL((attention_kernel<Ty, HD, 1, false, 3>), dim3(B * heads * 3), dim3(256), AttnCfg<HD>::LDS, s, qkv, out, D, heads,
                           scale * 1.4426950408889634f, blocked, (uint8_t*)nullptr);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(B * heads), dim3(256), AttnCfg<HD>::LDS, s, qkv, out, D, heads,
                       scale * 1.4426950408889634f, blocked, (uint8_t*)nullptr);
    return hipGetLastError();
}

hipError_t attention_launch(int dtype, const uint16_t* qkv, uint16_t* out, int B, int D, int heads, hipStream_t s, int qkv_blocked, uint8_t* mx_scales) {
    const int hd = D / heads;
    if (hd * heads != D) return hipErrorInvalidValue;
    if (mx_scales) {   // fp8 mode: MXFP8 output (head dim 64, fp16 operands)
        if (hd != 64 || dtype != DT_F16 || D % 128) return hipErrorInvalidValue;
        hipLaunchKernelGGL((attention_kernel<F16, 64, 1, true>), dim3(B * heads), dim3(256), AttnCfg<64>::LDS, s, qkv, out, D, heads,
                           (1.0f / 8.0f) * 1.4426950408889634f, qkv_blocked, mx_scales);
        return hipGetLastError();
    }
#define VP_ATT(HD)                                                                           \
    if (hd == HD)                                                                            \
        return dtype == DT_F16 ? launch<F16, HD>(qkv, out, B, D, heads, s, qkv_blocked) : launch<BF16, HD>(qkv, out, B, D, heads, s, qkv_blocked);
    VP_ATT(32) VP_ATT(64) VP_ATT(80)
#undef VP_ATT
    return hipErrorInvalidValue;
}

}  // namespace vp
