// Fused scaled-dot-product attention for the fixed 192-token ViTPose sequence
// (vit.py:164-176): one workgroup per (crop, head); the 192x192 score tile never
// leaves the CU.
//
//  S^T = K Q^T   (MFMA A = K rows from LDS, B = Q rows straight from HBM)
//  softmax over keys in fp32 registers (scale folded into the exponent)
//  O^T = V^T P^T (MFMA A = V^T from LDS, B = P^T = the S^T accumulators re-packed)
//
// Computing the TRANSPOSED products makes every lane own one query column
// (q = lane & 15): row max / row sum need only two cross-lane steps (xor 16, 32),
// the S^T accumulator registers of two 16-key tiles are directly the 8-element
// B fragment of the PV MFMA (the MFMA k index is a permutation-invariant sum, so V^T
// is stored in LDS with its keys permuted to match: pos = 32-block | g*8 + hi*4 + j),
// and the output fragment is 4 consecutive head-dim values of one query = one 8-byte
// store.  4 waves x 3 query tiles; K / V^T fragments read from LDS are shared by the
// wave's 3 query tiles.
#include "common.h"
#include "kernels.h"

namespace vp {

static constexpr int T = 192;

template <int HD> struct AttnCfg {
    static constexpr int HDP = (HD + 31) / 32 * 32;   // head dim padded to the MFMA k step
    static constexpr int KSTR = HDP * 2 + 16;         // K row stride in bytes (padded)
    static constexpr int VSTR = T * 2 + 16;           // V^T row stride in bytes
    static constexpr int K_BYTES = T * KSTR;
    static constexpr int LDS = K_BYTES + HD * VSTR;
};

template <class Ty, int HD>
__global__ __launch_bounds__(256, 2) void attention_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                            int D, int heads, float scale_log2e) {
    using C = AttnCfg<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vt = smem + C::K_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const size_t ld = (size_t)3 * D;
    const uint16_t* qbase = qkv + (size_t)b * T * ld + (size_t)h * HD;
    const uint16_t* kbase = qbase + D;
    const uint16_t* vbase = qbase + 2 * D;

    // ---- stage K (row major, zero padded to HDP) and V^T (key-permuted) into LDS ----
    constexpr int CH = HD / 8;          // 16-B chunks per row
    constexpr int CHP = C::HDP / 8;
    for (int c = tid; c < T * CHP; c += 256) {
        const int key = c / CHP, ch = c % CHP;
        u32x4 v = u32x4{0, 0, 0, 0};
        if (ch < CH) v = *(const u32x4*)(kbase + (size_t)key * ld + ch * 8);
        *(u32x4*)(Ks + key * C::KSTR + ch * 16) = v;
    }
    if constexpr (HD == 64) {
        // Conflict-free transposed staging of V (measured: the naive per-element scatter below cost 21 % of
        // the kernel in 8-16-way LDS write conflicts).  One wave instruction covers 8 keys x 8 d-chunks; lane
        // (key j, chunk ch) holds V[key][ch*8 .. +7].  Bank of Vt[d][pos] = (4 d + pos/2) mod 32, so
        //  * the 8 keys are {k..k+3, k+16..k+19} of a 32-key block: their permuted positions give pos/2 =
        //    0,0,1,1,2,2,3,3, and
        //  * in write step e lane ch stores its element (e + ch) & 7, i.e. d = 8 ch + ((e + ch) & 7): 8 distinct
        //    values of 4 d mod 32.  The per-lane element rotation is done once per chunk on the 128-bit register
        //    (v_alignbit by 16 (ch & 1) bits, then two conditional dword rotations).
        const int ch = lane & 7, j = lane >> 3;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int g4 = wave * 6 + i;                       // (32-key block, group of 4 keys)
            const int key = (g4 >> 2) * 32 + (g4 & 3) * 4 + (j & 3) + 16 * (j >> 2);
            u32x4 v = *(const u32x4*)(vbase + (size_t)key * ld + ch * 8);
            const int sh = (ch & 1) * 16;                      // rotate right by ch halfwords
            u32x4 w;
            w[0] = __builtin_amdgcn_alignbit(v[1], v[0], sh);
            w[1] = __builtin_amdgcn_alignbit(v[2], v[1], sh);
            w[2] = __builtin_amdgcn_alignbit(v[3], v[2], sh);
            w[3] = __builtin_amdgcn_alignbit(v[0], v[3], sh);
            if (ch & 2) w = u32x4{w[1], w[2], w[3], w[0]};
            if (ch & 4) w = u32x4{w[2], w[3], w[0], w[1]};
            const int pos = (key & ~31) | (((key >> 2) & 3) << 3) | (((key >> 4) & 1) << 2) | (key & 3);
            char* dst = Vt + pos * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = ch * 8 + ((e + ch) & 7);
                const uint32_t word = w[e >> 1];
                *(uint16_t*)(dst + d * C::VSTR) = (uint16_t)((e & 1) ? (word >> 16) : (word & 0xffff));
            }
        }
    } else {
    for (int c = tid; c < T * CH; c += 256) {
        const int key = c / CH, ch = c % CH;
        const u32x4 v = *(const u32x4*)(vbase + (size_t)key * ld + ch * 8);
        const int pos = (key & ~31) | (((key >> 2) & 3) << 3) | (((key >> 4) & 1) << 2) | (key & 3);
        char* dst = Vt + (ch * 8) * C::VSTR + pos * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            *(uint16_t*)(dst + (2 * e) * C::VSTR) = (uint16_t)(v[e] & 0xffff);
            *(uint16_t*)(dst + (2 * e + 1) * C::VSTR) = (uint16_t)(v[e] >> 16);
        }
    }
    }

    // ---- Q fragments (B operand: lane holds Q[q = lane&15][d = kk*32 + (lane>>4)*8 .. +7]) ----
    const int fr = lane & 15, fg = lane >> 4;
    constexpr int KS = C::HDP / 32;     // k steps of QK^T
    u32x4 qf[3][KS];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int q = (wave * 3 + t) * 16 + fr;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int d = kk * 32 + fg * 8;
            qf[t][kk] = (d < HD) ? *(const u32x4*)(qbase + (size_t)q * ld + d) : u32x4{0, 0, 0, 0};
        }
    }
    __syncthreads();

    // ---- S^T[key][q] for 12 key tiles x 3 query tiles ----
    f32x4 s[3][12];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int kt = 0; kt < 12; ++kt) s[t][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 12; ++kt) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const u32x4 kf = *(const u32x4*)(Ks + (kt * 16 + fr) * C::KSTR + kk * 64 + fg * 16);
#pragma unroll
            for (int t = 0; t < 3; ++t) s[t][kt] = mfma16<Ty>(kf, qf[t][kk], s[t][kt]);
        }
    }

    // ---- softmax over keys (per query column), fp32; P re-packed as PV B-fragments ----
    u32x4 pf[3][6];
    float inv_l[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
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
                const float p = __builtin_amdgcn_exp2f(s[t][kt][r] * scale_log2e - mb);
                s[t][kt][r] = p;
                l += p;
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        inv_l[t] = 1.0f / l;
#pragma unroll
        for (int kb = 0; kb < 6; ++kb) {
            pf[t][kb][0] = pack2<Ty>(s[t][2 * kb][0], s[t][2 * kb][1]);
            pf[t][kb][1] = pack2<Ty>(s[t][2 * kb][2], s[t][2 * kb][3]);
            pf[t][kb][2] = pack2<Ty>(s[t][2 * kb + 1][0], s[t][2 * kb + 1][1]);
            pf[t][kb][3] = pack2<Ty>(s[t][2 * kb + 1][2], s[t][2 * kb + 1][3]);
        }
    }

    // ---- O^T[d][q] = sum_key V^T[d][key] P^T[key][q] ----
    constexpr int DT = HD / 16;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        f32x4 o[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 6; ++kb) {
            const u32x4 vf = *(const u32x4*)(Vt + (dt * 16 + fr) * C::VSTR + kb * 64 + fg * 16);
#pragma unroll
            for (int t = 0; t < 3; ++t) o[t] = mfma16<Ty>(vf, pf[t][kb], o[t]);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int q = (wave * 3 + t) * 16 + fr;
            u32x2 w;
            w[0] = pack2<Ty>(o[t][0] * inv_l[t], o[t][1] * inv_l[t]);
            w[1] = pack2<Ty>(o[t][2] * inv_l[t], o[t][3] * inv_l[t]);
            *(u32x2*)(out + ((size_t)b * T + q) * D + h * HD + dt * 16 + fg * 4) = w;
        }
    }
}

template <class Ty, int HD>
static hipError_t launch(const uint16_t* qkv, uint16_t* out, int B, int D, int heads, hipStream_t s) {
    auto kern = attention_kernel<Ty, HD>;
    const float scale = 1.0f / sqrtf((float)HD);   // head_dim ** -0.5, vit.py:156
    hipLaunchKernelGGL(kern, dim3(B * heads), dim3(256), AttnCfg<HD>::LDS, s, qkv, out, D, heads,
                       scale * 1.4426950408889634f);
    return hipGetLastError();
}

hipError_t attention_launch(int dtype, const uint16_t* qkv, uint16_t* out, int B, int D, int heads, hipStream_t s) {
    const int hd = D / heads;
    if (hd * heads != D) return hipErrorInvalidValue;
#define VP_ATT(HD)                                                                           \
    if (hd == HD)                                                                            \
        return dtype == DT_F16 ? launch<F16, HD>(qkv, out, B, D, heads, s) : launch<BF16, HD>(qkv, out, B, D, heads, s);
    VP_ATT(32) VP_ATT(64) VP_ATT(80)
#undef VP_ATT
    return hipErrorInvalidValue;
}

}  // namespace vp
