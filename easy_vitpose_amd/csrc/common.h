// Shared device helpers for the gfx950 (CDNA4) ViTPose kernels.
// Wave = 64 lanes; MFMA 16x16x32 f16/bf16 with fp32 accumulation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vp {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

struct F16 {};   // operand type tags
struct BF16 {};

// ---- scalar conversions (fp32 <-> 16-bit storage, round to nearest even) ----
template <class T> __device__ __forceinline__ uint16_t to_bits(float v);
template <> __device__ __forceinline__ uint16_t to_bits<F16>(float v) {
    // saturate instead of overflowing to inf: activations are bounded in practice,
    // but a clamp is cheaper than a NaN three layers later
    v = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
    _Float16 h = (_Float16)v;
    return __builtin_bit_cast(uint16_t, h);
}
template <> __device__ __forceinline__ uint16_t to_bits<BF16>(float v) {
    return __builtin_bit_cast(uint16_t, (__bf16)v);   // v_cvt_pk_bf16_f32 on gfx950: round to nearest even, NaN kept quiet
}
template <class T> __device__ __forceinline__ float from_bits(uint16_t b);
template <> __device__ __forceinline__ float from_bits<F16>(uint16_t b) {
    return (float)__builtin_bit_cast(_Float16, b);
}
template <> __device__ __forceinline__ float from_bits<BF16>(uint16_t b) {
    return __builtin_bit_cast(float, (uint32_t)b << 16);
}
template <class T> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    return (uint32_t)to_bits<T>(lo) | ((uint32_t)to_bits<T>(hi) << 16);
}
template <> __device__ __forceinline__ uint32_t pack2<F16>(float lo, float hi) {    // 2 v_med3_f32 (saturate) + one v_cvt_pk_f16_f32
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 v = {(_Float16)__builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f), (_Float16)__builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f)};
    return __builtin_bit_cast(uint32_t, v);
}
template <> __device__ __forceinline__ uint32_t pack2<BF16>(float lo, float hi) {   // one v_cvt_pk_bf16_f32
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    const b2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// as pack2 for values known to be far inside the 16-bit range (no saturation clamp)
template <class T> __device__ __forceinline__ uint32_t pack2_nosat(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2_nosat<F16>(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
template <> __device__ __forceinline__ uint32_t pack2_nosat<BF16>(float lo, float hi) { return pack2<BF16>(lo, hi); }

// ---- MFMA: D(16x16) += A(16x32) * B(32x16) ----
// fragment layout (gfx950): A lane l holds A[row = l&15][k = (l>>4)*8 .. +7],
//                           B lane l holds B[k = (l>>4)*8 .. +7][col = l&15],
//                           C lane l holds C[row = (l>>4)*4 + r][col = l&15], r = 0..3
template <class T> __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mfma16<F16>(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma16<BF16>(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Two-plane residual stream: x = hi + lo with hi = round16(x), lo = round16(x - hi).  fp16: x is clamped to the fp16 range FIRST
// (+-65504), so hi cannot saturate away from x and lo stays a tiny correction (an unclamped |x| > 65504 would leave
// lo = x - 65504, which the unsaturated lo conversion turns into inf).  Same instruction count as clamping inside the hi conversion.
template <class T> __device__ __forceinline__ void split_planes2(float& a, float& b, uint32_t& hi, uint32_t& lo);
template <> __device__ __forceinline__ void split_planes2<F16>(float& a, float& b, uint32_t& hi, uint32_t& lo) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);
    b = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
    const h2 h = {(_Float16)a, (_Float16)b};
    hi = __builtin_bit_cast(uint32_t, h);
    const h2 l = {(_Float16)(a - (float)h[0]), (_Float16)(b - (float)h[1])};
    lo = __builtin_bit_cast(uint32_t, l);
}
template <> __device__ __forceinline__ void split_planes2<BF16>(float& a, float& b, uint32_t& hi, uint32_t& lo) {
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    const b2 h = {(__bf16)a, (__bf16)b};
    hi = __builtin_bit_cast(uint32_t, h);
    const b2 l = {(__bf16)(a - (float)h[0]), (__bf16)(b - (float)h[1])};
    lo = __builtin_bit_cast(uint32_t, l);
}

// LayerNorm folded into a GEMM epilogue: rstd * (acc - mean * s) + b as two explicit fused multiply-adds,
// so that every kernel that applies it rounds identically (bit-identity between tile configurations is the race screen)
__device__ __forceinline__ float ln_fold(float acc, float mean, float s, float rstd, float b) {
    return __builtin_fmaf(__builtin_fmaf(-mean, s, acc), rstd, b);
}

// Fused-LayerNorm statistics: fold the per-granule (sum, M2 about the granule mean) partials of one row, in granule order, into
// (mean, rstd) with the pairwise-merge identity  M2 = sum_g [M2_g + 64 (mean_g - mean)^2].  ONE definition for ln_finalize_kernel
// and for the GEMM epilogues that fold it in at small batch: identical operation sequence, hence identical bits.
__device__ __forceinline__ void ln_merge(const float* __restrict__ p, int tiles, float inv_d, float& mean, float& rstd) {
    float s1 = 0.f;
    for (int t = 0; t < tiles; ++t) s1 += p[2 * t];
    mean = s1 * inv_d;
    float m2 = 0.f;
    for (int t = 0; t < tiles; ++t) {
        const float d = __builtin_fmaf(p[2 * t], 1.0f / 64.0f, -mean);
        m2 += __builtin_fmaf(64.0f * d, d, p[2 * t + 1]);
    }
    rstd = rsqrtf(__builtin_fmaf(m2, inv_d, 1e-6f));
}

// The same merge on a REGISTER copy of the row's TILES granule statistics, fetched up front as TILES / 2 loads of 16 bytes (one memory latency instead of 2 x TILES dependent
// 4-byte loads): what ln_finalize_kernel_t and the consumer GEMMs' prologue (gemm.hip, GemmArgs::ln_part) run.  ln_merge on the copy: identical operations, identical bits.
template <int TILES>
__device__ __forceinline__ void ln_merge_row(const float* __restrict__ p, float inv_d, float& mean, float& rstd) {
    static_assert(TILES % 2 == 0, "a row of partials is a whole number of 16-byte pieces");
    const f32x4* src = (const f32x4*)p;
    float v[2 * TILES];
#pragma unroll
    for (int i = 0; i < TILES / 2; ++i) {
        const f32x4 q = src[i];
        v[4 * i] = q[0]; v[4 * i + 1] = q[1]; v[4 * i + 2] = q[2]; v[4 * i + 3] = q[3];
    }
    ln_merge(v, TILES, inv_d, mean, rstd);
}

// softmax numerator 2^(s scale - mb) with the multiply-add as ONE explicit fma: under -ffp-contract=fast hipcc fuses `s * scale - mb` for most
// elements and emits v_pk_mul + v_sub for a few, and WHICH ones depends on the surrounding code -- attention.hip and qkvattn.hip then differed in
// the last bit of ~1 % of the probabilities (round 4).  One definition, one rounding, for every kernel that must agree bit for bit.
__device__ __forceinline__ float softmax_p(float s, float scale_log2e, float mb) {
    return __builtin_amdgcn_exp2f(__builtin_fmaf(s, scale_log2e, -mb));
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

// async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)gsrc, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware, bijective block remap: the dispatcher places block b on XCD b % 8; give
// every XCD one contiguous range of logical tiles so neighbouring tiles (which share
// operand panels) hit the same private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// GELU (exact erf form, vit.py:127 nn.GELU) = max(x, 0) - |x| / 2 * erfc(|x| / sqrt 2), with erfc(t / sqrt 2) = 2^(-t P(t)), P a degree-4
// minimax fit of -log2(erfc(t / sqrt 2)) / t: one transcendental + 9 VALU instructions, |error| <= 1.2e-6 in fp32 (far below the 16-bit
// rounding of the result).  The factor 1/2 rides in the exponent (2^(-t P - 1)): one multiplication less than the round-2 form.
// ONE definition for every GEMM kernel: the fc1 configurations must agree bit for bit.
// Round 5: gelu_sat<T> = the same value already SATURATED to T's range, so that the 16-bit conversion behind it needs no clamp of its own (pack2_nosat):
// GELU(x) >= -0.17, and for x > 65504 the erfc term has underflowed to 0, so min(GELU(x), 65504) == GELU with max(x, 0) replaced by
// med3(x, 0, 65504) -- ONE instruction for the max and the clamp (the fc1 epilogue is VALU-issue-bound: 11.5 -> 10.5 instructions per value;
// a degree-3 fit was rejected: |error| 5e-5, the size of the 16-bit rounding of the result).  Bit-identical to pack2<T>(gelu_erf(x)) for every finite x.
__device__ __forceinline__ float gelu_core(float x, float relu) {
    const float a = fabsf(x);
    float q = fmaf(a, 5.204574411e-04f, -7.397505390e-03f);
    q = fmaf(q, a, 5.256122897e-02f);
    q = fmaf(q, a, 4.592546873e-01f);
    q = fmaf(q, a, 1.151091354e+00f);
    const float h = __builtin_amdgcn_exp2f(fmaf(-q, a, -1.0f));
    float r = fmaf(-a, h, relu);
    // the fp32 result is pinned here: with a saturation-free conversion directly behind it hipcc fused this multiply-add and the conversion into
    // v_fma_mixlo/hi_f16 -- ONE rounding instead of two -- in SOME instantiations (gemm8's bias variant, not its LayerNorm variant): 17 of 37 M
    // fc1 outputs then differed between tile configurations, whose bit identity is the race screen (tests/test_gpu_gemm_cfgs.py)
    asm("" : "+v"(r));
    return r;
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_core(x, fmaxf(x, 0.f)); }
template <class T> __device__ __forceinline__ float gelu_sat(float x);
template <> __device__ __forceinline__ float gelu_sat<F16>(float x) { return gelu_core(x, __builtin_amdgcn_fmed3f(x, 0.f, 65504.f)); }
template <> __device__ __forceinline__ float gelu_sat<BF16>(float x) { return gelu_erf(x); }   // bf16 conversions never clamp

}  // namespace vp
