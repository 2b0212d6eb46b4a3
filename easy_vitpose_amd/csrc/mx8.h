// MXFP8 (OCP e4m3 elements, one E8M0 power-of-two scale per block of 32 consecutive k) -- the operand format of the opt-in fp8 mode
// (vp_config.dtype = VP_DTYPE_FP8, BASELINE configs[4]).  gfx950's v_mfma_scale_f32_16x16x128_f8f6f4 takes the block scales as an
// operand, one byte per lane: lane group g = lane >> 4 supplies the scale of k block g (k = 32 g .. 32 g + 31 of the instruction's 128)
// for row lane & 15.  (The ELEMENTS of a lane are two 16-byte pieces, k = 16 g + [0, 16) in operand registers 0-3 and k = 64 + 16 g +
// [0, 16) in registers 4-7 -- measured, tools/mx_probe_diag.py -- which is exactly what the 16-bit kernels' two k-half fragment reads of a
// 128-byte LDS row fetch.)  De-quantisation therefore costs nothing in the K-loop, and a PRODUCER can quantise its output tile locally (a
// block of 32 output columns lives in 2-4 lanes of one wave), with no row-wide statistics.
//
// Layouts in HBM (activations [M, K], M a multiple of 64, K a multiple of 128):
//   codes   [M/64][K/128][64][128] bytes: a (64-row, 128-k) block is 8 KiB contiguous -- one LDS-DMA slot half of the 8-phase GEMM;
//   scales  [M/64][K/32][16][4]    bytes: the dword at (row group, k block kb, r) holds the E8M0 bytes of rows r, r+16, r+32, r+48, i.e. the
//           four 16-row MFMA fragments a wave multiplies against one weight fragment: one dword load per lane, byte j by `opsel`.
// Weights [N, K]: e4m3 codes row-major + one fp32 scale per output channel, applied in the epilogue (block scale = 1.0 = byte 127).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vp {

typedef int i32x8 __attribute__((ext_vector_type(8)));

// E8M0 scale byte of a block whose largest magnitude is `amax`: 2^(E - 127) with amax / 2^(E - 127) in [128, 256) -- one binade below
// e4m3's largest binade [256, 448], so no element can overflow the conversion whatever the rounding does; the element keeps its
// 3 mantissa bits down to 2^-6 of that (and 2^-9 as a subnormal).  amax = 0 (or subnormal-small): byte 0, every code 0.
__device__ __forceinline__ uint32_t mx_scale_byte(float amax) {
    const uint32_t ex = (__builtin_bit_cast(uint32_t, amax) >> 23) & 0xffu;
    return ex > 7u ? ex - 7u : 0u;
}
// 1 / 2^(E - 127) as a float (E <= 248: finite)
__device__ __forceinline__ float mx_inv_scale(uint32_t e) { return __builtin_bit_cast(float, (254u - e) << 23); }

// four floats -> four e4m3 codes in one dword (round to nearest even: v_cvt_pk_fp8_f32), element 0 in the low byte
__device__ __forceinline__ uint32_t mx_pack4(float a, float b, float c, float d, float inv) {
    int p = 0;
    p = __builtin_amdgcn_cvt_pk_fp8_f32(a * inv, b * inv, p, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c * inv, d * inv, p, true);
    return (uint32_t)p;
}

// byte offset of element (m, k) in the blocked code layout, and of the scale DWORD of (row group of m, k block kb = k / 32, r = m & 15)
__host__ __device__ __forceinline__ size_t mx_code_off(size_t m, size_t k, size_t K) {
    return (((m >> 6) * (K >> 7) + (k >> 7)) << 13) + ((m & 63) << 7) + (k & 127);
}
__host__ __device__ __forceinline__ size_t mx_scale_off(size_t m, size_t kb, size_t K) {
    return (((m >> 6) * (K >> 5) + kb) << 6) + ((m & 15) << 2) + ((m >> 4) & 3);   // byte (m >> 4) & 3 of the dword at ... + (m & 15) * 4
}

// D(16x16) += A(16x128) * B(128x16), e4m3 x e4m3, block scales: A = weights (scale 1.0), B = activations (byte `J` of `sb`, per lane)
template <int J>
__device__ __forceinline__ __attribute__((ext_vector_type(4))) float mfma_mx(i32x8 a, i32x8 b, __attribute__((ext_vector_type(4))) float c, int sb) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, J, sb);
}

}  // namespace vp

// ---- host side: fp32 -> OCP e4m3 (e4m3fn: no infinities, 0x7f / 0xff = NaN, max 448), round to nearest even, saturating ----
static inline uint8_t vp_host_e4m3(float v) {
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint8_t)(sign | 0x7f);                 // NaN
    float f;
    __builtin_memcpy(&f, &a, 4);
    if (f >= 464.0f) return (uint8_t)(sign | 0x7e);                     // >= halfway between 448 and the (absent) 480: saturate to 448
    if (f < 0.0009765625f) return sign;                                 // < 2^-10 = half the smallest subnormal: 0  (== 2^-10 ties to even = 0)
    int e = (int)(a >> 23) - 127;                                       // unbiased exponent of f
    if (e < -6) {                                                       // subnormal e4m3: multiples of 2^-9
        const float q = f * 512.0f;                                     // in [0.5, 8)
        int n = (int)q;
        const float r = q - (float)n;
        if (r > 0.5f || (r == 0.5f && (n & 1))) ++n;
        return (uint8_t)(sign | (uint8_t)n);                            // n == 8 is the smallest normal (0x08): same bits
    }
    uint32_t m = (a >> 20) & 0x7u;                                      // top 3 mantissa bits
    const uint32_t rem = a & 0xfffffu, half = 0x80000u;
    uint32_t code = ((uint32_t)(e + 7) << 3) | m;
    if (rem > half || (rem == half && (code & 1u))) ++code;             // carries into the exponent correctly
    if (code > 0x7eu) code = 0x7eu;
    return (uint8_t)(sign | code);
}
static inline float vp_host_e4m3_to_float(uint8_t c) {
    const int e = (c >> 3) & 0xf, m = c & 7;
    float v;
    if (e == 0) v = (float)m * 0.001953125f;                            // m * 2^-9
    else {
        uint32_t u = ((uint32_t)(e - 7 + 127) << 23) | ((uint32_t)m << 20);
        __builtin_memcpy(&v, &u, 4);
    }
    return (c & 0x80) ? -v : v;
}
