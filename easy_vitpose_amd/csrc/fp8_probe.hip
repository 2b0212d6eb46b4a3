// fp8 (OCP e4m3) probe for BASELINE config 5 -- a parity tap, not a product path.
//
// DESIGN.md section 6 documents the fp8 encoder as tolerance-infeasible from a CPU emulation (tests/fp8_budget.py: e4m3
// operands with one fp32 scale per weight row / per token row).  This translation unit confirms that emulation on the
// hardware it stands for: the quantisation a producer epilogue would do (x / scale -> v_cvt_pk_fp8_f32) and one GEMM through
// the K = 128 fp8 matrix instruction (__builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4, the 5 PFLOP/s pipe), so that
// tests/test_gpu_ops.py can compare the codes bit for bit with torch.float8_e4m3fn and the products with the emulation's.
// One wave per 16 x 16 output tile, operands straight from global memory: correctness only, no performance claim.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "mx8.h"

namespace vp {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));
typedef int i32x4_ __attribute__((ext_vector_type(4)));

// codes[r][k] = e4m3(src[r][k] / scale[r])  (IEEE fp32 division, then the hardware's round-to-nearest-even conversion)
__global__ void fp8_quantize_rows(const float* __restrict__ src, const float* __restrict__ scale, uint8_t* __restrict__ codes, int rows, int K) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (size_t)rows * K) return;
    const int r = (int)(i / K);
    const float s = scale[r];
    const float4 v = *(const float4*)(src + i);
    int packed = 0;
    packed = __builtin_amdgcn_cvt_pk_fp8_f32(__fdiv_rn(v.x, s), __fdiv_rn(v.y, s), packed, false);
    packed = __builtin_amdgcn_cvt_pk_fp8_f32(__fdiv_rn(v.z, s), __fdiv_rn(v.w, s), packed, true);
    *(int*)(codes + i) = packed;
}

// out[m][n] = a_scale[m] w_scale[n] sum_k A8[m][k] W8[n][k]; lane l of the wave holds row (l & 15), k = 32 (l >> 4) + [0, 32) of
// each 128-wide k-block of both operands; C: column n = l & 15, rows 4 (l >> 4) + [0, 4) (the f16 map: C/D is shape-determined)
__global__ __launch_bounds__(64) void fp8_gemm_16x16x128(const uint8_t* __restrict__ A8, const uint8_t* __restrict__ W8, const float* __restrict__ a_scale,
                                                         const float* __restrict__ w_scale, float* __restrict__ out, int M, int N, int K) {
    const int lane = threadIdx.x, m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const uint8_t* pa = A8 + (size_t)(m0 + (lane & 15)) * K + (lane >> 4) * 32;
    const uint8_t* pw = W8 + (size_t)(n0 + (lane & 15)) * K + (lane >> 4) * 32;
    f32x4_ acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 128) {
        const i32x8 a = *(const i32x8*)(pa + k);
        const i32x8 w = *(const i32x8*)(pw + k);
        // formats 0 / 0 = e4m3 x e4m3; both block scales constant 0 selects the plain (unscaled) form of the instruction
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, w, acc, 0, 0, 0, 0, 0, 0);
    }
    const int n = n0 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + (lane >> 4) * 4 + r;
        out[(size_t)m * N + n] = acc[r] * (a_scale[m] * w_scale[n]);
    }
}

// ---- MX probe (round 4): the block-scaled form of the same instruction, exactly as the fp8 mode's GEMM uses it ----
// rows of A [M, K] -> MXFP8: one thread per (row, 32-k block): amax -> E8M0 byte -> 32 codes; codes and scales in the layouts of mx8.h
__global__ void mx_quantize_rows(const float* __restrict__ src, uint8_t* __restrict__ codes, uint8_t* __restrict__ scales, int M, int K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kbs = K >> 5;
    if (i >= (size_t)M * kbs) return;
    const int m = (int)(i / kbs), kb = (int)(i % kbs);
    const float* p = src + (size_t)m * K + kb * 32;
    float v[32], amax = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) { v[e] = p[e]; amax = fmaxf(amax, fabsf(v[e])); }
    const uint32_t E = mx_scale_byte(amax);
    const float inv = mx_inv_scale(E);
    uint32_t* dst = (uint32_t*)(codes + mx_code_off(m, (size_t)kb * 32, K));
#pragma unroll
    for (int e = 0; e < 32; e += 4) dst[e >> 2] = mx_pack4(v[e], v[e + 1], v[e + 2], v[e + 3], inv);
    scales[mx_scale_off(m, kb, K)] = (uint8_t)E;
}

// out[m][n] = w_scale[n] sum_k dequant(A8)[m][k] W8[n][k]: one wave per (64 rows of m) x (16 columns n); operands SWAPPED into the
// instruction as in the GEMM kernels (weights = MFMA A operand, activations = B operand).  K LAYOUT OF THE INSTRUCTION (measured with
// tools/mx_probe_diag.py, round 4): of a lane's eight operand registers, registers 0-3 hold k = 16 g + [0, 16) and registers 4-7 hold
// k = 64 + 16 g + [0, 16) (g = lane >> 4) -- two 16-byte pieces 64 k apart, NOT 32 consecutive k --, and the scale byte lane group g
// supplies applies to k block g = [32 g, 32 g + 32), i.e. to registers 0-3 of lane groups 2 (g & 1), 2 (g & 1) + 1 ... : the block
// a scale belongs to is spread over two lane groups.  (The unscaled form is blind to this: any k permutation shared by both operands gives
// the same dot product.)  Row / column of a lane: l & 15, as for the 16-bit shapes; the activation scales of the four 16-row fragments j
// come in ONE dword per lane (mx8.h), selected by op_sel = j; C: lane l holds n = n0 + 4 (l >> 4) + e (e = 0..3) of m = m0 + 16 j + (l & 15)
__global__ __launch_bounds__(64) void mx_gemm_probe(const uint8_t* __restrict__ A8, const uint8_t* __restrict__ As, const uint8_t* __restrict__ W8,
                                                   const float* __restrict__ w_scale, float* __restrict__ out, int M, int N, int K) {
    const int lane = threadIdx.x, m0 = blockIdx.y * 64, n0 = blockIdx.x * 16;
    const int r = lane & 15, g = lane >> 4;
    f32x4_ acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4_{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 128) {
        auto frag = [&](const uint8_t* row) {     // the instruction's k layout: 16 bytes at k + 16 g, 16 bytes at k + 64 + 16 g
            const i32x4_ lo = *(const i32x4_*)(row + g * 16), hi = *(const i32x4_*)(row + 64 + g * 16);
            return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        const i32x8 w = frag(W8 + (size_t)(n0 + r) * K + k);
        const int sdw = *(const int*)(As + mx_scale_off(m0 + r, (k >> 5) + g, K));      // block g of this K-tile; bytes j = rows m0 + 16 j + r
        i32x8 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = frag(A8 + mx_code_off(m0 + j * 16 + r, k, K));
        acc[0] = mfma_mx<0>(w, x[0], acc[0], sdw);
        acc[1] = mfma_mx<1>(w, x[1], acc[1], sdw);
        acc[2] = mfma_mx<2>(w, x[2], acc[2], sdw);
        acc[3] = mfma_mx<3>(w, x[3], acc[3], sdw);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + g * 4 + e, m = m0 + j * 16 + r;
            out[(size_t)m * N + n] = acc[j][e] * w_scale[n];
        }
}

hipError_t mx_quantize_launch(const float* src, uint8_t* codes, uint8_t* scales, int M, int K, hipStream_t s) {
    if (M % 64 || K % 128) return hipErrorInvalidValue;
    const size_t na = (size_t)M * (K / 32);
    hipLaunchKernelGGL(mx_quantize_rows, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, src, codes, scales, M, K);
    return hipGetLastError();
}

hipError_t mx_probe_launch(const float* dA, const float* dW, const float* dWs, uint8_t* dA8, uint8_t* dAs, uint8_t* dW8, float* dOut,
                           int M, int N, int K, hipStream_t s) {
    const size_t na = (size_t)M * (K / 32), nw = (size_t)N * K / 4;
    hipLaunchKernelGGL(mx_quantize_rows, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, dA, dA8, dAs, M, K);
    hipLaunchKernelGGL(fp8_quantize_rows, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, dW, dWs, dW8, N, K);
    hipLaunchKernelGGL(mx_gemm_probe, dim3(N / 16, M / 64), dim3(64), 0, s, dA8, dAs, dW8, dWs, dOut, M, N, K);
    return hipGetLastError();
}

hipError_t fp8_probe_launch(const float* dA, const float* dW, const float* dAs, const float* dWs, uint8_t* dA8, uint8_t* dW8, float* dOut,
                            int M, int N, int K, hipStream_t s) {
    const size_t na = (size_t)M * K / 4, nw = (size_t)N * K / 4;
    hipLaunchKernelGGL(fp8_quantize_rows, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, dA, dAs, dA8, M, K);
    hipLaunchKernelGGL(fp8_quantize_rows, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, dW, dWs, dW8, N, K);
    hipLaunchKernelGGL(fp8_gemm_16x16x128, dim3(N / 16, M / 16), dim3(64), 0, s, dA8, dW8, dAs, dWs, dOut, M, N, K);
    return hipGetLastError();
}

}  // namespace vp
