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

namespace vp {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));

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

hipError_t fp8_probe_launch(const float* dA, const float* dW, const float* dAs, const float* dWs, uint8_t* dA8, uint8_t* dW8, float* dOut,
                            int M, int N, int K, hipStream_t s) {
    const size_t na = (size_t)M * K / 4, nw = (size_t)N * K / 4;
    hipLaunchKernelGGL(fp8_quantize_rows, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, dA, dAs, dA8, M, K);
    hipLaunchKernelGGL(fp8_quantize_rows, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, dW, dWs, dW8, N, K);
    hipLaunchKernelGGL(fp8_gemm_16x16x128, dim3(N / 16, M / 16), dim3(64), 0, s, dA8, dW8, dAs, dWs, dOut, M, N, K);
    return hipGetLastError();
}

}  // namespace vp
