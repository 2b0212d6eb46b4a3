// fp8 mode (vp_config.dtype = VP_DTYPE_FP8): the one elementwise pass the mode adds.  Between a residual GEMM (patch embed / attn.proj /
// mlp.fc2: two-plane residual stream + partial LayerNorm statistics per 64-column granule) and the qkv / fc1 GEMM that consumes
// LayerNorm(x): merge the row statistics (common.h::ln_merge, the code ln_finalize_kernel runs), normalise the hi plane and write it as
// MXFP8 -- e4m3 codes in 64 x 128 blocks + one E8M0 scale per 32 columns (csrc/mx8.h).  Replaces ln_finalize; gamma and beta are
// folded into the consumer's weights and bias at upload, so the consumer's epilogue is acc * w_scale + bias.
// HBM-bound: 2 D bytes in (hi plane) + 1.03 D bytes out per row.
#include "common.h"
#include "kernels.h"
#include "mx8.h"

namespace vp {

template <class T>
__global__ __launch_bounds__(256) void ln_quant_kernel(const uint16_t* __restrict__ x_hi, const float* __restrict__ ln_part, int tiles, float inv_d,
                                                       uint8_t* __restrict__ codes, uint8_t* __restrict__ scales, int M, int D) {
    __shared__ float2 st[64];
    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * 64;                    // this workgroup's 64-row group
    if (tid < 64) {
        float mean = 0.f, rstd = 0.f;
        if (g0 + tid < M) ln_merge(ln_part + (size_t)(g0 + tid) * tiles * 2, tiles, inv_d, mean, rstd);
        st[tid] = float2{mean, rstd};
    }
    __syncthreads();
    const int kbs = D >> 5;
    // work item = (r, kb): the four rows r, r + 16, r + 32, r + 48 of one 32-column block -> 4 x 32 codes and ONE scale dword;
    // consecutive threads take consecutive kb: a wave reads whole rows in 64-byte pieces
    for (int w = tid; w < 16 * kbs; w += 256) {
        const int r = w / kbs, kb = w - r * kbs;
        uint32_t sdw = 0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int lr = jj * 16 + r, m = g0 + lr;
            u32x4 o0 = {0, 0, 0, 0}, o1 = {0, 0, 0, 0};
            if (m < M) {
                const u32x4* src = (const u32x4*)(x_hi + (size_t)m * D + kb * 32);
                const float2 ms = st[lr];
                const float sh = -ms.x * ms.y;
                float v[32], amax = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 t = src[q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = __builtin_fmaf(from_bits<T>((uint16_t)(t[e] & 0xffffu)), ms.y, sh);
                        const float b = __builtin_fmaf(from_bits<T>((uint16_t)(t[e] >> 16)), ms.y, sh);
                        v[q * 8 + 2 * e] = a;
                        v[q * 8 + 2 * e + 1] = b;
                        amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b)));
                    }
                }
                const uint32_t E = mx_scale_byte(amax);
                const float inv = mx_inv_scale(E);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o0[q] = mx_pack4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3], inv);
                    o1[q] = mx_pack4(v[16 + 4 * q], v[17 + 4 * q], v[18 + 4 * q], v[19 + 4 * q], inv);
                }
                sdw |= E << (8 * jj);
            }
            u32x4* dst = (u32x4*)(codes + mx_code_off((size_t)m, (size_t)kb * 32, (size_t)D));
            dst[0] = o0;
            dst[1] = o1;
        }
        *(uint32_t*)(scales + ((((size_t)(g0 >> 6)) * kbs + kb) << 6) + r * 4) = sdw;
    }
}

hipError_t ln_quant_launch(int dtype, const uint16_t* x_hi, const float* ln_part, int tiles, uint8_t* codes, uint8_t* scales, int M, int Mp, int D,
                           hipStream_t s) {
    if (Mp % 64 || D % 128 || M > Mp) return hipErrorInvalidValue;
    if (dtype == DT_F16) hipLaunchKernelGGL(ln_quant_kernel<F16>, dim3(Mp / 64), dim3(256), 0, s, x_hi, ln_part, tiles, 1.0f / (float)D, codes, scales, M, D);
    else hipLaunchKernelGGL(ln_quant_kernel<BF16>, dim3(Mp / 64), dim3(256), 0, s, x_hi, ln_part, tiles, 1.0f / (float)D, codes, scales, M, D);
    return hipGetLastError();
}

}  // namespace vp
