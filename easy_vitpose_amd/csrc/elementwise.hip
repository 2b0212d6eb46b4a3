// HBM-bound helpers of the ViTPose path: patch gather (im2col) and LayerNorm.
#include "common.h"
#include "kernels.h"
#include "../../include/vitpose_hip.h"

namespace vp {

// ------------------------------------------------------------------ im2col
// PatchEmbed = Conv2d(3, D, k=16, s=16, padding=2) (vit.py:222): patch (py,px) covers
// rows 16py-2 .. 16py+13, cols 16px-2 .. 16px+13 of the 256x192 crop, zero outside.
// Row m = b*192 + py*12 + px of the patch matrix, column k = c*256 + ky*16 + kx
// (the flattening of the conv weight [D,3,16,16]).
// One block = one patch row (b, py): its input is 3 x 16 whole image rows (read as contiguous
// 16-byte pieces), its output 12 consecutive rows of the patch matrix (18 KiB contiguous); the
// transposition in between goes through LDS, so that both HBM sides are fully coalesced (the
// thread-per-output-chunk version fetched 2.1x the algorithmic bytes: 64-byte pieces of 128-byte lines).
// FLIP: the crop is mirrored left-right on the fly (flip-test, topdown_heatmap_simple_head.py:195-218).
template <class Ty, int FMT, bool FLIP>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ in, uint16_t* __restrict__ out, int B, int n_src) {
    constexpr int XS = 208;                                  // LDS row: x = -2 .. 205 (index x + 2)
    __shared__ __attribute__((aligned(16))) uint16_t tile[3 * 16 * XS];
    const int tid = threadIdx.x;
    const int bo = blockIdx.x >> 4, py = blockIdx.x & 15;    // output crop bo reads source crop min(bo, n_src - 1): the rows of a padded encoder batch (vitpose_api.hip forward_chunk) repeat the last crop
    const int b = min(bo, n_src - 1);
    const int ytop = 16 * py - 2;
    if (FMT == VP_INPUT_F32_NCHW) {
        // 3 channels x 16 rows x 48 float4
        for (int id = tid; id < 3 * 16 * 48; id += 256) {
            const int x4 = id % 48, ky = (id / 48) & 15, c = id / (48 * 16);
            const int y = ytop + ky;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((unsigned)y < 256u) v = *(const f32x4*)((const float*)in + (((size_t)b * 3 + c) * 256 + y) * 192 + x4 * 4);
            if (FLIP) {   // pixel x of the source row lands at x' = 191 - x
                uint32_t* dst = (uint32_t*)(tile + (c * 16 + ky) * XS + (188 - x4 * 4) + 2);
                dst[0] = pack2<Ty>(v[3], v[2]);
                dst[1] = pack2<Ty>(v[1], v[0]);
            } else {
                uint32_t* dst = (uint32_t*)(tile + (c * 16 + ky) * XS + x4 * 4 + 2);
                dst[0] = pack2<Ty>(v[0], v[1]);
                dst[1] = pack2<Ty>(v[2], v[3]);
            }
        }
    } else {
        // 16 rows x 576 bytes (192 px x RGB) as 36 16-byte pieces per row
        for (int id = tid; id < 16 * 36; id += 256) {
            const int q = id % 36, ky = id / 36;
            const int y = ytop + ky;
            u32x4 raw = u32x4{0, 0, 0, 0};
            const bool inside = (unsigned)y < 256u;
            if (inside) raw = *(const u32x4*)((const uint8_t*)in + ((size_t)b * 256 + y) * 576 + q * 16);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int off = q * 16 + e, x = off / 3, c = off - x * 3;
                // pre_img (easy_ViTPose/inference.py:316-317): float64 x/255, (x-MEAN)/STD, cast to fp32
                const double mean = c == 0 ? 0.485 : (c == 1 ? 0.456 : 0.406);
                const double stdv = c == 0 ? 0.229 : (c == 1 ? 0.224 : 0.225);
                const uint8_t u = (uint8_t)(raw[e >> 2] >> ((e & 3) * 8));
                const float f = inside ? (float)(((double)u / 255.0 - mean) / stdv) : 0.f;
                tile[(c * 16 + ky) * XS + (FLIP ? 191 - x : x) + 2] = to_bits<Ty>(f);
            }
        }
    }
    if (tid < 96) {   // left zero border (x = -2, -1) of every (c, ky) row
        tile[(tid >> 1) * XS + (tid & 1)] = 0;
    }
    __syncthreads();
    uint16_t* orow = out + ((size_t)bo * 192 + py * 12) * 768;
    for (int id = tid; id < 12 * 96; id += 256) {
        const int px = id / 96, kc = id - px * 96;
        const int c = kc >> 5, ky = (kc & 31) >> 1, kx0 = (kc & 1) * 8;
        *(u32x4*)(orow + (size_t)id * 8) = *(const u32x4*)(tile + (c * 16 + ky) * XS + 16 * px + kx0);
    }
}

hipError_t im2col_launch(int dtype, const void* crops, int fmt, uint16_t* out, int B, hipStream_t s, bool flip, int n_src) {
    if (B <= 0) return hipSuccess;
    if (n_src <= 0 || n_src > B) n_src = B;   // B output crops from n_src source crops (the last one repeated)
    const int grid = B * 16;   // one block per patch row
#define VP_I2C(TY, F)                                                                                              \
    do {                                                                                                           \
        if (flip) hipLaunchKernelGGL((im2col_kernel<TY, F, true>), dim3(grid), dim3(256), 0, s, crops, out, B, n_src);    \
        else hipLaunchKernelGGL((im2col_kernel<TY, F, false>), dim3(grid), dim3(256), 0, s, crops, out, B, n_src);        \
    } while (0)
    if (fmt == VP_INPUT_F32_NCHW) {
        if (dtype == DT_F16) VP_I2C(F16, VP_INPUT_F32_NCHW); else VP_I2C(BF16, VP_INPUT_F32_NCHW);
    } else if (fmt == VP_INPUT_U8_NHWC) {
        if (dtype == DT_F16) VP_I2C(F16, VP_INPUT_U8_NHWC); else VP_I2C(BF16, VP_INPUT_U8_NHWC);
    } else {
        return hipErrorInvalidValue;
    }
#undef VP_I2C
    return hipGetLastError();
}

// ------------------------------------------------------------- flip-test merge
// hm[n][k][y][x] = 0.5 (hm[n][k][y][x] + back[n][k][y][x]),  back = flip_back(hm_flipped) (post_transforms.py:110-147:
// channels swapped by the mirror pairs, then reversed in x), optionally shifted right by one pixel
// (topdown_heatmap_simple_head.py:213-215, `shift_heatmap`); the average is what the flip-test consumer takes.
__global__ __launch_bounds__(256) void flip_merge_kernel(float* __restrict__ hm, const float* __restrict__ hm_flipped,
                                                         const int32_t* __restrict__ partner, int K, int shift, size_t total) {
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
        const int x = (int)(id % 48);
        const size_t row = id / 48;                    // (n*K + k)*64 + y
        const int y = (int)(row & 63);
        const size_t nk = row >> 6;
        const int k = (int)(nk % K);
        const size_t n = nk / K;
        int xs = x;                                    // column of the flipped-back map before the shift
        if (shift && x > 0) xs = x - 1;
        const float b = hm_flipped[((n * K + partner[k]) * 64 + y) * 48 + (47 - xs)];
        hm[id] = 0.5f * (hm[id] + b);
    }
}

hipError_t flip_merge_launch(float* hm, const float* hm_flipped, const int32_t* partner, int N, int K, int shift, hipStream_t s) {
    const size_t total = (size_t)N * K * 3072;
    int grid = (int)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    if (total) hipLaunchKernelGGL(flip_merge_kernel, dim3(grid), dim3(256), 0, s, hm, hm_flipped, partner, K, shift, total);
    return hipGetLastError();
}

// --------------------------------------------------------------- LayerNorm
// nn.LayerNorm(eps=1e-6) (vit.py:274) over the fp32 residual stream, one wave per
// token row, row held in registers (D <= 1280 -> <= 5 float4 per lane), two-pass
// mean / variance in fp32, output rounded once to the GEMM operand type.
// PLANES: x is the two-plane 16-bit residual stream of the fused-LayerNorm path (x = hi + lo, gemm.hip).
template <class Ty, bool PLANES>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, uint16_t* __restrict__ out16,
                                                        float* __restrict__ out32, int M, int D, size_t plane) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;
    const f32x4* xr = (const f32x4*)(x + (size_t)row * D);
    f32x4 v[5];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int idx = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < nv) {
            if constexpr (PLANES) {
                const uint16_t* xh = (const uint16_t*)x + (size_t)row * D;
                const u32x2 h = ((const u32x2*)xh)[idx], l = ((const u32x2*)(xh + plane))[idx];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int sh = (e & 1) * 16;
                    v[i][e] = from_bits<Ty>((uint16_t)(h[e >> 1] >> sh)) + from_bits<Ty>((uint16_t)(l[e >> 1] >> sh));
                }
            } else {
                v[i] = xr[idx];
            }
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + 1e-6f);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const f32x4 g = ((const f32x4*)gamma)[idx];
            const f32x4 b = ((const f32x4*)beta)[idx];
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            if (out16) {
                u32x2 o;
                o[0] = pack2<Ty>(y[0], y[1]);
                o[1] = pack2<Ty>(y[2], y[3]);
                ((u32x2*)(out16 + (size_t)row * D))[idx] = o;
            }
            if (out32) ((f32x4*)(out32 + (size_t)row * D))[idx] = y;
        }
    }
}

hipError_t layernorm_launch(int dtype, const float* x, const float* gamma, const float* beta, uint16_t* out16,
                            float* out32, int M, int D, hipStream_t s, size_t plane) {
    if ((D & 3) || D > 1280) return hipErrorInvalidValue;
    const int grid = (M + 3) / 4;
#define VP_LN(TY, PL) hipLaunchKernelGGL((layernorm_kernel<TY, PL>), dim3(grid), dim3(256), 0, s, x, gamma, beta, out16, out32, M, D, plane)
    if (dtype == DT_F16) { if (plane) VP_LN(F16, true); else VP_LN(F16, false); }
    else { if (plane) VP_LN(BF16, true); else VP_LN(BF16, false); }
#undef VP_LN
    return hipGetLastError();
}


// sum over the 8 lanes of a DPP half-row (every lane receives it): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror -- the sequence of gemm.hip::row8_sum
__device__ __forceinline__ float row8_sum_dpp(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
    return x;
}

// Fused-LayerNorm helper: the residual GEMMs leave, per token row and 64-column granule, (sum, M2 about the
// granule mean).  Fold them in a FIXED order (deterministic, unlike atomics) into (mean, rstd) per row with the
// pairwise-merge identity  M2 = sum_g [M2_g + 64 (mean_g - mean)^2]  -- no E[x^2] - mean^2 cancellation.
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ partials, float* __restrict__ rowstat,
                                                          int M, int tiles, float inv_d) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float mean, rstd;
    ln_merge(partials + (size_t)m * tiles * 2, tiles, inv_d, mean, rstd);
    rowstat[2 * (size_t)m] = mean;
    rowstat[2 * (size_t)m + 1] = rstd;
}

// Round 5: the same merge with a row's partials fetched UP FRONT as 16-byte loads (one memory latency instead of 2 x tiles dependent 4-byte loads 8 bytes
// apart) and 64-thread workgroups (768 instead of 192 at 256 crops: every CU takes part).  ln_merge runs on the register copy: the same operations in the same
// order, hence the same bits (the kernel sits between every residual GEMM and its consumer: 24 launches per ViTPose-B forward, 64 per ViTPose-H).
template <int TILES>
__global__ __launch_bounds__(64) void ln_finalize_kernel_t(const float* __restrict__ partials, float* __restrict__ rowstat, int M, float inv_d) {
    static_assert(TILES % 2 == 0, "a row of partials is a whole number of 16-byte pieces");
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= M) return;
    const f32x4* src = (const f32x4*)(partials + (size_t)m * TILES * 2);
    float v[2 * TILES];
#pragma unroll
    for (int i = 0; i < TILES / 2; ++i) {
        const f32x4 q = src[i];
        v[4 * i] = q[0]; v[4 * i + 1] = q[1]; v[4 * i + 2] = q[2]; v[4 * i + 3] = q[3];
    }
    float mean, rstd;
    ln_merge(v, TILES, inv_d, mean, rstd);
    *(float2*)(rowstat + 2 * (size_t)m) = float2{mean, rstd};
}

// Split-K reduction of a residual GEMM (round 6, small batches; gemm.hip EPI_PARTIAL): the S fp32 partial products of every output element are added in the
// FIXED order s = 0 .. S - 1 (run-to-run deterministic; no atomics), then bias and the residual (hi + lo planes) exactly as the EPI_BIAS_RESID_LN epilogue adds
// them -- st = sum + bias, v = st + (hi + lo) -- and the row leaves as the two planes + the (sum, centred M2) statistics of its 64-column granules (8 lanes of
// 8 columns each, DPP adds in the epilogue's order).  One 8-column chunk per thread; every load of a thread is issued before the first add.
template <class T, int SMAX>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, const float* __restrict__ bias, uint16_t* __restrict__ x_hi,
                                                            size_t plane, float* __restrict__ stats_out, int M, int N) {
    const int cpr = N >> 3;
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = c < (size_t)M * cpr;
    const int m = ok ? (int)(c / cpr) : 0, ch = ok ? (int)(c - (size_t)m * cpr) : 0;
    const size_t off = (size_t)m * N + ch * 8, slab = (size_t)M * N;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (ok) {
        f32x4 p0[SMAX], p1[SMAX];
#pragma unroll
        for (int s = 0; s < SMAX; ++s)
            if (s < S) {
                p0[s] = *(const f32x4*)(part + s * slab + off);
                p1[s] = *(const f32x4*)(part + s * slab + off + 4);
            }
        const u32x4 ra = *(const u32x4*)(x_hi + off), rb = *(const u32x4*)(x_hi + plane + off);
        const f32x4 b0 = *(const f32x4*)(bias + ch * 8), b1 = *(const f32x4*)(bias + ch * 8 + 4);
        f32x4 a0 = p0[0], a1 = p1[0];
#pragma unroll
        for (int s = 1; s < SMAX; ++s)
            if (s < S) { a0 += p0[s]; a1 += p1[s]; }
        a0 += b0; a1 += b1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int sh = (e & 1) * 16;
            const float r = from_bits<T>((uint16_t)(ra[e >> 1] >> sh)) + from_bits<T>((uint16_t)(rb[e >> 1] >> sh));
            v[e] = (e < 4 ? a0[e] : a1[e - 4]) + r;
        }
        u32x4 oh, ol;
#pragma unroll
        for (int e = 0; e < 8; e += 2) { uint32_t h_, l_; split_planes2<T>(v[e], v[e + 1], h_, l_); oh[e >> 1] = h_; ol[e >> 1] = l_; }
        *(u32x4*)(x_hi + off) = oh;
        *(u32x4*)(x_hi + plane + off) = ol;
    }
    float s1 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    s1 = row8_sum_dpp(s1);
    const float mg = s1 * (1.0f / 64.0f);
    float s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float d = v[e] - mg;
        s2 = fmaf(d, d, s2);
    }
    s2 = row8_sum_dpp(s2);
    if (ok && (ch & 7) == 0) *(float2*)(stats_out + ((size_t)m * (N >> 6) + (ch >> 3)) * 2) = float2{s1, s2};
}

hipError_t splitk_reduce_launch(int dtype, const float* partials, int S, const float* bias, uint16_t* x_hi, size_t plane, float* stats_out, int M, int N,
                                hipStream_t s) {
    if (S < 1 || S > 8 || N % 64 != 0 || M <= 0) return hipErrorInvalidValue;
    const size_t chunks = (size_t)M * (N / 8);
    const dim3 grid((unsigned)((chunks + 255) / 256)), block(256);
    if (dtype == DT_F16) {
        if (S <= 4) hipLaunchKernelGGL((splitk_reduce_kernel<F16, 4>), grid, block, 0, s, partials, S, bias, x_hi, plane, stats_out, M, N);
        else hipLaunchKernelGGL((splitk_reduce_kernel<F16, 8>), grid, block, 0, s, partials, S, bias, x_hi, plane, stats_out, M, N);
    } else {
        if (S <= 4) hipLaunchKernelGGL((splitk_reduce_kernel<BF16, 4>), grid, block, 0, s, partials, S, bias, x_hi, plane, stats_out, M, N);
        else hipLaunchKernelGGL((splitk_reduce_kernel<BF16, 8>), grid, block, 0, s, partials, S, bias, x_hi, plane, stats_out, M, N);
    }
    return hipGetLastError();
}

hipError_t ln_finalize_launch(const float* partials, float* rowstat, int M, int tiles, int D, hipStream_t s) {
    const float inv_d = 1.0f / (float)D;
    const dim3 grid((M + 63) / 64), block(64);
    switch (tiles) {   // D / 64 of ViTPose-S / -B / -L / -H; anything else: the generic kernel
        case 6: hipLaunchKernelGGL(ln_finalize_kernel_t<6>, grid, block, 0, s, partials, rowstat, M, inv_d); break;
        case 12: hipLaunchKernelGGL(ln_finalize_kernel_t<12>, grid, block, 0, s, partials, rowstat, M, inv_d); break;
        case 16: hipLaunchKernelGGL(ln_finalize_kernel_t<16>, grid, block, 0, s, partials, rowstat, M, inv_d); break;
        case 20: hipLaunchKernelGGL(ln_finalize_kernel_t<20>, grid, block, 0, s, partials, rowstat, M, inv_d); break;
        default: hipLaunchKernelGGL(ln_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, s, partials, rowstat, M, tiles, inv_d);
    }
    return hipGetLastError();
}

template <class Ty>
__global__ void fill_random16_kernel(uint16_t* p, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        p[i] = to_bits<Ty>((float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f);
    }
}

hipError_t fill_random16(int dtype, uint16_t* p, size_t n, uint32_t seed, hipStream_t s) {
    if (dtype == DT_F16) hipLaunchKernelGGL(fill_random16_kernel<F16>, dim3(2048), dim3(256), 0, s, p, n, seed);
    else hipLaunchKernelGGL(fill_random16_kernel<BF16>, dim3(2048), dim3(256), 0, s, p, n, seed);
    return hipGetLastError();
}

}  // namespace vp

// ---------------------------------------------------------------------------
// Calibration micro-benchmarks (tools/ only): what this box's matrix pipe and HBM
// deliver, measured with the same compiler and launch path as the product kernels.
namespace vp {

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void peak_mfma_kernel(float* out, int iters) {
    u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    u32x4 b = {0x38003800u, 0x38003800u + threadIdx.x, 0x38003800u, 0x38003800u};
    if (SHAPE == 16) {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = mfma16<F16>(a, b, acc[i]);
        }
        f32x4 s = acc[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) s += acc[i];
        if (s[0] == 12345.f) out[threadIdx.x] = s[1];
    } else {
        typedef __attribute__((ext_vector_type(16))) float f32x16;
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[i], 0, 0, 0);
        }
        if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.f) out[threadIdx.x] = acc[0][1];
    }
}

__global__ __launch_bounds__(256) void peak_copy_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

// HBM ceiling, calibrated properly (peak_copy_kernel above keeps ONE 16-byte load in flight per lane and reads 4.6 TB/s,
// the guide's float4 copy 6.29).  A persistent grid walks chunks of 256 lanes x U x 16 bytes; the U loads of a chunk are issued back to
// back (U x 16 bytes in flight per lane, each wave instruction = 1 KiB contiguous), then the U stores.  MODE 0 = copy (n float4 in, n out),
// 1 = read only (the sum of the data is kept alive by a never-true store), 2 = write only.  NT = non-temporal loads / stores.
template <int MODE, int U, bool NT>
__global__ __launch_bounds__(256) void peak_stream_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n, uint32_t* sink) {
    const size_t chunk = (size_t)256 * U;
    const size_t nchunks = n / chunk;
    u32x4 keep = {0u, 0u, 0u, 0u};
    u32x4 fill = {threadIdx.x, 1u, 2u, 3u};
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
        u32x4 v[U];
        if (MODE != 2) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(in + base + (size_t)u * 256) : in[base + (size_t)u * 256];
        }
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) keep ^= v[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32x4 w = MODE == 2 ? fill : v[u];
                if (NT) __builtin_nontemporal_store(w, out + base + (size_t)u * 256);
                else out[base + (size_t)u * 256] = w;
            }
        }
    }
    if (MODE == 1 && (keep[0] ^ keep[1] ^ keep[2] ^ keep[3]) == 0x12345677u) sink[threadIdx.x] = keep[0];
}

// VALU issue-rate probe: per round 16 independent fp32 multiply-adds per lane as 16 v_fma_f32 (PK = 0) or as 8
// v_pk_fma_f32 on register pairs (PK = 1); NW waves per workgroup, one workgroup per CU; nothing else in the loop.  Is the packed form a
// throughput gain on this chip where no MFMA issues beside it (a GEMM epilogue)?
template <int NW, int PK>
__global__ __launch_bounds__(NW * 64, 1) void valu_probe_kernel(int iters, float* sink) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f32x2{1.0f + 0.001f * (float)(threadIdx.x + i), 0.5f + 0.002f * (float)(threadIdx.x + i)};
    const f32x2 m = {1.0001f, 0.9999f}, c = {0.25f, -0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (PK) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                } else {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i][0]) : "v"(m[0]), "v"(c[0]));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i][1]) : "v"(m[1]), "v"(c[1]));
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
    if (s == 12345.f) sink[threadIdx.x] = s;
}

// Where do the workgroups of a launch land?  Every workgroup records HW_REG_HW_ID, HW_REG_XCC_ID and the time it started (tools/hwid_probe.py):
// which wave slots / CU / XCD two co-resident 512-thread workgroups of a 2-per-CU launch get.
__global__ void hwid_probe_kernel(uint32_t* out, int spin) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID, 32 bits
        const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
        const unsigned long long t = __builtin_readcyclecounter();
        out[4 * blockIdx.x + 0] = hw;
        out[4 * blockIdx.x + 1] = xcc;
        out[4 * blockIdx.x + 2] = (uint32_t)t;
        out[4 * blockIdx.x + 3] = (uint32_t)(t >> 32);
        ((volatile char*)smem)[0] = 1;
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);   // stay resident so that the first round of workgroups fills the chip
}

hipError_t hwid_probe_launch(uint32_t* d_out, int blocks, int threads, int lds_bytes, int spin, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute((const void*)hwid_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(hwid_probe_kernel, dim3(blocks), dim3(threads), lds_bytes, s, d_out, spin);
    return hipGetLastError();
}

// LDS-DMA ceiling: every wave streams 1 KiB pieces global -> LDS (global_load_lds_dwordx4) from an L2-resident
// source, DEPTH pieces in flight, nothing else.  blocks_per_cu x 256 threads, 64 KiB LDS ring per block.
template <int DEPTH>
__global__ __launch_bounds__(256) void peak_glds_kernel(const char* __restrict__ src, size_t src_bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * 4 * 1024;
    size_t off = ((size_t)blockIdx.x * 4 + wave) * 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            glds16(src + (off & (src_bytes - 1)) + lane * 16, smem + ((wave * DEPTH + d) & 63) * 1024);   // src_bytes is a power of two
            off += stride;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (iters < 0) sink[threadIdx.x] = ((float*)smem)[threadIdx.x];
}

// Issue-interplay probe (tools/issue_probe.py): every wave runs `iters` rounds of 16 independent MFMAs (256 matrix-pipe cycles)
// and, per round, optionally (mode bit 0) ONE 16-byte-per-lane global store to a streaming address, (bit 1) ONE 1 KiB LDS-DMA
// load with 8 kept in flight, (bit 2) 48 independent VALU FMAs, (bit 3) 4 stores instead of 1.  NW waves per workgroup, one
// workgroup per CU: NW = 4 -> one wave per SIMD, NW = 8 -> two.  What a store / DMA / VALU block costs a wave whose matrix
// pipe is otherwise saturated is the difference to mode 0.
template <int NW, int mode>
__global__ __launch_bounds__(NW * 64, 1) void issue_probe_kernel(int iters, char* __restrict__ dst, const char* __restrict__ src,
                                                                 size_t mask, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    u32x4 b = {0x38003800u, 0x38003800u + threadIdx.x, 0x38003800u, 0x38003800u};
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = 1.0f + 0.001f * (float)(lane + i);
    size_t off = ((size_t)(blockIdx.x * NW + wave) * 1024 * 64) & mask;
    u32x4 data = {(uint32_t)lane, 1u, 2u, 3u};
    if (mode & 2) {
#pragma unroll
        for (int d = 0; d < 8; ++d) glds16(src + ((off + d * 1024) & mask) + lane * 16, smem + ((wave * 8 + d) & 127) * 1024);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] = mfma16<F16>(a, b, acc[i]);
            if (mode & 4) {
#pragma unroll
                for (int q = 0; q < 3; ++q) f[(i * 3 + q) & 7] = fmaf(f[(i * 3 + q) & 7], 1.0001f, 0.5f);
            }
            if (i == 3 && (mode & 1)) {
                *(u32x4*)(dst + off + lane * 16) = data;
                if (mode & 8) {
                    *(u32x4*)(dst + ((off + 1024) & mask) + lane * 16) = data;
                    *(u32x4*)(dst + ((off + 2048) & mask) + lane * 16) = data;
                    *(u32x4*)(dst + ((off + 3072) & mask) + lane * 16) = data;
                }
            }
            if (i == 9 && (mode & 2)) {
                glds16(src + off + lane * 16, smem + ((wave * 8 + (it & 7)) & 127) * 1024);
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            }
        }
        off = (off + 4096) & mask;
        data[1] += 1u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 s4 = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s4 += acc[i];
    float fs = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) fs += f[i];
    if (s4[0] + fs == 12345.f) sink[threadIdx.x] = s4[1] + ((float*)smem)[threadIdx.x];
}

// the same MFMA work per round as issue_probe_kernel mode 0 in the 32x32x16 shape (8 MFMAs of 32 matrix-pipe cycles), DEP = number
// of independent accumulators (8: no dependent pair closer than 8 MFMAs; 4 / 2: dependent issue distance 4 / 2)
template <int NW, int DEP>
__global__ __launch_bounds__(NW * 64, 1) void issue_probe32_kernel(int iters, float* sink) {
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    u32x4 b = {0x38003800u, 0x38003800u + threadIdx.x, 0x38003800u, 0x38003800u};
    f32x16 acc[DEP];
#pragma unroll
    for (int i = 0; i < DEP; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i % DEP] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[i % DEP], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < DEP; ++i) s += acc[i][0] + acc[i][5];
    if (s == 12345.f) sink[threadIdx.x] = s;
}

// kind 0: MFMA 16x16x32 f16, 1: MFMA 32x32x16 f16 (returns TFLOP/s); 2: float4 copy (returns TB/s read+write);
// 170 + (0 | 1 | 2: 8 / 4 / 2 independent accumulators) (+ 4: two waves per SIMD): the 32x32x16 probe, nanoseconds per round
// 100 + mode (+ 32: two waves per SIMD; mode bit 4: 1 MiB = L2-resident buffers): issue probe, returns nanoseconds per round
// 3 / 4 / 5 / 6: LDS-DMA stream from a 32 MiB / 1 GiB / 2 MiB / 256 KiB source, 2 blocks per CU (TB/s into LDS)
// Store-path probe (tools/store_probe.py): every wave issues ROUNDS x 16 stores of 16 bytes per lane, the way a gemm8 epilogue does (one 128 x 64
// block of 16-bit values per wave and "tile" = 16 KiB), into its own region of a buffer that is either small (stays in L2) or 1 GiB.
// PAT 0: a store instruction = 1 KiB contiguous; 1: 16 rows x 64 contiguous bytes at a row stride (4 lanes of a row adjacent);
// 2: 16 rows x four 16-byte pieces 32 bytes apart (the round-2/3 epilogue: a lane's 32 bytes of a row leave as two stores).
template <int PAT>
__global__ void store_probe_kernel(char* __restrict__ dst, size_t wave_bytes, size_t mask, int rounds, int row_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    const size_t base = ((size_t)blockIdx.x * nw + wave) * wave_bytes;
    const int frow = lane & 15, fg = lane >> 4;
    u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {   // i = (row group J, half h)
            const int J = i >> 1, h = i & 1;
            size_t off;
            if (PAT == 0) off = (size_t)i * 1024 + lane * 16;
            else if (PAT == 1) off = (size_t)(J * 16 + frow) * row_stride + h * 64 + fg * 16;
            else off = (size_t)(J * 16 + frow) * row_stride + fg * 32 + h * 16;
            off = (base + (size_t)r * 16384 + off) & mask;
            *(u32x4*)(dst + off) = v;
            v[1] += 1;
        }
    }
}

hipError_t peak_bench(int kind, double* result) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    if (kind >= 200 && kind < 248) {
        // 200 + 16 * waves code (0: 4, 1: 8, 2: 16 waves per CU) + 4 * pattern + 2 * (1 GiB instead of L2-resident) + (row stride 6144 instead of 128):
        // returns shader-clock-independent NANOSECONDS per store instruction and CU
        const int k = kind - 200, wc = k >> 4, pat = (k >> 2) & 3, big = (k >> 1) & 1, strided = k & 1;
        const int nw = wc == 0 ? 4 : wc == 1 ? 8 : 16;
        const size_t bytes = big ? ((size_t)1 << 30) : ((size_t)16 << 20);
        char* dst;
        hipMalloc((void**)&dst, bytes + (1 << 20));
        const int rounds = 64;
        const size_t wave_bytes = big ? bytes / (256 * nw) : (size_t)16384 * 4;   // small: every wave rewrites its own 64 KiB (chip total <= 16 MiB: L2 + MALL)
        const int row_stride = strided ? 6144 : 128;
        auto launch = [&]() {
            if (pat == 0) hipLaunchKernelGGL(store_probe_kernel<0>, dim3(256), dim3(nw * 64), 0, nullptr, dst, wave_bytes, bytes - 1, rounds, row_stride);
            else if (pat == 1) hipLaunchKernelGGL(store_probe_kernel<1>, dim3(256), dim3(nw * 64), 0, nullptr, dst, wave_bytes, bytes - 1, rounds, row_stride);
            else hipLaunchKernelGGL(store_probe_kernel<2>, dim3(256), dim3(nw * 64), 0, nullptr, dst, wave_bytes, bytes - 1, rounds, row_stride);
        };
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, nullptr);
            launch();
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
        }
        hipEventElapsedTime(&ms, e0, e1);
        *result = (double)ms * 1e6 / ((double)rounds * 16 * nw);   // ns per store instruction and CU
        hipFree(dst);
    } else if (kind == 0 || kind == 1) {
        float* d;
        hipMalloc(&d, 4096);
        const int iters = 20000, blocks = 256 * 2;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, nullptr);
            if (kind == 0) hipLaunchKernelGGL(peak_mfma_kernel<16>, dim3(blocks), dim3(256), 0, nullptr, d, iters);
            else hipLaunchKernelGGL(peak_mfma_kernel<32>, dim3(blocks), dim3(256), 0, nullptr, d, iters);
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
        }
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * (kind == 0 ? 16.0 * 16384 : 4.0 * 32768);
        *result = flops / (ms * 1e-3) / 1e12;
        hipFree(d);
    } else if (kind >= 170 && kind < 178) {
        const int dep = (kind - 170) & 3, two = (kind - 170) & 4;
        float* d;
        hipMalloc(&d, 4096);
        const int iters = 8000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, nullptr);
#define VP_P32(NW, D) hipLaunchKernelGGL((issue_probe32_kernel<NW, D>), dim3(256), dim3(NW * 64), 0, nullptr, iters, d)
            if (!two) { if (dep == 0) VP_P32(4, 8); else if (dep == 1) VP_P32(4, 4); else VP_P32(4, 2); }
            else { if (dep == 0) VP_P32(8, 8); else if (dep == 1) VP_P32(8, 4); else VP_P32(8, 2); }
#undef VP_P32
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
        }
        hipEventElapsedTime(&ms, e0, e1);
        *result = (double)ms * 1e6 / iters;
        hipFree(d);
    } else if (kind >= 100 && kind < 164) {
        const int mode = (kind - 100) & 31, nw = (kind - 100) & 32 ? 8 : 4;
        const size_t bytes = (mode & 16) ? ((size_t)1 << 20) : ((size_t)256 << 20);   // bit 4: all traffic inside 1 MiB (L2-resident)
        char *dst, *src; float* d;
        hipMalloc(&dst, bytes + 8192); hipMalloc(&src, bytes + 8192); hipMalloc(&d, 4096);
        hipMemset(src, 1, bytes + 8192);
        const int iters = 8000;
        hipError_t (*run)(int, int, char*, const char*, size_t, float*) = nullptr;
        switch (mode & 15) {
#define VP_PROBE(M) case M: run = [](int nw_, int it_, char* d_, const char* s_, size_t m_, float* k_) -> hipError_t { \
                if (nw_ == 4) { auto k = issue_probe_kernel<4, M>; hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
                    hipLaunchKernelGGL(k, dim3(256), dim3(256), 131072, nullptr, it_, d_, s_, m_, k_); } \
                else { auto k = issue_probe_kernel<8, M>; hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
                    hipLaunchKernelGGL(k, dim3(256), dim3(512), 131072, nullptr, it_, d_, s_, m_, k_); } \
                return hipGetLastError(); }; break;
            VP_PROBE(0) VP_PROBE(1) VP_PROBE(2) VP_PROBE(3) VP_PROBE(4) VP_PROBE(5) VP_PROBE(6) VP_PROBE(7)
            VP_PROBE(9) VP_PROBE(11) VP_PROBE(13) VP_PROBE(15)
#undef VP_PROBE
        }
        if (!run) { hipFree(dst); hipFree(src); hipFree(d); hipEventDestroy(e0); hipEventDestroy(e1); return hipErrorInvalidValue; }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, nullptr);
            run(nw, iters, dst, src, bytes - 1, d);
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
        }
        hipEventElapsedTime(&ms, e0, e1);
        *result = (double)ms * 1e6 / iters;
        hipFree(dst); hipFree(src); hipFree(d);
    } else if (kind >= 300 && kind < 492) {
        // 300 + 64 * mode (0 copy, 1 read, 2 write) + 32 * nt + 8 * ucode (U = 1, 2, 4, 8) + gridcode (workgroups = 256 x {2, 4, 8, 16, 32, 64}; 6: one per chunk; 7: 256 x 3):
        // TB/s of algorithmic traffic (copy: bytes read + bytes written) over 1 GiB per direction
        const int k = kind - 300, mode = k >> 6, nt = (k >> 5) & 1, uc = (k >> 3) & 3, gc = k & 7;
        const int U = 1 << uc;
        const size_t n = (size_t)1 << 26;   // 64 Mi x 16 bytes = 1 GiB
        u32x4 *a = nullptr, *b = nullptr; uint32_t* d = nullptr;
        if (mode != 2) { hipMalloc((void**)&a, n * 16); hipMemset(a, 1, n * 16); }
        if (mode != 1) { hipMalloc((void**)&b, n * 16); hipMemset(b, 2, n * 16); }
        hipMalloc((void**)&d, 4096);
        const size_t nchunks = n / ((size_t)256 * U);
        static const int gmul[8] = {2, 4, 8, 16, 32, 64, 0, 3};
        const unsigned grid = gc == 6 ? (unsigned)nchunks : 256u * gmul[gc];
#define VP_STREAM(M, UU, N) hipLaunchKernelGGL((peak_stream_kernel<M, UU, N>), dim3(grid), dim3(256), 0, nullptr, a, b, n, d)
#define VP_STREAM_U(M, N) do { if (U == 1) VP_STREAM(M, 1, N); else if (U == 2) VP_STREAM(M, 2, N); else if (U == 4) VP_STREAM(M, 4, N); else VP_STREAM(M, 8, N); } while (0)
#define VP_STREAM_N(M) do { if (nt) VP_STREAM_U(M, true); else VP_STREAM_U(M, false); } while (0)
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, nullptr);
            if (mode == 0) VP_STREAM_N(0); else if (mode == 1) VP_STREAM_N(1); else VP_STREAM_N(2);
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
#undef VP_STREAM_N
#undef VP_STREAM_U
#undef VP_STREAM
        *result = (mode == 0 ? 2.0 : 1.0) * n * 16 / (best * 1e-3) / 1e12;
        if (a) hipFree(a);
        if (b) hipFree(b);
        hipFree(d);
    } else if (kind >= 500 && kind < 504) {   // 500 + 2 * (two waves per SIMD) + packed: VALU probe, nanoseconds per round of 64 multiply-adds per lane
        const int pk = (kind - 500) & 1, two = (kind - 500) >> 1;
        float* d;
        hipMalloc(&d, 4096);
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, nullptr);
            if (!two) { if (pk) hipLaunchKernelGGL((valu_probe_kernel<4, 1>), dim3(256), dim3(256), 0, nullptr, iters, d); else hipLaunchKernelGGL((valu_probe_kernel<4, 0>), dim3(256), dim3(256), 0, nullptr, iters, d); }
            else { if (pk) hipLaunchKernelGGL((valu_probe_kernel<8, 1>), dim3(256), dim3(512), 0, nullptr, iters, d); else hipLaunchKernelGGL((valu_probe_kernel<8, 0>), dim3(256), dim3(512), 0, nullptr, iters, d); }
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
        }
        hipEventElapsedTime(&ms, e0, e1);
        *result = (double)ms * 1e6 / iters;
        hipFree(d);
    } else if (kind >= 3 && kind <= 12) {   // 7 .. 12: 64 / 96 / 128 / 192 / 256 / 384 MiB sources: is the 256 MB memory-side cache faster than HBM?
        static const size_t mib[] = {32, 1024, 2, 0, 64, 96, 128, 192, 256, 384};
        const size_t bytes = kind == 6 ? ((size_t)256 << 10) : (mib[kind - 3] << 20);
        char* a; float* d;
        hipMalloc(&a, bytes); hipMalloc(&d, 4096);
        hipMemset(a, 1, bytes);
        const int iters = 2000, blocks = 512, depth = 10;
        hipFuncSetAttribute((const void*)peak_glds_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, nullptr);
            hipLaunchKernelGGL(peak_glds_kernel<10>, dim3(blocks), dim3(256), 65536, nullptr, a, bytes, iters, d);
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
        }
        hipEventElapsedTime(&ms, e0, e1);
        *result = (double)blocks * 4 * depth * 1024.0 * iters / (ms * 1e-3) / 1e12;
        hipFree(a); hipFree(d);
    } else {
        const size_t n = (size_t)1 << 26;   // 64 Mi float4 = 1 GiB
        f32x4 *a, *b;
        hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
        hipMemset(a, 1, n * 16);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, nullptr);
            hipLaunchKernelGGL(peak_copy_kernel, dim3(256 * 16), dim3(256), 0, nullptr, a, b, n);
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
        }
        hipEventElapsedTime(&ms, e0, e1);
        *result = 2.0 * n * 16 / (ms * 1e-3) / 1e12;
        hipFree(a); hipFree(b);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return hipGetLastError();
}

}  // namespace vp

// ---------------------------------------------------------------------------
// Crop preparation on device: for every detected box, crop the frame, zero-pad to 3:4
// (pad_image, vit_utils/inference.py:41-70) and resize to 256x192 exactly as OpenCV's 8-bit INTER_LINEAR
// does (easy_ViTPose/inference.py:316): half-pixel centres, 11-bit fixed-point coefficients, int32
// horizontal pass, (((b0*(S0>>4))>>16)+((b1*(S1>>4))>>16)+2)>>2 vertical pass, 2x2 box average at exactly
// 2x.  Integer arithmetic -> bit-identical to easy_vitpose_amd/cropprep.py.  One block = one output row.
namespace vp {

struct AxisCoef { int s0, s1, a0, a1; };

__device__ __forceinline__ AxisCoef axis_coef(int d, int dsize, int ssize) {
    const double inv = (double)dsize / (double)ssize;
    const double scale = 1.0 / inv;
    float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
    int s = (int)floorf(f);
    f = __fsub_rn(f, (float)s);
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    AxisCoef c;
    c.s0 = s;
    c.s1 = min(s + 1, ssize - 1);
    c.a1 = (int)rintf(__fmul_rn(f, 2048.f));
    c.a0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
    return c;
}

__global__ __launch_bounds__(192) void crop_resize_kernel(const uint8_t* __restrict__ frame, int FH, int FW,
                                                          const int32_t* __restrict__ params, uint8_t* __restrict__ out) {
    const int crop = blockIdx.x >> 8, oy = blockIdx.x & 255, ox = threadIdx.x;
    const int32_t* p = params + crop * 8;
    const int x0 = p[0], y0 = p[1], cw = p[2], ch = p[3], left = p[4], top = p[5], pw = p[6], ph = p[7];
    auto px = [&](int Y, int X, int c) -> int {   // padded-canvas pixel
        const int yy = Y - top, xx = X - left;
        if ((unsigned)yy >= (unsigned)ch || (unsigned)xx >= (unsigned)cw) return 0;
        return frame[((size_t)(y0 + yy) * FW + (x0 + xx)) * 3 + c];
    };
    uint8_t* dst = out + (((size_t)crop * 256 + oy) * 192 + ox) * 3;
    if (pw == 384 && ph == 512) {                 // exactly 2x: OpenCV's INTER_LINEAR == fast INTER_AREA
#pragma unroll
        for (int c = 0; c < 3; ++c)
            dst[c] = (uint8_t)((px(2 * oy, 2 * ox, c) + px(2 * oy, 2 * ox + 1, c) + px(2 * oy + 1, 2 * ox, c) +
                                px(2 * oy + 1, 2 * ox + 1, c) + 2) >> 2);
        return;
    }
    const AxisCoef cx = axis_coef(ox, 192, pw), cy = axis_coef(oy, 256, ph);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int r0 = px(cy.s0, cx.s0, c) * cx.a0 + px(cy.s0, cx.s1, c) * cx.a1;
        const int r1 = px(cy.s1, cx.s0, c) * cx.a0 + px(cy.s1, cx.s1, c) * cx.a1;
        dst[c] = (uint8_t)((((cy.a0 * (r0 >> 4)) >> 16) + ((cy.a1 * (r1 >> 4)) >> 16) + 2) >> 2);
    }
}

hipError_t crop_resize_launch(const uint8_t* frame, int FH, int FW, const int32_t* params, uint8_t* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(crop_resize_kernel, dim3(n * 256), dim3(192), 0, s, frame, FH, FW, params, out);
    return hipGetLastError();
}

}  // namespace vp
