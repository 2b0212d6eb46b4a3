// Heatmap decode: keypoints_from_heatmaps(unbiased=True, use_udp=True) + postprocess
// (vit_utils/top_down_eval.py:493-641, easy_ViTPose/inference.py:187-205), with the
// reference's one-crop-at-a-time semantics (VitInference calls it with N == 1).
//
// HBM-bound: the only full read is one coalesced float4 sweep of each 64x48 map for
// the arg-max (first index wins ties, top_down_eval.py:106).  The DARK refinement
// (post_dark_udp, :354-415) needs the 11x11 Gaussian-blurred, clipped, log'ed map at
// only 7 positions around the arg-max, so instead of blurring 3072 pixels per joint
// (what the reference's N*K cv2.GaussianBlur calls do) the block evaluates those 7
// samples directly from the raw map (the re-read hits L2): 7 x 11 row sums
// (horizontal pass, fp32 like OpenCV's intermediate), then 7 column sums.
// The sample positions follow the reference's flat index arithmetic into the
// edge-padded map exactly -- including the wrap-around of negative indices that
// numpy fancy indexing performs when coords are -1 (max <= 0).
#include "common.h"
#include "kernels.h"

namespace vp {

static constexpr int HH = 64, WW = 48, HW = HH * WW;
static constexpr int PADSZ = (HH + 2) * (WW + 2);   // 3300

__device__ __forceinline__ int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

struct GaussK { float w[11]; };

__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ hm, const int32_t* __restrict__ org_wh,
                                                     float* __restrict__ out, int K, GaussK gk) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    __shared__ float s_part[7][11];
    __shared__ float s_samp[7];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x / K, k = blockIdx.x % K;
    const float* map = hm + (size_t)blockIdx.x * HW;

    // ---- arg-max / max over 3072 values, first index on ties (_get_max_preds, :82-114)
    float best = -INFINITY;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int v4 = tid + 256 * i;
        const f32x4 v = ((const f32x4*)map)[v4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (v[e] > best || bidx == 0x7fffffff) { best = v[e]; bidx = v4 * 4 + e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = bidx; }
    __syncthreads();
    best = s_val[0]; bidx = s_idx[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float ov = s_val[w];
        const int oi = s_idx[w];
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    const float maxval = best;
    const float cx = maxval > 0.f ? (float)(bidx % WW) : -1.f;
    const float cy = maxval > 0.f ? (float)(bidx / WW) : -1.f;

    // ---- 7 samples of log(clip(blur(map))) at the reference's flat padded indices (:389-402)
    // order: i_, ix1, iy1, ix1y1, ix1_y1_, ix1_, iy1_
    if (tid < 77) {
        const int s = tid / 11, ty = tid % 11;
        const int offs[7] = {0, 1, WW + 2, WW + 3, -WW - 3, -1, -2 - WW};
        const int total = K * PADSZ;
        int f = (int)cx + 1 + ((int)cy + 1) * (WW + 2) + k * PADSZ + offs[s];
        f %= total;
        if (f < 0) f += total;                        // numpy negative-index wrap (per-crop array)
        const int kk = f / PADSZ, rem = f % PADSZ;
        int py = rem / (WW + 2) - 1, px = rem % (WW + 2) - 1;
        py = min(max(py, 0), HH - 1);                 // np.pad(mode='edge')
        px = min(max(px, 0), WW - 1);
        const float* src = hm + ((size_t)n * K + kk) * HW + reflect101(py + ty - 5, HH) * WW;
        float acc = 0.f;
#pragma unroll
        for (int tx = 0; tx < 11; ++tx) acc += gk.w[tx] * src[reflect101(px + tx - 5, WW)];
        s_part[s][ty] = acc;
    }
    __syncthreads();
    if (tid < 7) {
        float acc = 0.f;
#pragma unroll
        for (int ty = 0; ty < 11; ++ty) acc += gk.w[ty] * s_part[tid][ty];
        acc = fminf(fmaxf(acc, 0.001f), 50.f);       // np.clip(.., 0.001, 50)  :386
        s_samp[tid] = logf(acc);                     // np.log                  :387
    }
    __syncthreads();
    if (tid == 0) {
        const float i_ = s_samp[0], ix1 = s_samp[1], iy1 = s_samp[2], ix1y1 = s_samp[3];
        const float ix1_y1_ = s_samp[4], ix1_ = s_samp[5], iy1_ = s_samp[6];
        const float dx = 0.5f * (ix1 - ix1_);
        const float dy = 0.5f * (iy1 - iy1_);
        const float dxx = ix1 - 2.f * i_ + ix1_;
        const float dyy = iy1 - 2.f * i_ + iy1_;
        const float dxy = 0.5f * (ix1y1 - ix1 - iy1 + i_ + i_ - ix1_ - iy1_ + ix1_y1_);
        // inv(H + eps_f32 * I) in float64 (:411-413)
        const double eps = 1.1920928955078125e-07;
        const double a = (double)dxx + eps, b = (double)dxy, d = (double)dyy + eps;
        const double det = a * d - b * b;
        const double ox = (d * (double)dx - b * (double)dy) / det;
        const double oy = (-b * (double)dx + a * (double)dy) / det;
        const float rx = (float)((double)cx - ox);   // coords -= H^-1 d, stored back as float32 (:414)
        const float ry = (float)((double)cy - oy);
        // transform_preds(use_udp=True) with center = (w//2, h//2), scale = (w, h)
        // (post_transforms.py:183-192, inference.py:200-204), float64 then float32
        int ow = 192, oh = 256;
        if (org_wh) { ow = org_wh[2 * n]; oh = org_wh[2 * n + 1]; }
        const double fx = (double)rx * ((double)ow / (WW - 1.0)) + (double)(ow / 2) - (double)ow * 0.5;
        const double fy = (double)ry * ((double)oh / (HH - 1.0)) + (double)(oh / 2) - (double)oh * 0.5;
        float* o = out + (size_t)blockIdx.x * 3;
        o[0] = (float)fy;                            // (y, x, conf)  inference.py:205
        o[1] = (float)fx;
        o[2] = maxval;
    }
}

hipError_t decode_launch(const float* hm, const int32_t* org_wh, float* out, int N, int K, hipStream_t s) {
    // OpenCV getGaussianKernel(11, sigma<=0): sigma = 0.3*((11-1)*0.5-1)+0.8 = 2.0, float32 weights, sum 1
    GaussK gk;
    double w[11], sum = 0.0;
    for (int i = 0; i < 11; ++i) { w[i] = exp(-((i - 5.0) * (i - 5.0)) / (2.0 * 2.0 * 2.0)); sum += w[i]; }
    for (int i = 0; i < 11; ++i) gk.w[i] = (float)(w[i] / sum);
    hipLaunchKernelGGL(decode_kernel, dim3(N * K), dim3(256), 0, s, hm, org_wh, out, K, gk);
    return hipGetLastError();
}

}  // namespace vp
