#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include "../../easy_vitpose_amd/csrc/common.h"
using namespace vp;
template <int TILES>
__global__ void k(const float* part, float* out_ref, float* out_pair, int M) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = tid >> 1;
    if (r >= M) return;
    float mean, rstd;
    ln_merge(part + (size_t)r * TILES * 2, TILES, 1.0f / (TILES * 64), mean, rstd);
    const float2 p = ln_pair_merge_any(part + (size_t)r * TILES * 2, TILES, tid & 1, 1.0f / (TILES * 64));
    if (!(tid & 1)) { out_ref[2 * r] = mean; out_ref[2 * r + 1] = rstd; }
    out_pair[2 * tid] = p.x; out_pair[2 * tid + 1] = p.y;
}
int main() {
    const int M = 4096;
    for (int T : {6, 12, 16, 20}) {
        std::vector<float> h((size_t)M * T * 2);
        for (auto& v : h) v = (float)rand() / RAND_MAX * 100.f - 20.f;
        for (size_t i = 1; i < h.size(); i += 2) h[i] = fabsf(h[i]) * 10;
        float *d, *o1, *o2;
        hipMalloc(&d, h.size() * 4); hipMalloc(&o1, M * 8); hipMalloc(&o2, M * 16);
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        if (T == 6) k<6><<<M * 2 / 256, 256>>>(d, o1, o2, M); else if (T == 12) k<12><<<M * 2 / 256, 256>>>(d, o1, o2, M);
        else if (T == 16) k<16><<<M * 2 / 256, 256>>>(d, o1, o2, M); else k<20><<<M * 2 / 256, 256>>>(d, o1, o2, M);
        std::vector<float> r(M * 2), p(M * 4);
        hipMemcpy(r.data(), o1, M * 8, hipMemcpyDeviceToHost); hipMemcpy(p.data(), o2, M * 16, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int m = 0; m < M; ++m) for (int l = 0; l < 2; ++l) for (int c = 0; c < 2; ++c) if (p[(2 * m + l) * 2 + c] != r[2 * m + c]) { if (bad < 4) printf("T=%d row %d lane %d comp %d: pair %g ref %g\n", T, m, l, c, p[(2 * m + l) * 2 + c], r[2 * m + c]); ++bad; }
        printf("tiles %d: %d mismatches of %d\n", T, bad, M * 4);
    }
    return 0;
}
