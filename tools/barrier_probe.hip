// What does a dependent kernel boundary cost inside a hipGraph, and what does a device-wide barrier inside ONE persistent kernel cost?  Decides whether a one-launch
// encoder for 1-crop calls (72 launches on ViTPose-B) could pay: it cannot -- a dependent launch of 256 workgroups that move 4 KiB each costs 1.9-3.1 us ALL IN inside a
// graph (profiles/small_batch_r6.txt, call 14), a counter barrier across the 8 XCDs 7-26 us.
// Standalone (GPU box):  hipcc -O3 --offload-arch=gfx950 tools/barrier_probe.hip -o /tmp/barrier_probe && /tmp/barrier_probe [workgroups [floats per workgroup]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void phase_kernel(const float* __restrict__ in, float* __restrict__ out, int words, int shift) {
    // every workgroup reads `words` floats another workgroup wrote in the previous phase and writes its own
    const int nb = gridDim.x, b = blockIdx.x, src = (b + shift) % nb;
    for (int i = threadIdx.x; i < words; i += blockDim.x) out[(size_t)b * words + i] = in[(size_t)src * words + i] + 1.0f;
}

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();   // release: this workgroup's stores reach device scope
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000) { *err = 1; break; }
        }
        __threadfence();
    }
    __syncthreads();
    return true;
}

__global__ void persistent_kernel(float* bufA, float* bufB, int words, int shift, int phases, unsigned* counter, int* err) {
    const int nb = gridDim.x, b = blockIdx.x, src = (b + shift) % nb;
    float* in = bufA; float* out = bufB;
    for (int p = 0; p < phases; ++p) {
        for (int i = threadIdx.x; i < words; i += blockDim.x) {
            float v = __builtin_nontemporal_load(in + (size_t)src * words + i);   // must not come from a stale L2 line of this XCD
            out[(size_t)b * words + i] = v + 1.0f;
        }
        grid_barrier(counter, (unsigned)(nb * (p + 1)), err);
        float* t = in; in = out; out = t;
    }
}

int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 256, words = argc > 2 ? atoi(argv[2]) : 1024, phases = 72, reps = 50;
    float *a, *b; unsigned* counter; int* err;
    CK(hipMalloc(&a, (size_t)nb * words * 4)); CK(hipMalloc(&b, (size_t)nb * words * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&err, 4));
    CK(hipMemset(a, 0, (size_t)nb * words * 4)); CK(hipMemset(err, 0, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int shift = nb / 8 * 3 + 1;   // a workgroup of another XCD
    // 1. a chain of `phases` kernels in a hipGraph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(nb), dim3(256), 0, s, (p & 1) ? b : a, (p & 1) ? a : b, words, shift);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipMemset(a, 0, (size_t)nb * words * 4));
    CK(hipEventRecord(e0, s)); for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> h((size_t)nb * words); CK(hipMemcpy(h.data(), a, h.size() * 4, hipMemcpyDeviceToHost));
    bool ok = true; for (float v : h) ok &= v == (float)(phases * reps);
    printf("graph of %d dependent kernels (%d workgroups, %d floats each): %.2f us per kernel   values %s\n", phases, nb, words, ms * 1e3 / (reps * phases), ok ? "exact" : "WRONG");
    // 2. one persistent kernel with device-wide barriers
    CK(hipMemset(a, 0, (size_t)nb * words * 4));
    for (int i = 0; i < 3; ++i) { CK(hipMemsetAsync(counter, 0, 4, s)); hipLaunchKernelGGL(persistent_kernel, dim3(nb), dim3(256), 0, s, a, b, words, shift, phases, counter, err); }
    CK(hipStreamSynchronize(s));
    CK(hipMemset(a, 0, (size_t)nb * words * 4));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) { CK(hipMemsetAsync(counter, 0, 4, s)); hipLaunchKernelGGL(persistent_kernel, dim3(nb), dim3(256), 0, s, a, b, words, shift, phases, counter, err); }
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), a, h.size() * 4, hipMemcpyDeviceToHost));
    int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    ok = true; for (float v : h) ok &= v == (float)(phases * reps);
    printf("one persistent kernel, %d device-wide barriers: %.2f us per phase (incl. 1/%d of a launch + memset)   values %s   spin bailout %d\n", phases, ms * 1e3 / (reps * phases), phases, ok ? "exact" : "WRONG", herr);
    return 0;
}
