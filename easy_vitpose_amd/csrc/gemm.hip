// MFMA GEMM for the ViT encoder / head:  C[m][n] = sum_k A[m][k] * W[n][k]
//
// gfx950 design (not a warp-tiled CUDA kernel recompiled):
//  * 128(m) x 128(n) x 64(k) block tile, 256 threads = 4 waves (2 x 2), each wave a
//    64 x 64 output tile = 4 x 4 MFMA 16x16x32 accumulators (fp32).
//  * operands are swapped into the MFMA: the W tile is the MFMA "A" operand (rows =
//    n), the activation tile is the "B" operand (cols = m).  A lane's 4 accumulator
//    registers are then 4 CONSECUTIVE n of one m: the epilogue reads bias/residual
//    and writes its result as one 8-byte (16-bit out) or 16-byte (fp32 out) access.
//  * both operand tiles go HBM -> LDS with global_load_lds (16 B per lane, no VGPR
//    round trip), double buffered, one barrier per k-step.  The LDS image must be
//    lane-linear, so the bank-conflict XOR swizzle (16-B slot ^= (row>>1)&7 inside a
//    128-B row) is applied to the per-lane SOURCE address and again on the
//    ds_read_b128 fragment reads (conflict-free for the b128 lane groups).
//  * blockIdx -> tile mapping is XCD-aware: each of the 8 XCDs walks a contiguous
//    range of tiles (n fastest), so the activation panel of an m-tile is fetched
//    into one private L2 and shared by the n-tiles that follow.
//  * the deconv layers are implicit GEMMs: ConvTranspose2d(k=4,s=2,p=1) splits into 4
//    output-parity classes, each a GEMM with K = 4*Cin whose A rows are gathered
//    (one 128-B chunk per row per k-step, zero row at the border) straight by the
//    global_load_lds source addresses -- no im2col buffer in HBM.
#include "common.h"
#include "kernels.h"

namespace vp {

static constexpr int BM = 128, BN = 128, BK = 64;
static constexpr int TILE_BYTES = 128 * BK * 2;      // 16 KiB per operand tile
static constexpr int STAGE_BYTES = 2 * TILE_BYTES;   // W tile + A tile
static constexpr int GEMM_LDS = 2 * STAGE_BYTES;     // double buffered: 64 KiB

__device__ __forceinline__ float gelu_erf(float x) {   // nn.GELU(approximate='none'), vit.py:127
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
}

template <class T, int EPI, int AMODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (g.M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int n0 = (bid % tiles_n) * BN, m0 = (bid / tiles_n) * BM;
    const int parity = (AMODE == A_DECONV) ? blockIdx.y : 0;
    const int K = g.K;

    // ---- staging: 4 x (8 rows x 128 B) pieces per operand per wave ----
    const int srow = wave * 8 + (lane >> 3);                          // + 32*q
    const int sslot = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);    // logical 16-B slot of this lane
    const uint16_t* wsrc[4];
    const uint16_t* asrc[4];
    int ai[4], aj[4];
    const uint16_t* W = g.W + (AMODE == A_DECONV ? (size_t)parity * ((size_t)tiles_n * BN) * K : 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = q * 32 + srow;
        wsrc[q] = W + (size_t)(n0 + r) * K + sslot * 8;
        int m = m0 + r;
        if (m > g.M - 1) m = g.M - 1;
        if (AMODE == A_DENSE) {
            asrc[q] = g.A + (size_t)m * K + sslot * 8;
        } else {
            const int j = m % g.Win, t = m / g.Win;
            ai[q] = t % g.Hin;
            aj[q] = j;
            asrc[q] = g.A + (size_t)m * g.Cin + sslot * 8;
        }
    }
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE_BYTES + wave * 1024;
        const int k0 = kt * BK;
        if (AMODE == A_DENSE) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                glds16(wsrc[q] + k0, base + q * 4096);
                glds16(asrc[q] + k0, base + TILE_BYTES + q * 4096);
            }
        } else {
            const int tap = k0 / g.Cin, c0 = k0 - tap * g.Cin;
            const int ti = tap >> 1, tj = tap & 1;
            // parity a (rows): a=0 -> taps ky=1 (di=0), ky=3 (di=-1); a=1 -> ky=0 (di=+1), ky=2 (di=0)
            const int pa = parity >> 1, pb = parity & 1;
            const int di = pa ? (ti ? 0 : 1) : (ti ? -1 : 0);
            const int dj = pb ? (tj ? 0 : 1) : (tj ? -1 : 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                glds16(wsrc[q] + k0, base + q * 4096);
                const bool ok = (unsigned)(ai[q] + di) < (unsigned)g.Hin && (unsigned)(aj[q] + dj) < (unsigned)g.Win;
                const uint16_t* p = ok ? asrc[q] + (ptrdiff_t)(di * g.Win + dj) * g.Cin + c0 : g.zero + sslot * 8;
                glds16(p, base + TILE_BYTES + q * 4096);
            }
        }
    };

    // ---- fragment read offsets ----
    const int wn = wave >> 1, wm = wave & 1;
    const int frow = lane & 15, fg = lane >> 4;
    const int foff = frow * 128 + ((fg ^ (frow >> 1)) << 4);   // k-half 0; k-half 1 = foff ^ 64
    const int woff = wn * 64 * 128 + foff;
    const int aoff = TILE_BYTES + wm * 64 * 128 + foff;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // tile kt landed for every wave; everyone is done reading the other buffer
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* sb = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 wf[4], af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wf[i] = *(const u32x4*)(sb + ((woff + i * 2048) ^ (kk << 6)));
                af[i] = *(const u32x4*)(sb + ((aoff + i * 2048) ^ (kk << 6)));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(wf[i], af[j], acc[i][j]);
        }
    }

    // ---- epilogue: lane owns 4 consecutive n (= nb..nb+3) of row m, per (i, j) ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + frow;
        if (m >= g.M) continue;
        size_t orow;
        if (EPI == EPI_DECONV) {
            const int jj = m % g.Win, t = m / g.Win, ii = t % g.Hin, img = t / g.Hin;
            orow = ((size_t)(img * 2 * g.Hin + 2 * ii + (parity >> 1)) * (2 * g.Win) + 2 * jj + (parity & 1)) * (size_t)g.ldo;
        } else if (EPI == EPI_HEATMAP) {
            const int img = m / 3072, p = m - img * 3072;
            orow = (size_t)img * g.Kp * 3072 + p;
        } else {
            orow = (size_t)m * g.ldo;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nb = n0 + wn * 64 + i * 16 + fg * 4;
            if (nb >= g.N) continue;
            f32x4 v = acc[i][j];
            if (EPI != EPI_POS) {
                const f32x4 b = *(const f32x4*)(g.bias + nb);
                v += b;
            }
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_DECONV) {
                if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                }
                if (EPI == EPI_DECONV) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                u32x2 o;
                o[0] = pack2<T>(v[0], v[1]);
                o[1] = pack2<T>(v[2], v[3]);
                *(u32x2*)((uint16_t*)g.out + orow + nb) = o;
            } else if (EPI == EPI_BIAS_RESID) {
                const f32x4 r = *(const f32x4*)(g.aux + orow + nb);
                *(f32x4*)((float*)g.out + orow + nb) = v + r;
            } else if (EPI == EPI_POS) {
                const f32x4 r = *(const f32x4*)(g.aux + (size_t)(m % 192) * g.ldo + nb);
                *(f32x4*)((float*)g.out + orow + nb) = v + r;
            } else {  // EPI_HEATMAP: out[(img*Kp + n) * 3072 + p]
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nb + r < g.N) ((float*)g.out)[orow + (size_t)(nb + r) * 3072] = v[r];
            }
        }
    }
}

template <class T, int EPI, int AMODE>
static hipError_t launch(const GemmArgs& a, hipStream_t s) {
    auto kern = gemm_kernel<T, EPI, AMODE>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    dim3 grid(tiles, AMODE == A_DECONV ? 4 : 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), GEMM_LDS, s, a);
    return hipGetLastError();
}

template <class T>
static hipError_t dispatch(int epi, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS: return launch<T, EPI_BIAS, A_DENSE>(a, s);
        case EPI_BIAS_GELU: return launch<T, EPI_BIAS_GELU, A_DENSE>(a, s);
        case EPI_BIAS_RESID: return launch<T, EPI_BIAS_RESID, A_DENSE>(a, s);
        case EPI_POS: return launch<T, EPI_POS, A_DENSE>(a, s);
        case EPI_DECONV: return launch<T, EPI_DECONV, A_DECONV>(a, s);
        case EPI_HEATMAP: return launch<T, EPI_HEATMAP, A_DENSE>(a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t gemm_launch(int dtype, int epi, const GemmArgs& a, hipStream_t s) {
    if (a.K % BK != 0 || a.M <= 0 || a.N <= 0) return hipErrorInvalidValue;
    if (epi == EPI_DECONV && (a.Cin % BK != 0 || a.K != 4 * a.Cin)) return hipErrorInvalidValue;
    if (epi != EPI_HEATMAP && (a.N % 4 != 0 || a.ldo % 4 != 0)) return hipErrorInvalidValue;  // 8/16-byte epilogue stores
    return dtype == DT_F16 ? dispatch<F16>(epi, a, s) : dispatch<BF16>(epi, a, s);
}

}  // namespace vp
