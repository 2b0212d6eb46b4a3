// MFMA GEMM for the ViT encoder / head:  C[m][n] = sum_k A[m][k] * W[n][k]
//
// gfx950 design (not a warp-tiled CUDA kernel recompiled):
//  * block tile BM(m) x BN(n) x BK(k), waves arranged NWM x NWN, each wave a WM x WN
//    output tile of MFMA 16x16x32 accumulators (fp32).  Tile shapes are template
//    configurations (TileCfg); gemm_launch picks one per problem.
//  * operands are swapped into the MFMA: the W tile is the MFMA "A" operand (rows =
//    n), the activation tile is the "B" operand (cols = m).  A lane's 4 accumulator
//    registers are then 4 CONSECUTIVE n of one m: the epilogue reads bias/residual
//    and writes its result as one 8-byte (16-bit out) or 16-byte (fp32 out) access.
//  * both operand tiles go HBM/L2 -> LDS with global_load_lds (16 B per lane, no VGPR
//    round trip) through a STAGES-deep ring, one barrier per k-step, counted
//    s_waitcnt vmcnt(N) so that STAGES-2 tiles stay in flight across the barrier
//    (raw s_barrier: __syncthreads() would drain the LDS-DMA queue).
//  * the LDS image of a tile is lane-linear (global_load_lds writes base + lane*16),
//    so the bank-conflict XOR swizzle of the 16-B slots is applied to the per-lane
//    SOURCE address and again on the ds_read_b128 fragment reads:
//       BK=64 (128-B rows): slot ^= (row>>1)&7      BK=32 (64-B rows): slot ^= 3*((row>>3)&1)
//    both are conflict-free for the four 16-lane groups ds_read_b128 is serviced in.
//  * blockIdx -> tile mapping: XCD-aware (each of the 8 XCDs walks a contiguous range
//    of tiles) and grouped (GROUP_M m-tiles x all n-tiles at a time), so the blocks
//    resident on one XCD share a small set of A and W panels in its private 4 MiB L2.
//  * the deconv layers are implicit GEMMs: ConvTranspose2d(k=4,s=2,p=1) splits into 4
//    output-parity classes, each a GEMM with K = 4*Cin whose A rows are gathered
//    (one row chunk per k-step, zero row at the border) straight by the
//    global_load_lds source addresses -- no im2col buffer in HBM.
#include "common.h"
#include "kernels.h"

namespace vp {

template <int BM_, int BN_, int BK_, int WM_, int WN_, int STAGES_>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, STAGES = STAGES_;
    static constexpr int NWM = BM / WM, NWN = BN / WN, NWAVES = NWM * NWN, NT = NWAVES * 64;
    static constexpr int ROWB = BK * 2;            // bytes per tile row
    static constexpr int SLOTS = ROWB / 16;        // 16-B slots per row (8 or 4)
    static constexpr int RPG = 1024 / ROWB;        // rows per global_load_lds wave-instruction (8 or 16)
    static constexpr int W_BYTES = BN * ROWB, A_BYTES = BM * ROWB, STAGE_BYTES = W_BYTES + A_BYTES;
    static constexpr int LDS = STAGES * STAGE_BYTES;
    static constexpr int WP = BN / RPG / NWAVES, AP = BM / RPG / NWAVES;   // pieces per wave per stage
    static constexpr int G = WP + AP;
    static constexpr int KK = BK / 32, TI = WN / 16, TJ = WM / 16;
    static_assert(BK == 64 || BK == 32, "BK");
    static_assert(WP * RPG * NWAVES == BN && AP * RPG * NWAVES == BM, "tile rows must split evenly over waves");
    static_assert(LDS <= 160 * 1024, "LDS");
};

template <int BK> __device__ __forceinline__ int swz(int row, int slot) {
    return BK == 64 ? (slot ^ ((row >> 1) & 7)) : (slot ^ (((row >> 3) & 1) * 3));
}

__device__ __forceinline__ float gelu_erf(float x) {   // nn.GELU(approximate='none'), vit.py:127
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <class T, int EPI, int AMODE, class C>
__global__ __launch_bounds__(C::NT, 2) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (g.N + C::BN - 1) / C::BN;
    const int tiles_m = (g.M + C::BM - 1) / C::BM;
    int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tm, tn;
    if (g.group_m > 1) {   // grouped order: GROUP_M m-tiles x all n-tiles, m fastest inside the group
        const int per_group = g.group_m * tiles_n;
        const int grp = bid / per_group, first_m = grp * g.group_m;
        const int gsz = min(tiles_m - first_m, g.group_m);
        const int r = bid - grp * per_group;
        tm = first_m + r % gsz;
        tn = r / gsz;
    } else {
        tm = bid / tiles_n;
        tn = bid - tm * tiles_n;
    }
    const int n0 = tn * C::BN, m0 = tm * C::BM;
    const int parity = (AMODE == A_DECONV) ? blockIdx.y : 0;
    const int K = g.K;

    // ---- staging addresses: piece p of an operand = rows [(p*NWAVES + wave)*RPG, +RPG) ----
    const int rip = lane / C::SLOTS, pslot = lane % C::SLOTS;
    const uint16_t* wsrc[C::WP];
    const uint16_t* asrc[C::AP];
    int ai[C::AP], aj[C::AP];
    const uint16_t* W = g.W + (AMODE == A_DECONV ? (size_t)parity * g.w_parity_stride : 0);
#pragma unroll
    for (int p = 0; p < C::WP; ++p) {
        const int r = (p * C::NWAVES + wave) * C::RPG + rip;
        wsrc[p] = W + (size_t)(n0 + r) * K + swz<C::BK>(r, pslot) * 8;
    }
#pragma unroll
    for (int p = 0; p < C::AP; ++p) {
        const int r = (p * C::NWAVES + wave) * C::RPG + rip;
        int m = m0 + r;
        if (m > g.M - 1) m = g.M - 1;
        const int sl = swz<C::BK>(r, pslot) * 8;
        if (AMODE == A_DENSE) {
            asrc[p] = g.A + (size_t)m * K + sl;
        } else {
            const int t = m / g.Win;
            ai[p] = t % g.Hin;
            aj[p] = m - t * g.Win;
            asrc[p] = g.A + (size_t)m * g.Cin + sl;
        }
    }
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * C::STAGE_BYTES + wave * 1024;
        const int k0 = kt * C::BK;
#pragma unroll
        for (int p = 0; p < C::WP; ++p) glds16(wsrc[p] + k0, base + p * (C::NWAVES * 1024));
        if (AMODE == A_DENSE) {
#pragma unroll
            for (int p = 0; p < C::AP; ++p) glds16(asrc[p] + k0, base + C::W_BYTES + p * (C::NWAVES * 1024));
        } else {
            const int tap = k0 / g.Cin, c0 = k0 - tap * g.Cin;
            const int ti = tap >> 1, tj = tap & 1;
            // parity a (rows): a=0 -> taps ky=1 (di=0), ky=3 (di=-1); a=1 -> ky=0 (di=+1), ky=2 (di=0)
            const int pa = parity >> 1, pb = parity & 1;
            const int di = pa ? (ti ? 0 : 1) : (ti ? -1 : 0);
            const int dj = pb ? (tj ? 0 : 1) : (tj ? -1 : 0);
#pragma unroll
            for (int p = 0; p < C::AP; ++p) {
                const bool ok = (unsigned)(ai[p] + di) < (unsigned)g.Hin && (unsigned)(aj[p] + dj) < (unsigned)g.Win;
                const uint16_t* src = ok ? asrc[p] + (ptrdiff_t)(di * g.Win + dj) * g.Cin + c0 : g.zero + pslot * 8;
                glds16(src, base + C::W_BYTES + p * (C::NWAVES * 1024));
            }
        }
    };

    // ---- fragment read offsets (bytes inside a stage) ----
    const int wn = wave / C::NWM, wm = wave % C::NWM;
    const int frow = lane & 15, fg = lane >> 4;
    const int foff = frow * C::ROWB + (swz<C::BK>(frow, fg) << 4);   // kk = 1 (BK=64): foff ^ 64
    const int woff = wn * C::WN * C::ROWB + foff;
    const int aoff = C::W_BYTES + wm * C::WM * C::ROWB + foff;

    f32x4 acc[C::TI][C::TJ];
#pragma unroll
    for (int i = 0; i < C::TI; ++i)
#pragma unroll
        for (int j = 0; j < C::TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / C::BK;
#pragma unroll
    for (int s = 0; s < C::STAGES - 1; ++s)
        if (s < nk) stage(s, s);
    int buf = 0, pbuf = C::STAGES - 1;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most STAGES-2 younger tiles are still in flight
        if (C::STAGES > 2 && kt + C::STAGES - 2 < nk) wait_vmcnt<C::G * (C::STAGES - 2)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // every wave's share of tile kt landed; everyone finished tile kt-1
        asm volatile("" ::: "memory");
        if (kt + C::STAGES - 1 < nk) stage(kt + C::STAGES - 1, pbuf);
        const char* sb = smem + buf * C::STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
            u32x4 wf[C::TI], af[C::TJ];
#pragma unroll
            for (int i = 0; i < C::TI; ++i) wf[i] = *(const u32x4*)(sb + ((woff + i * 16 * C::ROWB) ^ (kk << 6)));
#pragma unroll
            for (int j = 0; j < C::TJ; ++j) af[j] = *(const u32x4*)(sb + ((aoff + j * 16 * C::ROWB) ^ (kk << 6)));
#pragma unroll
            for (int i = 0; i < C::TI; ++i)
#pragma unroll
                for (int j = 0; j < C::TJ; ++j) acc[i][j] = mfma16<T>(wf[i], af[j], acc[i][j]);
        }
        buf = (buf + 1 == C::STAGES) ? 0 : buf + 1;
        pbuf = (pbuf + 1 == C::STAGES) ? 0 : pbuf + 1;
    }

    // ---- epilogue: lane owns 4 consecutive n (= nb..nb+3) of row m, per (i, j) ----
#pragma unroll
    for (int j = 0; j < C::TJ; ++j) {
        const int m = m0 + wm * C::WM + j * 16 + frow;
        if (m >= g.M) continue;
        size_t orow;
        if (EPI == EPI_DECONV) {
            const int t = m / g.Win, jj = m - t * g.Win, ii = t % g.Hin, img = t / g.Hin;
            orow = ((size_t)(img * 2 * g.Hin + 2 * ii + (parity >> 1)) * (2 * g.Win) + 2 * jj + (parity & 1)) * (size_t)g.ldo;
        } else if (EPI == EPI_HEATMAP) {
            const int img = m / 3072, p = m - img * 3072;
            orow = (size_t)img * g.Kp * 3072 + p;
        } else {
            orow = (size_t)m * g.ldo;
        }
#pragma unroll
        for (int i = 0; i < C::TI; ++i) {
            const int nb = n0 + wn * C::WN + i * 16 + fg * 4;
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

//                     BM   BN  BK  WM  WN  STAGES      LDS    waves
using Cfg0 = TileCfg<128, 128, 64, 64, 64, 2>;   //  64 KiB   4   (2 blocks / CU)
using Cfg1 = TileCfg<256, 128, 32, 128, 64, 3>;  //  72 KiB   4   (2 blocks / CU)
using Cfg2 = TileCfg<256, 256, 64, 128, 64, 2>;  // 128 KiB   8   (1 block / CU)
using Cfg3 = TileCfg<128, 128, 32, 64, 64, 4>;   //  64 KiB   4   (2 blocks / CU)
using Cfg4 = TileCfg<256, 128, 64, 64, 64, 2>;   //  96 KiB   8   (1 block / CU)
using Cfg5 = TileCfg<128, 256, 32, 64, 128, 3>;  //  72 KiB   4   (2 blocks / CU)

template <class T, int EPI, int AMODE, class C>
static hipError_t launch(const GemmArgs& a, hipStream_t s) {
    auto kern = gemm_kernel<T, EPI, AMODE, C>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    GemmArgs g = a;
    const int tiles_n = (a.N + C::BN - 1) / C::BN;
    g.w_parity_stride = (size_t)a.w_rows * a.K;
    if ((size_t)tiles_n * C::BN > (size_t)a.w_rows) return hipErrorInvalidValue;   // weight rows are padded at upload
    const int tiles = ((a.M + C::BM - 1) / C::BM) * tiles_n;
    dim3 grid(tiles, AMODE == A_DECONV ? 4 : 1);
    hipLaunchKernelGGL(kern, grid, dim3(C::NT), C::LDS, s, g);
    return hipGetLastError();
}

template <class T, int EPI, int AMODE>
static hipError_t by_variant(const GemmArgs& a, hipStream_t s) {
    switch (a.variant) {
        case 0: return launch<T, EPI, AMODE, Cfg0>(a, s);
        case 1: return launch<T, EPI, AMODE, Cfg1>(a, s);
        case 2: return launch<T, EPI, AMODE, Cfg2>(a, s);
        case 3: return launch<T, EPI, AMODE, Cfg3>(a, s);
        case 4: return launch<T, EPI, AMODE, Cfg4>(a, s);
        case 5: return launch<T, EPI, AMODE, Cfg5>(a, s);
    }
    return hipErrorInvalidValue;
}

template <class T>
static hipError_t dispatch(int epi, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS: return by_variant<T, EPI_BIAS, A_DENSE>(a, s);
        case EPI_BIAS_GELU: return by_variant<T, EPI_BIAS_GELU, A_DENSE>(a, s);
        case EPI_BIAS_RESID: return by_variant<T, EPI_BIAS_RESID, A_DENSE>(a, s);
        case EPI_POS: return by_variant<T, EPI_POS, A_DENSE>(a, s);
        case EPI_DECONV: return by_variant<T, EPI_DECONV, A_DECONV>(a, s);
        case EPI_HEATMAP: return by_variant<T, EPI_HEATMAP, A_DENSE>(a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t gemm_launch(int dtype, int epi, const GemmArgs& a, hipStream_t s) {
    if (a.K % 64 != 0 || a.M <= 0 || a.N <= 0) return hipErrorInvalidValue;
    if (epi == EPI_DECONV && (a.Cin % 64 != 0 || a.K != 4 * a.Cin)) return hipErrorInvalidValue;
    if (epi != EPI_HEATMAP && (a.N % 4 != 0 || a.ldo % 4 != 0)) return hipErrorInvalidValue;  // 8/16-byte epilogue stores
    return dtype == DT_F16 ? dispatch<F16>(epi, a, s) : dispatch<BF16>(epi, a, s);
}

}  // namespace vp
