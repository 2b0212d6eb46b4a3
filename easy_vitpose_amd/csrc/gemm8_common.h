// Shared pieces of the 8-phase GEMM kernels (gemm8.hip, gemm8d.hip): tile geometry, wait / barrier helpers, tile walk.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace vp {
namespace {

template <int BN_> struct G8 {
    static constexpr int BM = 256, BN = BN_, NT = 512;
    static constexpr int WH1 = BN - 128;            // rows of W half 1 (128 or 64)
    static constexpr int NF1 = WH1 / 64;            // n-fragments of a wave from W half 1 (2 or 1)
    static constexpr int TI = 2 + NF1, TJ = 8;      // fragments per wave: n, m
    static constexpr int HALF = 128 * 128;          // bytes of a 128-row slot (BK = 64 16-bit values per row)
    static constexpr int OFF_X0 = 0, OFF_X1 = HALF, OFF_W0 = 2 * HALF, OFF_W1 = 3 * HALF;
    static constexpr int BUF = 3 * HALF + WH1 * 128;
    static constexpr int RING = 2 * BUF;
    static constexpr int NW1 = WH1 / 64;            // DMA instructions per wave for W half 1
    static constexpr int INFLIGHT = 4 + NW1;        // DMAs of the three youngest slots (W0, X0, W1) at the counted wait
    static_assert(BN == 256 || BN == 192, "BN");
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void bar() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}


__device__ __forceinline__ float row8_sum8(float x) {   // identical to gemm.hip's row8_sum
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
    return x;
}

struct TileWalk {   // XCD-contiguous, grouped tile order (same as gemm.hip's persistent kernel)
    int tiles_m, tiles_n, group_m, base, cnt, j0, nloc;
    __device__ __forceinline__ void init(const GemmArgs& g, int BM, int BN) {
        tiles_n = (g.N + BN - 1) / BN;
        tiles_m = (g.M + BM - 1) / BM;
        group_m = g.group_m;
        const int ntiles = tiles_m * tiles_n;
        const int xcd = blockIdx.x & 7;
        j0 = blockIdx.x >> 3;
        nloc = gridDim.x >> 3;
        const int q = ntiles >> 3, r8 = ntiles & 7;
        base = (xcd < r8) ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
        cnt = q + (xcd < r8 ? 1 : 0);
    }
    __device__ __forceinline__ void origin(int t, int reverse, int BM, int BN, int& m0, int& n0) const {
        int bid = base + t;
        if (reverse) bid = tiles_m * tiles_n - 1 - bid;
        int tm, tn;
        if (group_m > 1) {
            const int per_group = group_m * tiles_n;
            const int grp = bid / per_group, first_m = grp * group_m;
            const int gsz = min(tiles_m - first_m, group_m);
            const int r = bid - grp * per_group;
            tm = first_m + r % gsz;
            tn = r / gsz;
        } else {
            tm = bid / tiles_n;
            tn = bid - tm * tiles_n;
        }
        m0 = tm * BM;
        n0 = tn * BN;
    }
};

}  // namespace
}  // namespace vp
