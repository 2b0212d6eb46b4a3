// Shared pieces of the 8-phase GEMM kernels (gemm8.hip, gemm8f.hip, qkvattn.hip): tile geometry, wait / barrier helpers, tile walk.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace vp {
namespace {

template <int BN_, int BM_ = 256> struct G8 {
    static constexpr int BM = BM_, BN = BN_, NT = 512;
    static constexpr int XR = BM / 2;               // rows of an X half (128; 96 for the 192-row tile)
    static constexpr int MJ = XR / 32;              // m-fragments of a wave per X half (4 or 3)
    static constexpr int WH1 = BN - 128;            // rows of W half 1 (128 or 64)
    static constexpr int NF1 = WH1 / 64;            // n-fragments of a wave from W half 1 (2 or 1)
    static constexpr int TI = 2 + NF1, TJ = 2 * MJ; // fragments per wave: n, m
    static constexpr int HALF = 128 * 128;          // bytes of a 128-row slot (BK = 64 16-bit values per row)
    static constexpr int XH = XR * 128;             // bytes of an X slot
    static constexpr int OFF_X0 = 0, OFF_X1 = XH, OFF_W0 = 2 * XH, OFF_W1 = 2 * XH + HALF;
    static constexpr int BUF = 2 * XH + HALF + WH1 * 128;
    static constexpr int RING = 2 * BUF;
    static constexpr int NW1 = WH1 / 64;            // DMA instructions per wave for W half 1
    static constexpr int INFLIGHT = 4 + NW1;        // BM = 256: DMAs of the three youngest slots (W0, X0, W1) at the counted wait
    // BM = 192: an X half is 12 DMA pieces for 8 waves.  Waves 0-3 issue two pieces of X0 and one of X1, waves 4-7 one of X0 and two of
    // X1: every wave issues 5 + NW1 pieces per K-tile, the steady-state counted wait (NKEEP) is the same for all of them, and only the
    // two waits that leave exactly one LB group in flight (tile boundary, ring start) differ by wave group.
    static constexpr int NKEEP = (BM == 256 ? 6 : 5) + NW1;
    static_assert(BN == 256 || BN == 192, "BN");
    static_assert(BM == 256 || (BM == 192 && BN == 256), "BM");
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
        // workgroups on this XCD.  A launch of fewer than 256 tiles is one workgroup per tile, whatever the count: XCD x then holds
        // (G >> 3) + (x < (G & 7)) workgroups and owns exactly as many tiles -- nobody has a second tile.  (Rounding the grid down to a multiple
        // of 8 sent the last 1-7 tiles of e.g. a 252-tile launch into a second round: ViTPose-L fc2 at 63 crops 108 -> 147 us.)
        nloc = (gridDim.x >> 3) + (xcd < (int)(gridDim.x & 7) ? 1 : 0);
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
