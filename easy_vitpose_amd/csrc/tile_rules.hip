// Tile rules: which kernel configuration runs one GEMM of the path -- pure host functions of the shape (no device, no handle), walked over
// every batch size of every model by tests/test_host_logic.py through the host-only taps at the end of this file.
#include "api_internal.h"

using namespace vpi;

namespace vpi {

// Tile of the 8-phase kernel for an [M, N] output (wide = 16-bit output, else residual epilogue); variant 0 = the 2-phase kernels run it.
// A pure function of the shape: tests/test_host_logic.py walks it over every batch size through the host-only tap vp_dbg_gemm8_pick.
//
// A candidate QUALIFIES (round 3, measured in situ at batch 32 - 256: ViTPose-B qkv at 216 / 432 tiles -15 % / -5 %, fc2 at 216 tiles -23 %, but
// fc1 / fc2 at 288 tiles = 56 % full +20 %; ViTPose-H fc2 at batch 128, 480 tiles: 329 -> 279 us) from 1.75 tiles per CU (448), or from 192 tiles
// when its last round is >= 80 % full.  Round 4 (`extended`; profiles/tile_sweep_r4.txt: isolated sweep + in-situ A/B at 40 - 256 crops) adds, from
// 7 680 rows on: a launch of ONE round from 192 tiles (fc2 at 88 crops: 198 tiles of 256 x 256, 110 -> 87 us), and a candidate whose
// rounds x tile area is below the 2-phase kernel's rounds x work of a CU per round (two 192 x 128 workgroups per CU; one when <= 256 tiles) --
// fc2 at 172 crops: 387 tiles of 256 x 256 = 2 rounds against 3 rounds of everything else, 200 -> 173 us.  Among the qualifying candidates the
// cheapest rounds x area wins (192 x 256 priced x 1.08: measured 1 - 8 % behind 256 x 192 at equal rounds; ties: the larger tile); without
// `extended` the 192 x 256 tile is only the fallback when no 256-row tile qualifies.  Where isolated and in-situ timings disagreed (fc2 at 52 / 128
// crops, ViTPose-L at 40, -S at 256: the 2-phase kernel finds `hid` in the caches and wins by 2 - 7 % in situ) the rule follows the in-situ result.
G8Pick pick_gemm8_tile(int M, int N, bool wide, int bm192_mask, long min_tiles, bool extended) {
    struct Cand { int bm, bn, variant; };
    static const Cand cands[3] = {{256, 256, 16}, {256, 192, 17}, {192, 256, 18}};
    const bool ext = extended && M >= 7680;
    const long t2 = (long)((M + 191) / 192) * ((N + 127) / 128);
    const double cost2 = t2 <= 256 ? 24576.0 : (double)((t2 + 511) / 512) * 49152.0;
    G8Pick pk{0, 0, 0, 0};
    double best = 0.0;
    bool have256 = false;
    for (int i = 0; i < 3; ++i) {
        const Cand& cd = cands[i];
        if (M % cd.bm || N % cd.bn || (wide && cd.variant == 17)) continue;
        if (cd.variant == 18 && (!(bm192_mask & (wide ? 2 : 1)) || (!ext && have256))) continue;   // round-3 behaviour: only when no 256-row tile qualifies
        const long t = (long)(M / cd.bm) * (N / cd.bn);
        if (t < 8) continue;
        const long rounds = (t + 255) / 256;
        const double f = (double)t / (double)(rounds * 256);   // share of 256 CUs x rounds that computes a tile (below 256 tiles: one workgroup per tile)
        const double cost = (double)rounds * cd.bm * cd.bn * (cd.variant == 18 ? 1.08 : 1.0);
        bool q = t >= min_tiles || (f >= 0.8 && t >= 192);
        if (ext) q = q || (rounds == 1 && t >= 192) || cost < 0.95 * cost2;
        if (!q) continue;
        if (cd.bm == 256) have256 = true;
        if (!pk.variant || (ext ? cost < 0.98 * best : f > (double)pk.tiles / (double)((pk.tiles + 255) / 256 * 256) + 1e-9)) {
            pk = {cd.variant, cd.bm, cd.bn, t};
            best = cost;
        }
    }
    return pk;
}

// Tile configuration of the 2-phase kernel (gemm.hip Cfg id) for one GEMM of the path -- a pure function of the epilogue and the shape: tests/test_host_logic.py
// walks it over every batch size of every model through the host-only tap vp_dbg_gemm2_pick (slots, rounds, the PIPE-6 precondition, the measured choices).
//
// Default: the 192(m) x 128(n) tile -- M is always a multiple of 192 tokens (one crop per m-tile), so the tile count divides evenly over 256 CUs x 2 workgroups at
// the BASELINE batch; best or tied for every encoder GEMM in the MI355X sweep (profiles/gemm_tune_r1.txt); residual GEMMs: the same tile as 8 waves; wide GEMMs use
// the grouped order.  Small batches (fewer than 384 such tiles, e.g. 8 crops per GPU of a sharded frame): tiles that still give the 256 CUs a workgroup each --
// 128 x 128 from 256 tiles on, else 64 x 64, and inside the 64 x 64 regime (round 5, measured IN SITU: tools/small_sweep.py, profiles/small_batch_r5.txt):
//   64 x 64 tiles are bound by the latency of every k-block (a workgroup retires STAGES - 1 k-blocks per round trip) and, with one workgroup per SIMD set, by the
//   ~500 cycles of wait + barrier + LDS round trip in front of the 8 MFMAs of a k-step.  Inside the step every layer's weights are first touched from HBM, so the round
//   trip is ~2 x what the isolated sweeps of rounds 2-3 (weights L2-resident) saw.  Every choice keeps the k order: bit-identical.
//   * <= 256 tiles of 32 x 64: Cfg31 = 32(m) x 64(n) tiles, 6-stage ring, TWO k-blocks per barrier (gemm.hip PIPE 6) -- twice the workgroups, half the MFMAs per wave
//     and k-block;  <= 256 tiles of 64 x 64: Cfg30 = that schedule on 64 x 64 tiles, one workgroup per CU;
//   * <= 512 tiles (all resident at the 2 workgroups per CU of the 4-stage ring): Cfg12; more tiles would run the deep rings in two rounds and lose against the 5
//     workgroups per CU of the 2-stage ring (Cfg9) -- except for long K (round 2: Cfg12 from K = 2048 on);
//   * residual GEMMs (attn.proj, mlp.fc2) with more than 512 tiles of 64 x 64 but <= 512 of 128(m) x 64(n) (12-28 crops): Cfg15 = that tile on a 3-stage ring, all
//     resident at 2 workgroups per CU: fc2 of 16 crops 42 -> 34 us (-B), 56.5 -> 44 (-L), of 12 crops 72 -> 54 (-H).  For the wide GEMMs the same tile is neutral.
//   attn.proj of 1-8 crops 17-20 -> 10-13 us, mlp.fc2 of one crop 24.5 -> 17-21.5 us, qkv / fc1 of one crop 18 -> 12 us; ViTPose-L 1 crop 1.90 -> 1.36 ms, 8 crops
//   2.50 -> 2.40 ms, 16 crops 3.62 -> 3.23 ms; -B 1 crop 0.73 -> 0.56 ms, 16 crops 1.48 -> 1.33 ms; -H 1 crop 3.00 -> 2.17 ms, 12 crops 5.49 -> 4.81 ms.
Tile2Pick pick_gemm2_tile(int epi, int M, int N, int K) {
    Tile2Pick tp;
    tp.variant = (epi == vp::EPI_BIAS_RESID || epi == vp::EPI_BIAS_RESID_LN) ? 11 : 8;
    tp.group_m = (epi == vp::EPI_BIAS || epi == vp::EPI_BIAS_GELU) ? 8 : 0;
    const long par_ = (epi == vp::EPI_DECONV) ? 4 : 1;   // the four output parities of a deconv are four GEMMs of one launch
    const long t192 = (long)((M + 191) / 192) * ((N + 127) / 128) * par_;
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * par_;
    const bool wide = epi == vp::EPI_BIAS || epi == vp::EPI_BIAS_GELU;
    if (t192 >= 384) {
        // Round 6 (profiles/small_batch_r6.txt calls 18-21): more than 512 tiles of 192 x 128 are a second, mostly empty round of the 2 workgroups per CU.  Where the 8-phase
        // kernel has no tile for the row count (it takes the GEMM first, gemm()), every choice below keeps the k order (bit-identical):
        //   * <= 256 tiles of 256 x 256 (ragged last m-tile): ONE round of the 2-phase 256 x 256 tile (Cfg3).  Wide GEMMs: ViTPose-L 17-21 crops mlp.fc1 48.5 -> 38.5 us (step
        //     -9 ... -10.7 %), -B 22-27 crops -5 ... -7 %, -H 13-15 crops -5 ... -7.6 %, -S 43-55 crops -2 ... -3.6 %.  Residual GEMMs (57-85 crops of ViTPose-L, 86-113 of -B that are
        //     no multiple of 4): mlp.fc2 140 -> 116-139 us, step -2.3 ... -6.2 %; with 272 such tiles it loses (+20 %): the rule asks for one round;
        //   * else a wide GEMM whose 128 x 128 tiles still fit two rounds of the 512 slots (ViTPose-S attn.qkv at 57-75 crops: 25.0 -> 21-24 us, step -0.8 ... -3.1 %): 128 x 128.
        const bool enc = wide || epi == vp::EPI_BIAS_RESID_LN;
        if (enc && t192 > 512 && K >= 384 && K % 128 == 0 && N % 256 == 0 && (long)((M + 255) / 256) * (N / 256) <= 256) {
            tp.variant = 3;
            tp.group_m = wide ? 8 : 0;
        } else if (wide && t192 > 512 && t128 <= 1024) {
            tp.variant = 1;
            tp.group_m = 0;
        }
        return tp;
    }
    // Round 6: a wide GEMM whose 128 x 128 tiles need a second, mostly empty round (257-384 tiles on 256 CUs) while its 192 x 128 tiles are ONE round (<= 256: M is a
    // multiple of 192) runs on the 8-wave 192 x 128 tile with a 3-stage ring (Cfg20, one workgroup per CU) in groups of 8 m-tiles, m fastest (an XCD then owns a few
    // weight n-tiles x all crops).  K >= 1024 only: ViTPose-B's 12 k-blocks do not amortise the deeper prologue (measured +3 %, profiles/small_batch_r5.txt call 10).
    // ViTPose-L, 7-8 crops (one GPU's share of BASELINE configs[3]): qkv 25.6 -> 23.5 us, fc1 27.8 -> 25.5 us per layer; ViTPose-H at 8 crops: qkv only (fc1: 320 tiles).
    if ((epi == vp::EPI_BIAS || epi == vp::EPI_BIAS_GELU) && K >= 1024 && K % 128 == 0 && M % 192 == 0 && N % 128 == 0 && t128 > 256 && t192 <= 256) {
        tp.variant = 20;
        tp.group_m = 8;
        return tp;
    }
    // Round 6 (calls 18-19): 128 x 128 tiles beyond the 512 slots of 2 workgroups per CU while 192 x 128 tiles fit them: the default tile (ViTPose-L 11 crops fc1 37.9 -> 30.3 us,
    // -B 15 crops 31.8 -> 26.6, -S 40 crops qkv 20.8 -> 17.4; the residual GEMMs of ViTPose-L 43-47 / -B 58-63 / -H 35-38 crops: fc2 118 -> 92 / 95 -> 74 / 146 -> 114 us, step -11 %)
    if ((wide || epi == vp::EPI_BIAS_RESID_LN) && t128 > 512) return tp;
    const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64) * par_;
    const long t128x64 = (long)((M + 127) / 128) * ((N + 63) / 64) * par_;
    if (epi == vp::EPI_BIAS_RESID_LN && t64 > 512) {
        // residual GEMMs beyond the 512 resident 64 x 64 tiles (round 5: 128 x 64 on a 3-stage ring up to 512 tiles, then 128 x 128).  Round 6 (call 18, all bit-identical):
        //   * 96(m) x 64(n) tiles (Cfg41, 4-stage ring, 2 workgroups per CU) up to 448 of them: ViTPose-L 11-14 crops fc2 41-42 -> 36-38 us, proj 18.5 -> 16.4 (step -3.7 ... -4.9 %),
        //     -B 15-16 crops -4 ... -5.6 %; at 480-512 tiles it loses (-L 16 crops 44 -> 49 us, -B 20 crops +1.8 %);
        //   * beyond 512 tiles of 128 x 64 the 128 x 128 tile ran one workgroup per CU on a 2-stage ring (fc2 of ViTPose-L 21-32 crops flat at 66-72 us): the 8-wave 192 x 128 tile
        //     on a 3-stage ring (Cfg20), one round of <= 256 tiles, 60-64 us, proj 27-29.5 -> 25-27: ViTPose-L 24 / 28 / 32 crops -5.5 / -5.6 / -4.9 %, -B 29 / 32 / 34 crops -4.9 / -4.5 / -4.9 %.
        const long t96x64 = (long)((M + 95) / 96) * ((N + 63) / 64);
        tp.group_m = 0;
        tp.variant = t96x64 <= 448 ? 41 : t128x64 <= 512 ? 15 : (t192 <= 256 && K % 128 == 0) ? 20 : 1;
        return tp;
    }
    tp.variant = (t128 >= 256) ? 1 : 9;
    tp.group_m = 0;
    if (tp.variant == 9) {
        const long t32 = (long)((M + 31) / 32) * ((N + 63) / 64) * par_;
        if (K % 128 == 0 && t32 <= 256) tp.variant = 31;
        else if (K % 128 == 0 && t64 <= 256) tp.variant = 30;
        else if (t64 <= 512) tp.variant = 12;
        else if (epi == vp::EPI_BIAS_RESID_LN && t128x64 <= 512) tp.variant = 15;
        else if (K >= 2048) tp.variant = 12;
    }
    return tp;
}

// Split-K of a residual GEMM at small batches (round 6; in-situ grid profiles/small_batch_r6.txt: 4 models x 1-12 crops x S in {2, 4} x six tiles, whole step timed).
// A call of ONE OR TWO crops leaves most CUs idle in mlp.fc2 -- 72-96 tiles of 32 x 64 per crop, each a serial chain of K / 64 = 48-80 k-blocks: four k ranges per tile
// (288-384 workgroups of 12-20 k-blocks) + the fixed-order reduction kernel win although the partial products make a round trip through L2:
//   1 crop : ViTPose-B 0.575 -> 0.524 ms (-8.9 %), -L 1.376 -> 1.223 (-11.1 %), -H 2.188 -> 1.897 (-13.3 %), -S -1.9 %;   2 crops: -3.9 % / -6.0 % / -4.6 % (B / L / H).
// From 4 crops on the unsplit GEMM fills the chip and the round trip of the partials loses (+1 ... +20 %), with one exception that is shipped: ViTPose-H's mlp.fc2 of
// 7-8 crops (K = 5120 = 80 k-blocks on 480 tiles of 64 x 64) as 4 k ranges of 128 x 128 tiles: 4.435 -> 4.076 ms (-8.1 %).  attn.proj (K = D) never gains.
// Returns S = 1 for everything else; the caller requires K % (128 S) == 0.  A pure function of the shape: tests/test_host_logic.py walks it (vp_dbg_splitk_pick).
SplitKPick pick_splitk(int M, int N, int K) {
    if (K < 3 * N || K % 512 != 0) return {1, 0};         // mlp.fc2 only (K = 4 N)
    if (M <= 192) return {4, N >= 1280 ? 12 : 31};         // one crop: 32 x 64 tiles (64 x 64 on a 4-stage ring for ViTPose-H: measured -13.3 % against -10.5 %)
    if (M <= 384 && K >= 3072) return {4, 12};             // two crops (not ViTPose-S: neutral)
    if (K >= 5120 && M > 1152 && M <= 1536) return {4, 1};  // ViTPose-H, 7-8 crops
    return {1, 0};
}

// (BM, BN, workgroups resident on 256 CUs) of a 2-phase tile configuration the rule above can return
static bool tile2_dims(int variant, int& bm, int& bn, int& slots) {
    switch (variant) {
        case 8: case 11: bm = 192; bn = 128; slots = 512; return true;
        case 1: bm = 128; bn = 128; slots = 512; return true;
        case 3: bm = 256; bn = 256; slots = 256; return true;
        case 9: bm = 64; bn = 64; slots = 1280; return true;
        case 12: bm = 64; bn = 64; slots = 512; return true;
        case 15: bm = 128; bn = 64; slots = 512; return true;
        case 20: bm = 192; bn = 128; slots = 256; return true;
        case 30: bm = 64; bn = 64; slots = 256; return true;
        case 31: bm = 32; bn = 64; slots = 512; return true;
        case 41: bm = 96; bn = 64; slots = 512; return true;
    }
    return false;
}

// Rounds x tile area x K of one MLP GEMM at M rows under the rules above: the persistent 8-phase kernel runs ceil(tiles / 256) full rounds (its 192-row tile priced x 1.08 as in
// pick_gemm8_tile); a 2-phase launch ceil(tiles / resident slots) rounds, priced x 1.15 (measured: the 2-phase 256 x 256 tile 33 us against 30 for the same one-round launch on
// the 8-phase kernel, the default tile 110 against 80-90).
static double mlp_gemm_cost(int epi, int M, int N, int K, bool wide, int bm192_mask, bool extended, bool gemm8) {
    if (gemm8) {
        const G8Pick pk = pick_gemm8_tile(M, N, wide, bm192_mask, 448, extended);
        if (pk.variant) return (double)((pk.tiles + 255) / 256) * pk.bm * pk.bn * (pk.bm == 192 ? 1.08 : 1.0) * K;
    }
    int bm = 192, bn = 128, slots = 512;
    if (!tile2_dims(pick_gemm2_tile(epi, M, N, K).variant, bm, bn, slots)) return 0.0;
    const long t = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    return (double)((t + slots - 1) / slots) * bm * bn * (slots / 256) * 1.15 * K;
}

// The batch the ENCODER runs for a chunk of n crops (round 6, profiles/small_batch_r6.txt calls 20 + 22): the 8-phase kernel's 256-row tiles need a row count that is a multiple
// of 256 = a multiple of 4 crops, so a batch of 65 crops of ViTPose-L ran mlp.fc2 in 137 us on 516 2-phase tiles where 68 crops take 99 us -- the whole step 9.93 against 9.02 ms.
// From 33 crops on the encoder therefore runs the next multiple of 4 crops (the padding rows repeat the last crop: im2col_launch n_src; every kernel of the path works row by row
// or crop by crop, so the real crops' results are bit for bit those of the unpadded run: test_padded_encoder_batch_is_bit_identical) whenever the cost above of mlp.fc1 + mlp.fc2
// drops by more than 5 % -- checked against the measured step times of every batch size in calls 17-20: 108 of the 112 padded sizes gain (2-9 %), 4 lose 1.0-4.2 %.  The head
// and the decode run the real crops only.  A pure function of (n, D): tests/test_host_logic.py walks it through vp_dbg_run_batch.
int pick_run_batch(int n, int D, int limit, int bm192_mask, bool extended, int gemm8_mask) {
    const int n4 = (n + 3) / 4 * 4;
    if (n < 33 || n4 == n || n4 > limit) return n;
    auto cost = [&](int m) {
        return mlp_gemm_cost(vp::EPI_BIAS_GELU, 192 * m, 4 * D, D, true, bm192_mask, extended, (gemm8_mask & 2) != 0) +
               mlp_gemm_cost(vp::EPI_BIAS_RESID_LN, 192 * m, D, 4 * D, false, bm192_mask, extended, (gemm8_mask & 1) != 0);
    };
    const double c0 = cost(n), c1 = cost(n4);
    return (c0 > 0.0 && c1 > 0.0 && c1 < 0.95 * c0) ? n4 : n;
}

}  // namespace vpi

extern "C" {

// HOST ONLY: the batch the encoder runs for a chunk of n crops of a model of embed dim D (pick_run_batch with the default switches; limit = the handle's padded workspace batch)
VP_API int vp_dbg_run_batch(int32_t n, int32_t D, int32_t limit) {
    if (n <= 0 || D <= 0) return VP_ERR_INVALID;
    return pick_run_batch(n, D, limit, 3, true, 0x7);
}


// HOST ONLY: the 8-phase tile the selection rule of gemm() picks for an [M, N] output (wide: qkv / fc1; else the residual GEMMs); returns the
// variant (0 = none: 2-phase kernels, 16 = 256 x 256, 17 = 256 x 192, 18 = 192 x 256) and its tile count
VP_API int vp_dbg_gemm8_pick(int32_t M, int32_t N, int32_t wide, int32_t bm192_mask, int32_t* tiles) {
    if (M <= 0 || N <= 0) return VP_ERR_INVALID;
    const G8Pick pk = pick_gemm8_tile(M, N, wide != 0, bm192_mask & 3, 448, !(bm192_mask & 4));
    if (tiles) *tiles = (int32_t)pk.tiles;
    return pk.variant;
}

// HOST ONLY: the tile configuration (gemm.hip Cfg id) the 2-phase selection rule picks for one GEMM: epi = kernels.h GemmEpi (0 bias, 1 bias + GELU, 4 deconv, 5 heatmap,
// 6 residual + statistics, 7 pos + statistics), shape [M, N] x K; *group_m = its tile-order group
VP_API int vp_dbg_gemm2_pick(int32_t epi, int32_t M, int32_t N, int32_t K, int32_t* group_m) {
    if (M <= 0 || N <= 0 || K <= 0) return VP_ERR_INVALID;
    const Tile2Pick tp = pick_gemm2_tile(epi, M, N, K);
    if (group_m) *group_m = tp.group_m;
    return tp.variant;
}

// HOST ONLY: the split-K rule for a residual GEMM of [M, N] x K: returns S (1 = one launch), *variant = the tile configuration of the partial products
VP_API int vp_dbg_splitk_pick(int32_t M, int32_t N, int32_t K, int32_t* variant) {
    if (M <= 0 || N <= 0 || K <= 0) return VP_ERR_INVALID;
    const SplitKPick sk = pick_splitk(M, N, K);
    if (variant) *variant = sk.variant;
    return sk.S;
}

}  // extern "C"
