// C ABI (include/vitpose_hip.h) of the MI355X-native ViTPose hot path:
// context + weight packer + forward orchestration.  No CPU fallback anywhere:
// every compute entry point needs a HIP device and fails with VP_ERR_HIP otherwise.
#include "api_internal.h"

using namespace vpi;

#ifndef VP_GRAPH_NULL_DEFAULT
#define VP_GRAPH_NULL_DEFAULT 1
#endif

namespace vpi {

thread_local std::string g_create_error;

int fail(vp_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

bool prof_begin(vp_ctx* c, int fam, double flops, double bytes) {
    if (!((c->prof >> fam) & 1u)) return false;
    std::pair<hipEvent_t, hipEvent_t> p;
    if (!c->ev_pool.empty()) {
        p = c->ev_pool.back();
        c->ev_pool.pop_back();
    } else {
        hipEventCreate(&p.first);
        hipEventCreate(&p.second);
    }
    hipEventRecord(p.first, c->stream);
    c->evs.push_back({p.first, p.second, fam, flops, bytes});
    return true;
}
void prof_end(vp_ctx* c, bool on) {
    if (on) hipEventRecord(c->evs.back().b, c->stream);
}
void prof_collect(vp_ctx* c) {
    for (auto& e : c->evs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            c->acc.ms[e.fam] += ms;
            c->acc.flops[e.fam] += e.flops;
            c->acc.bytes[e.fam] += e.bytes;
            c->acc.launches[e.fam] += 1;
        }
        c->ev_pool.push_back({e.a, e.b});
    }
    c->evs.clear();
}

// Tile configuration per GEMM family.  Defaults = best measured on MI355X (profiles/);
// experiments override with VP_GEMM_TUNE="fam:variant:group_m,..." (fam = VP_PROF_* index).
void apply_gemm_tuning(vp_ctx* c) {
#ifndef VP_TOOLS
    (void)c;
#else
    if (const char* t = getenv("VP_GEMM_TUNE")) {
        int f, v, gm, used = 0;
        while (sscanf(t, "%d:%d:%d%n", &f, &v, &gm, &used) == 3) {
            if (f >= 0 && f < VP_PROF_COUNT) { c->gemm_variant[f] = v; c->gemm_group_m[f] = gm; }
            t += used;
            if (*t == ',') ++t; else break;
        }
    }
#endif
}

int gemm(vp_ctx* c, int fam, int epi, const uint16_t* A, const uint16_t* W, const float* bias, void* out,
         const float* aux, int M, int N, int K, int ldo, int Hin, int Win, int Cin, const LnFuse* ln) {
    vp::GemmArgs g{};
    g.A = A; g.W = W; g.bias = bias; g.out = out; g.aux = aux;
    g.M = M; g.N = N; g.K = K; g.ldo = ldo;
    g.Hin = Hin; g.Win = Win; g.Cin = Cin; g.zero = c->zero; g.Kp = c->Kp;
    g.w_rows = (int)pad128((size_t)N);
    g.variant = c->gemm_variant[fam];
    g.group_m = c->gemm_group_m[fam];
    g.ablate = c->gemm_ablate | c->fam_ablate[fam];
    g.parity_fast = c->deconv_parity_fast;
    if (g.variant < 0) {
        const Tile2Pick tp = pick_gemm2_tile(epi, M, N, K);   // the 2-phase kernels' tile (the 8-phase kernel may take the GEMM over below)
        g.variant = tp.variant; g.group_m = tp.group_m;
    }
    if (c->persist_gemm && g.variant == 8 && (epi == vp::EPI_BIAS || epi == vp::EPI_BIAS_GELU) && K % 128 == 0 && ldo == N &&
        M % 192 == 0 && N % 128 == 0 && (long)(M / 192) * (N / 128) >= 1024)   // >= 2 tiles per resident workgroup
        g.persist = 1;
    if (ln) {
        g.a_blocked = ln->a_blocked; g.out_blocked = ln->out_blocked; g.reverse = ln->reverse;
        g.plane = ln->plane; g.stats_out = ln->stats_out; g.rowstat = ln->rowstat; g.ln_s = ln->ln_s;
        g.ln_part = ln->ln_part; g.ln_tiles = ln->ln_tiles; g.ln_inv_d = 1.0f / (float)K;
        if (ln->tiles_out) *ln->tiles_out = N / 64;   // partial statistics are written per 64 columns, whatever the tile
    }
    // large batches: the 8-phase persistent kernel (gemm8.hip), one 512-thread workgroup per CU on 256 x 256 (wide GEMMs) or
    // 256 x 192 (N = D: 768 = 4 x 192, three full rounds of 256 workgroups at batch 256) tiles, when every CU gets >= 1.75 tiles
    // (attn.proj, K = N = D, is HBM-bound and stays on the 192 x 128 tile with two workgroups per CU: measured 105 vs 112 us;
    // bit 3 of VP_GEMM8 moves it too)
    const bool is_proj = epi == vp::EPI_BIAS_RESID_LN && K <= N;
    const int g8bit = fam == VP_PROF_GEMM_FC2 ? 1 : fam == VP_PROF_GEMM_FC1 ? 2 : fam == VP_PROF_GEMM_QKV ? 4 : fam == VP_PROF_GEMM_PROJ ? 8 : 0;
    (void)is_proj;
    if (c->gemm_variant[fam] < 0 && (c->gemm8_mask & g8bit) &&
        (epi == vp::EPI_BIAS || epi == vp::EPI_BIAS_GELU || epi == vp::EPI_BIAS_RESID_LN)) {
        const bool wide = epi != vp::EPI_BIAS_RESID_LN;
        // tile shape: pick_gemm8_tile above (256 x 256; residual GEMMs also 256 x 192; 192 x 256 where the row count or the rounds ask for it)
#ifdef VP_TOOLS
        static const long min_tiles = [] { const char* e = getenv("VP_G8_MIN_TILES"); return e ? atol(e) : 448L; }();
#else
        const long min_tiles = 448;
#endif
        const G8Pick pk = pick_gemm8_tile(M, N, wide, c->g8_bm192, min_tiles, c->g8_cost_model);
        if (pk.variant && vp::gemm8_supported(epi, g, pk.bn, pk.bm)) {
            g.variant = pk.variant;
            g.group_m = fam == VP_PROF_GEMM_QKV ? 4 : fam == VP_PROF_GEMM_FC2 ? 2 : 8;   // measured sweep 0 / 2 / 4 / 8 / 16 / 32 (spread 2-3 %)
            g.persist = 0;
            g.stagger = c->g8_stagger;
        }
    }
    if (epi == vp::EPI_DECONV_FINAL) {   // deconv2 + final 1x1 conv in one kernel: the 256 x 256 tile (all channels of a pixel)
        g.variant = 3; g.group_m = 0; g.persist = 0;
        g.W2 = c->w_fin; g.bias2 = c->b_fin; g.out2 = c->hm;
    }
    if (g.ln_part) {   // only the one-tile-per-workgroup 2-phase kernel folds partial statistics itself
        g.persist = 0;
        if (g.variant >= 16 && g.variant <= 18) { g.variant = 8; g.group_m = 8; }
    }
    const bool deconv = epi == vp::EPI_DECONV || epi == vp::EPI_DECONV_FINAL;
    const double par = deconv ? 4.0 : 1.0;
    const double Nalg = (epi == vp::EPI_HEATMAP) ? (double)c->Kp : (double)N;   // heatmap: N counts the hi + lo weight rows
    double flops = 2.0 * M * Nalg * K * par;
    // algorithmic HBM bytes: each operand once, output once (+ residual read)
    const bool resid = epi == vp::EPI_BIAS_RESID || epi == vp::EPI_BIAS_RESID_LN;
    const bool f32out = resid || epi == vp::EPI_POS || epi == vp::EPI_POS_LN || epi == vp::EPI_HEATMAP;
    const double out_b = f32out ? 4.0 : 2.0;
    double bytes = 2.0 * M * (double)(deconv ? Cin : K) + 2.0 * N * (double)K * par + out_b * M * Nalg * par;
    if (epi == vp::EPI_DECONV_FINAL) {   // the 256-channel activations never reach HBM; the final conv's flops and the fp32 heatmaps count
        flops += 2.0 * M * par * (double)c->Kp * N;
        bytes += (4.0 * c->Kp - out_b * N) * M * par + 2.0 * c->fin_rows * N;
    }
    if (resid) bytes += 4.0 * M * (double)N;
    if (epi == vp::EPI_BIAS_RESID_LN || epi == vp::EPI_POS_LN) bytes += 8.0 * M * (double)(N / 64);   // partial row statistics
    char desc[192];
    desc[0] = 0;
    g.desc = desc; g.desc_cap = (int)sizeof(desc);   // the launch code names the kernel it resolved to (one snprintf per GEMM launch: vp_profile_kernel reports the LAST launch)
    // small batches: a residual GEMM as S partial products over k ranges + a fixed-order reduction (tile_rules.hip pick_splitk)
    if (epi == vp::EPI_BIAS_RESID_LN && c->splitk_ws && (size_t)M <= c->splitk_rows && c->gemm_variant[fam] < 0 && !(g.variant >= 16 && g.variant <= 18) &&
        out == (void*)aux && (fam == VP_PROF_GEMM_FC2 || fam == VP_PROF_GEMM_PROJ)) {
        const int which = fam == VP_PROF_GEMM_FC2 ? 1 : 0;
        SplitKPick sk = pick_splitk(M, N, K);
        if (c->splitk_force[which][0] > 0) sk = {c->splitk_force[which][0], c->splitk_force[which][1]};
        if (sk.S > 1 && sk.S <= SPLITK_MAX_S && K % (sk.S * 128) == 0) {
            vp::GemmArgs p = g;
            p.out = c->splitk_ws; p.aux = nullptr; p.bias = nullptr; p.stats_out = nullptr; p.plane = 0;
            p.variant = sk.variant; p.group_m = 0; p.splitk = sk.S; p.ldo = N; p.persist = 0;
            LAUNCH(c, fam, flops, bytes, vp::gemm_launch(c->dtype, vp::EPI_PARTIAL, p, c->stream));   // (flops / bytes stay the algorithmic figures of the GEMM; the partials' round trip is overhead)
            if (desc[0]) {
                char d2[224];
                snprintf(d2, sizeof(d2), "%s x split-K %d + splitk_reduce_kernel", desc, sk.S);
                if (c->kernel_desc[fam] != d2) c->kernel_desc[fam] = d2;
            }
            LAUNCH(c, fam, 0.0, 0.0, vp::splitk_reduce_launch(c->dtype, c->splitk_ws, sk.S, bias, (uint16_t*)out, g.plane, g.stats_out, M, N, c->stream));
            return VP_OK;
        }
    }
    LAUNCH(c, fam, flops, bytes, vp::gemm_launch(c->dtype, epi, g, c->stream));
    if (desc[0] && c->kernel_desc[fam] != desc) c->kernel_desc[fam] = desc;
    return VP_OK;
}

// fp8 mode: one encoder GEMM on MXFP8 operands (gemm8f.hip).  epi: EPI_BIAS (qkv -> 16-bit), EPI_BIAS_GELU (fc1 -> MXFP8 codes at `out`,
// block scales at out_scales), EPI_BIAS_RESID_LN (fc2: two-plane residual + row statistics).  M = padded token rows (multiple of 256).
int gemm_fp8(vp_ctx* c, int fam, int epi, const uint8_t* A8, const uint8_t* a_scales, const uint8_t* W8, const float* w_scale, const float* bias,
             void* out, uint8_t* out_scales, const float* aux, int M, int N, int K, const LnFuse* ln) {
    vp::GemmArgs g{};
    g.A = (const uint16_t*)A8; g.W = (const uint16_t*)W8; g.bias = bias; g.out = out; g.aux = aux;
    g.M = M; g.N = N; g.K = K; g.ldo = N;
    g.a_scales = a_scales; g.w_scale = w_scale; g.out_scales = out_scales;
    g.w_rows = (int)pad128((size_t)N);
    if (ln) {
        g.out_blocked = ln->out_blocked; g.reverse = ln->reverse;
        g.plane = ln->plane; g.stats_out = ln->stats_out;
        if (ln->tiles_out) *ln->tiles_out = N / 64;
    }
    // tile width: 256 for the wide GEMMs; the residual GEMM takes the width whose tile count fills the rounds of 256 persistent workgroups best
    int bn = 256;
    if (epi == vp::EPI_BIAS_RESID_LN) {
        double fill = -1.0;
        for (int cand : {256, 192}) {
            if (N % cand) continue;
            const long t = (long)(M / 256) * (N / cand);
            if (t < 8) continue;
            const double f = (double)t / (double)((t + 255) / 256 * 256);
            if (f > fill + 1e-9) { bn = cand; fill = f; }
        }
    }
    g.group_m = fam == VP_PROF_GEMM_QKV ? 4 : fam == VP_PROF_GEMM_FC2 ? 2 : 8;
    if (!vp::gemm8f_supported(epi, g, bn))
        return fail(c, VP_ERR_SHAPE, "fp8 mode: GEMM shape " + std::to_string(M) + " x " + std::to_string(N) + " x " + std::to_string(K) + " not supported by the MXFP8 kernel");
    const double flops = 2.0 * M * (double)N * K;
    double bytes = 1.0 * M * (double)K + M * (double)(K / 32) + 1.0 * N * (double)K;      // codes + block scales + weight codes
    bytes += epi == vp::EPI_BIAS ? 2.0 * M * (double)N : epi == vp::EPI_BIAS_GELU ? (1.0 + 1.0 / 32) * M * (double)N : 8.0 * M * (double)N + 8.0 * M * (double)(N / 64);
    char desc[192];
    desc[0] = 0;
    g.desc = desc; g.desc_cap = (int)sizeof(desc);
    LAUNCH(c, fam, flops, bytes, vp::gemm8f_launch(epi, g, bn, c->stream));
    if (desc[0] && c->kernel_desc[fam] != desc) c->kernel_desc[fam] = desc;
    return VP_OK;
}

}  // namespace vpi

namespace {

// forward of one chunk (n <= max_batch) with device-resident crops; heatmaps land in c->hm
int forward_chunk(vp_ctx* c, const void* d_crops, int fmt, int n_in, bool want_tokens, bool flip = false) {
    const int D = c->D;
    // the ENCODER's batch: n_in crops, or the next multiple of 4 where that buys the MLP GEMMs an 8-phase tile (tile_rules.hip pick_run_batch; rows n_in .. n - 1 repeat the
    // last crop and are never read by the head); the fp8 mode pads its rows itself
    const int n = (c->pad_batch && c->fuse_ln && !c->fp8) ? pick_run_batch(n_in, D, (c->maxb + 3) / 4 * 4, c->g8_bm192, c->g8_cost_model, c->gemm8_mask) : n_in;
    const int M = n * 192;
    const double in_b = (fmt == VP_INPUT_F32_NCHW ? 4.0 : 1.0) * n_in * 3.0 * 256 * 192;
    LAUNCH(c, VP_PROF_IM2COL, 0.0, in_b + 2.0 * M * 768, vp::im2col_launch(c->dtype, d_crops, fmt, c->hid, n, c->stream, flip, n_in));
    int rc;
    size_t plane = 0;   // != 0: the residual stream c->x is held as two 16-bit planes (fused-LayerNorm path)
    const bool qkv_blocked = c->blocked_qkv && D / c->heads == 64;
    if (c->fp8) {
        // fp8 mode.  The residual stream, the attention core, attn.proj, the head and the decode are the fp16 path's; qkv / fc1 / fc2 run
        // on MXFP8 operands.  Token rows are padded to Mp (multiple of the 256-row tile, >= 512): padding rows carry zeros into the
        // GEMMs and are never read by a kernel that works per crop.
        const int Mp = (int)std::max<size_t>((size_t)(M + 255) / 256 * 256, 512);
        plane = c->Mp * (size_t)D;                   // the planes are laid out for the handle's full padded row count
        uint16_t* xh = (uint16_t*)c->x;
        int tiles = 0;
        LnFuse prod; prod.plane = plane; prod.stats_out = c->ln_part; prod.tiles_out = &tiles;
        auto quant = [&]() -> int {                  // LayerNorm(x) -> MXFP8 (replaces ln_finalize; gamma / beta live in the consumer's weights / bias)
            LAUNCH(c, VP_PROF_LAYERNORM, 0.0, 8.0 * M * tiles + 2.0 * M * D + (1.0 + 1.0 / 32) * Mp * (double)D,
                   vp::ln_quant_launch(c->dtype, xh, c->ln_part, tiles, c->x8, c->xs8, M, Mp, D, c->stream));
            return VP_OK;
        };
        if ((rc = gemm(c, VP_PROF_GEMM_PATCH, vp::EPI_POS_LN, c->hid, c->w_patch, c->b_zero, c->x, c->pos, M, D, 768, D, 0, 0, 0, &prod))) return rc;
        for (int l = 0; l < c->L; ++l) {
            const Block& b = c->blocks[l];
            if ((rc = quant())) return rc;
            LnFuse cq; cq.out_blocked = qkv_blocked;
            if ((rc = gemm_fp8(c, VP_PROF_GEMM_QKV, vp::EPI_BIAS, c->x8, c->xs8, b.w_qkv8, b.ws_qkv, b.b_qkv, c->qkv, nullptr, nullptr, Mp, 3 * D, D, &cq))) return rc;
            LnFuse pp = prod;
            if (c->y8) {   // head dim 64: attention writes MXFP8, attn.proj runs on the fp8 kernel too
                LAUNCH(c, VP_PROF_ATTN, 4.0 * 192 * 192 * (double)D * n, 7.0 * M * D,
                       vp::attention_launch(c->dtype, c->qkv, (uint16_t*)c->y8, n, D, c->heads, c->stream, qkv_blocked ? 1 : 0, c->ys8));
                if ((rc = gemm_fp8(c, VP_PROF_GEMM_PROJ, vp::EPI_BIAS_RESID_LN, c->y8, c->ys8, b.w_proj8, b.ws_proj, b.b_proj, c->x, nullptr, c->x, Mp, D, D, &pp))) return rc;
            } else {
                LAUNCH(c, VP_PROF_ATTN, 4.0 * 192 * 192 * (double)D * n, 8.0 * M * D,
                       vp::attention_launch(c->dtype, c->qkv, c->y, n, D, c->heads, c->stream, qkv_blocked ? 1 : 0));
                if ((rc = gemm(c, VP_PROF_GEMM_PROJ, vp::EPI_BIAS_RESID_LN, c->y, b.w_proj, b.b_proj, c->x, c->x, M, D, D, D, 0, 0, 0, &pp))) return rc;
            }
            if ((rc = quant())) return rc;
            if ((rc = gemm_fp8(c, VP_PROF_GEMM_FC1, vp::EPI_BIAS_GELU, c->x8, c->xs8, b.w_fc18, b.ws_fc1, b.b_fc1, c->hid, c->hs8, nullptr, Mp, 4 * D, D, nullptr))) return rc;
            LnFuse p2 = prod;
            if ((rc = gemm_fp8(c, VP_PROF_GEMM_FC2, vp::EPI_BIAS_RESID_LN, (const uint8_t*)c->hid, c->hs8, b.w_fc28, b.ws_fc2, b.b_fc2, c->x, nullptr, c->x, Mp, D, 4 * D, &p2))) return rc;
        }
    } else if (c->fuse_ln) {
        // LayerNorm folded into the GEMMs on both sides of it.  The residual stream is kept as two 16-bit planes
        // (x = hi + lo, same bytes as fp32, >= 22 significant bits): every producer (patch embed, attn.proj,
        // mlp.fc2) writes the planes + partial row statistics, a tiny kernel folds those into (mean, rstd), and
        // qkv / fc1 multiply the hi plane -- the UN-normalised rows -- by gamma-folded weights and normalise in
        // their epilogue.  Saves the 151 MB read + 75 MB write of 24 of the 25 LayerNorm passes.
        plane = (size_t)M * D;
        uint16_t* xh = (uint16_t*)c->x;
        int tiles = 0;
        LnFuse prod; prod.plane = plane; prod.stats_out = c->ln_part; prod.tiles_out = &tiles;
        // small batches: the consumers fold the partial statistics themselves (same code, same bits) -- 2 x depth launches less
        // (round 6: with the consumers' merge on a register copy of the row's statistics instead of a bank-conflicted LDS image -- gemm.hip, GemmArgs::ln_part -- the fold wins at
        // every model and batch up to 8 crops, the one-round 192 x 128 tiles of ViTPose-L included: 8 crops 2.200 -> 2.143 ms against the ln_finalize launches, 4 crops 1.874 ->
        // 1.800, 1 crop 1.172 -> 1.111; beyond 8 crops it still loses (every column tile merges its rows again; ViTPose-H's fused qkv + attention tile needs rowstat):
        // profiles/small_batch_r6.txt call 11)
        const bool fold_stats = n <= c->graph_max_n_stats;
        // Round 6 (profiles/small_batch_r6.txt call 25): beyond 8 crops the fold did not lose because of the merge but because the (mean, rstd) area behind the ring pushes the
        // 80 KiB ring of the default 192 x 128 tile over half the CU's LDS -- ONE workgroup per CU instead of two (+9 ... +12 % per step).  Per consumer (attn.qkv reads LayerNorm-1,
        // mlp.fc1 LayerNorm-2): fold where its GEMM runs on a 2-phase tile that keeps its occupancy with the area (every configuration but the 80 KiB-ring ones), i.e. not on the
        // 8-phase kernel, not in a fused qkv + attention kernel (they read rowstat): ViTPose-S 9-28 crops -6.5 ... -8 %, -B 9-14 -2 ... -6 %, -L 9-10 -2.3 %; same code, same bits.
        auto folds = [&](int fam, int epi, int N, int g8bit) {
            if (fold_stats) return true;
            if (!c->fold_rule || n > 64 || c->gemm_variant[fam] >= 0) return false;
            if ((c->gemm8_mask & g8bit) && pick_gemm8_tile(M, N, true, c->g8_bm192, 448, c->g8_cost_model).variant) return false;
            const int v = pick_gemm2_tile(epi, M, N, D).variant;
            return v != 8 && v != 11;
        };
        // attn.qkv + attention core as ONE kernel per (pair of crops, head) from 108 tiles on (qkvattn.hip; bit-identical y; an odd batch's last crop fills both halves of its pair)
        // 128 - 1536 tiles: profiles/qkvattn_r4.txt.  Below (round 6, profiles/small_batch_r6.txt call 16): 108-120 tiles win or tie (ViTPose-B 17-20 crops -0.6 ... -6.5 %: at 19-20
        // crops the unfused qkv is 540 tiles of 128 x 128 on 512 slots; ViTPose-L 13-14 crops equal); 96 tiles and fewer lose (-B 16 crops +-0, 12 crops +3 %, -L 9-12 crops +2 ... +7 %)
        //   Call 26: with the per-consumer statistics fold (below) the two-launch path saves its LayerNorm-1 ln_finalize launches wherever the qkv GEMM's tile folds, and wins back
        //   108-127 tiles there (ViTPose-B 17-18 crops -1.8 / -2.2 %, -L 13-14 crops -2.9 / -3.4 %); where that GEMM would run on the default tile (-B 19-20) the fused kernel keeps them.
        static const bool qa_min_set = getenv("VP_QA_MIN_TILES") != nullptr;
        static const long qa_min_tiles = [] { const char* e = getenv("VP_QA_MIN_TILES"); return e ? atol(e) : 108L; }();
        // head dim 80: one crop x one head per 192 x 256 tile of the 8-phase kernel (gemm8.hip EPI_QKV_ATTN; bit-identical y), from 192 tiles on
        static const long qa80_min_tiles = [] { const char* e = getenv("VP_QA80_MIN_TILES"); return e ? atol(e) : 192L; }();   // wins from 12 crops x 16 heads on (profiles/qkvattn80_r5.txt)
        const bool has_qkvh = c->L > 0 && c->blocks[0].w_qkvh != nullptr;
        const bool qkv_can_fold = folds(VP_PROF_GEMM_QKV, vp::EPI_BIAS, 3 * D, 4);
        const long pair_tiles = (long)((n + 1) / 2) * c->heads;
        const bool want80 = has_qkvh && c->heads * 80 == D && (long)n * c->heads >= qa80_min_tiles;
        const bool want64 = has_qkvh && c->heads * 64 == D && pair_tiles >= qa_min_tiles && (qa_min_set || pair_tiles >= 128 || !qkv_can_fold);
        const bool fold1 = fold_stats || (!want80 && !want64 && qkv_can_fold);   // LayerNorm-1 -> attn.qkv
        const bool fold2 = folds(VP_PROF_GEMM_FC1, vp::EPI_BIAS_GELU, 4 * D, 2);                                       // LayerNorm-2 -> mlp.fc1
        auto finalize = [&](bool folded) -> int {
            if (folded) return VP_OK;
            LAUNCH(c, VP_PROF_LAYERNORM, 0.0, 8.0 * M * tiles + 8.0 * M, vp::ln_finalize_launch(c->ln_part, c->rowstat, M, tiles, D, c->stream));
            return VP_OK;
        };
        if ((rc = gemm(c, VP_PROF_GEMM_PATCH, vp::EPI_POS_LN, c->hid, c->w_patch, c->b_zero, c->x, c->pos, M, D, 768, D, 0, 0, 0, &prod))) return rc;
        if ((rc = finalize(fold1))) return rc;
        for (int l = 0; l < c->L; ++l) {
            const Block& b = c->blocks[l];
            LnFuse cq; cq.rowstat = c->rowstat; cq.ln_s = b.s_qkv; cq.reverse = (c->order_mask & 1) != 0; cq.out_blocked = qkv_blocked;
            if (fold1) { cq.rowstat = nullptr; cq.ln_part = c->ln_part; cq.ln_tiles = D / 64; }
            vp::QkvAttnArgs qa{};
            qa.x_hi = xh; qa.wh = b.w_qkvh; qa.bh = b.b_qkvh; qa.sh = b.s_qkvh; qa.rowstat = c->rowstat; qa.y = c->y;
            qa.npairs = (n + 1) / 2; qa.ncrops = n; qa.heads = c->heads; qa.D = D;   // odd n: the last crop fills both halves of its pair
            qa.scale_log2e = (1.0f / sqrtf(64.0f)) * 1.4426950408889634f;
            // (a shape the fused kernel rejects -- a chunk beyond its 32-bit row offsets, fewer than 8 tiles under a lowered VP_QA_MIN_TILES -- falls through
            // to the gemm + attention pair below, the way gemm() falls back when gemm8_supported says no: ADVICE r4)
            vp::GemmArgs g80{};
            if (b.w_qkvh && c->heads * 80 == D) {
                g80.A = xh; g80.W = b.w_qkvh; g80.bias = b.b_qkvh; g80.ln_s = b.s_qkvh; g80.rowstat = c->rowstat; g80.out = c->y;
                g80.M = M; g80.N = c->heads * 256; g80.K = D; g80.ldo = D; g80.w_rows = c->heads * 256; g80.variant = 18;
                // tile order: groups of 8 crops x all heads, crop fastest -- the 32 workgroups of an XCD then work on 8 crops x 4 heads at a time (12 operand K-slices
                // fetched per K-tile for 32 tiles instead of 18 with the head-fastest order: PMC traffic 800 -> ~450 MB per launch, profiles/qkvattn80_r5.txt)
                static const int qa80_group = [] { const char* e = getenv("VP_QA80_GROUP"); return e ? atoi(e) : 8; }();
                g80.group_m = qa80_group;
                g80.attn_scale_log2e = (1.0f / sqrtf(80.0f)) * 1.4426950408889634f;
                g80.ablate = c->gemm_ablate | c->fam_ablate[VP_PROF_GEMM_QKV];
            }
            if (g80.A && !fold1 && (long)n * c->heads >= qa80_min_tiles && c->gemm_variant[VP_PROF_GEMM_QKV] < 0 && vp::gemm8_supported(vp::EPI_QKV_ATTN, g80, 256, 192)) {
                char desc[192];
                desc[0] = 0;
                g80.desc = desc; g80.desc_cap = (int)sizeof(desc);
                LAUNCH(c, VP_PROF_GEMM_QKV, 2.0 * M * 3.0 * D * D + 4.0 * 192 * 192 * (double)D * n, 2.0 * M * D + 2.0 * 3 * D * (double)D + 2.0 * M * D,
                       vp::gemm_launch(c->dtype, vp::EPI_QKV_ATTN, g80, c->stream));
                if (desc[0] && c->kernel_desc[VP_PROF_GEMM_QKV] != desc) c->kernel_desc[VP_PROF_GEMM_QKV] = desc;
            } else if (want64 && b.w_qkvh && !fold1 && c->gemm_variant[VP_PROF_GEMM_QKV] < 0 && vp::qkvattn_supported(qa)) {
                char desc[96];
                desc[0] = 0;
                LAUNCH(c, VP_PROF_GEMM_QKV, 2.0 * M * 3.0 * D * D + 4.0 * 192 * 192 * (double)D * n, 2.0 * M * D + 2.0 * 3 * D * (double)D + 2.0 * M * D,
                       vp::qkvattn_launch(c->dtype, qa, c->stream, desc, (int)sizeof(desc)));
                if (desc[0] && c->kernel_desc[VP_PROF_GEMM_QKV] != desc) c->kernel_desc[VP_PROF_GEMM_QKV] = desc;
            } else {
            if ((rc = gemm(c, VP_PROF_GEMM_QKV, vp::EPI_BIAS, xh, b.w_qkv, b.b_qkv, c->qkv, nullptr, M, 3 * D, D, 3 * D, 0, 0, 0, &cq))) return rc;
            LAUNCH(c, VP_PROF_ATTN, 4.0 * 192 * 192 * (double)D * n, 8.0 * M * D,
                   vp::attention_launch(c->dtype, c->qkv, c->y, n, D, c->heads, c->stream, qkv_blocked ? 1 : 0));
            }
            LnFuse pp = prod; pp.reverse = (c->order_mask & 2) != 0;
            if ((rc = gemm(c, VP_PROF_GEMM_PROJ, vp::EPI_BIAS_RESID_LN, c->y, b.w_proj, b.b_proj, c->x, c->x, M, D, D, D, 0, 0, 0, &pp))) return rc;
            if ((rc = finalize(fold2))) return rc;
            LnFuse c1; c1.rowstat = c->rowstat; c1.ln_s = b.s_fc1; c1.out_blocked = c->blocked_hid; c1.reverse = (c->order_mask & 4) != 0;
            if (fold2) { c1.rowstat = nullptr; c1.ln_part = c->ln_part; c1.ln_tiles = D / 64; }
            if ((rc = gemm(c, VP_PROF_GEMM_FC1, vp::EPI_BIAS_GELU, xh, b.w_fc1, b.b_fc1, c->hid, nullptr, M, 4 * D, D, 4 * D, 0, 0, 0, &c1))) return rc;
            LnFuse p2 = prod; p2.a_blocked = c->blocked_hid; p2.reverse = (c->order_mask & 8) != 0;
            if ((rc = gemm(c, VP_PROF_GEMM_FC2, vp::EPI_BIAS_RESID_LN, c->hid, b.w_fc2, b.b_fc2, c->x, c->x, M, D, 4 * D, D, 0, 0, 0, &p2))) return rc;
            if (l + 1 < c->L && (rc = finalize(fold1))) return rc;   // last block: last_norm below is a standalone pass
        }
    } else {
    if ((rc = gemm(c, VP_PROF_GEMM_PATCH, vp::EPI_POS, c->hid, c->w_patch, c->b_zero, c->x, c->pos, M, D, 768, D))) return rc;
    for (int l = 0; l < c->L; ++l) {
        const Block& b = c->blocks[l];
        LAUNCH(c, VP_PROF_LAYERNORM, 0.0, 6.0 * M * D,
               vp::layernorm_launch(c->dtype, c->x, b.ln1_g, b.ln1_b, c->y, nullptr, M, D, c->stream));
        LnFuse q0; q0.out_blocked = qkv_blocked;
        if ((rc = gemm(c, VP_PROF_GEMM_QKV, vp::EPI_BIAS, c->y, b.w_qkv, b.b_qkv, c->qkv, nullptr, M, 3 * D, D, 3 * D, 0, 0, 0, &q0))) return rc;
        LAUNCH(c, VP_PROF_ATTN, 4.0 * 192 * 192 * (double)D * n, 8.0 * M * D,
               vp::attention_launch(c->dtype, c->qkv, c->y, n, D, c->heads, c->stream, qkv_blocked ? 1 : 0));
        if ((rc = gemm(c, VP_PROF_GEMM_PROJ, vp::EPI_BIAS_RESID, c->y, b.w_proj, b.b_proj, c->x, c->x, M, D, D, D))) return rc;
        LAUNCH(c, VP_PROF_LAYERNORM, 0.0, 6.0 * M * D,
               vp::layernorm_launch(c->dtype, c->x, b.ln2_g, b.ln2_b, c->y, nullptr, M, D, c->stream));
        if ((rc = gemm(c, VP_PROF_GEMM_FC1, vp::EPI_BIAS_GELU, c->y, b.w_fc1, b.b_fc1, c->hid, nullptr, M, 4 * D, D, 4 * D))) return rc;
        if ((rc = gemm(c, VP_PROF_GEMM_FC2, vp::EPI_BIAS_RESID, c->hid, b.w_fc2, b.b_fc2, c->x, c->x, M, D, 4 * D, D))) return rc;
    }
    }
    // last_norm, head and decode: the caller's n_in crops (the planes are laid out for the encoder's row count)
    const int nh = n_in, Mh = nh * 192;
    LAUNCH(c, VP_PROF_LAYERNORM, 0.0, 6.0 * Mh * D,
           vp::layernorm_launch(c->dtype, c->x, c->lnf_g, c->lnf_b, c->y, want_tokens ? c->tok : nullptr, Mh, D, c->stream, plane));
    // head: tokens [n,16,12,D] (NHWC view of [n*192, D]) -> [n,32,24,256] -> [n,64,48,256] -> heatmaps [n,Kp,64,48]
    if ((rc = gemm(c, VP_PROF_GEMM_DECONV, vp::EPI_DECONV, c->y, c->w_d1, c->b_d1, c->d1, nullptr, nh * 192, 256, 4 * D, 256, 16, 12, D))) return rc;
    // large batches: the final 1x1 conv rides in deconv2's epilogue (gemm.hip EPI_DECONV_FINAL, bit-identical heatmaps) and the
    // [n,64,48,256] tensor is never written; small batches keep the two launches on tiles that still fill 256 CUs
    if (c->fuse_head && c->gemm_variant[VP_PROF_GEMM_DECONV] < 0 && (long)nh * 12 >= 512)
        return gemm(c, VP_PROF_GEMM_DECONV, vp::EPI_DECONV_FINAL, c->d1, c->w_d2, c->b_d2, nullptr, nullptr, nh * 768, 256, 1024, 256, 32, 24, 256);
    if ((rc = gemm(c, VP_PROF_GEMM_DECONV, vp::EPI_DECONV, c->d1, c->w_d2, c->b_d2, c->d2, nullptr, nh * 768, 256, 1024, 256, 32, 24, 256))) return rc;
    if ((rc = gemm(c, VP_PROF_GEMM_FINAL, vp::EPI_HEATMAP, c->d2, c->w_fin, c->b_fin, c->hm, nullptr, nh * 3072, (int)c->fin_rows, 256, 0))) return rc;
    return VP_OK;
}

int decode_chunk(vp_ctx* c, const int32_t* d_wh, float* d_out, int n) {
    LAUNCH(c, VP_PROF_DECODE, 0.0, 4.0 * n * c->Kp * 3072.0 + 12.0 * n * c->Kp,
           vp::decode_launch(c->hm, d_wh, d_out, n, c->Kp, c->stream));
    return VP_OK;
}

// forward + decode of one chunk.  Small chunks (<= graph_max_n crops, profiling off) are launch-bound -- ~110-290 launches of a few
// microseconds each -- so the second time the same (n, format, buffers) combination is seen the chunk is captured into a hipGraph
// and from then on replayed with one hipGraphLaunch.
int run_chunk(vp_ctx* c, const void* d_src, int fmt, int nb, const int32_t* d_wh, float* d_out) {
    int rc;
    auto eager = [&]() -> int {
        if ((rc = forward_chunk(c, d_src, fmt, nb, false))) return rc;
        return decode_chunk(c, d_wh, d_out, nb);
    };
    // a caller's legacy default stream (torch's usual current stream): VP_GRAPH_NULL=1 replays the graph there too (a caller that synchronises after
    // every call pays the 110-290 eager launches' host time); 0 = plain launches (equal when calls are enqueued back to back)
    static const bool graph_null = [] { const char* e = getenv("VP_GRAPH_NULL"); return e ? atoi(e) != 0 : VP_GRAPH_NULL_DEFAULT; }();
    if (nb > c->graph_max_n || c->prof != 0 || (c->stream == nullptr && !graph_null)) return eager();
    vp_ctx::GraphEntry* ge = nullptr;
    for (auto& g : c->graphs)
        if (g.n == nb && g.fmt == fmt && g.src == d_src && g.wh == d_wh && g.out == d_out) { ge = &g; break; }
    if (ge && ge->exec) {
        const hipError_t el = hipGraphLaunch(ge->exec, c->stream);
        if (el == hipSuccess) return VP_OK;
        (void)hipGetLastError();   // a stream that does not take graph launches: nothing was enqueued -- this key runs eagerly from now on
        hipGraphExecDestroy(ge->exec);
        ge->exec = nullptr;
        ge->no_graph = true;
        return eager();
    }
    if (ge && ge->no_graph) return eager();   // capture or instantiation failed once for this key: it stays on the eager path
    if (!ge) {   // first sighting: run eagerly (also performs every one-time function-attribute set-up outside a capture), remember the key
        ge = &c->graphs[c->graph_victim++ & 3];
        if (ge->exec) {   // eviction: the graph may still be executing on the stream (vp_infer_device with sync = 0)
            HIPCHK(c, hipStreamSynchronize(c->stream));
            hipGraphExecDestroy(ge->exec);
            ge->exec = nullptr;
        }
        ge->n = nb; ge->fmt = fmt; ge->src = d_src; ge->wh = d_wh; ge->out = d_out; ge->seen = 1; ge->no_graph = false;
        return eager();
    }
    // second sighting: capture.  Any failure of the capture machinery (not of the launches themselves) marks the key "do not
    // graph" and the chunk runs eagerly now and from now on -- a handle never gets stuck retrying a capture.
    hipGraph_t graph = nullptr;
    hipStream_t target = c->stream;   // the capture itself always runs on the handle's own stream (nothing executes during a capture); the graph is launched on `target`
    c->stream = c->own_stream;
    hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { c->stream = target; (void)hipGetLastError(); ge->no_graph = true; return eager(); }
    rc = forward_chunk(c, d_src, fmt, nb, false);
    if (!rc) rc = decode_chunk(c, d_wh, d_out, nb);
    e = hipStreamEndCapture(c->stream, &graph);
    c->stream = target;
    if (rc) {   // a launch failed INSIDE the capture (e.g. a capture-illegal call): nothing has executed -- drop the graph, clear the sticky
                // error and run the chunk eagerly, now and from now on; an error is reported only if the eager run fails too
        if (graph) hipGraphDestroy(graph);
        (void)hipGetLastError();
        ge->exec = nullptr;
        ge->no_graph = true;
        c->err.clear();
        return eager();
    }
    if (e == hipSuccess && graph) {
        e = hipGraphInstantiate(&ge->exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
    } else if (e == hipSuccess) {
        e = hipErrorUnknown;
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ge->exec = nullptr;
        ge->no_graph = true;
        return eager();
    }
    HIPCHK(c, hipGraphLaunch(ge->exec, c->stream));
    return VP_OK;
}

size_t crop_bytes(int fmt) { return (size_t)3 * 256 * 192 * (fmt == VP_INPUT_F32_NCHW ? 4 : 1); }

// Work on the handle's buffers is about to be enqueued on stream s: order it behind whatever the previous call left on ANOTHER stream.
//   own -> own, caller A -> caller A: nothing (in-order streams);  own -> caller: record on own, caller waits;  caller -> own / another caller: wait for ev_sw,
//   which vp_infer_device_stream recorded on the caller's stream behind its launches (that stream itself is never touched again: it may be gone).
int adopt_stream(vp_ctx* c, hipStream_t s) {
    if (!c->foreign_pending && s == c->own_stream) return VP_OK;
    if (c->foreign_pending && (const void*)s == c->last_stream_id && s != c->own_stream) return VP_OK;
    if (!c->ev_sw) HIPCHK(c, hipEventCreateWithFlags(&c->ev_sw, hipEventDisableTiming));
    if (!c->foreign_pending) HIPCHK(c, hipEventRecord(c->ev_sw, c->own_stream));
    HIPCHK(c, hipStreamWaitEvent(s, c->ev_sw, 0));
    if (s == c->own_stream) c->foreign_pending = false;
    return VP_OK;
}

int check_ready(vp_ctx* c, int fmt, int n, const void* p0, const void* p1, bool adopt_own = true) {
    if (!c) return VP_ERR_INVALID;
    if (!c->loaded) return fail(c, VP_ERR_STATE, "weights not loaded: call vp_load_weights first");
    if (fmt != VP_INPUT_F32_NCHW && fmt != VP_INPUT_U8_NHWC) return fail(c, VP_ERR_INVALID, "unknown input_format");
    if (n < 0 || (n > 0 && (!p0 || !p1))) return fail(c, VP_ERR_INVALID, "null buffer or negative batch");
    hipError_t e = hipSetDevice(c->cfg.device_id);
    if (e != hipSuccess) return fail(c, VP_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return adopt_own ? adopt_stream(c, c->own_stream) : VP_OK;   // every entry but the caller-stream path works on the handle's own stream
}

}  // namespace

extern "C" {

int vp_abi_version(void) { return VP_ABI_VERSION; }
int vp_group_destroy(vp_group_handle g);

int vp_create(vp_handle* out, const vp_config* cfg) {
    if (!out || !cfg) return fail(nullptr, VP_ERR_INVALID, "null argument");
    *out = nullptr;
    const int D = cfg->embed_dim, h = cfg->num_heads;
    if (D <= 0 || h <= 0 || D % h != 0 || D % 128 != 0 || D > 1280)
        return fail(nullptr, VP_ERR_INVALID, "embed_dim must be a multiple of 128 (<= 1280) and divisible by num_heads");
    const int hd = D / h;
    if (hd != 32 && hd != 64 && hd != 80) return fail(nullptr, VP_ERR_INVALID, "head_dim must be 32, 64 or 80");
    if (cfg->depth <= 0 || cfg->num_keypoints <= 0 || cfg->num_keypoints > 1024 || cfg->max_batch <= 0)
        return fail(nullptr, VP_ERR_INVALID, "depth, num_keypoints and max_batch must be positive");
    if (cfg->dtype != VP_DTYPE_F16 && cfg->dtype != VP_DTYPE_BF16 && cfg->dtype != VP_DTYPE_FP8) return fail(nullptr, VP_ERR_INVALID, "unknown dtype");
    // fp8 mode: the MXFP8 kernel has no small-tile fallback -- at the minimum padded row count (512) the narrowest GEMM (attn.proj / mlp.fc2,
    // N = D) must still have the 8 tiles gemm8f_supported asks for: 2 x D / 192 (or / 256) >= 8 -> D >= 768.  Rejected HERE, not at infer time.
    if (cfg->dtype == VP_DTYPE_FP8 && (D % 256 != 0 || D < 768))
        return fail(nullptr, VP_ERR_INVALID, "the fp8 mode needs embed_dim >= 768 and a multiple of 256 (K-tiles of 128, an even number of them; >= 8 tiles per GEMM at the smallest batch): ViTPose-B / -L / -H");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, VP_ERR_HIP, std::string("no HIP device available (this library has no CPU fallback): ") +
                                             (e != hipSuccess ? hipGetErrorString(e) : "device count 0"));
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, VP_ERR_INVALID, "device_id out of range");
    vp_ctx* c = new vp_ctx();
    c->cfg = *cfg;
    c->D = D; c->L = cfg->depth; c->heads = h; c->Kp = cfg->num_keypoints;
    c->dtype = cfg->dtype == VP_DTYPE_BF16 ? vp::DT_BF16 : vp::DT_F16;   // fp8 mode: everything that is not one of the three MXFP8 GEMMs runs as fp16
    c->fp8 = cfg->dtype == VP_DTYPE_FP8;
    c->maxb = cfg->max_batch;
    apply_gemm_tuning(c);
    auto bail = [&](int rc) { g_create_error = c->err; vp_destroy(c); return rc; };
    if ((e = hipSetDevice(cfg->device_id)) != hipSuccess) { c->err = hipGetErrorString(e); return bail(VP_ERR_HIP); }
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { c->err = hipGetErrorString(e); return bail(VP_ERR_HIP); }
    c->own_stream = c->stream;
    if (const char* f = getenv("VP_CALLER_STREAM")) c->caller_stream_max_n = atoi(f) == 1 ? 16 : atoi(f);   // 0 = always fence against the caller's stream with events
    const size_t B = (size_t)((c->maxb + 3) / 4 * 4);   // workspaces: the encoder may run the next multiple of 4 crops (forward_chunk)
    // fp8 mode: workspaces indexed by token row are sized for the padded row count the MXFP8 GEMM tiles need
    c->Mp = std::max<size_t>((B * 192 + 255) / 256 * 256, 512);
    const size_t M = c->fp8 ? c->Mp : B * 192;
    int rc = 0;
    void* stage = nullptr;
    if ((rc = dalloc(c, (char**)&stage, B * crop_bytes(VP_INPUT_F32_NCHW)))) return bail(rc);
    c->in_stage = stage;
    if ((rc = dalloc(c, &c->wh_stage, B * 2))) return bail(rc);
    if ((rc = dalloc(c, &c->x, M * D))) return bail(rc);
    if ((rc = dalloc(c, &c->y, M * D))) return bail(rc);
    // switches between SHIPPED, tested code paths (each has a parity test that flips it): standalone LayerNorm passes, one-tile-per-
    // workgroup wide GEMMs, eager small batches, un-fused head
    if (const char* f = getenv("VP_FUSE_LN")) c->fuse_ln = atoi(f) != 0;
    if (const char* f = getenv("VP_PERSIST")) c->persist_gemm = atoi(f) != 0;
    if (const char* f = getenv("VP_GRAPH")) c->graph_max_n = atoi(f) == 1 ? 16 : atoi(f);   // 0 = off, 1 = default, n > 1: capture chunks of up to n crops
    if (const char* f = getenv("VP_FUSE_HEAD")) c->fuse_head = atoi(f) != 0;
    if (const char* f = getenv("VP_FOLD_STATS")) { c->graph_max_n_stats = atoi(f); c->fold_rule = false; }   // an explicit threshold: fold up to that many crops, nowhere else
    if (const char* f = getenv("VP_BLOCKED_QKV")) c->blocked_qkv = atoi(f) != 0;
    if (const char* f = getenv("VP_FUSE_QKV_ATTN")) c->fuse_qkv_attn = atoi(f) != 0;
    if (const char* f = getenv("VP_DECONV_PARITY_FAST")) c->deconv_parity_fast = atoi(f) != 0;
    if (const char* f = getenv("VP_G8_COST")) c->g8_cost_model = atoi(f) != 0;
    if (const char* f = getenv("VP_PAD_BATCH")) c->pad_batch = atoi(f) != 0;
    if (const char* f = getenv("VP_G8_BM192")) c->g8_bm192 = atoi(f);   // mask: 1 = residual GEMMs, 2 = wide GEMMs may take the 192 x 256 tile of the 8-phase kernel (0: never)
    if (const char* f = getenv("VP_SPLITK")) {   // 0 = off; "fc2:S:variant,proj:S:variant" = override the rule (measurement sweeps)
        if (!strchr(f, ':')) c->splitk_on = atoi(f) != 0;
        else {
            const char* t = f;
            while (*t) {
                const int which = !strncmp(t, "fc2:", 4) ? 1 : !strncmp(t, "proj:", 5) ? 0 : -1;
                if (which < 0) break;
                t = strchr(t, ':') + 1;
                int S = 0, v = 0, used = 0;
                if (sscanf(t, "%d:%d%n", &S, &v, &used) != 2) break;
                c->splitk_force[which][0] = S; c->splitk_force[which][1] = v;
                t += used;
                if (*t == ',') ++t;
            }
        }
    }
    if (const char* f = getenv("VP_GEMM8")) c->gemm8_mask = atoi(f);   // which GEMMs may take the 8-phase kernel (1 fc2, 2 fc1, 4 qkv, 8 proj; 0 = the 2-phase kernels everywhere)
#ifdef VP_TOOLS   // development switches of the measurement build (tools/)
    if (const char* f = getenv("VP_BLOCKED_HID")) c->blocked_hid = atoi(f) != 0;
    if (const char* f = getenv("VP_ORDER")) c->order_mask = atoi(f);
    if (const char* f = getenv("VP_G8_STAGGER")) c->g8_stagger = atoi(f);
    if (const char* t = getenv("VP_ABLATE_FAM")) {   // e.g. "2:64,1:64" = non-temporal stores in the qkv and fc1 epilogues
        int f, b, used = 0;
        while (sscanf(t, "%d:%d%n", &f, &b, &used) == 2) {
            if (f >= 0 && f < VP_PROF_COUNT) c->fam_ablate[f] = b;
            t += used;
            if (*t == ',') ++t; else break;
        }
    }
#endif
    if (c->fp8 && !c->fuse_ln) { c->err = "the fp8 mode is built on the fused-LayerNorm path (unset VP_FUSE_LN)"; return bail(VP_ERR_INVALID); }
    if (c->fuse_ln) {
        if ((rc = dalloc(c, &c->ln_part, M * (size_t)(D / 64) * 2))) return bail(rc);
        if ((rc = dalloc(c, &c->rowstat, M * 2))) return bail(rc);
    }
    if (c->fp8) {
        if ((rc = dalloc(c, &c->x8, M * D)) || (rc = dalloc(c, &c->xs8, M * (size_t)(D / 32))) || (rc = dalloc(c, &c->hs8, M * (size_t)(4 * D / 32)))) return bail(rc);
        if (hipMemset(c->x, 0, M * D * 4) != hipSuccess) { c->err = "hipMemset"; return bail(VP_ERR_HIP); }   // padding rows of the planes: read by fc2's residual epilogue
        if (hd == 64 && !getenv("VP_FP8_PROJ16")) {   // VP_FP8_PROJ16=1: attn.proj stays on the fp16 kernels (parity test flips it)
            if ((rc = dalloc(c, &c->y8, M * D)) || (rc = dalloc(c, &c->ys8, M * (size_t)(D / 32)))) return bail(rc);
            if (hipMemset(c->y8, 0, M * D) != hipSuccess || hipMemset(c->ys8, 0, M * (size_t)(D / 32)) != hipSuccess) { c->err = "hipMemset"; return bail(VP_ERR_HIP); }   // padding rows: zero codes
        }
    }
    if ((rc = dalloc(c, &c->qkv, M * 3 * D))) return bail(rc);
    if ((rc = dalloc(c, &c->hid, M * 4 * D))) return bail(rc);
    if ((rc = dalloc(c, &c->d1, B * 768 * 256))) return bail(rc);
    if ((rc = dalloc(c, &c->d2, B * 3072 * 256))) return bail(rc);
    if ((rc = dalloc(c, &c->hm, B * c->Kp * 3072))) return bail(rc);
    if ((rc = dalloc(c, &c->kp, B * c->Kp * 3))) return bail(rc);
    if (c->fuse_ln && !c->fp8 && c->splitk_on) {
        c->splitk_rows = (size_t)std::min<int>(c->maxb, SPLITK_MAX_CROPS) * 192;
        if ((rc = dalloc(c, &c->splitk_ws, (size_t)SPLITK_MAX_S * c->splitk_rows * D))) return bail(rc);
    }
    if ((rc = dalloc(c, &c->zero, (size_t)256))) return bail(rc);
    if (hipMemset(c->zero, 0, 512) != hipSuccess) { c->err = "hipMemset"; return bail(VP_ERR_HIP); }
    *out = c;
    return VP_OK;
}
int vp_infer_device(vp_handle c, const void* d_crops, int32_t fmt, int32_t n, const int32_t* d_org_wh, float* d_out, int32_t sync) {
    int rc = check_ready(c, fmt, n, d_crops, d_out);
    if (rc) return rc;
    for (int off = 0; off < n; off += c->maxb) {
        const int nb = (n - off < c->maxb) ? n - off : c->maxb;
        const char* src = (const char*)d_crops + (size_t)off * crop_bytes(fmt);
        if ((rc = run_chunk(c, src, fmt, nb, d_org_wh ? d_org_wh + 2 * (size_t)off : nullptr, d_out + (size_t)off * c->Kp * 3))) return rc;
    }
    if (sync) HIPCHK(c, hipStreamSynchronize(c->stream));
    return VP_OK;
}

int vp_infer_device_stream(vp_handle c, const void* d_crops, int32_t fmt, int32_t n, const int32_t* d_org_wh, float* d_out, void* caller_stream) {
    int rc = check_ready(c, fmt, n, d_crops, d_out, false);
    if (rc) return rc;
    hipStream_t cs = (hipStream_t)caller_stream;
    if (n > 0 && n <= c->caller_stream_max_n && n <= c->maxb && cs != c->own_stream) {
        // Small batches: the chunk's launches (or its hipGraph) go onto the caller's stream itself -- in order with its producers and consumers by construction, no
        // cross-stream dependency per call (two of them cost ~0.1 ms of a 0.6-2.4 ms step: profiles/small_batch_r5.txt).  ev_sw, recorded behind the launches, is
        // what the next call on any other stream (and vp_synchronize / vp_destroy) waits for.
        if ((rc = adopt_stream(c, cs))) return rc;
        c->stream = cs;
        rc = run_chunk(c, d_crops, fmt, n, d_org_wh, d_out);
        c->stream = c->own_stream;
        if (!c->ev_sw && hipEventCreateWithFlags(&c->ev_sw, hipEventDisableTiming) != hipSuccess) return fail(c, VP_ERR_HIP, "hipEventCreateWithFlags(ev_sw)");
        const hipError_t er = hipEventRecord(c->ev_sw, cs);   // also when rc != 0: part of the chunk may be enqueued
        c->foreign_pending = true;
        c->last_stream_id = (const void*)cs;
        if (rc) return rc;
        if (er != hipSuccess) return fail(c, VP_ERR_HIP, std::string("hipEventRecord(ev_sw): ") + hipGetErrorString(er));
        return VP_OK;
    }
    if ((rc = adopt_stream(c, c->own_stream))) return rc;
    if (!c->ev_in) HIPCHK(c, hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    if (!c->ev_out) HIPCHK(c, hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
    // everything the caller enqueued on its stream so far (the producers of d_crops / d_org_wh) happens before the library's kernels ...
    HIPCHK(c, hipEventRecord(c->ev_in, cs));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_in, 0));
    if ((rc = vp_infer_device(c, d_crops, fmt, n, d_org_wh, d_out, 0))) return rc;
    // ... and whatever the caller enqueues afterwards (consumers of d_out) happens after them
    HIPCHK(c, hipEventRecord(c->ev_out, c->stream));
    HIPCHK(c, hipStreamWaitEvent(cs, c->ev_out, 0));
    return VP_OK;
}

int vp_infer(vp_handle c, const void* crops, int32_t fmt, int32_t n, const int32_t* org_wh, float* out) {
    int rc = check_ready(c, fmt, n, crops, out);
    if (rc) return rc;
    if (n > c->maxb && !c->slots[0].busy && !c->slots[1].busy) {
        // more than one chunk: through the two asynchronous slots, so that the upload of chunk i+1 and the download of chunk i-1
        // run under the compute of chunk i (fully overlapped when the caller's buffers are pinned; pageable buffers still work,
        // the runtime stages them synchronously)
        int pending[2], np = 0;
        for (int off = 0; off < n; off += c->maxb) {
            const int nb = (n - off < c->maxb) ? n - off : c->maxb;
            int32_t slot = -1;
            rc = vp_infer_submit(c, (const char*)crops + (size_t)off * crop_bytes(fmt), fmt, nb, org_wh ? org_wh + 2 * (size_t)off : nullptr,
                                 out + (size_t)off * c->Kp * 3, &slot);
            if (rc) { for (int i = 0; i < np; ++i) vp_infer_wait(c, pending[i]); return rc; }
            pending[np++] = slot;
            if (np == 2) {
                if ((rc = vp_infer_wait(c, pending[0]))) { vp_infer_wait(c, pending[1]); return rc; }
                pending[0] = pending[1]; np = 1;
            }
        }
        for (int i = 0; i < np; ++i)
            if ((rc = vp_infer_wait(c, pending[i]))) return rc;
        return VP_OK;
    }
    for (int off = 0; off < n; off += c->maxb) {
        const int nb = (n - off < c->maxb) ? n - off : c->maxb;
        const char* src = (const char*)crops + (size_t)off * crop_bytes(fmt);
        HIPCHK(c, hipMemcpyAsync(c->in_stage, src, (size_t)nb * crop_bytes(fmt), hipMemcpyHostToDevice, c->stream));
        if (org_wh) HIPCHK(c, hipMemcpyAsync(c->wh_stage, org_wh + 2 * (size_t)off, (size_t)nb * 8, hipMemcpyHostToDevice, c->stream));
        if ((rc = run_chunk(c, c->in_stage, fmt, nb, org_wh ? c->wh_stage : nullptr, c->kp))) return rc;
        HIPCHK(c, hipMemcpyAsync(out + (size_t)off * c->Kp * 3, c->kp, (size_t)nb * c->Kp * 12, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return VP_OK;
}

// ---- asynchronous host path ----------------------------------------------------------------------------------------------
void* vp_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void vp_host_free(void* p) { if (p) hipHostFree(p); }

// stage_out: the D2H targets the slot's own pinned buffer and vp_infer_wait copies it to `out` (the group path: the caller's
// memory may be pageable, and an asynchronous copy to pageable memory is host-synchronous -- it would hold the submission until the
// compute is over and serialise the devices of a group)
// true when `p` is page-locked host memory the runtime knows (hipHostMalloc / vp_host_alloc / hipHostRegister): only such memory is
// copied asynchronously as it is
static bool host_ptr_is_pinned(const void* p) {
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // unknown to the runtime = pageable
    return a.type == hipMemoryTypeHost;
}

static int submit_impl(vp_handle c, const void* crops, int32_t fmt, int32_t n, const int32_t* org_wh, float* out, int32_t* slot_out, bool stage_out) {
    int rc = check_ready(c, fmt, n, crops, out);
    if (rc) return rc;
    if (!slot_out) return fail(c, VP_ERR_INVALID, "null slot pointer");
    if (n <= 0 || n > c->maxb) return fail(c, VP_ERR_INVALID, "vp_infer_submit takes 1 .. max_batch crops per call");
    if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->d2h_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
    const int si = c->next_slot;
    vp_ctx::Slot& sl = c->slots[si];
    if (sl.busy) return fail(c, VP_ERR_STATE, "both slots in flight: call vp_infer_wait first");
    if (!sl.in) {
        char* q;
        if ((rc = dalloc(c, &q, (size_t)c->maxb * crop_bytes(VP_INPUT_F32_NCHW)))) return rc;
        sl.in = q;
    }
    if (!sl.wh && (rc = dalloc(c, &sl.wh, (size_t)c->maxb * 2))) return rc;
    if (!sl.kp && (rc = dalloc(c, &sl.kp, (size_t)c->maxb * c->Kp * 3))) return rc;
    if (!sl.h2d) HIPCHK(c, hipEventCreateWithFlags(&sl.h2d, hipEventDisableTiming));
    if (!sl.done) HIPCHK(c, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (!sl.out) HIPCHK(c, hipEventCreateWithFlags(&sl.out, hipEventDisableTiming));
    // copy stream: H2D of this call (the slot's previous D2H finished: vp_infer_wait was called on it)
    const size_t in_bytes = (size_t)n * crop_bytes(fmt);
    if (stage_out && !host_ptr_is_pinned(crops)) {
        // the group path with pageable caller memory: staged through the slot's pinned buffer in 4 MiB pieces, so that this call never
        // waits for a device (hipMemcpyAsync from pageable memory held member i + 1's submission behind member i's staging)
        if (sl.host_in_cap < in_bytes) {
            if (sl.host_in) { HIPCHK(c, hipStreamSynchronize(c->copy_stream)); hipHostFree(sl.host_in); sl.host_in = nullptr; sl.host_in_cap = 0; }
            void* q = nullptr;
            HIPCHK(c, hipHostMalloc(&q, in_bytes, hipHostMallocDefault));
            sl.host_in = (char*)q; sl.host_in_cap = in_bytes;
        }
        const size_t piece = (size_t)4 << 20;
        for (size_t off = 0; off < in_bytes; off += piece) {
            const size_t len = in_bytes - off < piece ? in_bytes - off : piece;
            std::memcpy(sl.host_in + off, (const char*)crops + off, len);
            HIPCHK(c, hipMemcpyAsync((char*)sl.in + off, sl.host_in + off, len, hipMemcpyHostToDevice, c->copy_stream));
        }
    } else {
        HIPCHK(c, hipMemcpyAsync(sl.in, crops, in_bytes, hipMemcpyHostToDevice, c->copy_stream));
    }
    if (org_wh) HIPCHK(c, hipMemcpyAsync(sl.wh, org_wh, (size_t)n * 8, hipMemcpyHostToDevice, c->copy_stream));
    HIPCHK(c, hipEventRecord(sl.h2d, c->copy_stream));
    // compute stream: after the upload
    HIPCHK(c, hipStreamWaitEvent(c->stream, sl.h2d, 0));
    if ((rc = run_chunk(c, sl.in, fmt, n, org_wh ? sl.wh : nullptr, sl.kp))) return rc;
    HIPCHK(c, hipEventRecord(sl.done, c->stream));
    // download stream: D2H of the keypoints after the compute
    HIPCHK(c, hipStreamWaitEvent(c->d2h_stream, sl.done, 0));
    float* dst = out;
    sl.user_out = nullptr;
    if (stage_out) {
        if (!sl.host_kp) {
            void* q = nullptr;
            HIPCHK(c, hipHostMalloc(&q, (size_t)c->maxb * c->Kp * 12, hipHostMallocDefault));
            sl.host_kp = (float*)q;
        }
        dst = sl.host_kp;
        sl.user_out = out;
        sl.out_bytes = (size_t)n * c->Kp * 12;
    }
    HIPCHK(c, hipMemcpyAsync(dst, sl.kp, (size_t)n * c->Kp * 12, hipMemcpyDeviceToHost, c->d2h_stream));
    HIPCHK(c, hipEventRecord(sl.out, c->d2h_stream));
    sl.busy = true;
    *slot_out = si;
    c->next_slot = si ^ 1;
    return VP_OK;
}

int vp_infer_submit(vp_handle c, const void* crops, int32_t fmt, int32_t n, const int32_t* org_wh, float* out, int32_t* slot_out) {
    return submit_impl(c, crops, fmt, n, org_wh, out, slot_out, false);
}

int vp_infer_wait(vp_handle c, int32_t slot) {
    if (!c || slot < 0 || slot > 1) return VP_ERR_INVALID;
    vp_ctx::Slot& sl = c->slots[slot];
    if (!sl.busy) return fail(c, VP_ERR_STATE, "slot not in flight");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    HIPCHK(c, hipEventSynchronize(sl.out));
    if (sl.user_out) { std::memcpy(sl.user_out, sl.host_kp, sl.out_bytes); sl.user_out = nullptr; }
    sl.busy = false;
    return VP_OK;
}

int vp_infer_flip(vp_handle c, const void* crops, int32_t fmt, int32_t n, const int32_t* org_wh, const int32_t* flip_pairs,
                  int32_t n_pairs, int32_t shift_heatmap, float* out, float* heatmaps) {
    int rc = check_ready(c, fmt, n, crops, out ? (const void*)out : (const void*)heatmaps);
    if (rc) return rc;
    if (n_pairs < 0 || (n_pairs > 0 && !flip_pairs)) return fail(c, VP_ERR_INVALID, "bad flip_pairs");
    if (n == 0) return VP_OK;   // nothing to do (and no async upload of the stack-lifetime partner table left in flight)
    std::vector<int32_t> partner(c->Kp);
    for (int k = 0; k < c->Kp; ++k) partner[k] = k;
    for (int i = 0; i < n_pairs; ++i) {
        const int a = flip_pairs[2 * i], b = flip_pairs[2 * i + 1];
        if (a < 0 || b < 0 || a >= c->Kp || b >= c->Kp) return fail(c, VP_ERR_INVALID, "flip pair index out of range");
        partner[a] = b;
        partner[b] = a;
    }
    const size_t hm_elems = (size_t)c->maxb * c->Kp * 3072;
    if (!c->hm_keep && (rc = dalloc(c, &c->hm_keep, hm_elems))) return rc;
    if (!c->partner && (rc = dalloc(c, &c->partner, (size_t)c->Kp))) return rc;
    HIPCHK(c, hipMemcpy(c->partner, partner.data(), (size_t)c->Kp * 4, hipMemcpyHostToDevice));   // synchronous: `partner` is a local
    for (int off = 0; off < n; off += c->maxb) {
        const int nb = (n - off < c->maxb) ? n - off : c->maxb;
        const char* src = (const char*)crops + (size_t)off * crop_bytes(fmt);
        HIPCHK(c, hipMemcpyAsync(c->in_stage, src, (size_t)nb * crop_bytes(fmt), hipMemcpyHostToDevice, c->stream));
        if (org_wh) HIPCHK(c, hipMemcpyAsync(c->wh_stage, org_wh + 2 * (size_t)off, (size_t)nb * 8, hipMemcpyHostToDevice, c->stream));
        if ((rc = forward_chunk(c, c->in_stage, fmt, nb, false, false))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->hm_keep, c->hm, (size_t)nb * c->Kp * 3072 * 4, hipMemcpyDeviceToDevice, c->stream));
        if ((rc = forward_chunk(c, c->in_stage, fmt, nb, false, true))) return rc;         // mirrored crops -> c->hm
        HIPCHK(c, vp::flip_merge_launch(c->hm_keep, c->hm, c->partner, nb, c->Kp, shift_heatmap ? 1 : 0, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->hm, c->hm_keep, (size_t)nb * c->Kp * 3072 * 4, hipMemcpyDeviceToDevice, c->stream));
        if (heatmaps) HIPCHK(c, hipMemcpyAsync(heatmaps + (size_t)off * c->Kp * 3072, c->hm, (size_t)nb * c->Kp * 3072 * 4, hipMemcpyDeviceToHost, c->stream));
        if (out) {
            if ((rc = decode_chunk(c, org_wh ? c->wh_stage : nullptr, c->kp, nb))) return rc;
            HIPCHK(c, hipMemcpyAsync(out + (size_t)off * c->Kp * 3, c->kp, (size_t)nb * c->Kp * 12, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return VP_OK;
}

int vp_infer_heatmaps(vp_handle c, const void* crops, int32_t fmt, int32_t n, float* heatmaps) {
    int rc = check_ready(c, fmt, n, crops, heatmaps);
    if (rc) return rc;
    for (int off = 0; off < n; off += c->maxb) {
        const int nb = (n - off < c->maxb) ? n - off : c->maxb;
        HIPCHK(c, hipMemcpyAsync(c->in_stage, (const char*)crops + (size_t)off * crop_bytes(fmt), (size_t)nb * crop_bytes(fmt), hipMemcpyHostToDevice, c->stream));
        if ((rc = forward_chunk(c, c->in_stage, fmt, nb, false))) return rc;
        HIPCHK(c, hipMemcpyAsync(heatmaps + (size_t)off * c->Kp * 3072, c->hm, (size_t)nb * c->Kp * 3072 * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return VP_OK;
}

int vp_infer_tokens(vp_handle c, const void* crops, int32_t fmt, int32_t n, float* tokens) {
    int rc = check_ready(c, fmt, n, crops, tokens);
    if (rc) return rc;
    if (!c->tok && (rc = dalloc(c, &c->tok, (size_t)c->maxb * 192 * c->D))) return rc;
    for (int off = 0; off < n; off += c->maxb) {
        const int nb = (n - off < c->maxb) ? n - off : c->maxb;
        HIPCHK(c, hipMemcpyAsync(c->in_stage, (const char*)crops + (size_t)off * crop_bytes(fmt), (size_t)nb * crop_bytes(fmt), hipMemcpyHostToDevice, c->stream));
        if ((rc = forward_chunk(c, c->in_stage, fmt, nb, true))) return rc;
        HIPCHK(c, hipMemcpyAsync(tokens + (size_t)off * 192 * c->D, c->tok, (size_t)nb * 192 * c->D * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return VP_OK;
}

int vp_infer_frame(vp_handle c, const uint8_t* frame, int32_t fh, int32_t fw, const int32_t* crop_params, int32_t n, float* out) {
    int rc = check_ready(c, VP_INPUT_U8_NHWC, n, frame, out);
    if (rc) return rc;
    if (fh <= 0 || fw <= 0 || (n > 0 && !crop_params)) return fail(c, VP_ERR_INVALID, "bad frame geometry");
    const size_t fbytes = (size_t)fh * fw * 3;
    if (fbytes > c->frame_cap) {
        void* q = nullptr;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMalloc(&q, fbytes + 256));
        if (c->frame_stage) {        // release the smaller staging buffer now: growing resolutions must not grow device memory
            for (auto it = c->allocs.begin(); it != c->allocs.end(); ++it)
                if (*it == (void*)c->frame_stage) { c->allocs.erase(it); break; }
            hipFree(c->frame_stage);
        }
        c->allocs.push_back(q);
        c->frame_stage = (uint8_t*)q;
        c->frame_cap = fbytes;
    }
    if (!c->cparams && (rc = dalloc(c, &c->cparams, (size_t)c->maxb * 8))) return rc;
    for (int i = 0; i < n; ++i) {
        const int32_t* p = crop_params + 8 * (size_t)i;
        if (p[0] < 0 || p[1] < 0 || p[2] <= 0 || p[3] <= 0 || p[0] + p[2] > fw || p[1] + p[3] > fh || p[4] < 0 || p[5] < 0 ||
            p[4] + p[2] > p[6] || p[5] + p[3] > p[7])
            return fail(c, VP_ERR_INVALID, "crop " + std::to_string(i) + " lies outside the frame / its padded canvas");
    }
    HIPCHK(c, hipMemcpyAsync(c->frame_stage, frame, fbytes, hipMemcpyHostToDevice, c->stream));
    for (int off = 0; off < n; off += c->maxb) {
        const int nb = (n - off < c->maxb) ? n - off : c->maxb;
        HIPCHK(c, hipMemcpyAsync(c->cparams, crop_params + 8 * (size_t)off, (size_t)nb * 32, hipMemcpyHostToDevice, c->stream));
        LAUNCH(c, VP_PROF_IM2COL, 0.0, (double)nb * 256 * 192 * 3 * 5,
               vp::crop_resize_launch(c->frame_stage, fh, fw, c->cparams, (uint8_t*)c->in_stage, nb, c->stream));
        // decode scales by the padded-canvas size (pw, ph) of each crop = the image pre_img receives
        std::vector<int32_t> wh((size_t)nb * 2);
        for (int i = 0; i < nb; ++i) { wh[2 * i] = crop_params[8 * (size_t)(off + i) + 6]; wh[2 * i + 1] = crop_params[8 * (size_t)(off + i) + 7]; }
        HIPCHK(c, hipMemcpyAsync(c->wh_stage, wh.data(), (size_t)nb * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));   // wh is a stack-lifetime host buffer
        if ((rc = run_chunk(c, c->in_stage, VP_INPUT_U8_NHWC, nb, c->wh_stage, c->kp))) return rc;
        HIPCHK(c, hipMemcpyAsync(out + (size_t)off * c->Kp * 3, c->kp, (size_t)nb * c->Kp * 12, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return VP_OK;
}

int vp_decode_only(int32_t device_id, const float* heatmaps, int32_t n, int32_t k, const int32_t* org_wh, float* out) {
    if (!heatmaps || !out || n <= 0 || k <= 0) return fail(nullptr, VP_ERR_INVALID, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, VP_ERR_HIP, "no HIP device available (no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, VP_ERR_INVALID, "device_id out of range");
    vp_ctx* c = nullptr;   // errors below are reported through the create-error slot
    HIPCHK(c, hipSetDevice(device_id));
    float *d_hm = nullptr, *d_out = nullptr;
    int32_t* d_wh = nullptr;
    const size_t hb = (size_t)n * k * 3072 * 4, ob = (size_t)n * k * 12;
    int rc = VP_OK;
    hipError_t e = hipMalloc((void**)&d_hm, hb);
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, ob);
    if (e == hipSuccess) e = hipMemcpy(d_hm, heatmaps, hb, hipMemcpyHostToDevice);
    if (e == hipSuccess && org_wh) {
        e = hipMalloc((void**)&d_wh, (size_t)n * 8);
        if (e == hipSuccess) e = hipMemcpy(d_wh, org_wh, (size_t)n * 8, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = vp::decode_launch(d_hm, d_wh, d_out, n, k, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d_out, ob, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(nullptr, VP_ERR_HIP, std::string("vp_decode_only: ") + hipGetErrorString(e));
    if (d_hm) hipFree(d_hm);
    if (d_out) hipFree(d_out);
    if (d_wh) hipFree(d_wh);
    return rc;
}

// ---- multi-GPU group (one process, N devices): crops sharded contiguously, weights replicated ------------------------------
struct vp_group {
    std::vector<vp_ctx*> h;
    int peer_missing = 0;        // ordered device pairs without peer access (their all-gather copies are staged through the host)
    std::vector<float*> d_all;   // per device: [max_total, K, 3] keypoints of EVERY shard (vp_group_infer_allgather)
    size_t all_cap = 0;
    std::string err;
};
namespace { thread_local std::string g_group_error; }

int vp_group_create(vp_group_handle* out, const vp_config* cfg, const int32_t* device_ids, int32_t n_devices) {
    if (!out || !cfg || !device_ids || n_devices <= 0) { g_group_error = "null argument"; return VP_ERR_INVALID; }
    *out = nullptr;
    vp_group* g = new vp_group();
    for (int i = 0; i < n_devices; ++i) {
        vp_config c = *cfg;
        c.device_id = device_ids[i];
        vp_handle h = nullptr;
        int rc = vp_create(&h, &c);
        if (rc) { g_group_error = std::string("device ") + std::to_string(device_ids[i]) + ": " + vp_last_error(nullptr); vp_group_destroy(g); return rc; }
        g->h.push_back(h);
    }
    // peer access for the device-side all-gather (xGMI links are point to point: one copy per pair)
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < n_devices; ++j)
            if (i != j) {
                hipSetDevice(device_ids[i]);
                int can = 0;
                bool ok = false;
                if (hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[j]) == hipSuccess && can) {
                    hipError_t e = hipDeviceEnablePeerAccess(device_ids[j], 0);
                    ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
                }
                if (!ok) { (void)hipGetLastError(); ++g->peer_missing; }   // not fatal: hipMemcpyPeerAsync stages such a pair through the host
            }
    *out = g;
    return VP_OK;
}

int vp_group_size(vp_group_handle g) { return g ? (int)g->h.size() : 0; }
int vp_group_peer_access_missing(vp_group_handle g) { return g ? g->peer_missing : -1; }

int vp_group_load_weights(vp_group_handle g, const vp_tensor_desc* tensors, int32_t n_tensors) {
    if (!g) return VP_ERR_INVALID;
    for (auto* h : g->h) {
        int rc = vp_load_weights(h, tensors, n_tensors);
        if (rc) { g->err = h->err; return rc; }
    }
    return VP_OK;
}

// shard i of n crops over w devices: [off, off + cnt), contiguous, ceil(n / w) per device (the last ones may be short or empty)
static void group_shard(int n, int w, int i, int& off, int& cnt) {
    const int per = (n + w - 1) / w;
    off = per * i < n ? per * i : n;
    cnt = n - off < per ? n - off : per;
}

// the whole plan of a call: rounds of (devices x max_batch) crops, entry e = round * w + device -> [offs[e], offs[e] + cnts[e])
static int group_plan(int n, int w, int maxb, std::vector<int>& offs, std::vector<int>& cnts) {
    offs.clear(); cnts.clear();
    if (n < 0 || w <= 0 || maxb <= 0) return -1;
    const long per_round = (long)w * maxb;
    for (long r0 = 0; r0 < n; r0 += per_round) {
        const int nr = (int)(n - r0 < per_round ? n - r0 : per_round);
        for (int i = 0; i < w; ++i) {
            int off, cnt;
            group_shard(nr, w, i, off, cnt);
            offs.push_back((int)r0 + off);
            cnts.push_back(cnt);
        }
    }
    return (int)offs.size();
}

int vp_dbg_group_plan(int32_t n, int32_t w, int32_t maxb, int32_t* offs, int32_t* cnts, int32_t cap) {
    std::vector<int> o, k;
    const int e = group_plan(n, w, maxb, o, k);
    if (e < 0 || cap < 0 || (cap > 0 && (!offs || !cnts))) return -1;
    for (int i = 0; i < e && i < cap; ++i) { offs[i] = o[i]; cnts[i] = k[i]; }
    return e;
}

// The two-phase schedule of a group call, as ONE function for the real path (group_run) and for the host-only trace
// (vp_dbg_group_trace): per round of `w` plan entries, phase 1 calls submit(member, off, cnt) for EVERY member with work before phase 2
// calls wait(member) for any of them.  submit returns 0 or an error code; on an error every member already submitted in this round
// is waited for (drained) before the error is returned, so no slot of any member stays in flight.
extern "C++" {
template <class Submit, class Wait>
static int group_rounds(const std::vector<int>& offs, const std::vector<int>& cnts, int w, Submit submit, Wait wait) {
    const int entries = (int)offs.size();
    for (int e0 = 0; e0 < entries; e0 += w) {
        std::vector<char> inflight(w, 0);
        auto drain = [&](int from) { for (int i = from; i < w; ++i) if (inflight[i]) { wait(i); inflight[i] = 0; } };
        for (int i = 0; i < w; ++i) {                    // phase 1 -- enqueue on every member; nothing here waits for a device
            if (cnts[e0 + i] <= 0) continue;
            const int rc = submit(i, offs[e0 + i], cnts[e0 + i]);
            if (rc) { drain(0); return rc; }
            inflight[i] = 1;
        }
        for (int i = 0; i < w; ++i)                      // phase 2 -- collect
            if (inflight[i]) {
                inflight[i] = 0;
                const int rc = wait(i);
                if (rc) { drain(i + 1); return rc; }
            }
    }
    return VP_OK;
}
}   // extern "C++"

// host-only: the order in which a call of n crops on w members of max_batch maxb submits (+ (member + 1)) and waits (- (member + 1)),
// with stub members (tests/test_host_logic.py: every submission of a round precedes its first wait)
int vp_dbg_group_trace(int32_t n, int32_t w, int32_t maxb, int32_t* trace, int32_t cap) {
    std::vector<int> offs, cnts;
    if (group_plan(n, w, maxb, offs, cnts) < 0 || cap < 0 || (cap > 0 && !trace)) return -1;
    int len = 0;
    auto put = [&](int v) { if (len < cap) trace[len] = v; ++len; };
    group_rounds(offs, cnts, w, [&](int i, int, int) { put(i + 1); return 0; }, [&](int i) { put(-(i + 1)); return 0; });
    return len;
}

static int group_run(vp_group* g, const void* crops, int32_t fmt, int32_t n, const int32_t* org_wh, float* out, float* const* d_all) {
    if (!g || n < 0 || (n > 0 && (!crops || (!out && !d_all)))) return VP_ERR_INVALID;
    const int w = (int)g->h.size();
    const int K = g->h[0]->Kp;
    std::vector<float> scratch;
    if (!out) { scratch.resize((size_t)n * K * 3); out = scratch.data(); }
    std::vector<int> offs, cnts;
    if (group_plan(n, w, g->h[0]->maxb, offs, cnts) < 0) return VP_ERR_INVALID;
    std::vector<int> slot(w, -1);
    // phase 1 per member: upload (pinned caller memory as it is, pageable memory through the member's pinned staging buffer), model,
    // decode, download into the member's pinned staging buffer, and the peer copies of the device-side all-gather -- all enqueued, none
    // waited for, so the members compute concurrently.  phase 2: wait for the member's download, copy its slice to the caller's buffer.
    auto submit = [&](int i, int off, int cnt) -> int {
        vp_ctx* c = g->h[i];
        int rc = submit_impl(c, (const char*)crops + (size_t)off * crop_bytes(fmt), fmt, cnt, org_wh ? org_wh + 2 * (size_t)off : nullptr,
                             out + (size_t)off * K * 3, &slot[i], true);
        if (rc) { g->err = c->err; slot[i] = -1; return rc; }
        if (d_all) {   // all-gather on the device side: this shard's keypoints to every device's copy, peer to peer, on the owner's stream
            for (int j = 0; j < w; ++j) {
                hipError_t e = hipMemcpyPeerAsync(d_all[j] + (size_t)off * K * 3, g->h[j]->cfg.device_id, c->slots[slot[i]].kp,
                                                  c->cfg.device_id, (size_t)cnt * K * 12, c->stream);
                if (e != hipSuccess) {
                    g->err = std::string("hipMemcpyPeerAsync: ") + hipGetErrorString(e);
                    vp_infer_wait(c, slot[i]); slot[i] = -1;   // this member is not marked in flight yet: collect it here
                    return VP_ERR_HIP;
                }
            }
        }
        return VP_OK;
    };
    auto wait = [&](int i) -> int {
        int rc = vp_infer_wait(g->h[i], slot[i]);
        slot[i] = -1;
        if (!rc && d_all) rc = vp_synchronize(g->h[i]);
        if (rc) g->err = g->h[i]->err;
        return rc;
    };
    return group_rounds(offs, cnts, w, submit, wait);
}

int vp_group_infer(vp_group_handle g, const void* crops, int32_t fmt, int32_t n, const int32_t* org_wh, float* out) {
    if (!out && n > 0) return VP_ERR_INVALID;
    return group_run(g, crops, fmt, n, org_wh, out, nullptr);
}

int vp_group_infer_allgather(vp_group_handle g, const void* crops, int32_t fmt, int32_t n, const int32_t* org_wh, float* const* d_all, float* out) {
    if (!g || !d_all) return VP_ERR_INVALID;
    return group_run(g, crops, fmt, n, org_wh, out, d_all);
}

vp_handle vp_group_member(vp_group_handle g, int32_t i) { return (g && i >= 0 && i < (int)g->h.size()) ? g->h[i] : nullptr; }

int vp_group_destroy(vp_group_handle g) {
    if (!g) return VP_OK;
    for (auto* h : g->h) vp_destroy(h);
    delete g;
    return VP_OK;
}

const char* vp_group_last_error(vp_group_handle g) { return g ? g->err.c_str() : g_group_error.c_str(); }

void* vp_stream(vp_handle c) { return c ? (void*)c->stream : nullptr; }

int vp_synchronize(vp_handle c) {
    if (!c) return VP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    if (c->foreign_pending && c->ev_sw) HIPCHK(c, hipEventSynchronize(c->ev_sw));   // the last call ran on a caller's stream
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VP_OK;
}

int vp_set_profiling(vp_handle c, int32_t enable) {
    if (!c) return VP_ERR_INVALID;
    c->prof = (uint32_t)enable;   // bitmask over VP_PROF_* families (-1 = all)
    return VP_OK;
}

int vp_reset_profile(vp_handle c) {
    if (!c) return VP_ERR_INVALID;
    if (c->foreign_pending && c->ev_sw) hipEventSynchronize(c->ev_sw);   // a small stream-ordered call recorded its timing events on the CALLER's stream
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    std::memset(&c->acc, 0, sizeof(c->acc));
    return VP_OK;
}

int vp_get_profile(vp_handle c, vp_profile* out) {
    if (!c || !out) return VP_ERR_INVALID;
    if (c->foreign_pending && c->ev_sw) HIPCHK(c, hipEventSynchronize(c->ev_sw));   // as vp_synchronize: the last call may have run (and been timed) on a caller's stream
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    *out = c->acc;
    return VP_OK;
}

int vp_profile_kernel(vp_handle c, int32_t family, char* buf, int32_t cap) {
    if (!c || family < 0 || family >= VP_PROF_COUNT || !buf || cap <= 0) return VP_ERR_INVALID;
    snprintf(buf, (size_t)cap, "%s", c->kernel_desc[family].c_str());
    return VP_OK;
}

int vp_destroy(vp_handle c) {
    if (!c) return VP_OK;
    hipSetDevice(c->cfg.device_id);
    if (c->foreign_pending && c->ev_sw) hipEventSynchronize(c->ev_sw);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto& e : c->evs) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (auto& p : c->ev_pool) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    for (auto& sl : c->slots) { if (sl.h2d) hipEventDestroy(sl.h2d); if (sl.done) hipEventDestroy(sl.done); if (sl.out) hipEventDestroy(sl.out); if (sl.host_kp) hipHostFree(sl.host_kp); if (sl.host_in) hipHostFree(sl.host_in); }
    if (c->ev_in) hipEventDestroy(c->ev_in);
    if (c->ev_out) hipEventDestroy(c->ev_out);
    if (c->ev_sw) hipEventDestroy(c->ev_sw);
    for (auto& ge : c->graphs) if (ge.exec) hipGraphExecDestroy(ge.exec);
    for (void* p : c->allocs) hipFree(p);
    if (c->copy_stream) { hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); }
    if (c->d2h_stream) { hipStreamSynchronize(c->d2h_stream); hipStreamDestroy(c->d2h_stream); }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return VP_OK;
}

const char* vp_last_error(vp_handle c) { return c ? c->err.c_str() : g_create_error.c_str(); }

}  // extern "C"
