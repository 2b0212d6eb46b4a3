// The weight packer behind vp_load_weights (include/vitpose_hip.h): host fp32 tensors named as in the reference's state dict
// (SURVEY.md 8a "State-dict schema") -> the device operands of the kernels: 16-bit rounding, BatchNorm folded into the deconv weights,
// LayerNorm's gamma / beta folded into qkv / fc1, pos + cls + conv bias pre-added, the deconvs re-tiled into four output-parity GEMM
// operands, the final 1x1 conv as hi + lo pairs, e4m3 codes + scales in the fp8 mode.
#include "api_internal.h"

using namespace vpi;

namespace vpi {

// fp32 -> 16-bit storage on the host (round to nearest even), same as the device paths
uint16_t host_to_bits(float v, int dtype) {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    if (dtype == vp::DT_BF16) {
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    }
    // IEEE binary16, RNE, with subnormals; saturate to +-65504
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);   // >= 65520 rounds past max -> saturate
    if (a < 0x33000001u) return (uint16_t)sign;                // < 2^-25 -> 0
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;
    if (e < -14) {                                             // subnormal half
        const int shift = -14 - e + 13;
        const uint32_t r = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        uint32_t h = r;
        if (rem > half || (rem == half && (r & 1))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3ffu);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return (uint16_t)(sign | h);
}

int upload_f32(vp_ctx* c, float** dst, const float* src, size_t n, size_t npad) {
    if (npad < n) npad = n;
    std::vector<float> tmp(npad, 0.f);
    std::memcpy(tmp.data(), src, n * 4);
    int rc = dalloc(c, dst, npad);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(*dst, tmp.data(), npad * 4, hipMemcpyHostToDevice));
    return VP_OK;
}

// rows x cols fp32 matrix -> 16-bit, rows padded with zeros to rows_pad
int upload_mat(vp_ctx* c, uint16_t** dst, const float* src, size_t rows, size_t cols, size_t rows_pad) {
    std::vector<uint16_t> tmp(rows_pad * cols, 0);
    for (size_t i = 0; i < rows * cols; ++i) tmp[i] = host_to_bits(src[i], c->dtype);
    int rc = dalloc(c, dst, rows_pad * cols);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(*dst, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
    return VP_OK;
}

float host_from_bits(uint16_t h, int dtype) {
    uint32_t u;
    if (dtype == vp::DT_BF16) {
        u = (uint32_t)h << 16;
    } else {
        const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu;
        if (e == 0) {
            if (m == 0) u = sign;
            else { int sh = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; ++sh; }
                   u = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3ffu) << 13); }
        } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
        else u = sign | ((e + 112) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

size_t pad128(size_t n) { return (n + 255) / 256 * 256; }

// final 1x1 conv weights as a hi + lo pair of 16-bit values (W = hi + lo to ~22 bits): 16-row groups interleaved
// [16 hi rows][16 lo rows] so that the two MFMA accumulator fragments a lane sums in the EPI_HEATMAP epilogue are
// the hi and lo products of the SAME output columns.  The GEMM is HBM-bound on its A operand, so the doubled MFMA
// work is free, and the final layer's weight rounding (9 % of the heatmap error variance, tests/precision_budget.py)
// disappears.  Physical rows: 32 * ceil(Kp / 16).
int upload_final(vp_ctx* c, uint16_t** dst, const float* src, size_t kp, size_t cols, size_t* rows_phys) {
    const size_t groups = (kp + 15) / 16, rows = groups * 32, rows_pad = pad128(rows);
    std::vector<uint16_t> tmp(rows_pad * cols, 0);
    for (size_t n = 0; n < kp; ++n)
        for (size_t k = 0; k < cols; ++k) {
            const float w = src[n * cols + k];
            const uint16_t hi = host_to_bits(w, c->dtype);
            const uint16_t lo = host_to_bits(w - host_from_bits(hi, c->dtype), c->dtype);
            const size_t r = (n / 16) * 32 + (n % 16);
            tmp[r * cols + k] = hi;
            tmp[(r + 16) * cols + k] = lo;
        }
    *rows_phys = rows;
    int rc = dalloc(c, dst, rows_pad * cols);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(*dst, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
    return VP_OK;
}

// LayerNorm folded into the following nn.Linear (y = LN(x) W^T + b):
//   W'[n][k] = gamma[k] W[n][k] (rounded to the operand type), s[n] = sum_k W'[n][k] (of the ROUNDED values, so the
//   identity  LN(x).W^T = rstd (x.W'^T - mean s) + c  holds exactly for what the MFMA multiplies), c[n] = sum_k beta[k] W[n][k] + b[n]
int upload_ln_folded(vp_ctx* c, uint16_t** w_out, float** s_out, float** c_out, const float* W, const float* b,
                     const float* gamma, const float* beta, size_t N, size_t K) {
    const size_t rows_pad = pad128(N);
    std::vector<uint16_t> wq(rows_pad * K, 0);
    std::vector<float> s(rows_pad, 0.f), cc(rows_pad, 0.f);
    for (size_t n = 0; n < N; ++n) {
        double ss = 0.0, sc = 0.0;
        for (size_t k = 0; k < K; ++k) {
            const uint16_t q = host_to_bits(gamma[k] * W[n * K + k], c->dtype);
            wq[n * K + k] = q;
            ss += (double)host_from_bits(q, c->dtype);
            sc += (double)beta[k] * (double)W[n * K + k];
        }
        s[n] = (float)ss;
        cc[n] = (float)(sc + (double)b[n]);
    }
    int rc = dalloc(c, w_out, rows_pad * K);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(*w_out, wq.data(), wq.size() * 2, hipMemcpyHostToDevice));
    if ((rc = upload_f32(c, s_out, s.data(), rows_pad))) return rc;
    return upload_f32(c, c_out, cc.data(), rows_pad);
}   // weight rows: multiple of the largest BN tile (256)

// fp8 mode: rows of W [N, K] (optionally with LayerNorm's gamma folded in: W'[n][k] = gamma[k] W[n][k]) -> OCP e4m3 codes with one fp32
// scale per output channel (max |row| / 448), rows zero-padded to a multiple of 256; c_out (optional) = sum_k beta[k] W[n][k] + b[n]
int upload_fp8_rows(vp_ctx* c, uint8_t** w_out, float** ws_out, float** c_out, const float* W, const float* b, const float* gamma,
                    const float* beta, size_t N, size_t K) {
    const size_t rows_pad = pad128(N);
    std::vector<uint8_t> wq(rows_pad * K, 0);
    std::vector<float> ws(rows_pad, 1.f), cc(rows_pad, 0.f), row(K);
    for (size_t n = 0; n < N; ++n) {
        float amax = 0.f;
        double sc = 0.0;
        for (size_t k = 0; k < K; ++k) {
            row[k] = gamma ? gamma[k] * W[n * K + k] : W[n * K + k];
            amax = std::fmax(amax, std::fabs(row[k]));
            if (beta) sc += (double)beta[k] * (double)W[n * K + k];
        }
        const float sn = amax > 0.f ? amax / 448.0f : 1.0f;
        ws[n] = sn;
        for (size_t k = 0; k < K; ++k) wq[n * K + k] = vp_host_e4m3(row[k] / sn);
        cc[n] = (float)(sc + (b ? (double)b[n] : 0.0));
    }
    int rc = dalloc(c, w_out, rows_pad * K);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(*w_out, wq.data(), wq.size(), hipMemcpyHostToDevice));
    if ((rc = upload_f32(c, ws_out, ws.data(), rows_pad))) return rc;
    if (c_out) return upload_f32(c, c_out, cc.data(), rows_pad);
    return VP_OK;
}

// ConvTranspose2d(Cin, 256, 4, s=2, p=1, bias=False) + BatchNorm2d(eval, eps=1e-5)
// (topdown_heatmap_simple_head.py:291-321) -> 4 output-parity GEMM operands
//   Wp[parity=(a,b)][o][t*Cin + c] = w[c][o][ky(a,ti)][kx(b,tj)] * gamma[o]/sqrt(var[o]+eps),  t = ti*2+tj
//   a=0: ti=0 -> ky=1 (input row i), ti=1 -> ky=3 (row i-1);  a=1: ti=0 -> ky=0 (row i+1), ti=1 -> ky=2 (row i)
//   bias[o] = beta[o] - mean[o]*scale[o]
int pack_deconv(vp_ctx* c, Lookup& lk, int idx, int Cin, uint16_t** w_out, float** b_out) {
    const std::string h = "keypoint_head.deconv_layers.";
    const float *w, *g, *b, *mu, *var;
    int rc;
    if ((rc = lk.get(h + std::to_string(idx) + ".weight", (int64_t)Cin * 256 * 16, &w))) return rc;
    if ((rc = lk.get(h + std::to_string(idx + 1) + ".weight", 256, &g))) return rc;
    if ((rc = lk.get(h + std::to_string(idx + 1) + ".bias", 256, &b))) return rc;
    if ((rc = lk.get(h + std::to_string(idx + 1) + ".running_mean", 256, &mu))) return rc;
    if ((rc = lk.get(h + std::to_string(idx + 1) + ".running_var", 256, &var))) return rc;
    std::vector<float> scale(256), bias(256);
    for (int o = 0; o < 256; ++o) {
        scale[o] = g[o] / std::sqrt(var[o] + 1e-5f);
        bias[o] = b[o] - mu[o] * scale[o];
    }
    const size_t K = (size_t)4 * Cin;
    std::vector<float> wp((size_t)4 * 256 * K);
    for (int pa = 0; pa < 2; ++pa)
        for (int pb = 0; pb < 2; ++pb)
            for (int o = 0; o < 256; ++o)
                for (int ti = 0; ti < 2; ++ti)
                    for (int tj = 0; tj < 2; ++tj) {
                        const int ky = pa ? (ti ? 2 : 0) : (ti ? 3 : 1);
                        const int kx = pb ? (tj ? 2 : 0) : (tj ? 3 : 1);
                        float* dst = &wp[(((size_t)(pa * 2 + pb) * 256 + o) * 4 + (ti * 2 + tj)) * Cin];
                        for (int ci = 0; ci < Cin; ++ci)
                            dst[ci] = w[(((size_t)ci * 256 + o) * 4 + ky) * 4 + kx] * scale[o];
                    }
    if ((rc = upload_mat(c, w_out, wp.data(), (size_t)4 * 256, K, (size_t)4 * 256))) return rc;
    return upload_f32(c, b_out, bias.data(), 256);
}

}  // namespace vpi

extern "C" {

int vp_load_weights(vp_handle c, const vp_tensor_desc* tensors, int32_t n_tensors) {
    if (!c || !tensors || n_tensors <= 0) return fail(c, VP_ERR_INVALID, "null argument");
    if (c->loaded) return fail(c, VP_ERR_STATE, "weights already loaded on this handle");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    Lookup lk;
    lk.c = c;
    for (int i = 0; i < n_tensors; ++i)
        if (tensors[i].name) lk.map[tensors[i].name] = &tensors[i];
    const int D = c->D;
    const size_t DD = (size_t)D * D;
    int rc;
    const float *p, *q;
    // patch embed + positional embedding (vit.py:222, :382): aux[t] = pos[1+t] + pos[0] + conv bias
    if ((rc = lk.get("backbone.patch_embed.proj.weight", (int64_t)D * 768, &p))) return rc;
    if ((rc = upload_mat(c, &c->w_patch, p, D, 768, pad128(D)))) return rc;
    if ((rc = lk.get("backbone.pos_embed", (int64_t)193 * D, &p))) return rc;
    if ((rc = lk.get("backbone.patch_embed.proj.bias", D, &q))) return rc;
    {
        std::vector<float> pos((size_t)192 * D);
        for (int t = 0; t < 192; ++t)
            for (int d = 0; d < D; ++d) pos[(size_t)t * D + d] = (p[(size_t)(1 + t) * D + d] + p[d]) + q[d];
        if ((rc = upload_f32(c, &c->pos, pos.data(), pos.size()))) return rc;
    }
    c->blocks.resize(c->L);
    for (int l = 0; l < c->L; ++l) {
        Block& b = c->blocks[l];
        const std::string pre = "backbone.blocks." + std::to_string(l) + ".";
        const float *g1, *be1, *g2, *be2, *wq, *bq, *w1, *b1;
        if ((rc = lk.get(pre + "norm1.weight", D, &g1)) || (rc = lk.get(pre + "norm1.bias", D, &be1)) ||
            (rc = lk.get(pre + "norm2.weight", D, &g2)) || (rc = lk.get(pre + "norm2.bias", D, &be2)) ||
            (rc = lk.get(pre + "attn.qkv.weight", (int64_t)3 * DD, &wq)) || (rc = lk.get(pre + "attn.qkv.bias", 3 * D, &bq)) ||
            (rc = lk.get(pre + "mlp.fc1.weight", (int64_t)4 * DD, &w1)) || (rc = lk.get(pre + "mlp.fc1.bias", 4 * D, &b1)))
            return rc;
        if (c->fp8) {
            if ((rc = upload_fp8_rows(c, &b.w_qkv8, &b.ws_qkv, &b.b_qkv, wq, bq, g1, be1, 3 * (size_t)D, D))) return rc;
            if ((rc = upload_fp8_rows(c, &b.w_fc18, &b.ws_fc1, &b.b_fc1, w1, b1, g2, be2, 4 * (size_t)D, D))) return rc;
        } else if (c->fuse_ln) {
            if ((rc = upload_ln_folded(c, &b.w_qkv, &b.s_qkv, &b.b_qkv, wq, bq, g1, be1, 3 * (size_t)D, D))) return rc;
            if (c->fuse_qkv_attn && D / c->heads == 64) {   // head-major copies for the fused qkv + attention kernel
                if ((rc = dalloc(c, &b.w_qkvh, 3 * (size_t)D * D)) || (rc = dalloc(c, &b.b_qkvh, 3 * (size_t)D)) || (rc = dalloc(c, &b.s_qkvh, 3 * (size_t)D))) return rc;
                HIPCHK(c, vp::qkv_head_major_launch(b.w_qkv, b.b_qkv, b.s_qkv, b.w_qkvh, b.b_qkvh, b.s_qkvh, D, D, nullptr));
            } else if (c->fuse_qkv_attn && c->heads * 80 == D && D % 128 == 0) {   // head dim 80 (ViTPose-H): [q_h | k_h | v_h | 16 zero rows] per head (gemm8.hip EPI_QKV_ATTN)
                const size_t rows = (size_t)c->heads * 256;
                if ((rc = dalloc(c, &b.w_qkvh, rows * D)) || (rc = dalloc(c, &b.b_qkvh, rows)) || (rc = dalloc(c, &b.s_qkvh, rows))) return rc;
                HIPCHK(c, vp::qkv_head_major80_launch(b.w_qkv, b.b_qkv, b.s_qkv, b.w_qkvh, b.b_qkvh, b.s_qkvh, D, D, c->heads, nullptr));
            }
            if ((rc = upload_ln_folded(c, &b.w_fc1, &b.s_fc1, &b.b_fc1, w1, b1, g2, be2, 4 * (size_t)D, D))) return rc;
        } else {
            if ((rc = upload_f32(c, &b.ln1_g, g1, D)) || (rc = upload_f32(c, &b.ln1_b, be1, D)) ||
                (rc = upload_f32(c, &b.ln2_g, g2, D)) || (rc = upload_f32(c, &b.ln2_b, be2, D)))
                return rc;
            if ((rc = upload_mat(c, &b.w_qkv, wq, 3 * (size_t)D, D, pad128(3 * (size_t)D))) || (rc = upload_f32(c, &b.b_qkv, bq, 3 * (size_t)D))) return rc;
            if ((rc = upload_mat(c, &b.w_fc1, w1, 4 * (size_t)D, D, pad128(4 * (size_t)D))) || (rc = upload_f32(c, &b.b_fc1, b1, 4 * (size_t)D))) return rc;
        }
        if ((rc = lk.get(pre + "attn.proj.weight", (int64_t)DD, &p)) || (rc = upload_mat(c, &b.w_proj, p, D, D, pad128(D)))) return rc;
        if (c->y8 && (rc = upload_fp8_rows(c, &b.w_proj8, &b.ws_proj, nullptr, p, nullptr, nullptr, nullptr, D, D))) return rc;
        if ((rc = lk.get(pre + "attn.proj.bias", D, &p)) || (rc = upload_f32(c, &b.b_proj, p, D))) return rc;
        if ((rc = lk.get(pre + "mlp.fc2.weight", (int64_t)4 * DD, &p))) return rc;
        if (c->fp8) { if ((rc = upload_fp8_rows(c, &b.w_fc28, &b.ws_fc2, nullptr, p, nullptr, nullptr, nullptr, D, 4 * (size_t)D))) return rc; }
        else if ((rc = upload_mat(c, &b.w_fc2, p, D, 4 * (size_t)D, pad128(D)))) return rc;
        if ((rc = lk.get(pre + "mlp.fc2.bias", D, &p)) || (rc = upload_f32(c, &b.b_fc2, p, D))) return rc;
    }
    if ((rc = lk.get("backbone.last_norm.weight", D, &p)) || (rc = upload_f32(c, &c->lnf_g, p, D))) return rc;
    if ((rc = lk.get("backbone.last_norm.bias", D, &p)) || (rc = upload_f32(c, &c->lnf_b, p, D))) return rc;
    if ((rc = pack_deconv(c, lk, 0, D, &c->w_d1, &c->b_d1))) return rc;
    if ((rc = pack_deconv(c, lk, 3, 256, &c->w_d2, &c->b_d2))) return rc;
    if ((rc = lk.get("keypoint_head.final_layer.weight", (int64_t)c->Kp * 256, &p)) ||
        (rc = upload_final(c, &c->w_fin, p, c->Kp, 256, &c->fin_rows))) return rc;
    if ((rc = lk.get("keypoint_head.final_layer.bias", c->Kp, &p)) || (rc = upload_f32(c, &c->b_fin, p, c->Kp, pad128(c->Kp)))) return rc;
    {
        std::vector<float> z(pad128(4 * (size_t)D), 0.f);
        if ((rc = upload_f32(c, &c->b_zero, z.data(), z.size()))) return rc;
    }
    HIPCHK(c, hipDeviceSynchronize());
    c->loaded = true;
    return VP_OK;
}

}  // extern "C"
