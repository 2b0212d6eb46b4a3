// C ABI (include/vitpose_hip.h) of the MI355X-native ViTPose hot path:
// context + weight packer + forward orchestration.  No CPU fallback anywhere:
// every compute entry point needs a HIP device and fails with VP_ERR_HIP otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/vitpose_hip.h"
#ifdef VP_TOOLS
#include "../../include/vitpose_hip_tools.h"
#endif
#include "kernels.h"
#include "mx8.h"

namespace {

thread_local std::string g_create_error;

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

struct Block {
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;          // standalone-LayerNorm path only
    uint16_t *w_qkv, *w_proj, *w_fc1, *w_fc2;      // fused path: w_qkv / w_fc1 carry LayerNorm's gamma
    float *b_qkv, *b_proj, *b_fc1, *b_fc2;         // fused path: b_qkv / b_fc1 = W.beta + b
    float *s_qkv, *s_fc1;                          // fused path: row sums of the (rounded) folded weights
    uint16_t* w_qkvh = nullptr;                    // head dim 64: head-major copies for the fused qkv + attention kernel (qkvattn.hip)
    float *b_qkvh = nullptr, *s_qkvh = nullptr;
    // fp8 mode: e4m3 codes [rows padded to 256][K] + one fp32 scale per output channel (LayerNorm's gamma folded into qkv / fc1 first)
    uint8_t *w_qkv8 = nullptr, *w_fc18 = nullptr, *w_fc28 = nullptr, *w_proj8 = nullptr;   // w_proj8: head dim 64 only (the attention kernel's MXFP8 output)
    float *ws_qkv = nullptr, *ws_fc1 = nullptr, *ws_fc2 = nullptr, *ws_proj = nullptr;
};

}  // namespace

struct vp_ctx {
    vp_config cfg;
    int D, L, heads, Kp, dtype, maxb;
    hipStream_t stream = nullptr;
    std::string err;
    bool loaded = false;
    std::vector<void*> allocs;
    // weights
    uint16_t* w_patch = nullptr;
    float* pos = nullptr;
    std::vector<Block> blocks;
    float *lnf_g = nullptr, *lnf_b = nullptr;
    uint16_t *w_d1 = nullptr, *w_d2 = nullptr, *w_fin = nullptr;
    size_t fin_rows = 0;   // physical (hi/lo interleaved) rows of w_fin
    float *b_d1 = nullptr, *b_d2 = nullptr, *b_fin = nullptr, *b_zero = nullptr;
    uint16_t* zero = nullptr;
    // workspaces
    void* in_stage = nullptr;
    int32_t* wh_stage = nullptr;
    float* x = nullptr;
    uint16_t *y = nullptr, *qkv = nullptr, *hid = nullptr, *d1 = nullptr, *d2 = nullptr;
    float *hm = nullptr, *kp = nullptr, *tok = nullptr;
    float* hm_keep = nullptr;         // flip-test: heatmaps of the un-flipped crops while the flipped pass runs
    int32_t* partner = nullptr;       // flip-test: mirror joint per joint
    int g8_stagger = 0;               // gemm8: start delay per XCD in sleep quanta (VP_G8_STAGGER)
    int gemm8_mask = 0x7;             // GEMMs on the 8-phase kernel at large batch: 1 fc2, 2 fc1, 4 qkv, 8 proj (VP_GEMM8; proj measured slower)
    bool persist_gemm = true;         // qkv / fc1 as persistent workgroups at large batch (VP_PERSIST=0: one tile per workgroup)
    int order_mask = 8;               // tile walk last-to-first per GEMM: bit0 qkv, bit1 proj, bit2 fc1, bit3 fc2 (VP_ORDER)
    bool blocked_hid = true;          // mlp hidden activations in the 64x64-blocked layout (VP_BLOCKED_HID=0: row-major)
    bool blocked_qkv = true;          // qkv in the same blocked layout when the head dim is 64 (a (crop, head) slab = three contiguous 8 KiB blocks; VP_BLOCKED_QKV=0: row-major)
    bool fuse_ln = true;              // LayerNorm folded into the GEMMs on both sides of it (VP_FUSE_LN=0: standalone passes)
    bool fuse_qkv_attn = true;        // head dim 64, even batches of >= 128 (pair, head) tiles: attn.qkv + attention core in one kernel (VP_FUSE_QKV_ATTN=0: two launches)
    int g8_bm192 = 3;                 // the 8-phase kernel's 192 x 256 tile is a candidate for: 1 = the residual GEMMs, 2 = the wide GEMMs
    bool g8_cost_model = true;        // tile selection with the round-4 extensions (VP_G8_COST=0: the round-3 thresholds + the 192-row fallback)
    bool deconv_parity_fast = true;   // head: the four output parities of a deconv tile run side by side on one XCD (VP_DECONV_PARITY_FAST=0: parity-major launch order)
    float *ln_part = nullptr, *rowstat = nullptr;   // partial row statistics [M][D/64][2], (mean, rstd) [M][2]
    // fp8 mode (vp_config.dtype = VP_DTYPE_FP8; csrc/mx8.h, gemm8f.hip, quant8.hip): qkv / fc1 / fc2 on MXFP8 operands.  Token rows are
    // padded to Mp (a multiple of the 256-row GEMM tile, >= 512); x8 / xs8 = LayerNorm(x) as MXFP8 codes / scales, hs8 = block scales of
    // the MXFP8 `hid` (its codes live in c->hid)
    bool fp8 = false;
    size_t Mp = 0;
    uint8_t *x8 = nullptr, *xs8 = nullptr, *hs8 = nullptr;
    uint8_t *y8 = nullptr, *ys8 = nullptr;          // head dim 64: the attention output as MXFP8 (A operand of the fp8 attn.proj)
    // asynchronous host path (vp_infer_submit / vp_infer_wait): two slots, each with its own device staging, so that the
    // H2D of call i+1 and the D2H of call i-1 run on the copy stream under the compute of call i
    struct Slot {
        void* in = nullptr; int32_t* wh = nullptr; float* kp = nullptr; hipEvent_t h2d = nullptr, done = nullptr, out = nullptr; bool busy = false;
        // staged download (the group path): the D2H lands in this pinned buffer and vp_infer_wait copies it to the caller's `user_out`,
        // so the submission never blocks on the compute whatever kind of host memory the caller owns
        float* host_kp = nullptr; float* user_out = nullptr; size_t out_bytes = 0;
        // staged upload (the group path with PAGEABLE caller memory): an asynchronous H2D from pageable memory is host-synchronous (the
        // runtime stages it and waits), so the crops go through this pinned buffer in pieces -- host memcpy of piece k+1 under the DMA of piece k
        char* host_in = nullptr; size_t host_in_cap = 0;
    };
    Slot slots[2];
    hipStream_t copy_stream = nullptr;   // H2D of the asynchronous path
    hipStream_t d2h_stream = nullptr;    // D2H on its own stream: an in-order copy stream would hold the next upload behind `wait compute; download`
    int next_slot = 0;
    // small batches: the whole forward + decode of a chunk captured once per (n, input format, source pointer) into a hipGraph and
    // replayed (170+ launches of a few microseconds each are launch-bound below ~16 crops); VP_GRAPH=0 disables
    struct GraphEntry { hipGraphExec_t exec = nullptr; int n = 0, fmt = -1, seen = 0; bool no_graph = false; const void* src = nullptr; const int32_t* wh = nullptr; float* out = nullptr; };
    GraphEntry graphs[4];
    int graph_victim = 0;
    bool fuse_head = true;            // VP_FUSE_HEAD=0: deconv2 and the final 1x1 conv as two launches at every batch size
    int graph_max_n = 16;
    int graph_max_n_stats = 8;        // batches of <= this many crops: the consumer GEMMs (qkv, fc1) merge the LayerNorm partial statistics of their tile rows
                                      // themselves (once per row and tile, in the prologue: gemm.hip) and the 2 x depth ln_finalize launches disappear -- same
                                      // ln_merge, bit-identical.  Measured (profiles/fold_stats_r3.txt): -7...-12 % per step at 1-8 crops, +0...+20 % at
                                      // 16-48 (every column tile merges its rows again): threshold 8.  VP_FOLD_STATS=n moves it (0 = always ln_finalize).
                                      // Round 2 merged per LANE in the epilogue (16 x redundant): slower than ln_finalize even at 8 crops (3.89 vs 2.97 ms).
    hipEvent_t ev_in = nullptr, ev_out = nullptr;   // vp_infer_device_stream: ordering against the caller's stream
    // vp_infer_device_stream at small batches (round 5): the launches go onto the CALLER's stream (c->stream points at it for the duration of that call) instead of
    // being fenced against it with two cross-stream events per call (~0.1 ms at 1-16 crops).  The handle's workspaces are then used from more than one stream over
    // time: `adopt_stream` orders a call behind the previous one whenever the stream changes.
    hipStream_t own_stream = nullptr;       // the handle's compute stream (== stream outside that call)
    const void* last_stream_id = nullptr;   // identity of the caller's stream the workspaces were last used on (compared, never dereferenced: the caller may have destroyed it)
    bool foreign_pending = false;           // the last user was a caller's stream: ev_sw, recorded behind its launches, is what work on any other stream waits for
    hipEvent_t ev_sw = nullptr;
    int caller_stream_max_n = 16;           // batches up to this many crops take that path (VP_CALLER_STREAM=0: off)
    uint8_t* frame_stage = nullptr;   // device copy of the current video frame (vp_infer_frame)
    size_t frame_cap = 0;
    int32_t* cparams = nullptr;       // per-crop geometry [max_batch, 8]
    // profiling
    uint32_t prof = 0;   // bit f = time kernel family f
    int gemm_variant[VP_PROF_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};   // tile cfg per GEMM family, -1 = default rule
    int gemm_group_m[VP_PROF_COUNT] = {0};
    int gemm_ablate = 0;   // profiling only
    int fam_ablate[VP_PROF_COUNT] = {0};   // VP_TOOLS: per-family ablation / experiment bits in the forward pass (VP_ABLATE_FAM="fam:bits,...")
    struct Ev { hipEvent_t a, b; int fam; double flops, bytes; };
    std::vector<Ev> evs;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    vp_profile acc{};
    std::string kernel_desc[VP_PROF_COUNT];   // name of the kernel the last launch of each family resolved to (vp_profile_kernel)
};

namespace {

int fail(vp_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HIPCHK(c, expr)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return fail((c), VP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));   \
    } while (0)

template <class T> int dalloc(vp_ctx* c, T** p, size_t count) {
    void* q = nullptr;
    HIPCHK(c, hipMalloc(&q, count * sizeof(T) + 256));
    c->allocs.push_back(q);
    *p = (T*)q;
    return VP_OK;
}

int upload_f32(vp_ctx* c, float** dst, const float* src, size_t n, size_t npad = 0) {
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

struct Lookup {
    std::unordered_map<std::string, const vp_tensor_desc*> map;
    vp_ctx* c;
    int get(const std::string& name, int64_t numel, const float** out) {
        auto it = map.find(name);
        if (it == map.end()) return fail(c, VP_ERR_MISSING_TENSOR, "missing key in state dict: " + name);
        if (it->second->numel != numel || it->second->data == nullptr)
            return fail(c, VP_ERR_SHAPE, "size mismatch for " + name + ": expected " + std::to_string(numel) +
                                             " elements, got " + std::to_string(it->second->numel));
        *out = it->second->data;
        return VP_OK;
    }
};

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

#define LAUNCH(c, fam, flops, bytes, expr)   \
    do {                                     \
        const bool on__ = prof_begin((c), (fam), (flops), (bytes)); \
        hipError_t e__ = (expr);             \
        prof_end((c), on__);                 \
        if (e__ != hipSuccess)               \
            return fail((c), VP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

struct LnFuse {
    bool a_blocked = false, out_blocked = false, reverse = false;   // 64x64-blocked activation layout on the A / output side (kernels.h)
    size_t plane = 0;                 // producer: elements between the hi and lo planes of the residual stream
    float* stats_out = nullptr;       // producer: partial row statistics
    const float* rowstat = nullptr;   // consumer: (mean, rstd) per row
    const float* ln_s = nullptr;      // consumer: row sums of the folded weights
    const float* ln_part = nullptr;   // consumer at small batch: the producer's partial statistics instead of rowstat
    int ln_tiles = 0;
    int* tiles_out = nullptr;         // producer: number of n-tiles written per row
};

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
struct G8Pick { int variant, bm, bn; long tiles; };
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
struct Tile2Pick { int variant, group_m; };
Tile2Pick pick_gemm2_tile(int epi, int M, int N, int K) {
    Tile2Pick tp;
    tp.variant = (epi == vp::EPI_BIAS_RESID || epi == vp::EPI_BIAS_RESID_LN) ? 11 : 8;
    tp.group_m = (epi == vp::EPI_BIAS || epi == vp::EPI_BIAS_GELU) ? 8 : 0;
    const long par_ = (epi == vp::EPI_DECONV) ? 4 : 1;   // the four output parities of a deconv are four GEMMs of one launch
    const long t192 = (long)((M + 191) / 192) * ((N + 127) / 128) * par_;
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * par_;
    if (t192 >= 384) return tp;
    tp.variant = (t128 >= 256) ? 1 : 9;
    tp.group_m = 0;
    if (tp.variant == 9) {
        const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64) * par_;
        const long t32 = (long)((M + 31) / 32) * ((N + 63) / 64) * par_;
        const long t128x64 = (long)((M + 127) / 128) * ((N + 63) / 64) * par_;
        if (K % 128 == 0 && t32 <= 256) tp.variant = 31;
        else if (K % 128 == 0 && t64 <= 256) tp.variant = 30;
        else if (t64 <= 512) tp.variant = 12;
        else if (epi == vp::EPI_BIAS_RESID_LN && t128x64 <= 512) tp.variant = 15;
        else if (K >= 2048) tp.variant = 12;
    }
    return tp;
}

int gemm(vp_ctx* c, int fam, int epi, const uint16_t* A, const uint16_t* W, const float* bias, void* out,
         const float* aux, int M, int N, int K, int ldo, int Hin = 0, int Win = 0, int Cin = 0, const LnFuse* ln = nullptr) {
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

// forward of one chunk (n <= max_batch) with device-resident crops; heatmaps land in c->hm
int forward_chunk(vp_ctx* c, const void* d_crops, int fmt, int n, bool want_tokens, bool flip = false) {
    const int D = c->D, M = n * 192;
    const double in_b = (fmt == VP_INPUT_F32_NCHW ? 4.0 : 1.0) * n * 3.0 * 256 * 192;
    LAUNCH(c, VP_PROF_IM2COL, 0.0, in_b + 2.0 * M * 768, vp::im2col_launch(c->dtype, d_crops, fmt, c->hid, n, c->stream, flip));
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
        const bool fold_stats = n <= c->graph_max_n_stats;
        auto finalize = [&]() -> int {
            if (fold_stats) return VP_OK;
            LAUNCH(c, VP_PROF_LAYERNORM, 0.0, 8.0 * M * tiles + 8.0 * M, vp::ln_finalize_launch(c->ln_part, c->rowstat, M, tiles, D, c->stream));
            return VP_OK;
        };
        if ((rc = gemm(c, VP_PROF_GEMM_PATCH, vp::EPI_POS_LN, c->hid, c->w_patch, c->b_zero, c->x, c->pos, M, D, 768, D, 0, 0, 0, &prod))) return rc;
        if ((rc = finalize())) return rc;
        for (int l = 0; l < c->L; ++l) {
            const Block& b = c->blocks[l];
            LnFuse cq; cq.rowstat = c->rowstat; cq.ln_s = b.s_qkv; cq.reverse = (c->order_mask & 1) != 0; cq.out_blocked = qkv_blocked;
            if (fold_stats) { cq.rowstat = nullptr; cq.ln_part = c->ln_part; cq.ln_tiles = D / 64; }
            // attn.qkv + attention core as ONE kernel per (pair of crops, head) from 128 tiles on (qkvattn.hip; bit-identical y; an odd batch's last crop fills both halves of its pair)
            static const long qa_min_tiles = [] { const char* e = getenv("VP_QA_MIN_TILES"); return e ? atol(e) : 128L; }();   // the fused kernel wins from 128 tiles on (measured sweep 128 - 1536 tiles: profiles/qkvattn_r4.txt)
            vp::QkvAttnArgs qa{};
            qa.x_hi = xh; qa.wh = b.w_qkvh; qa.bh = b.b_qkvh; qa.sh = b.s_qkvh; qa.rowstat = c->rowstat; qa.y = c->y;
            qa.npairs = (n + 1) / 2; qa.ncrops = n; qa.heads = c->heads; qa.D = D;   // odd n: the last crop fills both halves of its pair
            qa.scale_log2e = (1.0f / sqrtf(64.0f)) * 1.4426950408889634f;
            // (a shape the fused kernel rejects -- a chunk beyond its 32-bit row offsets, fewer than 8 tiles under a lowered VP_QA_MIN_TILES -- falls through
            // to the gemm + attention pair below, the way gemm() falls back when gemm8_supported says no: ADVICE r4)
            // head dim 80: one crop x one head per 192 x 256 tile of the 8-phase kernel (gemm8.hip EPI_QKV_ATTN; bit-identical y), from 192 tiles on
            static const long qa80_min_tiles = [] { const char* e = getenv("VP_QA80_MIN_TILES"); return e ? atol(e) : 192L; }();   // wins from 12 crops x 16 heads on (profiles/qkvattn80_r5.txt)
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
            if (g80.A && !fold_stats && (long)n * c->heads >= qa80_min_tiles && c->gemm_variant[VP_PROF_GEMM_QKV] < 0 && vp::gemm8_supported(vp::EPI_QKV_ATTN, g80, 256, 192)) {
                char desc[192];
                desc[0] = 0;
                g80.desc = desc; g80.desc_cap = (int)sizeof(desc);
                LAUNCH(c, VP_PROF_GEMM_QKV, 2.0 * M * 3.0 * D * D + 4.0 * 192 * 192 * (double)D * n, 2.0 * M * D + 2.0 * 3 * D * (double)D + 2.0 * M * D,
                       vp::gemm_launch(c->dtype, vp::EPI_QKV_ATTN, g80, c->stream));
                if (desc[0] && c->kernel_desc[VP_PROF_GEMM_QKV] != desc) c->kernel_desc[VP_PROF_GEMM_QKV] = desc;
            } else if (b.w_qkvh && c->heads * 64 == D && !fold_stats && (long)((n + 1) / 2) * c->heads >= qa_min_tiles && c->gemm_variant[VP_PROF_GEMM_QKV] < 0 && vp::qkvattn_supported(qa)) {
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
            if ((rc = finalize())) return rc;
            LnFuse c1; c1.rowstat = c->rowstat; c1.ln_s = b.s_fc1; c1.out_blocked = c->blocked_hid; c1.reverse = (c->order_mask & 4) != 0;
            if (fold_stats) { c1.rowstat = nullptr; c1.ln_part = c->ln_part; c1.ln_tiles = D / 64; }
            if ((rc = gemm(c, VP_PROF_GEMM_FC1, vp::EPI_BIAS_GELU, xh, b.w_fc1, b.b_fc1, c->hid, nullptr, M, 4 * D, D, 4 * D, 0, 0, 0, &c1))) return rc;
            LnFuse p2 = prod; p2.a_blocked = c->blocked_hid; p2.reverse = (c->order_mask & 8) != 0;
            if ((rc = gemm(c, VP_PROF_GEMM_FC2, vp::EPI_BIAS_RESID_LN, c->hid, b.w_fc2, b.b_fc2, c->x, c->x, M, D, 4 * D, D, 0, 0, 0, &p2))) return rc;
            if (l + 1 < c->L && (rc = finalize())) return rc;   // last block: last_norm below is a standalone pass
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
    LAUNCH(c, VP_PROF_LAYERNORM, 0.0, 6.0 * M * D,
           vp::layernorm_launch(c->dtype, c->x, c->lnf_g, c->lnf_b, c->y, want_tokens ? c->tok : nullptr, M, D, c->stream, plane));
    // head: tokens [n,16,12,D] (NHWC view of [n*192, D]) -> [n,32,24,256] -> [n,64,48,256] -> heatmaps [n,Kp,64,48]
    if ((rc = gemm(c, VP_PROF_GEMM_DECONV, vp::EPI_DECONV, c->y, c->w_d1, c->b_d1, c->d1, nullptr, n * 192, 256, 4 * D, 256, 16, 12, D))) return rc;
    // large batches: the final 1x1 conv rides in deconv2's epilogue (gemm.hip EPI_DECONV_FINAL, bit-identical heatmaps) and the
    // [n,64,48,256] tensor is never written; small batches keep the two launches on tiles that still fill 256 CUs
    if (c->fuse_head && c->gemm_variant[VP_PROF_GEMM_DECONV] < 0 && (long)n * 12 >= 512)
        return gemm(c, VP_PROF_GEMM_DECONV, vp::EPI_DECONV_FINAL, c->d1, c->w_d2, c->b_d2, nullptr, nullptr, n * 768, 256, 1024, 256, 32, 24, 256);
    if ((rc = gemm(c, VP_PROF_GEMM_DECONV, vp::EPI_DECONV, c->d1, c->w_d2, c->b_d2, c->d2, nullptr, n * 768, 256, 1024, 256, 32, 24, 256))) return rc;
    if ((rc = gemm(c, VP_PROF_GEMM_FINAL, vp::EPI_HEATMAP, c->d2, c->w_fin, c->b_fin, c->hm, nullptr, n * 3072, (int)c->fin_rows, 256, 0))) return rc;
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
    if (nb > c->graph_max_n || c->prof != 0 || c->stream == nullptr) return eager();   // (a caller's legacy default stream: plain launches; graph replay and eager launches run the same when calls are enqueued back to back)
    vp_ctx::GraphEntry* ge = nullptr;
    for (auto& g : c->graphs)
        if (g.n == nb && g.fmt == fmt && g.src == d_src && g.wh == d_wh && g.out == d_out) { ge = &g; break; }
    if (ge && ge->exec) {
        HIPCHK(c, hipGraphLaunch(ge->exec, c->stream));
        return VP_OK;
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
    const size_t B = (size_t)c->maxb;
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
    if (const char* f = getenv("VP_FOLD_STATS")) c->graph_max_n_stats = atoi(f);
    if (const char* f = getenv("VP_BLOCKED_QKV")) c->blocked_qkv = atoi(f) != 0;
    if (const char* f = getenv("VP_FUSE_QKV_ATTN")) c->fuse_qkv_attn = atoi(f) != 0;
    if (const char* f = getenv("VP_DECONV_PARITY_FAST")) c->deconv_parity_fast = atoi(f) != 0;
    if (const char* f = getenv("VP_G8_COST")) c->g8_cost_model = atoi(f) != 0;
    if (const char* f = getenv("VP_G8_BM192")) c->g8_bm192 = atoi(f);   // mask: 1 = residual GEMMs, 2 = wide GEMMs may take the 192 x 256 tile of the 8-phase kernel (0: never)
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
    if ((rc = dalloc(c, &c->zero, (size_t)256))) return bail(rc);
    if (hipMemset(c->zero, 0, 512) != hipSuccess) { c->err = "hipMemset"; return bail(VP_ERR_HIP); }
    *out = c;
    return VP_OK;
}

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
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    std::memset(&c->acc, 0, sizeof(c->acc));
    return VP_OK;
}

int vp_get_profile(vp_handle c, vp_profile* out) {
    if (!c || !out) return VP_ERR_INVALID;
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

// ---------------------------------------------------------------------------
// Debug / parity taps: run ONE kernel on host fp32 data (operands are rounded to
// `dtype` exactly as the production packer / producers do).  Used by tests/ only.
// ---------------------------------------------------------------------------
namespace {
vp_ctx* dbg_ctx(int device, int dtype) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        g_create_error = "no HIP device available (no CPU fallback)";
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    vp_ctx* c = new vp_ctx();
    c->cfg.device_id = device;
    c->dtype = dtype == VP_DTYPE_F16 ? vp::DT_F16 : vp::DT_BF16;
    apply_gemm_tuning(c);
    return c;
}
int dbg_finish(vp_ctx* c, int rc) {
    if (rc) g_create_error = c->err;
    vp_destroy(c);
    return rc;
}
// device 16-bit -> host fp32
int download16(vp_ctx* c, const uint16_t* d, float* out, size_t n) {
    std::vector<uint16_t> t(n);
    HIPCHK(c, hipMemcpy(t.data(), d, n * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
        if (c->dtype == vp::DT_BF16) {
            uint32_t u = (uint32_t)t[i] << 16;
            std::memcpy(&out[i], &u, 4);
        } else {
            const uint32_t h = t[i], sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff;
            uint32_t u;
            if (e == 0) {
                if (m == 0) u = sign;
                else { int sh = 0; uint32_t mm = m; while (!(mm & 0x400)) { mm <<= 1; ++sh; }
                       u = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3ff) << 13); }
            } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
            else u = sign | ((e + 112) << 23) | (m << 13);
            std::memcpy(&out[i], &u, 4);
        }
    }
    return VP_OK;
}
}  // namespace

extern "C" {

// out = epilogue(A[M,K] . W[N,K]^T): epi 0 bias->16bit, 1 bias+gelu->16bit, 2 bias+aux[M,N]->fp32, 3 aux[m%192]->fp32
VP_API int vp_dbg_gemm(int32_t device, int32_t dtype, int32_t epi, int32_t M, int32_t N, int32_t K, const float* A,
                       const float* W, const float* bias, const float* aux, float* out) {
    if (epi < 0 || epi > 3 || M <= 0 || N <= 0 || K <= 0 || K % 64) return fail(nullptr, VP_ERR_INVALID, "bad gemm test shape");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    uint16_t *dA, *dW, *dO16 = nullptr;
    float *dB, *dAux = nullptr, *dO32 = nullptr;
    int rc;
    const size_t MN = (size_t)M * N;
    if ((rc = upload_mat(c, &dA, A, M, K, M))) return dbg_finish(c, rc);
    if ((rc = upload_mat(c, &dW, W, N, K, pad128(N)))) return dbg_finish(c, rc);
    if ((rc = upload_f32(c, &dB, bias, N, pad128(N)))) return dbg_finish(c, rc);
    if (epi >= 2) {
        if ((rc = upload_f32(c, &dAux, aux, epi == 2 ? MN : (size_t)192 * N))) return dbg_finish(c, rc);
        if ((rc = dalloc(c, &dO32, MN))) return dbg_finish(c, rc);
    } else if ((rc = dalloc(c, &dO16, MN))) return dbg_finish(c, rc);
    if ((rc = dalloc(c, &c->zero, (size_t)256))) return dbg_finish(c, rc);
    rc = gemm(c, 0, epi, dA, dW, dB, epi >= 2 ? (void*)dO32 : (void*)dO16, dAux, M, N, K, N);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(c, VP_ERR_HIP, "gemm kernel failed");
    if (!rc) {
        if (epi >= 2) { if (hipMemcpy(out, dO32, MN * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(c, VP_ERR_HIP, "D2H"); }
        else rc = download16(c, dO16, out, MN);
    }
    return dbg_finish(c, rc);
}

// qkv [B*192, 3*D] fp32 -> out [B*192, D] fp32 (attention core, vit.py:167-176)
VP_API int vp_dbg_attention(int32_t device, int32_t dtype, int32_t B, int32_t D, int32_t heads, const float* qkv, float* out) {
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    uint16_t *dq, *dout;
    int rc;
    const size_t M = (size_t)B * 192;
    if ((rc = upload_mat(c, &dq, qkv, M, 3 * (size_t)D, M))) return dbg_finish(c, rc);
    if ((rc = dalloc(c, &dout, M * D))) return dbg_finish(c, rc);
    hipError_t e = vp::attention_launch(c->dtype, dq, dout, B, D, heads, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("attention: ") + hipGetErrorString(e)));
    return dbg_finish(c, download16(c, dout, out, M * D));
}

// attn.qkv + attention core in one kernel (qkvattn.hip): x [2 npairs 192, D] (rounded to dtype), Wqkv [3D, D], bias [3D] -> out [M, D] (as fp32).
// Run with neutral LayerNorm statistics (mean 0, rstd 1, row sums 0: ln_fold(acc, 0, 0, 1, b) == acc + b exactly), so the result must equal
// vp_dbg_gemm(epi 0) followed by vp_dbg_attention bit for bit.
VP_API int vp_dbg_qkvattn(int32_t device, int32_t dtype, int32_t npairs, int32_t D, int32_t heads, const float* x, const float* W, const float* bias, float* out) {
    if (npairs <= 0 || D <= 0 || heads <= 0 || !x || !W || !bias || !out) return fail(nullptr, VP_ERR_INVALID, "bad qkvattn test shape");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    const size_t M = (size_t)npairs * 384;
    uint16_t *dx, *dw, *dwh, *dy;
    float *db, *dbh, *ds, *dsh, *drow;
    int rc;
    std::vector<float> zeros(3 * (size_t)D, 0.f), row(2 * M);
    for (size_t m = 0; m < M; ++m) { row[2 * m] = 0.f; row[2 * m + 1] = 1.f; }
    if ((rc = upload_mat(c, &dx, x, M, D, M)) || (rc = upload_mat(c, &dw, W, 3 * (size_t)D, D, pad128(3 * (size_t)D))) || (rc = upload_f32(c, &db, bias, 3 * (size_t)D)) ||
        (rc = upload_f32(c, &ds, zeros.data(), 3 * (size_t)D)) || (rc = upload_f32(c, &drow, row.data(), 2 * M)) || (rc = dalloc(c, &dwh, 3 * (size_t)D * D)) ||
        (rc = dalloc(c, &dbh, 3 * (size_t)D)) || (rc = dalloc(c, &dsh, 3 * (size_t)D)) || (rc = dalloc(c, &dy, M * D)))
        return dbg_finish(c, rc);
    if (heads * 80 == D) {   // head dim 80: gemm8.hip EPI_QKV_ATTN on the 192 x 256 tile (one crop x one head), head-major weights of heads * 256 rows
        uint16_t* dwh80; float *dbh80, *dsh80;
        const size_t rows = (size_t)heads * 256;
        if ((rc = dalloc(c, &dwh80, rows * D)) || (rc = dalloc(c, &dbh80, rows)) || (rc = dalloc(c, &dsh80, rows))) return dbg_finish(c, rc);
        hipError_t e8 = vp::qkv_head_major80_launch(dw, db, ds, dwh80, dbh80, dsh80, D, D, heads, nullptr);
        vp::GemmArgs g80{};
        g80.A = dx; g80.W = dwh80; g80.bias = dbh80; g80.ln_s = dsh80; g80.rowstat = drow; g80.out = dy;
        g80.M = (int)M; g80.N = heads * 256; g80.K = D; g80.ldo = D; g80.w_rows = heads * 256; g80.variant = 18;
        g80.attn_scale_log2e = (1.0f / sqrtf(80.0f)) * 1.4426950408889634f;
        if (e8 == hipSuccess && !vp::gemm8_supported(vp::EPI_QKV_ATTN, g80, 256, 192)) return dbg_finish(c, fail(c, VP_ERR_INVALID, "shape not supported by the fused qkv + attention tile (head dim 80)"));
        if (e8 == hipSuccess) e8 = vp::gemm_launch(c->dtype, vp::EPI_QKV_ATTN, g80, nullptr);
        if (e8 == hipSuccess) e8 = hipDeviceSynchronize();
        if (e8 != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("qkvattn (head dim 80): ") + hipGetErrorString(e8)));
        return dbg_finish(c, download16(c, dy, out, M * D));
    }
    hipError_t e = vp::qkv_head_major_launch(dw, db, ds, dwh, dbh, dsh, D, D, nullptr);
    vp::QkvAttnArgs qa{};
    qa.x_hi = dx; qa.wh = dwh; qa.bh = dbh; qa.sh = dsh; qa.rowstat = drow; qa.y = dy; qa.npairs = npairs; qa.ncrops = 2 * npairs; qa.heads = heads; qa.D = D;
    const float scale = 1.0f / sqrtf(64.0f);
    qa.scale_log2e = scale * 1.4426950408889634f;
    if (e == hipSuccess && !vp::qkvattn_supported(qa)) return dbg_finish(c, fail(c, VP_ERR_INVALID, "shape not supported by the fused qkv + attention kernel"));
    if (e == hipSuccess) e = vp::qkvattn_launch(c->dtype, qa, nullptr, nullptr, 0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("qkvattn: ") + hipGetErrorString(e)));
    return dbg_finish(c, download16(c, dy, out, M * D));
}

#ifdef VP_TOOLS
// tools/qkvattn_phases.py: average milliseconds of the fused qkv + attention kernel on random operands, optionally with phases compiled out
VP_API int vp_dbg_qkvattn_bench(int32_t device, int32_t npairs, int32_t D, int32_t heads, int32_t iters, int32_t ablate, float* ms_out) {
    vp_ctx* c = dbg_ctx(device, VP_DTYPE_F16);
    if (!c) return VP_ERR_HIP;
    const size_t M = (size_t)npairs * 384;
    uint16_t *dx, *dwh, *dy;
    float *dbh, *dsh, *drow;
    int rc;
    if ((rc = dalloc(c, &dx, M * D)) || (rc = dalloc(c, &dwh, 3 * (size_t)D * D)) || (rc = dalloc(c, &dy, M * D)) || (rc = dalloc(c, &dbh, 3 * (size_t)D)) ||
        (rc = dalloc(c, &dsh, 3 * (size_t)D)) || (rc = dalloc(c, &drow, 2 * M)))
        return dbg_finish(c, rc);
    vp::fill_random16(c->dtype, dx, M * D, 1u, nullptr);
    vp::fill_random16(c->dtype, dwh, 3 * (size_t)D * D, 2u, nullptr);
    hipMemset(dbh, 0, 3 * (size_t)D * 4); hipMemset(dsh, 0, 3 * (size_t)D * 4); hipMemset(drow, 0, 2 * M * 4);
    vp::QkvAttnArgs qa{};
    qa.x_hi = dx; qa.wh = dwh; qa.bh = dbh; qa.sh = dsh; qa.rowstat = drow; qa.y = dy; qa.npairs = npairs; qa.ncrops = 2 * npairs; qa.heads = heads; qa.D = D; qa.ablate = ablate;
    qa.scale_log2e = 0.125f * 1.4426950408889634f;
    vp::GemmArgs g80{};   // head dim 80: gemm8.hip EPI_QKV_ATTN (heads * 256 head-major rows: the 3 D^2 buffer is larger than heads * 256 * D)
    const bool h80 = heads * 80 == D;
    g80.A = dx; g80.W = dwh; g80.bias = dbh; g80.ln_s = dsh; g80.rowstat = drow; g80.out = dy;
    g80.M = (int)M; g80.N = heads * 256; g80.K = D; g80.ldo = D; g80.w_rows = heads * 256; g80.variant = 18; g80.ablate = ablate;
    g80.attn_scale_log2e = (1.0f / sqrtf(80.0f)) * 1.4426950408889634f;
    auto launch = [&]() { return h80 ? vp::gemm_launch(c->dtype, vp::EPI_QKV_ATTN, g80, nullptr) : vp::qkvattn_launch(c->dtype, qa, nullptr, nullptr, 0); };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = launch();
    hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("qkvattn bench: ") + hipGetErrorString(e)));
    return dbg_finish(c, VP_OK);
}
#endif

// LayerNorm(eps 1e-6): x [M,D] fp32 -> out16 (as fp32) [M,D] and out32 [M,D]
VP_API int vp_dbg_layernorm(int32_t device, int32_t dtype, int32_t M, int32_t D, const float* x, const float* gamma,
                            const float* beta, float* out16, float* out32) {
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    float *dx, *dg, *db, *d32;
    uint16_t* d16;
    int rc;
    const size_t MD = (size_t)M * D;
    if ((rc = upload_f32(c, &dx, x, MD)) || (rc = upload_f32(c, &dg, gamma, D)) || (rc = upload_f32(c, &db, beta, D)) ||
        (rc = dalloc(c, &d32, MD)) || (rc = dalloc(c, &d16, MD))) return dbg_finish(c, rc);
    hipError_t e = vp::layernorm_launch(c->dtype, dx, dg, db, d16, d32, M, D, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out32, d32, MD * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("layernorm: ") + hipGetErrorString(e)));
    return dbg_finish(c, download16(c, d16, out16, MD));
}

// ConvTranspose2d(Cin,256,4,2,1,bias=False)+BN(eval)+ReLU on NHWC x [B,Hin,Win,Cin] fp32 -> NHWC [B,2Hin,2Win,256] fp32.
// tensors = {"keypoint_head.deconv_layers.0.weight", ".1.weight", ".1.bias", ".1.running_mean", ".1.running_var"}
VP_API int vp_dbg_deconv(int32_t device, int32_t dtype, int32_t B, int32_t Hin, int32_t Win, int32_t Cin, const float* x,
                         const vp_tensor_desc* tensors, int32_t n_tensors, float* out) {
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    Lookup lk;
    lk.c = c;
    for (int i = 0; i < n_tensors; ++i) lk.map[tensors[i].name] = &tensors[i];
    uint16_t *dx, *dw, *dout;
    float* db;
    int rc;
    const size_t Min = (size_t)B * Hin * Win;
    if ((rc = pack_deconv(c, lk, 0, Cin, &dw, &db))) return dbg_finish(c, rc);
    if ((rc = upload_mat(c, &dx, x, Min, Cin, Min))) return dbg_finish(c, rc);
    if ((rc = dalloc(c, &dout, Min * 4 * 256))) return dbg_finish(c, rc);
    if ((rc = dalloc(c, &c->zero, (size_t)256))) return dbg_finish(c, rc);
    if (hipMemset(c->zero, 0, 512) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "memset"));
    rc = gemm(c, 0, vp::EPI_DECONV, dx, dw, db, dout, nullptr, (int)Min, 256, 4 * Cin, 256, Hin, Win, Cin);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(c, VP_ERR_HIP, "deconv kernel failed");
    if (!rc) rc = download16(c, dout, out, Min * 4 * 256);
    return dbg_finish(c, rc);
}


#ifdef VP_TOOLS
// tools/gemm_timeline.py: one persistent launch of the qkv / fc1 shape with per-tile phase stamps (shader cycles) of wave 0 of
// every workgroup: stamps[wg][tile][8] = (main loop start, main loop end, epilogue end, 5 stamps inside k-step 5: top, after the
// barrier, after the global_load_lds issues, after the first MFMA block, end), up to 32 tiles per workgroup.
VP_API int vp_dbg_gemm_timeline(int32_t device, int32_t dtype, int32_t epi, int32_t M, int32_t N, int32_t K, uint64_t* stamps,
                                int32_t max_wg) {
    if ((epi != 0 && epi != 1) || !stamps) return fail(nullptr, VP_ERR_INVALID, "bad timeline request");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    uint16_t *dA, *dW, *dO;
    float* dB;
    unsigned long long* dS;
    int rc;
    const size_t wrows = pad128(N), nst = (size_t)max_wg * 32 * 8;
    if ((rc = dalloc(c, &dA, (size_t)M * K)) || (rc = dalloc(c, &dW, wrows * K)) || (rc = dalloc(c, &dB, wrows)) ||
        (rc = dalloc(c, &dO, (size_t)M * N)) || (rc = dalloc(c, &dS, nst)) || (rc = dalloc(c, &c->zero, (size_t)256)))
        return dbg_finish(c, rc);
    vp::fill_random16(c->dtype, dA, (size_t)M * K, 1u, nullptr);
    vp::fill_random16(c->dtype, dW, wrows * K, 2u, nullptr);
    hipMemset(dB, 0, wrows * 4);
    hipMemset(dS, 0, nst * 8);
    vp::GemmArgs g{};
    g.A = dA; g.W = dW; g.bias = dB; g.out = dO; g.M = M; g.N = N; g.K = K; g.ldo = N; g.zero = c->zero;
    g.w_rows = (int)wrows; g.variant = 8; g.group_m = 8; g.persist = 1;
    hipError_t e = vp::gemm_launch(c->dtype, epi, g, nullptr);          // warm
    g.ablate = 32 | (getenv("VP_TL_ABL") ? atoi(getenv("VP_TL_ABL")) : 0); g.stats_out = (float*)dS;
    if (e == hipSuccess) e = vp::gemm_launch(c->dtype, epi, g, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(stamps, dS, nst * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("timeline: ") + hipGetErrorString(e)));
    return dbg_finish(c, VP_OK);
}
#endif  // VP_TOOLS

// Time `iters` launches of one GEMM configuration on random device operands (HIP events).
// epi as in vp_dbg_gemm (0..3); returns average milliseconds per launch in *ms_out.
VP_API int vp_dbg_gemm_bench(int32_t device, int32_t dtype, int32_t epi, int32_t variant, int32_t group_m, int32_t M,
                             int32_t N, int32_t K, int32_t iters, float* ms_out) {
    if (epi < 0 || epi > 3 || M <= 0 || N <= 0 || K <= 0 || K % 64 || iters <= 0 || !ms_out)
        return fail(nullptr, VP_ERR_INVALID, "bad gemm bench shape");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    uint16_t *dA, *dW, *dO16 = nullptr;
    float *dB, *dAux = nullptr, *dO32 = nullptr;
    int rc;
    const size_t MN = (size_t)M * N, wrows = pad128(N);
    if ((rc = dalloc(c, &dA, (size_t)M * K)) || (rc = dalloc(c, &dW, wrows * K)) || (rc = dalloc(c, &dB, wrows)) ||
        (rc = dalloc(c, &c->zero, (size_t)256)))
        return dbg_finish(c, rc);
    if (epi >= 2) { if ((rc = dalloc(c, &dO32, MN)) || (rc = dalloc(c, &dAux, (size_t)192 * N))) return dbg_finish(c, rc); }
    else if ((rc = dalloc(c, &dO16, MN))) return dbg_finish(c, rc);
    vp::fill_random16(c->dtype, dA, (size_t)M * K, 1u, nullptr);
    vp::fill_random16(c->dtype, dW, wrows * K, 2u, nullptr);
    hipMemset(dB, 0, wrows * 4);
    if (dO32) hipMemset(dO32, 0, MN * 4);
    if (dAux) hipMemset(dAux, 0, (size_t)192 * N * 4);
    c->gemm_variant[0] = variant & 0xff;
    c->gemm_group_m[0] = group_m;
    c->gemm_ablate = variant >> 8;   // tools only: ablation flags in the high bits
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    void* outp = epi >= 2 ? (void*)dO32 : (void*)dO16;
    const float* aux = epi == 2 ? dO32 : dAux;
    for (int i = 0; i < 2 && !rc; ++i) rc = gemm(c, 0, epi, dA, dW, dB, outp, aux, M, N, K, N);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters && !rc; ++i) rc = gemm(c, 0, epi, dA, dW, dB, outp, aux, M, N, K, N);
    hipEventRecord(e1, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(c, VP_ERR_HIP, "gemm bench kernel failed");
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return dbg_finish(c, rc);
}


// ---- production-configuration GEMM taps (tests/test_gpu_gemm_cfgs.py, tools/gemm8_check.py) ----
}  // extern "C"
namespace {
// a GEMM launch on RANDOM device operands in any production configuration: epi = kernels.h GemmEpi 0, 1 (optionally with the
// LayerNorm-consumer fold), 2, 3, 6; flags: 1 persist, 2 out_blocked, 4 a_blocked, 8 reverse, 16 LayerNorm-consumer fold
struct RandCase {
    vp::GemmArgs g{};
    size_t out_bytes = 0, stats_floats = 0;
    void* out[2] = {nullptr, nullptr};
    float* stats[2] = {nullptr, nullptr};
};
int make_rand_case(vp_ctx* c, RandCase& rc, int epi, int flags, int M, int N, int K, int nout) {
    uint16_t *dA, *dW, *dAux16 = nullptr;
    float *dB, *dAux32 = nullptr, *dRow = nullptr, *dS = nullptr;
    int r;
    const size_t MN = (size_t)M * N, wrows = pad128(N);
    if ((r = dalloc(c, &dA, (size_t)M * K)) || (r = dalloc(c, &dW, wrows * K)) || (r = dalloc(c, &c->zero, (size_t)256))) return r;
    vp::fill_random16(c->dtype, dA, (size_t)M * K, 1u, nullptr);
    vp::fill_random16(c->dtype, dW, wrows * K, 2u, nullptr);
    std::vector<float> hb(wrows), hs(wrows), hr((size_t)M * 2);
    uint32_t lcg = 12345u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((lcg >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hb) v = rnd();
    for (auto& v : hs) v = 4.f * rnd();
    for (size_t i = 0; i < (size_t)M; ++i) { hr[2 * i] = 0.2f * rnd(); hr[2 * i + 1] = 1.f + 0.4f * rnd(); }
    if ((r = upload_f32(c, &dB, hb.data(), wrows))) return r;
    vp::GemmArgs& g = rc.g;
    g.A = dA; g.W = dW; g.bias = dB; g.M = M; g.N = N; g.K = K; g.ldo = N; g.zero = c->zero; g.Kp = c->Kp;
    g.w_rows = (int)wrows;
    g.persist = (flags & 1) != 0; g.out_blocked = (flags & 2) != 0; g.a_blocked = (flags & 4) != 0; g.reverse = (flags & 8) != 0;
    if (flags & 16) {
        if ((r = upload_f32(c, &dRow, hr.data(), (size_t)M * 2)) || (r = upload_f32(c, &dS, hs.data(), wrows))) return r;
        g.rowstat = dRow; g.ln_s = dS;
    }
    if (epi == vp::EPI_BIAS || epi == vp::EPI_BIAS_GELU) {
        rc.out_bytes = MN * 2;
    } else if (epi == vp::EPI_BIAS_RESID || epi == vp::EPI_POS) {
        rc.out_bytes = MN * 4;
        const size_t na = epi == vp::EPI_BIAS_RESID ? MN : (size_t)192 * N;
        if ((r = dalloc(c, &dAux32, na))) return r;
        std::vector<float> ha(na);
        for (auto& v : ha) v = 2.f * rnd();
        HIPCHK(c, hipMemcpy(dAux32, ha.data(), na * 4, hipMemcpyHostToDevice));
        g.aux = dAux32;
    } else if (epi == vp::EPI_BIAS_RESID_LN) {
        rc.out_bytes = MN * 4;   // hi plane + lo plane
        rc.stats_floats = (size_t)M * (N / 64) * 2;
        if ((r = dalloc(c, &dAux16, 2 * MN))) return r;
        vp::fill_random16(c->dtype, dAux16, MN, 3u, nullptr);
        vp::fill_random16(c->dtype, dAux16 + MN, MN, 4u, nullptr);
        g.aux = (const float*)dAux16;
        g.plane = MN;
    } else {
        return fail(c, VP_ERR_INVALID, "unsupported epilogue for the random GEMM case");
    }
    for (int i = 0; i < nout; ++i) {
        char* o;
        if ((r = dalloc(c, &o, rc.out_bytes))) return r;
        HIPCHK(c, hipMemset(o, 0xff, rc.out_bytes));
        rc.out[i] = o;
        if (rc.stats_floats) {
            if ((r = dalloc(c, &rc.stats[i], rc.stats_floats))) return r;
            HIPCHK(c, hipMemset(rc.stats[i], 0xff, rc.stats_floats * 4));
        }
    }
    HIPCHK(c, hipDeviceSynchronize());
    return VP_OK;
}
}  // namespace
extern "C" {

// ONE launch of any production GEMM configuration on HOST fp32 data (tests/test_gpu_gemm_cfgs.py): operands are rounded to
// `dtype` exactly as the packer / producing kernels round them, layouts (64x64-blocked A / output, two-plane residual stream,
// hi+lo final-conv weights) are built and undone here.
//   epi 0 / 1: out[M,N] 16-bit (returned as fp32); rowstat [M,2] + ln_s [N] non-NULL = LayerNorm-consumer fold
//   epi 2 / 3: out[M,N] fp32, aux = residual [M,N] / pos [192,N]
//   epi 6 / 7: aux = fp32 residual [M,N] (split into hi + lo planes on upload) / pos [192,N]; out = hi + lo planes summed;
//              stats [M, N/64, 2] = (sum, centred M2) per 64-column granule
//   epi 5:     W = final 1x1 conv weight [N = Kp, K = 256], A = [M = B 3072, 256]; out = heatmaps [B, Kp, 3072] fp32
// flags: 1 persistent, 2 out_blocked, 4 a_blocked, 8 reverse
VP_API int vp_dbg_gemm_case(int32_t device, int32_t dtype, int32_t epi, int32_t variant, int32_t group_m, int32_t flags, int32_t M,
                            int32_t N, int32_t K, const float* A, const float* W, const float* bias, const float* aux, const float* rowstat,
                            const float* ln_s, float* out, float* stats) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 64 || !A || !W || !bias || !out) return fail(nullptr, VP_ERR_INVALID, "bad gemm case");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    c->Kp = N;
    int r;
    const size_t MN = (size_t)M * N, wrows = pad128(N);
    const bool ablk = (flags & 4) != 0, oblk = (flags & 2) != 0;
    uint16_t *dA, *dW;
    float *dB, *dAux32 = nullptr, *dRow = nullptr, *dS = nullptr, *dStats = nullptr;
    uint16_t* dAux16 = nullptr;
    void* dOut = nullptr;
    // A (optionally in the 64x64-blocked layout [M/64][K/64][64][64])
    {
        std::vector<uint16_t> ha((size_t)M * K);
        for (size_t m = 0; m < (size_t)M; ++m)
            for (size_t k = 0; k < (size_t)K; ++k) {
                const size_t dst = ablk ? ((((m >> 6) * (K >> 6) + (k >> 6)) << 12) + ((m & 63) << 6) + (k & 63)) : m * K + k;
                ha[dst] = host_to_bits(A[m * K + k], c->dtype);
            }
        if ((r = dalloc(c, &dA, (size_t)M * K))) return dbg_finish(c, r);
        if (hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "H2D"));
    }
    size_t fin_rows = 0;
    if (epi == vp::EPI_HEATMAP) { if ((r = upload_final(c, &dW, W, N, K, &fin_rows))) return dbg_finish(c, r); }
    else if ((r = upload_mat(c, &dW, W, N, K, wrows))) return dbg_finish(c, r);
    if ((r = upload_f32(c, &dB, bias, N, wrows)) || (r = dalloc(c, &c->zero, (size_t)256))) return dbg_finish(c, r);
    vp::GemmArgs g{};
    g.A = dA; g.W = dW; g.bias = dB; g.M = M; g.N = N; g.K = K; g.ldo = N; g.zero = c->zero; g.Kp = N;
    g.w_rows = (int)wrows; g.variant = variant; g.group_m = group_m;
    g.persist = (flags & 1) != 0; g.out_blocked = oblk; g.a_blocked = ablk; g.reverse = (flags & 8) != 0;
    if (rowstat && ln_s) {
        if ((r = upload_f32(c, &dRow, rowstat, (size_t)M * 2)) || (r = upload_f32(c, &dS, ln_s, N, wrows))) return dbg_finish(c, r);
        g.rowstat = dRow; g.ln_s = dS;
    }
    size_t out_bytes = 0;
    const bool prod = epi == vp::EPI_BIAS_RESID_LN || epi == vp::EPI_POS_LN;
    if (epi == vp::EPI_BIAS || epi == vp::EPI_BIAS_GELU) out_bytes = MN * 2;
    else if (epi == vp::EPI_BIAS_RESID || epi == vp::EPI_POS || prod) out_bytes = MN * 4;
    else if (epi == vp::EPI_HEATMAP) { out_bytes = MN * 4; g.N = (int)fin_rows; g.ldo = 0; g.w_rows = (int)pad128(fin_rows); }
    else return dbg_finish(c, fail(c, VP_ERR_INVALID, "unsupported epilogue"));
    if (epi == vp::EPI_BIAS_RESID || epi == vp::EPI_POS || epi == vp::EPI_POS_LN) {
        if (!aux) return dbg_finish(c, fail(c, VP_ERR_INVALID, "aux required"));
        if ((r = upload_f32(c, &dAux32, aux, epi == vp::EPI_BIAS_RESID ? MN : (size_t)192 * N))) return dbg_finish(c, r);
        g.aux = dAux32;
    }
    if (epi == vp::EPI_BIAS_RESID_LN) {
        if (!aux) return dbg_finish(c, fail(c, VP_ERR_INVALID, "aux required"));
        std::vector<uint16_t> hp(2 * MN);
        for (size_t i = 0; i < MN; ++i) {
            const uint16_t hi = host_to_bits(aux[i], c->dtype);
            hp[i] = hi;
            hp[MN + i] = host_to_bits(aux[i] - host_from_bits(hi, c->dtype), c->dtype);
        }
        if ((r = dalloc(c, &dAux16, 2 * MN))) return dbg_finish(c, r);
        if (hipMemcpy(dAux16, hp.data(), hp.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "H2D"));
        g.aux = (const float*)dAux16;
    }
    if (prod) {
        g.plane = MN;
        if ((r = dalloc(c, &dStats, (size_t)M * (N / 64) * 2))) return dbg_finish(c, r);
        g.stats_out = dStats;
    }
    char* o;
    if ((r = dalloc(c, &o, out_bytes))) return dbg_finish(c, r);
    dOut = o;
    hipMemset(dOut, 0xff, out_bytes);
    g.out = dOut;
    hipError_t e = vp::gemm_launch(c->dtype, epi, g, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("gemm case: ") + hipGetErrorString(e)));
    if (epi == vp::EPI_BIAS || epi == vp::EPI_BIAS_GELU) {
        std::vector<float> t(MN);
        if ((r = download16(c, (const uint16_t*)dOut, t.data(), MN))) return dbg_finish(c, r);
        for (size_t m = 0; m < (size_t)M; ++m)
            for (size_t n = 0; n < (size_t)N; ++n) {
                const size_t src = oblk ? ((((m >> 6) * ((size_t)N >> 6) + (n >> 6)) << 12) + ((m & 63) << 6) + (n & 63)) : m * N + n;
                out[m * N + n] = t[src];
            }
    } else if (prod) {
        std::vector<float> hi(MN), lo(MN);
        if ((r = download16(c, (const uint16_t*)dOut, hi.data(), MN)) || (r = download16(c, (const uint16_t*)dOut + MN, lo.data(), MN))) return dbg_finish(c, r);
        for (size_t i = 0; i < MN; ++i) out[i] = hi[i] + lo[i];
        if (stats && hipMemcpy(stats, dStats, (size_t)M * (N / 64) * 8, hipMemcpyDeviceToHost) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "D2H"));
    } else {
        if (hipMemcpy(out, dOut, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "D2H"));
    }
    return dbg_finish(c, VP_OK);
}

// average milliseconds per launch of one production GEMM configuration on random operands
VP_API int vp_dbg_gemm_bench2(int32_t device, int32_t dtype, int32_t epi, int32_t variant, int32_t group_m, int32_t flags, int32_t M,
                              int32_t N, int32_t K, int32_t iters, float* ms_out) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 64 || iters <= 0 || !ms_out) return fail(nullptr, VP_ERR_INVALID, "bad gemm bench shape");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    RandCase rc;
    int r = make_rand_case(c, rc, epi, flags, M, N, K, 1);
    if (r) return dbg_finish(c, r);
    vp::GemmArgs g = rc.g;
    g.variant = variant & 0xff; g.group_m = group_m; g.ablate = variant >> 8;
    g.out = rc.out[0]; g.stats_out = rc.stats[0];
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = vp::gemm_launch(c->dtype, epi, g, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = vp::gemm_launch(c->dtype, epi, g, nullptr);
    hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("gemm bench2: ") + hipGetErrorString(e)));
    return dbg_finish(c, VP_OK);
}

#ifdef VP_TOOLS
// tools/gemm8_timeline.py: one gemm8 launch (variant 16 / 17, epi 0 / 1) with cycle stamps of waves 0 and 4 of every workgroup:
// stamps[wg][group][tile < 16][8] = (main loop begin, main loop end, epilogue end, P4 wait of K-tile 0 begin / end, of K-tile 1 begin / end, 0)
VP_API int vp_dbg_gemm8_timeline(int32_t device, int32_t dtype, int32_t epi, int32_t variant, int32_t flags, int32_t ablate, int32_t M,
                                 int32_t N, int32_t K, uint64_t* stamps, int32_t max_wg) {
    if ((epi != 0 && epi != 1 && epi != vp::EPI_BIAS_RESID_LN) || !stamps || max_wg < 256) return fail(nullptr, VP_ERR_INVALID, "bad timeline request");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    RandCase rc;
    int r = make_rand_case(c, rc, epi, flags, M, N, K, 1);
    if (r) return dbg_finish(c, r);
    unsigned long long* dS;
    const size_t nst = (size_t)max_wg * 2 * 16 * 8;
    if ((r = dalloc(c, &dS, nst))) return dbg_finish(c, r);
    hipMemset(dS, 0, nst * 8);
    vp::GemmArgs g = rc.g;
    g.variant = variant; g.group_m = 8; g.out = rc.out[0];
    if (epi == vp::EPI_BIAS_RESID_LN) g.stats_out = rc.stats[0];
    hipError_t e = vp::gemm_launch(c->dtype, epi, g, nullptr);   // warm
    g.ablate = 32 | ablate;
    if (epi == vp::EPI_BIAS_RESID_LN) { g.stats_out = rc.stats[0]; g.ln_part = (const float*)dS; }   // the residual GEMM writes real statistics: stamps go to the unused ln_part
    else g.stats_out = (float*)dS;
    if (e == hipSuccess) e = vp::gemm_launch(c->dtype, epi, g, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(stamps, dS, nst * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("gemm8 timeline: ") + hipGetErrorString(e)));
    return dbg_finish(c, VP_OK);
}
#endif  // VP_TOOLS

// run two configurations of the same GEMM on the same random operands `reps` times each and compare every output byte
// (and the row statistics): the race / schedule screen for kernels whose arithmetic order is identical by construction
VP_API int vp_dbg_gemm_compare(int32_t device, int32_t dtype, int32_t epi, int32_t variant_a, int32_t group_a, int32_t flags_a,
                               int32_t variant_b, int32_t group_b, int32_t flags_b, int32_t M, int32_t N, int32_t K, int32_t reps,
                               uint64_t* n_mismatch, double* max_abs_diff) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 64 || reps <= 0 || !n_mismatch || !max_abs_diff) return fail(nullptr, VP_ERR_INVALID, "bad gemm compare shape");
    if ((flags_a & (2 | 4 | 16)) != (flags_b & (2 | 4 | 16))) return fail(nullptr, VP_ERR_INVALID, "layout / fold flags must agree");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    RandCase rc;
    int r = make_rand_case(c, rc, epi, flags_a, M, N, K, 2);
    if (r) return dbg_finish(c, r);
    *n_mismatch = 0; *max_abs_diff = 0.0;
    std::vector<uint16_t> ha(rc.out_bytes / 2), hb2(rc.out_bytes / 2);
    std::vector<float> sa(rc.stats_floats), sb(rc.stats_floats);
    for (int rep = 0; rep < reps; ++rep) {
        for (int w = 0; w < 2; ++w) {
            vp::GemmArgs g = rc.g;
            const int fl = w ? flags_b : flags_a;
            g.variant = w ? variant_b : variant_a; g.group_m = w ? group_b : group_a;
            g.persist = (fl & 1) != 0; g.reverse = (fl & 8) != 0;
            g.out = rc.out[w]; g.stats_out = rc.stats[w];
            hipMemsetAsync(rc.out[w], 0xff, rc.out_bytes, nullptr);
            hipError_t e = vp::gemm_launch(c->dtype, epi, g, nullptr);
            if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("gemm compare launch ") + (w ? "B: " : "A: ") + hipGetErrorString(e)));
        }
        hipError_t e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(ha.data(), rc.out[0], rc.out_bytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(hb2.data(), rc.out[1], rc.out_bytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess && rc.stats_floats) e = hipMemcpy(sa.data(), rc.stats[0], rc.stats_floats * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && rc.stats_floats) e = hipMemcpy(sb.data(), rc.stats[1], rc.stats_floats * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("gemm compare: ") + hipGetErrorString(e)));
        const bool f32out = (epi == vp::EPI_BIAS_RESID || epi == vp::EPI_POS);
        if (f32out) {
            const float* fa = (const float*)ha.data(); const float* fb = (const float*)hb2.data();
            for (size_t i = 0; i < rc.out_bytes / 4; ++i)
                if (std::memcmp(&fa[i], &fb[i], 4)) { ++*n_mismatch; const double d = std::fabs((double)fa[i] - (double)fb[i]); if (!(d <= *max_abs_diff)) *max_abs_diff = d; }
        } else {
            for (size_t i = 0; i < ha.size(); ++i)
                if (ha[i] != hb2[i]) {
                    ++*n_mismatch;
                    const double d = std::fabs((double)host_from_bits(ha[i], c->dtype) - (double)host_from_bits(hb2[i], c->dtype));
                    if (!(d <= *max_abs_diff)) *max_abs_diff = d;
                }
        }
        for (size_t i = 0; i < sa.size(); ++i)
            if (std::memcmp(&sa[i], &sb[i], 4)) { ++*n_mismatch; const double d = std::fabs((double)sa[i] - (double)sb[i]); if (!(d <= *max_abs_diff)) *max_abs_diff = d; }
    }
    return dbg_finish(c, VP_OK);
}

// frame + crop geometry -> the uint8 [n,256,192,3] crops the model is fed (device crop/pad/resize kernel alone)
VP_API int vp_dbg_crop_prep(int32_t device, const uint8_t* frame, int32_t fh, int32_t fw, const int32_t* crop_params, int32_t n, uint8_t* out) {
    if (!frame || !crop_params || !out || n <= 0 || fh <= 0 || fw <= 0) return fail(nullptr, VP_ERR_INVALID, "bad argument");
    vp_ctx* c = dbg_ctx(device, VP_DTYPE_F16);
    if (!c) return VP_ERR_HIP;
    uint8_t *df, *dout;
    int32_t* dp;
    int rc;
    const size_t fb = (size_t)fh * fw * 3, ob = (size_t)n * 256 * 192 * 3;
    if ((rc = dalloc(c, &df, fb)) || (rc = dalloc(c, &dout, ob)) || (rc = dalloc(c, &dp, (size_t)n * 8))) return dbg_finish(c, rc);
    hipError_t e = hipMemcpy(df, frame, fb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dp, crop_params, (size_t)n * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = vp::crop_resize_launch(df, fh, fw, dp, dout, n, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dout, ob, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(c, VP_ERR_HIP, std::string("crop_prep: ") + hipGetErrorString(e));
    return dbg_finish(c, rc);
}

// BASELINE config 5 probe: rows quantised to OCP e4m3 on device + one GEMM through v_mfma_f32_16x16x128_f8f6f4 (fp8_probe.hip)
VP_API int vp_dbg_fp8_gemm(int32_t device, int32_t M, int32_t N, int32_t K, const float* A, const float* a_scale, const float* W,
                           const float* w_scale, float* out, uint8_t* a_codes, uint8_t* w_codes) {
    if (M <= 0 || N <= 0 || K <= 0 || M % 16 || N % 16 || K % 128 || !A || !W || !a_scale || !w_scale || !out)
        return fail(nullptr, VP_ERR_INVALID, "bad fp8 probe shape");
    vp_ctx* c = dbg_ctx(device, VP_DTYPE_F16);
    if (!c) return VP_ERR_HIP;
    float *dA, *dW, *dAs, *dWs, *dO;
    uint8_t *dA8, *dW8;
    int rc;
    if ((rc = upload_f32(c, &dA, A, (size_t)M * K)) || (rc = upload_f32(c, &dW, W, (size_t)N * K)) || (rc = upload_f32(c, &dAs, a_scale, M)) ||
        (rc = upload_f32(c, &dWs, w_scale, N)) || (rc = dalloc(c, &dO, (size_t)M * N)) || (rc = dalloc(c, &dA8, (size_t)M * K)) ||
        (rc = dalloc(c, &dW8, (size_t)N * K)))
        return dbg_finish(c, rc);
    hipError_t e = vp::fp8_probe_launch(dA, dW, dAs, dWs, dA8, dW8, dO, M, N, K, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dO, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && a_codes) e = hipMemcpy(a_codes, dA8, (size_t)M * K, hipMemcpyDeviceToHost);
    if (e == hipSuccess && w_codes) e = hipMemcpy(w_codes, dW8, (size_t)N * K, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(c, VP_ERR_HIP, std::string("fp8 probe: ") + hipGetErrorString(e));
    return dbg_finish(c, rc);
}

// MX probe (round 4): A -> MXFP8 on device (mx8.h layouts), W -> e4m3 with the per-row scale given; out = block-scaled MFMA product.
// a_codes [M*K] (blocked layout), a_scales [M*K/32] (packed dword layout), w_codes [N*K] may be NULL.
VP_API int vp_dbg_mx_gemm(int32_t device, int32_t M, int32_t N, int32_t K, const float* A, const float* W, const float* w_scale, float* out,
                          uint8_t* a_codes, uint8_t* a_scales, uint8_t* w_codes) {
    if (M <= 0 || N <= 0 || K <= 0 || M % 64 || N % 16 || K % 128 || !A || !W || !w_scale || !out) return fail(nullptr, VP_ERR_INVALID, "bad mx probe shape");
    vp_ctx* c = dbg_ctx(device, VP_DTYPE_F16);
    if (!c) return VP_ERR_HIP;
    float *dA, *dW, *dWs, *dO;
    uint8_t *dA8, *dAs, *dW8;
    int rc;
    if ((rc = upload_f32(c, &dA, A, (size_t)M * K)) || (rc = upload_f32(c, &dW, W, (size_t)N * K)) || (rc = upload_f32(c, &dWs, w_scale, N)) ||
        (rc = dalloc(c, &dO, (size_t)M * N)) || (rc = dalloc(c, &dA8, (size_t)M * K)) || (rc = dalloc(c, &dAs, (size_t)M * K / 32)) ||
        (rc = dalloc(c, &dW8, (size_t)N * K)))
        return dbg_finish(c, rc);
    hipError_t e = vp::mx_probe_launch(dA, dW, dWs, dA8, dAs, dW8, dO, M, N, K, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dO, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && a_codes) e = hipMemcpy(a_codes, dA8, (size_t)M * K, hipMemcpyDeviceToHost);
    if (e == hipSuccess && a_scales) e = hipMemcpy(a_scales, dAs, (size_t)M * K / 32, hipMemcpyDeviceToHost);
    if (e == hipSuccess && w_codes) e = hipMemcpy(w_codes, dW8, (size_t)N * K, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(c, VP_ERR_HIP, std::string("mx probe: ") + hipGetErrorString(e));
    return dbg_finish(c, rc);
}

// ONE launch of the MXFP8 GEMM kernel (gemm8f.hip) on host fp32 data (tests/test_gpu_fp8.py).  A [M,K] is quantised to MXFP8 on device
// (mx_quantize_launch: the layouts of csrc/mx8.h), W [N,K] on the host exactly as the weight packer does (per-output-channel scale);
// a_deq / w_deq return what the codes and scales stand for, so that the test can restate the product exactly.
//   epi 0: out = a.w^T * w_scale + bias, rounded to fp16        epi 1: out = gelu(...) as MXFP8 (returned de-quantised)
//   epi 6: out = ... + aux (two-plane residual, returned as hi + lo), stats [M, N/64, 2]
VP_API int vp_dbg_gemm_fp8_case(int32_t device, int32_t epi, int32_t M, int32_t N, int32_t K, const float* A, const float* W, const float* bias,
                                const float* aux, float* out, float* stats, float* a_deq, float* w_deq) {
    if (M <= 0 || N <= 0 || K <= 0 || M % 256 || K % 256 || N % 64 || !A || !W || !bias || !out || (epi != 0 && epi != 1 && epi != 6))
        return fail(nullptr, VP_ERR_INVALID, "bad fp8 gemm case");
    vp_ctx* c = dbg_ctx(device, VP_DTYPE_F16);
    if (!c) return VP_ERR_HIP;
    int r;
    const size_t MN = (size_t)M * N, MK = (size_t)M * K;
    float *dA, *dB, *dWs, *dStats = nullptr;
    uint8_t *dA8, *dAs, *dW8, *dOs = nullptr;
    uint16_t* dAux16 = nullptr;
    char* dOut;
    if ((r = upload_f32(c, &dA, A, MK)) || (r = dalloc(c, &dA8, MK)) || (r = dalloc(c, &dAs, MK / 32)) ||
        (r = upload_fp8_rows(c, &dW8, &dWs, nullptr, W, nullptr, nullptr, nullptr, N, K)) || (r = upload_f32(c, &dB, bias, N, pad128(N))))
        return dbg_finish(c, r);
    hipError_t e = vp::mx_quantize_launch(dA, dA8, dAs, M, K, nullptr);
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "mx quantize"));
    const size_t out_bytes = epi == 0 ? MN * 2 : epi == 1 ? MN : MN * 4;
    if ((r = dalloc(c, &dOut, out_bytes))) return dbg_finish(c, r);
    hipMemset(dOut, 0xff, out_bytes);
    LnFuse ln;
    if (epi == 1 && (r = dalloc(c, &dOs, MN / 32))) return dbg_finish(c, r);
    if (epi == 6) {
        if (!aux) return dbg_finish(c, fail(c, VP_ERR_INVALID, "aux required"));
        std::vector<uint16_t> hp(2 * MN);
        for (size_t i = 0; i < MN; ++i) {
            const uint16_t hi = host_to_bits(aux[i], c->dtype);
            hp[i] = hi;
            hp[MN + i] = host_to_bits(aux[i] - host_from_bits(hi, c->dtype), c->dtype);
        }
        if ((r = dalloc(c, &dAux16, 2 * MN)) || (r = dalloc(c, &dStats, (size_t)M * (N / 64) * 2))) return dbg_finish(c, r);
        if (hipMemcpy(dAux16, hp.data(), hp.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "H2D"));
        ln.plane = MN; ln.stats_out = dStats;
    }
    r = gemm_fp8(c, epi == 0 ? VP_PROF_GEMM_QKV : epi == 1 ? VP_PROF_GEMM_FC1 : VP_PROF_GEMM_FC2, epi, dA8, dAs, dW8, dWs, dB, dOut, dOs,
                 (const float*)dAux16, M, N, K, &ln);
    if (!r && hipDeviceSynchronize() != hipSuccess) r = fail(c, VP_ERR_HIP, "fp8 gemm kernel failed");
    if (r) return dbg_finish(c, r);
    // what the operands stand for
    {
        std::vector<uint8_t> ca(MK), sa(MK / 32);
        if (hipMemcpy(ca.data(), dA8, MK, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(sa.data(), dAs, MK / 32, hipMemcpyDeviceToHost) != hipSuccess)
            return dbg_finish(c, fail(c, VP_ERR_HIP, "D2H"));
        if (a_deq)
            for (size_t m = 0; m < (size_t)M; ++m)
                for (size_t k = 0; k < (size_t)K; ++k)
                    a_deq[m * K + k] = vp_host_e4m3_to_float(ca[vp::mx_code_off(m, k, K)]) * std::ldexp(1.0f, (int)sa[vp::mx_scale_off(m, k >> 5, K)] - 127);
        if (w_deq) {
            std::vector<uint8_t> cw((size_t)N * K);
            std::vector<float> sw(N);
            if (hipMemcpy(cw.data(), dW8, cw.size(), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(sw.data(), dWs, (size_t)N * 4, hipMemcpyDeviceToHost) != hipSuccess)
                return dbg_finish(c, fail(c, VP_ERR_HIP, "D2H"));
            for (size_t n = 0; n < (size_t)N; ++n)
                for (size_t k = 0; k < (size_t)K; ++k) w_deq[n * K + k] = vp_host_e4m3_to_float(cw[n * K + k]) * sw[n];
        }
    }
    if (epi == 0) {
        r = download16(c, (const uint16_t*)dOut, out, MN);
    } else if (epi == 1) {
        std::vector<uint8_t> co(MN), so(MN / 32);
        if (hipMemcpy(co.data(), dOut, MN, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(so.data(), dOs, MN / 32, hipMemcpyDeviceToHost) != hipSuccess)
            return dbg_finish(c, fail(c, VP_ERR_HIP, "D2H"));
        for (size_t m = 0; m < (size_t)M; ++m)
            for (size_t n = 0; n < (size_t)N; ++n)
                out[m * N + n] = vp_host_e4m3_to_float(co[vp::mx_code_off(m, n, N)]) * std::ldexp(1.0f, (int)so[vp::mx_scale_off(m, n >> 5, N)] - 127);
    } else {
        std::vector<float> hi(MN), lo(MN);
        if ((r = download16(c, (const uint16_t*)dOut, hi.data(), MN)) || (r = download16(c, (const uint16_t*)dOut + MN, lo.data(), MN))) return dbg_finish(c, r);
        for (size_t i = 0; i < MN; ++i) out[i] = hi[i] + lo[i];
        if (stats && hipMemcpy(stats, dStats, (size_t)M * (N / 64) * 8, hipMemcpyDeviceToHost) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "D2H"));
    }
    return dbg_finish(c, r);
}

// host-only: fp32 -> OCP e4m3 codes with the library's own converter (the one the weight packer of the fp8 mode uses)
VP_API int vp_dbg_host_e4m3(const float* in, uint8_t* out, int64_t n) {
    if (!in || !out || n < 0) return VP_ERR_INVALID;
    for (int64_t i = 0; i < n; ++i) out[i] = vp_host_e4m3(in[i]);
    return VP_OK;
}

// Calibration: kind 0/1 = MFMA-only loop (16x16x32 / 32x32x16 f16) in TFLOP/s, 2 = float4 copy in TB/s (read+write).
#ifdef VP_TOOLS
VP_API int vp_dbg_hwid_probe(int32_t device, int32_t blocks, int32_t threads, int32_t lds_bytes, int32_t spin, uint32_t* out) {
    if (!out || blocks <= 0 || blocks > 65536 || threads <= 0 || threads > 1024 || lds_bytes < 16 || lds_bytes > 160 * 1024 || spin < 0) return fail(nullptr, VP_ERR_INVALID, "bad argument");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, VP_ERR_HIP, "no HIP device");
    uint32_t* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)blocks * 16) != hipSuccess) return fail(nullptr, VP_ERR_HIP, "hipMalloc");
    hipError_t e = vp::hwid_probe_launch(d, blocks, threads, lds_bytes, spin, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    hipFree(d);
    return e == hipSuccess ? VP_OK : fail(nullptr, VP_ERR_HIP, hipGetErrorString(e));
}
#endif

VP_API int vp_dbg_peak(int32_t device, int32_t kind, double* result) {
    const bool known = (kind >= 0 && kind <= 12) || (kind >= 100 && kind < 164) || (kind >= 170 && kind < 178) || (kind >= 200 && kind < 248) ||
                       (kind >= 300 && kind < 492) || (kind >= 500 && kind < 504);
    if (!result || !known) return fail(nullptr, VP_ERR_INVALID, "bad argument");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, VP_ERR_HIP, "no HIP device");
    hipError_t e = vp::peak_bench(kind, result);
    return e == hipSuccess ? VP_OK : fail(nullptr, VP_ERR_HIP, hipGetErrorString(e));
}

}  // extern "C"
