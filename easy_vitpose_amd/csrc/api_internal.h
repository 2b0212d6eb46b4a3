// Internal header shared by the translation units behind the C ABI (include/vitpose_hip.h):
//   vitpose_api.hip  -- handle, streams, forward orchestration, the ABI entry points, the multi-device group
//   weights.hip      -- the weight packer (vp_load_weights: BN / LayerNorm folding, 16-bit / e4m3 conversion, deconv re-tiling)
//   tile_rules.hip   -- which GEMM tile runs a shape (pure host functions + their host-only taps)
//   debug_taps.hip   -- vp_dbg_*: one kernel on host data (parity tests), and the measurement build's timing taps
// Everything here is library-internal (hidden visibility); nothing of it appears in the public headers.
#pragma once
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

namespace vpi {

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

}  // namespace vpi

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
    std::vector<vpi::Block> blocks;
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
    bool pad_batch = true;            // the encoder runs the next multiple of 4 crops where that buys an 8-phase tile (tile_rules.hip pick_run_batch; VP_PAD_BATCH=0: never)
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
    bool fold_rule = true;            // beyond graph_max_n_stats crops: fold per consumer where its tile keeps its occupancy with the statistics area (forward_chunk; off when VP_FOLD_STATS is set)
    int graph_max_n_stats = 8;        // batches of <= this many crops: the consumer GEMMs (qkv, fc1) merge the LayerNorm partial statistics of their tile rows
                                      // themselves (once per row and tile, in the prologue: gemm.hip) and the 2 x depth ln_finalize launches disappear -- same
                                      // ln_merge, bit-identical.  Measured (profiles/fold_stats_r3.txt): -7...-12 % per step at 1-8 crops, +0...+20 % at
                                      // 16-48 (every column tile merges its rows again): threshold 8.  VP_FOLD_STATS=n moves it (0 = always ln_finalize).
                                      // Round 6 (merge on a register copy, profiles/small_batch_r6.txt call 11): -2.6 ... -6.5 % against ln_finalize at 1-8 crops.
                                      // Round 6, call 25: the '+0 ... +20 %' beyond 8 crops was the statistics area behind the default tile's 80 KiB ring (one workgroup per CU
                                      // instead of two), not the merge: beyond this threshold each consumer folds on its own where its tile keeps its occupancy (fold_rule).
                                      // Round 2 merged per LANE in the epilogue (16 x redundant): slower than ln_finalize even at 8 crops (3.89 vs 2.97 ms).
    // split-K for the residual GEMMs of small batches (round 6; tile_rules.hip pick_splitk, gemm.hip EPI_PARTIAL, elementwise.hip splitk_reduce_kernel): fp32 partial
    // products [S][M][D] of up to splitk_rows token rows.  VP_SPLITK=0 switches it off (the parity test flips it); VP_SPLITK="fc2:S:variant,proj:S:variant" overrides the rule.
    float* splitk_ws = nullptr;
    size_t splitk_rows = 0;
    bool splitk_on = true;
    int splitk_force[2][2] = {{0, 0}, {0, 0}};   // [0 = proj, 1 = fc2][S, variant]; S = 0: the rule decides
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

namespace vpi {

extern thread_local std::string g_create_error;

int fail(vp_ctx* c, int code, const std::string& msg);

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

// ---- weights.hip
uint16_t host_to_bits(float v, int dtype);     // fp32 -> 16-bit storage on the host (round to nearest even), same as the device paths
float host_from_bits(uint16_t h, int dtype);
size_t pad128(size_t n);
int upload_f32(vp_ctx* c, float** dst, const float* src, size_t n, size_t npad = 0);
int upload_mat(vp_ctx* c, uint16_t** dst, const float* src, size_t rows, size_t cols, size_t rows_pad);
int upload_final(vp_ctx* c, uint16_t** dst, const float* src, size_t kp, size_t cols, size_t* rows_phys);
int upload_ln_folded(vp_ctx* c, uint16_t** w_out, float** s_out, float** c_out, const float* W, const float* b,
                     const float* gamma, const float* beta, size_t N, size_t K);
int upload_fp8_rows(vp_ctx* c, uint8_t** w_out, float** ws_out, float** c_out, const float* W, const float* b, const float* gamma,
                    const float* beta, size_t N, size_t K);

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

int pack_deconv(vp_ctx* c, Lookup& lk, int idx, int Cin, uint16_t** w_out, float** b_out);

// ---- vitpose_api.hip
bool prof_begin(vp_ctx* c, int fam, double flops, double bytes);
void prof_end(vp_ctx* c, bool on);
void prof_collect(vp_ctx* c);
void apply_gemm_tuning(vp_ctx* c);

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

// ---- tile_rules.hip
struct G8Pick { int variant, bm, bn; long tiles; };
G8Pick pick_gemm8_tile(int M, int N, bool wide, int bm192_mask, long min_tiles, bool extended);
struct Tile2Pick { int variant, group_m; };
Tile2Pick pick_gemm2_tile(int epi, int M, int N, int K);
// split-K of a residual GEMM (attn.proj, mlp.fc2) at small batches: S k ranges on tile configuration `variant` (S = 1: no split)
struct SplitKPick { int S, variant; };
SplitKPick pick_splitk(int M, int N, int K);
int pick_run_batch(int n, int D, int limit, int bm192_mask, bool extended, int gemm8_mask);   // the batch the encoder runs for a chunk of n crops (>= n, a multiple of 4 when padded)
constexpr int SPLITK_MAX_S = 8, SPLITK_MAX_CROPS = 32;

// ---- vitpose_api.hip: one GEMM of the path through the tile rules (also what the vp_dbg_gemm* taps launch)
int gemm(vp_ctx* c, int fam, int epi, const uint16_t* A, const uint16_t* W, const float* bias, void* out,
         const float* aux, int M, int N, int K, int ldo, int Hin = 0, int Win = 0, int Cin = 0, const LnFuse* ln = nullptr);
int gemm_fp8(vp_ctx* c, int fam, int epi, const uint8_t* A8, const uint8_t* a_scales, const uint8_t* W8, const float* w_scale, const float* bias,
             void* out, uint8_t* out_scales, const float* aux, int M, int N, int K, const LnFuse* ln);

}  // namespace vpi
