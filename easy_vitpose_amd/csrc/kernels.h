// Internal launch interface between the C-ABI translation unit and the kernel
// translation units.  Everything here is host-side C++; kernels live in *.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Two builds of the same sources (easy_vitpose_amd/build.py): the PRODUCT library (libvitpose_hip.so) and, with -DVP_TOOLS, the
// measurement library tools/ loads (libvitpose_hip_tools.so): ablation flags and start stagger inside the GEMM loops, cycle
// stamps, the experimental tile configurations and kernel variants, the development environment switches.  In the product
// build GemmArgs::ablate / ::stagger read as the constant 0, so none of those branches exists in its kernels.
#ifdef VP_TOOLS
#define VP_ABLATE(g) ((g).ablate)
#define VP_STAGGER(g) ((g).stagger)
#else
#define VP_ABLATE(g) 0
#define VP_STAGGER(g) 0
#endif

namespace vp {

enum { DT_F16 = 0, DT_BF16 = 1 };

// ----------------------------------------------------------------------- GEMM
// C[m][n] = sum_k A[m][k] * W[n][k]   (A activations, W = nn.Linear weight [out,in])
enum GemmEpi {
    EPI_BIAS = 0,        // + bias            -> 16-bit [M, ldo]           (qkv)
    EPI_BIAS_GELU = 1,   // gelu(+ bias)      -> 16-bit [M, ldo]           (fc1)
    EPI_BIAS_RESID = 2,  // + bias + aux[m]   -> fp32   [M, ldo] (aux may alias out: proj, fc2)
    EPI_POS = 3,         // + aux[m % 192]    -> fp32   [M, ldo]           (patch embed; bias folded into aux)
    EPI_DECONV = 4,      // relu(+ bias)      -> 16-bit NHWC, output-parity scatter (deconv + folded BN + ReLU)
    EPI_HEATMAP = 5,     // + bias            -> fp32 NCHW heatmaps [B, Kp, 64*48]  (final 1x1 conv)
    EPI_BIAS_RESID_LN = 6,  // EPI_BIAS_RESID on the two-plane residual stream + fused-LayerNorm row statistics
    EPI_POS_LN = 7,         // EPI_POS        writing the two-plane residual stream + row statistics
    EPI_DECONV_FINAL = 8,   // EPI_DECONV with the final 1x1 conv (EPI_HEATMAP) fused behind it: the 256-channel activations stay in LDS,
                            // fp32 NCHW heatmaps at out2 (256 x 256 tile only: N == 256, variant 3)
    EPI_QKV_ATTN = 9,       // gemm8.hip, 192 x 256 tiles only (round 5): attn.qkv of ONE crop x ONE head of head dim 80 (W = head-major [q | k | v | 16 zero rows],
                            // N = heads * 256) with the LayerNorm-consumer fold, q / k / v handed to the attention core through LDS; out = y [M, K] 16-bit
    EPI_PARTIAL = 10,       // split-K (round 6, small batches): workgroup (split s, tile) multiplies k in [s K / S, (s + 1) K / S) and writes its fp32 partial tile to
                            // out[s][M][ldo] -- no bias, no residual; splitk_reduce_launch adds the S partials in the fixed order s = 0 .. S - 1 (deterministic), then
                            // bias + residual planes, and writes planes + row statistics exactly like the EPI_BIAS_RESID_LN epilogue
};
enum GemmAMode { A_DENSE = 0, A_DECONV = 1 };

struct GemmArgs {
    const uint16_t* A;    // dense: [M, K] row-major.  deconv: NHWC source [B, Hin, Win, Cin]
    const uint16_t* W;    // [w_rows, K] row-major, zero padded rows (deconv: 4 parity slabs of [w_rows, K])
    const float* bias;    // [Npad]
    void* out;
    const float* aux;     // residual [M, ldo] / pos [192, ldo]
    int M, N, K, ldo;     // N = real columns (stores are masked to n < N)
    int Hin, Win, Cin;    // deconv geometry (K = 4 * Cin)
    const uint16_t* zero; // >= 128 B of zeros (deconv border taps)
    int Kp;               // heatmap: number of keypoints (== N)
    // EPI_DECONV_FINAL: final-layer weights as [16 hi][16 lo] row groups x 256 channels, bias [Kp], heatmaps fp32 [B, Kp, 2Hin, 2Win]
    const uint16_t* W2;
    const float* bias2;
    float* out2;
    int w_rows;           // rows W is padded to at upload (multiple of 256); deconv parity slab = w_rows * K
    int variant;          // tile configuration (gemm.hip Cfg0..)
    int group_m;          // grouped tile order: m-tiles per group (<= 1: plain n-fastest order)
    size_t w_parity_stride;  // filled by gemm_launch
    // ---- fused LayerNorm ----
    // producer side (EPI_BIAS_RESID_LN / EPI_POS_LN): the residual stream is two 16-bit planes, x = hi + lo,
    // hi = round16(x) at out / aux, lo = round16(x - hi) `plane` elements behind it; besides the rows the partial
    // row statistics (sum, M2 about the granule mean) of every 64-column granule go to stats_out[(m*(N/64) + n/64)*2]
    size_t plane;
    float* stats_out;
    // consumer side (EPI_BIAS / EPI_BIAS_GELU): A is the UN-normalised 16-bit residual stream, W has LayerNorm's
    // gamma folded in; the epilogue applies  v = rstd_m * (acc - mean_m * ln_s[n]) + bias[n]  with
    // rowstat[m] = (mean, rstd), ln_s[n] = sum_k W'[n][k], bias[n] = sum_k beta_k W[n][k] + b[n]
    const float* rowstat;
    const float* ln_s;
    // small batches: instead of rowstat, the producer's partial statistics [M][ln_tiles][2]; the consumer epilogue folds them itself
    // (common.h::ln_merge, the same code ln_finalize_kernel runs) and the 2 x depth ln_finalize launches disappear
    const float* ln_part;
    int ln_tiles;
    float ln_inv_d;       // 1 / D as the host computes it for ln_finalize_launch
    // 64x64-blocked activation layout [M/64][K/64][64][64]: every operand tile of the consuming GEMM is a run of
    // contiguous 8 KiB blocks (sequential DRAM bursts instead of 128-byte pieces at a row stride).  out_blocked: this
    // GEMM writes its 16-bit output that way (EPI_BIAS / EPI_BIAS_GELU, ldo == N); a_blocked: A is read that way.
    int a_blocked, out_blocked;
    // tile order last-to-first: a consumer of a tensor larger than the 256 MB Infinity Cache then starts with the
    // rows its producer wrote last (still cached) instead of the ones already evicted
    int reverse;
    // persistent workgroups whose operand ring runs on across tile boundaries (gemm.hip gemm_persist_kernel):
    // EPI_BIAS / EPI_BIAS_GELU on the default tile, K % 128 == 0
    int persist;
    // gemm8.hip: start of XCD x (= blockIdx & 7) delayed by x * stagger * 64 * 127 shader cycles, so that the eight XCDs reach
    // their tile boundaries -- the store bursts of the epilogue -- at different times instead of saturating HBM together
    int stagger;
    int ablate;           // VP_TOOLS builds only (tools/gemm_ablate.py): 1 = no operand loads after the prologue, 2 = every tile loads the A rows of m-tile 0 (A L2-resident), 8 = no epilogue stores
    char* desc;           // host side: when non-null the launch code writes the resolved kernel's name here (desc_cap bytes)
    int desc_cap;
    // ---- fp8 mode (gemm8f.hip; csrc/mx8.h): A = MXFP8 codes in 64 x 128 blocks (at `A`) + packed E8M0 block scales, W = e4m3 codes
    // row-major [w_rows, K] (at `W`) + one fp32 scale per output channel; EPI_BIAS_GELU writes its output as MXFP8 (codes at `out`, K of the
    // consumer = ldo, scales at out_scales)
    float attn_scale_log2e;   // EPI_QKV_ATTN: head_dim^-0.5 * log2(e)
    int splitk;        // EPI_PARTIAL: number of k ranges S (K % (S * BK) == 0; grid = tiles * S)
    int parity_fast;   // deconv: 1 = the four output parities of a tile are consecutive logical blocks (same XCD, shared input rows); 0 = parity on blockIdx.y
    const uint8_t* a_scales;
    const float* w_scale;
    uint8_t* out_scales;
};
hipError_t gemm_launch(int dtype, int epi, const GemmArgs& a, hipStream_t s);
// 8-phase persistent kernel (gemm8.hip): 256 x bn tiles (bn = 256 or 192), EPI_BIAS / EPI_BIAS_GELU / EPI_BIAS_RESID_LN.
// Selected through GemmArgs::variant 16 (bn 256) / 17 (bn 192) in gemm_launch.
bool gemm8_supported(int epi, const GemmArgs& a, int bn, int bm = 256);
hipError_t gemm8_launch(int dtype, int epi, const GemmArgs& a, int bn, hipStream_t s, int bm = 256);
// name of the kernel a launch resolves to, as the profiler prints it minus the namespace; written by the launch code when
// GemmArgs::desc != nullptr (vp_profile_kernel)

// 8-phase kernel on MXFP8 operands (gemm8f.hip): the encoder GEMMs of the opt-in fp8 mode.  K % 256 == 0, K >= 512, M % 256 == 0.
bool gemm8f_supported(int epi, const GemmArgs& a, int bn);
hipError_t gemm8f_launch(int epi, const GemmArgs& a, int bn, hipStream_t s);
// fp8 mode, between a residual GEMM and qkv / fc1 (quant8.hip): merges the producer's partial row statistics (ln_merge), normalises the
// hi plane of the residual stream and writes it as MXFP8 (codes [Mp/64][D/128][64][128], scales packed: mx8.h); rows >= M (padding up to
// the multiple of 256 the GEMM tiles need) are written as zeros.  LayerNorm's gamma / beta live in the consumer's weights / bias.
hipError_t ln_quant_launch(int dtype, const uint16_t* x_hi, const float* ln_part, int tiles, uint8_t* codes, uint8_t* scales, int M, int Mp, int D,
                           hipStream_t s);
// fp32 rows [M, K] -> MXFP8 in the same layouts (parity taps)
hipError_t mx_quantize_launch(const float* src, uint8_t* codes, uint8_t* scales, int M, int K, hipStream_t s);

int gemm_tile_bn(int variant);   // BN of a tile configuration (number of n-tiles = ceil(N / BN))
// fill a 16-bit buffer with pseudo-random values in [-1, 1) (benchmark operands)
hipError_t fill_random16(int dtype, uint16_t* p, size_t n, uint32_t seed, hipStream_t s);

// partial row statistics [M][tiles][2] (sum, centred M2 per 64-column granule) -> rowstat [M][2] (mean, rstd), LayerNorm eps 1e-6
hipError_t ln_finalize_launch(const float* partials, float* rowstat, int M, int tiles, int D, hipStream_t s);
// split-K reduction of a residual GEMM (EPI_PARTIAL partials [S][M][N] fp32): x = ((p_0 + p_1) + ... + p_{S-1}) + bias + (x_hi + x_lo), written back as
// the two residual planes (in place: hi at x_hi, lo `plane` elements behind) + the partial row statistics [M][N/64][2] the LayerNorm consumers fold --
// the arithmetic of the EPI_BIAS_RESID_LN epilogue behind a fixed-order sum of the partials.  N % 64 == 0.
hipError_t splitk_reduce_launch(int dtype, const float* partials, int S, const float* bias, uint16_t* x_hi, size_t plane, float* stats_out, int M, int N,
                                hipStream_t s);

// fp8_probe.hip (parity tap of BASELINE config 5): rows of A [M,K] / W [N,K] -> e4m3 codes (x / scale[row]), out = scaled product
// through v_mfma_f32_16x16x128_f8f6f4
hipError_t fp8_probe_launch(const float* dA, const float* dW, const float* dAs, const float* dWs, uint8_t* dA8, uint8_t* dW8, float* dOut,
                            int M, int N, int K, hipStream_t s);

// MX probe: rows of A -> MXFP8 (e4m3 + one E8M0 scale per 32 k, layouts of mx8.h), W -> e4m3 per-row scale; product through the block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 with the operand roles and the packed scale dwords of the fp8 mode's GEMM.  M % 64 == 0, N % 16 == 0, K % 128 == 0
hipError_t mx_probe_launch(const float* dA, const float* dW, const float* dWs, uint8_t* dA8, uint8_t* dAs, uint8_t* dW8, float* dOut,
                           int M, int N, int K, hipStream_t s);

// calibration micro-benchmarks (tools/): kind 0/1 = MFMA 16x16x32 / 32x32x16 f16 TFLOP/s, 2 = float4 copy TB/s
hipError_t peak_bench(int kind, double* result);
// tools/hwid_probe.py: every workgroup of a launch records (HW_REG_HW_ID, HW_REG_XCC_ID, start cycle lo, hi) -> d_out[blocks][4]
hipError_t hwid_probe_launch(uint32_t* d_out, int blocks, int threads, int lds_bytes, int spin, hipStream_t s);

// ------------------------------------------------------------------ attention
// qkv [B*192, 3*D] 16-bit (columns = [q | k | v] x heads x head_dim, vit.py:166-167)
// out [B*192, D]   16-bit (columns = heads x head_dim, vit.py:176)
// qkv_blocked (head dim 64 only): qkv in the 64 x 64-blocked layout the GEMMs write with GemmArgs::out_blocked
// mx_scales != nullptr (fp8 mode, head dim 64): the output is written as MXFP8 (codes at `out`, 64 x 128-blocked; E8M0 scales at mx_scales)
hipError_t attention_launch(int dtype, const uint16_t* qkv, uint16_t* out, int B, int D, int heads, hipStream_t s, int qkv_blocked = 0,
                            uint8_t* mx_scales = nullptr);

// attn.qkv + attention core in one kernel (qkvattn.hip; head dim 64): tile = (pair of crops, head); the qkv tensor never reaches HBM.
// wh / bh / sh = head-major copies of the LayerNorm-folded qkv weights / bias / row sums (qkv_head_major_launch).  y bit-identical to
// gemm (EPI_BIAS, LayerNorm-consumer fold) + attention_launch.
struct QkvAttnArgs {
    const uint16_t* x_hi;     // hi plane of the residual stream [ncrops 192, D] (un-normalised 16-bit rows)
    const uint16_t* wh;       // [heads 192, D]
    const float* bh;          // [heads 192]
    const float* sh;          // [heads 192]
    const float* rowstat;     // (mean, rstd) per token row
    uint16_t* y;              // attention output [ncrops 192, D]
    int npairs, heads, D;
    int ncrops;               // 2 npairs, or 2 npairs - 1: the last pair of an odd batch runs its only crop in both halves (identical results stored twice)
    float scale_log2e;        // head_dim^-0.5 * log2(e)
    int ablate;               // VP_TOOLS builds only (tools/qkvattn_phases.py): 1 = no attention phase, 2 = no epilogue (LDS hand-over), 4 = no K-loop, 8 = no ring restart wait
};
bool qkvattn_supported(const QkvAttnArgs& a);
hipError_t qkvattn_launch(int dtype, const QkvAttnArgs& a, hipStream_t s, char* desc, int desc_cap);
hipError_t qkv_head_major_launch(const uint16_t* w, const float* b, const float* s, uint16_t* wh, float* bh, float* sh, int D, int K, hipStream_t st);
// head dim 80 (ViTPose-H; round 5): the fused kernel is gemm8.hip's 192 x 256 tile with EPI_QKV_ATTN -- one crop x one head per tile.  Head-major copy of the
// LayerNorm-folded qkv weights: rows h 256 + [0, 80) = q_h, [80, 160) = k_h, [160, 240) = v_h, [240, 256) = zeros; bias and row sums alike (heads * 256 entries)
hipError_t qkv_head_major80_launch(const uint16_t* w, const float* b, const float* s, uint16_t* wh, float* bh, float* sh, int D, int K, int heads, hipStream_t st);

// ---------------------------------------------------------------- elementwise
// fp32 [M, D] -> LayerNorm(eps 1e-6) -> 16-bit [M, D] (out16) and/or fp32 (out32), either may be null.
// plane != 0: x is the two-plane 16-bit residual stream (hi at x, lo `plane` elements behind) instead of fp32.
hipError_t layernorm_launch(int dtype, const float* x, const float* gamma, const float* beta,
                            uint16_t* out16, float* out32, int M, int D, hipStream_t s, size_t plane = 0);
// crops -> im2col patch matrix [B*192, 768] 16-bit (k = c*256 + ky*16 + kx, zero border of 2 px)
hipError_t im2col_launch(int dtype, const void* crops, int input_format, uint16_t* out, int B, hipStream_t s, bool flip = false, int n_src = 0);   // n_src < B: output crops n_src .. B - 1 repeat source crop n_src - 1
// flip-test: hm = 0.5 (hm + flip_back(hm_flipped)); partner[k] = mirror joint of k (k itself if unpaired)
hipError_t flip_merge_launch(float* hm, const float* hm_flipped, const int32_t* partner, int N, int K, int shift, hipStream_t s);

// frame u8 [FH,FW,3] + params int32 [n,8] (x0,y0,cw,ch,left,top,pw,ph) -> crops u8 [n,256,192,3]
hipError_t crop_resize_launch(const uint8_t* frame, int FH, int FW, const int32_t* params, uint8_t* out, int n, hipStream_t s);

// --------------------------------------------------------------------- decode
// heatmaps fp32 [N, K, 64, 48] -> out fp32 [N, K, 3] (y, x, conf); org_wh int32 [N,2] or null
hipError_t decode_launch(const float* hm, const int32_t* org_wh, float* out, int N, int K, hipStream_t s);

}  // namespace vp
