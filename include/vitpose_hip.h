/*
 * vitpose_hip.h -- C ABI of the MI355X-native ViTPose hot path (libvitpose_hip.so).
 *
 * Drop-in boundary: the reference's backend slot `VitInference._inference`
 * (easy_ViTPose/inference.py:207-219, assigned at :169-172, called at :268) whose
 * existing implementations are `_inference_torch` (:320-328) and `_inference_onnx`
 * (:330-337).  A backend takes one cropped+padded RGB crop, runs
 *     pre_img (:314-318) -> ViTPose.forward (vit_models/model.py:23-24)
 *             -> postprocess (:187-205)
 * and returns float32 [1, K, 3] = (y, x, conf) in crop pixels.  This library is
 * the batched form of exactly that call: N crops in, [N, K, 3] out.
 *
 * Plain C: opaque handle, plain pointers and sizes, int status codes.  No torch
 * types, no C++ types.  All entry points are thread-compatible, not thread-safe:
 * calls on one handle must be serialised by the caller (the reference's
 * VitInference is single-threaded and stateful too, inference.py:112-116).
 * Nothing here falls back to a CPU implementation: without a gfx950 device every
 * compute entry point returns VP_ERR_HIP.
 */
#ifndef VITPOSE_HIP_H
#define VITPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VP_ABI_VERSION 4
#define VP_API __attribute__((visibility("default")))

/* status codes (0 = ok).  The Python host maps them onto the exception types the
 * reference raises at the same places (AssertionError / ValueError / KeyError,
 * easy_ViTPose/inference.py:91-92,127-128,138-144; vit_utils/util.py:33). */
enum {
    VP_OK = 0,
    VP_ERR_INVALID = 1,        /* bad argument / unsupported shape          */
    VP_ERR_HIP = 2,            /* HIP runtime error, no device, OOM         */
    VP_ERR_STATE = 3,          /* weights not loaded yet, handle destroyed  */
    VP_ERR_MISSING_TENSOR = 4, /* state-dict key absent (load_state_dict KeyError analogue) */
    VP_ERR_SHAPE = 5           /* state-dict tensor has the wrong size      */
};

/* arithmetic type of the GEMM/attention operands (accumulation is always fp32; LayerNorm / softmax statistics,
 * heatmaps and the decode are fp32; the residual stream is held as a hi + lo pair of 16-bit planes, >= 22 bits) */
enum {
    VP_DTYPE_F16 = 0,
    VP_DTYPE_BF16 = 1,
    /* OPT-IN, never a default: the encoder's qkv / fc1 / fc2 GEMMs on MXFP8 operands (OCP e4m3 + one power-of-two scale per 32 k,
     * csrc/mx8.h) through the block-scaled fp8 matrix instruction; residual stream, attention core, attn.proj, head and decode as in
     * VP_DTYPE_F16.  BASELINE configs[4].  Does NOT meet the north_star's 1e-3 on confidences (e4m3 carries 3 mantissa bits; measured
     * bounds in tests/test_gpu_fp8.py and DESIGN.md section 6); coordinates stay within +-0.5 px on peaked maps.  ViTPose-B / -L / -H only. */
    VP_DTYPE_FP8 = 2
};

/* layout of the crop batch handed to vp_infer* */
enum {
    VP_INPUT_F32_NCHW = 0, /* float32 [N,3,256,192], already normalised = output of pre_img (inference.py:314-318) */
    VP_INPUT_U8_NHWC = 1   /* uint8   [N,256,192,3] RGB crops; (x/255-mean)/std (inference.py:32-33,316-317) is applied on device */
};

typedef struct vp_ctx* vp_handle;
typedef struct vp_group* vp_group_handle; /* N handles in one process, one per device (vp_group_*) */

/* Model shape = one row of configs/ViTPose_common.py:65-195 + the dataset's
 * out_channels (configs/ViTPose_<dataset>.py).  Input is fixed at 256x192,
 * heatmaps at 64x48 (ViTPose_common.py:29-31). */
typedef struct vp_config {
    int32_t embed_dim;     /* 384 / 768 / 1024 / 1280 */
    int32_t depth;         /* 12 / 12 / 24 / 32       */
    int32_t num_heads;     /* 12 / 12 / 16 / 16  (head_dim must be 32, 64 or 80) */
    int32_t num_keypoints; /* 17 / 25 / 133 / ...      */
    int32_t dtype;         /* VP_DTYPE_*              */
    int32_t device_id;     /* HIP device ordinal      */
    int32_t max_batch;     /* workspace is sized for this many crops; larger N is processed in chunks */
} vp_config;

/* One tensor of the checkpoint, host float32, named exactly as in the reference's
 * state dict (`backbone.pos_embed`, `backbone.blocks.3.attn.qkv.weight`,
 * `keypoint_head.deconv_layers.1.running_var`, ... ; schema in SURVEY.md 8a,
 * loaded by the reference at easy_ViTPose/inference.py:162-166). */
typedef struct vp_tensor_desc {
    const char* name;
    const float* data;
    int64_t numel;
} vp_tensor_desc;

/* Per-kernel-family timing collected with HIP events on the handle's stream. */
#define VP_PROF_GEMM_PROJ 0  /* attn.proj GEMM (bias + residual planes + LayerNorm statistics), HBM-bound */
#define VP_PROF_GEMM_FC1 1   /* fc1 GEMM (bias + GELU)                     */
#define VP_PROF_GEMM_QKV 2   /* qkv GEMM (bias)                            */
#define VP_PROF_GEMM_PATCH 3 /* patch-embed GEMM (bias + pos)              */
#define VP_PROF_GEMM_DECONV 4
#define VP_PROF_GEMM_FINAL 5
#define VP_PROF_ATTN 6
#define VP_PROF_LAYERNORM 7
#define VP_PROF_IM2COL 8
#define VP_PROF_DECODE 9
#define VP_PROF_GEMM_FC2 10  /* mlp.fc2 GEMM (same epilogue as proj, K = 4 D) */
#define VP_PROF_COUNT 11

typedef struct vp_profile {
    double ms[VP_PROF_COUNT];      /* summed kernel time since the last reset  */
    double flops[VP_PROF_COUNT];   /* summed algorithmic FLOPs (2*M*N*K)       */
    double bytes[VP_PROF_COUNT];   /* summed algorithmic HBM bytes              */
    int64_t launches[VP_PROF_COUNT];
} vp_profile;

VP_API int vp_abi_version(void);

/* Create a context on cfg->device_id: stream, workspaces.  Replaces the model
 * construction `ViTPose(model_cfg)` + `.to(device)` (inference.py:156-167). */
VP_API int vp_create(vp_handle* out, const vp_config* cfg);

/* Upload a checkpoint: folds BatchNorm (eval, eps 1e-5) into the deconv weights,
 * pre-adds pos_embed[:,1:]+pos_embed[:,:1] (vit.py:382), re-tiles the transposed-conv
 * weights into 4 output-parity GEMM operands, converts matrices to cfg.dtype.
 * Replaces `load_state_dict` (inference.py:162-166).  Missing key ->
 * VP_ERR_MISSING_TENSOR, wrong numel -> VP_ERR_SHAPE. */
VP_API int vp_load_weights(vp_handle h, const vp_tensor_desc* tensors, int32_t n_tensors);

/* The hot path on host buffers: H2D crops, model, decode, D2H keypoints; returns
 * when `out` is complete.  Batched replacement of `_inference_torch`
 * (inference.py:320-328).  org_wh = N x (org_w, org_h) int32 of each crop before
 * pre_img's resize (NULL = 192x256 for all); out = float32 [N, K, 3] (y, x, conf).
 * Numerics contract: every call is run-to-run bit-identical (no atomics anywhere) and within the parity tolerances of the
 * reference (+-0.5 px, confidences 1e-3).  Equal BITS for a crop at every batch size hold inside the one-launch kernel family;
 * a chunk of one or two crops (ViTPose-H: also seven or eight) runs mlp.fc2 as four k ranges + a fixed-order reduction (a
 * different fp32 accumulation order), so such a call differs from the same crop inside a larger batch in the last bits.
 * VP_SPLITK=0 (read at vp_create) pins a handle to the one-launch family. */
VP_API int vp_infer(vp_handle h, const void* crops, int32_t input_format, int32_t n,
             const int32_t* org_wh, float* out);

/* Same with device-resident buffers (e.g. torch tensors' data_ptr()); enqueued on
 * the handle's stream, returns after the stream has been synchronised when
 * `sync` != 0.  d_org_wh may be NULL. */
VP_API int vp_infer_device(vp_handle h, const void* d_crops, int32_t input_format, int32_t n,
                    const int32_t* d_org_wh, float* d_out, int32_t sync);

/* vp_infer_device ordered against the CALLER's stream (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream; NULL =
 * the legacy default stream): the library's kernels start after everything enqueued on caller_stream so far (the producers
 * of d_crops / d_org_wh), and work enqueued on caller_stream afterwards (consumers of d_out) starts after them.  Returns
 * without synchronising.  This is the entry a framework should use; plain vp_infer_device leaves the ordering to the caller.
 * Batches of <= 16 crops (that fit max_batch; VP_CALLER_STREAM=0: never) are enqueued ON caller_stream itself -- same ordering guarantee, without the two cross-stream
 * events of the general path (~0.1 ms of a 0.6-2.4 ms call).  The handle orders its next call behind that work by itself, whatever entry or stream it comes through,
 * and vp_synchronize / vp_destroy wait for it; caller_stream is not touched again after the call returns (it may be destroyed once its work has completed). */
VP_API int vp_infer_device_stream(vp_handle h, const void* d_crops, int32_t input_format, int32_t n,
                                  const int32_t* d_org_wh, float* d_out, void* caller_stream);

/* Asynchronous host path: the H2D copy of call i+1 and the D2H copy of call i-1 run on a copy stream under the compute of
 * call i (two slots).  vp_infer_submit enqueues one batch (1 .. max_batch crops) and returns a slot id at once; `crops`,
 * `org_wh` must stay valid until the upload has happened and `out` until vp_infer_wait(slot) has returned.  For the copies
 * to be truly asynchronous the host buffers must be pinned: vp_host_alloc / vp_host_free (hipHostMalloc).  At most two
 * submissions may be in flight. */
VP_API void* vp_host_alloc(size_t bytes);
VP_API void vp_host_free(void* p);
VP_API int vp_infer_submit(vp_handle h, const void* crops, int32_t input_format, int32_t n, const int32_t* org_wh,
                           float* out, int32_t* slot);
VP_API int vp_infer_wait(vp_handle h, int32_t slot);

/* Multi-GPU in ONE process (the reference has no inference-side parallelism to mirror, inference.py:167,259-272
 * -- this is the build's own contract): one handle per device, weights replicated, the crops of a call sharded contiguously
 * (ceil(n / devices) each), every device running its shard concurrently (asynchronous submit on each handle).
 * vp_group_infer gathers the keypoints in the host buffer `out` [n, K, 3] (each device's D2H lands in its slice).
 * vp_group_infer_allgather additionally leaves ALL keypoints on EVERY device: d_all[i] = device buffer [n, K, 3] on device i;
 * each shard is copied peer to peer (hipMemcpyPeerAsync over the xGMI link of the pair) on its owner's stream -- the
 * all-gather of north_star without a collective library in the C ABI (the one-process-per-GPU Python host uses RCCL:
 * easy_vitpose_amd/parallel.py).  `out` may be NULL there.  cfg->device_id is ignored. */
/* Concurrency: a call enqueues upload + model + decode + download on EVERY member first and only then waits
 * for the members in turn and copies their slices into `out`.  Neither direction blocks the enqueue on a device: the download lands
 * in a pinned per-member staging buffer; the upload is asynchronous as it is when `crops` is page-locked memory the runtime knows
 * (vp_host_alloc / hipHostMalloc / hipHostRegister: checked with hipPointerGetAttributes) and otherwise goes through a pinned
 * per-member staging buffer in 4 MiB pieces (host memcpy of piece k + 1 under the DMA of piece k).  Pass pinned crops for full overlap:
 * with pageable memory the calling thread's memcpy (~10 GB/s: ~4 ms per 256 u8 crops, ~15 ms per 256 f32 crops) is what serialises the members.
 * Ordering of vp_group_infer_allgather: the peer copies into d_all[j] run on the OWNER's private stream; nothing orders them
 * against work the caller has in flight on device j, so d_all[*] must be idle (already synchronised) when the call starts;
 * the call returns after every copy has completed (host-synchronised), so consumers need no further ordering.  Pairs without
 * peer access are not an error (the runtime stages those copies through the host); vp_group_peer_access reports them. */
VP_API int vp_group_create(vp_group_handle* out, const vp_config* cfg, const int32_t* device_ids, int32_t n_devices);
/* number of ordered device pairs (i != j) of the group for which peer access could NOT be enabled (0 on an xGMI-connected node) */
VP_API int vp_group_peer_access_missing(vp_group_handle g);
VP_API int vp_group_size(vp_group_handle g);
VP_API vp_handle vp_group_member(vp_group_handle g, int32_t i);
VP_API int vp_group_load_weights(vp_group_handle g, const vp_tensor_desc* tensors, int32_t n_tensors);
VP_API int vp_group_infer(vp_group_handle g, const void* crops, int32_t input_format, int32_t n, const int32_t* org_wh, float* out);
VP_API int vp_group_infer_allgather(vp_group_handle g, const void* crops, int32_t input_format, int32_t n, const int32_t* org_wh,
                                    float* const* d_all, float* out);
VP_API int vp_group_destroy(vp_group_handle g);
VP_API const char* vp_group_last_error(vp_group_handle g);

/* Whole-frame entry: the crop loop of VitInference.inference (inference.py:259-266) on device.
 * frame = uint8 RGB [fh, fw, 3] on the host; crop_params = n x 8 int32 {x0, y0, cw, ch, left_pad, top_pad, pw, ph}:
 * the +10 px padded & clipped box (:261-262) and the zero-pad geometry of pad_image (vit_utils/inference.py:41-70).
 * One H2D of the frame; crop + zero-pad + OpenCV-style 8-bit bilinear resize (:316) + normalisation run on device.
 * out = float32 [n, K, 3] (y, x, conf) in padded-crop pixels; the caller adds (y0 - top_pad, x0 - left_pad) as :270 does. */
VP_API int vp_infer_frame(vp_handle h, const uint8_t* frame, int32_t fh, int32_t fw, const int32_t* crop_params,
                          int32_t n, float* out);

/* Flip-test inference (the optional accuracy mode of the reference head, topdown_heatmap_simple_head.py:195-218 with
 * flip_back of vit_utils/post_processing/post_transforms.py:110-147; `flip_test=True, shift_heatmap=False` in
 * configs/ViTPose_common.py:91-93): the model runs on the crops and on their left-right mirror, the mirrored heatmaps are
 * flipped back (joint pairs swapped, x reversed, optionally shifted one pixel right) and averaged with the first ones, the
 * average is decoded.  flip_pairs = n_pairs x 2 joint indices (the dataset's mirror pairs; the reference ships none).
 * out [n,K,3] and / or heatmaps [n,K,64,48] (either may be NULL, not both). */
VP_API int vp_infer_flip(vp_handle h, const void* crops, int32_t input_format, int32_t n, const int32_t* org_wh,
                         const int32_t* flip_pairs, int32_t n_pairs, int32_t shift_heatmap, float* out, float* heatmaps);

/* Parity/debug taps.  Heatmaps = ViTPose.forward output, float32 [N, K, 64, 48]. */
VP_API int vp_infer_heatmaps(vp_handle h, const void* crops, int32_t input_format, int32_t n, float* heatmaps);
/* Backbone output after last_norm (vit.py:387), float32 [N, 192, D]. */
VP_API int vp_infer_tokens(vp_handle h, const void* crops, int32_t input_format, int32_t n, float* tokens);

/* Decode alone: keypoints_from_heatmaps(unbiased=True, use_udp=True) + postprocess
 * (vit_utils/top_down_eval.py:493-641, easy_ViTPose/inference.py:187-205), one crop
 * at a time semantics.  heatmaps float32 [N, K, 64, 48] on the host. */
VP_API int vp_decode_only(int32_t device_id, const float* heatmaps, int32_t n, int32_t k,
                   const int32_t* org_wh, float* out);

/* HIP stream of the handle (a hipStream_t), for callers that order their own work against it. */
VP_API void* vp_stream(vp_handle h);
VP_API int vp_synchronize(vp_handle h);

/* HIP-event timing per kernel family: family_mask bit f (1 << VP_PROF_*) turns timing of
 * family f on (two event records around each of its launches); -1 = all, 0 = off. */
VP_API int vp_set_profiling(vp_handle h, int32_t family_mask);
VP_API int vp_reset_profile(vp_handle h);
VP_API int vp_get_profile(vp_handle h, vp_profile* out);
/* Name of the kernel the LAST launch of family `family` (VP_PROF_*) ran on, as the profiler prints it minus the namespace
 * (e.g. "gemm8_kernel<F16, 1, G8<256>>", "gemm_kernel<F16, 6, 0, TileCfg<192, 128, 64, 48, 64, 2, 1, 0>>"), written by the
 * launch code itself: bench.py reads `roofline.kernel` from here instead of restating the selection rule.  Empty string when
 * the family has not been launched yet.  Returns VP_ERR_INVALID for a bad family / NULL buffer. */
VP_API int vp_profile_kernel(vp_handle h, int32_t family, char* buf, int32_t cap);

VP_API int vp_destroy(vp_handle h);

/* Last error text of this handle (or of the failed vp_create when h == NULL).
 * The pointer stays valid until the next call on the same handle. */
VP_API const char* vp_last_error(vp_handle h);

/* ---- parity taps (used by tests/ only): run ONE kernel on host fp32 data; operands are
 * rounded to `dtype` exactly as the production packer / producing kernels round them. ---- */
/* out[M,N] = epilogue(A[M,K] . W[N,K]^T); epi: 0 = +bias (nn.Linear, vit.py:166), 1 = gelu(+bias)
 * (Mlp fc1+act, vit.py:137-138), 2 = +bias +aux[M,N] (residual add, vit.py:203-204),
 * 3 = +aux[m % 192] (patch embed + pos, vit.py:382).  K % 64 == 0. */
VP_API int vp_dbg_gemm(int32_t device_id, int32_t dtype, int32_t epi, int32_t M, int32_t N, int32_t K,
                       const float* A, const float* W, const float* bias, const float* aux, float* out);
/* qkv [B*192, 3*D] -> attention core output [B*192, D]  (vit.py:167-176) */
VP_API int vp_dbg_attention(int32_t device_id, int32_t dtype, int32_t B, int32_t D, int32_t heads,
                            const float* qkv, float* out);
/* attn.qkv + attention core in ONE kernel (csrc/qkvattn.hip; head dim 64): x [2 npairs 192, D], Wqkv [3D, D], bias [3D] -> [M, D].  Neutral
 * LayerNorm statistics: the result must equal vp_dbg_gemm(epi 0) + vp_dbg_attention bit for bit.  npairs * heads >= 8. */
VP_API int vp_dbg_qkvattn(int32_t device_id, int32_t dtype, int32_t npairs, int32_t D, int32_t heads, const float* x, const float* W,
                          const float* bias, float* out);
/* LayerNorm(eps=1e-6) of x [M,D]: out16 = result rounded to dtype (returned as fp32), out32 = fp32 result */
VP_API int vp_dbg_layernorm(int32_t device_id, int32_t dtype, int32_t M, int32_t D, const float* x,
                            const float* gamma, const float* beta, float* out16, float* out32);
/* ConvTranspose2d(Cin,256,4,2,1)+BN(eval)+ReLU (topdown_heatmap_simple_head.py:303-319) on NHWC
 * x [B,Hin,Win,Cin] -> NHWC [B,2Hin,2Win,256]; tensors named keypoint_head.deconv_layers.{0,1}.* */
VP_API int vp_dbg_deconv(int32_t device_id, int32_t dtype, int32_t B, int32_t Hin, int32_t Win, int32_t Cin,
                         const float* x, const vp_tensor_desc* tensors, int32_t n_tensors, float* out);
/* the device crop / zero-pad / resize kernel alone: uint8 crops [n, 256, 192, 3] */
VP_API int vp_dbg_crop_prep(int32_t device_id, const uint8_t* frame, int32_t fh, int32_t fw, const int32_t* crop_params,
                            int32_t n, uint8_t* out);
/* one launch of a production GEMM configuration on HOST data (layouts built / undone inside): variant = tile configuration (gemm.hip Cfg id;
 * 16 / 17 / 18 = the 8-phase kernel of gemm8.hip with 256x256 / 256x192 / 192x256 tiles); flags: 1 persistent workgroups, 2 64x64-blocked output,
 * 4 64x64-blocked A operand, 8 reversed tile walk, 16 LayerNorm-consumer fold, bits 8-11 = S > 1: split-K (epi 6 only: S partial products over k ranges +
 * the fixed-order reduction kernel, the small-batch path of attn.proj / mlp.fc2); epi 0-3, 5 (final 1x1 conv with hi+lo
 * weights -> heatmaps [M/3072, N, 3072]), 6, 7; rowstat [M,2] + ln_s [N] = LayerNorm-consumer fold; stats [M, N/64, 2] (epi 6 / 7) */
VP_API int vp_dbg_gemm_case(int32_t device_id, int32_t dtype, int32_t epi, int32_t variant, int32_t group_m, int32_t flags,
                            int32_t M, int32_t N, int32_t K, const float* A, const float* W, const float* bias, const float* aux,
                            const float* rowstat, const float* ln_s, float* out, float* stats);
/* The sharding plan of vp_group_infer for n crops on w devices of max_batch maxb -- HOST ONLY, no device needed: the exact
 * function group_run executes.  Rounds of w * maxb crops; inside a round device i takes [off, off + cnt) with ceil(nr / w) crops
 * per device (trailing devices short or empty).  Entry e = round * w + device: offs[e], cnts[e].  Returns the number of
 * entries (rounds * w), also when it exceeds `cap` (nothing is written beyond cap); < 0 on bad arguments. */
VP_API int vp_dbg_group_plan(int32_t n, int32_t w, int32_t maxb, int32_t* offs, int32_t* cnts, int32_t cap);
/* HOST ONLY: which tile of the 8-phase GEMM kernel the selection rule picks for an [M, N] output -- 0 = none (2-phase kernels), 16 = 256 x 256,
 * 17 = 256 x 192, 18 = 192 x 256; wide != 0: 16-bit-output GEMMs (qkv, fc1), else the residual GEMMs; bm192_mask bits 1 / 2 as VP_G8_BM192, bit 4 = the
 * round-3 thresholds (as VP_G8_COST=0); *tiles (may be NULL) = its tile count */
VP_API int vp_dbg_gemm8_pick(int32_t M, int32_t N, int32_t wide, int32_t bm192_mask, int32_t* tiles);
/* HOST ONLY: which tile configuration of the 2-phase GEMM kernel (gemm.hip Cfg id: 8 / 11 = 192 x 128 with 4 / 8 waves, 1 = 128 x 128, 9 / 12 = 64 x 64 on a 2- / 4-stage
 * ring, 15 = 128 x 64 3-stage, 30 / 31 = 64 x 64 / 32 x 64 with two k-blocks per barrier) the selection rule picks for one GEMM of the path: epi = 0 bias (qkv), 1 bias + GELU
 * (fc1), 4 deconv, 5 final 1x1 conv, 6 residual + row statistics (attn.proj, mlp.fc2), 7 patch embed; [M, N] output, depth K; *group_m (may be NULL) = its tile-order group.
 * (Large batches: the 8-phase kernel takes the encoder GEMMs over where vp_dbg_gemm8_pick says so.) */
VP_API int vp_dbg_gemm2_pick(int32_t epi, int32_t M, int32_t N, int32_t K, int32_t* group_m);
/* HOST ONLY: the split-K rule of the residual GEMMs at small batches (attn.proj / mlp.fc2 of [M, N] x K): returns the number of k ranges S (1 = the one-launch
 * residual epilogue; S > 1 = S partial products + the fixed-order reduction kernel), *variant (may be NULL) = the gemm.hip Cfg id of the partial products */
VP_API int vp_dbg_splitk_pick(int32_t M, int32_t N, int32_t K, int32_t* variant);
/* HOST ONLY: the batch the ENCODER runs for a chunk of n crops of a model of embed dim D (>= n; limit = the largest batch the workspaces hold).  From 33 crops on a chunk
 * that is no multiple of 4 crops runs the next multiple of 4 where that moves mlp.fc1 / mlp.fc2 onto the 8-phase kernel's 256-row tiles (or saves a round of them): the padding
 * rows repeat the last crop, every kernel works row by row or crop by crop, so the real crops' keypoints are bit for bit those of the unpadded run; the head and the decode run
 * the real crops only.  VP_PAD_BATCH=0 switches the padding off. */
VP_API int vp_dbg_run_batch(int32_t n, int32_t D, int32_t limit);
/* The two-phase schedule of a group call -- HOST ONLY, stub members: the order in which group_run would submit to (+ (member + 1)) and
 * wait for (- (member + 1)) its members for n crops on w devices of max_batch maxb.  Within every round all submissions precede the
 * first wait: no member's enqueue waits for another member's compute.  Returns the trace length (also beyond `cap`); < 0 on bad arguments. */
VP_API int vp_dbg_group_trace(int32_t n, int32_t w, int32_t maxb, int32_t* trace, int32_t cap);
/* BASELINE config 5 probe (fp8 is documented tolerance-infeasible; this confirms the CPU emulation the
 * finding rests on, tests/fp8_budget.py, on the hardware): rows of A [M,K] and W [N,K] are quantised on device to OCP e4m3
 * (x / scale[row], v_cvt_pk_fp8_f32) and multiplied through v_mfma_f32_16x16x128_f8f6f4:
 * out[m][n] = a_scale[m] w_scale[n] sum_k a8 w8.  a_codes / w_codes (uint8 [M,K] / [N,K], may be NULL) return the codes.
 * M, N multiples of 16, K a multiple of 128. */
VP_API int vp_dbg_fp8_gemm(int32_t device_id, int32_t M, int32_t N, int32_t K, const float* A, const float* a_scale, const float* W,
                           const float* w_scale, float* out, uint8_t* a_codes, uint8_t* w_codes);
/* MX probe (the opt-in fp8 mode's operand format, csrc/mx8.h): rows of A [M,K] are quantised on device to MXFP8 -- OCP e4m3 codes with
 * one E8M0 power-of-two scale per block of 32 consecutive k -- W [N,K] to e4m3 with the given per-row scale, and multiplied through the
 * BLOCK-SCALED v_mfma_scale_f32_16x16x128_f8f6f4 with the operand roles and packed scale dwords of the production GEMM.  Returns the
 * product and (optionally) the codes / scale bytes in the library's layouts so that a test can restate the arithmetic exactly.
 * M % 64 == 0, N % 16 == 0, K % 128 == 0. */
VP_API int vp_dbg_mx_gemm(int32_t device_id, int32_t M, int32_t N, int32_t K, const float* A, const float* W, const float* w_scale, float* out,
                          uint8_t* a_codes, uint8_t* a_scales, uint8_t* w_codes);
/* ONE launch of the fp8 mode's GEMM kernel (csrc/gemm8f.hip) on host fp32 data: A [M,K] is quantised to MXFP8 on device, W [N,K] on the host
 * as the weight packer does; a_deq [M,K] / w_deq [N,K] (may be NULL) return the values the codes and scales stand for.  epi 0: out = fp16(a.w^T
 * + bias); epi 1: out = gelu(a.w^T + bias) written as MXFP8 and returned de-quantised; epi 6: out = a.w^T + bias + aux as the two-plane residual
 * (hi + lo returned), stats [M, N/64, 2].  M % 256 == 0, K % 256 == 0, K >= 512, N % 256 == 0 (epi 6: N % 192 == 0 or N % 256 == 0). */
VP_API int vp_dbg_gemm_fp8_case(int32_t device_id, int32_t epi, int32_t M, int32_t N, int32_t K, const float* A, const float* W, const float* bias,
                                const float* aux, float* out, float* stats, float* a_deq, float* w_deq);
/* HOST ONLY: fp32 -> OCP e4m3 codes with the converter the fp8 weight packer uses (round to nearest even, saturating at 448) */
VP_API int vp_dbg_host_e4m3(const float* in, uint8_t* out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* VITPOSE_HIP_H */
