/*
 * vitpose_hip_tools.h -- entry points of the MEASUREMENT build only (libvitpose_hip_tools.so = the sources of
 * libvitpose_hip.so compiled with -DVP_TOOLS, easy_vitpose_amd/build.py): cycle-stamp timelines of the GEMM kernels.  The
 * measurement build additionally honours GemmArgs::ablate / ::stagger inside the kernels, instantiates the experimental tile
 * configurations (gemm.hip Cfg0/2/4/5/6/7/10/13/14/15) and reads the
 * development environment switches.  tools/ loads it; the product path and tests/ never do.
 */
#ifndef VITPOSE_HIP_TOOLS_H
#define VITPOSE_HIP_TOOLS_H

#include "vitpose_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tools/gemm_timeline.py: per-tile phase stamps (shader cycles) of one persistent qkv / fc1 launch:
 * stamps[max_wg][32][8] = (main loop start, main loop end, epilogue end, 5 stamps inside k-step 5) of wave 0 of every
 * workgroup. */
VP_API int vp_dbg_gemm_timeline(int32_t device, int32_t dtype, int32_t epi, int32_t M, int32_t N, int32_t K, uint64_t* stamps,
                                int32_t max_wg);
/* tools/gemm8_timeline.py: cycle stamps of one gemm8 launch, stamps[max_wg >= 256][2 wave groups][16 tiles][8] */
VP_API int vp_dbg_gemm8_timeline(int32_t device_id, int32_t dtype, int32_t epi, int32_t variant, int32_t flags, int32_t ablate,
                                 int32_t M, int32_t N, int32_t K, uint64_t* stamps, int32_t max_wg);

/* tools/qkvattn_phases.py: average milliseconds per launch of the fused qkv + attention kernel (csrc/qkvattn.hip) on random operands; ablate bits:
 * 1 = no attention phase, 2 = no epilogue (hand-over of q / k / v through LDS), 4 = K-loop cut to four K-tiles */
VP_API int vp_dbg_qkvattn_bench(int32_t device_id, int32_t npairs, int32_t D, int32_t heads, int32_t iters, int32_t ablate, float* ms_out);

/* tools/hwid_probe.py: where the workgroups of a launch land.  out[blocks][4] = (HW_REG_HW_ID, HW_REG_XCC_ID, start cycle lo, hi) of every
 * workgroup of a `blocks` x `threads` launch with `lds_bytes` of dynamic LDS whose workgroups stay resident for ~spin x 4096 cycles. */
VP_API int vp_dbg_hwid_probe(int32_t device_id, int32_t blocks, int32_t threads, int32_t lds_bytes, int32_t spin, uint32_t* out);

/* ---- timing taps (moved out of the product library in round 6: the product .so exports parity taps only) ---- */
/* average milliseconds of `iters` launches of one GEMM tile configuration on random device operands */
VP_API int vp_dbg_gemm_bench(int32_t device_id, int32_t dtype, int32_t epi, int32_t variant, int32_t group_m,
                             int32_t M, int32_t N, int32_t K, int32_t iters, float* ms_out);
/* production GEMM configurations on RANDOM device operands (tools/gemm8_check.py, tests/test_gpu_gemm_cfgs.py):
 * epi = 0 bias, 1 bias+gelu, 2 bias+residual (fp32), 3 pos (fp32), 6 bias + two-plane residual + LayerNorm row statistics;
 * variant = tile configuration (gemm.hip Cfg0-11; 16 / 17 = the 8-phase kernel of gemm8.hip with 256x256 / 256x192 tiles);
 * flags: 1 persistent workgroups, 2 64x64-blocked output, 4 64x64-blocked A operand, 8 reversed tile walk, 16 LayerNorm-consumer fold.
 * bench2: average milliseconds per launch.  compare: both configurations on the same operands, `reps` times; counts every
 * differing output element / statistic (two kernels with the same accumulation order must agree bit for bit). */
VP_API int vp_dbg_gemm_bench2(int32_t device_id, int32_t dtype, int32_t epi, int32_t variant, int32_t group_m, int32_t flags,
                              int32_t M, int32_t N, int32_t K, int32_t iters, float* ms_out);
VP_API int vp_dbg_gemm_compare(int32_t device_id, int32_t dtype, int32_t epi, int32_t variant_a, int32_t group_a, int32_t flags_a,
                               int32_t variant_b, int32_t group_b, int32_t flags_b, int32_t M, int32_t N, int32_t K, int32_t reps,
                               uint64_t* n_mismatch, double* max_abs_diff);
/* calibration of the box: kind 0/1 = MFMA-only loop 16x16x32 / 32x32x16 f16 (TFLOP/s), 2 = float4 copy (TB/s) */
VP_API int vp_dbg_peak(int32_t device_id, int32_t kind, double* result);


#ifdef __cplusplus
}
#endif
#endif /* VITPOSE_HIP_TOOLS_H */
