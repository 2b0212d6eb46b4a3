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

#ifdef __cplusplus
}
#endif
#endif /* VITPOSE_HIP_TOOLS_H */
