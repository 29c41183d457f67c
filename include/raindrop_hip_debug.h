/*
 * raindrop_hip_debug.h -- profiling hooks of libraindrop_hip.so.  NOT part of the C-ABI (include/raindrop_hip.h): no host binding
 * in raindrop_amd/_lib.py, no stability promise; used by tools/*_timing.py only.  Each setter registers (process-wide) a device
 * buffer into which the named kernels write clock64() stamps per phase, or restores normal operation when passed NULL; the
 * pointer travels to the kernels as an argument of the launches enqueued while it is registered.
 */
#ifndef RAINDROP_HIP_DEBUG_H
#define RAINDROP_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

void rd_debug_set_stamps(void* device_u64);          /* k_msg_fwd_fused / k_msg_bwd_fused   (tools/fused_timing.py)   */
void rd_debug_set_rowgemm_stamps(void* device_u64);  /* k_rowgemm                            (tools/rowgemm_timing.py) */
void rd_debug_set_attn_stamps(void* device_u64);     /* k_attn_*_one_b16*                    (tools/attn_timing.py)    */
void rd_debug_set_gemm_stamps(void* device_u64);     /* k_gemm_bf16x3                        (tools/gemm_timing.py)    */
void rd_debug_set_encfuse_stamps(void* device_u64);  /* k_enc_post_fwd / k_enc_pre_bwd       (tools/encfuse_timing.py) */
void rd_debug_set_head_stamps(void* device_u64);     /* k_head_rows                          (tools/head_timing.py)    */
void rd_debug_set_attnfuse_stamps(void* device_u64); /* k_attn_fwd_fused / k_attn_bwd_fused  (tools/attnfuse_timing.py) */
void rd_debug_ifetch_probe(void* ticks_device_u64, void* scratch_64_floats, void* stream);   /* 32 KB of straight-line code, one wave: clock64 ticks (bench.py config.box; tools/probe_clocks.hip) */
void rd_debug_set_splitk_want(int workgroups);       /* target workgroup count of the split-K weight-gradient plan     */

#ifdef __cplusplus
}
#endif
#endif /* RAINDROP_HIP_DEBUG_H */
