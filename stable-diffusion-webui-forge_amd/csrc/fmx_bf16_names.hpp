// Symbol names of the bfloat16 build of the 16-bit kernels (compiled with -DFMX_ELEM_BF16, see fmx_common.hpp and the Makefile).
// Included before anything else, so the prototypes of include/fmx.h, the definitions and the cross-file launchers are all renamed
// consistently; the element-type-independent entry points of those files are left out of this build (#ifndef FMX_ELEM_BF16).
#pragma once
#define fmx_gemm_conv_f16 fmx_gemm_conv_bf16
#define fmx_attention_f16 fmx_attention_bf16
#define fmx_softmax_rows_f16 fmx_softmax_rows_bf16
#define fmx_layernorm_f16 fmx_layernorm_bf16
#define fmx_layernorm_padded_f16 fmx_layernorm_padded_bf16
#define fmx_layernorm_mod_f16 fmx_layernorm_mod_bf16
#define fmx_rmsnorm_f16 fmx_rmsnorm_bf16
#define fmx_flux_qk_norm_rope_f16 fmx_flux_qk_norm_rope_bf16
// the VAE in bfloat16 (the reference's VAE type on bf16-capable parts, backend/memory_management.py:190-205, :840-855): GroupNorm, the GEMM
// with output statistics, the 512-wide single-head attention
#define fmx_gemm_conv_stats_f16 fmx_gemm_conv_stats_bf16
#define fmx_groupnorm_stats_f16 fmx_groupnorm_stats_bf16
#define fmx_groupnorm_apply_f16 fmx_groupnorm_apply_bf16
#define fmx_attention_single_head512_f16 fmx_attention_single_head512_bf16
#define fmx_conv3x3_narrow_f16 fmx_conv3x3_narrow_bf16
#define fmx_launch_gn_stats fmx_launch_gn_stats_bf16
#define fmx_launch_gn_finalize fmx_launch_gn_finalize_bf16
#define fmx_conv3x3_gn_silu_f16 fmx_conv3x3_gn_silu_bf16
#define fmx_conv3x3_up2x_f16 fmx_conv3x3_up2x_bf16
#define fmx_conv3x3_narrow_gn_silu_f16 fmx_conv3x3_narrow_gn_silu_bf16
// host-side C++ symbols shared between the GEMM files
#define fmx_launch_gemm256p fmx_launch_gemm256p_bf16
#define fmx_launch_gemm4w fmx_launch_gemm4w_bf16
#define GemmParams GemmParamsBf16
#define GemmEpilogue GemmEpilogueBf16
#define FastEpilogue FastEpilogueBf16
