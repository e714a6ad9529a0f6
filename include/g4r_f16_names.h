/* g4r_f16_names.h -- the IEEE-half (fp16) instantiation of the inference entry points.
 *
 * The reference's serving path runs fp16 end to end: gpt4roi/app.py:74-98 loads the model with torch_dtype=float16, the
 * boxes are `.half()` at :271 and the image at :296.  libgpt4roi_hip.so therefore carries every kernel of the inference
 * path twice, compiled from the SAME sources (gpt4roi_amd/csrc/g4r_common.h: `h16_t`, G4R_MFMA_32X32X16): once storing
 * bfloat16 (the reference's training dtype, train_stage1.sh:19) under the names of g4r_kernels.h, and once storing IEEE
 * half under the names below -- identical signatures and contracts, every `void*` tensor that g4r_kernels.h documents as
 * bf16 is fp16 instead; fp32 accumulation, statistics and logits are unchanged.
 *
 * A C consumer that wants the fp16 entry points declared includes this file BEFORE g4r_kernels.h / g4r_roi_align.h in a
 * translation unit of its own (the macros rename the declarations), or simply dlsym()s the names on the right.
 * Training-only entry points (g4r_train.h) exist in bf16 only, as the reference trains in bf16.
 */
#ifndef G4R_F16_NAMES_H
#define G4R_F16_NAMES_H
#define g4r_gemm_bf16_nt_partials           g4r_gemm_f16_nt_partials
#define g4r_gemm_qkv_rope_bf16              g4r_gemm_qkv_rope_f16
#define g4r_gemm_bf16_nt                    g4r_gemm_f16_nt
#define g4r_gemv_rmsnorm_bf16               g4r_gemv_rmsnorm_f16
#define g4r_gemv_batch_bf16                 g4r_gemv_batch_f16
#define g4r_gemv_attn_merge_bf16            g4r_gemv_attn_merge_f16
#define g4r_conv3x3_nhwc_bf16               g4r_conv3x3_nhwc_f16
#define g4r_conv3x3_mlvl_nhwc_bf16          g4r_conv3x3_mlvl_nhwc_f16
#define g4r_attn2_dispatch                  g4r_attn2_dispatch_f16
#define g4r_flash_attn_fwd_bf16             g4r_flash_attn_fwd_f16
#define g4r_attn_decode_bf16                g4r_attn_decode_f16
#define g4r_attn_decode_ragged_bf16         g4r_attn_decode_ragged_f16
#define g4r_groupnorm_affine_mlvl_nhwc_bf16 g4r_groupnorm_affine_mlvl_nhwc_f16
#define g4r_layernorm_bf16                  g4r_layernorm_f16
#define g4r_rmsnorm_bf16                    g4r_rmsnorm_f16
#define g4r_layernorm_splitk_bf16           g4r_layernorm_splitk_f16
#define g4r_rmsnorm_splitk_bf16             g4r_rmsnorm_splitk_f16
#define g4r_groupnorm_affine_nhwc_bf16      g4r_groupnorm_affine_nhwc_f16
#define g4r_upsample_coord_nhwc_bf16        g4r_upsample_coord_nhwc_f16
#define g4r_fuse_shuffle_mlvl_nhwc_bf16     g4r_fuse_shuffle_mlvl_nhwc_f16
#define g4r_fuse_shuffle_nhwc_bf16          g4r_fuse_shuffle_nhwc_f16
#define g4r_im2col_patch14_f32              g4r_im2col_patch14_f32_f16
#define g4r_vit_assemble_bf16               g4r_vit_assemble_f16
#define g4r_rope_qkv_bf16                   g4r_rope_qkv_f16
#define g4r_swiglu_bf16                     g4r_swiglu_f16
#define g4r_splice_embed_bf16               g4r_splice_embed_f16
#define g4r_add_rows_bf16                   g4r_add_rows_f16
#define g4r_cast_f32_to_bf16                g4r_cast_f32_to_f16
#define g4r_conv3x3_weight_layout_bf16      g4r_conv3x3_weight_layout_f16
#define g4r_roi_align_mlvl_nhwc_bf16        g4r_roi_align_mlvl_nhwc_f16
#endif
