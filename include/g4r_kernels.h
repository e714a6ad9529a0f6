/*
 * g4r_kernels.h -- C ABI of the remaining gfx950 kernels of the region-feature path
 * (everything except RoIAlign, which has its own header g4r_roi_align.h).
 *
 * The reference has no native boundary for these stages: it reaches cuBLAS / cuDNN / ATen
 * through torch.nn modules.  Each entry point below names the reference call site whose
 * arithmetic it carries (paths relative to /root/reference).  Common contract: device
 * pointers owned by the caller, no allocation, no retained state, asynchronous on `stream`
 * (hipStream_t), int status return (g4r_roi_align.h: G4R_OK / G4R_ERR_*).  "bf16" pointers are
 * bfloat16 bit patterns (uint16); strides are in ELEMENTS.
 */
#ifndef G4R_KERNELS_H
#define G4R_KERNELS_H

#ifdef __cplusplus
extern "C" {
#endif

/*
 * C[M,N] = act(A[M,K] . W[N,K]^T + bias) + residual          (torch.nn.Linear layout)
 *   - mm_projector              llava/model/llava.py:52,76 ; gpt4roi/models/spi_llava.py:89-97
 *   - flatten_linear / updims / pos_embedd   gpt4roi/models/layers.py:260-270, 326-329
 *   - 1x1 input_conv (as a GEMM over NHWC pixels)   layers.py:129-132, 191
 *   - every projection inside CLIP ViT-L/14 and LLaMA-7B (HF modules called at
 *     spi_llava.py:66-67 and :198-205) and lm_head (llava.py:235-238)
 * A, W, residual bf16; bias fp32 [N] or NULL; C bf16 (out_f32 = 0) or fp32 (out_f32 = 1).
 * act: 0 none, 1 relu, 2 quick_gelu, 3 silu.  K % 64 == 0 runs the MFMA kernel (lda, ldw
 * multiples of 8); any other K runs a scalar kernel (tiny layers only, no residual).
 * splits > 1: split-K with `workspace` of splits*M*N floats.
 * tile_cfg (the tiles the shipped library holds; gpt4roi_amd/kernels.py pick_tile chooses among them by whole waves of the
 * 256 CUs): 0 = 128x128 two-stage, 4 = 64x128, 7 = 128x128 x 8 waves ring of 4, 13 = 64x128 ring of 3, 14 = 64x64 ring of
 * 4, 24 = 256x256 ring ping-pong (also the route for operands of 2 GiB and more), 28 = 192x256 ring ping-pong, 34 = 256x256
 * one wave per SIMD, K tiles of 64 -- the production tile; launches of more than one wave of its tiles with N % 256 == 0,
 * K >= 2048, 16-bit output and no K slices run its PERSISTENT form (one workgroup per CU walks the tiles; bit-identical).
 * 36 = tile 34 with the persistent form offered at every K (tests of its short-K paths).
 * Any other number returns G4R_ERR_INVALID_ARG (superseded forms exist in the tools build only, -DG4R_TOOLS_BUILD).
 */
int g4r_gemm_bf16_nt(const void* A, const void* W, void* C, const float* bias, const void* residual,
                     float* workspace, int M, int N, int K, int lda, int ldw, int ldc, int ldr,
                     int act, int out_f32, int splits, int tile_cfg, void* stream);

/*
 * 3x3 / stride 1 / pad 1 convolution over NHWC bf16 as an implicit GEMM on MFMA:
 *   Y[b,y,x,:] = act( sum_g sum_tap W[:, g, tap, :] . X_g[b, y+dy, x+dx, :] + bias )
 *   - MLVLFuseModule.fuse_convs (ConvModule 3x3, no bias)   gpt4roi/models/layers.py:133-144
 *   - MlvlRoIExtractor.pconvs, all `groups` levels summed in one accumulator
 *     (layers.py:257-259, 321-324): X_g = X + g * x_group_stride, images = RoIs of 14x14.
 * X [groups][batch, H, Wd, Cin]; W [Cout][groups][9][Cin] (tap = ky*3+kx); Y [batch, H, Wd, Cout].
 * `zeros`: >= 128 bytes of device zeros (padding source).  Cin % 64 == 0.
 */
int g4r_conv3x3_nhwc_bf16(const void* X, const void* W, void* Y, const float* bias, const void* zeros,
                          float* workspace, int batch, int H, int Wd, int Cin, int Cout, int groups,
                          long x_group_stride, int act, int out_f32, int splits, int tile_cfg,
                          void* stream);

/*
 * The 3x3 / pad 1 convolutions of one MLVLFuseModule round over ALL pyramid levels in one implicit GEMM
 * (gpt4roi/models/layers.py:218-236 runs the same ConvModule on every level): X and Y hold the levels' NHWC maps stacked
 * [level][batch][y][x][C] (level l has level_h[l] x level_w[l] pixels), W as for g4r_conv3x3_nhwc_bf16 (groups = 1),
 * bias nullable.  Same arithmetic per output pixel as one g4r_conv3x3_nhwc_bf16 call per level.
 */
int g4r_conv3x3_mlvl_nhwc_bf16(const void* X, const void* W, void* Y, const float* bias, const void* zeros, int n_levels,
                               const int* level_h, const int* level_w, int batch, int Cin, int Cout, int act,
                               void* stream);

/*
 * softmax(Q K^T * scale [+ causal mask]) V, bf16, head_dim 64 (CLIP ViT-L/14) or 128 (LLaMA-7B).
 * Replaces the attention inside HF CLIPAttention / LlamaAttention (called from
 * spi_llava.py:66-67, 198-205) and the flash-attn patch
 * (llava/train/llama_flash_attn_monkey_patch.py:15-91).
 * Q [B, Tq, H*D], K/V [B, Tk, H*D], O [B, Tq, H*D] addressed through row / batch strides.
 * causal: query i attends keys <= i + (Tk - Tq)  (KV-cache decode when Tq < Tk).
 * kv_len_dev (nullable): when set, Tk = *kv_len_dev + Tq is computed on the device (the argument Tk is then
 * only an upper bound): *kv_len_dev = positions already cached before this call, so a decode step can be
 * replayed from a hipGraph while the cache grows.
 * lse (nullable): fp32 [B, H, Tq], receives the log2-domain log-sum-exp of the scaled scores for the backward
 * (g4r_train.h).
 */
int g4r_flash_attn_fwd_bf16(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq,
                            int Tk, int head_dim, long q_row, long k_row, long v_row, long o_row,
                            long q_batch, long k_batch, long v_batch, long o_batch, float scale,
                            int causal, const int* kv_len_dev, float* lse, void* stream);

/*
 * One activation row through a projection, with the RMSNorm in front of it fused in: the per-token form of
 * `LlamaRMSNorm` + `nn.Linear` that the decode loop of gpt4roi/app.py:293-300 runs 4x per layer (HF LlamaDecoderLayer).
 *   C [N] = act(W [N, K] . h + bias) + residual,  h = gamma ? bf16(bf16(x * rsqrt(mean(x^2) + eps)) * gamma) : x
 * x [K] bf16, gamma fp32 [K] or null, W bf16 rows `ldw` elements apart, C bf16 (or fp32 with out_f32), act as in
 * g4r_gemm_bf16_nt (4 = SwiGLU over interleaved (gate, up) rows: C has N/2 entries).  K >= 512, K % 8 == 0,
 * K <= 8192 with gamma (<= 32768 without).  h is bit-identical to g4r_rmsnorm_bf16's output.
 */
int g4r_gemv_rmsnorm_bf16(const void* x, const float* gamma, float eps, const void* W, void* C, const float* bias,
                          const void* residual, int N, int K, int ldw, int act, int out_f32, void* stream);

/*
 * B = 2..16 activation rows through a projection (round 5, csrc/gemv_mfma.hip): the batched form of g4r_gemv_rmsnorm_bf16 --
 * the decode step of several sequences sharing ONE pass over the weights, what HF `generate()` runs per token for a batch
 * (gpt4roi/app.py:293-300 with a batched prompt).  C [B, N] = act(W [N, K] . h_b + bias) + residual_b with
 * h_b as above (gamma null: h_b = x_b); x / C / residual rows are `ldx` / `ldc` / `ldr` elements apart; act 4 = SwiGLU over
 * interleaved (gate, up) rows (C has N / 2 columns).  K >= 512, K % 64 == 0.  h_b is bit-identical to g4r_rmsnorm_bf16's output;
 * the products run on the matrix pipe (v_mfma_f32_16x16x32, fp32 accumulation), the weights stream straight into its operand
 * registers.  variant: 0 (tools: other wave counts / row blocks / LDS budgets, see csrc/gemv_mfma.hip).
 */
int g4r_gemv_batch_bf16(const void* x, int B, long ldx, const float* gamma, float eps, const void* W, void* C, long ldc,
                        const float* bias, const void* residual, long ldr, int N, int K, int ldw, int act, int out_f32,
                        int variant, void* stream);

/*
 * Single-query attention over a KV cache: the per-token step of the decode loop the reference reaches through HF
 * `generate()` (gpt4roi/app.py:293-300 -> LlamaAttention with past_key_values).  Q/O [H*head_dim] bf16; K/V cache rows
 * `k_row`/`v_row` elements apart; the first Tk rows are attended, Tk = *kv_len_dev + 1 when kv_len_dev is given (read on
 * the device: the call is replayed from a hipGraph).  The keys of a head are split over `splits` workgroups that merge
 * inside the launch: workspace = H*splits*(head_dim+2) floats, counters = H uint32, zero before the first call (each call
 * leaves them zero).  Same result as g4r_flash_attn_fwd_bf16 with Tq = 1, causal, up to fp32 summation order.
 * qkv (nullable): the raw q|k|v projection row [3*H*head_dim] of the new token.  When given, Q is ignored and the call
 * also performs g4r_rope_qkv_bf16 for that token: q and k are rotated with row Tk-1 of the cos/sin tables
 * ([maxT, head_dim/2] fp32), the rotated k and v are written to row Tk-1 of the caches (LlamaAttention's
 * apply_rotary_pos_emb + cache append).
 * defer_merge: leave the per-split partials in `workspace` for g4r_gemv_attn_merge_bf16 (O and counters unused).
 * batch: number of equal-length sequences served by the launch (decode of B requests sharing one weight stream, SURVEY.md
 * 8d config 5); sequence b uses Q/qkv + b*q_batch, K + b*k_batch, V + b*v_batch, O + b*o_batch (elements) and the b-th set
 * of workspace (H*splits*(head_dim+2) floats) and counters (H).
 */
int g4r_attn_decode_bf16(const void* Q, const void* qkv, const float* cos_tab, const float* sin_tab, void* K, void* V,
                         void* O, float* workspace, unsigned* counters, int H, int head_dim, int Tk, long k_row,
                         long v_row, float scale, int splits, const int* kv_len_dev, int defer_merge, int batch,
                         long q_batch, long k_batch, long v_batch, long o_batch, void* stream);

/*
 * The same launch for a RAGGED batch -- prompts of different lengths, or a batch that carried a padding mask
 * (HF LlamaModel with attention_mask: masked keys are never attended and positions count the padded layout; the reference
 * reaches it through generate(), llava/model/llava.py:263-283; its training path unpads with the mask,
 * llava/train/llama_flash_attn_monkey_patch.py:60-85).  The pad rows are squeezed out of the cache at prefill, so sequence
 * b holds kv_lens_dev[b] rows: it attends rows [0, kv_lens_dev[b]] and appends the new token at row kv_lens_dev[b].  The
 * RoPE position of the new tokens is *rope_pos_dev for every sequence (the padded-layout position; null: the cache row).
 * `qkv` [batch, 3*H*head_dim] raw projection rows (always the fused form); other arguments as above.
 */
int g4r_attn_decode_ragged_bf16(const void* qkv, const float* cos_tab, const float* sin_tab, void* K, void* V, void* O,
                                float* workspace, unsigned* counters, int H, int head_dim, long k_row, long v_row,
                                float scale, int splits, const int* kv_lens_dev, const int* rope_pos_dev, int batch,
                                long q_batch, long k_batch, long v_batch, long o_batch, void* stream);

/*
 * o_proj of the decode step with the merge of the attention partials fused into its input staging:
 * C [N] = W [N, K] . a + bias + residual, a [K = H*head_dim] = the attention output assembled from
 * `partials` [H][splits][head_dim + 2] fp32 as written by g4r_attn_decode_bf16(..., defer_merge = 1).  Bit-identical to
 * g4r_attn_decode_bf16(defer_merge = 0) followed by g4r_gemv_rmsnorm_bf16(gamma = null).
 */
int g4r_gemv_attn_merge_bf16(const float* partials, int splits, int head_dim, const void* W, void* C, const float* bias,
                             const void* residual, int N, int K, int ldw, int out_f32, void* stream);

/* The per-level launches of a fuse round merged (round 3): GroupNorm affines of every level of a stacked pyramid in two
 * launches, and the "_single_shuffle" input assembly of every target level in one (gpt4roi/models/layers.py:152-195, the
 * same arithmetic as g4r_groupnorm_affine_nhwc_bf16 / g4r_fuse_shuffle_nhwc_bf16 per level). */
int g4r_groupnorm_affine_mlvl_nhwc_bf16(const void* x, const float* gamma, const float* beta, float* partial,
                                        float* scale_shift, int n_levels, const int* level_hw, int B, int C, int G,
                                        float eps, void* stream);
int g4r_fuse_shuffle_mlvl_nhwc_bf16(const void* const* maps, const float* const* affines, const int* heights,
                                    const int* widths, const int* top_idx, const int* down_idx, void* const* outs,
                                    int n_levels, int B, int C, void* stream);

/* Split-K GEMM with the reduce deferred to the consumer, and the consumer: the K-slice reduce of the LLaMA down_proj
 * (+ residual) folded into the following RMSNorm (HF LlamaDecoderLayer: x = x + mlp(...); h = input_layernorm(x) of the next
 * layer, spi_llava.py:198-205).  Bit-identical to g4r_gemm_bf16_nt(splits) + g4r_rmsnorm_bf16. */
int g4r_gemm_bf16_nt_partials(const void* A, const void* W, float* workspace, int M, int N, int K, int lda, int ldw,
                              int splits, int tile_cfg, int* splits_out, void* stream);
int g4r_rmsnorm_splitk_bf16(const float* partials, int splits, const void* residual, long ldr, void* x_out, long ldxo,
                            const float* gamma, void* y, long ldy, int rows, int cols, float eps, void* stream);
int g4r_layernorm_splitk_bf16(const float* partials, int splits, const float* bias, const void* residual, long ldr,
                              void* x_out, long ldxo, const float* gamma, const float* beta, void* y, long ldy, int rows,
                              int cols, float eps, void* stream);   /* CLIP fc2 (+bias, +residual) -> next layer_norm1 */

/* Fused q|k|v projection + RoPE + KV-cache append (HF LlamaAttention.forward: q/k/v_proj, apply_rotary_pos_emb, cache update;
 * the arithmetic spi_llava.py:198-205 delegates to).  See csrc/gemm_bf16.hip for the argument contract. */
int g4r_gemm_qkv_rope_bf16(const void* A, const void* W, int B, int T, int K, int lda, int ldw, int heads, int head_dim,
                           void* q_out, void* k_cache, void* v_cache, long cache_row, long cache_batch,
                           const float* cos_tab, const float* sin_tab, int pos0, int tile_cfg, void* stream);

/* LayerNorm over the last dim (CLIP pre_layrnorm / layer_norm1,2; pos_embedd LayerNorms
 * gpt4roi/models/layers.py:260-267).  gamma/beta fp32.  relu_in: apply ReLU to x first. */
int g4r_layernorm_bf16(const void* x, const float* gamma, const float* beta, void* y, int rows, int cols,
                       long ldx, long ldy, float eps, int relu_in, void* stream);

/* LLaMA RMSNorm (HF LlamaRMSNorm, reached from spi_llava.py:198-205). */
int g4r_rmsnorm_bf16(const void* x, const float* gamma, void* y, int rows, int cols, long ldx, long ldy,
                     float eps, void* stream);

/*
 * GroupNorm statistics of ConvModule's GN (layers.py:133-144; mmcv/cnn/bricks/norm.py:101-107)
 * over an NHWC bf16 map, returned as a per-(image, channel) affine  y = a*x + s :
 *   scale_shift [B, 2, C] fp32 (a then s).  `partial` = workspace of B*256*G*2 floats.
 * The normalisation itself (+ReLU) is applied by the consumers (g4r_fuse_shuffle_nhwc_bf16,
 * g4r_roi_align_mlvl_nhwc_*), so the normalised map is never materialised.
 */
int g4r_groupnorm_affine_nhwc_bf16(const void* x, const float* gamma, const float* beta, float* partial,
                                   float* scale_shift, int B, int HW, int C, int G, float eps,
                                   void* stream);

/*
 * MLVLROIQueryModule.forward pyramid build + MLVLFuseModule coordinate concat
 * (gpt4roi/models/layers.py:225-232, 117-127, 183-189): bilinear align_corners=True resize of a
 * ViT level [B, Hin*Win, ldin-strided C] to [B, H, W, Cpad] with channels C, C+1 = x, y
 * coordinates in linspace(-1,1) and zero padding to Cpad (a multiple of 64 for the 1x1-conv GEMM).
 */
int g4r_upsample_coord_nhwc_bf16(const void* in, void* out, int B, int Hin, int Win,
                                 long in_batch_stride, int ldin, int H, int W, int C, int Cpad,
                                 void* stream);

/*
 * MLVLFuseModule._single_shuffle input for one level (layers.py:152-180):
 *   out = cat[ own[:, :C/2], resize(top[:, 3C/4:]), resize(down[:, C/2:3C/4]) ]
 * bilinear align_corners=True in fp32; *_affine (nullable, [B,2,C]) = deferred GN+ReLU of the
 * source (see g4r_groupnorm_affine_nhwc_bf16).
 */
int g4r_fuse_shuffle_nhwc_bf16(const void* own, const float* own_affine, int H, int W, const void* top,
                               const float* top_affine, int Ht, int Wt, const void* down,
                               const float* down_affine, int Hd, int Wd, void* out, int B, int C,
                               void* stream);

/* CLIP patch embedding front end (HF CLIPVisionEmbeddings): image fp32 NCHW [B,3,S,S] ->
 * [B*(S/14)^2, Kpad] bf16 rows (k = c*196 + ky*14 + kx, zero padded), then
 * tokens = cat(cls, patches) + position_embedding. */
int g4r_im2col_patch14_f32(const float* img, void* out, int B, int S, int Kpad, void* stream);
int g4r_vit_assemble_bf16(const void* patch, const void* cls, const void* pos, void* tok, int B, int n,
                          int C, void* stream);

/* LLaMA: rotary embedding of q and k (HF rotate_half convention) + KV-cache append.
 * qkv [T, 3*heads*head_dim]; cos/sin [max_pos, head_dim/2] fp32; caches [max_pos, heads*head_dim]. */
int g4r_rope_qkv_bf16(const void* qkv, const float* cos_tab, const float* sin_tab, void* q_out,
                      void* k_cache, void* v_cache, int T, int heads, int head_dim, int pos0,
                      const int* pos_dev /* nullable: position read from device memory */, void* stream);
/* LLaMA MLP gate: out[T,F] = silu(gate_up[:, :F]) * gate_up[:, F:]. */
int g4r_swiglu_bf16(const void* gate_up, void* out, int T, int F, void* stream);

/*
 * Token embedding + image-patch splice + <bbox> region-token injection in one gather; replaces
 * the per-sample host loop of gpt4roi/models/spi_llava.py:99-196.  status[b] bit flags:
 * 1 patch count != n_patch, 2 #<bbox> != #regions, 4 no <im_start> before the patch run,
 * 8 no <im_end> after it, 16 patch run not contiguous (the reference raises ValueError).
 * spi_offset: [B+1] int32 prefix sums of regions per sample (NULL = no regions).
 */
int g4r_splice_embed_bf16(const long* ids, const void* embed, const void* img, const void* spi,
                          const int* spi_offset, void* out, int* status, int B, int T, int C,
                          int n_patch, long patch_id, long bbox_id, long im_start_id, long im_end_id,
                          int vocab, void* stream);

/* Device-side greedy step: tok = argmax(logits[:N]); out_ids[*step] = tok; ++*step; ++*pos  (nothing
 * returns to the host, so generate(do_sample=False)'s per-token loop -- llava.py:263-283, app.py:294-300
 * with sampling off -- can be replayed from a hipGraph). */
int g4r_greedy_advance_f32(const float* logits, int N, long* tok, long* out_ids, int* step, int* pos,
                           int max_steps, void* stream);

/* Batched greedy step (B equal-length sequences decoded together): tok[b] = tok32[b] = nxt[b] (the per-row argmax from
 * g4r_argmax_rows_f32), out_ids[b][*step] = nxt[b] (row stride max_steps), then ++*step, ++*pos -- all on the device. */
int g4r_batch_advance(const long* nxt, int B, long* tok, int* tok32, long* out_ids, int* step, int* pos, int max_steps,
                      void* stream);
/* Ragged batch: `pos` is an array of npos counters (the B per-sequence cache lengths and the shared RoPE position of
 * g4r_attn_decode_ragged_bf16), every one advanced by 1. */
int g4r_batch_advance_ragged(const long* nxt, int B, long* tok, int* tok32, long* out_ids, int* step, int* pos, int npos,
                             int max_steps, void* stream);
/* greedy decode: out[r] = argmax(logits[r, :N]) (lowest index on ties). */
int g4r_argmax_rows_f32(const float* logits, long ld, int rows, int N, long* out, void* stream);
/* y = a + b[row % brows]  ("fuse_roi_feats + pos_embedd", layers.py:328). */
int g4r_add_rows_bf16(const void* a, const void* b, void* y, long rows, int C, long brows, void* stream);
int g4r_cast_f32_to_bf16(const float* x, void* y, long n, void* stream);
/* torch conv weight [Co][Ci][3][3] fp32 -> the kernels' 16-bit layouts in one pass.  transposed = 0: the forward weight
 * rows of g4r_conv3x3_nhwc_bf16, dst[co*ld + off + tap*Ci + ci] = w[co][ci][tap] (off = g*9*Ci for group g of a grouped
 * conv); transposed = 1: the data-gradient weight, dst[ci*ld + off + tap*Co + co] = w[co][ci][8 - tap] (the 180-degree
 * rotated, channel-transposed filter of conv2d's backward; `dst` = the row block of group g). */
int g4r_conv3x3_weight_layout_bf16(const float* w, void* dst, int Co, int Ci, long ld, long off, int transposed, void* stream);

/*
 * Image front end: uint8 HWC (RGB, or BGR with bgr = 1) -> fp32 CHW [3, out_h, out_w],
 * (pixel/255 - mean) / std, bilinear resize with align_corners = False -- `image_processor.preprocess` followed by
 * F.interpolate of gpt4roi/app.py:125-136; the Resize + Normalize stages of the dataset pipelines
 * (gpt4roi/datasets/refcoco.py:69-85).  row_bytes = bytes between image rows (>= 3 * width).
 */
int g4r_image_preprocess_u8_f32(const void* image, int height, int width, long row_bytes, int bgr, float* out,
                                int out_h, int out_w, float mean_r, float mean_g, float mean_b, float std_r,
                                float std_g, float std_b, void* stream);

/*
 * One sampling step of generate(do_sample=True, temperature, top_k, top_p) with the decode state on the device
 * (gpt4roi/app.py:293-300 calls HF generate with do_sample=True, temperature=0.2; HF's warper order is temperature ->
 * top-k -> top-p, GenerationConfig defaults top_k = 50, top_p = 1.0): keeps the tokens whose logit is >= the top_k-th
 * largest (top_k = 0: all), optionally the top-p nucleus of those, and draws by inverse CDF in ascending vocabulary
 * order with u = Philox4x32-10(counter = (*step, 0, 0, 0), key = *seed)[0] >> 8 scaled to [0, 1).
 * tok[0] = id; out_ids[*step] = id; u_out[*step] = u (nullable); ++*step; ++*pos.  top_k in [0, 1024]; top_p < 1 needs
 * top_k >= 1.  Same device-state contract as g4r_greedy_advance_f32 (hipGraph-replayable; the seed is read from memory).
 */
int g4r_sample_advance_f32(const float* logits, int N, float temperature, int top_k, float top_p,
                           const unsigned long long* seed, long* tok, long* out_ids, int* step, int* pos,
                           int max_steps, float* u_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* G4R_KERNELS_H */
