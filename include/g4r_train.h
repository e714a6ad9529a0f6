/*
 * g4r_train.h -- C ABI of the training rows of the region-feature path (SURVEY.md 8a rows a2, a12, a16, a18;
 * 8d configs 3 and 4): the backward and optimizer kernels the reference obtains from torch autograd,
 * flash-attn's backward and torch.optim.AdamW under HF Trainer
 * (/root/reference/gpt4roi/train/train.py:698-712, train_stage1.sh, train_stage2.sh).
 *
 * Same conventions as g4r_kernels.h: raw device pointers + sizes + hipStream_t (as void*), int status
 * (0 = ok; text via g4r_last_error()), the caller owns every buffer, no allocation, no global state.
 * bf16 tensors are raw 16-bit patterns.  Matrix gradients use the forward GEMM / implicit-GEMM entry points of
 * g4r_kernels.h on transposed operands (g4r_transpose_bf16 below).
 */
#ifndef G4R_TRAIN_H
#define G4R_TRAIN_H

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Backward of g4r_flash_attn_fwd_bf16 (which now also returns `lse`, the log2-domain log-sum-exp of the scaled
 * scores, [B, H, Tq] fp32).  Replaces autograd through HF LlamaAttention / flash_attn_unpadded_qkvpacked_func's
 * backward (llava/train/llama_flash_attn_monkey_patch.py:15-91).  delta [B, H, Tq] fp32 is scratch
 * (rowsum(dO * O)).  dQ/dK/dV are fully written (no accumulation, no atomics: bit-reproducible).
 */
int g4r_flash_attn_bwd_bf16(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                            const float* lse, float* delta, void* dQ, void* dK, void* dV, int B, int H, int Tq,
                            int Tk, int head_dim, long q_row, long k_row, long v_row, long o_row, long do_row,
                            long dq_row, long dk_row, long dv_row, long q_batch, long k_batch, long v_batch,
                            long o_batch, long do_batch, long dq_batch, long dk_batch, long dv_batch,
                            float scale, int causal, void* stream);

/* HF LlamaRMSNorm backward.  dx = dres + d(rmsnorm)/dx (dres nullable: the residual-stream gradient that
 * bypasses the norm); dgamma (nullable, fp32 [cols]) is accumulated with atomics. */
int g4r_rmsnorm_bwd_bf16(const void* x, const float* gamma, const void* dy, const void* dres, void* dx,
                         float* dgamma, int rows, int cols, long ldx, long lddy, long lddres, long lddx, float eps,
                         void* stream);

/* nn.LayerNorm backward (pos_embedd, gpt4roi/models/layers.py:260-267); relu_in as in the forward kernel.
 * dx nullable (first layer); dgamma / dbeta fp32 [cols], accumulated. */
int g4r_layernorm_bwd_bf16(const void* x, const float* gamma, const void* dy, void* dx, float* dgamma, float* dbeta,
                           int rows, int cols, long ldx, long lddy, long lddx, float eps, int relu_in, void* stream);

/* SiLU(gate) * up over INTERLEAVED columns (gate_up [T, 2F]: column 2c = gate_c, 2c+1 = up_c, the layout the
 * fused gate|up GEMM writes); the training forward keeps gate_up for the backward. */
int g4r_swiglu_il_bf16(const void* gate_up, void* out, int T, int F, void* stream);
int g4r_swiglu_il_bwd_bf16(const void* gate_up, const void* dy, void* dgate_up, int T, int F, void* stream);

/* Backward of g4r_rope_qkv_bf16: d(qkv)[t] = [R(pos)^-1 dq[t], R(pos)^-1 dk[t], dv[t]], dqkv [T, 3*H*D]. */
int g4r_rope_qkv_bwd_bf16(const void* dq, const void* dk, const void* dv, const float* cos_tab, const float* sin_tab,
                          void* dqkv, int T, int heads, int head_dim, int pos0, long ldq, long ldk, long ldv,
                          void* stream);
/* The same for the stacked rows of a whole batch in one launch: rows = B * period, row r at position pos0 + r % period. */
int g4r_rope_qkv_bwd_batch_bf16(const void* dq, const void* dk, const void* dv, const float* cos_tab, const float* sin_tab,
                                void* dqkv, int rows, int period, int heads, int head_dim, int pos0, long ldq, long ldk,
                                long ldv, void* stream);

/*
 * Token cross entropy of llava/model/llava.py:240-252 (labels already shifted by the caller; label < 0 =
 * ignore_index).  logits fp32 [rows, ld]; *loss_sum += sum over valid rows of (lse - logit[label]);
 * dlogits (nullable) bf16 [rows, ldd] = (softmax - onehot) * *grad_scale, columns N..n_pad zero-filled so the
 * buffer can feed a GEMM whose reduction length must be a multiple of 64.
 */
int g4r_cross_entropy_f32(const float* logits, const long* labels, void* dlogits, float* loss_sum,
                          const float* grad_scale, int rows, int N, long ld, long ldd, int n_pad, void* stream);

/* out[c][r] = in[r][c] for r < R, zero for R <= r < R_pad (bf16).  Weight gradients are NT GEMMs over the token
 * / pixel axis: dW[N,K] = dY^T[N, M] . (X^T[K, M])^T. */
int g4r_transpose_bf16(const void* in, void* out, int R, int C, long ld_in, long ld_out, int R_pad, void* stream);

/* out[c] += sum_r x[r][c]  (bias gradients). */
int g4r_colsum_bf16(const void* x, float* out, int M, int N, long ld, void* stream);

/* dx = dy * (y > 0). */
int g4r_relu_bwd_bf16(const void* y, const void* dy, void* dx, long n, void* stream);

/* dst[i] = src[idx[i]] (rows of C bf16 values; idx < 0 -> zeros): gradient of the image-patch / <bbox> splice
 * (gpt4roi/models/spi_llava.py:99-196) with respect to the projector output and the region embeddings. */
int g4r_gather_rows_bf16(const void* src, const int* idx, void* dst, int n, int C, long ld_src, long ld_dst,
                         void* stream);

/* out[idx[r]][:] += src[r][:] (src bf16 rows, out fp32, idx < 0 = skip; fp32 atomics): gradient of the token
 * embedding lookup inside the splice (rows that took an embedding, spi_llava.py:99-196 / HF embed_tokens). */
int g4r_scatter_add_rows_f32(const void* src, const int* idx, float* out, int n, int C, long ld_src, long ld_out,
                             void* stream);

/* torch.optim.AdamW step on fp32 master weights (decoupled weight decay); grad bf16 or fp32, multiplied by
 * grad_scale first (gradient averaging / clipping); param_bf16 (nullable) receives the rounded copy the kernels
 * read.  step >= 1. */
int g4r_adamw_f32(float* param, const void* grad, int grad_is_bf16, float* exp_avg, float* exp_avg_sq,
                  void* param_bf16, long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int step, float grad_scale, void* stream);

/* ---- region module (gpt4roi/models/layers.py:96-335) ------------------------------------------------------ */

/* Per-(image, group) mean and rstd of a raw NHWC conv output: stats [B, G, 2] fp32.  The forward
 * (g4r_groupnorm_affine_nhwc_bf16) keeps only the folded affine, the backward recomputes the statistics.
 * partial: scratch, B*256*G*2 floats. */
int g4r_groupnorm_stats_nhwc_bf16(const void* x, float* partial, float* stats, int B, int HW, int C, int G, float eps,
                                  void* stream);

/* Backward of ConvModule's GroupNorm + ReLU (mmcv conv_module.py:196-206) applied to the raw conv output z:
 * y = relu(a*z + s).  dy fp32 [B, HW, C] = gradient w.r.t. y; affine [B, 2, C] from the forward; stats from
 * g4r_groupnorm_stats_nhwc_bf16; dgamma/dbeta fp32 [C] are accumulated; gsum scratch [B, G, 2]; dz bf16. */
int g4r_gn_relu_bwd_nhwc_bf16(const void* z, const float* dy, const float* affine, const float* gamma,
                              const float* stats, float* dgamma, float* dbeta, float* gsum, void* dz, int B, int HW,
                              int C, int G, void* stream);

/* Transpose of g4r_fuse_shuffle_nhwc_bf16 (MLVLFuseModule._single_shuffle, layers.py:152-180): the gradient of
 * one level's conv input is scattered (fp32 atomics, bilinear align_corners weights) into the gradient maps of
 * the level itself (channels [0, C/2)), its coarser neighbour (channels [3C/4, C)) and its finer neighbour
 * (channels [C/2, 3C/4)). */
int g4r_fuse_shuffle_bwd_nhwc_bf16(const void* dinp, int H, int W, float* d_own, float* d_top, int Ht, int Wt,
                                   float* d_down, int Hd, int Wd, int B, int C, void* stream);

/* The same transpose as a gather, one call per SOURCE level (what the region module's backward uses): d_src fp32
 * [B, H, W, C] is fully written from the conv-input gradients of the level itself (dinp_own), of the finer target
 * that read this level as its coarser neighbour (dinp_fine, nullable) and of the coarser target that read it as its
 * finer neighbour (dinp_coarse, nullable); self_top / self_down: the level is its own neighbour (last / first
 * level, layers.py:108-112).  No atomics, bit-reproducible. */
int g4r_fuse_shuffle_bwd_gather_nhwc_bf16(float* d_src, const void* dinp_own, int H, int W, const void* dinp_fine,
                                          int Hf, int Wf, const void* dinp_coarse, int Hc, int Wc, int self_top,
                                          int self_down, int B, int C, void* stream);

/* 3x3 weight gradient read straight from NHWC operands (round 4; csrc/gemm_tn.hip).  The reference gets these from
 * torch autograd for the convs of gpt4roi/models/layers.py:129-144,191-195,321-325.
 * g4r_nhwc_pad_bf16: src [B][H][W][C] -> the zero-bordered grid [B][H+2][W+2][C] starting at row `row0` of dst (interior
 * only; border and guard rows stay as the caller zeroed them).
 * g4r_conv3x3_wgrad_nhwc_bf16: dw [Cout][Cin][3][3] fp32, summed over the n_levels (1..4) map geometries that share the
 * weight (the levels of a fuse round; 1 for a plain conv).  Level l: dy_pads[l] [krows][Cout] (row = bordered pixel index,
 * krows = B (H_l+2) (W_l+2) rounded up to 32, zero border) and x_pads[l] [guard + krows + guard][Cin] (guard = W_l + 3
 * zero rows in front of bordered pixel 0 and behind the last row).  The pixel axis of every level is cut into slices of
 * about slice_tiles K tiles of 32 pixels; partials = fp32 workspace [slices][9][Cout][Cin] with slices =
 * g4r_conv3x3_wgrad_nhwc_slices(same arguments) (< 0: shape not supported).  Cout, Cin multiples of 256;
 * accumulate != 0: dw += . */
int g4r_nhwc_pad_bf16(const void* src, void* dst, int B, int H, int W, int C, long row0, void* stream);
/* C [M][N] fp32 (+)= A^T B, A [K][lda] (element (k, m)), B [K][ldb] (element (k, n)) bf16: torch autograd's grad_weight of
 * an nn.Linear, dW [N_out][K_in] = dY^T X, read from the row-major operands (no transposed copies; csrc/gemm_tn.hip).
 * M, N, lda, ldb multiples of 8; K any.  slices = 1, accumulate = 0: direct store (ldc = row stride of C).  Otherwise the K
 * axis is cut into `slices` ranges with fp32 partials [slices][M][N] and a reduce writes / adds to a dense C (ldc == N). */
int g4r_gemm_tn_bf16(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, float* partials,
                     int slices, int accumulate, void* stream);
int g4r_conv3x3_wgrad_nhwc_slices(int n_levels, const int* heights, const int* widths, int B, int Cin, int Cout,
                                  int slice_tiles);
int g4r_conv3x3_wgrad_nhwc_bf16(const void* const* dy_pads, const void* const* x_pads, int n_levels, const int* heights,
                                const int* widths, int B, int Cin, int Cout, int slice_tiles, float* partials, float* dw,
                                int accumulate, void* stream);

/* NHWC bf16 [B, H, W, C] -> channel-major rows with a zero border:
 *   dst[s][c][base + b*seg + (y+1)*Wp + (x+1) - (s - n_shift/2)] = src[b][y][x][c],   dst [n_shift, C, ltot]
 * The 3x3 weight gradient is then 9 NT GEMMs over the pixel axis,
 *   dW[co][ci][ky][kx] = sum_p dY^T[co][p] * X^T_{kx}[ci][p + (ky-1)*Wp],
 * with every operand row 16-byte aligned (n_shift = 3 for X, 1 for dY).  dst must be zero-initialised once. */
int g4r_nhwc_to_cm_padded_bf16(const void* src, void* dst, int B, int H, int W, int C, int Wp, long seg, long base,
                               long ltot, int n_shift, void* stream);

/* Backward of g4r_roi_align_mlvl_nhwc_bf16 (g4r_roi_align.h): dout bf16 with element (l, n, ph, pw, c) at
 * dout[l*lvl_stride + ((n*PH + ph)*PW + pw)*pix_stride + c]; grads[l] fp32 NHWC [B, H_l, W_l, C] accumulate the
 * gradient w.r.t. the (post GroupNorm+ReLU) feature maps.  Replaces roi_align_backward
 * (mmcv-1.4.7/mmcv/ops/csrc/common/cuda/roi_align_cuda_kernel.cuh:111-210) for the fused module. */
int g4r_roi_align_mlvl_nhwc_bwd_bf16(const void* dout, long lvl_stride, long pix_stride, float* const* grads,
                                     const int* heights, const int* widths, const float* scales, int levels,
                                     const float* rois, int batch, int channels, int n_rois, int pooled_h,
                                     int pooled_w, int sampling_ratio, int aligned, void* stream);

/*
 * The same gradient WITHOUT atomics: gather per (level, image, map row, 48-column tile, 256-channel chunk) over the RoIs
 * of that image (g4r_roi_align_mlvl_nhwc_bwd_bf16 above keeps the reference's atomicAdd scatter,
 * roi_align_cuda_kernel.cuh:141-148,197-204).  Every element of every gradient map is WRITTEN exactly once -- the maps
 * need no zero-fill -- in a fixed summation order (bit-reproducible).  roi_offsets (nullable): int32 [batch + 1], the
 * RoIs of image b are rows roi_offsets[b] .. roi_offsets[b+1]-1 (RoIs grouped by image, as layers.py:295-302 builds
 * them); NULL = scan all n_rois and test the batch index.
 */
int g4r_roi_align_mlvl_nhwc_bwd_gather_bf16(const void* dout, long lvl_stride, long pix_stride, float* const* grads,
                                            const int* heights, const int* widths, const float* scales, int levels,
                                            const float* rois, const int* roi_offsets, int batch, int channels,
                                            int n_rois, int pooled_h, int pooled_w, int sampling_ratio, int aligned,
                                            void* stream);

/*
 * Multi-tensor clip + AdamW: the foreach that `torch.nn.utils.clip_grad_norm_` + `torch.optim.AdamW.step()` run under HF
 * Trainer (gpt4roi/train/train.py:698-712), as two launches over a device-resident table of n_tensors tensors.
 * Table arrays (device memory): *_ptrs = uint64 device addresses; numel [n]; g_is_bf16 [n] (gradient dtype per tensor);
 * chunk_start [n + 1] = prefix sums of ceil(numel / 4096); n_chunks = chunk_start[n].
 *   g4r_multi_sumsq     : total[0] = sum over every tensor of sum(grad^2) (fp64, fixed reduction order); `partial` is a
 *                         workspace of n_chunks doubles.
 *   g4r_multi_adamw_f32 : grad' = grad * pre_scale * min(1, max_norm / (pre_scale * sqrt(total_sq[0]) + 1e-6)) (max_norm <= 0
 *                         or total_sq NULL: no clipping), then the AdamW update of g4r_adamw_f32 on p / m / v; pb_ptrs[i] != 0
 *                         receives the bf16 copy the kernels read.  The clip coefficient is computed on the device: no
 *                         host synchronisation between backward and update.
 */
int g4r_multi_sumsq(const void* g_ptrs, const long* numel, const int* g_is_bf16, const int* chunk_start, int n_tensors,
                    int n_chunks, double* partial, double* total, void* stream);
int g4r_multi_adamw_f32(const void* p_ptrs, const void* g_ptrs, const void* m_ptrs, const void* v_ptrs, const void* pb_ptrs,
                        const long* numel, const int* g_is_bf16, const int* chunk_start, int n_tensors, int n_chunks,
                        const double* total_sq, float max_norm, float pre_scale, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* G4R_TRAIN_H */
