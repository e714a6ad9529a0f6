/*
 * g4r_roi_align.h -- C ABI of the MI355X (gfx950) RoIAlign kernels.
 *
 * These entry points are what the reference's native boundary for RoIAlign binds:
 *   mmcv._ext.roi_align_forward / roi_align_backward
 *     (/root/reference/mmcv-1.4.7/mmcv/ops/csrc/pytorch/pybind.cpp:191-199, 611-620,
 *      dispatched at csrc/pytorch/roi_align.cpp:5-41, CUDA launchers at
 *      csrc/pytorch/cuda/roi_align_cuda.cu:5-58).
 *
 * Contract (same ownership rules as the reference):
 *   - every pointer is a DEVICE pointer owned by the caller, already allocated;
 *     nothing is allocated, freed or retained here; no global state;
 *   - `stream` is a hipStream_t (NULL = the null stream); launches are asynchronous;
 *   - return value: G4R_OK, or a G4R_ERR_* code (the Python wrapper raises RuntimeError,
 *     as the reference's TORCH_CHECK / AT_CUDA_CHECK do);
 *   - layouts: `input` NCHW contiguous [batch, channels, height, width];
 *     `rois` [n_rois, 5] = (batch_index, x1, y1, x2, y2) in the same dtype as input;
 *     `output` / `argmax_*` [n_rois, channels, pooled_h, pooled_w];
 *     pool_mode 0 = max, 1 = avg (mmcv/ops/roi_align.py:77); argmax_* are only touched
 *     for max; `grad_input` must be zero-filled by the caller (roi_align.py:113).
 *   - a RoI whose batch index is outside [0, batch) produces zeros / no gradient instead
 *     of the reference's out-of-bounds read.
 */
#ifndef G4R_ROI_ALIGN_H
#define G4R_ROI_ALIGN_H

#ifdef __cplusplus
extern "C" {
#endif

#define G4R_OK 0
#define G4R_ERR_INVALID_ARG 1
#define G4R_ERR_LAUNCH 2
#define G4R_ERR_UNSUPPORTED 3

/* Library identification: returns the ABI version (bumped on signature changes). */
int g4r_abi_version(void);
/* Human-readable text of the last HIP error seen by this thread ("" if none). */
const char* g4r_last_error(void);

/* ---- drop-in op: replaces roi_align_forward_impl<CUDA> (roi_align_cuda.cu:5-30) ---- */
int g4r_roi_align_forward_f32(const float* input, const float* rois, float* output,
                              float* argmax_y, float* argmax_x, int batch, int channels,
                              int height, int width, int n_rois, int pooled_h, int pooled_w,
                              float spatial_scale, int sampling_ratio, int pool_mode,
                              int aligned, void* stream);
int g4r_roi_align_forward_f64(const double* input, const double* rois, double* output,
                              double* argmax_y, double* argmax_x, int batch, int channels,
                              int height, int width, int n_rois, int pooled_h, int pooled_w,
                              float spatial_scale, int sampling_ratio, int pool_mode,
                              int aligned, void* stream);
/* fp16 storage (IEEE binary16 bit patterns), fp32 arithmetic. */
int g4r_roi_align_forward_f16(const void* input, const void* rois, void* output,
                              void* argmax_y, void* argmax_x, int batch, int channels,
                              int height, int width, int n_rois, int pooled_h, int pooled_w,
                              float spatial_scale, int sampling_ratio, int pool_mode,
                              int aligned, void* stream);

/* ---- drop-in op: replaces roi_align_backward_impl<CUDA> (roi_align_cuda.cu:32-58) ---- */
int g4r_roi_align_backward_f32(const float* grad_output, const float* rois,
                               const float* argmax_y, const float* argmax_x,
                               float* grad_input, int batch, int channels, int height,
                               int width, int n_rois, int pooled_h, int pooled_w,
                               float spatial_scale, int sampling_ratio, int pool_mode,
                               int aligned, void* stream);
int g4r_roi_align_backward_f64(const double* grad_output, const double* rois,
                               const double* argmax_y, const double* argmax_x,
                               double* grad_input, int batch, int channels, int height,
                               int width, int n_rois, int pooled_h, int pooled_w,
                               float spatial_scale, int sampling_ratio, int pool_mode,
                               int aligned, void* stream);
int g4r_roi_align_backward_f16(const void* grad_output, const void* rois,
                               const void* argmax_y, const void* argmax_x, void* grad_input,
                               int batch, int channels, int height, int width, int n_rois,
                               int pooled_h, int pooled_w, float spatial_scale,
                               int sampling_ratio, int pool_mode, int aligned, void* stream);

/*
 * ---- fused region path: the 4 per-level RoIAlign calls of MlvlRoIExtractor.forward
 *      (/root/reference/gpt4roi/models/layers.py:307-313) in ONE launch ----
 * feats[l]  : device pointer to level l, NHWC [batch, heights[l], widths[l], channels]
 *             (bf16 bit patterns for *_bf16, float for *_f32)
 * rois      : [n_rois, 5] float32 (batch_index, x1, y1, x2, y2), image pixels
 * output    : [levels, n_rois, pooled_h, pooled_w, channels], same dtype as feats.
 *             The reference computes on an fp32 copy of the map and rounds the result
 *             back to the model dtype (layers.py:311-313); so does this kernel
 *             (fp32 taps, fp32 accumulation in the reference's summation order, one
 *             final rounding).
 * avg pooling, sampling_ratio > 0 only (the regime GPT4RoI runs: 14x14, sr 2, aligned).
 * channels must be a multiple of 8.  levels <= 8.
 * affines   : optional HOST array of `levels` device pointers (entries may be NULL): a deferred
 *             GroupNorm+ReLU, [batch, 2, channels] float32 = per-(image, channel) scale `a` then
 *             shift `s`; each texel is read as relu(a*x + s).  This is ConvModule's conv->GN->ReLU
 *             order (mmcv/cnn/bricks/conv_module.py:196-206) folded into the gather, so the last
 *             fuse round's normalised map (layers.py:193-195) is never written to HBM.
 * feats/affines/heights/widths/scales are HOST arrays of length `levels`.
 */
int g4r_roi_align_mlvl_nhwc_bf16(const void* const* feats, const float* const* affines,
                                 const int* heights,
                                 const int* widths, const float* scales, int levels,
                                 const float* rois, void* output, int batch, int channels,
                                 int n_rois, int pooled_h, int pooled_w, int sampling_ratio,
                                 int aligned, void* stream);
int g4r_roi_align_mlvl_nhwc_f32(const void* const* feats, const float* const* affines,
                                const int* heights,
                                const int* widths, const float* scales, int levels,
                                const float* rois, void* output, int batch, int channels,
                                int n_rois, int pooled_h, int pooled_w, int sampling_ratio,
                                int aligned, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* G4R_ROI_ALIGN_H */
