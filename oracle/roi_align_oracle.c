/*
 * oracle/roi_align_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the RoIAlign algorithm the
 * reference runs on its hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (gpt4roi_amd/) never does and fails loudly when the HIP library is absent.
 *
 * What it restates (all paths relative to /root/reference/mmcv-1.4.7/mmcv/ops/csrc):
 *   - forward  : pytorch/cpu/roi_align.cpp:110-214 (ROIAlignForward) with the
 *                tap table of :23-108 (pre_calc_for_bilinear_interpolate)
 *   - backward : pytorch/cpu/roi_align.cpp:270-382 (ROIAlignBackward) with
 *                :216-263 (bilinear_interpolate_gradient)
 *   - the same maths as the CUDA kernels common/cuda/roi_align_cuda_kernel.cuh:17-210
 *     and common/cuda/common_cuda_helper.hpp:28-119.
 *
 * Pinning: tests/test_oracle_roi_align.py checks this file against
 *   (1) the three hand-computed known-answer cases (output AND input-gradient)
 *       of mmcv-1.4.7/tests/test_ops/test_roi_align.py:14-32, and
 *   (2) oracle/_ref (the reference's own CPU sources compiled unmodified) on
 *       seeded inputs, through the fixtures committed under tests/golden/.
 *
 * Layout contract (as the reference op): input NCHW contiguous, rois [n,5] =
 * (batch_idx, x1, y1, x2, y2) in input-image pixels, output [n,C,ph,pw].
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_ERR_NEGATIVE_ROI 1 /* cpu/roi_align.cpp:137-139 AT_ASSERTM */
#define ORACLE_ERR_ALLOC 2
#define ORACLE_ERR_BATCH_INDEX 3

/* One bilinear sample = four (offset, weight) taps into an H x W plane. */
#define DEFINE_ORACLE(T, SUFFIX)                                                         \
  typedef struct {                                                                        \
    int off[4];                                                                           \
    T w[4];                                                                               \
    int valid;                                                                            \
  } tap_##SUFFIX;                                                                         \
                                                                                          \
  /* cpu/roi_align.cpp:42-102 and :216-263: out-of-map test, clamp, corner weights. */    \
  static void make_tap_##SUFFIX(T y, T x, int height, int width, tap_##SUFFIX* t) {       \
    if (y < (T)-1.0 || y > (T)height || x < (T)-1.0 || x > (T)width) {                    \
      t->valid = 0;                                                                       \
      for (int k = 0; k < 4; ++k) {                                                       \
        t->off[k] = 0;                                                                    \
        t->w[k] = (T)0;                                                                   \
      }                                                                                   \
      return;                                                                             \
    }                                                                                     \
    if (y <= (T)0) y = (T)0;                                                              \
    if (x <= (T)0) x = (T)0;                                                              \
    int y0 = (int)y, x0 = (int)x, y1, x1;                                                 \
    if (y0 >= height - 1) {                                                               \
      y1 = y0 = height - 1;                                                               \
      y = (T)y0;                                                                          \
    } else {                                                                              \
      y1 = y0 + 1;                                                                        \
    }                                                                                     \
    if (x0 >= width - 1) {                                                                \
      x1 = x0 = width - 1;                                                                \
      x = (T)x0;                                                                          \
    } else {                                                                              \
      x1 = x0 + 1;                                                                        \
    }                                                                                     \
    T ly = y - (T)y0, lx = x - (T)x0;                                                     \
    T hy = (T)1. - ly, hx = (T)1. - lx;                                                   \
    t->valid = 1;                                                                         \
    t->off[0] = y0 * width + x0;                                                          \
    t->off[1] = y0 * width + x1;                                                          \
    t->off[2] = y1 * width + x0;                                                          \
    t->off[3] = y1 * width + x1;                                                          \
    t->w[0] = hy * hx;                                                                    \
    t->w[1] = hy * lx;                                                                    \
    t->w[2] = ly * hx;                                                                    \
    t->w[3] = ly * lx;                                                                    \
  }                                                                                       \
                                                                                          \
  /* RoI box -> start / bin size / grid, cpu/roi_align.cpp:125-153 (same at :290-309). */  \
  static int roi_geometry_##SUFFIX(const T* roi, T scale, int aligned, int ph, int pw,    \
                                   int sampling_ratio, T* start_h, T* start_w, T* bin_h,  \
                                   T* bin_w, int* grid_h, int* grid_w) {                  \
    T offset = aligned ? (T)0.5 : (T)0.0;                                                 \
    T sw = roi[1] * scale - offset;                                                       \
    T sh = roi[2] * scale - offset;                                                       \
    T ew = roi[3] * scale - offset;                                                       \
    T eh = roi[4] * scale - offset;                                                       \
    T rw = ew - sw, rh = eh - sh;                                                         \
    if (aligned) {                                                                        \
      if (!(rw >= 0 && rh >= 0)) return ORACLE_ERR_NEGATIVE_ROI;                          \
    } else {                                                                              \
      if (rw < (T)1.) rw = (T)1.;                                                         \
      if (rh < (T)1.) rh = (T)1.;                                                         \
    }                                                                                     \
    *start_h = sh;                                                                        \
    *start_w = sw;                                                                        \
    *bin_h = rh / (T)ph;                                                                  \
    *bin_w = rw / (T)pw;                                                                  \
    *grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf((float)(rh / ph));         \
    *grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf((float)(rw / pw));         \
    return ORACLE_OK;                                                                     \
  }                                                                                       \
                                                                                          \
  int oracle_roi_align_forward_##SUFFIX(                                                  \
      const T* input, const T* rois, T* output, T* argmax_y, T* argmax_x, int batch,      \
      int n_rois, int channels, int height, int width, int pooled_h, int pooled_w,        \
      T spatial_scale, int sampling_ratio, int pool_mode, int aligned) {                  \
    const int bins = pooled_h * pooled_w;                                                 \
    for (int n = 0; n < n_rois; ++n) {                                                    \
      const T* roi = rois + 5 * n;                                                        \
      const int b = (int)roi[0];                                                          \
      if (b < 0 || b >= batch) return ORACLE_ERR_BATCH_INDEX;                             \
      T sh, sw, bh, bw;                                                                   \
      int gh, gw;                                                                         \
      int rc = roi_geometry_##SUFFIX(roi, spatial_scale, aligned, pooled_h, pooled_w,     \
                                     sampling_ratio, &sh, &sw, &bh, &bw, &gh, &gw);       \
      if (rc) return rc;                                                                  \
      const int per_bin = gh * gw;                                                        \
      const T count = (T)(per_bin > 1 ? per_bin : 1); /* :152 zero grid -> 0/1 */         \
      const size_t n_taps = (size_t)bins * (size_t)(per_bin > 0 ? per_bin : 0);           \
      tap_##SUFFIX* taps = (tap_##SUFFIX*)malloc((n_taps ? n_taps : 1) * sizeof(*taps));  \
      T* ys = (T*)malloc((n_taps ? n_taps : 1) * sizeof(T));                              \
      T* xs = (T*)malloc((n_taps ? n_taps : 1) * sizeof(T));                              \
      if (!taps || !ys || !xs) {                                                          \
        free(taps);                                                                       \
        free(ys);                                                                         \
        free(xs);                                                                         \
        return ORACLE_ERR_ALLOC;                                                          \
      }                                                                                   \
      size_t k = 0;                                                                       \
      for (int ph = 0; ph < pooled_h; ++ph)                                               \
        for (int pw = 0; pw < pooled_w; ++pw)                                             \
          for (int iy = 0; iy < gh; ++iy) {                                               \
            /* sample coordinate expression kept in the reference's order, :34-41 */      \
            const T y = sh + ph * bh + (T)(iy + .5f) * bh / (T)gh;                        \
            for (int ix = 0; ix < gw; ++ix, ++k) {                                        \
              const T x = sw + pw * bw + (T)(ix + .5f) * bw / (T)gw;                      \
              ys[k] = y;                                                                  \
              xs[k] = x;                                                                  \
              make_tap_##SUFFIX(y, x, height, width, &taps[k]);                           \
            }                                                                             \
          }                                                                               \
      for (int c = 0; c < channels; ++c) {                                                \
        const T* plane = input + ((size_t)b * channels + c) * height * width;             \
        T* out = output + ((size_t)n * channels + c) * bins;                              \
        k = 0;                                                                            \
        for (int bin = 0; bin < bins; ++bin) {                                            \
          T acc = (T)0, best = (T)-10000; /* :176 CPU max-pool seed */                    \
          T by = (T)-1.f, bx = (T)-1.f;                                                   \
          for (int s = 0; s < per_bin; ++s, ++k) {                                        \
            const tap_##SUFFIX* t = &taps[k];                                             \
            const T v = t->w[0] * plane[t->off[0]] + t->w[1] * plane[t->off[1]] +         \
                        t->w[2] * plane[t->off[2]] + t->w[3] * plane[t->off[3]];          \
            if (v > best) {                                                               \
              best = v;                                                                   \
              by = ys[k];                                                                 \
              bx = xs[k];                                                                 \
            }                                                                             \
            acc += v;                                                                     \
          }                                                                               \
          if (pool_mode == 0) {                                                           \
            out[bin] = best;                                                              \
            argmax_y[((size_t)n * channels + c) * bins + bin] = by;                       \
            argmax_x[((size_t)n * channels + c) * bins + bin] = bx;                       \
          } else {                                                                        \
            out[bin] = acc / count;                                                       \
          }                                                                               \
        }                                                                                 \
      }                                                                                   \
      free(taps);                                                                         \
      free(ys);                                                                           \
      free(xs);                                                                           \
    }                                                                                     \
    return ORACLE_OK;                                                                     \
  }                                                                                       \
                                                                                          \
  /* grad_input must be zero-filled by the caller (mmcv/ops/roi_align.py:113).  */        \
  int oracle_roi_align_backward_##SUFFIX(                                                 \
      const T* grad_output, const T* rois, const T* argmax_y, const T* argmax_x,          \
      T* grad_input, int batch, int n_rois, int channels, int height, int width,          \
      int pooled_h, int pooled_w, T spatial_scale, int sampling_ratio, int pool_mode,     \
      int aligned) {                                                                      \
    const int bins = pooled_h * pooled_w;                                                 \
    for (int n = 0; n < n_rois; ++n) {                                                    \
      const T* roi = rois + 5 * n;                                                        \
      const int b = (int)roi[0];                                                          \
      if (b < 0 || b >= batch) return ORACLE_ERR_BATCH_INDEX;                             \
      T sh, sw, bh, bw;                                                                   \
      int gh, gw;                                                                         \
      int rc = roi_geometry_##SUFFIX(roi, spatial_scale, aligned, pooled_h, pooled_w,     \
                                     sampling_ratio, &sh, &sw, &bh, &bw, &gh, &gw);       \
      if (rc) return rc;                                                                  \
      const T count = (T)(gh * gw); /* :333, no max(.,1) in backward */                   \
      for (int c = 0; c < channels; ++c) {                                                \
        T* gplane = grad_input + ((size_t)b * channels + c) * height * width;             \
        const size_t obase = ((size_t)n * channels + c) * bins;                           \
        for (int ph = 0; ph < pooled_h; ++ph)                                             \
          for (int pw = 0; pw < pooled_w; ++pw) {                                         \
            const size_t oi = obase + (size_t)ph * pooled_w + pw;                         \
            const T g = grad_output[oi];                                                  \
            tap_##SUFFIX t;                                                               \
            if (pool_mode == 0) {                                                         \
              const T y = argmax_y[oi], x = argmax_x[oi];                                 \
              if (y != (T)-1.f) {                                                         \
                make_tap_##SUFFIX(y, x, height, width, &t);                               \
                if (t.valid)                                                              \
                  for (int q = 0; q < 4; ++q) gplane[t.off[q]] += g * t.w[q];             \
              }                                                                           \
            } else {                                                                      \
              for (int iy = 0; iy < gh; ++iy) {                                           \
                const T y = sh + ph * bh + (T)(iy + .5f) * bh / (T)gh;                    \
                for (int ix = 0; ix < gw; ++ix) {                                         \
                  const T x = sw + pw * bw + (T)(ix + .5f) * bw / (T)gw;                  \
                  make_tap_##SUFFIX(y, x, height, width, &t);                             \
                  if (t.valid)                                                            \
                    for (int q = 0; q < 4; ++q) gplane[t.off[q]] += g * t.w[q] / count;   \
                }                                                                         \
              }                                                                           \
            }                                                                             \
          }                                                                               \
      }                                                                                   \
    }                                                                                     \
    return ORACLE_OK;                                                                     \
  }

DEFINE_ORACLE(float, f32)
DEFINE_ORACLE(double, f64)
