// roi_align.hip -- RoIAlign for MI355X (gfx950), written from scratch for CDNA4.
//
// Replaces, behind the C ABI of include/g4r_roi_align.h, the reference's
//   roi_align_forward_cuda_kernel / roi_align_backward_cuda_kernel
//   (/root/reference/mmcv-1.4.7/mmcv/ops/csrc/common/cuda/roi_align_cuda_kernel.cuh:17-210,
//    bilinear helpers common/cuda/common_cuda_helper.hpp:28-119).
// Same sampling rules, different machine mapping:
//   * the bilinear taps of a RoI are separable (row taps x column taps) and shared by all
//     channels, so a workgroup first builds the two 1-D tap tables of its RoI in LDS and then
//     only gathers -- the CUDA kernel recomputes all of it per output element;
//   * NCHW drop-in kernel: one workgroup = (RoI, slab of channels); consecutive lanes own
//     consecutive bins of one channel plane => output stores are fully coalesced and the 16
//     gathers of a bin stay inside the RoI's window of one plane (L1/L2 resident);
//   * NHWC multi-level kernel (the fused region path): one wave = one bin, lanes over channels,
//     every corner is one contiguous 16 B/lane (1 KiB/wave) load, 4 levels in one launch, the
//     workgroup -> (level, RoI) map is XCD-aware so a RoI's window stays in one XCD's L2.
// Arithmetic follows the reference's expression order (no FMA contraction in this file), so
// fp32 results agree with the reference CPU implementation to the last bit on most inputs.
#include <float.h>
#include <hip/hip_fp16.h>

#include "g4r_common.h"

#pragma clang fp contract(off)

namespace {

template <typename CT>
struct Tap1D {
  CT coord;  // the un-clamped sample coordinate (what max-pool records as argmax)
  CT frac;   // l = v - low   (h = 1 - l)
  int lo, hi;
  int valid;
};

// roi_align_cuda_kernel.cuh / common_cuda_helper.hpp:33-56, one axis at a time.
template <typename CT>
__device__ __forceinline__ Tap1D<CT> make_tap1d(CT v, int size) {
  Tap1D<CT> t;
  t.coord = v;
  if (v < (CT)-1.0 || v > (CT)size) {
    t.frac = (CT)0;
    t.lo = t.hi = 0;
    t.valid = 0;
    return t;
  }
  if (v <= (CT)0) v = (CT)0;
  int lo = (int)v, hi;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    v = (CT)lo;
  } else {
    hi = lo + 1;
  }
  t.frac = v - (CT)lo;
  t.lo = lo;
  t.hi = hi;
  t.valid = 1;
  return t;
}

template <typename CT>
struct RoiGeom {
  CT start_h, start_w, bin_h, bin_w;
  int grid_h, grid_w;
  int batch;
};

// roi_align_cuda_kernel.cuh:31-62
template <typename CT, typename RT>
__device__ __forceinline__ RoiGeom<CT> roi_geometry(const RT* roi, CT scale, int aligned, int PH,
                                                    int PW, int sr) {
  RoiGeom<CT> g;
  g.batch = (int)(CT)roi[0];
  const CT offset = aligned ? (CT)0.5 : (CT)0.0;
  const CT sw = (CT)roi[1] * scale - offset;
  const CT sh = (CT)roi[2] * scale - offset;
  const CT ew = (CT)roi[3] * scale - offset;
  const CT eh = (CT)roi[4] * scale - offset;
  CT rw = ew - sw, rh = eh - sh;
  if (!aligned) {
    rw = rw > (CT)1. ? rw : (CT)1.;
    rh = rh > (CT)1. ? rh : (CT)1.;
  }
  g.start_h = sh;
  g.start_w = sw;
  g.bin_h = rh / (CT)PH;
  g.bin_w = rw / (CT)PW;
  g.grid_h = sr > 0 ? sr : (int)ceilf((float)(rh / (CT)PH));
  g.grid_w = sr > 0 ? sr : (int)ceilf((float)(rw / (CT)PW));
  return g;
}

template <typename CT>
__device__ __forceinline__ CT sample_coord(CT start, CT bin, int p, int i, int grid) {
  // "roi_start + p * bin + (i + .5f) * bin / grid", the reference's order (cuh:66-72)
  return start + (CT)p * bin + (CT)((float)i + .5f) * bin / (CT)grid;
}

template <typename T> __device__ __forceinline__ float ld(const float* p) { return *p; }
__device__ __forceinline__ float load_as(const float* p, float) { return *p; }
__device__ __forceinline__ double load_as(const double* p, double) { return *p; }
__device__ __forceinline__ float load_as(const __half* p, float) { return __half2float(*p); }
__device__ __forceinline__ void store_as(float* p, float v) { *p = v; }
__device__ __forceinline__ void store_as(double* p, double v) { *p = v; }
__device__ __forceinline__ void store_as(__half* p, float v) { *p = __float2half(v); }

// ---------------------------------------------------------------------------------------------
// NCHW forward (the mmcv drop-in).  grid = (n_rois, ceil(C / cpb)), block = 256.
// ---------------------------------------------------------------------------------------------
template <typename T, typename CT, int MAXTAB>
__global__ __launch_bounds__(256) void roi_align_fwd_nchw_kernel(
    const T* __restrict__ in, const T* __restrict__ rois, T* __restrict__ out,
    T* __restrict__ amy, T* __restrict__ amx, int B, int C, int H, int W, int PH, int PW, CT scale,
    int sr, int pool_mode, int aligned, int cpb) {
  __shared__ Tap1D<CT> ytab[MAXTAB];
  __shared__ Tap1D<CT> xtab[MAXTAB];
  const int n = blockIdx.x;
  const int c0 = blockIdx.y * cpb;
  const int c1 = min(C, c0 + cpb);
  const int tid = threadIdx.x;
  const RoiGeom<CT> g = roi_geometry<CT>(rois + (size_t)5 * n, scale, aligned, PH, PW, sr);
  const int gh = g.grid_h, gw = g.grid_w;
  const int ny = PH * gh, nx = PW * gw;
  const bool use_tab = ny <= MAXTAB && nx <= MAXTAB;  // workgroup-uniform
  if (use_tab) {
    for (int i = tid; i < ny; i += 256)
      ytab[i] = make_tap1d<CT>(sample_coord<CT>(g.start_h, g.bin_h, i / gh, i % gh, gh), H);
    for (int i = tid; i < nx; i += 256)
      xtab[i] = make_tap1d<CT>(sample_coord<CT>(g.start_w, g.bin_w, i / gw, i % gw, gw), W);
    __syncthreads();
  }
  const int bins = PH * PW;
  const int total = (c1 - c0) * bins;
  const bool batch_ok = g.batch >= 0 && g.batch < B;
  const int per_bin = gh * gw;
  const CT count = (CT)(per_bin > 1 ? per_bin : 1);
  for (int idx = tid; idx < total; idx += 256) {
    const int c = c0 + idx / bins;
    const int bin = idx % bins;
    const int ph = bin / PW, pw = bin % PW;
    const size_t oi = ((size_t)n * C + c) * bins + bin;
    if (!batch_ok) {
      store_as(out + oi, (CT)0);
      if (pool_mode == 0) {
        store_as(amy + oi, (CT)-1.f);
        store_as(amx + oi, (CT)-1.f);
      }
      continue;
    }
    const T* plane = in + ((size_t)g.batch * C + c) * H * W;
    CT acc = (CT)0;
    CT best = (CT)-FLT_MAX, by = (CT)-1.f, bx = (CT)-1.f;
    for (int iy = 0; iy < gh; ++iy) {
      const Tap1D<CT> ty = use_tab ? ytab[ph * gh + iy]
                                   : make_tap1d<CT>(sample_coord<CT>(g.start_h, g.bin_h, ph, iy, gh), H);
      for (int ix = 0; ix < gw; ++ix) {
        const Tap1D<CT> tx = use_tab ? xtab[pw * gw + ix]
                                     : make_tap1d<CT>(sample_coord<CT>(g.start_w, g.bin_w, pw, ix, gw), W);
        CT val = (CT)0;
        if (ty.valid && tx.valid) {
          const CT ly = ty.frac, lx = tx.frac;
          const CT hy = (CT)1. - ly, hx = (CT)1. - lx;
          const CT v1 = load_as(plane + ty.lo * W + tx.lo, CT());
          const CT v2 = load_as(plane + ty.lo * W + tx.hi, CT());
          const CT v3 = load_as(plane + ty.hi * W + tx.lo, CT());
          const CT v4 = load_as(plane + ty.hi * W + tx.hi, CT());
          const CT w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
          val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
        }
        if (val > best) {
          best = val;
          by = ty.coord;
          bx = tx.coord;
        }
        acc += val;
      }
    }
    if (pool_mode == 0) {
      store_as(out + oi, best);
      store_as(amy + oi, by);
      store_as(amx + oi, bx);
    } else {
      store_as(out + oi, acc / count);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// NCHW backward: scatter with hardware float atomics (as the reference does, cuh:111-210).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_accum(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_accum(double* p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_accum(__half* p, float v) {
  // 16-bit add through a CAS on the enclosing aligned dword
  unsigned int* word = (unsigned int*)((uintptr_t)p & ~(uintptr_t)3);
  const bool upper = ((uintptr_t)p & 2) != 0;
  unsigned int old = *word, assumed;
  do {
    assumed = old;
    unsigned short h = upper ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xffffu);
    __half hv = __ushort_as_half(h);
    unsigned short nh = __half_as_ushort(__float2half(__half2float(hv) + v));
    unsigned int repl = upper ? ((assumed & 0x0000ffffu) | ((unsigned int)nh << 16))
                              : ((assumed & 0xffff0000u) | nh);
    old = atomicCAS(word, assumed, repl);
  } while (old != assumed);
}

template <typename T, typename CT>
__global__ __launch_bounds__(256) void roi_align_bwd_nchw_kernel(
    const T* __restrict__ gout, const T* __restrict__ rois, const T* __restrict__ amy,
    const T* __restrict__ amx, T* __restrict__ gin, long nthreads, int B, int C, int H, int W,
    int PH, int PW, CT scale, int sr, int pool_mode, int aligned) {
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < nthreads;
       index += (long)blockDim.x * gridDim.x) {
    const int pw = (int)(index % PW);
    const int ph = (int)((index / PW) % PH);
    const int c = (int)((index / PW / PH) % C);
    const int n = (int)(index / PW / PH / C);
    const CT go = load_as(gout + index, CT());
    const RoiGeom<CT> g = roi_geometry<CT>(rois + (size_t)5 * n, scale, aligned, PH, PW, sr);
    if (g.batch < 0 || g.batch >= B) continue;
    T* gplane = gin + ((size_t)g.batch * C + c) * H * W;
    if (pool_mode == 0) {
      const CT y = load_as(amy + index, CT()), x = load_as(amx + index, CT());
      if (y != (CT)-1.f) {
        const Tap1D<CT> ty = make_tap1d<CT>(y, H), tx = make_tap1d<CT>(x, W);
        if (ty.valid && tx.valid) {
          const CT ly = ty.frac, lx = tx.frac, hy = (CT)1. - ly, hx = (CT)1. - lx;
          atomic_accum(gplane + ty.lo * W + tx.lo, go * (hy * hx));
          atomic_accum(gplane + ty.lo * W + tx.hi, go * (hy * lx));
          atomic_accum(gplane + ty.hi * W + tx.lo, go * (ly * hx));
          atomic_accum(gplane + ty.hi * W + tx.hi, go * (ly * lx));
        }
      }
    } else {
      const int gh = g.grid_h, gw = g.grid_w;
      const CT count = (CT)(gh * gw);
      for (int iy = 0; iy < gh; ++iy) {
        const Tap1D<CT> ty = make_tap1d<CT>(sample_coord<CT>(g.start_h, g.bin_h, ph, iy, gh), H);
        for (int ix = 0; ix < gw; ++ix) {
          const Tap1D<CT> tx = make_tap1d<CT>(sample_coord<CT>(g.start_w, g.bin_w, pw, ix, gw), W);
          if (ty.valid && tx.valid) {
            const CT ly = ty.frac, lx = tx.frac, hy = (CT)1. - ly, hx = (CT)1. - lx;
            atomic_accum(gplane + ty.lo * W + tx.lo, go * (hy * hx) / count);
            atomic_accum(gplane + ty.lo * W + tx.hi, go * (hy * lx) / count);
            atomic_accum(gplane + ty.hi * W + tx.lo, go * (ly * hx) / count);
            atomic_accum(gplane + ty.hi * W + tx.hi, go * (ly * lx) / count);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// NHWC multi-level forward (fused region path).  One launch = all levels of
// MlvlRoIExtractor.forward (gpt4roi/models/layers.py:307-313).
// workgroup = (level, roi, bin-row ph); wave = bin; lane = 8 consecutive channels.
// ---------------------------------------------------------------------------------------------
#define G4R_MAX_LEVELS 8
struct MlvlArgs {
  const void* feat[G4R_MAX_LEVELS];
  const float* affine[G4R_MAX_LEVELS];  // deferred GroupNorm+ReLU: [B, 2, C] (a then s) or null
  int H[G4R_MAX_LEVELS];
  int W[G4R_MAX_LEVELS];
  float scale[G4R_MAX_LEVELS];
};

struct Vec8 {
  float v[8];
};
__device__ __forceinline__ Vec8 load8(const h16_t* p) {
  const uint4v r = *reinterpret_cast<const uint4v*>(p);
  Vec8 o;
  o.v[0] = h16lo(r.x); o.v[1] = h16hi(r.x); o.v[2] = h16lo(r.y); o.v[3] = h16hi(r.y);
  o.v[4] = h16lo(r.z); o.v[5] = h16hi(r.z); o.v[6] = h16lo(r.w); o.v[7] = h16hi(r.w);
  return o;
}
__device__ __forceinline__ Vec8 load8(const float* p) {
  const float4v a = *reinterpret_cast<const float4v*>(p);
  const float4v b = *reinterpret_cast<const float4v*>(p + 4);
  Vec8 o;
  o.v[0] = a.x; o.v[1] = a.y; o.v[2] = a.z; o.v[3] = a.w;
  o.v[4] = b.x; o.v[5] = b.y; o.v[6] = b.z; o.v[7] = b.w;
  return o;
}
__device__ __forceinline__ void store8(h16_t* p, const Vec8& a) {
  uint4v r;
  r.x = pack_h16x2(a.v[0], a.v[1]); r.y = pack_h16x2(a.v[2], a.v[3]);
  r.z = pack_h16x2(a.v[4], a.v[5]); r.w = pack_h16x2(a.v[6], a.v[7]);
  *reinterpret_cast<uint4v*>(p) = r;
}
__device__ __forceinline__ void store8(float* p, const Vec8& a) {
  float4v x = {a.v[0], a.v[1], a.v[2], a.v[3]}, y = {a.v[4], a.v[5], a.v[6], a.v[7]};
  *reinterpret_cast<float4v*>(p) = x;
  *reinterpret_cast<float4v*>(p + 4) = y;
}

#define MLVL_MAX_XTAB 256
#define MLVL_MAX_YTAB 16
#define MLVL_STAGE_COLS 32    // LDS-staged path when the bin row touches <= 32 map columns
#define MLVL_STAGE_CH 256     // channels per staging pass (32 cols x 256 ch x 4 B = 32 KB)

// STAGED = true (the bf16 region path): when the RoI is narrow on this level (bin width < ~2 px, so
// neighbouring samples share texels) the workgroup first combines the map rows of its bin row
// vertically into LDS,   S[x][c] = sum_iy ( hy * v[y_lo][x][c] + ly * v[y_hi][x][c] ),
// reading every needed texel of the row pair ONCE with coalesced 16-B loads, and then resolves the
// horizontal taps of all PW bins from LDS.  Wide RoIs (no texel sharing) keep the direct gather.
// The staged path re-associates the 16-term sum (agreement ~1e-7 relative before the bf16
// rounding); STAGED = false (fp32 instantiation) keeps the reference's summation order exactly.
template <typename TI, bool STAGED>
__global__ __launch_bounds__(256) void roi_align_mlvl_nhwc_kernel(MlvlArgs a,
                                                                  const float* __restrict__ rois,
                                                                  TI* __restrict__ out, int L, int B,
                                                                  int C, int N, int PH, int PW, int sr,
                                                                  int aligned) {
  __shared__ Tap1D<float> xtab[MLVL_MAX_XTAB];
  __shared__ Tap1D<float> ytab[MLVL_MAX_YTAB];
  __shared__ __attribute__((aligned(16))) float stage[STAGED ? MLVL_STAGE_COLS * MLVL_STAGE_CH : 4];
  __shared__ int xrange[2];
  // XCD-aware decode: workgroup b runs on XCD b % 8 (observed dispatch); give all PH rows of
  // one (level, roi) group to the same XCD so its window of the map is fetched into one L2.
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int q = bid >> 3;
  const int ph = q % PH;
  const int grp = (q / PH) * 8 + xcd;
  if (grp >= L * N) return;
  const int l = grp / N, n = grp % N;
  const int H = a.H[l], W = a.W[l];
  const TI* feat = reinterpret_cast<const TI*>(a.feat[l]);
  const RoiGeom<float> g = roi_geometry<float>(rois + (size_t)5 * n, a.scale[l], aligned, PH, PW, sr);
  const int tid = threadIdx.x;
  if (tid == 0) {
    xrange[0] = 0x7fffffff;
    xrange[1] = -1;
  }
  if (tid < sr) ytab[tid] = make_tap1d<float>(sample_coord<float>(g.start_h, g.bin_h, ph, tid, sr), H);
  __syncthreads();
  for (int i = tid; i < PW * sr; i += 256) {
    const Tap1D<float> t = make_tap1d<float>(sample_coord<float>(g.start_w, g.bin_w, i / sr, i % sr, sr), W);
    xtab[i] = t;
    if (STAGED && t.valid) {
      atomicMin(&xrange[0], t.lo);
      atomicMax(&xrange[1], t.hi);
    }
  }
  __syncthreads();
  const bool batch_ok = g.batch >= 0 && g.batch < B;
  const int wave = tid >> 6, lane = tid & 63;
  const int nvec = C >> 3;
  const float count = (float)(sr * sr);
  const int bi = batch_ok ? g.batch : 0;
  const TI* img = feat + (size_t)bi * H * W * C;
  const float* aff = a.affine[l] ? a.affine[l] + (size_t)bi * 2 * C : nullptr;
  TI* obase = out + (((size_t)l * N + n) * PH + ph) * PW * C;

  const int x_first = xrange[0];
  const int ncols = xrange[1] - xrange[0] + 1;
  if (STAGED && batch_ok && ncols >= 1 && ncols <= MLVL_STAGE_COLS) {
    const int vl = tid & 31, sub = tid >> 5;  // 32 vector lanes (256 channels) x 8 column/bin lanes
    for (int c0 = 0; c0 < C; c0 += MLVL_STAGE_CH) {
      const int cv = c0 + vl * 8;
      const bool ch_ok = cv < C;
      Vec8 ga, gs;
      if (aff && ch_ok) {
        ga = load8(aff + cv);
        gs = load8(aff + C + cv);
      }
      // phase 1: vertical combine of the <= 2*sr map rows into LDS, one texel vector per (col, lane)
      for (int col = sub; col < ncols; col += 8) {
        Vec8 acc;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc.v[k] = 0.f;
        if (ch_ok) {
          const int x = x_first + col;
          for (int iy = 0; iy < sr; ++iy) {
            const Tap1D<float> ty = ytab[iy];
            if (!ty.valid) continue;
            const float ly = ty.frac, hy = 1.f - ly;
            Vec8 v1 = load8(img + ((size_t)ty.lo * W + x) * C + cv);
            Vec8 v3 = load8(img + ((size_t)ty.hi * W + x) * C + cv);
            if (aff) {
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                v1.v[k] = fmaxf(ga.v[k] * v1.v[k] + gs.v[k], 0.f);
                v3.v[k] = fmaxf(ga.v[k] * v3.v[k] + gs.v[k], 0.f);
              }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc.v[k] += hy * v1.v[k] + ly * v3.v[k];
          }
        }
        float* sp = stage + col * MLVL_STAGE_CH + vl * 8;
        *reinterpret_cast<float4v*>(sp) = float4v{acc.v[0], acc.v[1], acc.v[2], acc.v[3]};
        *reinterpret_cast<float4v*>(sp + 4) = float4v{acc.v[4], acc.v[5], acc.v[6], acc.v[7]};
      }
      __syncthreads();
      // phase 2: horizontal taps of every bin from LDS
      if (ch_ok) {
        for (int pw = sub; pw < PW; pw += 8) {
          Vec8 acc;
#pragma unroll
          for (int k = 0; k < 8; ++k) acc.v[k] = 0.f;
          for (int ix = 0; ix < sr; ++ix) {
            const Tap1D<float> tx = xtab[pw * sr + ix];
            if (!tx.valid) continue;
            const float lx = tx.frac, hx = 1.f - lx;
            const float* s0 = stage + (tx.lo - x_first) * MLVL_STAGE_CH + vl * 8;
            const float* s1 = stage + (tx.hi - x_first) * MLVL_STAGE_CH + vl * 8;
            const float4v a0 = *reinterpret_cast<const float4v*>(s0), a1 = *reinterpret_cast<const float4v*>(s0 + 4);
            const float4v b0 = *reinterpret_cast<const float4v*>(s1), b1 = *reinterpret_cast<const float4v*>(s1 + 4);
            acc.v[0] += hx * a0.x + lx * b0.x; acc.v[1] += hx * a0.y + lx * b0.y;
            acc.v[2] += hx * a0.z + lx * b0.z; acc.v[3] += hx * a0.w + lx * b0.w;
            acc.v[4] += hx * a1.x + lx * b1.x; acc.v[5] += hx * a1.y + lx * b1.y;
            acc.v[6] += hx * a1.z + lx * b1.z; acc.v[7] += hx * a1.w + lx * b1.w;
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) acc.v[k] = acc.v[k] / count;
          store8(obase + (size_t)pw * C + cv, acc);
        }
      }
      __syncthreads();
    }
    return;
  }

  // direct gather (wide RoIs; fp32 instantiation; RoIs with a bad batch index -> zeros)
  for (int pw = wave; pw < PW; pw += 4) {
    TI* orow = obase + (size_t)pw * C;
    for (int v = lane; v < nvec; v += 64) {
      Vec8 acc;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc.v[k] = 0.f;
      Vec8 ga, gs;
      if (aff) {
        ga = load8(aff + v * 8);
        gs = load8(aff + C + v * 8);
      }
      if (batch_ok) {
        for (int iy = 0; iy < sr; ++iy) {
          const Tap1D<float> ty = ytab[iy];
          for (int ix = 0; ix < sr; ++ix) {
            const Tap1D<float> tx = xtab[pw * sr + ix];
            if (ty.valid && tx.valid) {
              const float ly = ty.frac, lx = tx.frac, hy = 1.f - ly, hx = 1.f - lx;
              const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
              Vec8 v1 = load8(img + ((size_t)ty.lo * W + tx.lo) * C + v * 8);
              Vec8 v2 = load8(img + ((size_t)ty.lo * W + tx.hi) * C + v * 8);
              Vec8 v3 = load8(img + ((size_t)ty.hi * W + tx.lo) * C + v * 8);
              Vec8 v4 = load8(img + ((size_t)ty.hi * W + tx.hi) * C + v * 8);
              if (aff) {  // ConvModule's GN + ReLU applied to the raw conv output, per texel
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                  v1.v[k] = fmaxf(ga.v[k] * v1.v[k] + gs.v[k], 0.f);
                  v2.v[k] = fmaxf(ga.v[k] * v2.v[k] + gs.v[k], 0.f);
                  v3.v[k] = fmaxf(ga.v[k] * v3.v[k] + gs.v[k], 0.f);
                  v4.v[k] = fmaxf(ga.v[k] * v4.v[k] + gs.v[k], 0.f);
                }
              }
#pragma unroll
              for (int k = 0; k < 8; ++k)
                acc.v[k] += w1 * v1.v[k] + w2 * v2.v[k] + w3 * v3.v[k] + w4 * v4.v[k];
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc.v[k] = acc.v[k] / count;
      store8(orow + v * 8, acc);
    }
  }
}

template <typename T, typename CT>
int launch_fwd_nchw(const T* in, const T* rois, T* out, T* amy, T* amx, int B, int C, int H, int W,
                    int N, int PH, int PW, float scale, int sr, int pool_mode, int aligned,
                    void* stream) {
  G4R_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && N >= 0 && PH > 0 && PW > 0, "roi_align: bad shape");
  G4R_REQUIRE(pool_mode == 0 || pool_mode == 1, "roi_align: pool_mode must be 0 (max) or 1 (avg)");
  if (N == 0 || C == 0) return G4R_OK;
  G4R_REQUIRE(in && rois && out, "roi_align: null pointer");
  G4R_REQUIRE(pool_mode == 1 || (amy && amx), "roi_align: max pooling needs argmax buffers");
  const int bins = PH * PW;
  int cpb = g4r_ceil_div(4096, bins);
  if (cpb > C) cpb = C;
  const int gy = g4r_ceil_div(C, cpb);
  G4R_REQUIRE(gy <= 65535, "roi_align: too many channel slabs");
  dim3 grid(N, gy);
  hipLaunchKernelGGL((roi_align_fwd_nchw_kernel<T, CT, 512>), grid, dim3(256), 0, (hipStream_t)stream,
                     in, rois, out, amy, amx, B, C, H, W, PH, PW, (CT)scale, sr, pool_mode, aligned, cpb);
  G4R_CHECK_LAUNCH("roi_align_forward");
  return G4R_OK;
}

template <typename T, typename CT>
int launch_bwd_nchw(const T* gout, const T* rois, const T* amy, const T* amx, T* gin, int B, int C,
                    int H, int W, int N, int PH, int PW, float scale, int sr, int pool_mode,
                    int aligned, void* stream) {
  G4R_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && N >= 0 && PH > 0 && PW > 0, "roi_align: bad shape");
  G4R_REQUIRE(pool_mode == 0 || pool_mode == 1, "roi_align: pool_mode must be 0 (max) or 1 (avg)");
  const long nthreads = (long)N * C * PH * PW;
  if (nthreads == 0) return G4R_OK;
  G4R_REQUIRE(gout && rois && gin, "roi_align: null pointer");
  G4R_REQUIRE(pool_mode == 1 || (amy && amx), "roi_align: max pooling needs argmax buffers");
  long blocks = (nthreads + 255) / 256;
  if (blocks > 256L * 32) blocks = 256L * 32;  // grid-stride beyond 8k workgroups
  hipLaunchKernelGGL((roi_align_bwd_nchw_kernel<T, CT>), dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, gout, rois, amy, amx, gin, nthreads, B, C, H, W, PH, PW,
                     (CT)scale, sr, pool_mode, aligned);
  G4R_CHECK_LAUNCH("roi_align_backward");
  return G4R_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward of the fused multi-level NHWC RoIAlign (training row a12 for the region module): scatters
// d(roi_feats) into fp32 NHWC gradient maps of the POST GroupNorm+ReLU features (the deferred affine's own
// backward is g4r_gn_relu_bwd_nhwc_bf16).  Same sample geometry as the forward kernel; 16 corner taps per
// output bin, fp32 atomics as in the reference's roi_align_backward_cuda_kernel
// (mmcv-1.4.7/mmcv/ops/csrc/common/cuda/roi_align_cuda_kernel.cuh:111-210).
// dout element (l, n, ph, pw, c) = dout[l*lvl_stride + ((n*PH + ph)*PW + pw)*pix_stride + c]  (bf16).
// ---------------------------------------------------------------------------------------------
struct MlvlGradArgs {
  float* grad[G4R_MAX_LEVELS];
  int H[G4R_MAX_LEVELS];
  int W[G4R_MAX_LEVELS];
  float scale[G4R_MAX_LEVELS];
};

#define MLVL_BWD_MAX_PW 16

// Narrow RoIs (the bin row touches <= 32 map columns; most RoIs on the coarse levels): the 2x2 samples of
// neighbouring bins hit the same texels, so the block first folds the horizontal taps of its whole bin row into
// LDS,  S[x][c] = sum_{pw,ix} (hx or lx) * d[pw][c] / count,  and then issues ONE atomic per (map row, column,
// channel) instead of one per (bin, sample, corner): 4*ncols instead of 16*PW atomics per channel.
__global__ __launch_bounds__(256) void roi_align_mlvl_nhwc_bwd_kernel(MlvlGradArgs a, const float* __restrict__ rois,
                                                                      const h16_t* __restrict__ dout, long lvl_stride,
                                                                      long pix_stride, int L, int B, int C, int N,
                                                                      int PH, int PW, int sr, int aligned) {
  __shared__ Tap1D<float> xtab[MLVL_MAX_XTAB];
  __shared__ Tap1D<float> ytab[MLVL_MAX_YTAB];
  __shared__ __attribute__((aligned(16))) float stage[MLVL_STAGE_COLS * MLVL_STAGE_CH];
  __shared__ __attribute__((aligned(16))) float dstage[MLVL_BWD_MAX_PW * MLVL_STAGE_CH];
  __shared__ int xrange[2];
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int q = bid >> 3;
  const int ph = q % PH;
  const int grp = (q / PH) * 8 + xcd;
  if (grp >= L * N) return;
  const int l = grp / N, n = grp % N;
  const int H = a.H[l], W = a.W[l];
  const RoiGeom<float> g = roi_geometry<float>(rois + (size_t)5 * n, a.scale[l], aligned, PH, PW, sr);
  const int tid = threadIdx.x;
  if (tid == 0) {
    xrange[0] = 0x7fffffff;
    xrange[1] = -1;
  }
  if (tid < sr) ytab[tid] = make_tap1d<float>(sample_coord<float>(g.start_h, g.bin_h, ph, tid, sr), H);
  __syncthreads();
  for (int i = tid; i < PW * sr; i += 256) {
    const Tap1D<float> t = make_tap1d<float>(sample_coord<float>(g.start_w, g.bin_w, i / sr, i % sr, sr), W);
    xtab[i] = t;
    if (t.valid) {
      atomicMin(&xrange[0], t.lo);
      atomicMax(&xrange[1], t.hi);
    }
  }
  __syncthreads();
  if (g.batch < 0 || g.batch >= B) return;
  const int wave = tid >> 6, lane = tid & 63;
  const int nvec = C >> 3;
  const float inv_count = 1.f / (float)(sr * sr);
  float* gm = a.grad[l] + (size_t)g.batch * H * W * C;
  const h16_t* dbase = dout + (size_t)l * lvl_stride + ((size_t)n * PH + ph) * PW * pix_stride;

  const int x_first = xrange[0];
  const int ncols = xrange[1] - xrange[0] + 1;
  if (ncols >= 1 && ncols <= MLVL_STAGE_COLS && PW <= MLVL_BWD_MAX_PW) {
    const int vl = tid & 31, sub = tid >> 5;  // 32 vector lanes (256 channels) x 8 column / bin lanes
    for (int c0 = 0; c0 < C; c0 += MLVL_STAGE_CH) {
      const int cv = c0 + vl * 8;
      const bool ch_ok = cv < C;
      // phase 0: this bin row's output gradient (already divided by the sample count) into LDS
      for (int pw = sub; pw < PW; pw += 8) {
        Vec8 d;
#pragma unroll
        for (int k = 0; k < 8; ++k) d.v[k] = 0.f;
        if (ch_ok) d = load8(dbase + (size_t)pw * pix_stride + cv);
        float* dp = dstage + pw * MLVL_STAGE_CH + vl * 8;
        *reinterpret_cast<float4v*>(dp) = float4v{d.v[0] * inv_count, d.v[1] * inv_count, d.v[2] * inv_count, d.v[3] * inv_count};
        *reinterpret_cast<float4v*>(dp + 4) = float4v{d.v[4] * inv_count, d.v[5] * inv_count, d.v[6] * inv_count, d.v[7] * inv_count};
      }
      __syncthreads();
      // phase 1: horizontal taps folded per map column
      for (int col = sub; col < ncols; col += 8) {
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int t = 0; t < PW * sr; ++t) {
          const Tap1D<float> tx = xtab[t];
          if (!tx.valid) continue;
          const float lx = tx.frac, hx = 1.f - lx;
          const float w = (tx.lo - x_first == col ? hx : 0.f) + (tx.hi - x_first == col ? lx : 0.f);
          if (w == 0.f) continue;
          const float* dp = dstage + (t / sr) * MLVL_STAGE_CH + vl * 8;
          const float4v d0 = *reinterpret_cast<const float4v*>(dp), d1 = *reinterpret_cast<const float4v*>(dp + 4);
          acc[0] += w * d0.x; acc[1] += w * d0.y; acc[2] += w * d0.z; acc[3] += w * d0.w;
          acc[4] += w * d1.x; acc[5] += w * d1.y; acc[6] += w * d1.z; acc[7] += w * d1.w;
        }
        float* sp = stage + col * MLVL_STAGE_CH + vl * 8;
        *reinterpret_cast<float4v*>(sp) = float4v{acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<float4v*>(sp + 4) = float4v{acc[4], acc[5], acc[6], acc[7]};
      }
      __syncthreads();
      // phase 2: vertical taps, one atomic per (map row, column, channel)
      if (ch_ok) {
        for (int col = sub; col < ncols; col += 8) {
          const float* sp = stage + col * MLVL_STAGE_CH + vl * 8;
          const float4v s0 = *reinterpret_cast<const float4v*>(sp), s1 = *reinterpret_cast<const float4v*>(sp + 4);
          const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
          for (int iy = 0; iy < sr; ++iy) {
            const Tap1D<float> ty = ytab[iy];
            if (!ty.valid) continue;
            const float ly = ty.frac, hy = 1.f - ly;
            float* p1 = gm + ((size_t)ty.lo * W + x_first + col) * C + cv;
            float* p3 = gm + ((size_t)ty.hi * W + x_first + col) * C + cv;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              unsafeAtomicAdd(p1 + k, hy * sv[k]);
              unsafeAtomicAdd(p3 + k, ly * sv[k]);
            }
          }
        }
      }
      __syncthreads();
    }
    return;
  }

  for (int pw = wave; pw < PW; pw += 4) {
    for (int v = lane; v < nvec; v += 64) {
      Vec8 d = load8(dbase + (size_t)pw * pix_stride + v * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) d.v[k] *= inv_count;
      for (int iy = 0; iy < sr; ++iy) {
        const Tap1D<float> ty = ytab[iy];
        if (!ty.valid) continue;
        for (int ix = 0; ix < sr; ++ix) {
          const Tap1D<float> tx = xtab[pw * sr + ix];
          if (!tx.valid) continue;
          const float ly = ty.frac, lx = tx.frac, hy = 1.f - ly, hx = 1.f - lx;
          float* p1 = gm + ((size_t)ty.lo * W + tx.lo) * C + v * 8;
          float* p2 = gm + ((size_t)ty.lo * W + tx.hi) * C + v * 8;
          float* p3 = gm + ((size_t)ty.hi * W + tx.lo) * C + v * 8;
          float* p4 = gm + ((size_t)ty.hi * W + tx.hi) * C + v * 8;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            unsafeAtomicAdd(p1 + k, d.v[k] * hy * hx);
            unsafeAtomicAdd(p2 + k, d.v[k] * hy * lx);
            unsafeAtomicAdd(p3 + k, d.v[k] * ly * hx);
            unsafeAtomicAdd(p4 + k, d.v[k] * ly * lx);
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Backward of the fused multi-level NHWC RoIAlign WITHOUT atomics (round 2): the transpose of the forward written as
// a GATHER per output texel row.  workgroup = (level, image, map row y, tile of 48 columns, chunk of 256 channels);
// thread = one channel.  The workgroup walks the RoIs of its image; a RoI touches row y through the sample rows whose
// y_low or y_high equals y (weight hy / ly, the separable taps of roi_align_cuda_kernel.cuh:141-148), and each such
// bin row folds its PW x sr horizontal samples into the LDS row buffer S[x][c] -- every S[.][c] is only ever touched
// by the thread that owns channel c, so there is no race and no atomic.  After the last RoI the row tile is written
// once: every texel of every gradient map is stored exactly once (texels under no RoI get their 0 here: the maps need
// no zero-fill), the summation order is fixed (bit-reproducible, unlike the reference's atomicAdd order), and HBM sees
// one coalesced fp32 write per texel instead of a read-modify-write per corner tap.
// ---------------------------------------------------------------------------------------------
#define MLVL_GATHER_TW 48
#define MLVL_GATHER_CH 256

__global__ __launch_bounds__(256) void roi_align_mlvl_nhwc_bwd_gather_kernel(
    MlvlGradArgs a, const float* __restrict__ rois, const int* __restrict__ roi_offsets, const h16_t* __restrict__ dout,
    long lvl_stride, long pix_stride, int L, int B, int C, int N, int PH, int PW, int sr, int aligned) {
  __shared__ float S[MLVL_GATHER_TW * MLVL_GATHER_CH];
  __shared__ Tap1D<float> xtab[MLVL_MAX_XTAB];
  __shared__ float wrow[MLVL_MAX_XTAB];     // per bin row ph: summed vertical weight of its samples onto map row y
  __shared__ int any_row;
  const int tid = threadIdx.x;
  // block -> (level, image, row, column tile)
  int r = blockIdx.x, l = 0, tiles = 1;
  for (; l < L; ++l) {
    tiles = (a.W[l] + MLVL_GATHER_TW - 1) / MLVL_GATHER_TW;
    const int nb = B * a.H[l] * tiles;
    if (r < nb) break;
    r -= nb;
  }
  if (l >= L) return;
  const int H = a.H[l], W = a.W[l];
  const int xt = r % tiles, y = (r / tiles) % H, b = r / (tiles * H);
  const int x0 = xt * MLVL_GATHER_TW;
  const int tw = min(MLVL_GATHER_TW, W - x0);
  const int c = blockIdx.y * MLVL_GATHER_CH + tid;
  const bool ch_ok = c < C;
  for (int i = tid; i < MLVL_GATHER_TW * MLVL_GATHER_CH; i += 256) S[i] = 0.f;
  const float inv_count = 1.f / (float)(sr * sr);
  const int n0 = roi_offsets ? roi_offsets[b] : 0, n1 = roi_offsets ? roi_offsets[b + 1] : N;
  for (int n = n0; n < n1; ++n) {
    const RoiGeom<float> g = roi_geometry<float>(rois + (size_t)5 * n, a.scale[l], aligned, PH, PW, sr);
    if (g.batch != b) continue;                       // uniform across the workgroup
    __syncthreads();                                  // the previous RoI's tables are no longer read
    if (tid == 0) any_row = 0;
    __syncthreads();
    if (tid < PH) {
      float w = 0.f;
      for (int iy = 0; iy < sr; ++iy) {
        const Tap1D<float> ty = make_tap1d<float>(sample_coord<float>(g.start_h, g.bin_h, tid, iy, sr), H);
        if (!ty.valid) continue;
        const float ly = ty.frac, hy = 1.f - ly;
        if (ty.lo == y) w += hy;
        if (ty.hi == y) w += ly;
      }
      wrow[tid] = w;
      if (w != 0.f) any_row = 1;
    }
    for (int i = tid - 64; i >= 0 && i < PW * sr; i += 192)
      xtab[i] = make_tap1d<float>(sample_coord<float>(g.start_w, g.bin_w, i / sr, i % sr, sr), W);
    __syncthreads();
    if (!any_row || !ch_ok) continue;
    const h16_t* dn = dout + (size_t)l * lvl_stride + (size_t)n * PH * PW * pix_stride + c;
    for (int ph = 0; ph < PH; ++ph) {
      const float wy = wrow[ph];
      if (wy == 0.f) continue;
      const float wq = wy * inv_count;
      for (int pw = 0; pw < PW; ++pw) {
        float d = 0.f;
        bool loaded = false;
        for (int ix = 0; ix < sr; ++ix) {
          const Tap1D<float> tx = xtab[pw * sr + ix];
          if (!tx.valid) continue;
          const int xl = tx.lo - x0, xh = tx.hi - x0;
          const bool in_l = xl >= 0 && xl < tw, in_h = xh >= 0 && xh < tw;
          if (!in_l && !in_h) continue;
          if (!loaded) {
            d = h16_to_f32(dn[((size_t)ph * PW + pw) * pix_stride]) * wq;
            loaded = true;
          }
          const float lx = tx.frac, hx = 1.f - lx;
          if (in_l) S[xl * MLVL_GATHER_CH + tid] += hx * d;
          if (in_h) S[xh * MLVL_GATHER_CH + tid] += lx * d;
        }
      }
    }
  }
  __syncthreads();
  if (!ch_ok) return;
  float* gm = a.grad[l] + (((size_t)b * H + y) * W + x0) * C + c;
  for (int x = 0; x < tw; ++x) gm[(size_t)x * C] = S[x * MLVL_GATHER_CH + tid];
}

template <typename TI>
int launch_mlvl(const void* const* feats, const float* const* affines, const int* heights, const int* widths, const float* scales,
                int levels, const float* rois, void* output, int B, int C, int N, int PH, int PW,
                int sr, int aligned, void* stream) {
  G4R_REQUIRE(levels > 0 && levels <= G4R_MAX_LEVELS, "roi_align_mlvl: 1..8 levels");
  G4R_REQUIRE(C > 0 && (C % 8) == 0, "roi_align_mlvl: channels must be a multiple of 8");
  G4R_REQUIRE(PH > 0 && PW > 0 && B > 0 && N >= 0, "roi_align_mlvl: bad shape");
  if (sr <= 0 || sr > MLVL_MAX_YTAB || PW * sr > MLVL_MAX_XTAB)
    return g4r_note_error(G4R_ERR_UNSUPPORTED, "roi_align_mlvl: needs 0 < sampling_ratio <= 16, PW*sr <= 256");
  if (N == 0) return G4R_OK;
  G4R_REQUIRE(feats && heights && widths && scales && rois && output, "roi_align_mlvl: null pointer");
  MlvlArgs a;
  for (int l = 0; l < G4R_MAX_LEVELS; ++l) {
    const int s = l < levels ? l : 0;
    a.feat[l] = feats[s];
    a.affine[l] = affines ? affines[s] : nullptr;
    a.H[l] = heights[s];
    a.W[l] = widths[s];
    a.scale[l] = scales[s];
    G4R_REQUIRE(feats[s] && heights[s] > 0 && widths[s] > 0, "roi_align_mlvl: bad level");
  }
  const long groups = (long)levels * N;
  const long blocks = ((groups + 7) / 8) * 8 * PH;
  G4R_REQUIRE(blocks < 2147483647L, "roi_align_mlvl: grid too large");
  hipLaunchKernelGGL((roi_align_mlvl_nhwc_kernel<TI, sizeof(TI) == 2>), dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, a, rois, (TI*)output, levels, B, C, N, PH, PW, sr, aligned);
  G4R_CHECK_LAUNCH("roi_align_mlvl_nhwc");
  return G4R_OK;
}

}  // namespace

extern "C" {

#ifndef G4R_F16   // the drop-in NCHW op has its own dtype instantiations (f32 / f64 / f16)
int g4r_roi_align_forward_f32(const float* input, const float* rois, float* output, float* argmax_y,
                              float* argmax_x, int batch, int channels, int height, int width,
                              int n_rois, int pooled_h, int pooled_w, float spatial_scale,
                              int sampling_ratio, int pool_mode, int aligned, void* stream) {
  return launch_fwd_nchw<float, float>(input, rois, output, argmax_y, argmax_x, batch, channels, height,
                                       width, n_rois, pooled_h, pooled_w, spatial_scale, sampling_ratio,
                                       pool_mode, aligned, stream);
}
int g4r_roi_align_forward_f64(const double* input, const double* rois, double* output,
                              double* argmax_y, double* argmax_x, int batch, int channels, int height,
                              int width, int n_rois, int pooled_h, int pooled_w, float spatial_scale,
                              int sampling_ratio, int pool_mode, int aligned, void* stream) {
  return launch_fwd_nchw<double, double>(input, rois, output, argmax_y, argmax_x, batch, channels,
                                         height, width, n_rois, pooled_h, pooled_w, spatial_scale,
                                         sampling_ratio, pool_mode, aligned, stream);
}
int g4r_roi_align_forward_f16(const void* input, const void* rois, void* output, void* argmax_y,
                              void* argmax_x, int batch, int channels, int height, int width,
                              int n_rois, int pooled_h, int pooled_w, float spatial_scale,
                              int sampling_ratio, int pool_mode, int aligned, void* stream) {
  return launch_fwd_nchw<__half, float>((const __half*)input, (const __half*)rois, (__half*)output,
                                        (__half*)argmax_y, (__half*)argmax_x, batch, channels, height,
                                        width, n_rois, pooled_h, pooled_w, spatial_scale, sampling_ratio,
                                        pool_mode, aligned, stream);
}

int g4r_roi_align_backward_f32(const float* grad_output, const float* rois, const float* argmax_y,
                               const float* argmax_x, float* grad_input, int batch, int channels,
                               int height, int width, int n_rois, int pooled_h, int pooled_w,
                               float spatial_scale, int sampling_ratio, int pool_mode, int aligned,
                               void* stream) {
  return launch_bwd_nchw<float, float>(grad_output, rois, argmax_y, argmax_x, grad_input, batch, channels,
                                       height, width, n_rois, pooled_h, pooled_w, spatial_scale,
                                       sampling_ratio, pool_mode, aligned, stream);
}
int g4r_roi_align_backward_f64(const double* grad_output, const double* rois, const double* argmax_y,
                               const double* argmax_x, double* grad_input, int batch, int channels,
                               int height, int width, int n_rois, int pooled_h, int pooled_w,
                               float spatial_scale, int sampling_ratio, int pool_mode, int aligned,
                               void* stream) {
  return launch_bwd_nchw<double, double>(grad_output, rois, argmax_y, argmax_x, grad_input, batch,
                                         channels, height, width, n_rois, pooled_h, pooled_w,
                                         spatial_scale, sampling_ratio, pool_mode, aligned, stream);
}
int g4r_roi_align_backward_f16(const void* grad_output, const void* rois, const void* argmax_y,
                               const void* argmax_x, void* grad_input, int batch, int channels,
                               int height, int width, int n_rois, int pooled_h, int pooled_w,
                               float spatial_scale, int sampling_ratio, int pool_mode, int aligned,
                               void* stream) {
  return launch_bwd_nchw<__half, float>((const __half*)grad_output, (const __half*)rois,
                                        (const __half*)argmax_y, (const __half*)argmax_x,
                                        (__half*)grad_input, batch, channels, height, width, n_rois,
                                        pooled_h, pooled_w, spatial_scale, sampling_ratio, pool_mode,
                                        aligned, stream);
}

#endif  // !G4R_F16

int g4r_roi_align_mlvl_nhwc_bf16(const void* const* feats, const float* const* affines, const int* heights, const int* widths,
                                 const float* scales, int levels, const float* rois, void* output,
                                 int batch, int channels, int n_rois, int pooled_h, int pooled_w,
                                 int sampling_ratio, int aligned, void* stream) {
  return launch_mlvl<h16_t>(feats, affines, heights, widths, scales, levels, rois, output, batch, channels,
                             n_rois, pooled_h, pooled_w, sampling_ratio, aligned, stream);
}
#ifndef G4R_F16   // fp32 maps and the training-only backward: one instantiation
int g4r_roi_align_mlvl_nhwc_f32(const void* const* feats, const float* const* affines, const int* heights, const int* widths,
                                const float* scales, int levels, const float* rois, void* output,
                                int batch, int channels, int n_rois, int pooled_h, int pooled_w,
                                int sampling_ratio, int aligned, void* stream) {
  return launch_mlvl<float>(feats, affines, heights, widths, scales, levels, rois, output, batch, channels,
                            n_rois, pooled_h, pooled_w, sampling_ratio, aligned, stream);
}

int g4r_roi_align_mlvl_nhwc_bwd_bf16(const void* dout, long lvl_stride, long pix_stride, float* const* grads,
                                     const int* heights, const int* widths, const float* scales, int levels,
                                     const float* rois, int batch, int channels, int n_rois, int pooled_h,
                                     int pooled_w, int sampling_ratio, int aligned, void* stream) {
  G4R_REQUIRE(levels > 0 && levels <= G4R_MAX_LEVELS, "roi_align_mlvl_bwd: 1..8 levels");
  G4R_REQUIRE(channels > 0 && (channels % 8) == 0 && pix_stride % 8 == 0 && lvl_stride % 8 == 0,
              "roi_align_mlvl_bwd: channels / strides must be multiples of 8");
  G4R_REQUIRE(pooled_h > 0 && pooled_w > 0 && batch > 0 && n_rois >= 0, "roi_align_mlvl_bwd: bad shape");
  if (sampling_ratio <= 0 || sampling_ratio > MLVL_MAX_YTAB || pooled_w * sampling_ratio > MLVL_MAX_XTAB)
    return g4r_note_error(G4R_ERR_UNSUPPORTED, "roi_align_mlvl_bwd: needs 0 < sampling_ratio <= 16, PW*sr <= 256");
  if (n_rois == 0) return G4R_OK;
  G4R_REQUIRE(dout && grads && heights && widths && scales && rois, "roi_align_mlvl_bwd: null pointer");
  MlvlGradArgs a;
  for (int l = 0; l < G4R_MAX_LEVELS; ++l) {
    const int s = l < levels ? l : 0;
    a.grad[l] = grads[s];
    a.H[l] = heights[s];
    a.W[l] = widths[s];
    a.scale[l] = scales[s];
    G4R_REQUIRE(grads[s] && heights[s] > 0 && widths[s] > 0, "roi_align_mlvl_bwd: bad level");
  }
  const long groups = (long)levels * n_rois;
  const long blocks = ((groups + 7) / 8) * 8 * pooled_h;
  G4R_REQUIRE(blocks < 2147483647L, "roi_align_mlvl_bwd: grid too large");
  hipLaunchKernelGGL(roi_align_mlvl_nhwc_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, rois,
                     (const h16_t*)dout, lvl_stride, pix_stride, levels, batch, channels, n_rois, pooled_h, pooled_w,
                     sampling_ratio, aligned);
  G4R_CHECK_LAUNCH("roi_align_mlvl_nhwc_bwd");
  return G4R_OK;
}

int g4r_roi_align_mlvl_nhwc_bwd_gather_bf16(const void* dout, long lvl_stride, long pix_stride, float* const* grads,
                                            const int* heights, const int* widths, const float* scales, int levels,
                                            const float* rois, const int* roi_offsets, int batch, int channels,
                                            int n_rois, int pooled_h, int pooled_w, int sampling_ratio, int aligned,
                                            void* stream) {
  G4R_REQUIRE(levels > 0 && levels <= G4R_MAX_LEVELS, "roi_align_mlvl_bwd_gather: 1..8 levels");
  G4R_REQUIRE(channels > 0 && (channels % 8) == 0, "roi_align_mlvl_bwd_gather: channels must be a multiple of 8");
  G4R_REQUIRE(pooled_h > 0 && pooled_w > 0 && batch > 0 && n_rois >= 0, "roi_align_mlvl_bwd_gather: bad shape");
  if (sampling_ratio <= 0 || sampling_ratio > MLVL_MAX_YTAB || pooled_w * sampling_ratio > MLVL_MAX_XTAB ||
      pooled_h > MLVL_MAX_XTAB || pooled_h > 256)
    return g4r_note_error(G4R_ERR_UNSUPPORTED, "roi_align_mlvl_bwd_gather: needs 0 < sampling_ratio <= 16, PW*sr <= 256, PH <= 256");
  G4R_REQUIRE(grads && heights && widths && scales && (n_rois == 0 || (dout && rois)), "roi_align_mlvl_bwd_gather: null pointer");
  MlvlGradArgs a;
  long blocks = 0;
  for (int l = 0; l < G4R_MAX_LEVELS; ++l) {
    const int s = l < levels ? l : 0;
    a.grad[l] = grads[s];
    a.H[l] = heights[s];
    a.W[l] = widths[s];
    a.scale[l] = scales[s];
    G4R_REQUIRE(grads[s] && heights[s] > 0 && widths[s] > 0, "roi_align_mlvl_bwd_gather: bad level");
    if (l < levels) blocks += (long)batch * heights[s] * ((widths[s] + MLVL_GATHER_TW - 1) / MLVL_GATHER_TW);
  }
  G4R_REQUIRE(blocks < 2147483647L, "roi_align_mlvl_bwd_gather: grid too large");
  hipLaunchKernelGGL(roi_align_mlvl_nhwc_bwd_gather_kernel, dim3((unsigned)blocks, (channels + MLVL_GATHER_CH - 1) / MLVL_GATHER_CH),
                     dim3(256), 0, (hipStream_t)stream, a, rois, roi_offsets, (const h16_t*)dout, lvl_stride, pix_stride,
                     levels, batch, channels, n_rois, pooled_h, pooled_w, sampling_ratio, aligned);
  G4R_CHECK_LAUNCH("roi_align_mlvl_nhwc_bwd_gather");
  return G4R_OK;
}

#endif  // !G4R_F16

}  // extern "C"
