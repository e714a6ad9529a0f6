// norm.hip -- row normalisations of the region-feature path (HBM-bound, one pass each).
//   LayerNorm  : CLIP ViT pre_layrnorm / layer_norm1 / layer_norm2 (HF CLIPEncoderLayer, the
//                arithmetic spi_llava.py:66-67 delegates to) and pos_embedd's LayerNorms
//                (gpt4roi/models/layers.py:260-267)
//   RMSNorm    : LLaMA input/post-attention/final norms (spi_llava.py:198-205 -> HF LlamaRMSNorm)
//   GroupNorm  : statistics of ConvModule's GN(64 groups, eps 1e-5)
//                (gpt4roi/models/layers.py:133-144; mmcv/cnn/bricks/norm.py:101-107) over NHWC.
// bf16 storage, fp32 arithmetic, 16-byte vector accesses, wave64 shuffles for reductions.
#include "g4r_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

constexpr int NORM_MAXV = 4;  // 16-B vectors per thread: rows up to 256*4*8 = 8192 elements

// One workgroup per row; the row lives in registers (single HBM read).  cols % 8 == 0.
//   y = (x - mean) * rstd * gamma + beta   (biased variance, eps inside the sqrt)
__global__ __launch_bounds__(256) void layernorm_bf16_kernel(const h16_t* __restrict__ x,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             h16_t* __restrict__ y, int rows, int cols,
                                                             long ldx, long ldy, float eps, int relu_in) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const h16_t* xr = x + (size_t)row * ldx;
  const int nvec = cols >> 3;
  float f[NORM_MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      const uint4v r = *reinterpret_cast<const uint4v*>(xr + v * 8);
      f[i][0] = h16lo(r.x); f[i][1] = h16hi(r.x); f[i][2] = h16lo(r.y); f[i][3] = h16hi(r.y);
      f[i][4] = h16lo(r.z); f[i][5] = h16hi(r.z); f[i][6] = h16lo(r.w); f[i][7] = h16hi(r.w);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (relu_in) f[i][k] = fmaxf(f[i][k], 0.f);
        s += f[i][k];
      }
    }
  }
  const float mean = block_sum(s, red) / (float)cols;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i)
    if (threadIdx.x + i * 256 < nvec) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float d = f[i][k] - mean;
        s2 += d * d;
      }
    }
  const float rstd = rsqrtf(block_sum(s2, red) / (float)cols + eps);
  h16_t* yr = y + (size_t)row * ldy;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      const float4v g0 = *reinterpret_cast<const float4v*>(gamma + v * 8);
      const float4v g1 = *reinterpret_cast<const float4v*>(gamma + v * 8 + 4);
      const float4v b0 = *reinterpret_cast<const float4v*>(beta + v * 8);
      const float4v b1 = *reinterpret_cast<const float4v*>(beta + v * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (f[i][k] - mean) * rstd * g[k] + b[k];
      uint4v w;
      w.x = pack_h16x2(o[0], o[1]); w.y = pack_h16x2(o[2], o[3]);
      w.z = pack_h16x2(o[4], o[5]); w.w = pack_h16x2(o[6], o[7]);
      *reinterpret_cast<uint4v*>(yr + v * 8) = w;
    }
  }
}

// One workgroup per row.  y = x * rsqrt(mean(x^2) + eps) * gamma   (HF LlamaRMSNorm: the
// normalised value is rounded to the storage dtype BEFORE the gamma multiply).
__global__ __launch_bounds__(256) void rmsnorm_bf16_kernel(const h16_t* __restrict__ x,
                                                           const float* __restrict__ gamma,
                                                           h16_t* __restrict__ y, int rows, int cols,
                                                           long ldx, long ldy, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const h16_t* xr = x + (size_t)row * ldx;
  const int nvec = cols >> 3;
  float f[NORM_MAXV][8];
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      const uint4v r = *reinterpret_cast<const uint4v*>(xr + v * 8);
      f[i][0] = h16lo(r.x); f[i][1] = h16hi(r.x); f[i][2] = h16lo(r.y); f[i][3] = h16hi(r.y);
      f[i][4] = h16lo(r.z); f[i][5] = h16hi(r.z); f[i][6] = h16lo(r.w); f[i][7] = h16hi(r.w);
#pragma unroll
      for (int k = 0; k < 8; ++k) s2 += f[i][k] * f[i][k];
    }
  }
  const float rstd = rsqrtf(block_sum(s2, red) / (float)cols + eps);
  h16_t* yr = y + (size_t)row * ldy;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      const float4v g0 = *reinterpret_cast<const float4v*>(gamma + v * 8);
      const float4v g1 = *reinterpret_cast<const float4v*>(gamma + v * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      float o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = h16_to_f32(f32_to_h16(f[i][k] * rstd)) * g[k];
      uint4v w;
      w.x = pack_h16x2(o[0], o[1]); w.y = pack_h16x2(o[2], o[3]);
      w.z = pack_h16x2(o[4], o[5]); w.w = pack_h16x2(o[6], o[7]);
      *reinterpret_cast<uint4v*>(yr + v * 8) = w;
    }
  }
}

// RMSNorm of a row that is still a sum of split-K partials: x = sum_s partial[s][row] (+ residual), rounded to bf16 and
// stored (the residual stream), then y = rmsnorm(x) -- splitk_reduce_kernel and rmsnorm_bf16_kernel in one pass over the
// row, same summation order (s = 0, 1, ...), same rounding points, same element -> thread map: bit-identical to the two
// launches (one launch and one 6 MB round trip less per LLaMA layer).
__global__ __launch_bounds__(256) void rmsnorm_splitk_bf16_kernel(const float* __restrict__ partials, int splits,
                                                                  long slice_stride, const h16_t* __restrict__ residual,
                                                                  long ldr, h16_t* __restrict__ xout, long ldxo,
                                                                  const float* __restrict__ gamma, h16_t* __restrict__ y,
                                                                  long ldy, int cols, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const int nvec = cols >> 3;
  float f[NORM_MAXV][8];
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      float acc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = 0.f;
      const float* src = partials + (size_t)row * cols + v * 8;
      for (int s = 0; s < splits; ++s) {
        const float4v a = *reinterpret_cast<const float4v*>(src + (size_t)s * slice_stride);
        const float4v b = *reinterpret_cast<const float4v*>(src + (size_t)s * slice_stride + 4);
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
      }
      if (residual) {
        const uint4v r = *reinterpret_cast<const uint4v*>(residual + (size_t)row * ldr + v * 8);
        acc[0] += h16lo(r.x); acc[1] += h16hi(r.x); acc[2] += h16lo(r.y); acc[3] += h16hi(r.y);
        acc[4] += h16lo(r.z); acc[5] += h16hi(r.z); acc[6] += h16lo(r.w); acc[7] += h16hi(r.w);
      }
      uint4v w;
      w.x = pack_h16x2(acc[0], acc[1]); w.y = pack_h16x2(acc[2], acc[3]);
      w.z = pack_h16x2(acc[4], acc[5]); w.w = pack_h16x2(acc[6], acc[7]);
      *reinterpret_cast<uint4v*>(xout + (size_t)row * ldxo + v * 8) = w;
      f[i][0] = h16lo(w.x); f[i][1] = h16hi(w.x); f[i][2] = h16lo(w.y); f[i][3] = h16hi(w.y);
      f[i][4] = h16lo(w.z); f[i][5] = h16hi(w.z); f[i][6] = h16lo(w.w); f[i][7] = h16hi(w.w);
#pragma unroll
      for (int k = 0; k < 8; ++k) s2 += f[i][k] * f[i][k];
    }
  }
  const float rstd = rsqrtf(block_sum(s2, red) / (float)cols + eps);
  h16_t* yr = y + (size_t)row * ldy;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      const float4v g0 = *reinterpret_cast<const float4v*>(gamma + v * 8);
      const float4v g1 = *reinterpret_cast<const float4v*>(gamma + v * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      float o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = h16_to_f32(f32_to_h16(f[i][k] * rstd)) * g[k];
      uint4v w;
      w.x = pack_h16x2(o[0], o[1]); w.y = pack_h16x2(o[2], o[3]);
      w.z = pack_h16x2(o[4], o[5]); w.w = pack_h16x2(o[6], o[7]);
      *reinterpret_cast<uint4v*>(yr + v * 8) = w;
    }
  }
}

// LayerNorm of a row that is still a sum of split-K partials: x = sum_s partial[s][row] + bias (+ residual), rounded to bf16
// and stored (the residual stream), then y = layernorm(x) -- splitk_reduce_kernel and layernorm_bf16_kernel in one pass,
// same summation order and rounding points: bit-identical to the two launches (CLIP fc2 -> next block's layer_norm1).
__global__ __launch_bounds__(256) void layernorm_splitk_bf16_kernel(const float* __restrict__ partials, int splits,
                                                                    long slice_stride, const float* __restrict__ bias,
                                                                    const h16_t* __restrict__ residual, long ldr,
                                                                    h16_t* __restrict__ xout, long ldxo,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, h16_t* __restrict__ y,
                                                                    long ldy, int cols, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const int nvec = cols >> 3;
  float f[NORM_MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      float acc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = 0.f;
      const float* src = partials + (size_t)row * cols + v * 8;
      for (int sl = 0; sl < splits; ++sl) {
        const float4v a = *reinterpret_cast<const float4v*>(src + (size_t)sl * slice_stride);
        const float4v b = *reinterpret_cast<const float4v*>(src + (size_t)sl * slice_stride + 4);
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
      }
      if (bias) {
        const float4v b0 = *reinterpret_cast<const float4v*>(bias + v * 8), b1 = *reinterpret_cast<const float4v*>(bias + v * 8 + 4);
        acc[0] += b0.x; acc[1] += b0.y; acc[2] += b0.z; acc[3] += b0.w;
        acc[4] += b1.x; acc[5] += b1.y; acc[6] += b1.z; acc[7] += b1.w;
      }
      if (residual) {
        const uint4v r = *reinterpret_cast<const uint4v*>(residual + (size_t)row * ldr + v * 8);
        acc[0] += h16lo(r.x); acc[1] += h16hi(r.x); acc[2] += h16lo(r.y); acc[3] += h16hi(r.y);
        acc[4] += h16lo(r.z); acc[5] += h16hi(r.z); acc[6] += h16lo(r.w); acc[7] += h16hi(r.w);
      }
      uint4v w;
      w.x = pack_h16x2(acc[0], acc[1]); w.y = pack_h16x2(acc[2], acc[3]);
      w.z = pack_h16x2(acc[4], acc[5]); w.w = pack_h16x2(acc[6], acc[7]);
      *reinterpret_cast<uint4v*>(xout + (size_t)row * ldxo + v * 8) = w;
      f[i][0] = h16lo(w.x); f[i][1] = h16hi(w.x); f[i][2] = h16lo(w.y); f[i][3] = h16hi(w.y);
      f[i][4] = h16lo(w.z); f[i][5] = h16hi(w.z); f[i][6] = h16lo(w.w); f[i][7] = h16hi(w.w);
#pragma unroll
      for (int k = 0; k < 8; ++k) s += f[i][k];
    }
  }
  const float mean = block_sum(s, red) / (float)cols;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i)
    if (threadIdx.x + i * 256 < nvec) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float d = f[i][k] - mean;
        s2 += d * d;
      }
    }
  const float rstd = rsqrtf(block_sum(s2, red) / (float)cols + eps);
  h16_t* yr = y + (size_t)row * ldy;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      const float4v g0 = *reinterpret_cast<const float4v*>(gamma + v * 8);
      const float4v g1 = *reinterpret_cast<const float4v*>(gamma + v * 8 + 4);
      const float4v b0 = *reinterpret_cast<const float4v*>(beta + v * 8);
      const float4v b1 = *reinterpret_cast<const float4v*>(beta + v * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (f[i][k] - mean) * rstd * g[k] + b[k];
      uint4v w;
      w.x = pack_h16x2(o[0], o[1]); w.y = pack_h16x2(o[2], o[3]);
      w.z = pack_h16x2(o[4], o[5]); w.w = pack_h16x2(o[6], o[7]);
      *reinterpret_cast<uint4v*>(yr + v * 8) = w;
    }
  }
}

// GroupNorm statistics over NHWC bf16, two launches, no atomics:
//   (1) partial[b][chunk][g] = (sum, sumsq) over a chunk of pixels   grid = (chunks, B)
//   (2) reduce the chunks in fp64 and emit the per-(b, channel) affine y = a*x + s
// block = 256 threads; thread = one 8-channel vector, strided pixels.
__global__ __launch_bounds__(256) void gn_stats_nhwc_kernel(const h16_t* __restrict__ x,
                                                            float* __restrict__ partial, int HW, int C,
                                                            int G, int pix_per_block) {
  __shared__ float red[2][256];
  const int b = blockIdx.y;
  const int nvec = C >> 3;             // vectors per pixel (divides 256)
  const int tid = threadIdx.x;
  const int cv = tid % nvec;
  const int pl = tid / nvec;           // pixel lane
  const int plc = 256 / nvec;          // pixel lanes per block (>= 1)
  const int p0 = blockIdx.x * pix_per_block;
  int p1 = p0 + pix_per_block;
  if (p1 > HW) p1 = HW;
  float s = 0.f, s2 = 0.f;
  const h16_t* base = x + ((size_t)b * HW) * C + cv * 8;
  for (int p = p0 + pl; p < p1; p += plc) {
    const uint4v r = *reinterpret_cast<const uint4v*>(base + (size_t)p * C);
    const float f[8] = {h16lo(r.x), h16hi(r.x), h16lo(r.y), h16hi(r.y),
                        h16lo(r.z), h16hi(r.z), h16lo(r.w), h16hi(r.w)};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s += f[k];
      s2 += f[k] * f[k];
    }
  }
  red[0][tid] = s;
  red[1][tid] = s2;
  __syncthreads();
  // group g owns vectors [g*vpg, (g+1)*vpg), vpg = (C/G)/8
  const int vpg = (C / G) >> 3;
  if (tid < G) {
    float ts = 0.f, ts2 = 0.f;
    for (int l = 0; l < plc; ++l)
      for (int v = 0; v < vpg; ++v) {
        const int t = l * nvec + tid * vpg + v;
        ts += red[0][t];
        ts2 += red[1][t];
      }
    float* dst = partial + (((size_t)b * gridDim.x + blockIdx.x) * G + tid) * 2;
    dst[0] = ts;
    dst[1] = ts2;
  }
}

// one wave per (image, group): shuffle-reduce the chunk partials in fp64, then the wave writes the
// affine of its C/G channels.  grid = ceil(B*G / 4), block = 256.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ scale_shift, int B, int C,
                                                          int G, int chunks, double count, float eps) {
  const int bg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bg >= B * G) return;
  const int lane = threadIdx.x & 63;
  const int b = bg / G, g = bg % G;
  double sum = 0.0, sq = 0.0;
  for (int k = lane; k < chunks; k += 64) {
    const float* src = partial + (((size_t)b * chunks + k) * G + g) * 2;
    sum += (double)src[0];
    sq += (double)src[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o);
    sq += __shfl_xor(sq, o);
  }
  const double mean = sum / count;
  double var = sq / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const int cpg = C / G;
  for (int i = lane; i < cpg; i += 64) {
    const int c = g * cpg + i;
    const float a = gamma[c] * rstd;
    scale_shift[(size_t)b * 2 * C + c] = a;
    scale_shift[(size_t)b * 2 * C + C + c] = beta[c] - (float)mean * a;
  }
}

// ---- the same two steps for EVERY level of a pyramid stacked in one buffer ([level][b][y][x][C]: kernels.MlvlMaps) ----
#define G4R_GN_MAX_LEVELS 4
struct GnLevels {
  long pix0[G4R_GN_MAX_LEVELS];     // first pixel row of level l in the stacked buffer
  int hw[G4R_GN_MAX_LEVELS];        // pixels per image
  int ppb[G4R_GN_MAX_LEVELS];       // pixels per chunk
  int chunks[G4R_GN_MAX_LEVELS];    // chunks per image
  int blk_end[G4R_GN_MAX_LEVELS];   // running workgroup count: level l owns [blk_end[l-1], blk_end[l]) = B * chunks[l] of them
  int n;
};
// partial layout: [level][b][256 chunk slots][G][2] floats; statistics identical to gn_stats_nhwc_kernel per (level, b)
__global__ __launch_bounds__(256) void gn_stats_mlvl_kernel(const h16_t* __restrict__ x, float* __restrict__ partial,
                                                            GnLevels a, int B, int C, int G) {
  __shared__ float red[2][256];
  int l = 0;
#pragma unroll
  for (int q = 0; q < G4R_GN_MAX_LEVELS - 1; ++q)
    if (q + 1 < a.n && (int)blockIdx.x >= a.blk_end[q]) l = q + 1;
  const int local = (int)blockIdx.x - (l == 0 ? 0 : a.blk_end[l - 1]);
  const int HW = a.hw[l], chunks = a.chunks[l], ppb = a.ppb[l];
  const int b = local / chunks, chunk = local - b * chunks;
  const int nvec = C >> 3;
  const int tid = threadIdx.x;
  const int cv = tid % nvec;
  const int pl = tid / nvec;
  const int plc = 256 / nvec;
  const int p0 = chunk * ppb;
  int p1 = p0 + ppb;
  if (p1 > HW) p1 = HW;
  float s = 0.f, s2 = 0.f;
  const h16_t* base = x + ((size_t)a.pix0[l] + (size_t)b * HW) * C + cv * 8;
  // four 16-byte loads in flight per thread (the per-level kernel walks one pixel at a time: 2.6 TB/s); the accumulation
  // order per thread is unchanged, so the sums are the same bits
  for (int p = p0 + pl; p < p1; p += 4 * plc) {
    uint4v r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pp = p + u * plc;
      r[u] = *reinterpret_cast<const uint4v*>(base + (size_t)(pp < p1 ? pp : p) * C);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (p + u * plc < p1) {
        const float f[8] = {h16lo(r[u].x), h16hi(r[u].x), h16lo(r[u].y), h16hi(r[u].y),
                            h16lo(r[u].z), h16hi(r[u].z), h16lo(r[u].w), h16hi(r[u].w)};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s += f[k];
          s2 += f[k] * f[k];
        }
      }
    }
  }
  red[0][tid] = s;
  red[1][tid] = s2;
  __syncthreads();
  const int vpg = (C / G) >> 3;
  if (tid < G) {
    float ts = 0.f, ts2 = 0.f;
    for (int q = 0; q < plc; ++q)
      for (int v = 0; v < vpg; ++v) {
        const int t = q * nvec + tid * vpg + v;
        ts += red[0][t];
        ts2 += red[1][t];
      }
    float* dst = partial + ((((size_t)l * B + b) * 256 + chunk) * G + tid) * 2;
    dst[0] = ts;
    dst[1] = ts2;
  }
}

// one wave per (level, image, group); scale_shift [level][B][2][C]
__global__ __launch_bounds__(256) void gn_finalize_mlvl_kernel(const float* __restrict__ partial,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               float* __restrict__ scale_shift, GnLevels a, int B,
                                                               int C, int G, float eps) {
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= a.n * B * G) return;
  const int lane = threadIdx.x & 63;
  const int g = idx % G, b = (idx / G) % B, l = idx / (G * B);
  const int chunks = a.chunks[l];
  double sum = 0.0, sq = 0.0;
  for (int k = lane; k < chunks; k += 64) {
    const float* src = partial + ((((size_t)l * B + b) * 256 + k) * G + g) * 2;
    sum += (double)src[0];
    sq += (double)src[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o);
    sq += __shfl_xor(sq, o);
  }
  const double count = (double)a.hw[l] * (double)(C / G);
  const double mean = sum / count;
  double var = sq / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const int cpg = C / G;
  float* ss = scale_shift + ((size_t)l * B + b) * 2 * C;
  for (int i = lane; i < cpg; i += 64) {
    const int c = g * cpg + i;
    const float av = gamma[c] * rstd;
    ss[c] = av;
    ss[C + c] = beta[c] - (float)mean * av;
  }
}

}  // namespace

extern "C" {

/* GroupNorm affines of every level of a stacked pyramid (kernels.MlvlMaps: [level][b][y][x][C] bf16) in two launches.
 * partial: n_levels * B * 256 * G * 2 floats; scale_shift: [n_levels][B][2][C] floats.  Same arithmetic (chunking,
 * fp32 chunk sums, fp64 chunk reduction) as g4r_groupnorm_affine_nhwc_bf16 per level: bit-identical affines. */
int g4r_groupnorm_affine_mlvl_nhwc_bf16(const void* x, const float* gamma, const float* beta, float* partial,
                                        float* scale_shift, int n_levels, const int* level_hw, int B, int C, int G,
                                        float eps, void* stream) {
  G4R_REQUIRE(n_levels >= 1 && n_levels <= G4R_GN_MAX_LEVELS && B > 0 && C > 0 && G > 0 && C % G == 0 &&
                  (C / G) % 8 == 0, "groupnorm_mlvl: 1..4 levels, channels per group a multiple of 8");
  const int nvec = C / 8;
  G4R_REQUIRE(nvec <= 256 && 256 % nvec == 0 && G <= 256, "groupnorm_mlvl: C/8 must divide 256");
  G4R_REQUIRE(x && gamma && beta && partial && scale_shift && level_hw, "groupnorm_mlvl: null pointer");
  GnLevels a;
  long pix = 0;
  int blocks = 0;
  for (int l = 0; l < G4R_GN_MAX_LEVELS; ++l) {
    const int HW = level_hw[l < n_levels ? l : 0];
    G4R_REQUIRE(HW > 0, "groupnorm_mlvl: bad level size");
    int ppb = g4r_ceil_div(HW, 256);
    if (ppb < 32) ppb = 32;
    a.pix0[l] = pix;
    a.hw[l] = HW;
    a.ppb[l] = ppb;
    a.chunks[l] = g4r_ceil_div(HW, ppb);
    if (l < n_levels) {
      pix += (long)B * HW;
      blocks += B * a.chunks[l];
    }
    a.blk_end[l] = blocks;
  }
  a.n = n_levels;
  hipLaunchKernelGGL(gn_stats_mlvl_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, partial, a,
                     B, C, G);
  G4R_CHECK_LAUNCH("gn_stats_mlvl");
  hipLaunchKernelGGL(gn_finalize_mlvl_kernel, dim3(g4r_ceil_div((long)n_levels * B * G, 4)), dim3(256), 0,
                     (hipStream_t)stream, partial, gamma, beta, scale_shift, a, B, C, G, eps);
  G4R_CHECK_LAUNCH("gn_finalize_mlvl");
  return G4R_OK;
}

int g4r_layernorm_bf16(const void* x, const float* gamma, const float* beta, void* y, int rows, int cols,
                       long ldx, long ldy, float eps, int relu_in, void* stream) {
  G4R_REQUIRE(rows >= 0 && cols > 0 && (cols % 8) == 0, "layernorm: cols must be a multiple of 8");
  if (rows == 0) return G4R_OK;
  G4R_REQUIRE(x && gamma && beta && y && (ldx % 8) == 0 && (ldy % 8) == 0, "layernorm: bad pointer/stride");
  G4R_REQUIRE(cols <= 256 * NORM_MAXV * 8, "layernorm: cols <= 8192");
  hipLaunchKernelGGL(layernorm_bf16_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)x, gamma, beta, (h16_t*)y, rows, cols, ldx, ldy, eps, relu_in);
  G4R_CHECK_LAUNCH("layernorm");
  return G4R_OK;
}

int g4r_rmsnorm_bf16(const void* x, const float* gamma, void* y, int rows, int cols, long ldx, long ldy,
                     float eps, void* stream) {
  G4R_REQUIRE(rows >= 0 && cols > 0 && (cols % 8) == 0, "rmsnorm: cols must be a multiple of 8");
  if (rows == 0) return G4R_OK;
  G4R_REQUIRE(x && gamma && y && (ldx % 8) == 0 && (ldy % 8) == 0, "rmsnorm: bad pointer/stride");
  G4R_REQUIRE(cols <= 256 * NORM_MAXV * 8, "rmsnorm: cols <= 8192");
  hipLaunchKernelGGL(rmsnorm_bf16_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)x, gamma, (h16_t*)y, rows, cols, ldx, ldy, eps);
  G4R_CHECK_LAUNCH("rmsnorm");
  return G4R_OK;
}

/* x_out[row] = bf16(sum_s partials[s][row] + bias + residual[row]); y[row] = layernorm(x_out[row]; gamma, beta, eps): the
 * K-slice reduce of CLIP's fc2 (+ bias, + residual) folded into the next block's layer_norm1.  Bit-identical to
 * g4r_gemm_bf16_nt(splits, bias, residual) + g4r_layernorm_bf16. */
int g4r_layernorm_splitk_bf16(const float* partials, int splits, const float* bias, const void* residual, long ldr,
                              void* x_out, long ldxo, const float* gamma, const float* beta, void* y, long ldy, int rows,
                              int cols, float eps, void* stream) {
  G4R_REQUIRE(rows >= 0 && cols > 0 && (cols % 8) == 0 && splits >= 1, "layernorm_splitk: cols must be a multiple of 8");
  if (rows == 0) return G4R_OK;
  G4R_REQUIRE(partials && x_out && gamma && beta && y && (ldr % 8) == 0 && (ldxo % 8) == 0 && (ldy % 8) == 0,
              "layernorm_splitk: bad pointer/stride");
  G4R_REQUIRE(cols <= 256 * NORM_MAXV * 8, "layernorm_splitk: cols <= 8192");
  hipLaunchKernelGGL(layernorm_splitk_bf16_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, partials, splits,
                     (long)rows * cols, bias, (const h16_t*)residual, ldr, (h16_t*)x_out, ldxo, gamma, beta, (h16_t*)y, ldy,
                     cols, eps);
  G4R_CHECK_LAUNCH("layernorm_splitk");
  return G4R_OK;
}

/* x_out[row] = bf16(sum_s partials[s][row] + residual[row]); y[row] = rmsnorm(x_out[row]; gamma, eps).  partials: fp32
 * [splits][rows][cols] as written by g4r_gemm_bf16_nt_partials.  Bit-identical to the split-K reduce of g4r_gemm_bf16_nt
 * (no bias, no activation, + residual) followed by g4r_rmsnorm_bf16. */
int g4r_rmsnorm_splitk_bf16(const float* partials, int splits, const void* residual, long ldr, void* x_out, long ldxo,
                            const float* gamma, void* y, long ldy, int rows, int cols, float eps, void* stream) {
  G4R_REQUIRE(rows >= 0 && cols > 0 && (cols % 8) == 0 && splits >= 1, "rmsnorm_splitk: cols must be a multiple of 8");
  if (rows == 0) return G4R_OK;
  G4R_REQUIRE(partials && x_out && gamma && y && (ldr % 8) == 0 && (ldxo % 8) == 0 && (ldy % 8) == 0,
              "rmsnorm_splitk: bad pointer/stride");
  G4R_REQUIRE(cols <= 256 * NORM_MAXV * 8, "rmsnorm_splitk: cols <= 8192");
  hipLaunchKernelGGL(rmsnorm_splitk_bf16_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, partials, splits,
                     (long)rows * cols, (const h16_t*)residual, ldr, (h16_t*)x_out, ldxo, gamma, (h16_t*)y, ldy, cols, eps);
  G4R_CHECK_LAUNCH("rmsnorm_splitk");
  return G4R_OK;
}

// partial: workspace of B * G4R_GN_MAX_CHUNKS(256) * G * 2 floats; scale_shift: [B, 2, C] floats.
int g4r_groupnorm_affine_nhwc_bf16(const void* x, const float* gamma, const float* beta, float* partial,
                                   float* scale_shift, int B, int HW, int C, int G, float eps,
                                   void* stream) {
  G4R_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0 && (C / G) % 8 == 0,
              "groupnorm: channels per group must be a multiple of 8");
  const int nvec = C / 8;
  G4R_REQUIRE(nvec <= 256 && 256 % nvec == 0 && G <= 256, "groupnorm: C/8 must divide 256");
  G4R_REQUIRE(x && gamma && beta && partial && scale_shift, "groupnorm: null pointer");
  int chunks = 256;                       // x B workgroups, each streaming >= 32 pixels
  int ppb = g4r_ceil_div(HW, chunks);
  if (ppb < 32) ppb = 32;
  chunks = g4r_ceil_div(HW, ppb);
  hipLaunchKernelGGL(gn_stats_nhwc_kernel, dim3(chunks, B), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)x, partial, HW, C, G, ppb);
  G4R_CHECK_LAUNCH("gn_stats");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(g4r_ceil_div((long)B * G, 4)), dim3(256), 0,
                     (hipStream_t)stream, partial, gamma, beta, scale_shift, B, C, G, chunks,
                     (double)HW * (double)(C / G), eps);
  G4R_CHECK_LAUNCH("gn_finalize");
  return G4R_OK;
}

}  // extern "C"
