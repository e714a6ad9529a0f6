// spi_backward.hip -- backward kernels of the region-feature module (8d config 3:
// stage-1 training updates exactly this module, /root/reference/train_stage1.sh + gpt4roi/train/train.py:698-712).
// The reference gets these gradients from autograd through MLVLFuseModule / MlvlRoIExtractor
// (gpt4roi/models/layers.py:96-335).  Gradient maps are accumulated in fp32 NHWC ("d_y" = gradient with respect
// to the POST GroupNorm+ReLU feature map of a round); matrix gradients reuse the forward GEMM / implicit-GEMM.
//   groupnorm_stats      per-(image, group) mean / rstd of a raw conv output (recomputed, the forward keeps only
//                        the folded affine)
//   gn_relu_bwd          y = relu(GN(z)):  d_y (fp32) -> dz (bf16), dgamma, dbeta        (two passes)
//   fuse_shuffle_bwd     transpose of the channel shuffle + align_corners bilinear resampling (layers.py:152-180)
//   nhwc_to_cm_padded    NHWC map -> channel-major, zero-padded rows (+ x-shifted copies): operands of the 3x3
//                        weight-gradient GEMMs  dW[co,ci,ky,kx] = sum_p dY^T[co][p] * X^T[ci][p + (ky-1)*Wp + (kx-1)]
#include "g4r_common.h"

namespace {

struct F8 {
  float v[8];
};
__device__ __forceinline__ F8 ld8(const h16_t* p) {
  const uint4v r = *reinterpret_cast<const uint4v*>(p);
  F8 a;
  a.v[0] = h16lo(r.x); a.v[1] = h16hi(r.x); a.v[2] = h16lo(r.y); a.v[3] = h16hi(r.y);
  a.v[4] = h16lo(r.z); a.v[5] = h16hi(r.z); a.v[6] = h16lo(r.w); a.v[7] = h16hi(r.w);
  return a;
}
__device__ __forceinline__ void st8(h16_t* p, const F8& a) {
  uint4v w;
  w.x = pack_h16x2(a.v[0], a.v[1]); w.y = pack_h16x2(a.v[2], a.v[3]);
  w.z = pack_h16x2(a.v[4], a.v[5]); w.w = pack_h16x2(a.v[6], a.v[7]);
  *reinterpret_cast<uint4v*>(p) = w;
}
__device__ __forceinline__ F8 ld8f(const float* p) {
  const float4v a = *reinterpret_cast<const float4v*>(p);
  const float4v b = *reinterpret_cast<const float4v*>(p + 4);
  F8 r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}

inline int grid_for(long total) {
  long b = (total + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- per-(image, group) statistics: chunk partials, then one wave per (b, g) ------------------------------
__global__ __launch_bounds__(256) void gn_stats_partial_kernel(const h16_t* __restrict__ x, float* __restrict__ partial,
                                                               int HW, int C, int G, int pix_per_block) {
  __shared__ float red[2][256];
  const int b = blockIdx.y;
  const int nvec = C >> 3;
  const int tid = threadIdx.x;
  const int cv = tid % nvec, pl = tid / nvec, plc = 256 / nvec;
  const int p0 = blockIdx.x * pix_per_block;
  int p1 = p0 + pix_per_block;
  if (p1 > HW) p1 = HW;
  float s = 0.f, s2 = 0.f;
  const h16_t* base = x + ((size_t)b * HW) * C + cv * 8;
  for (int p = p0 + pl; p < p1; p += plc) {
    const F8 f = ld8(base + (size_t)p * C);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s += f.v[k];
      s2 += f.v[k] * f.v[k];
    }
  }
  red[0][tid] = s;
  red[1][tid] = s2;
  __syncthreads();
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

__global__ __launch_bounds__(256) void gn_stats_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats,
                                                                int B, int G, int chunks, double count, float eps) {
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
  if (lane == 0) {
    const double mean = sum / count;
    double var = sq / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[(size_t)bg * 2] = (float)mean;
    stats[(size_t)bg * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// ---- GN + ReLU backward, pass 1: reductions ---------------------------------------------------------------
//   dyh = d_y * (a*z + s > 0);  xhat = (z - mean) * rstd
//   dgamma[c] += sum dyh * xhat;  dbeta[c] += sum dyh           (over images and pixels)
//   gsum[b][g] += (sum gamma*dyh, sum gamma*dyh*xhat)           (over the group's pixels x channels)
__global__ __launch_bounds__(256) void gn_relu_bwd_reduce_kernel(const h16_t* __restrict__ z, const float* __restrict__ dy,
                                                                 const float* __restrict__ aff, const float* __restrict__ gamma,
                                                                 const float* __restrict__ stats, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, float* __restrict__ gsum, int HW,
                                                                 int C, int G, int pix_per_block) {
  __shared__ float red[2][256];
  const int b = blockIdx.y;
  const int nvec = C >> 3;
  const int tid = threadIdx.x;
  const int cv = tid % nvec, pl = tid / nvec, plc = 256 / nvec;
  const int vpg = (C / G) >> 3;
  const int g = cv / vpg;
  const int p0 = blockIdx.x * pix_per_block;
  int p1 = p0 + pix_per_block;
  if (p1 > HW) p1 = HW;
  const F8 a = ld8f(aff + (size_t)b * 2 * C + cv * 8);
  const F8 s = ld8f(aff + (size_t)b * 2 * C + C + cv * 8);
  const F8 gm = ld8f(gamma + cv * 8);
  const float mean = stats[((size_t)b * G + g) * 2], rstd = stats[((size_t)b * G + g) * 2 + 1];
  float db[8], dg[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) db[k] = dg[k] = 0.f;
  // four pixels' loads (4 x 48 B per lane) go out before the first is used: with one load pair in flight the pass ran at
  // ~1.2 TB/s (profiles/r03_train_step_kernel_stats.csv: 565 us average per launch), i.e. on the HBM latency, not its bandwidth
  auto accumulate = [&](const F8& zv, const F8& d) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float dyh = (a.v[k] * zv.v[k] + s.v[k] > 0.f) ? d.v[k] : 0.f;
      const float xh = (zv.v[k] - mean) * rstd;
      db[k] += dyh;
      dg[k] += dyh * xh;
      s1 += gm.v[k] * dyh;
      s2 += gm.v[k] * dyh * xh;
    }
  };
  int p = p0 + pl;
  for (; p + 3 * plc < p1; p += 4 * plc) {
    F8 zv[4], d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t off = ((size_t)b * HW + p + u * plc) * C + cv * 8;
      zv[u] = ld8(z + off);
      d[u] = ld8f(dy + off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) accumulate(zv[u], d[u]);
  }
  for (; p < p1; p += plc) {
    const size_t off = ((size_t)b * HW + p) * C + cv * 8;
    accumulate(ld8(z + off), ld8f(dy + off));
  }
  // the workgroup's pixel lanes are combined in LDS before the atomics, and the launch uses ~512 workgroups instead of 2048:
  // 8.4 M fp32 atomics onto 2 x 1024 addresses per launch were the cost of this pass, not its 1.8 GB of input
  __shared__ float comb[2][2048];
  const bool combine = plc > 1 && C <= 2048;          // wave-uniform
  if (combine) {
    for (int l = 1; l < plc; ++l) {                   // lanes pl = 1 .. plc-1 hand their sums to lane pl = 0, one at a time
      if (pl == l) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          comb[0][cv * 8 + k] = dg[k];
          comb[1][cv * 8 + k] = db[k];
        }
      }
      __syncthreads();
      if (pl == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          dg[k] += comb[0][cv * 8 + k];
          db[k] += comb[1][cv * 8 + k];
        }
      }
      __syncthreads();
    }
  }
  if (!combine || pl == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      unsafeAtomicAdd(dgamma + cv * 8 + k, dg[k]);
      unsafeAtomicAdd(dbeta + cv * 8 + k, db[k]);
    }
  }
  red[0][tid] = s1;
  red[1][tid] = s2;
  __syncthreads();
  if (tid < G) {
    float t1 = 0.f, t2 = 0.f;
    for (int l = 0; l < plc; ++l)
      for (int v = 0; v < vpg; ++v) {
        const int t = l * nvec + tid * vpg + v;
        t1 += red[0][t];
        t2 += red[1][t];
      }
    unsafeAtomicAdd(gsum + ((size_t)b * G + tid) * 2, t1);
    unsafeAtomicAdd(gsum + ((size_t)b * G + tid) * 2 + 1, t2);
  }
}

// ---- pass 2: dz = rstd * (gamma*dyh - m1 - xhat*m2),  m = gsum / (HW * C/G) -------------------------------
__global__ __launch_bounds__(256) void gn_relu_bwd_apply_kernel(const h16_t* __restrict__ z, const float* __restrict__ dy,
                                                                const float* __restrict__ aff, const float* __restrict__ gamma,
                                                                const float* __restrict__ stats, const float* __restrict__ gsum,
                                                                h16_t* __restrict__ dz, int B, int HW, int C, int G) {
  const int nvec = C >> 3;
  const int vpg = (C / G) >> 3;
  const float inv_n = 1.f / ((float)HW * (float)(C / G));
  const long total = (long)B * HW * nvec;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cv = (int)(i % nvec);
    const long pix = i / nvec;
    const int b = (int)(pix / HW);
    const int g = cv / vpg;
    const F8 a = ld8f(aff + (size_t)b * 2 * C + cv * 8);
    const F8 s = ld8f(aff + (size_t)b * 2 * C + C + cv * 8);
    const F8 gm = ld8f(gamma + cv * 8);
    const float mean = stats[((size_t)b * G + g) * 2], rstd = stats[((size_t)b * G + g) * 2 + 1];
    const float m1 = gsum[((size_t)b * G + g) * 2] * inv_n, m2 = gsum[((size_t)b * G + g) * 2 + 1] * inv_n;
    const size_t off = (size_t)pix * C + cv * 8;
    const F8 zv = ld8(z + off);
    const F8 d = ld8f(dy + off);
    F8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float dyh = (a.v[k] * zv.v[k] + s.v[k] > 0.f) ? d.v[k] : 0.f;
      const float xh = (zv.v[k] - mean) * rstd;
      o.v[k] = rstd * (gm.v[k] * dyh - m1 - xh * m2);
    }
    st8(dz + off, o);
  }
}

// ---- transpose of fuse_shuffle (elementwise.hip): scatter d_inp into the three source gradients ---------------
struct Lerp {
  int i0, i1;
  float w1;
};
__device__ __forceinline__ Lerp lerp_ac(int dst, int in_size, int out_size) {
  Lerp l;
  if (out_size <= 1) {
    l.i0 = l.i1 = 0;
    l.w1 = 0.f;
    return l;
  }
  const float scale = (float)(in_size - 1) / (float)(out_size - 1);
  const float src = scale * (float)dst;
  int i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  l.i0 = i0;
  l.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l.w1 = src - (float)i0;
  return l;
}

struct ShuffleDst {
  float* g;  // fp32 [B, H, W, C]
  int H, W;
};

__device__ __forceinline__ void add8(float* p, const F8& v, float w) {
#pragma unroll
  for (int k = 0; k < 8; ++k) unsafeAtomicAdd(p + k, v.v[k] * w);
}

__device__ __forceinline__ void scatter_dst(const ShuffleDst& s, int b, int y, int x, int H, int W, int C, int c0,
                                            const F8& v) {
  float* base = s.g + (size_t)b * s.H * s.W * C + c0;
  if (s.H == H && s.W == W) {
    add8(base + ((size_t)y * W + x) * C, v, 1.f);
    return;
  }
  const Lerp ly = lerp_ac(y, s.H, H), lx = lerp_ac(x, s.W, W);
  const float wy1 = ly.w1, wy0 = 1.f - wy1, wx1 = lx.w1, wx0 = 1.f - wx1;
  add8(base + ((size_t)ly.i0 * s.W + lx.i0) * C, v, wy0 * wx0);
  add8(base + ((size_t)ly.i0 * s.W + lx.i1) * C, v, wy0 * wx1);
  add8(base + ((size_t)ly.i1 * s.W + lx.i0) * C, v, wy1 * wx0);
  add8(base + ((size_t)ly.i1 * s.W + lx.i1) * C, v, wy1 * wx1);
}

__global__ __launch_bounds__(256) void fuse_shuffle_bwd_kernel(const h16_t* __restrict__ dinp, ShuffleDst own,
                                                               ShuffleDst top, ShuffleDst down, int B, int C) {
  const int H = own.H, W = own.W;
  const int nvec = C >> 3;
  const int R = C >> 1, S = C >> 2;
  const long total = (long)B * H * W * nvec;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % nvec);
    const long pix = i / nvec;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int b = (int)(pix / ((long)W * H));
    const int c = v * 8;
    const F8 d = ld8(dinp + (size_t)pix * C + c);
    if (c < R)
      scatter_dst(own, b, y, x, H, W, C, c, d);
    else if (c < R + S)
      scatter_dst(top, b, y, x, H, W, C, c + S, d);
    else
      scatter_dst(down, b, y, x, H, W, C, c - S, d);
  }
}

// ---- the same transpose as a GATHER, one launch per SOURCE level: every gradient element is written once --------
// (no zero-fill, no atomics; ~5x faster than the scatter on MI355X where 10^8 fp32 atomics per round dominate).
//   channels [0, R)      <- d_inp of the level itself
//   channels [R, R+S)    <- channel c+S of the COARSER target (which read this level as `down`) and, for level 0,
//                           of the level itself (identity resize)
//   channels [R+S, C)    <- channel c-S of the FINER target (which read this level as `top`) and, for the last
//                           level, of the level itself
// For a resampled target the candidate target rows/columns around src/scale are re-tested with the forward's own
// lerp_ac(), so the weights are exactly the forward's.
struct GatherTgt {
  const h16_t* g;  // bf16 [B, H, W, C] or null
  int H, W;
};

__device__ __forceinline__ void cand_range(int j, int in_size, int out_size, int& lo, int& hi) {
  if (out_size <= 1 || in_size <= 1) {
    lo = 0;
    hi = out_size - 1;
    return;
  }
  const float inv = (float)(out_size - 1) / (float)(in_size - 1);
  lo = (int)floorf((float)(j - 1) * inv) - 1;
  hi = (int)ceilf((float)(j + 1) * inv) + 1;
  if (lo < 0) lo = 0;
  if (hi > out_size - 1) hi = out_size - 1;
}

__device__ __forceinline__ void gather_tgt(const GatherTgt& t, int b, int ys, int xs, int Hs, int Ws, int C, int ch,
                                           F8& acc) {
  int ylo, yhi, xlo, xhi;
  cand_range(ys, Hs, t.H, ylo, yhi);
  cand_range(xs, Ws, t.W, xlo, xhi);
  const h16_t* base = t.g + (size_t)b * t.H * t.W * C + ch;
  for (int i = ylo; i <= yhi; ++i) {
    const Lerp ly = lerp_ac(i, Hs, t.H);
    const float wy = (ly.i0 == ys ? 1.f - ly.w1 : 0.f) + (ly.i1 == ys ? ly.w1 : 0.f);
    if (wy == 0.f) continue;
    for (int j = xlo; j <= xhi; ++j) {
      const Lerp lx = lerp_ac(j, Ws, t.W);
      const float wx = (lx.i0 == xs ? 1.f - lx.w1 : 0.f) + (lx.i1 == xs ? lx.w1 : 0.f);
      if (wx == 0.f) continue;
      const F8 d = ld8(base + ((size_t)i * t.W + j) * C);
      const float w = wy * wx;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc.v[k] += w * d.v[k];
    }
  }
}

__global__ __launch_bounds__(256) void fuse_shuffle_bwd_gather_kernel(float* __restrict__ dsrc, const h16_t* __restrict__ own,
                                                                      GatherTgt fine, GatherTgt coarse, int self_top,
                                                                      int self_down, int B, int H, int W, int C) {
  const int nvec = C >> 3;
  const int R = C >> 1, S = C >> 2;
  const long total = (long)B * H * W * nvec;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % nvec);
    const long pix = i / nvec;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int b = (int)(pix / ((long)W * H));
    const int c = v * 8;
    F8 acc;
    if (c < R) {
      acc = ld8(own + (size_t)pix * C + c);
    } else if (c < R + S) {
      if (self_down) acc = ld8(own + (size_t)pix * C + c + S);
      else {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc.v[k] = 0.f;
      }
      if (coarse.g) gather_tgt(coarse, b, y, x, H, W, C, c + S, acc);
    } else {
      if (self_top) acc = ld8(own + (size_t)pix * C + c - S);
      else {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc.v[k] = 0.f;
      }
      if (fine.g) gather_tgt(fine, b, y, x, H, W, C, c - S, acc);
    }
    float* o = dsrc + (size_t)pix * C + c;
    *reinterpret_cast<float4v*>(o) = float4v{acc.v[0], acc.v[1], acc.v[2], acc.v[3]};
    *reinterpret_cast<float4v*>(o + 4) = float4v{acc.v[4], acc.v[5], acc.v[6], acc.v[7]};
  }
}

// ---- NHWC [B, H, W, C] -> channel-major padded rows ------------------------------------------------------------
// dst[s][c][base + b*seg + (y+1)*Wp + (x+1) - dx_s] = src[b][y][x][c],  dx_s = s - (n_shift >> 1)
// (n_shift = 1: the plain copy; n_shift = 3: copies pre-shifted by -1, 0, +1 columns so that every tap's operand
// starts 16-byte aligned).  The destination is zero-initialised once by the caller; pad positions are never written.
__global__ __launch_bounds__(256) void nhwc_to_cm_padded_kernel(const h16_t* __restrict__ src, h16_t* __restrict__ dst,
                                                                int B, int H, int W, int C, int Wp, long seg, long base,
                                                                long ltot, int n_shift) {
  __shared__ h16_t tile[64][66];
  const int xt = (W + 63) / 64;
  const int bx = blockIdx.x % xt;
  const int y = (blockIdx.x / xt) % H;
  const int b = blockIdx.x / (xt * H);
  const int c0 = blockIdx.y * 64;
  const int x0 = bx * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {  // j = pixel, tx = channel
    const int x = x0 + j, c = c0 + tx;
    tile[j][tx] = (x < W && c < C) ? src[(((size_t)b * H + y) * W + x) * C + c] : (h16_t)0;
  }
  __syncthreads();
  const long row = base + (long)b * seg + (long)(y + 1) * Wp + 1;
  for (int j = ty; j < 64; j += 4) {  // j = channel, tx = pixel
    const int c = c0 + j, x = x0 + tx;
    if (c < C && x < W) {
      const h16_t v = tile[tx][j];
      for (int s = 0; s < n_shift; ++s) {
        const int dx = s - (n_shift >> 1);
        dst[((size_t)s * C + c) * ltot + row + x - dx] = v;
      }
    }
  }
}

}  // namespace

extern "C" {

int g4r_groupnorm_stats_nhwc_bf16(const void* x, float* partial, float* stats, int B, int HW, int C, int G, float eps,
                                  void* stream) {
  G4R_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0 && (C / G) % 8 == 0, "groupnorm_stats: bad shape");
  const int nvec = C / 8;
  G4R_REQUIRE(nvec <= 256 && 256 % nvec == 0 && G <= 256, "groupnorm_stats: C/8 must divide 256");
  G4R_REQUIRE(x && partial && stats, "groupnorm_stats: null pointer");
  int chunks = 256;
  int ppb = g4r_ceil_div(HW, chunks);
  if (ppb < 32) ppb = 32;
  chunks = g4r_ceil_div(HW, ppb);
  hipLaunchKernelGGL(gn_stats_partial_kernel, dim3(chunks, B), dim3(256), 0, (hipStream_t)stream, (const h16_t*)x,
                     partial, HW, C, G, ppb);
  G4R_CHECK_LAUNCH("gn_stats_partial");
  hipLaunchKernelGGL(gn_stats_finalize_kernel, dim3(g4r_ceil_div((long)B * G, 4)), dim3(256), 0, (hipStream_t)stream,
                     partial, stats, B, G, chunks, (double)HW * (double)(C / G), eps);
  G4R_CHECK_LAUNCH("gn_stats_finalize");
  return G4R_OK;
}

int g4r_gn_relu_bwd_nhwc_bf16(const void* z, const float* dy, const float* affine, const float* gamma,
                              const float* stats, float* dgamma, float* dbeta, float* gsum, void* dz, int B, int HW,
                              int C, int G, void* stream) {
  G4R_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0 && (C / G) % 8 == 0, "gn_relu_bwd: bad shape");
  const int nvec = C / 8;
  G4R_REQUIRE(nvec <= 256 && 256 % nvec == 0 && G <= 256, "gn_relu_bwd: C/8 must divide 256");
  G4R_REQUIRE(z && dy && affine && gamma && stats && dgamma && dbeta && gsum && dz, "gn_relu_bwd: null pointer");
  int chunks = 512 / B;                               // ~512 workgroups per launch (two per CU): few atomics, enough loads in flight
  if (chunks < 16) chunks = 16;
  if (chunks > 256) chunks = 256;
  int ppb = g4r_ceil_div(HW, chunks);
  if (ppb < 32) ppb = 32;
  chunks = g4r_ceil_div(HW, ppb);
  hipError_t e = hipMemsetAsync(gsum, 0, (size_t)B * G * 2 * sizeof(float), (hipStream_t)stream);
  if (e != hipSuccess) return g4r_note_hip_error(e, "gn_relu_bwd: memset");
  hipLaunchKernelGGL(gn_relu_bwd_reduce_kernel, dim3(chunks, B), dim3(256), 0, (hipStream_t)stream, (const h16_t*)z,
                     dy, affine, gamma, stats, dgamma, dbeta, gsum, HW, C, G, ppb);
  G4R_CHECK_LAUNCH("gn_relu_bwd_reduce");
  hipLaunchKernelGGL(gn_relu_bwd_apply_kernel, dim3(grid_for((long)B * HW * nvec)), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)z, dy, affine, gamma, stats, gsum, (h16_t*)dz, B, HW, C, G);
  G4R_CHECK_LAUNCH("gn_relu_bwd_apply");
  return G4R_OK;
}

int g4r_fuse_shuffle_bwd_nhwc_bf16(const void* dinp, int H, int W, float* d_own, float* d_top, int Ht, int Wt,
                                   float* d_down, int Hd, int Wd, int B, int C, void* stream) {
  G4R_REQUIRE(B > 0 && H > 0 && W > 0 && C % 32 == 0, "fuse_shuffle_bwd: bad shape");
  G4R_REQUIRE(dinp && d_own && d_top && d_down, "fuse_shuffle_bwd: null pointer");
  ShuffleDst own = {d_own, H, W}, top = {d_top, Ht, Wt}, down = {d_down, Hd, Wd};
  hipLaunchKernelGGL(fuse_shuffle_bwd_kernel, dim3(grid_for((long)B * H * W * (C / 8))), dim3(256), 0,
                     (hipStream_t)stream, (const h16_t*)dinp, own, top, down, B, C);
  G4R_CHECK_LAUNCH("fuse_shuffle_bwd");
  return G4R_OK;
}

int g4r_fuse_shuffle_bwd_gather_nhwc_bf16(float* d_src, const void* dinp_own, int H, int W, const void* dinp_fine,
                                          int Hf, int Wf, const void* dinp_coarse, int Hc, int Wc, int self_top,
                                          int self_down, int B, int C, void* stream) {
  G4R_REQUIRE(B > 0 && H > 0 && W > 0 && C % 32 == 0, "fuse_shuffle_bwd_gather: bad shape");
  G4R_REQUIRE(d_src && dinp_own, "fuse_shuffle_bwd_gather: null pointer");
  GatherTgt fine = {(const h16_t*)dinp_fine, Hf, Wf}, coarse = {(const h16_t*)dinp_coarse, Hc, Wc};
  hipLaunchKernelGGL(fuse_shuffle_bwd_gather_kernel, dim3(grid_for((long)B * H * W * (C / 8))), dim3(256), 0,
                     (hipStream_t)stream, d_src, (const h16_t*)dinp_own, fine, coarse, self_top, self_down, B, H, W, C);
  G4R_CHECK_LAUNCH("fuse_shuffle_bwd_gather");
  return G4R_OK;
}

int g4r_nhwc_to_cm_padded_bf16(const void* src, void* dst, int B, int H, int W, int C, int Wp, long seg, long base,
                               long ltot, int n_shift, void* stream) {
  G4R_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Wp >= W + 2 && seg >= (long)(H + 2) * Wp, "nhwc_to_cm: bad shape");
  G4R_REQUIRE(base >= 1 && base + (long)B * seg + 1 <= ltot && (n_shift == 1 || n_shift == 3), "nhwc_to_cm: bad layout");
  G4R_REQUIRE(src && dst, "nhwc_to_cm: null pointer");
  const long blocks = (long)B * H * ((W + 63) / 64);
  G4R_REQUIRE(blocks < 2147483647L, "nhwc_to_cm: grid too large");
  hipLaunchKernelGGL(nhwc_to_cm_padded_kernel, dim3((unsigned)blocks, g4r_ceil_div(C, 64)), dim3(256), 0,
                     (hipStream_t)stream, (const h16_t*)src, (h16_t*)dst, B, H, W, C, Wp, seg, base, ltot, n_shift);
  G4R_CHECK_LAUNCH("nhwc_to_cm_padded");
  return G4R_OK;
}

}  // extern "C"
