// backward.hip -- the HBM-bound backward / optimizer kernels of the training rows (SURVEY.md 8d configs 3/4,
// 8a rows a2, a16, a18): what autograd + torch.optim.AdamW do for the reference under HF Trainer
// (/root/reference/gpt4roi/train/train.py:698-712).  bf16 storage, fp32 arithmetic, 16-byte accesses.
//   rmsnorm_bwd / layernorm_bwd   HF LlamaRMSNorm, pos_embedd LayerNorms (gpt4roi/models/layers.py:260-267)
//   swiglu_il (+bwd)              SiLU(gate)*up over the interleaved (gate, up) columns the fused GEMM produces
//   rope_qkv_bwd                  inverse rotation of dq, dk + pass-through of dv into the fused d(qkv) buffer
//   cross_entropy                 shifted-label CE of llava/model/llava.py:240-252, loss and dlogits in one pass
//   transpose                     [R, C] -> [C, R_pad] (zero padded): operands of the weight-gradient GEMMs
//   colsum / relu_bwd / gather_rows / scatter helpers
//   adamw                         fused AdamW on fp32 master weights + bf16 kernel copy
#include "g4r_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
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

__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

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

constexpr int NORM_MAXV = 4;  // rows up to 8192 elements

// ---- RMSNorm backward: y = bf16(x * rstd) * gamma -------------------------------------------------
//   dx = dres + rstd * (gamma*dy - xhat * mean(gamma*dy*xhat)),  xhat = x * rstd
//   dgamma[c] += dy * xhat  (fp32 atomics, only when dgamma != null: stage 2)
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const h16_t* __restrict__ x, const float* __restrict__ gamma,
                                                          const h16_t* __restrict__ dy, const h16_t* __restrict__ dres,
                                                          h16_t* __restrict__ dx, float* __restrict__ dgamma,
                                                          int rows, int cols, long ldx, long lddy, long lddr, long lddx,
                                                          float eps, int rows_per_block) {
  __shared__ float red[4];
  const int nvec = cols >> 3;
  // dgamma: a workgroup walks `rows_per_block` rows and keeps the column sums in registers, so the atomics per
  // column drop from one per row to one per workgroup (767 rows hammering 4096 addresses cost 238 us per call)
  float dg[NORM_MAXV][8];
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) dg[i][k] = 0.f;
  for (int rr = 0; rr < rows_per_block; ++rr) {
    const int row = blockIdx.x * rows_per_block + rr;
    if (row >= rows) break;  // uniform across the workgroup
    F8 xv[NORM_MAXV], gv[NORM_MAXV];
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_MAXV; ++i) {
      const int v = threadIdx.x + i * 256;
      if (v < nvec) {
        xv[i] = ld8(x + (size_t)row * ldx + v * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) s2 += xv[i].v[k] * xv[i].v[k];
      }
    }
    const float rstd = rsqrtf(block_sum(s2, red) / (float)cols + eps);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_MAXV; ++i) {
      const int v = threadIdx.x + i * 256;
      if (v < nvec) {
        const F8 g = ld8f(gamma + v * 8);
        const F8 d = ld8(dy + (size_t)row * lddy + v * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float xh = xv[i].v[k] * rstd;
          dg[i][k] += d.v[k] * xh;
          gv[i].v[k] = g.v[k] * d.v[k];
          dot += gv[i].v[k] * xh;
        }
      }
    }
    const float mdot = block_sum(dot, red) / (float)cols;
#pragma unroll
    for (int i = 0; i < NORM_MAXV; ++i) {
      const int v = threadIdx.x + i * 256;
      if (v < nvec) {
        F8 o;
        F8 r;
        if (dres) r = ld8(dres + (size_t)row * lddr + v * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float xh = xv[i].v[k] * rstd;
          o.v[k] = rstd * (gv[i].v[k] - xh * mdot) + (dres ? r.v[k] : 0.f);
        }
        st8(dx + (size_t)row * lddx + v * 8, o);
      }
    }
  }
  if (dgamma) {
#pragma unroll
    for (int i = 0; i < NORM_MAXV; ++i) {
      const int v = threadIdx.x + i * 256;
      if (v < nvec) {
#pragma unroll
        for (int k = 0; k < 8; ++k) unsafeAtomicAdd(dgamma + v * 8 + k, dg[i][k]);
      }
    }
  }
}

// ---- LayerNorm backward: y = (relu?(x) - mean) * rstd * gamma + beta ---------------------------------
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const h16_t* __restrict__ x, const float* __restrict__ gamma,
                                                            const h16_t* __restrict__ dy, h16_t* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int cols, long ldx, long lddy, long lddx, float eps,
                                                            int relu_in) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const int nvec = cols >> 3;
  F8 xv[NORM_MAXV], gv[NORM_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      xv[i] = ld8(x + (size_t)row * ldx + v * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (relu_in) xv[i].v[k] = fmaxf(xv[i].v[k], 0.f);
        s += xv[i].v[k];
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
        const float d = xv[i].v[k] - mean;
        s2 += d * d;
      }
    }
  const float rstd = rsqrtf(block_sum(s2, red) / (float)cols + eps);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      const F8 g = ld8f(gamma + v * 8);
      const F8 d = ld8(dy + (size_t)row * lddy + v * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (xv[i].v[k] - mean) * rstd;
        if (dgamma) unsafeAtomicAdd(dgamma + v * 8 + k, d.v[k] * xh);
        if (dbeta) unsafeAtomicAdd(dbeta + v * 8 + k, d.v[k]);
        gv[i].v[k] = g.v[k] * d.v[k];
        sg += gv[i].v[k];
        sgx += gv[i].v[k] * xh;
      }
    }
  }
  const float mg = block_sum(sg, red) / (float)cols;
  const float mgx = block_sum(sgx, red) / (float)cols;
  if (!dx) return;
#pragma unroll
  for (int i = 0; i < NORM_MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      F8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (xv[i].v[k] - mean) * rstd;
        float g = rstd * (gv[i].v[k] - mg - xh * mgx);
        if (relu_in && xv[i].v[k] <= 0.f) g = 0.f;
        o.v[k] = g;
      }
      st8(dx + (size_t)row * lddx + v * 8, o);
    }
  }
}

// ---- SwiGLU over interleaved (gate, up) columns: gu [T, 2F] with column 2c = gate_c, 2c+1 = up_c ----
__global__ __launch_bounds__(256) void swiglu_il_kernel(const h16_t* __restrict__ gu, h16_t* __restrict__ out, int T,
                                                        int F) {
  const int nvec = F >> 2;  // 4 outputs = 8 interleaved inputs per thread
  const long total = (long)T * nvec;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % nvec);
    const long t = i / nvec;
    const F8 a = ld8(gu + t * 2 * F + v * 8);
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g = a.v[2 * k], u = a.v[2 * k + 1];
      o[k] = g / (1.f + __expf(-g)) * u;
    }
    const uint2v w = {pack_h16x2(o[0], o[1]), pack_h16x2(o[2], o[3])};
    *reinterpret_cast<uint2v*>(out + t * F + v * 4) = w;
  }
}

__global__ __launch_bounds__(256) void swiglu_il_bwd_kernel(const h16_t* __restrict__ gu, const h16_t* __restrict__ dy,
                                                            h16_t* __restrict__ dgu, int T, int F) {
  const int nvec = F >> 2;
  const long total = (long)T * nvec;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % nvec);
    const long t = i / nvec;
    const F8 a = ld8(gu + t * 2 * F + v * 8);
    const uint2v dw = *reinterpret_cast<const uint2v*>(dy + t * F + v * 4);
    const float d[4] = {h16lo(dw.x), h16hi(dw.x), h16lo(dw.y), h16hi(dw.y)};
    F8 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g = a.v[2 * k], u = a.v[2 * k + 1];
      const float sg = 1.f / (1.f + __expf(-g));
      const float silu = g * sg;
      o.v[2 * k] = d[k] * u * (sg + silu * (1.f - sg));  // d silu / dg = sg * (1 + g * (1 - sg))
      o.v[2 * k + 1] = d[k] * silu;
    }
    st8(dgu + t * 2 * F + v * 8, o);
  }
}

// ---- rotary backward: d(qkv)[t] = [R^-1 dq, R^-1 dk, dv] ---------------------------------------------
__global__ __launch_bounds__(256) void rope_qkv_bwd_kernel(const h16_t* __restrict__ dq, const h16_t* __restrict__ dk,
                                                           const h16_t* __restrict__ dv, const float* __restrict__ cs,
                                                           const float* __restrict__ sn, h16_t* __restrict__ dqkv,
                                                           int T, int Hh, int D, int pos0, long ldq, long ldk, long ldv,
                                                           int period) {
  const int half = D >> 1;
  const int hv = half >> 3;
  const long total = (long)T * Hh * hv;
  const int HD = Hh * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % hv);
    const int h = (int)((i / hv) % Hh);
    const int t = (int)(i / ((long)hv * Hh));
    const int pos = pos0 + (period > 0 ? t % period : t);      // rows of several sequences stacked: positions restart
    const F8 c = ld8f(cs + (size_t)pos * half + v * 8);
    const F8 s = ld8f(sn + (size_t)pos * half + v * 8);
    const size_t off = (size_t)h * D + v * 8;
    h16_t* out = dqkv + (size_t)t * 3 * HD;
    {
      const F8 a = ld8(dq + (size_t)t * ldq + off), b = ld8(dq + (size_t)t * ldq + off + half);
      F8 o1, o2;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        o1.v[k] = a.v[k] * c.v[k] + b.v[k] * s.v[k];
        o2.v[k] = b.v[k] * c.v[k] - a.v[k] * s.v[k];
      }
      st8(out + off, o1);
      st8(out + off + half, o2);
    }
    {
      const F8 a = ld8(dk + (size_t)t * ldk + off), b = ld8(dk + (size_t)t * ldk + off + half);
      F8 o1, o2;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        o1.v[k] = a.v[k] * c.v[k] + b.v[k] * s.v[k];
        o2.v[k] = b.v[k] * c.v[k] - a.v[k] * s.v[k];
      }
      st8(out + HD + off, o1);
      st8(out + HD + off + half, o2);
    }
    *reinterpret_cast<uint4v*>(out + 2 * HD + off) = *reinterpret_cast<const uint4v*>(dv + (size_t)t * ldv + off);
    *reinterpret_cast<uint4v*>(out + 2 * HD + off + half) =
        *reinterpret_cast<const uint4v*>(dv + (size_t)t * ldv + off + half);
  }
}

// ---- cross entropy (one workgroup per row) -----------------------------------------------------------
//   label < 0 (HF ignore_index -100): no loss, zero gradient.
//   loss_sum += lse - logit[label];  dlogits = (softmax - onehot) * *grad_scale, zero in the pad columns.
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, const long* __restrict__ labels,
                                                            h16_t* __restrict__ dlogits, float* __restrict__ loss_sum,
                                                            const float* __restrict__ grad_scale, int N, long ld,
                                                            long ldd, int n_pad) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const float* lr = logits + (size_t)row * ld;
  h16_t* dr = dlogits ? dlogits + (size_t)row * ldd : nullptr;
  const long label = labels[row];
  if (label < 0 || label >= N) {
    if (dr)
      for (int c = threadIdx.x; c < n_pad; c += 256) dr[c] = 0;
    return;
  }
  float m = -INFINITY;
  for (int c = threadIdx.x; c < N; c += 256) m = fmaxf(m, lr[c]);
  m = block_max(m, red);
  float s = 0.f;
  for (int c = threadIdx.x; c < N; c += 256) s += __expf(lr[c] - m);
  s = block_sum(s, red);
  const float lse = m + __logf(s);
  if (threadIdx.x == 0) unsafeAtomicAdd(loss_sum, lse - lr[label]);
  if (!dr) return;
  const float gs = *grad_scale;
  const float inv = 1.f / s;
  for (int c = threadIdx.x; c < n_pad; c += 256) {
    float g = 0.f;
    if (c < N) g = (__expf(lr[c] - m) * inv - (c == label ? 1.f : 0.f)) * gs;
    dr[c] = f32_to_h16(g);
  }
}

// ---- transpose: out[c][r] = in[r][c]; rows R..R_pad of the (transposed) output are zero -----------------
__global__ __launch_bounds__(256) void transpose_kernel(const h16_t* __restrict__ in, h16_t* __restrict__ out, int R,
                                                        int C, long ld_in, long ld_out, int R_pad) {
  __shared__ h16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < C) ? in[(size_t)r * ld_in + c] : (h16_t)0;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int c = c0 + j, r = r0 + tx;
    if (c < C && r < R_pad) out[(size_t)c * ld_out + r] = tile[tx][j];
  }
}

// ---- column sums of a bf16 [M, N] matrix -> fp32 [N] (bias gradients); accumulates ----------------------
__global__ __launch_bounds__(256) void colsum_kernel(const h16_t* __restrict__ x, float* __restrict__ out, int M, int N,
                                                     long ld, int rows_per_block) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  const int r0 = blockIdx.y * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += h16_to_f32(x[(size_t)r * ld + c]);
  unsafeAtomicAdd(out + c, s);
}

// ---- dx = dy where y > 0 else 0 ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_bwd_kernel(const h16_t* __restrict__ y, const h16_t* __restrict__ dy,
                                                       h16_t* __restrict__ dx, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const F8 a = ld8(y + i * 8), d = ld8(dy + i * 8);
    F8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = a.v[k] > 0.f ? d.v[k] : 0.f;
    st8(dx + i * 8, o);
  }
}

// ---- dst[i] = src[idx[i]] (rows of C bf16): gradient of the <bbox> / image-patch splice ------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const h16_t* __restrict__ src, const int* __restrict__ idx,
                                                          h16_t* __restrict__ dst, int n, int C, long ld_src,
                                                          long ld_dst) {
  const int nvec = C >> 3;
  const long total = (long)n * nvec;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % nvec);
    const int r = (int)(i / nvec);
    const int s = idx[r];
    uint4v w = {0u, 0u, 0u, 0u};
    if (s >= 0) w = *reinterpret_cast<const uint4v*>(src + (size_t)s * ld_src + v * 8);
    *reinterpret_cast<uint4v*>(dst + (size_t)r * ld_dst + v * 8) = w;
  }
}

// ---- out[idx[r]] += src[r] (fp32 accumulation of bf16 rows; idx < 0 = skip): embedding-table gradient -------------
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const h16_t* __restrict__ src, const int* __restrict__ idx,
                                                               float* __restrict__ out, int n, int C, long ld_src,
                                                               long ld_out) {
  const int nvec = C >> 3;
  const long total = (long)n * nvec;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % nvec);
    const int r = (int)(i / nvec);
    const int d = idx[r];
    if (d < 0) continue;
    const F8 a = ld8(src + (size_t)r * ld_src + v * 8);
    float* o = out + (size_t)d * ld_out + v * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) unsafeAtomicAdd(o + k, a.v[k]);
  }
}

// ---- AdamW (decoupled weight decay, torch.optim.AdamW semantics) ----------------------------------------
template <typename G>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const G* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, h16_t* __restrict__ p_bf16, long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                    float gscale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float gi;
    if (sizeof(G) == 2) gi = h16_to_f32(reinterpret_cast<const h16_t*>(g)[i]);
    else gi = reinterpret_cast<const float*>(g)[i];
    gi *= gscale;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (p_bf16) p_bf16[i] = f32_to_h16(pi);
  }
}

}  // namespace

extern "C" {

int g4r_rmsnorm_bwd_bf16(const void* x, const float* gamma, const void* dy, const void* dres, void* dx,
                         float* dgamma, int rows, int cols, long ldx, long lddy, long lddres, long lddx, float eps,
                         void* stream) {
  G4R_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 256 * NORM_MAXV * 8, "rmsnorm_bwd: bad shape");
  if (rows == 0) return G4R_OK;
  G4R_REQUIRE(x && gamma && dy && dx, "rmsnorm_bwd: null pointer");
  G4R_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && lddres % 8 == 0, "rmsnorm_bwd: bad stride");
  const int rpb = dgamma ? 8 : 1;
  hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(g4r_ceil_div(rows, rpb)), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)x, gamma, (const h16_t*)dy, (const h16_t*)dres, (h16_t*)dx, dgamma, rows, cols, ldx,
                     lddy, lddres, lddx, eps, rpb);
  G4R_CHECK_LAUNCH("rmsnorm_bwd");
  return G4R_OK;
}

int g4r_layernorm_bwd_bf16(const void* x, const float* gamma, const void* dy, void* dx, float* dgamma, float* dbeta,
                           int rows, int cols, long ldx, long lddy, long lddx, float eps, int relu_in, void* stream) {
  G4R_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 256 * NORM_MAXV * 8, "layernorm_bwd: bad shape");
  if (rows == 0) return G4R_OK;
  G4R_REQUIRE(x && gamma && dy, "layernorm_bwd: null pointer");
  G4R_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, "layernorm_bwd: bad stride");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, gamma,
                     (const h16_t*)dy, (h16_t*)dx, dgamma, dbeta, cols, ldx, lddy, lddx, eps, relu_in);
  G4R_CHECK_LAUNCH("layernorm_bwd");
  return G4R_OK;
}

int g4r_swiglu_il_bf16(const void* gate_up, void* out, int T, int F, void* stream) {
  G4R_REQUIRE(T >= 0 && F > 0 && F % 4 == 0, "swiglu_il: bad shape");
  if (T == 0) return G4R_OK;
  G4R_REQUIRE(gate_up && out, "swiglu_il: null pointer");
  hipLaunchKernelGGL(swiglu_il_kernel, dim3(grid_for((long)T * (F / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)gate_up, (h16_t*)out, T, F);
  G4R_CHECK_LAUNCH("swiglu_il");
  return G4R_OK;
}

int g4r_swiglu_il_bwd_bf16(const void* gate_up, const void* dy, void* dgate_up, int T, int F, void* stream) {
  G4R_REQUIRE(T >= 0 && F > 0 && F % 4 == 0, "swiglu_il_bwd: bad shape");
  if (T == 0) return G4R_OK;
  G4R_REQUIRE(gate_up && dy && dgate_up, "swiglu_il_bwd: null pointer");
  hipLaunchKernelGGL(swiglu_il_bwd_kernel, dim3(grid_for((long)T * (F / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)gate_up, (const h16_t*)dy, (h16_t*)dgate_up, T, F);
  G4R_CHECK_LAUNCH("swiglu_il_bwd");
  return G4R_OK;
}

static int rope_qkv_bwd_launch(const void* dq, const void* dk, const void* dv, const float* cos_tab, const float* sin_tab,
                               void* dqkv, int T, int heads, int head_dim, int pos0, long ldq, long ldk, long ldv, int period,
                               void* stream) {
  G4R_REQUIRE(T >= 0 && heads > 0 && head_dim % 16 == 0 && pos0 >= 0 && period >= 0, "rope_qkv_bwd: bad shape");
  if (T == 0) return G4R_OK;
  G4R_REQUIRE(dq && dk && dv && cos_tab && sin_tab && dqkv, "rope_qkv_bwd: null pointer");
  G4R_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0, "rope_qkv_bwd: bad stride");
  const long total = (long)T * heads * (head_dim / 16);
  hipLaunchKernelGGL(rope_qkv_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const h16_t*)dq,
                     (const h16_t*)dk, (const h16_t*)dv, cos_tab, sin_tab, (h16_t*)dqkv, T, heads, head_dim, pos0,
                     ldq, ldk, ldv, period);
  G4R_CHECK_LAUNCH("rope_qkv_bwd");
  return G4R_OK;
}

int g4r_rope_qkv_bwd_bf16(const void* dq, const void* dk, const void* dv, const float* cos_tab, const float* sin_tab,
                          void* dqkv, int T, int heads, int head_dim, int pos0, long ldq, long ldk, long ldv,
                          void* stream) {
  return rope_qkv_bwd_launch(dq, dk, dv, cos_tab, sin_tab, dqkv, T, heads, head_dim, pos0, ldq, ldk, ldv, 0, stream);
}

// The rows of a whole batch in one launch: `rows` = B * period stacked rows, row r sits at position pos0 + r % period.
int g4r_rope_qkv_bwd_batch_bf16(const void* dq, const void* dk, const void* dv, const float* cos_tab, const float* sin_tab,
                                void* dqkv, int rows, int period, int heads, int head_dim, int pos0, long ldq, long ldk,
                                long ldv, void* stream) {
  G4R_REQUIRE(period > 0 && rows % period == 0, "rope_qkv_bwd_batch: rows must be a multiple of the sequence length");
  return rope_qkv_bwd_launch(dq, dk, dv, cos_tab, sin_tab, dqkv, rows, heads, head_dim, pos0, ldq, ldk, ldv, period, stream);
}

int g4r_cross_entropy_f32(const float* logits, const long* labels, void* dlogits, float* loss_sum,
                          const float* grad_scale, int rows, int N, long ld, long ldd, int n_pad, void* stream) {
  G4R_REQUIRE(rows >= 0 && N > 0 && n_pad >= N, "cross_entropy: bad shape");
  if (rows == 0) return G4R_OK;
  G4R_REQUIRE(logits && labels && loss_sum, "cross_entropy: null pointer");
  G4R_REQUIRE(!dlogits || (grad_scale && ldd >= n_pad), "cross_entropy: dlogits needs grad_scale and ldd >= n_pad");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, labels,
                     (h16_t*)dlogits, loss_sum, grad_scale, N, ld, ldd, n_pad);
  G4R_CHECK_LAUNCH("cross_entropy");
  return G4R_OK;
}

int g4r_transpose_bf16(const void* in, void* out, int R, int C, long ld_in, long ld_out, int R_pad, void* stream) {
  G4R_REQUIRE(R >= 0 && C >= 0 && R_pad >= R && ld_in >= C && ld_out >= R_pad, "transpose: bad shape");
  if (R_pad == 0 || C == 0) return G4R_OK;
  G4R_REQUIRE(in && out, "transpose: null pointer");
  hipLaunchKernelGGL(transpose_kernel, dim3(g4r_ceil_div(C, 64), g4r_ceil_div(R_pad, 64)), dim3(256), 0,
                     (hipStream_t)stream, (const h16_t*)in, (h16_t*)out, R, C, ld_in, ld_out, R_pad);
  G4R_CHECK_LAUNCH("transpose");
  return G4R_OK;
}

int g4r_colsum_bf16(const void* x, float* out, int M, int N, long ld, void* stream) {
  G4R_REQUIRE(M >= 0 && N > 0 && ld >= N, "colsum: bad shape");
  if (M == 0) return G4R_OK;
  G4R_REQUIRE(x && out, "colsum: null pointer");
  const int rpb = 256;
  hipLaunchKernelGGL(colsum_kernel, dim3(g4r_ceil_div(N, 256), g4r_ceil_div(M, rpb)), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)x, out, M, N, ld, rpb);
  G4R_CHECK_LAUNCH("colsum");
  return G4R_OK;
}

int g4r_relu_bwd_bf16(const void* y, const void* dy, void* dx, long n, void* stream) {
  G4R_REQUIRE(n >= 0 && n % 8 == 0, "relu_bwd: n must be a multiple of 8");
  if (n == 0) return G4R_OK;
  G4R_REQUIRE(y && dy && dx, "relu_bwd: null pointer");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const h16_t*)y,
                     (const h16_t*)dy, (h16_t*)dx, n / 8);
  G4R_CHECK_LAUNCH("relu_bwd");
  return G4R_OK;
}

int g4r_gather_rows_bf16(const void* src, const int* idx, void* dst, int n, int C, long ld_src, long ld_dst,
                         void* stream) {
  G4R_REQUIRE(n >= 0 && C > 0 && C % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, "gather_rows: bad shape");
  if (n == 0) return G4R_OK;
  G4R_REQUIRE(src && idx && dst, "gather_rows: null pointer");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)n * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)src, idx, (h16_t*)dst, n, C, ld_src, ld_dst);
  G4R_CHECK_LAUNCH("gather_rows");
  return G4R_OK;
}

int g4r_scatter_add_rows_f32(const void* src, const int* idx, float* out, int n, int C, long ld_src, long ld_out,
                             void* stream) {
  G4R_REQUIRE(n >= 0 && C > 0 && C % 8 == 0 && ld_src % 8 == 0, "scatter_add_rows: bad shape");
  if (n == 0) return G4R_OK;
  G4R_REQUIRE(src && idx && out, "scatter_add_rows: null pointer");
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for((long)n * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)src, idx, out, n, C, ld_src, ld_out);
  G4R_CHECK_LAUNCH("scatter_add_rows");
  return G4R_OK;
}

int g4r_adamw_f32(float* param, const void* grad, int grad_is_bf16, float* exp_avg, float* exp_avg_sq,
                  void* param_bf16, long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int step, float grad_scale, void* stream) {
  G4R_REQUIRE(n >= 0 && step >= 1, "adamw: bad shape / step");
  if (n == 0) return G4R_OK;
  G4R_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adamw: null pointer");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  if (grad_is_bf16)
    hipLaunchKernelGGL(adamw_kernel<h16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param,
                       (const h16_t*)grad, exp_avg, exp_avg_sq, (h16_t*)param_bf16, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2, grad_scale);
  else
    hipLaunchKernelGGL(adamw_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param,
                       (const float*)grad, exp_avg, exp_avg_sq, (h16_t*)param_bf16, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2, grad_scale);
  G4R_CHECK_LAUNCH("adamw");
  return G4R_OK;
}

}  // extern "C"
