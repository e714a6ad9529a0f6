// elementwise.hip -- HBM-bound glue kernels of the region-feature path (NHWC, bf16 storage,
// fp32 arithmetic, 16-byte accesses, one pass each).  Each kernel names the reference lines
// whose arithmetic it restates for MI355X.
#include "g4r_common.h"

namespace {

struct F8 {
  float v[8];
};
__device__ __forceinline__ F8 ld8(const h16_t* p) {
  const uint4v r = *reinterpret_cast<const uint4v*>(p);
  F8 o;
  o.v[0] = h16lo(r.x); o.v[1] = h16hi(r.x); o.v[2] = h16lo(r.y); o.v[3] = h16hi(r.y);
  o.v[4] = h16lo(r.z); o.v[5] = h16hi(r.z); o.v[6] = h16lo(r.w); o.v[7] = h16hi(r.w);
  return o;
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
  F8 o = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
  return o;
}

// torch.linspace(-1, 1, n)[i] in fp32 (symmetric evaluation, as ATen's linspace kernel does)
__device__ __forceinline__ float linspace_pm1(int i, int n) {
  if (n == 1) return -1.f;
  const float step = 2.f / (float)(n - 1);
  return i < n / 2 ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

// align_corners=True source coordinate (F.interpolate, bilinear): src = dst * (in-1)/(out-1)
struct Lerp {
  int i0, i1;
  float w1;  // weight of i1; weight of i0 = 1 - w1
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

// ---------------------------------------------------------------------------------------------
// a6 + first half of a7: bilinear (align_corners=True) upsample of one ViT level to H x W,
// concatenated with the two coordinate channels and zero padding up to Cpad channels.
//   MLVLROIQueryModule.forward  gpt4roi/models/layers.py:225-232
//   MLVLFuseModule.generate_coordinate / forward  layers.py:117-127, 183-189
// in : [B, Hin*Win (+ row offset), ldin] bf16 token-major (= NHWC);  out: [B, H, W, Cpad] bf16.
// The interpolation result is rounded to bf16 before the coordinate concat, as F.interpolate on
// a bf16 tensor does.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample_coord_kernel(const h16_t* __restrict__ in,
                                                             h16_t* __restrict__ out, int B, int Hin,
                                                             int Win, long in_batch_stride, int ldin,
                                                             int H, int W, int C, int Cpad) {
  const int nvec = Cpad >> 3;
  const long total = (long)B * H * W * nvec;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % nvec);
    const long pix = i / nvec;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int b = (int)(pix / ((long)W * H));
    F8 o;
    if (v * 8 < C) {
      const Lerp ly = lerp_ac(y, Hin, H), lx = lerp_ac(x, Win, W);
      const h16_t* base = in + (size_t)b * in_batch_stride + v * 8;
      const F8 p00 = ld8(base + ((size_t)ly.i0 * Win + lx.i0) * ldin);
      const F8 p01 = ld8(base + ((size_t)ly.i0 * Win + lx.i1) * ldin);
      const F8 p10 = ld8(base + ((size_t)ly.i1 * Win + lx.i0) * ldin);
      const F8 p11 = ld8(base + ((size_t)ly.i1 * Win + lx.i1) * ldin);
      const float wy1 = ly.w1, wy0 = 1.f - wy1, wx1 = lx.w1, wx0 = 1.f - wx1;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        o.v[k] = wy0 * (wx0 * p00.v[k] + wx1 * p01.v[k]) + wy1 * (wx0 * p10.v[k] + wx1 * p11.v[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = v * 8 + k;
        o.v[k] = c == C ? linspace_pm1(x, W) : (c == C + 1 ? linspace_pm1(y, H) : 0.f);
      }
    }
    st8(out + (size_t)pix * Cpad + v * 8, o);
  }
}

// ---------------------------------------------------------------------------------------------
// a7: one "_single_shuffle" input assembly for ONE target level (layers.py:152-180):
//   out[:, 0:R)        = own      [:, 0:R)
//   out[:, R:R+S)      = interp(top [:, R+S : R+2S))      (coarser neighbour, fp32 bilinear,
//   out[:, R+S:R+2S)   = interp(down[:, R   : R+S))        finer neighbour,   align_corners=True)
// with R = C/2, S = C/4.  Each source may carry a deferred GroupNorm+ReLU (ConvModule order conv
// -> GN -> ReLU, mmcv/cnn/bricks/conv_module.py:196-206): y = relu(a*x + s), a/s per (batch,
// channel) from g4r_groupnorm_affine_nhwc_bf16; null = source already final (round 0 feeds the
// biased 1x1-conv output directly, layers.py:191).
// All maps NHWC bf16; result rounded to bf16 once (the conv input cast under autocast).
// ---------------------------------------------------------------------------------------------
struct ShuffleSrc {
  const h16_t* x;
  const float* affine;  // [B, 2, C] or null
  int H, W;
};

// the bilinear blend of the shuffle kernels with its contractions pinned (wy0 (wx0 p00 + wx1 p01) + wy1 (wx0 p10 + wx1 p11)):
// the per-level and the merged kernel must round alike, whatever the compiler would fuse in either context
__device__ __forceinline__ float blend4(float wy0, float wy1, float wx0, float wx1, float p00, float p01, float p10,
                                        float p11) {
  const float t0 = __builtin_fmaf(wx1, p01, wx0 * p00);
  const float t1 = __builtin_fmaf(wx1, p11, wx0 * p10);
  return __builtin_fmaf(wy1, t1, wy0 * t0);
}

__device__ __forceinline__ F8 gn_relu(const F8& x, const float* aff, int b, int C, int c0) {
  if (!aff) return x;
  const F8 a = ld8f(aff + (size_t)b * 2 * C + c0);
  const F8 s = ld8f(aff + (size_t)b * 2 * C + C + c0);
  F8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o.v[k] = fmaxf(__builtin_fmaf(a.v[k], x.v[k], s.v[k]), 0.f);
  return o;
}

__device__ __forceinline__ F8 sample_src(const ShuffleSrc& s, int b, int y, int x, int H, int W, int C,
                                         int c0) {
  const h16_t* base = s.x + (size_t)b * s.H * s.W * C + c0;
  if (s.H == H && s.W == W)  // identity resize (level is its own neighbour at the ends)
    return gn_relu(ld8(base + ((size_t)y * W + x) * C), s.affine, b, C, c0);
  const Lerp ly = lerp_ac(y, s.H, H), lx = lerp_ac(x, s.W, W);
  const F8 p00 = gn_relu(ld8(base + ((size_t)ly.i0 * s.W + lx.i0) * C), s.affine, b, C, c0);
  const F8 p01 = gn_relu(ld8(base + ((size_t)ly.i0 * s.W + lx.i1) * C), s.affine, b, C, c0);
  const F8 p10 = gn_relu(ld8(base + ((size_t)ly.i1 * s.W + lx.i0) * C), s.affine, b, C, c0);
  const F8 p11 = gn_relu(ld8(base + ((size_t)ly.i1 * s.W + lx.i1) * C), s.affine, b, C, c0);
  const float wy1 = ly.w1, wy0 = 1.f - wy1, wx1 = lx.w1, wx0 = 1.f - wx1;
  F8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o.v[k] = blend4(wy0, wy1, wx0, wx1, p00.v[k], p01.v[k], p10.v[k], p11.v[k]);
  return o;
}

__device__ __forceinline__ void shuffle_item(const ShuffleSrc& own, const ShuffleSrc& top, const ShuffleSrc& down,
                                             h16_t* __restrict__ out, int C, long i) {
  const int H = own.H, W = own.W;
  const int nvec = C >> 3;
  const int R = C >> 1, S = C >> 2;
  const int v = (int)(i % nvec);
  const long pix = i / nvec;
  const int x = (int)(pix % W);
  const int y = (int)((pix / W) % H);
  const int b = (int)(pix / ((long)W * H));
  const int c = v * 8;
  F8 o;
  if (c < R)
    o = sample_src(own, b, y, x, H, W, C, c);
  else if (c < R + S)
    o = sample_src(top, b, y, x, H, W, C, c + S);   // top[:, R+S + (c-R)]
  else
    o = sample_src(down, b, y, x, H, W, C, c - S);  // down[:, R + (c-R-S)]
  st8(out + (size_t)pix * C + c, o);
}

__global__ __launch_bounds__(256) void fuse_shuffle_kernel(ShuffleSrc own, ShuffleSrc top, ShuffleSrc down,
                                                           h16_t* __restrict__ out, int B, int C) {
  const long total = (long)B * own.H * own.W * (C >> 3);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
    shuffle_item(own, top, down, out, C, i);
}

// The same assembly for EVERY target level of a fuse round in one launch (the four per-level launches of round 2 cost
// 4-57 us each: the small levels are pure launch latency).  Restructured for the memory system as well: a workgroup owns
// a run of pixels of one (level, image) and a thread ONE 8-channel vector for all of them, so its source (own / top /
// down) and the deferred GroupNorm affine of its channels are fixed -- loaded once instead of per corner per pixel (the
// first merged version spent 10 of its 12.5 loads per item on the affine table and ran at 2.3 TB/s) -- and the pixel
// coordinates cost one division per pixel instead of three per item.  Same expressions per element: bit-identical.
#define G4R_SHUFFLE_MAX_LEVELS 4
struct ShuffleLevels {
  ShuffleSrc own[G4R_SHUFFLE_MAX_LEVELS], top[G4R_SHUFFLE_MAX_LEVELS], down[G4R_SHUFFLE_MAX_LEVELS];
  h16_t* out[G4R_SHUFFLE_MAX_LEVELS];
  int chunks[G4R_SHUFFLE_MAX_LEVELS];    // pixel chunks per image on level l
  int blk_end[G4R_SHUFFLE_MAX_LEVELS];   // running workgroup count: level l owns [blk_end[l-1], blk_end[l]) = B * chunks[l]
  int n;
};

__device__ __forceinline__ F8 affine_relu(const F8& x, const F8& ga, const F8& gs, bool on) {
  if (!on) return x;
  F8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o.v[k] = fmaxf(__builtin_fmaf(ga.v[k], x.v[k], gs.v[k]), 0.f);
  return o;
}

__global__ __launch_bounds__(256) void fuse_shuffle_mlvl_kernel(ShuffleLevels a, int B, int C, int ppb) {
  int l = 0;
#pragma unroll
  for (int q = 0; q < G4R_SHUFFLE_MAX_LEVELS - 1; ++q)
    if (q + 1 < a.n && (int)blockIdx.x >= a.blk_end[q]) l = q + 1;
  const int local = (int)blockIdx.x - (l == 0 ? 0 : a.blk_end[l - 1]);
  const int chunks = a.chunks[l];
  const int b = local / chunks, chunk = local - b * chunks;
  const int nvec = C >> 3;
  const int v = threadIdx.x % nvec, plane = threadIdx.x / nvec, pstep = 256 / nvec;
  const int R = C >> 1, S = C >> 2;
  const int c = v * 8;
  // this thread's source and the channel it reads there
  ShuffleSrc src = a.own[l];
  int cs = c;
  if (c >= R + S) { src = a.down[l]; cs = c - S; }       // down[:, R + (c-R-S)]
  else if (c >= R) { src = a.top[l]; cs = c + S; }       // top[:, R+S + (c-R)]
  const int H = a.own[l].H, W = a.own[l].W;
  const bool aff = src.affine != nullptr;
  F8 ga, gs;
  if (aff) {
    ga = ld8f(src.affine + (size_t)b * 2 * C + cs);
    gs = ld8f(src.affine + (size_t)b * 2 * C + C + cs);
  }
  const h16_t* base = src.x + (size_t)b * src.H * src.W * C + cs;
  const bool same = src.H == H && src.W == W;
  h16_t* out = a.out[l] + (size_t)b * H * W * C + c;
  const int HW = H * W;
  int p1 = (chunk + 1) * ppb;
  if (p1 > HW) p1 = HW;
#pragma unroll 4
  for (int p = chunk * ppb + plane; p < p1; p += pstep) {       // independent pixels: let several gathers be in flight
    const int y = p / W, x = p - y * W;
    F8 o;
    if (same) {
      o = affine_relu(ld8(base + (size_t)p * C), ga, gs, aff);
    } else {
      const Lerp ly = lerp_ac(y, src.H, H), lx = lerp_ac(x, src.W, W);
      const F8 p00 = affine_relu(ld8(base + ((size_t)ly.i0 * src.W + lx.i0) * C), ga, gs, aff);
      const F8 p01 = affine_relu(ld8(base + ((size_t)ly.i0 * src.W + lx.i1) * C), ga, gs, aff);
      const F8 p10 = affine_relu(ld8(base + ((size_t)ly.i1 * src.W + lx.i0) * C), ga, gs, aff);
      const F8 p11 = affine_relu(ld8(base + ((size_t)ly.i1 * src.W + lx.i1) * C), ga, gs, aff);
      const float wy1 = ly.w1, wy0 = 1.f - wy1, wx1 = lx.w1, wx0 = 1.f - wx1;
#pragma unroll
      for (int k = 0; k < 8; ++k) o.v[k] = blend4(wy0, wy1, wx0, wx1, p00.v[k], p01.v[k], p10.v[k], p11.v[k]);
    }
    st8(out + (size_t)p * C, o);
  }
}

// ---------------------------------------------------------------------------------------------
// CLIP ViT front end (HF CLIPVisionEmbeddings: conv14/14 no bias -> cat(cls) -> + pos-emb).
// ---------------------------------------------------------------------------------------------
// img [B,3,S,S] float32 NCHW -> patches [B*P*P, Kpad] bf16, k = c*196 + ky*14 + kx (conv weight
// flatten order), zero padded to Kpad.
__global__ __launch_bounds__(256) void im2col_patch14_kernel(const float* __restrict__ img,
                                                             h16_t* __restrict__ out, int B, int S,
                                                             int P, int Kpad) {
  const long total = (long)B * P * P * Kpad;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int k = (int)(i % Kpad);
    const long t = i / Kpad;
    float v = 0.f;
    if (k < 588) {
      const int c = k / 196, r = k % 196, ky = r / 14, kx = r % 14;
      const int px = (int)(t % P), py = (int)((t / P) % P), b = (int)(t / ((long)P * P));
      v = img[(((size_t)b * 3 + c) * S + py * 14 + ky) * S + px * 14 + kx];
    }
    out[i] = f32_to_h16(v);
  }
}

// tokens[b, 0] = cls + pos[0]; tokens[b, 1+i] = patch[b*n+i] + pos[1+i]      (bf16, C % 8 == 0)
__global__ __launch_bounds__(256) void vit_assemble_kernel(const h16_t* __restrict__ patch,
                                                           const h16_t* __restrict__ cls,
                                                           const h16_t* __restrict__ pos,
                                                           h16_t* __restrict__ tok, int B, int n, int C) {
  const int nvec = C >> 3;
  const long total = (long)B * (n + 1) * nvec;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % nvec);
    const long row = i / nvec;
    const int s = (int)(row % (n + 1)), b = (int)(row / (n + 1));
    const F8 a = s == 0 ? ld8(cls + v * 8) : ld8(patch + ((size_t)b * n + s - 1) * C + v * 8);
    const F8 p = ld8(pos + (size_t)s * C + v * 8);
    F8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = h16_to_f32(f32_to_h16(a.v[k])) + p.v[k];
    st8(tok + row * C + v * 8, o);
  }
}

// ---------------------------------------------------------------------------------------------
// LLaMA glue (the arithmetic spi_llava.py:198-205 delegates to HF LlamaModel).
// ---------------------------------------------------------------------------------------------
// qkv [T, 3*Hh*D] bf16 (q | k | v) -> RoPE(q) in place layout qout [T, Hh*D]; RoPE(k) and v appended
// to the caches at rows pos0.. : kcache/vcache [maxT, Hh*D].  cos/sin [maxT, D/2] fp32.
// rotate_half convention: x' = x*cos + rot(x)*sin, rot(x) = cat(-x[D/2:], x[:D/2]).
__global__ __launch_bounds__(256) void rope_qkv_kernel(const h16_t* __restrict__ qkv,
                                                       const float* __restrict__ cs,
                                                       const float* __restrict__ sn,
                                                       h16_t* __restrict__ qout,
                                                       h16_t* __restrict__ kcache,
                                                       h16_t* __restrict__ vcache, int T, int Hh, int D,
                                                       int pos0, const int* __restrict__ pos_dev) {
  if (pos_dev) pos0 = *pos_dev;  // decode loop replayed from a hipGraph: the position lives on the device
  const int half = D >> 1;
  const int hv = half >> 3;  // 8-wide vectors per half head
  const long total = (long)T * Hh * hv;
  const int HD = Hh * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % hv);
    const int h = (int)((i / hv) % Hh);
    const int t = (int)(i / ((long)hv * Hh));
    const int pos = pos0 + t;
    const F8 c = ld8f(cs + (size_t)pos * half + v * 8);
    const F8 s = ld8f(sn + (size_t)pos * half + v * 8);
    const size_t off = (size_t)h * D + v * 8;
    const h16_t* row = qkv + (size_t)t * 3 * HD;
    {
      const F8 a = ld8(row + off), b = ld8(row + off + half);
      F8 o1, o2;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        o1.v[k] = __builtin_fmaf(a.v[k], c.v[k], -(b.v[k] * s.v[k]));   // pinned: the fused epilogue form
        o2.v[k] = __builtin_fmaf(b.v[k], c.v[k], a.v[k] * s.v[k]);      // (g4r_gemm_qkv_rope_bf16) rounds alike
      }
      st8(qout + (size_t)t * HD + off, o1);
      st8(qout + (size_t)t * HD + off + half, o2);
    }
    {
      const F8 a = ld8(row + HD + off), b = ld8(row + HD + off + half);
      F8 o1, o2;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        o1.v[k] = __builtin_fmaf(a.v[k], c.v[k], -(b.v[k] * s.v[k]));   // pinned: the fused epilogue form
        o2.v[k] = __builtin_fmaf(b.v[k], c.v[k], a.v[k] * s.v[k]);      // (g4r_gemm_qkv_rope_bf16) rounds alike
      }
      st8(kcache + (size_t)pos * HD + off, o1);
      st8(kcache + (size_t)pos * HD + off + half, o2);
    }
    *reinterpret_cast<uint4v*>(vcache + (size_t)pos * HD + off) =
        *reinterpret_cast<const uint4v*>(row + 2 * HD + off);
    *reinterpret_cast<uint4v*>(vcache + (size_t)pos * HD + off + half) =
        *reinterpret_cast<const uint4v*>(row + 2 * HD + off + half);
  }
}

// gu [T, 2*F] bf16 (gate | up) -> out [T, F] = silu(gate) * up
__global__ __launch_bounds__(256) void swiglu_kernel(const h16_t* __restrict__ gu, h16_t* __restrict__ out,
                                                     int T, int F) {
  const int nvec = F >> 3;
  const long total = (long)T * nvec;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % nvec);
    const long t = i / nvec;
    const F8 g = ld8(gu + t * 2 * F + v * 8), u = ld8(gu + t * 2 * F + F + v * 8);
    F8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float sg = h16_to_f32(f32_to_h16(g.v[k] / (1.f + __expf(-g.v[k]))));  // silu rounds to bf16
      o.v[k] = sg * u.v[k];
    }
    st8(out + t * F + v * 8, o);
  }
}

// ---------------------------------------------------------------------------------------------
// a15: embedding lookup + image-patch splice + <bbox> region-token injection in one gather
// (replaces the per-sample host loop of gpt4roi/models/spi_llava.py:99-196).
//   ids [B, T] int64.  For sample b: tokens equal to patch_id take consecutive rows of
//   img[b] ([B, n_patch, C]); tokens equal to bbox_id take consecutive rows of spi (rows
//   spi_offset[b] ..); everything else takes embed[id].  The reference requires the patch run to
//   sit right after <im_start> and be followed by <im_end> (:125-128) and #<bbox> == n_i (:149-157);
//   violations set status[b] != 0 instead of raising on the device.
// grid = (token chunks, samples): every workgroup redoes the cheap block-wide rank scan of its sample
// (T is ~1-2k ids) and then gathers `tok_per_block` token rows; chunk 0 writes status.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splice_embed_kernel(const long* __restrict__ ids,
                                                           const h16_t* __restrict__ embed,
                                                           const h16_t* __restrict__ img,
                                                           const h16_t* __restrict__ spi,
                                                           const int* __restrict__ spi_offset,
                                                           h16_t* __restrict__ out, int* __restrict__ status,
                                                           int T, int C, int n_patch, long patch_id,
                                                           long bbox_id, long im_start_id, long im_end_id,
                                                           int vocab, int tok_per_block) {
  extern __shared__ int sh[];  // [T] patch rank, [T] bbox rank, [8] scratch
  int* prank = sh;
  int* brank = sh + T;
  __shared__ int carry[2];
  __shared__ int wsum[2][4];
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long* row = ids + (size_t)b * T;
  if (tid == 0) carry[0] = carry[1] = 0;
  __syncthreads();
  for (int base = 0; base < T; base += 256) {
    const int t = base + tid;
    const long id = t < T ? row[t] : -1;
    const int fp = id == patch_id, fb = id == bbox_id;
    const unsigned long long mp = __ballot(fp), mb = __ballot(fb);
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int pp = __popcll(mp & below), pb = __popcll(mb & below);
    if (lane == 0) {
      wsum[0][wave] = __popcll(mp);
      wsum[1][wave] = __popcll(mb);
    }
    __syncthreads();
    int op = carry[0], ob = carry[1];
    for (int w = 0; w < wave; ++w) {
      op += wsum[0][w];
      ob += wsum[1][w];
    }
    if (t < T) {
      prank[t] = fp ? op + pp : -1;
      brank[t] = fb ? ob + pb : -1;
    }
    __syncthreads();
    if (tid == 0) {
      carry[0] += wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
      carry[1] += wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
    }
    __syncthreads();
  }
  // structural checks, in parallel over the tokens
  __shared__ int first_sh, flags_sh;
  if (tid == 0) {
    first_sh = -1;
    flags_sh = 0;
  }
  __syncthreads();
  for (int t = tid; t < T; t += 256)
    if (prank[t] == 0) first_sh = t;  // rank 0 occurs at most once
  __syncthreads();
  const int first = first_sh;
  for (int t = tid; t < T; t += 256)
    if (prank[t] >= 0 && t - prank[t] != first) atomicOr(&flags_sh, 16);  // patch run not contiguous
  __syncthreads();
  if (tid == 0 && blockIdx.x == 0) {
    int st = flags_sh;
    const int np = carry[0], nb = carry[1];
    if (np != 0 && np != n_patch) st |= 1;                       // wrong number of patch tokens
    const int want = spi_offset ? spi_offset[b + 1] - spi_offset[b] : 0;
    if (nb != want) st |= 2;                                       // #<bbox> != #regions
    if (np > 0) {
      // ids < 0: a checkpoint without mm_use_im_start_end -- the patch run stands alone (spi_llava.py:158-196)
      if (im_start_id >= 0 && (first < 1 || row[first - 1] != im_start_id)) st |= 4;    // <im_start> must precede
      if (im_end_id >= 0 && (first + n_patch >= T || row[first + n_patch] != im_end_id)) st |= 8;  // <im_end> must follow
    }
    status[b] = st;
  }
  const int nvec = C >> 3;
  const int so = spi_offset ? spi_offset[b] : 0;
  // rows of `spi` this sample owns: a prompt with MORE <bbox> tokens than regions must not read the next sample's
  // rows (or past the allocation); the surplus tokens take their embedding row and status bit 2 reports the
  // mismatch (the reference raises on the host instead, spi_llava.py:149-157)
  const int n_region = spi_offset ? spi_offset[b + 1] - so : 0;
  const int t0 = blockIdx.x * tok_per_block;
  const int t1 = min(T, t0 + tok_per_block);
  for (int i = t0 * nvec + tid; i < t1 * nvec; i += 256) {
    const int t = i / nvec, v = i % nvec;
    const h16_t* src;
    if (prank[t] >= 0 && prank[t] < n_patch)
      src = img + ((size_t)b * n_patch + prank[t]) * C;
    else if (brank[t] >= 0 && brank[t] < n_region && spi)
      src = spi + ((size_t)so + brank[t]) * C;
    else {
      long id = row[t];
      if (id < 0) id = 0;
      if (id >= vocab) id = vocab - 1;
      src = embed + (size_t)id * C;
    }
    *reinterpret_cast<uint4v*>(out + ((size_t)b * T + t) * C + v * 8) =
        *reinterpret_cast<const uint4v*>(src + v * 8);
  }
}

// greedy decode: argmax over a logits row (first maximum wins, like torch.argmax on ties at the
// lowest index).  One workgroup per row.
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ logits, long ld, int N,
                                                          long* __restrict__ out) {
  __shared__ float bv[256];
  __shared__ int bi[256];
  const float* row = logits + (size_t)blockIdx.x * ld;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = threadIdx.x; i < N; i += 256) {
    const float v = row[i];
    if (v > best || (v == best && i < idx)) {
      best = v;
      idx = i;
    }
  }
  bv[threadIdx.x] = best;
  bi[threadIdx.x] = idx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float v = bv[threadIdx.x + s];
      const int i = bi[threadIdx.x + s];
      if (v > bv[threadIdx.x] || (v == bv[threadIdx.x] && i < bi[threadIdx.x])) {
        bv[threadIdx.x] = v;
        bi[threadIdx.x] = i;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = bi[0];
}

// Device-side greedy step (generate(do_sample=False), llava.py:263-283 feeds only the last token):
// tok = argmax(logits[0, :N]); out_ids[*step] = tok; ++*step; ++*pos.  Everything stays on the GPU so
// the whole decode step can be replayed from a hipGraph; the host reads out_ids when it wants to.
// 1024 threads, 16-byte loads issued eight at a time (the 128 KB logits row comes from L2 in a couple of round trips;
// the first version's 256-thread scalar loop took 37 us per token).  Ties go to the lowest index, as torch.argmax.
__global__ __launch_bounds__(1024) void greedy_advance_kernel(const float* __restrict__ logits, int N,
                                                              long* __restrict__ tok, long* __restrict__ out_ids,
                                                              int* __restrict__ step, int* __restrict__ pos,
                                                              int max_steps) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  float best = -INFINITY;
  int idx = 0x7fffffff;
  auto take = [&](float v, int i) {
    if (v > best || (v == best && i < idx)) { best = v; idx = i; }
  };
  const int nvec = N >> 2;
  for (int v0 = threadIdx.x; v0 < nvec; v0 += 8 * 1024) {
    float4v r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int v = v0 + u * 1024;
      r[u] = *reinterpret_cast<const float4v*>(logits + 4 * (size_t)(v < nvec ? v : nvec - 1));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int v = v0 + u * 1024;
      if (v < nvec) { take(r[u].x, 4 * v); take(r[u].y, 4 * v + 1); take(r[u].z, 4 * v + 2); take(r[u].w, 4 * v + 3); }
    }
  }
  if (threadIdx.x == 0)
    for (int i = nvec * 4; i < N; ++i) take(logits[i], i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v = __shfl_xor(best, o);
    const int i = __shfl_xor(idx, o);
    take(v, i);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) take(bv[w], bi[w]);
    const int st = *step;
    tok[0] = idx;
    if (st < max_steps) out_ids[st] = idx;
    *step = st + 1;
    *pos = *pos + 1;
  }
}

// Batched form of the device-side greedy step: nxt [B] = the argmax of each sequence's logits row (g4r_argmax_rows_f32);
// tok / tok32 [B] receive it (int64 for the caller, int32 for the embedding gather), out_ids [B][max_steps] its slot
// *step; the shared counters advance once.  One block: nobody reads *step after it moved.
__global__ __launch_bounds__(64) void batch_advance_kernel(const long* __restrict__ nxt, int B, long* __restrict__ tok,
                                                           int* __restrict__ tok32, long* __restrict__ out_ids,
                                                           int* __restrict__ step, int* __restrict__ pos, int max_steps,
                                                           int npos) {
  const int st = *step;
  for (int b = threadIdx.x; b < B; b += 64) {
    const long t = nxt[b];
    tok[b] = t;
    tok32[b] = (int)t;
    if (st < max_steps) out_ids[(size_t)b * max_steps + st] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) *step = st + 1;
  for (int i = threadIdx.x; i < npos; i += 64) pos[i] += 1;      // one shared position, or one per sequence (+ the RoPE one)
}

// y[i] = a[i] + b[row(i) % brows]  (bf16; used for "+ pos_embedd" style adds), C % 8 == 0
__global__ __launch_bounds__(256) void add_rows_kernel(const h16_t* __restrict__ a,
                                                       const h16_t* __restrict__ b, h16_t* __restrict__ y,
                                                       long rows, int C, long brows) {
  const int nvec = C >> 3;
  const long total = rows * nvec;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % nvec);
    const long r = i / nvec;
    const F8 x = ld8(a + r * C + v * 8), z = ld8(b + (r % brows) * C + v * 8);
    F8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = x.v[k] + z.v[k];
    st8(y + r * C + v * 8, o);
  }
}

// float32 -> bf16 and bf16 -> float32 casts (weights / feature hand-over)
// torch conv weight [Co][Ci][3][3] fp32 -> a kernel layout in the 16-bit storage type, one pass (the region module's weights
// change every training step: rounds 1-3 re-derived both layouts with torch permute / flip / cat / cast passes):
//   transposed = 0: dst[co * ld + off + tap * Ci + ci] = w[co][ci][tap]          rows of the implicit-GEMM forward weight
//   transposed = 1: dst[(ci) * ld + off + tap * Co + co] = w[co][ci][8 - tap]    rows of the data-gradient weight: the conv
//                   with the 180-degree rotated, channel-transposed filter (what conv2d's backward applies to dY)
__global__ __launch_bounds__(256) void conv3x3_weight_layout_kernel(const float* __restrict__ w, h16_t* __restrict__ dst,
                                                                     int Co, int Ci, long ld, long off, int transposed) {
  const long n = (long)Co * Ci;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    long co, ci;
    if (transposed) { ci = i / Co; co = i - ci * Co; } else { co = i / Ci; ci = i - co * Ci; }   // fastest index = written dim
    const float* src = w + (co * Ci + ci) * 9;
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = src[t];
    if (transposed) {
      h16_t* o = dst + ci * ld + off + co;
#pragma unroll
      for (int t = 0; t < 9; ++t) o[(long)t * Co] = f32_to_h16(v[8 - t]);
    } else {
      h16_t* o = dst + co * ld + off + ci;
#pragma unroll
      for (int t = 0; t < 9; ++t) o[(long)t * Ci] = f32_to_h16(v[t]);
    }
  }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, h16_t* __restrict__ y,
                                                            long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = f32_to_h16(x[i]);
}

inline int grid_for(long work_items) {
  long b = (work_items + 255) / 256;
  if (b > 256L * 16) b = 256L * 16;
  if (b < 1) b = 1;
  return (int)b;
}

// ---------------------------------------------------------------------------------------------
// Image front end: uint8 HWC image -> normalised, bilinearly resized fp32 CHW tensor, the
// step right before the ViT.  Replaces, on the device, `image_processor.preprocess` + F.interpolate(size,
// mode='bilinear', align_corners=False) of gpt4roi/app.py:125-136 and the Resize/Normalize stages of the dataset
// pipelines (gpt4roi/datasets/refcoco.py:69-85).  Normalisation is affine and interpolation linear, so doing
// them in one pass equals normalise-then-resize.  PyTorch's half-pixel convention:
//   src = max(0, (dst + 0.5) * in/out - 0.5), i0 = floor(src), i1 = min(i0 + 1, in - 1).
// ---------------------------------------------------------------------------------------------
struct NormParams {
  float mean[3], inv_std[3];
};

__global__ __launch_bounds__(256) void image_preprocess_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                               int H, int W, long row_bytes, int bgr, int OH, int OW,
                                                               NormParams np) {
  const long total = (long)OH * OW;
  const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % OW), oy = (int)(i / OW);
    float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
    if (fy < 0.f) fy = 0.f;
    if (fx < 0.f) fx = 0.f;
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > H - 1) y0 = H - 1;
    if (x0 > W - 1) x0 = W - 1;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const uint8_t* r0 = src + (size_t)y0 * row_bytes;
    const uint8_t* r1 = src + (size_t)y1 * row_bytes;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int sc = bgr ? 2 - c : c;
      const float p00 = r0[x0 * 3 + sc], p01 = r0[x1 * 3 + sc], p10 = r1[x0 * 3 + sc], p11 = r1[x1 * 3 + sc];
      const float v = (1.f - ly) * ((1.f - lx) * p00 + lx * p01) + ly * ((1.f - lx) * p10 + lx * p11);
      dst[(size_t)c * total + i] = (v * (1.f / 255.f) - np.mean[c]) * np.inv_std[c];
    }
  }
}

}  // namespace

extern "C" {

int g4r_upsample_coord_nhwc_bf16(const void* in, void* out, int B, int Hin, int Win, long in_batch_stride,
                                 int ldin, int H, int W, int C, int Cpad, void* stream) {
  G4R_REQUIRE(B > 0 && Hin > 0 && Win > 0 && H > 0 && W > 0, "upsample_coord: bad shape");
  G4R_REQUIRE(C % 8 == 0 && Cpad % 8 == 0 && Cpad >= C + 2 && ldin % 8 == 0, "upsample_coord: channels");
  G4R_REQUIRE(in && out, "upsample_coord: null pointer");
  const long total = (long)B * H * W * (Cpad / 8);
  hipLaunchKernelGGL(upsample_coord_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)in, (h16_t*)out, B, Hin, Win, in_batch_stride, ldin, H, W, C, Cpad);
  G4R_CHECK_LAUNCH("upsample_coord");
  return G4R_OK;
}

/* n_levels (<= 4) target levels in one launch.  For target level t: own = maps[t], top = maps[top_idx[t]], down =
 * maps[down_idx[t]] (layers.py:108-112: the neighbour, or the level itself at the ends of the pyramid); maps[l] is NHWC
 * [B, heights[l], widths[l], C] bf16 with the optional deferred GroupNorm+ReLU affine affines[l] ([B, 2, C] fp32 or null;
 * `affines` itself may be null); outs[t] receives the assembled conv input of level t. */
int g4r_fuse_shuffle_mlvl_nhwc_bf16(const void* const* maps, const float* const* affines, const int* heights,
                                    const int* widths, const int* top_idx, const int* down_idx, void* const* outs,
                                    int n_levels, int B, int C, void* stream) {
  G4R_REQUIRE(n_levels >= 1 && n_levels <= G4R_SHUFFLE_MAX_LEVELS && B > 0, "fuse_shuffle_mlvl: 1..4 levels");
  G4R_REQUIRE(C % 32 == 0, "fuse_shuffle_mlvl: C must be a multiple of 32");
  G4R_REQUIRE(maps && heights && widths && top_idx && down_idx && outs, "fuse_shuffle_mlvl: null pointer");
  const int nvec = C / 8;
  G4R_REQUIRE(nvec <= 256 && 256 % nvec == 0, "fuse_shuffle_mlvl: C/8 must divide 256");
  ShuffleLevels a;
  const int ppb = 64;                  // pixels per workgroup: 64 x C x 2 B = 128 KB written per workgroup at C = 1024
  int blocks = 0;
  for (int t = 0; t < G4R_SHUFFLE_MAX_LEVELS; ++t) {
    const int l = t < n_levels ? t : 0;
    const int tp = top_idx[l], dn = down_idx[l];
    G4R_REQUIRE(tp >= 0 && tp < n_levels && dn >= 0 && dn < n_levels && maps[l] && outs[l] && heights[l] > 0 &&
                    widths[l] > 0, "fuse_shuffle_mlvl: bad level");
    a.own[t] = ShuffleSrc{(const h16_t*)maps[l], affines ? affines[l] : nullptr, heights[l], widths[l]};
    a.top[t] = ShuffleSrc{(const h16_t*)maps[tp], affines ? affines[tp] : nullptr, heights[tp], widths[tp]};
    a.down[t] = ShuffleSrc{(const h16_t*)maps[dn], affines ? affines[dn] : nullptr, heights[dn], widths[dn]};
    a.out[t] = (h16_t*)outs[l];
    a.chunks[t] = g4r_ceil_div((long)heights[l] * widths[l], ppb);
    if (t < n_levels) blocks += B * a.chunks[t];
    a.blk_end[t] = blocks;
  }
  a.n = n_levels;
  hipLaunchKernelGGL(fuse_shuffle_mlvl_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, B, C, ppb);
  G4R_CHECK_LAUNCH("fuse_shuffle_mlvl");
  return G4R_OK;
}

int g4r_fuse_shuffle_nhwc_bf16(const void* own, const float* own_affine, int H, int W, const void* top,
                               const float* top_affine, int Ht, int Wt, const void* down,
                               const float* down_affine, int Hd, int Wd, void* out, int B, int C,
                               void* stream) {
  G4R_REQUIRE(B > 0 && H > 0 && W > 0 && Ht > 0 && Wt > 0 && Hd > 0 && Wd > 0, "fuse_shuffle: bad shape");
  G4R_REQUIRE(C % 32 == 0, "fuse_shuffle: C must be a multiple of 32");
  G4R_REQUIRE(own && top && down && out, "fuse_shuffle: null pointer");
  ShuffleSrc so = {(const h16_t*)own, own_affine, H, W};
  ShuffleSrc st = {(const h16_t*)top, top_affine, Ht, Wt};
  ShuffleSrc sd = {(const h16_t*)down, down_affine, Hd, Wd};
  const long total = (long)B * H * W * (C / 8);
  hipLaunchKernelGGL(fuse_shuffle_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, so, st,
                     sd, (h16_t*)out, B, C);
  G4R_CHECK_LAUNCH("fuse_shuffle");
  return G4R_OK;
}

int g4r_im2col_patch14_f32(const float* img, void* out, int B, int S, int Kpad, void* stream) {
  G4R_REQUIRE(B > 0 && S > 0 && S % 14 == 0 && Kpad >= 588, "im2col_patch14: bad shape");
  G4R_REQUIRE(img && out, "im2col_patch14: null pointer");
  const int P = S / 14;
  const long total = (long)B * P * P * Kpad;
  hipLaunchKernelGGL(im2col_patch14_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img,
                     (h16_t*)out, B, S, P, Kpad);
  G4R_CHECK_LAUNCH("im2col_patch14");
  return G4R_OK;
}

int g4r_vit_assemble_bf16(const void* patch, const void* cls, const void* pos, void* tok, int B, int n, int C,
                          void* stream) {
  G4R_REQUIRE(B > 0 && n > 0 && C % 8 == 0, "vit_assemble: bad shape");
  G4R_REQUIRE(patch && cls && pos && tok, "vit_assemble: null pointer");
  const long total = (long)B * (n + 1) * (C / 8);
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)patch, (const h16_t*)cls, (const h16_t*)pos, (h16_t*)tok, B, n, C);
  G4R_CHECK_LAUNCH("vit_assemble");
  return G4R_OK;
}

int g4r_rope_qkv_bf16(const void* qkv, const float* cos_tab, const float* sin_tab, void* q_out, void* k_cache,
                      void* v_cache, int T, int heads, int head_dim, int pos0, const int* pos_dev, void* stream) {
  G4R_REQUIRE(T >= 0 && heads > 0 && head_dim % 16 == 0 && pos0 >= 0, "rope_qkv: bad shape");
  if (T == 0) return G4R_OK;
  G4R_REQUIRE(qkv && cos_tab && sin_tab && q_out && k_cache && v_cache, "rope_qkv: null pointer");
  const long total = (long)T * heads * (head_dim / 16);
  hipLaunchKernelGGL(rope_qkv_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)qkv, cos_tab, sin_tab, (h16_t*)q_out, (h16_t*)k_cache, (h16_t*)v_cache, T,
                     heads, head_dim, pos0, pos_dev);
  G4R_CHECK_LAUNCH("rope_qkv");
  return G4R_OK;
}

int g4r_swiglu_bf16(const void* gate_up, void* out, int T, int F, void* stream) {
  G4R_REQUIRE(T >= 0 && F > 0 && F % 8 == 0, "swiglu: bad shape");
  if (T == 0) return G4R_OK;
  G4R_REQUIRE(gate_up && out, "swiglu: null pointer");
  hipLaunchKernelGGL(swiglu_kernel, dim3(grid_for((long)T * (F / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)gate_up, (h16_t*)out, T, F);
  G4R_CHECK_LAUNCH("swiglu");
  return G4R_OK;
}

int g4r_splice_embed_bf16(const long* ids, const void* embed, const void* img, const void* spi,
                          const int* spi_offset, void* out, int* status, int B, int T, int C, int n_patch,
                          long patch_id, long bbox_id, long im_start_id, long im_end_id, int vocab,
                          void* stream) {
  G4R_REQUIRE(B > 0 && T > 0 && C % 8 == 0 && vocab > 0 && n_patch >= 0, "splice_embed: bad shape");
  G4R_REQUIRE(T <= 16384, "splice_embed: T <= 16384");
  G4R_REQUIRE(ids && embed && out && status, "splice_embed: null pointer");
  G4R_REQUIRE(n_patch == 0 || img, "splice_embed: image features missing");
  const int tpb = 4;
  hipLaunchKernelGGL(splice_embed_kernel, dim3(g4r_ceil_div(T, tpb), B), dim3(256), 2 * T * sizeof(int),
                     (hipStream_t)stream, ids, (const h16_t*)embed, (const h16_t*)img, (const h16_t*)spi,
                     spi_offset, (h16_t*)out, status, T, C, n_patch, patch_id, bbox_id, im_start_id, im_end_id,
                     vocab, tpb);
  G4R_CHECK_LAUNCH("splice_embed");
  return G4R_OK;
}

#ifndef G4R_F16   // fp32 in, integers out: one instantiation serves both storage types
int g4r_argmax_rows_f32(const float* logits, long ld, int rows, int N, long* out, void* stream) {
  G4R_REQUIRE(rows >= 0 && N > 0, "argmax_rows: bad shape");
  if (rows == 0) return G4R_OK;
  G4R_REQUIRE(logits && out, "argmax_rows: null pointer");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ld, N, out);
  G4R_CHECK_LAUNCH("argmax_rows");
  return G4R_OK;
}

int g4r_greedy_advance_f32(const float* logits, int N, long* tok, long* out_ids, int* step, int* pos,
                           int max_steps, void* stream) {
  G4R_REQUIRE(N > 0 && max_steps >= 0, "greedy_advance: bad shape");
  G4R_REQUIRE(logits && tok && out_ids && step && pos, "greedy_advance: null pointer");
  G4R_REQUIRE(((uintptr_t)logits & 15) == 0, "greedy_advance: logits must be 16-byte aligned");
  hipLaunchKernelGGL(greedy_advance_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, N, tok, out_ids,
                     step, pos, max_steps);
  G4R_CHECK_LAUNCH("greedy_advance");
  return G4R_OK;
}

int g4r_batch_advance(const long* nxt, int B, long* tok, int* tok32, long* out_ids, int* step, int* pos, int max_steps,
                      void* stream) {
  G4R_REQUIRE(B > 0 && max_steps >= 0, "batch_advance: bad shape");
  G4R_REQUIRE(nxt && tok && tok32 && out_ids && step && pos, "batch_advance: null pointer");
  hipLaunchKernelGGL(batch_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nxt, B, tok, tok32, out_ids, step,
                     pos, max_steps, 1);
  G4R_CHECK_LAUNCH("batch_advance");
  return G4R_OK;
}

// Same step for a ragged batch: `pos` holds npos counters (the B cache lengths and the shared RoPE position), all advanced.
int g4r_batch_advance_ragged(const long* nxt, int B, long* tok, int* tok32, long* out_ids, int* step, int* pos, int npos,
                             int max_steps, void* stream) {
  G4R_REQUIRE(B > 0 && max_steps >= 0 && npos >= 1, "batch_advance_ragged: bad shape");
  G4R_REQUIRE(nxt && tok && tok32 && out_ids && step && pos, "batch_advance_ragged: null pointer");
  hipLaunchKernelGGL(batch_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nxt, B, tok, tok32, out_ids, step,
                     pos, max_steps, npos);
  G4R_CHECK_LAUNCH("batch_advance_ragged");
  return G4R_OK;
}

#endif  // !G4R_F16

int g4r_add_rows_bf16(const void* a, const void* b, void* y, long rows, int C, long brows, void* stream) {
  G4R_REQUIRE(rows >= 0 && C % 8 == 0 && brows > 0, "add_rows: bad shape");
  if (rows == 0) return G4R_OK;
  G4R_REQUIRE(a && b && y, "add_rows: null pointer");
  hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)a, (const h16_t*)b, (h16_t*)y, rows, C, brows);
  G4R_CHECK_LAUNCH("add_rows");
  return G4R_OK;
}

int g4r_conv3x3_weight_layout_bf16(const float* w, void* dst, int Co, int Ci, long ld, long off, int transposed, void* stream) {
  G4R_REQUIRE(Co > 0 && Ci > 0 && off >= 0 && ld >= off + 9L * (transposed ? Co : Ci), "conv3x3_weight_layout: bad shape");
  G4R_REQUIRE(w && dst, "conv3x3_weight_layout: null pointer");
  hipLaunchKernelGGL(conv3x3_weight_layout_kernel, dim3(grid_for((long)Co * Ci)), dim3(256), 0, (hipStream_t)stream, w,
                     (h16_t*)dst, Co, Ci, ld, off, transposed);
  G4R_CHECK_LAUNCH("conv3x3_weight_layout");
  return G4R_OK;
}

int g4r_cast_f32_to_bf16(const float* x, void* y, long n, void* stream) {
  if (n <= 0) return G4R_OK;
  G4R_REQUIRE(x && y, "cast: null pointer");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x,
                     (h16_t*)y, n);
  G4R_CHECK_LAUNCH("cast_f32_bf16");
  return G4R_OK;
}

#ifndef G4R_F16   // u8 -> fp32: independent of the storage type
int g4r_image_preprocess_u8_f32(const void* image, int height, int width, long row_bytes, int bgr, float* out,
                                int out_h, int out_w, float mean_r, float mean_g, float mean_b, float std_r,
                                float std_g, float std_b, void* stream) {
  G4R_REQUIRE(height > 0 && width > 0 && out_h > 0 && out_w > 0 && row_bytes >= 3L * width, "image_preprocess: bad shape");
  G4R_REQUIRE(image && out, "image_preprocess: null pointer");
  G4R_REQUIRE(std_r > 0.f && std_g > 0.f && std_b > 0.f, "image_preprocess: std must be positive");
  NormParams np = {{mean_r, mean_g, mean_b}, {1.f / std_r, 1.f / std_g, 1.f / std_b}};
  hipLaunchKernelGGL(image_preprocess_kernel, dim3(grid_for((long)out_h * out_w)), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)image, out, height, width, row_bytes, bgr, out_h, out_w, np);
  G4R_CHECK_LAUNCH("image_preprocess");
  return G4R_OK;
}
#endif  // !G4R_F16

}  // extern "C"
