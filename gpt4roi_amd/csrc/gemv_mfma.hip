// gemv_mfma.hip -- the projections of a BATCHED decode step (2..16 sequences, one new token each) as a weight-streaming kernel
// whose arithmetic runs on the matrix pipe (round 5).
//
//   C [B, N] = act(W [N, K] . h_b + bias) + residual_b,   h_b = gamma ? bf16(bf16(x_b * rsqrt(mean(x_b^2) + eps)) * gamma) : x_b
//
// What HF `generate()` computes per token for a batch of prompts (gpt4roi/app.py:293-300; SURVEY.md 8d config 5 decodes 8
// requests together): LlamaRMSNorm + nn.Linear on [B, 1, K].  The step is HBM-bound (every 16-bit weight once per token for the
// whole batch), so the kernel is built like the single-row GEMV (gemm_bf16.hip, gemv_bf16_kernel) and NOT like a GEMM tile:
//   * a workgroup owns 16 consecutive output rows n; its 8 waves split K in interleaved steps of 64 (two or four row blocks per
//     workgroup -- fewer, fatter workgroups amortising the staged rows -- measured 1.4-3x slower: the kernel lives on the number of
//     workgroups streaming);
//   * the weights never touch LDS: a lane loads 16 bytes of W straight into the register layout of the `A` operand of
//     v_mfma_f32_16x16x32 (lane l: row n0 + l % 16, k-chunk l / 16), non-temporal, U steps (2 loads each) in flight per lane;
//   * the B activation rows are staged ONCE per workgroup in LDS (row stride K * 2 + 16 bytes: the 16 lanes of a read group hit 16
//     different 16-byte slots), the RMSNorm in front fused in with rmsnorm_bf16_kernel's own element -> thread map and summation
//     order (bit-identical normalised rows); a lane reads the `B` operand (batch row l % 16, k-chunk l / 16) with one
//     ds_read_b128, lanes of batch rows >= B feed zeros;
//   * one MFMA per 1 KiB of weights: 16 matrix-pipe cycles per wave against >= 60 cycles of HBM time for the same bytes, so the
//     arithmetic is free (the packed-dot VALU form of the same kernel was VALU-bound from B = 4 on:
//     profiles/r05_batched_gemv_negative.txt);
//   * the waves' 16 x 16 partial results meet in LDS, wave 0 finishes (bias -> activation -> residual -> one rounding; SwiGLU
//     over interleaved (gate, up) rows) and stores 4 consecutive n per lane.
#include "g4r_common.h"

namespace {

struct GemvMfmaArgs {
  const h16_t* x; long ldx;
  const float* gamma; float eps;
  const h16_t* W; int ldw;
  void* C; long ldc;
  const float* bias;
  const h16_t* residual; long ldr;
  int B, N, K, act, out_f32;
  int KC;                       // K elements staged per pass (multiple of 64); one pass when the rows fit the LDS budget
  int early;                    // the first weight block issued before the staging is waited for: 0 never, 1 without a fused norm, 2 always
};

__device__ __forceinline__ float gm_act(float v, int act) {
  if (act == 1) return v > 0.f ? v : 0.f;
  if (act == 2) return v / (1.f + __expf(-1.702f * v));
  if (act == 3) return v / (1.f + __expf(-v));
  return v;
}

template <int NWV, int U, bool NORM, int RB, bool EARLY, int RPH>
__global__ __launch_bounds__(NWV * 64) void gemv_mfma_kernel(GemvMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) char gm_smem[];
  __shared__ float red[4];
  __shared__ float rstd_s[16];
  __shared__ __attribute__((aligned(16))) float part[NWV][RB][64][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K, B = a.B;
  const int rowb = a.KC * 2 + 16;                      // bytes per staged row
  const int n0 = blockIdx.x * 16 * RB;                 // RB blocks of 16 output rows per workgroup (the staged rows serve them all)
  const int fr = lane & 15, fq = lane >> 4;            // MFMA operand coordinates: row / batch column, k-chunk of 8
  const h16_t* wrow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    int wn = n0 + rb * 16 + fr;
    if (wn > a.N - 1) wn = a.N - 1;
    wrow[rb] = a.W + (size_t)wn * a.ldw + fq * 8;
  }
  const bool col_live = fr < B;
  const char* xrow = gm_smem + (col_live ? fr : 0) * rowb + fq * 16;

  // the first block of this wave's weight stream (steps wave, wave + NWV, ... of the first pass): issued right AFTER the loads of
  // the wave's share of the staging and BEFORE it waits for them -- vector memory returns in order, so the (older) x loads are not
  // delayed, and the stream runs from the first cycles of the workgroup instead of after the staging's round trips and barriers
  // (round 6; the single-row GEMV does the same).  early = 0: debug, tools/decode_bench.py --ab
  uint4v w0[U][RB], w1[U][RB];
  auto load_w = [&](int c0, int s0, int nsteps) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int s = s0 + u * NWV;
      if (s > nsteps - 1) s = nsteps - 1;             // (clamped: the duplicate is not accumulated)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const h16_t* p = wrow[rb] + c0 + s * 64;
        w0[u][rb] = __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(p));
        w1[u][rb] = __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(p + 32));
      }
    }
  };
  const int nsteps0 = (K < a.KC ? K : a.KC) >> 6;
  bool pre = false;
  const int nvec = K >> 3;
  // one-pass fused norm with at most 4 rows per 256 threads (B <= 8 at eight waves, K <= 4096): the 256 threads of "half" h take
  // rows h, h + NH, h + 2 NH, ... -- each with rmsnorm_bf16_kernel's own element -> thread map (v = t + 256 i) and summation
  // order, so rstd has the bits of the separate launch -- ALL rows in ONE memory round trip (the general form below walks the rows
  // one round trip + two barriers each), and a thread normalises and stages exactly the vectors it summed: x is read once.
  constexpr int NH = NWV / 4;
  const bool fast = NORM && NH >= 1 && a.KC >= K && nvec <= 512 && B <= RPH * NH;   // RPH rows per 256 threads (2 or 4: registers)
  bool staged = false;
  if (NORM && fast) {
    __shared__ float red2[16][4];
    const int half = tid >> 8, t = tid & 255;
    uint4v xr[RPH][2];
    float4v gq[2][2];
#pragma unroll
    for (int j = 0; j < RPH; ++j) {
      const int b = half + NH * j;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = t + 256 * i;
        if (b < B && v < nvec) xr[j][i] = *reinterpret_cast<const uint4v*>(a.x + (size_t)b * a.ldx + (size_t)v * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = t + 256 * i;
      if (v < nvec) {
        gq[i][0] = *reinterpret_cast<const float4v*>(a.gamma + v * 8);
        gq[i][1] = *reinterpret_cast<const float4v*>(a.gamma + v * 8 + 4);
      }
    }
    if (EARLY) { load_w(0, wave, nsteps0); pre = true; }
    float s2[RPH];
#pragma unroll
    for (int j = 0; j < RPH; ++j) s2[j] = 0.f;
#pragma unroll
    for (int j = 0; j < RPH; ++j) {
      const int b = half + NH * j;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = t + 256 * i;
        if (b < B && v < nvec) {
          const uint4v r4 = xr[j][i];
          const float f[8] = {h16lo(r4.x), h16hi(r4.x), h16lo(r4.y), h16hi(r4.y), h16lo(r4.z), h16hi(r4.z), h16lo(r4.w), h16hi(r4.w)};
#pragma unroll
          for (int k = 0; k < 8; ++k) s2[j] += f[k] * f[k];
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s2[j] += __shfl_xor(s2[j], o);
      if (lane == 0 && b < B) red2[b][wave & 3] = s2[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPH; ++j) {
      const int b = half + NH * j;
      if (b < B) {
        const float rs = rsqrtf((red2[b][0] + red2[b][1] + red2[b][2] + red2[b][3]) / (float)K + a.eps);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int v = t + 256 * i;
          if (v < nvec) {
            uint4v r4 = xr[j][i];
            const float4v g0 = gq[i][0], g1 = gq[i][1];
            const float f[8] = {h16lo(r4.x), h16hi(r4.x), h16lo(r4.y), h16hi(r4.y), h16lo(r4.z), h16hi(r4.z), h16lo(r4.w), h16hi(r4.w)};
            const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = h16_to_f32(f32_to_h16(f[k] * rs)) * g[k];
            r4.x = pack_h16x2(o[0], o[1]); r4.y = pack_h16x2(o[2], o[3]);
            r4.z = pack_h16x2(o[4], o[5]); r4.w = pack_h16x2(o[6], o[7]);
            *reinterpret_cast<uint4v*>(gm_smem + b * rowb + v * 16) = r4;
          }
        }
      }
    }
    staged = true;
  } else if (NORM) {
    // rstd of every row, exactly as rmsnorm_bf16_kernel's 256 threads sum it (norm.hip): v = tid + 256 i, xor-shuffle, red[0..3]
    for (int b = 0; b < B; ++b) {
      float s2 = 0.f;
      if (tid < 256) {
        for (int v = tid; v < nvec; v += 256) {
          const uint4v r4 = *reinterpret_cast<const uint4v*>(a.x + (size_t)b * a.ldx + (size_t)v * 8);
          const float f[8] = {h16lo(r4.x), h16hi(r4.x), h16lo(r4.y), h16hi(r4.y), h16lo(r4.z), h16hi(r4.z), h16lo(r4.w), h16hi(r4.w)};
#pragma unroll
          for (int k = 0; k < 8; ++k) s2 += f[k] * f[k];
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
      __syncthreads();
      if (lane == 0 && tid < 256) red[wave] = s2;
      __syncthreads();
      if (tid == 0) rstd_s[b] = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + a.eps);
    }
    __syncthreads();
  }

  float4v acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] = float4v{0.f, 0.f, 0.f, 0.f};
  const int nsteps_total = K >> 6;                     // steps of 64 of K
  for (int c0 = 0; c0 < K; c0 += a.KC) {
    const int cn = K - c0 < a.KC ? K - c0 : a.KC;      // elements of this pass
    // ---- stage the B rows of this pass ----
    const int cvec = cn >> 3;
    if (!NORM && EARLY && c0 == 0) {
      // first pass without a norm: the thread's vector of the first four rows waits in registers while the first weight block is issued
      constexpr int XP = 4, NT = NWV * 64;
      uint4v xr[XP];
#pragma unroll
      for (int j = 0; j < XP; ++j)
        if (j < B && tid < cvec) xr[j] = *reinterpret_cast<const uint4v*>(a.x + (size_t)j * a.ldx + (size_t)tid * 8);
      load_w(0, wave, nsteps0);
      pre = true;
#pragma unroll
      for (int j = 0; j < XP; ++j)
        if (j < B && tid < cvec) *reinterpret_cast<uint4v*>(gm_smem + j * rowb + tid * 16) = xr[j];
      for (int b = 0; b < B; ++b)
        for (int v = tid + (b < XP ? NT : 0); v < cvec; v += NT)
          *reinterpret_cast<uint4v*>(gm_smem + b * rowb + v * 16) = *reinterpret_cast<const uint4v*>(a.x + (size_t)b * a.ldx + (size_t)v * 8);
    } else if (!staged) {
    for (int i = tid; i < B * cvec; i += NWV * 64) {
      const int b = i / cvec, v = i - b * cvec;
      uint4v r4 = *reinterpret_cast<const uint4v*>(a.x + (size_t)b * a.ldx + (size_t)(c0 + v * 8));
      if (NORM) {
        const float rs = rstd_s[b];
        const float4v g0 = *reinterpret_cast<const float4v*>(a.gamma + c0 + v * 8);
        const float4v g1 = *reinterpret_cast<const float4v*>(a.gamma + c0 + v * 8 + 4);
        const float f[8] = {h16lo(r4.x), h16hi(r4.x), h16lo(r4.y), h16hi(r4.y), h16lo(r4.z), h16hi(r4.z), h16lo(r4.w), h16hi(r4.w)};
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = h16_to_f32(f32_to_h16(f[k] * rs)) * g[k];
        r4.x = pack_h16x2(o[0], o[1]); r4.y = pack_h16x2(o[2], o[3]);
        r4.z = pack_h16x2(o[4], o[5]); r4.w = pack_h16x2(o[6], o[7]);
      }
      *reinterpret_cast<uint4v*>(gm_smem + b * rowb + v * 16) = r4;
    }
    }
    __syncthreads();
    // ---- stream this wave's steps of the pass: step s covers K [c0 + 64 s, + 64), wave w takes s = w, w + NWV, ... ----
    const int nsteps = cn >> 6;
    auto consume = [&](int s0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int s = s0 + u * NWV;
        if (s < nsteps) {
          uint4v x0 = *reinterpret_cast<const uint4v*>(xrow + s * 128);
          uint4v x1 = *reinterpret_cast<const uint4v*>(xrow + s * 128 + 64);
          if (!col_live) {
            x0 = uint4v{0u, 0u, 0u, 0u};
            x1 = uint4v{0u, 0u, 0u, 0u};
          }
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            acc[rb] = G4R_MFMA_16X16X32(__builtin_bit_cast(h16x8, w0[u][rb]), __builtin_bit_cast(h16x8, x0), acc[rb], 0, 0, 0);
            acc[rb] = G4R_MFMA_16X16X32(__builtin_bit_cast(h16x8, w1[u][rb]), __builtin_bit_cast(h16x8, x1), acc[rb], 0, 0, 0);
          }
        }
      }
    };
    int s0 = wave;
    if (EARLY && pre && c0 == 0) {                      // the block that has been in flight since before the staging
      consume(s0);
      s0 += NWV * U;
    }
    for (; s0 < nsteps; s0 += NWV * U) {
      load_w(c0, s0, nsteps);
      consume(s0);
    }
    if (c0 + a.KC < K) __syncthreads();                 // the next pass overwrites the staged rows
  }
  (void)nsteps_total;
  // ---- the waves' partial 16 x 16 results meet in LDS ----
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) *reinterpret_cast<float4v*>(&part[wave][rb][lane][0]) = acc[rb];
  __syncthreads();
  if (wave >= RB) return;                               // wave rb finishes row block rb
  const int rbw = wave;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < NWV; ++w) {
    const float4v p4 = *reinterpret_cast<const float4v*>(&part[w][rbw][lane][0]);
    v[0] += p4.x; v[1] += p4.y; v[2] += p4.z; v[3] += p4.w;
  }
  // D layout of v_mfma_f32_16x16x32: lane holds rows n0 + 4 (l / 16) + i, column (batch row) l % 16
  const int b = fr, n = n0 + rbw * 16 + fq * 4;
  if (b >= B || n >= a.N) return;
  if (a.act == 4) {                                      // (gate, up) row pairs -> N / 2 outputs
    h16_t* dst = reinterpret_cast<h16_t*>(a.C) + (size_t)b * a.ldc + (n >> 1);
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
      if (n + i + 1 < a.N) {
        const float g = v[i], u = v[i + 1];
        const float sg = h16lo(pack_h16x2(g / (1.f + __expf(-g)), 0.f));
        dst[i >> 1] = f32_to_h16(sg * u);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (n + i >= a.N) break;
    float y = v[i];
    if (a.bias) y += a.bias[n + i];
    y = gm_act(y, a.act);
    if (a.residual) y += h16_to_f32(a.residual[(size_t)b * a.ldr + n + i]);
    if (a.out_f32)
      reinterpret_cast<float*>(a.C)[(size_t)b * a.ldc + n + i] = y;
    else
      reinterpret_cast<h16_t*>(a.C)[(size_t)b * a.ldc + n + i] = f32_to_h16(y);
  }
}

template <int NWV, int U, bool NORM, int RB, bool EARLY, int RPH>
int gm_launch_k(GemvMfmaArgs& a, hipStream_t stream) {
  auto kern = gemv_mfma_kernel<NWV, U, NORM, RB, EARLY, RPH>;
  const size_t lds = (size_t)a.B * (a.KC * 2 + 16);
  static G4rPerDeviceOnce attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    if (e != hipSuccess) { attr_set.failed(); return g4r_note_hip_error(e, "gemv_mfma: hipFuncSetAttribute"); }
  }
  hipLaunchKernelGGL(kern, dim3(g4r_ceil_div(a.N, 16 * RB)), dim3(NWV * 64), lds, stream, a);
  G4R_CHECK_LAUNCH("gemv_mfma");
  return G4R_OK;
}

// the fused norm's one-round-trip form holds RPH rows per 256 threads in registers: 2 up to B = NWV / 2 rows, else 4
template <int NWV, int U, bool NORM, int RB = 1>
int gm_launch(GemvMfmaArgs& a, hipStream_t stream) {
  const bool few = !NORM || a.B <= 2 * (NWV / 4);
  // measured inside the batched decode step (profiles/r06_decode_batch_ab.txt): the early block pays on the launches WITHOUT a
  // norm (o_proj); with the norm fused its registers (88-106 against 52-70) cost a resident workgroup per CU and it loses
  const bool early = a.early == 2 || (a.early == 1 && !NORM);
  if (early) return few ? gm_launch_k<NWV, U, NORM, RB, true, 2>(a, stream) : gm_launch_k<NWV, U, NORM, RB, true, 4>(a, stream);
  return few ? gm_launch_k<NWV, U, NORM, RB, false, 2>(a, stream) : gm_launch_k<NWV, U, NORM, RB, false, 4>(a, stream);
}

}  // namespace

extern "C" {

// See include/g4r_kernels.h.  variant (tools only; 0 = production = 8 waves x 4 steps in flight; profiles/r05_gemv_batch.txt):
// 2 = 4 waves x 8 steps, 3 = two 16-row blocks per workgroup (fewer, fatter workgroups: slower), 5 = 4 waves x 4 steps, 6 / 7 = production
// without / with the early weight loads on every launch (the A/B of round 6; production: only without a fused norm), >= 16 = the LDS budget of the staged rows in KB
int g4r_gemv_batch_bf16(const void* x, int B, long ldx, const float* gamma, float eps, const void* W, void* C, long ldc,
                        const float* bias, const void* residual, long ldr, int N, int K, int ldw, int act, int out_f32,
                        int variant, void* stream) {
  G4R_REQUIRE(B >= 2 && B <= 16, "gemv_batch: 2 <= B <= 16 rows (one row: g4r_gemv_rmsnorm_bf16; more: g4r_gemm_bf16_nt)");
  G4R_REQUIRE(N > 0 && K >= 512 && K % 64 == 0 && ldw % 8 == 0 && ldw >= K && ldx % 8 == 0 && ldx >= K,
              "gemv_batch: K >= 512 and a multiple of 64, ldw / ldx multiples of 8");
  G4R_REQUIRE(x && W && C, "gemv_batch: null pointer");
  G4R_REQUIRE(act >= 0 && act <= 4, "gemv_batch: act must be 0..4");
  G4R_REQUIRE(act != 4 || (N % 4 == 0 && !residual && !bias && !out_f32), "gemv_batch: swiglu needs N % 4 == 0, 16-bit output");
  GemvMfmaArgs a = {};
  a.x = (const h16_t*)x; a.ldx = ldx; a.gamma = gamma; a.eps = eps; a.W = (const h16_t*)W; a.ldw = ldw; a.C = C; a.ldc = ldc;
  a.bias = bias; a.residual = (const h16_t*)residual; a.ldr = ldr; a.B = B; a.N = N; a.K = K; a.act = act; a.out_f32 = out_f32;
  // staged rows: B x (KC x 2 + 16) bytes within an LDS budget that leaves several workgroups per CU (the kernel lives on loads
  // in flight: at 82 KB -- 16 rows x 2560 -- one workgroup per CU ran at half the rate of four); equal passes, KC a multiple of
  // 64 x 8 (whole rounds of the eight-wave form).  variant >= 16: the budget in KB (tools)
  a.early = variant == 6 ? 0 : variant == 7 ? 2 : 1;
  if (variant == 6 || variant == 7) variant = 0;
  const int budget = variant >= 16 ? variant * 1024 : 98304;     // (measured: fewer passes beat more workgroups per CU)
  if (variant >= 16) variant = 1;
  int kc = ((budget / B - 16) / 2) / 512 * 512;
  if (kc < 512) kc = 512;
  const int passes = g4r_ceil_div(K, kc);
  kc = g4r_ceil_div(g4r_ceil_div(K, passes), 512) * 512;
  a.KC = kc >= K ? K : kc;
  hipStream_t st = (hipStream_t)stream;
  if (gamma) {
    if (variant == 1) return gm_launch<8, 4, true>(a, st);
    if (variant == 2) return gm_launch<4, 8, true>(a, st);
    if (variant == 3) return gm_launch<8, 4, true, 2>(a, st);
    if (variant == 5) return gm_launch<4, 4, true>(a, st);
    return gm_launch<8, 4, true>(a, st);
  }
  if (variant == 1) return gm_launch<8, 4, false>(a, st);
  if (variant == 2) return gm_launch<4, 8, false>(a, st);
  if (variant == 3) return gm_launch<8, 4, false, 2>(a, st);
  if (variant == 5) return gm_launch<4, 4, false>(a, st);
  return gm_launch<8, 4, false>(a, st);
}

}  // extern "C"
