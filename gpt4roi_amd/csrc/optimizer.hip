// optimizer.hip -- multi-tensor global gradient norm + AdamW for the training rows (gfx950).
//
// What the reference runs per step under HF Trainer (gpt4roi/train/train.py:698-712; llava_trainer.py:59-162):
// `clip_grad_norm_(parameters, 1.0)` then `torch.optim.AdamW.step()` -- a foreach over every trainable tensor (45 for the
// region module, ~340 when the 7B decoder trains in stage 2).  Round 1 launched one AdamW kernel per tensor and read the
// norm back to the host (16 % of the stage-2 step in a Python loop).  Here the whole update is TWO launches over a table of
// tensors, and the clip coefficient never leaves the device:
//   g4r_multi_sumsq      : per-chunk sums of grad^2 (fp32 reads, fp64 partials) -> one fp64 total (fixed reduction order:
//                          bit-reproducible for a given table)
//   g4r_multi_adamw_f32  : p, m, v (fp32) updated from grad * min(1, max_norm / (sqrt(total) + 1e-6)); the bf16 copy the
//                          kernels read is written in the same pass where the table names one
// HBM-bound: 16 B read + 12 B (+2 B) written per parameter; one workgroup streams one 4096-element chunk with 16-byte
// accesses.  The table (pointers, sizes, chunk prefix) lives in device memory and is re-uploaded only when a pointer changes.
#include "g4r_common.h"

namespace {

constexpr int CHUNK = 4096;  // elements per workgroup (256 threads x 16)

struct MultiArgs {
  const unsigned long long* p;      // [n] fp32 master pointers
  const unsigned long long* g;      // [n] gradient pointers
  const unsigned long long* m;      // [n] exp_avg
  const unsigned long long* v;      // [n] exp_avg_sq
  const unsigned long long* pb;     // [n] bf16 copy pointers (0 = none)
  const long* numel;                // [n]
  const int* g_is_bf16;             // [n]
  const int* chunk_start;           // [n + 1] prefix sum of ceil(numel / CHUNK)
  int n;
};

__device__ __forceinline__ int find_tensor(const int* __restrict__ chunk_start, int n, int b) {
  int lo = 0, hi = n;  // chunk_start[lo] <= b < chunk_start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_start[mid] <= b) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void multi_sumsq_kernel(MultiArgs a, double* __restrict__ partial) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  const int t = find_tensor(a.chunk_start, a.n, b);
  const long off = (long)(b - a.chunk_start[t]) * CHUNK;
  const long n = a.numel[t];
  double s = 0.0;
  if (a.g_is_bf16[t]) {
    const h16_t* g = reinterpret_cast<const h16_t*>(a.g[t]) + off;
    for (long i = threadIdx.x; i < CHUNK && off + i < n; i += 256) {
      const float x = h16_to_f32(g[i]);
      s += (double)x * x;
    }
  } else {
    const float* g = reinterpret_cast<const float*>(a.g[t]) + off;
    const long lim = (n - off < CHUNK ? n - off : CHUNK);
    if ((lim & 3) == 0 && ((a.g[t] + off * 4) & 15) == 0) {
      for (long i = threadIdx.x * 4; i < lim; i += 1024) {
        const float4v x = *reinterpret_cast<const float4v*>(g + i);
        s += (double)x.x * x.x + (double)x.y * x.y + (double)x.z * x.z + (double)x.w * x.w;
      }
    } else {
      for (long i = threadIdx.x; i < lim; i += 256) s += (double)g[i] * g[i];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[b] = red[0] + red[1] + red[2] + red[3];
}

// one workgroup: total = sum of the partials in index order (fixed order -> reproducible)
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const double* __restrict__ partial, int n, double* __restrict__ total) {
  __shared__ double red[256];
  double s = 0.0;
  const int per = (n + 255) / 256;
  const int i0 = threadIdx.x * per, i1 = min(n, i0 + per);
  for (int i = i0; i < i1; ++i) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += red[i];
    total[0] = t;
  }
}

__global__ __launch_bounds__(256) void multi_adamw_kernel(MultiArgs a, const double* __restrict__ total_sq, float max_norm,
                                                          float pre_scale, float lr, float b1, float b2, float eps, float wd,
                                                          float bc1, float bc2) {
  const int b = blockIdx.x;
  const int t = find_tensor(a.chunk_start, a.n, b);
  const long off = (long)(b - a.chunk_start[t]) * CHUNK;
  const long n = a.numel[t];
  // torch.nn.utils.clip_grad_norm_: coefficient max_norm / (total_norm + 1e-6), clamped to 1
  float gscale = pre_scale;
  if (max_norm > 0.f && total_sq) {
    const float norm = (float)sqrt(total_sq[0]) * pre_scale;
    const float c = max_norm / (norm + 1e-6f);
    gscale *= c < 1.f ? c : 1.f;
  }
  float* p = reinterpret_cast<float*>(a.p[t]) + off;
  float* m = reinterpret_cast<float*>(a.m[t]) + off;
  float* v = reinterpret_cast<float*>(a.v[t]) + off;
  h16_t* pb = a.pb[t] ? reinterpret_cast<h16_t*>(a.pb[t]) + off : nullptr;
  const bool gb = a.g_is_bf16[t] != 0;
  const h16_t* g16 = reinterpret_cast<const h16_t*>(a.g[t]) + off;
  const float* g32 = reinterpret_cast<const float*>(a.g[t]) + off;
  const long lim = (n - off < CHUNK ? n - off : CHUNK);
  const float decay = 1.f - lr * wd, ib1 = 1.f / bc1, ib2 = 1.f / bc2;
  for (long i = threadIdx.x; i < lim; i += 256) {
    const float gi = (gb ? h16_to_f32(g16[i]) : g32[i]) * gscale;
    float pi = p[i] * decay;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    pi -= lr * (mi * ib1) / (sqrtf(vi * ib2) + eps);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (pb) pb[i] = f32_to_h16(pi);
  }
}

}  // namespace

extern "C" {

// table: 5 pointer arrays + numel + dtype flags + chunk prefix, all in DEVICE memory (see include/g4r_train.h)
int g4r_multi_sumsq(const void* g_ptrs, const long* numel, const int* g_is_bf16, const int* chunk_start, int n_tensors,
                    int n_chunks, double* partial, double* total, void* stream) {
  G4R_REQUIRE(n_tensors >= 0 && n_chunks >= 0, "multi_sumsq: bad table size");
  G4R_REQUIRE(total, "multi_sumsq: null total");
  if (n_tensors == 0 || n_chunks == 0) {
    hipError_t e = hipMemsetAsync(total, 0, sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) return g4r_note_hip_error(e, "multi_sumsq: memset");
    return G4R_OK;
  }
  G4R_REQUIRE(g_ptrs && numel && g_is_bf16 && chunk_start && partial, "multi_sumsq: null pointer");
  MultiArgs a = {};
  a.g = (const unsigned long long*)g_ptrs; a.numel = numel; a.g_is_bf16 = g_is_bf16; a.chunk_start = chunk_start;
  a.n = n_tensors;
  hipLaunchKernelGGL(multi_sumsq_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, a, partial);
  G4R_CHECK_LAUNCH("multi_sumsq");
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n_chunks, total);
  G4R_CHECK_LAUNCH("sumsq_finish");
  return G4R_OK;
}

int g4r_multi_adamw_f32(const void* p_ptrs, const void* g_ptrs, const void* m_ptrs, const void* v_ptrs, const void* pb_ptrs,
                        const long* numel, const int* g_is_bf16, const int* chunk_start, int n_tensors, int n_chunks,
                        const double* total_sq, float max_norm, float pre_scale, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int step, void* stream) {
  G4R_REQUIRE(n_tensors >= 0 && n_chunks >= 0 && step >= 1, "multi_adamw: bad table size / step");
  if (n_tensors == 0 || n_chunks == 0) return G4R_OK;
  G4R_REQUIRE(p_ptrs && g_ptrs && m_ptrs && v_ptrs && pb_ptrs && numel && g_is_bf16 && chunk_start, "multi_adamw: null pointer");
  MultiArgs a = {};
  a.p = (const unsigned long long*)p_ptrs; a.g = (const unsigned long long*)g_ptrs;
  a.m = (const unsigned long long*)m_ptrs; a.v = (const unsigned long long*)v_ptrs;
  a.pb = (const unsigned long long*)pb_ptrs; a.numel = numel; a.g_is_bf16 = g_is_bf16; a.chunk_start = chunk_start;
  a.n = n_tensors;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(multi_adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, a, total_sq, max_norm, pre_scale, lr,
                     beta1, beta2, eps, weight_decay, bc1, bc2);
  G4R_CHECK_LAUNCH("multi_adamw");
  return G4R_OK;
}

}  // extern "C"
