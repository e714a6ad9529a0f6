// sampling.hip -- device-side token sampling for generate(do_sample=True) (gfx950).
//
// The reference samples through HF `generate(do_sample=True, temperature=0.2, ...)` (gpt4roi/app.py:293-300); HF applies
// its logits warpers in the order temperature -> top-k -> top-p (top_k = 50 and top_p = 1.0 are the GenerationConfig
// defaults of the pinned transformers) and then draws from the remaining distribution.  This kernel is that step with the
// token id, position and step counter resident on the device (so the per-token decode step stays hipGraph-replayable,
// like g4r_greedy_advance_f32) and a counter-based generator: the uniform of step s is Philox4x32-10(counter = (s,0,0,0),
// key = seed) -- reproducible for a fixed seed whatever the launch geometry.  The draw is an inverse CDF over the kept
// tokens in ascending vocabulary order; oracle/sampler_oracle.py restates it (and is pinned to the Random123 known-answer
// vectors for Philox).
#include "g4r_common.h"

namespace {

__device__ __forceinline__ uint32_t order_key(float x) {          // monotone float -> uint map
  const uint32_t b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

constexpr int LIST_CAP = 1024;

// One workgroup of 1024 threads.  top_k in [1, LIST_CAP]: list path (kept tokens gathered into LDS, everything else by
// thread 0 on <= ~top_k entries).  top_k == 0: every token kept, chunked inverse CDF (top_p must be 1).
// The logits row is staged ONCE in LDS (`in_lds`: N * 4 B of dynamic LDS, 128 KB for the 32 006-token vocabulary) and the
// six passes over it (max, four radix-select passes, gather) read it from there; the first version walked the row in
// global memory with 256 threads, ~140 us per token.
constexpr int SNT = 1024;
__global__ __launch_bounds__(SNT) void sample_advance_kernel(const float* __restrict__ glogits, int N, int in_lds, float inv_temp,
                                                             int top_k, float top_p,
                                                             const unsigned long long* __restrict__ seed,
                                                             long* __restrict__ tok, long* __restrict__ out_ids,
                                                             int* __restrict__ step, int* __restrict__ pos,
                                                             int max_steps, float* __restrict__ u_out) {
  extern __shared__ __attribute__((aligned(16))) float staged[];
  __shared__ float red[SNT];
  __shared__ unsigned hist[256];
  __shared__ unsigned sel_prefix, sel_mask, sel_k, list_n;
  __shared__ int list_idx[LIST_CAP];
  __shared__ float list_e[LIST_CAP];
  __shared__ double chunk_sum[256];
  const int tid = threadIdx.x;
  const float* logits = glogits;
  if (in_lds) {
    for (int i = tid; i < N; i += SNT) staged[i] = glogits[i];
    __syncthreads();
    logits = staged;
  }
  // ---- max logit ----
  float m = -INFINITY;
  for (int i = tid; i < N; i += SNT) m = fmaxf(m, logits[i]);
  red[tid] = m;
  __syncthreads();
  for (int s = SNT / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
    __syncthreads();
  }
  m = red[0];
  __syncthreads();
  // ---- the uniform of this step ----
  const int st = *step;
  uint32_t r4[4];
  const unsigned long long sd = seed ? *seed : 0ull;
  philox4x32_10((uint32_t)st, 0u, 0u, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), r4);
  const float u = (float)(r4[0] >> 8) * (1.0f / 16777216.0f);        // 24-bit uniform in [0, 1)
  int choice = -1;
  const int k_eff = top_k < N ? top_k : N;   // HF: top_k = min(top_k, vocabulary)
  uint32_t kth = 0;                          // order key of the k-th largest logit (0: every token kept)
  bool use_list = false;
  if (top_k >= 1 && (k_eff < N || N <= LIST_CAP)) {
    // ---- k-th largest logit by MSB radix select on the order-preserving key (ties with the k-th are all kept, as
    //      HF's TopKLogitsWarper does: scores < kth are removed) ----
    if (tid == 0) { sel_prefix = 0; sel_mask = 0; sel_k = (unsigned)k_eff; }
    for (int pass = 3; pass >= 0; --pass) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const unsigned pre = sel_prefix, msk = sel_mask;
      const int sh = pass * 8;
      for (int i = tid; i < N; i += SNT) {
        const uint32_t k = order_key(logits[i]);
        if ((k & msk) == pre) atomicAdd(&hist[(k >> sh) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned cum = 0, k_rem = sel_k;
        int b = 255;
        for (; b > 0; --b) {
          if (cum + hist[b] >= k_rem) break;
          cum += hist[b];
        }
        sel_k = k_rem - cum;
        sel_prefix = pre | ((unsigned)b << sh);
        sel_mask = msk | (255u << sh);
      }
      __syncthreads();
    }
    kth = sel_prefix;
    if (tid == 0) list_n = 0;
    __syncthreads();
    for (int i = tid; i < N; i += SNT) {
      const float l = logits[i];
      if (order_key(l) >= kth) {
        const unsigned slot = atomicAdd(&list_n, 1u);
        if (slot < LIST_CAP) {
          list_idx[slot] = i;
          list_e[slot] = expf((l - m) * inv_temp);
        }
      }
    }
    __syncthreads();
    // more than LIST_CAP tokens tie at the k-th logit (degenerate rows, e.g. constant logits): the kept set does not
    // fit the list -> chunked draw over the tokens with key >= kth below (top_p is not applied in that case)
    use_list = list_n <= (unsigned)LIST_CAP;
    if (use_list) {
      // ascending vocabulary order by rank: entry a goes to slot #{b : idx[b] < idx[a]} (the indices are distinct); one
      // thread per entry instead of thread 0's insertion sort (~30 us at top_k = 50)
      const int n = (int)list_n;
      int my_i = 0, rank = 0;
      float my_e = 0.f;
      if (tid < n) {
        my_i = list_idx[tid];
        my_e = list_e[tid];
        for (int b = 0; b < n; ++b) rank += list_idx[b] < my_i;
      }
      __syncthreads();
      if (tid < n) { list_idx[rank] = my_i; list_e[rank] = my_e; }
      __syncthreads();
    }
    if (use_list && tid == 0) {
      int n = (int)list_n;
      double Z = 0.0;
      for (int a = 0; a < n; ++a) Z += (double)list_e[a];
      if (top_p < 1.0f) {
        // HF TopPLogitsWarper: a token stays iff the probability mass of the STRICTLY more probable tokens is < top_p
        // (ties ordered by vocabulary index); at least one token stays.
        for (int a = 0; a < n; ++a) {
          double above = 0.0;
          for (int b = 0; b < n; ++b)
            if (list_e[b] > list_e[a] || (list_e[b] == list_e[a] && list_idx[b] < list_idx[a])) above += (double)list_e[b];
          if (!(above < (double)top_p * Z)) list_idx[a] = -1 - list_idx[a];   // mark removed
        }
        double Z2 = 0.0;
        for (int a = 0; a < n; ++a)
          if (list_idx[a] >= 0) Z2 += (double)list_e[a];
        Z = Z2;
      }
      const double target = (double)u * Z;
      double acc = 0.0;
      int last = -1;
      for (int a = 0; a < n; ++a) {
        if (list_idx[a] < 0) continue;
        last = list_idx[a];
        acc += (double)list_e[a];
        if (acc > target) { choice = last; break; }
      }
      if (choice < 0) choice = last;
    }
  }
  if (!use_list) {
    // ---- chunked inverse CDF in ascending vocabulary order over the tokens with key >= kth (kth = 0: all) ----
    const int chunk = (N + 255) / 256;
    if (tid < 256) {
      const int c0 = tid * chunk, c1 = min(N, c0 + chunk);
      double s = 0.0;
      for (int i = c0; i < c1; ++i)
        if (order_key(logits[i]) >= kth) s += (double)expf((logits[i] - m) * inv_temp);
      chunk_sum[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
      double Z = 0.0;
      for (int c = 0; c < 256; ++c) Z += chunk_sum[c];
      const double target = (double)u * Z;
      double acc = 0.0;
      int c = 0;
      for (; c < 255; ++c) {
        if (acc + chunk_sum[c] > target) break;
        acc += chunk_sum[c];
      }
      const int b0 = c * chunk, b1 = min(N, b0 + chunk);
      choice = -1;
      for (int i = b0; i < b1; ++i) {
        if (order_key(logits[i]) < kth) continue;
        choice = i;                       // the last kept token seen: the fallback if rounding leaves acc <= target
        acc += (double)expf((logits[i] - m) * inv_temp);
        if (acc > target) break;
      }
      if (choice < 0)                     // rounding pushed the target past the last chunk: the last kept token
        for (int i = N - 1; i >= 0; --i)
          if (order_key(logits[i]) >= kth) { choice = i; break; }
    }
  }
  if (tid == 0) {
    if (choice < 0) choice = 0;
    tok[0] = choice;
    if (st < max_steps) out_ids[st] = choice;
    if (u_out && st < max_steps) u_out[st] = u;
    *step = st + 1;
    *pos = *pos + 1;
  }
}

}  // namespace

extern "C" int g4r_sample_advance_f32(const float* logits, int N, float temperature, int top_k, float top_p,
                                      const unsigned long long* seed, long* tok, long* out_ids, int* step, int* pos,
                                      int max_steps, float* u_out, void* stream) {
  G4R_REQUIRE(logits && tok && out_ids && step && pos, "sample_advance: null pointer");
  G4R_REQUIRE(N > 0 && temperature > 0.f, "sample_advance: N > 0 and temperature > 0");
  G4R_REQUIRE(top_p > 0.f && top_p <= 1.f, "sample_advance: top_p in (0, 1]");
  G4R_REQUIRE(top_k >= 0 && top_k <= LIST_CAP, "sample_advance: top_k in [0, 1024] (0 = disabled)");
  if (top_p < 1.f && !(top_k >= 1 && (top_k < N || N <= LIST_CAP)))
    return g4r_note_error(G4R_ERR_UNSUPPORTED, "sample_advance: top_p < 1 needs top_k in [1, 1024] (below the vocabulary size)");
  const int in_lds = (size_t)N * 4 <= 140 * 1024;       // + 15 KB of static LDS: within the 160 KB of a CU
  const size_t lds = in_lds ? (size_t)N * 4 : 0;
  static G4rPerDeviceOnce attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sample_advance_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    if (e != hipSuccess) { attr_set.failed(); return g4r_note_hip_error(e, "sample_advance: hipFuncSetAttribute"); }
  }
  hipLaunchKernelGGL(sample_advance_kernel, dim3(1), dim3(SNT), lds, (hipStream_t)stream, logits, N, in_lds,
                     1.0f / temperature, top_k, top_p, seed, tok, out_ids, step, pos, max_steps, u_out);
  G4R_CHECK_LAUNCH("sample_advance");
  return G4R_OK;
}
