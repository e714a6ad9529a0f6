// attention.hip -- fused softmax(Q K^T * scale [+ causal mask]) V for gfx950, bf16 MFMA, fp32
// online softmax.  Serves both transformers of the path:
//   CLIP ViT-L/14 self-attention  (16 heads x 64, S = P^2+1, bidirectional)  -- SURVEY 8a row a4
//   LLaMA-7B self-attention       (32 heads x 128, causal, KV cache)         -- row a16; the region
//   tokens are ordinary query rows attending to the cached K/V (north_star: "region-token x KV").
// The reference delegates this arithmetic to HF transformers (spi_llava.py:66-67, 198-205) or to
// flash-attn (llava/train/llama_flash_attn_monkey_patch.py:15-91).
//
// Mapping: workgroup = 2 waves = 64 query rows of one (batch, head); wave = 32 query rows.
//   S^T = K Q^T  with v_mfma_f32_32x32x16_bf16 (K tile from LDS as the A operand, Q fragments in
//   registers as B) so that every lane owns ONE query column: its 32 score registers, its running
//   max / sum and all of its O^T accumulator registers belong to the same query -> softmax needs
//   one cross-lane exchange (lane ^ 32) and the O rescale is a per-lane scalar.
//   O^T += V^T P^T : V is staged TRANSPOSED in LDS (keys contiguous) so the A operand is two
//   8-byte reads; the key order inside each 16-key MFMA step is permuted to match the registers the
//   lane already holds for P (no P shuffle, no LDS round trip for P).
#include "g4r_common.h"

namespace {


constexpr int KVB = 64;       // keys per tile
constexpr int VT_LD = 68;     // Vt row stride in elements (136 B: conflict-free ds_read_b64)

struct AttnArgs {
  const h16_t* Q;
  const h16_t* K;
  const h16_t* V;
  h16_t* O;
  long q_row, k_row, v_row, o_row;      // row strides (elements)
  long q_batch, k_batch, v_batch, o_batch;
  int Tq, Tk, H;
  float scale;
  int causal;                            // query i sees keys <= i + (Tk - Tq)
  const int* kv_len_dev;                 // optional: Tk = *kv_len_dev + Tq, read on the device (graph-replayed decode)
  float* lse;                            // optional [B][H][Tq]: log2-domain log-sum-exp of the scaled scores (training)
};

template <int D>
__device__ __forceinline__ int k_swz(int row) {
  return D == 128 ? (row & 15) : ((row >> 1) & 7);
}

// NW waves per workgroup = 32*NW query rows.  K/V tiles are software-pipelined through registers:
// the global loads of tile j+1 are issued right after tile j has been written to LDS and stay in
// flight during the QK^T / softmax / PV of tile j.
template <int D, int NW>
__global__ __launch_bounds__(NW * 64, 2) void flash_attn_fwd_kernel(AttnArgs p) {
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;             // query rows per workgroup
  constexpr int SLOTS = D / 8;            // 16-B slots per key row
  constexpr int KSTEPS = D / 16;          // MFMA k-steps over the head dim
  constexpr int DB = D / 32;              // 32-row blocks of O^T
  constexpr int NK = KVB * SLOTS / NT;    // K slots staged per thread
  constexpr int NVU = 16 * SLOTS / NT;    // V units (4 keys x one 8-wide d slot) per thread
  static_assert(KVB * SLOTS % NT == 0 && 16 * SLOTS % NT == 0, "staging split");
  if (p.kv_len_dev) p.Tk = *p.kv_len_dev + p.Tq;  // cached positions before this call + the new rows
  __shared__ __attribute__((aligned(16))) h16_t Ks[KVB * D];
  __shared__ __attribute__((aligned(16))) h16_t Vt[D * VT_LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, ql = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qblock = blockIdx.x * QB;
  const int qi = qblock + wave * 32 + ql;
  const int off = p.Tk - p.Tq;
  const h16_t* Qb = p.Q + (size_t)b * p.q_batch + (size_t)h * D;
  const h16_t* Kb = p.K + (size_t)b * p.k_batch + (size_t)h * D;
  const h16_t* Vb = p.V + (size_t)b * p.v_batch + (size_t)h * D;

  // Q fragments (B operand: lane holds Q[q = lane&31][d = kk*16 + hi*8 .. +7])
  h16x8 qf[KSTEPS];
  {
    const int qr = qi < p.Tq ? qi : p.Tq - 1;
    const h16_t* qrow = Qb + (size_t)qr * p.q_row + hi * 8;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) qf[kk] = *reinterpret_cast<const h16x8*>(qrow + kk * 16);
  }

  float16v oacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sc2 = p.scale * 1.4426950408889634f;  // scores are kept as s*scale*log2(e)

  int kend = p.Tk;
  if (p.causal) {
    const int last = qblock + QB - 1 + off + 1;  // one past the last key any row of this block sees
    if (last < kend) kend = last;
  }

  uint4v kreg[NK], vreg[NVU][4];
  auto fetch = [&](int j0) {
#pragma unroll
    for (int it = 0; it < NK; ++it) {
      const int pk = it * NT + tid;
      const int row = pk / SLOTS, s = pk % SLOTS;
      int key = j0 + row;
      if (key > p.Tk - 1) key = p.Tk - 1;
      kreg[it] = *reinterpret_cast<const uint4v*>(Kb + (size_t)key * p.k_row + s * 8);
    }
#pragma unroll
    for (int u = 0; u < NVU; ++u) {
      const int pu = u * NT + tid;
      const int kq = pu % 16, vs = pu / 16;   // key quad, d slot
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int key = j0 + kq * 4 + j;
        if (key > p.Tk - 1) key = p.Tk - 1;
        vreg[u][j] = *reinterpret_cast<const uint4v*>(Vb + (size_t)key * p.v_row + vs * 8);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < NK; ++it) {
      const int pk = it * NT + tid;
      const int row = pk / SLOTS, s = pk % SLOTS;
      *reinterpret_cast<uint4v*>(reinterpret_cast<char*>(Ks) + row * (D * 2) + ((s ^ k_swz<D>(row)) << 4)) = kreg[it];
    }
    // V^T: this thread holds 4 consecutive keys x 8 d values -> for each d one 8-byte store of 4 keys
#pragma unroll
    for (int u = 0; u < NVU; ++u) {
      const int pu = u * NT + tid;
      const int kq = pu % 16, vs = pu / 16;
      const unsigned w[4][4] = {{vreg[u][0].x, vreg[u][0].y, vreg[u][0].z, vreg[u][0].w},
                                {vreg[u][1].x, vreg[u][1].y, vreg[u][1].z, vreg[u][1].w},
                                {vreg[u][2].x, vreg[u][2].y, vreg[u][2].z, vreg[u][2].w},
                                {vreg[u][3].x, vreg[u][3].y, vreg[u][3].z, vreg[u][3].w}};
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // word e holds d = 2e (low half) and 2e+1 (high half)
        const uint2v lo = {(w[0][e] & 0xffffu) | (w[1][e] << 16), (w[2][e] & 0xffffu) | (w[3][e] << 16)};
        const uint2v hi2 = {(w[0][e] >> 16) | (w[1][e] & 0xffff0000u), (w[2][e] >> 16) | (w[3][e] & 0xffff0000u)};
        *reinterpret_cast<uint2v*>(Vt + (vs * 8 + 2 * e) * VT_LD + kq * 4) = lo;
        *reinterpret_cast<uint2v*>(Vt + (vs * 8 + 2 * e + 1) * VT_LD + kq * 4) = hi2;
      }
    }
  };

  if (kend > 0) fetch(0);
  for (int j0 = 0; j0 < kend; j0 += KVB) {
    commit();
    __syncthreads();
    if (j0 + KVB < kend) fetch(j0 + KVB);  // in flight during this tile's MFMAs

    // ---- S^T = K Q^T for two 32-key blocks ----
    float16v sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
      const int row = kb * 32 + ql;
      const char* krow = reinterpret_cast<const char*>(Ks) + row * (D * 2);
      const int sw = k_swz<D>(row);
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        const h16x8 kf = *reinterpret_cast<const h16x8*>(krow + (((kk * 2 + hi) ^ sw) << 4));
        sacc[kb] = G4R_MFMA_32X32X16(kf, qf[kk], sacc[kb], 0, 0, 0);
      }
    }

    // ---- online softmax for this lane's query (keys: j0 + kb*32 + (r&3) + 8*(r>>2) + 4*hi),
    //      in the log2 domain: p = 2^(s*scale*log2(e) - m) is one v_exp_f32 per score ----
    const bool need_mask = (j0 + KVB > p.Tk) || (p.causal && j0 + KVB - 1 > qblock + wave * 32 + off);
    float mt = -INFINITY;
    if (need_mask) {  // wave-uniform: only the diagonal / last tiles pay for the per-key tests
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          float s = sacc[kb][r] * sc2;
          if (key >= p.Tk || (p.causal && key > qi + off)) s = -INFINITY;
          sacc[kb][r] = s;
          mt = fmaxf(mt, s);
        }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sacc[kb][r] *= sc2;
          mt = fmaxf(mt, sacc[kb][r]);
        }
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;  // fully masked so far: 2^(-inf - 0) = 0
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);  // m_run = -inf -> 0
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(sacc[kb][r] - m_use);
        sacc[kb][r] = e;
        rs += e;
      }
    rs += __shfl_xor(rs, 32);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        uint4v pw;
        pw.x = pack_h16x2(sacc[kb][hf * 8 + 0], sacc[kb][hf * 8 + 1]);
        pw.y = pack_h16x2(sacc[kb][hf * 8 + 2], sacc[kb][hf * 8 + 3]);
        pw.z = pack_h16x2(sacc[kb][hf * 8 + 4], sacc[kb][hf * 8 + 5]);
        pw.w = pack_h16x2(sacc[kb][hf * 8 + 6], sacc[kb][hf * 8 + 7]);
        const h16x8 pf = __builtin_bit_cast(h16x8, pw);
        const int kbase = kb * 32 + hf * 16 + 4 * hi;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const h16_t* vrow = Vt + (d * 32 + ql) * VT_LD + kbase;
          const uint2v lo = *reinterpret_cast<const uint2v*>(vrow);
          const uint2v hi2 = *reinterpret_cast<const uint2v*>(vrow + 8);
          const uint4v vw = {lo.x, lo.y, hi2.x, hi2.y};
          oacc[d] = G4R_MFMA_32X32X16(__builtin_bit_cast(h16x8, vw), pf, oacc[d], 0, 0, 0);
        }
      }
    __syncthreads();
  }

  // ---- normalise and store: lane owns query qi, d = db*32 + 8*g + 4*hi + (0..3) ----
  if (qi < p.Tq) {
    if (p.lse && hi == 0) p.lse[((size_t)b * p.H + h) * p.Tq + qi] = m_run + __log2f(l_run);
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    h16_t* orow = p.O + (size_t)b * p.o_batch + (size_t)qi * p.o_row + (size_t)h * D;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint2v w = {pack_h16x2(oacc[d][g * 4] * inv, oacc[d][g * 4 + 1] * inv),
                          pack_h16x2(oacc[d][g * 4 + 2] * inv, oacc[d][g * 4 + 3] * inv)};
        *reinterpret_cast<uint2v*>(orow + d * 32 + g * 8 + 4 * hi) = w;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Single-query attention over the KV cache: the decode step of row a17 (one new token attends the
// `*kv_len_dev + 1` cached positions).  HBM/L2-bound: 2 * Tk * H * D bf16 of K/V per layer and token,
// 16 flops per byte -- no MFMA.  The tiled kernel above runs this case with ONE workgroup per head
// (32 workgroups, 25 us at 800 keys); here the keys of a head are split over S workgroups (flash-decoding):
//   grid (S, H), 4 waves; a wave takes 4 keys per step: 16 lanes x 16 B cover one 256-B K (and V) row of the
//   head, so every load instruction moves 1 KiB of whole rows; q . k is reduced over the 16 lanes with 4 DPP
//   shuffles; each 16-lane group keeps its own online-softmax state (m, l, o[8 dims per lane]) in fp32.
//   The 16 states of a workgroup are merged through LDS into one un-normalised partial (m, l, o[D]).
//   Partials meet through `ws`; the LAST workgroup of a head to arrive (agent-scope acq_rel counter, the
//   protocol of cdna_hip_programming.md section 6 G16) merges the S partials, writes the bf16 output and
//   re-arms the counter, so there is no second launch and no memset node in the captured graph.  (First version: an
//   acq_rel counter -- correct, but every workgroup then pays an L2 write-back + invalidate; see the hand-off below.)
// p is kept in fp32 for the PV product (the tiled kernel rounds it to bf16 for the MFMA).
// ---------------------------------------------------------------------------------------------
struct DecodeAttnArgs {
  const h16_t* Q;     // [H * D] (already rotated), or null when `qkv` is given
  const h16_t* qkv;   // optional raw projection row [3 * H * D] (q | k | v) of the new token: RoPE is applied here, the
                       // rotated k and the v are appended to the caches at row Tk - 1 (what g4r_rope_qkv_bf16 would do)
  const float* cs;     // cos / sin tables [maxT][D / 2] (with qkv)
  const float* sn;
  h16_t* K;           // [Tmax][k_row]
  h16_t* V;
  h16_t* O;           // [H * D]
  float* ws;           // [H][S][D + 2]
  unsigned* cnt;       // [H], zero before the first call; left zero by every call
  long k_row, v_row;
  int Tk, S, H;
  float scale;
  const int* kv_len_dev;
  int defer;           // 1: leave the S partials in ws for the consumer (g4r_gemv_attn_merge_bf16); O and cnt unused
  long q_batch, k_batch, v_batch, o_batch;   // elements between the sequences of a batch (blockIdx.z)
  int kv_len_stride;   // 0: the sequences share *kv_len_dev; 1: sequence z reads kv_len_dev[z] (ragged batch)
  const int* rope_pos_dev;   // optional (with qkv): RoPE position of the new token when it is not its cache row -- a prompt
                             // whose pad positions were squeezed out of the cache keeps the positions of the padded layout
};

__device__ __forceinline__ void unpack8(const uint4v& r, float* f) {
  f[0] = h16lo(r.x); f[1] = h16hi(r.x); f[2] = h16lo(r.y); f[3] = h16hi(r.y);
  f[4] = h16lo(r.z); f[5] = h16hi(r.z); f[6] = h16lo(r.w); f[7] = h16hi(r.w);
}

template <int D>
__global__ __launch_bounds__(256) void attn_decode_kernel(DecodeAttnArgs p) {
  constexpr int LPK = D / 8;        // lanes per key (16 B each)
  constexpr int KPW = 64 / LPK;     // keys per wave and step
  constexpr int NST = 4 * KPW;      // online-softmax states per workgroup = keys per workgroup step
  constexpr int NB = 8;             // steps whose loads are issued together (2 * NB 16-B loads in flight per lane)
  constexpr int SLD = D + 2;
  __shared__ float st[NST][SLD];
  __shared__ float wgt[64];
  __shared__ unsigned arrival;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane % LPK, ks = lane / LPK;
  const int s = blockIdx.x, h = blockIdx.y, S = p.S;
  {
    const size_t bz = blockIdx.z;              // sequence of the batch: its own q / qkv row, cache slot, output row, partials
    if (p.Q) p.Q += bz * p.q_batch;
    if (p.qkv) p.qkv += bz * p.q_batch;
    p.K += bz * p.k_batch;
    p.V += bz * p.v_batch;
    if (p.O) p.O += bz * p.o_batch;
    if (p.ws) p.ws += bz * (size_t)p.H * S * (D + 2);
    if (p.cnt) p.cnt += bz * p.H;
  }
  const int Tk = p.kv_len_dev ? p.kv_len_dev[(size_t)blockIdx.z * p.kv_len_stride] + 1 : p.Tk;
  int chunk = (Tk + S - 1) / S;
  chunk = (chunk + NST - 1) / NST * NST;
  const int j_begin = s * chunk;
  const int j_end = j_begin + chunk < Tk ? j_begin + chunk : Tk;
  const float sc2 = p.scale * 1.4426950408889634f;
  const h16_t* Kb = p.K + (size_t)h * D + sub * 8;
  const h16_t* Vb = p.V + (size_t)h * D + sub * 8;
  uint4v kv[NB], vv[NB];
  auto load_block = [&](int j0) {
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int j = j0 + u * NST + ks;
      int jc = j < j_end ? j : j_end - 1;
      if (jc < 0) jc = 0;
      kv[u] = __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(Kb + (size_t)jc * p.k_row));
      vv[u] = __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(Vb + (size_t)jc * p.v_row));
    }
  };
  const int j_first = j_begin + wave * KPW;
  load_block(j_first);                 // in flight while q (and the new k) are being rotated below
  float q[8];
  uint4v k_new = {0u, 0u, 0u, 0u}, v_new = {0u, 0u, 0u, 0u};
  const bool fused = p.qkv != nullptr;
  if (fused) {
    // RoPE of this head's q and k rows at position Tk - 1, rotate_half convention, same expressions and bf16
    // roundings as rope_qkv_kernel (elementwise.hip); lane `sub` holds 8 dims, its partner half is lane sub ^ (LPK/2)
    constexpr int HL = LPK / 2;
    const int pos = Tk - 1, v8 = (sub & (HL - 1)) * 8;
    const int rpos = p.rope_pos_dev ? *p.rope_pos_dev : pos;
    const bool second = sub >= HL;
    const size_t own = (size_t)h * D + sub * 8, oth = (size_t)h * D + (sub ^ HL) * 8;
    const size_t HD = (size_t)p.H * D;
    float c[8], sn[8], a[8], b[8];
    {
      const float4v c0 = *reinterpret_cast<const float4v*>(p.cs + (size_t)rpos * (D / 2) + v8);
      const float4v c1 = *reinterpret_cast<const float4v*>(p.cs + (size_t)rpos * (D / 2) + v8 + 4);
      const float4v s0 = *reinterpret_cast<const float4v*>(p.sn + (size_t)rpos * (D / 2) + v8);
      const float4v s1 = *reinterpret_cast<const float4v*>(p.sn + (size_t)rpos * (D / 2) + v8 + 4);
      c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
      sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
    }
    float o[8];
    unpack8(*reinterpret_cast<const uint4v*>(p.qkv + own), a);
    unpack8(*reinterpret_cast<const uint4v*>(p.qkv + oth), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = second ? __builtin_fmaf(a[e], c[e], b[e] * sn[e]) : __builtin_fmaf(a[e], c[e], -(b[e] * sn[e]));
    {
      const uint4v qr = {pack_h16x2(o[0], o[1]), pack_h16x2(o[2], o[3]), pack_h16x2(o[4], o[5]), pack_h16x2(o[6], o[7])};
      unpack8(qr, q);
    }
    unpack8(*reinterpret_cast<const uint4v*>(p.qkv + HD + own), a);
    unpack8(*reinterpret_cast<const uint4v*>(p.qkv + HD + oth), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = second ? __builtin_fmaf(a[e], c[e], b[e] * sn[e]) : __builtin_fmaf(a[e], c[e], -(b[e] * sn[e]));
    k_new = {pack_h16x2(o[0], o[1]), pack_h16x2(o[2], o[3]), pack_h16x2(o[4], o[5]), pack_h16x2(o[6], o[7])};
    v_new = *reinterpret_cast<const uint4v*>(p.qkv + 2 * HD + own);
    if (s == 0 && wave == 0 && ks == 0) {                      // append the new row for the tokens to come
      *reinterpret_cast<uint4v*>(p.K + (size_t)pos * p.k_row + own) = k_new;
      *reinterpret_cast<uint4v*>(p.V + (size_t)pos * p.v_row + own) = v_new;
    }
  } else {
    unpack8(*reinterpret_cast<const uint4v*>(p.Q + (size_t)h * D + sub * 8), q);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) q[e] *= sc2;
  float m = -INFINITY, l = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j0 = j_first; j0 < j_end; j0 += NST * NB) {
    if (j0 != j_first) load_block(j0);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int j = j0 + u * NST + ks;
      if (fused && j == Tk - 1) { kv[u] = k_new; vv[u] = v_new; }   // the row being appended: from registers
      float kf[8], vf[8];
      unpack8(kv[u], kf);
      unpack8(vv[u], vf);
      float sdot = q[0] * kf[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) sdot = fmaf(q[e], kf[e], sdot);
#pragma unroll
      for (int x = 1; x < LPK; x <<= 1) sdot += __shfl_xor(sdot, x);
      if (j < j_end) {
        const float mn = fmaxf(m, sdot);
        const float alpha = exp2f(m - mn), pj = exp2f(sdot - mn);
        l = l * alpha + pj;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = o[e] * alpha + pj * vf[e];
        m = mn;
      }
    }
  }
  // ---- merge the NST states of this workgroup ----
  {
    float* row = st[wave * KPW + ks];
#pragma unroll
    for (int e = 0; e < 8; ++e) row[sub * 8 + e] = o[e];
    if (sub == 0) { row[D] = m; row[D + 1] = l; }
  }
  __syncthreads();
  if (tid < 64) {
    float mm = -INFINITY;
    for (int i = 0; i < NST; ++i) mm = fmaxf(mm, st[i][D]);
    float w = 0.f;
    if (tid < NST) w = st[tid][D] == -INFINITY ? 0.f : exp2f(st[tid][D] - mm);
    wgt[tid] = w;
  }
  __syncthreads();
  float M = -INFINITY, L = 0.f, acc = 0.f;
  if (tid < D) {
    for (int i = 0; i < NST; ++i) {
      M = fmaxf(M, st[i][D]);
      L += wgt[i] * st[i][D + 1];
      acc += wgt[i] * st[i][tid];
    }
  }
  if (S == 1 && !p.defer) {
    if (tid < D) p.O[(size_t)h * D + tid] = f32_to_h16(acc / L);
    return;
  }
  if (p.defer) {                      // the consumer merges: plain stores, the kernel boundary publishes them
    float* mine = p.ws + ((size_t)h * S + s) * SLD;
    if (tid < D) {
      mine[tid] = acc;
      if (tid == 0) { mine[D] = M; mine[D + 1] = L; }
    }
    return;
  }
  // ---- hand-off without fences (cdna_hip_programming.md section 6 G16, write-through variant): the partial is stored with
  // agent-scope relaxed atomics (sc1 write-through: it bypasses this XCD's non-coherent L2), drained (vmcnt(0)) before the
  // arrival counter is bumped; the last workgroup to arrive reads the partials with agent-scope relaxed loads (sc1).  An
  // acq_rel counter instead costs an L2 write-back + invalidate per workgroup: 20-30 us per launch measured. ----
  float* mine = p.ws + ((size_t)h * S + s) * SLD;
  if (tid < D) {
    __hip_atomic_store(mine + tid, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
      __hip_atomic_store(mine + D, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + D + 1, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) arrival = __hip_atomic_fetch_add(p.cnt + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (arrival != (unsigned)(S - 1)) return;
  float* part = p.ws + (size_t)h * S * SLD;
  if (tid < 64) wgt[tid] = tid < S ? __hip_atomic_load(part + tid * SLD + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -INFINITY;
  __syncthreads();
  float mm = -INFINITY;
  for (int i = 0; i < S; ++i) mm = fmaxf(mm, wgt[i]);
  if (tid < D) {
    float Lt = 0.f, a2 = 0.f;
    for (int i0 = 0; i0 < S; i0 += 16) {
      float pl[16], po[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {       // all loads of a batch issued before the first use
        const int i = i0 + u < S ? i0 + u : S - 1;
        pl[u] = __hip_atomic_load(part + i * SLD + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        po[u] = __hip_atomic_load(part + i * SLD + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (i0 + u < S) {
          const float mi = wgt[i0 + u];
          const float w = mi == -INFINITY ? 0.f : exp2f(mi - mm);
          Lt += w * pl[u];
          a2 += w * po[u];
        }
      }
    }
    p.O[(size_t)h * D + tid] = f32_to_h16(a2 / Lt);
  }
  if (tid == 0) __hip_atomic_store(p.cnt + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

static int g_attn_variant = 0;   // tools only: 1 = force the first-form kernel (head_dim 128: 4 waves), 2 = its 2-wave form,
                                 // >= 10: the second form (attention_v2.hip) with NWG * 10 + NG

// attention_v2.hip: key groups inside the workgroup, K/V by LDS-DMA, V through ds_read_b64_tr_b16
int g4r_attn2_dispatch(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq, int Tk, int head_dim,
                       long q_row, long k_row, long v_row, long o_row, long q_batch, long k_batch, long v_batch,
                       long o_batch, float scale, int causal, const int* kv_len_dev, float* lse, int variant,
                       void* stream);

extern "C" {

#ifndef G4R_F16
void g4r_attn_debug_variant(int v) { g_attn_variant = v; }
#endif

// Q [B][Tq][H*D-strided rows], K/V [B][Tk][...], O [B][Tq][...]; all bf16; row/batch strides in
// elements (multiples of 8).  head_dim 64 or 128.  causal: query i attends keys <= i + (Tk - Tq).
int g4r_flash_attn_fwd_bf16(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq, int Tk,
                            int head_dim, long q_row, long k_row, long v_row, long o_row, long q_batch,
                            long k_batch, long v_batch, long o_batch, float scale, int causal,
                            const int* kv_len_dev, float* lse, void* stream) {
  G4R_REQUIRE(B > 0 && H > 0 && Tq >= 0 && Tk > 0, "flash_attn: bad shape");
  G4R_REQUIRE(head_dim == 64 || head_dim == 128, "flash_attn: head_dim must be 64 or 128");
  if (Tq == 0) return G4R_OK;
  G4R_REQUIRE(Q && K && V && O, "flash_attn: null pointer");
  G4R_REQUIRE(q_row % 8 == 0 && k_row % 8 == 0 && v_row % 8 == 0 && o_row % 4 == 0 && q_batch % 8 == 0 &&
                  k_batch % 8 == 0 && v_batch % 8 == 0 && o_batch % 4 == 0,
              "flash_attn: strides must keep 16-byte alignment");
  G4R_REQUIRE(!causal || Tk >= Tq, "flash_attn: causal needs Tk >= Tq");
  // the prefill shapes of the path go to the second form; a handful of query rows against a long cache (the host-loop
  // decode, Tq < 32) stays on the first form, whose 2 x 64-row workgroups waste less on an almost empty query block
  // second form: every launch with Tq >= 32 (since its second pass it also wins the many-block launches of the training
  // batch: 8 x 699 tokens 89.4 vs 93.2 us, profiles/r03_attention_time_final.txt)
  const bool second_form = g_attn_variant >= 10 || (g_attn_variant == 0 && Tq >= 32);
  // the second form addresses K / V through 32-bit buffer offsets: a head's rows must span < 2 GiB (it traps otherwise)
  const long kv_row = k_row > v_row ? k_row : v_row;
  const bool spans_ok = ((long)(Tk > 0 ? Tk - 1 : 0) * kv_row + head_dim) * 2 < 0x7fffffffL;
  if (second_form && spans_ok && o_row % 8 == 0 && o_batch % 8 == 0)   // 16-byte O rows
    return g4r_attn2_dispatch(Q, K, V, O, B, H, Tq, Tk, head_dim, q_row, k_row, v_row, o_row, q_batch, k_batch, v_batch,
                              o_batch, scale, causal, kv_len_dev, lse, g_attn_variant >= 10 ? g_attn_variant : 0, stream);
  AttnArgs a = {(const h16_t*)Q, (const h16_t*)K, (const h16_t*)V, (h16_t*)O, q_row, k_row, v_row, o_row,
                q_batch, k_batch, v_batch, o_batch, Tq, Tk, H, scale, causal, kv_len_dev, lse};
  // 64 query rows per workgroup (2 waves): ~2x the workgroups of a 128-row block for the short
  // sequences of this path (577 / ~800 tokens) and finer causal load balance
  if (head_dim == 64) {
    dim3 grid(g4r_ceil_div(Tq, 64), H, B);
    hipLaunchKernelGGL((flash_attn_fwd_kernel<64, 2>), grid, dim3(128), 0, (hipStream_t)stream, a);
  } else {
    // head_dim 128: 4 waves (128 query rows) share the staging registers of a tile.  The 2-wave form (64 rows: twice the
    // workgroups -- LLaMA prefill at T = 767 has only 6 x 32 = 192 of them for 256 CUs) spills 72 B per lane and pays the
    // K/V staging twice per query row: 66.0 vs 38.6 us at T = 767, 175 vs 104 at T = 2048 (tools/attn_probe.py), so it
    // is a probe variant only.
    const bool two = g_attn_variant == 2;
    if (two) {
      dim3 grid(g4r_ceil_div(Tq, 64), H, B);
      hipLaunchKernelGGL((flash_attn_fwd_kernel<128, 2>), grid, dim3(128), 0, (hipStream_t)stream, a);
    } else {
      dim3 grid(g4r_ceil_div(Tq, 128), H, B);
      hipLaunchKernelGGL((flash_attn_fwd_kernel<128, 4>), grid, dim3(256), 0, (hipStream_t)stream, a);
    }
  }
  G4R_CHECK_LAUNCH("flash_attn_fwd");
  return G4R_OK;
}

// One query row (the token being decoded) against the first Tk = *kv_len_dev + 1 (or `Tk` when kv_len_dev is null) rows
// of a KV cache: Q/O [H*D] bf16, K/V rows `k_row`/`v_row` elements apart.  `splits` workgroups per head share the keys;
// workspace = H*splits*(D+2) floats, counters = H uint32 that are zero before the first call (every call leaves them zero).
// With `qkv` (the raw q|k|v projection row of the new token, [3*H*D]) the call also does what g4r_rope_qkv_bf16 does for
// that token: q and k are rotated with cos/sin row Tk-1, k and v are appended to the caches at row Tk-1 (Q is ignored).
// batch > 1: `batch` sequences of equal length in one launch (grid z): sequence b reads Q/qkv + b*q_batch, the cache slot
// K/V + b*k_batch / v_batch, writes O + b*o_batch; workspace / counters hold `batch` consecutive sets.
// defer_merge: the S partials (un-normalised o, max, sum per split) stay in the workspace and the consumer assembles the
// output (g4r_gemv_attn_merge_bf16, the o_proj of the decode step) -- no hand-off inside this launch; O/counters unused.
// Replaces the Tq = 1 case of g4r_flash_attn_fwd_bf16 in the decode loop the reference reaches through HF generate()
// (gpt4roi/app.py:293-300 -> transformers LlamaAttention with past_key_values).
static int attn_decode_launch(const void* Q, const void* qkv, const float* cos_tab, const float* sin_tab, void* K, void* V,
                         void* O, float* workspace, unsigned* counters, int H, int head_dim, int Tk, long k_row,
                         long v_row, float scale, int splits, const int* kv_len_dev, int defer_merge, int batch,
                         long q_batch, long k_batch, long v_batch, long o_batch, int kv_len_stride,
                              const int* rope_pos_dev, void* stream) {
  G4R_REQUIRE(H > 0 && (Tk > 0 || kv_len_dev) && batch >= 1 && batch <= 65535, "attn_decode: bad shape");
  G4R_REQUIRE(head_dim == 64 || head_dim == 128, "attn_decode: head_dim must be 64 or 128");
  G4R_REQUIRE((Q || qkv) && K && V && (O || defer_merge), "attn_decode: null pointer");
  G4R_REQUIRE(!defer_merge || workspace, "attn_decode: defer_merge needs the workspace");
  G4R_REQUIRE(!qkv || (cos_tab && sin_tab), "attn_decode: the fused RoPE needs the cos/sin tables");
  G4R_REQUIRE(splits >= 1 && splits <= 64, "attn_decode: splits must be in [1, 64]");
  G4R_REQUIRE(splits == 1 || defer_merge || (workspace && counters), "attn_decode: split keys need workspace and counters");
  G4R_REQUIRE(k_row % 8 == 0 && v_row % 8 == 0 && q_batch % 8 == 0 && k_batch % 8 == 0 && v_batch % 8 == 0 &&
                  o_batch % 8 == 0, "attn_decode: strides must keep 16-byte alignment");
  DecodeAttnArgs a = {(const h16_t*)Q, (const h16_t*)qkv, cos_tab, sin_tab, (h16_t*)K, (h16_t*)V, (h16_t*)O,
                      workspace, counters, k_row, v_row, Tk, splits, H, scale, kv_len_dev, defer_merge,
                      q_batch, k_batch, v_batch, o_batch, kv_len_stride, rope_pos_dev};
  dim3 grid(splits, H, batch);
  if (head_dim == 64)
    hipLaunchKernelGGL((attn_decode_kernel<64>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((attn_decode_kernel<128>), grid, dim3(256), 0, (hipStream_t)stream, a);
  G4R_CHECK_LAUNCH("attn_decode");
  return G4R_OK;
}

int g4r_attn_decode_bf16(const void* Q, const void* qkv, const float* cos_tab, const float* sin_tab, void* K, void* V,
                         void* O, float* workspace, unsigned* counters, int H, int head_dim, int Tk, long k_row,
                         long v_row, float scale, int splits, const int* kv_len_dev, int defer_merge, int batch,
                         long q_batch, long k_batch, long v_batch, long o_batch, void* stream) {
  return attn_decode_launch(Q, qkv, cos_tab, sin_tab, K, V, O, workspace, counters, H, head_dim, Tk, k_row, v_row, scale,
                            splits, kv_len_dev, defer_merge, batch, q_batch, k_batch, v_batch, o_batch, 0, nullptr, stream);
}

// Ragged batch: sequence z attends its own first kv_lens_dev[z] + 1 rows and appends the new row at kv_lens_dev[z]; the
// RoPE position of the new tokens is *rope_pos_dev for every sequence (null: the cache row).  This is what HF's decoder
// does for a batch with a padding mask (positions count the padded layout, masked keys are not attended; the reference
// reaches it through generate() with attention_mask, llava/model/llava.py:263-283) once the pad rows have been squeezed
// out of the cache.
int g4r_attn_decode_ragged_bf16(const void* qkv, const float* cos_tab, const float* sin_tab, void* K, void* V, void* O,
                                float* workspace, unsigned* counters, int H, int head_dim, long k_row, long v_row,
                                float scale, int splits, const int* kv_lens_dev, const int* rope_pos_dev, int batch,
                                long q_batch, long k_batch, long v_batch, long o_batch, void* stream) {
  G4R_REQUIRE(qkv && kv_lens_dev, "attn_decode_ragged: needs the projection rows and the per-sequence lengths");
  return attn_decode_launch(nullptr, qkv, cos_tab, sin_tab, K, V, O, workspace, counters, H, head_dim, 0, k_row, v_row,
                            scale, splits, kv_lens_dev, 0, batch, q_batch, k_batch, v_batch, o_batch, 1, rope_pos_dev,
                            stream);
}

}  // extern "C"
