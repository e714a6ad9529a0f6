// attention_bwd.hip -- backward of softmax(Q K^T * scale [+ causal mask]) V for gfx950 (bf16 MFMA, fp32
// accumulation).  Training rows of the path: the reference gets this gradient
// from autograd through HF LlamaAttention or from flash-attn's backward
// (llava/train/llama_flash_attn_monkey_patch.py:15-91).
//
// Two kernels, each the mirror image of the forward kernel's register layout so that no probability
// tile ever needs a transpose:
//   dQ     : "lane owns a query" (exactly the forward orientation).  S^T = K Q^T and dP^T = V dO^T land
//            with one query per lane, so LSE and Delta are lane scalars; dQ^T += K^T dS^T uses the packed
//            dS^T registers as the B operand and K^T as the A operand, read TRANSPOSED (ds_read_b64_tr_b16) from the
//            row-major K tile in LDS.
//   dK, dV : "lane owns a key".  S = Q K^T and dP = dO V^T with the key block's K/V fragments resident in
//            registers and the streamed query tile in LDS; dV^T += dO^T P and dK^T += Q^T dS read Q / dO
//            transposed from the same row-major tiles.  LSE / Delta of the 64 streamed queries sit in LDS.
// P is recomputed from the forward's log2-domain LSE: P = 2^(s*scale*log2e - lse).  Delta = rowsum(dO*O)
// comes from a small pre-pass.  Nothing is atomically accumulated: results are bit-reproducible.
#include "g4r_common.h"

namespace {


constexpr int TB = 64;     // rows of the streamed tile (keys for dQ, queries for dK/dV)

struct AttnBwdArgs {
  const h16_t *Q, *K, *V, *O, *dO;
  h16_t *dQ, *dK, *dV;
  const float* lse;  // [B][H][Tq], log2 domain (forward)
  float* delta;      // [B][H][Tq]
  long q_row, k_row, v_row, o_row, do_row, dq_row, dk_row, dv_row;
  long q_batch, k_batch, v_batch, o_batch, do_batch, dq_batch, dk_batch, dv_batch;
  int Tq, Tk, H;
  float scale;
  int causal;
};

// XOR swizzle of the 16-byte slots of a row-major tile row.  D = 128 (16 slots per 256-byte row): the two bit pairs of
// row & 15 swapped -- 16 consecutive rows still take 16 different slot offsets (the ds_read_b128 fragment reads of 8 / 16
// consecutive rows stay conflict-free, as with the plain row & 15 of rounds 1-3), AND four consecutive rows r0..r0+3 (r0 a
// multiple of 4) differ in slot bits 2-3, which is what the transposing reads below need: a 16-lane group of
// ds_read_b64_tr_b16 fetches 4 rows x 32 bytes, and with this swizzle those land in four different 64-byte bank groups.
template <int D>
__device__ __forceinline__ int row_swz(int row) {
  return D == 128 ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((row >> 1) & 7);
}

// 64 rows x D of a [rows, H*D]-strided matrix -> LDS, row-major with the 16-byte slots of a row XOR-swizzled
template <int D, int NT>
__device__ __forceinline__ void stage_rows(h16_t* dst, const h16_t* src, long row_stride, int r0, int rmax,
                                           int tid) {
  constexpr int SLOTS = D / 8;
#pragma unroll
  for (int it = 0; it < TB * SLOTS / NT; ++it) {
    const int pk = it * NT + tid;
    const int row = pk / SLOTS, s = pk % SLOTS;
    int r = r0 + row;
    if (r > rmax - 1) r = rmax - 1;
    const uint4v v = *reinterpret_cast<const uint4v*>(src + (size_t)r * row_stride + s * 8);
    *reinterpret_cast<uint4v*>(reinterpret_cast<char*>(dst) + row * (D * 2) + ((s ^ row_swz<D>(row)) << 4)) = v;
  }
}

// Round 4: ONE global pass per streamed tile and a register prefetch.  A thread holds 4 consecutive rows x one 16-byte slot
// and writes the row-major swizzled LDS image from them (4 x 16-byte stores; until the transposing reads below it also wrote a
// transposed image, 8 x 8-byte stores from the same registers), and
// `tile_load` for tile i+1 is issued before the MFMAs of tile i (the kernels run ONE workgroup per CU at 284-358 registers per
// lane: without the prefetch nothing overlapped the load latency, and a 64-row tile took ~11 us against ~1 us of MFMAs).
template <int D, int NT>
struct TileRegs {
  static_assert(16 * (D / 8) == NT, "one (row quad, slot) unit per thread");
  uint4v w[4];
};

template <int D, int NT>
__device__ __forceinline__ void tile_load(TileRegs<D, NT>& t, const h16_t* src, long row_stride, int r0, int rmax, int tid) {
  const int rq = tid % 16, vs = tid / 16;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int r = r0 + rq * 4 + j;
    if (r > rmax - 1) r = rmax - 1;
    t.w[j] = *reinterpret_cast<const uint4v*>(src + (size_t)r * row_stride + vs * 8);
  }
}

template <int D, int NT>
__device__ __forceinline__ void tile_store_rows(const TileRegs<D, NT>& t, h16_t* dst, int tid) {
  const int rq = tid % 16, vs = tid / 16;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = rq * 4 + j;
    *reinterpret_cast<uint4v*>(reinterpret_cast<char*>(dst) + row * (D * 2) + ((vs ^ row_swz<D>(row)) << 4)) = t.w[j];
  }
}

template <int D>
__device__ __forceinline__ h16x8 frag_rows(const h16_t* tile, int row, int kk, int hi) {
  return *reinterpret_cast<const h16x8*>(reinterpret_cast<const char*>(tile) + row * (D * 2) +
                                          (((kk * 2 + hi) ^ row_swz<D>(row)) << 4));
}

// A operand = the TRANSPOSE of a row-major tile (rows = the 64 streamed keys / queries, columns = head dim), read with
// ds_read_b64_tr_b16 straight from the row-major image (round 4; rounds 1-3 wrote a second, transposed image of every tile
// with 8-byte stores from packed registers).  The MFMA's A row is the head-dim index d = 32 * dblk + (lane & 31); its 8
// contraction indices are tile rows base + 4 hi + {0..3} and base + 8 + 4 hi + {0..3} -- the order the C-layout registers of
// the other operand (P / dS, packed by pack8) already have.  A 16-lane group fetches [4 rows][16 columns]: lane c of it
// supplies the 8 bytes at row (c >> 2), columns 4 (c & 3)..+3 and receives column c of the four rows.
template <int D>
struct TrLane {
  int off[2][D / 32];     // byte offset inside a tile (without the 16-row block) for the two reads, per 32-wide head-dim block
  __device__ __forceinline__ void init(int lane) {
    const int c = lane & 15, g = (lane >> 4) & 1, hi = lane >> 5, r4 = c >> 2;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int rowl = 4 * hi + r4 + 8 * rd;                  // row & 15: all the swizzle depends on
      const int sw = row_swz<D>(rowl);
#pragma unroll
      for (int d = 0; d < D / 32; ++d) {
        const int slot = d * 4 + g * 2 + ((c >> 1) & 1);
        off[rd][d] = rowl * (D * 2) + ((slot ^ sw) << 4) + (c & 1) * 8;
      }
    }
  }
};

template <int D>
__device__ __forceinline__ h16x8 frag_tr(const h16_t* tile, const TrLane<D>& tl, int dblk, int base16) {
  // base16: first row of the 16-row block (a multiple of 16)
  const char* t = reinterpret_cast<const char*>(tile) + base16 * (D * 2);
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(t + tl.off[0][dblk]));
  const short4v up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(t + tl.off[1][dblk]));
  const short8 vv = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
  return __builtin_bit_cast(h16x8, vv);
}

__device__ __forceinline__ h16x8 pack8(const float* v) {
  const uint4v w = {pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3]), pack_h16x2(v[4], v[5]),
                    pack_h16x2(v[6], v[7])};
  return __builtin_bit_cast(h16x8, w);
}

// ---------------------------------------------------------------------------------------------
// Delta[b][h][i] = sum_d dO[b][i][h*D + d] * O[b][i][h*D + d]
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnBwdArgs p, int B) {
  constexpr int LPR = D / 8;  // lanes per (row, head)
  const long total = (long)B * p.Tq * p.H * LPR;
  // LPR divides 64 and total is a multiple of LPR: a shuffle group is either wholly in range or wholly out
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    float acc = 0.f;
    const int l = (int)(i % LPR);
    const long rh = i / LPR;
    const int h = (int)(rh % p.H);
    const long bi = rh / p.H;
    const int q = (int)(bi % p.Tq);
    const int b = (int)(bi / p.Tq);
    {
      const uint4v o = *reinterpret_cast<const uint4v*>(p.O + (size_t)b * p.o_batch + (size_t)q * p.o_row + h * D + l * 8);
      const uint4v g = *reinterpret_cast<const uint4v*>(p.dO + (size_t)b * p.do_batch + (size_t)q * p.do_row + h * D + l * 8);
      acc = h16lo(o.x) * h16lo(g.x) + h16hi(o.x) * h16hi(g.x) + h16lo(o.y) * h16lo(g.y) +
            h16hi(o.y) * h16hi(g.y) + h16lo(o.z) * h16lo(g.z) + h16hi(o.z) * h16hi(g.z) +
            h16lo(o.w) * h16lo(g.w) + h16hi(o.w) * h16hi(g.w);
    }
#pragma unroll
    for (int s = LPR / 2; s >= 1; s >>= 1) acc += __shfl_xor(acc, s);
    if (l == 0) p.delta[((size_t)b * p.H + h) * p.Tq + q] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// dQ: one workgroup = NW waves = 32*NW queries of one (batch, head); streams the K/V tiles.
// ---------------------------------------------------------------------------------------------
template <int D, int NW>
__global__ __launch_bounds__(NW * 64, 1) void attn_bwd_dq_kernel(AttnBwdArgs p) {
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;
  constexpr int KSTEPS = D / 16;
  constexpr int DB = D / 32;
  __shared__ __attribute__((aligned(16))) h16_t Ks[TB * D];
  __shared__ __attribute__((aligned(16))) h16_t Vs[TB * D];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, ql = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qblock = blockIdx.x * QB;
  const int qi = qblock + wave * 32 + ql;
  const int off = p.Tk - p.Tq;
  const h16_t* Kb = p.K + (size_t)b * p.k_batch + (size_t)h * D;
  const h16_t* Vb = p.V + (size_t)b * p.v_batch + (size_t)h * D;
  const int qr = qi < p.Tq ? qi : p.Tq - 1;

  h16x8 qf[KSTEPS], dof[KSTEPS];
  {
    const h16_t* qrow = p.Q + (size_t)b * p.q_batch + (size_t)qr * p.q_row + (size_t)h * D + hi * 8;
    const h16_t* grow = p.dO + (size_t)b * p.do_batch + (size_t)qr * p.do_row + (size_t)h * D + hi * 8;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
      qf[kk] = *reinterpret_cast<const h16x8*>(qrow + kk * 16);
      dof[kk] = *reinterpret_cast<const h16x8*>(grow + kk * 16);
    }
  }
  const float lse_i = p.lse[((size_t)b * p.H + h) * p.Tq + qr];
  const float delta_i = p.delta[((size_t)b * p.H + h) * p.Tq + qr];
  TrLane<D> trl;
  trl.init(lane);
  const float sc2 = p.scale * 1.4426950408889634f;

  float16v oacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;

  int kend = p.Tk;
  if (p.causal) {
    const int last = qblock + QB - 1 + off + 1;
    if (last < kend) kend = last;
  }
  TileRegs<D, NT> kreg, vreg;
  if (kend > 0) {
    tile_load<D, NT>(kreg, Kb, p.k_row, 0, p.Tk, tid);
    tile_load<D, NT>(vreg, Vb, p.v_row, 0, p.Tk, tid);
  }
  for (int j0 = 0; j0 < kend; j0 += TB) {
    tile_store_rows<D, NT>(kreg, Ks, tid);
    tile_store_rows<D, NT>(vreg, Vs, tid);
    __syncthreads();
    if (j0 + TB < kend) {                                   // the next tile's rows travel while this tile is multiplied
      tile_load<D, NT>(kreg, Kb, p.k_row, j0 + TB, p.Tk, tid);
      tile_load<D, NT>(vreg, Vb, p.v_row, j0 + TB, p.Tk, tid);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float16v sacc, dpacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f, dpacc[r] = 0.f;
      const int row = kb * 32 + ql;
      // all 2 x KSTEPS fragments of this key block first, in registers of their own, THEN the MFMAs: with one wave per SIMD
      // a ds_read -> s_waitcnt -> v_mfma chain per step (what the compiler made of the fused loop) exposes the LDS latency
      // 2 x KSTEPS times per block
      h16x8 kfr[KSTEPS], vfr[KSTEPS];
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        kfr[kk] = frag_rows<D>(Ks, row, kk, hi);
        vfr[kk] = frag_rows<D>(Vs, row, kk, hi);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        sacc = G4R_MFMA_32X32X16(kfr[kk], qf[kk], sacc, 0, 0, 0);
        dpacc = G4R_MFMA_32X32X16(vfr[kk], dof[kk], dpacc, 0, 0, 0);
      }
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = j0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const bool valid = key < p.Tk && qi < p.Tq && (!p.causal || key <= qi + off);
        const float e = __builtin_amdgcn_exp2f(sacc[r] * sc2 - lse_i);     // unconditional: a select, not a branch per element
        const float pr = valid ? e : 0.f;
        ds[r] = pr * (dpacc[r] - delta_i) * p.scale;
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const h16x8 pf = pack8(ds + hf * 8);
#pragma unroll
        for (int d = 0; d < DB; ++d)
          oacc[d] = G4R_MFMA_32X32X16(frag_tr<D>(Ks, trl, d, kb * 32 + hf * 16), pf, oacc[d], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if (qi < p.Tq) {
    h16_t* orow = p.dQ + (size_t)b * p.dq_batch + (size_t)qi * p.dq_row + (size_t)h * D;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint2v w = {pack_h16x2(oacc[d][g * 4], oacc[d][g * 4 + 1]),
                          pack_h16x2(oacc[d][g * 4 + 2], oacc[d][g * 4 + 3])};
        *reinterpret_cast<uint2v*>(orow + d * 32 + g * 8 + 4 * hi) = w;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// dK, dV: one workgroup = NW waves = 32*NW keys of one (batch, head); streams the Q / dO tiles.
// ---------------------------------------------------------------------------------------------
template <int D, int NW>
__global__ __launch_bounds__(NW * 64, 1) void attn_bwd_dkv_kernel(AttnBwdArgs p) {
  constexpr int NT = NW * 64;
  constexpr int KB = NW * 32;
  constexpr int KSTEPS = D / 16;
  constexpr int DB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char dkv_smem[];  // 32.5 KB at D = 128
  h16_t* Qs = reinterpret_cast<h16_t*>(dkv_smem);
  h16_t* Gs = Qs + TB * D;        // dO rows
  float* lse_s = reinterpret_cast<float*>(Gs + TB * D);
  float* delta_s = lse_s + TB;
  TrLane<D> trl;
  trl.init(threadIdx.x & 63);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, ql = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const int kblock = blockIdx.x * KB;
  const int kj = kblock + wave * 32 + ql;  // this lane's key
  const int off = p.Tk - p.Tq;
  const h16_t* Qb = p.Q + (size_t)b * p.q_batch + (size_t)h * D;
  const h16_t* Gb = p.dO + (size_t)b * p.do_batch + (size_t)h * D;
  const int kr = kj < p.Tk ? kj : p.Tk - 1;

  h16x8 kf[KSTEPS], vf[KSTEPS];
  {
    const h16_t* krow = p.K + (size_t)b * p.k_batch + (size_t)kr * p.k_row + (size_t)h * D + hi * 8;
    const h16_t* vrow = p.V + (size_t)b * p.v_batch + (size_t)kr * p.v_row + (size_t)h * D + hi * 8;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
      kf[kk] = *reinterpret_cast<const h16x8*>(krow + kk * 16);
      vf[kk] = *reinterpret_cast<const h16x8*>(vrow + kk * 16);
    }
  }
  const float sc2 = p.scale * 1.4426950408889634f;
  float16v dkacc[DB], dvacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dkacc[d][r] = 0.f, dvacc[d][r] = 0.f;

  int q_begin = 0;
  if (p.causal) {
    q_begin = kblock - off;  // first query that sees the first key of this block
    if (q_begin < 0) q_begin = 0;
    q_begin = (q_begin / TB) * TB;
  }
  const float* lse_b = p.lse + ((size_t)b * p.H + h) * p.Tq;
  const float* delta_b = p.delta + ((size_t)b * p.H + h) * p.Tq;
  TileRegs<D, NT> qreg, greg;
  float lse_r = 0.f, delta_r = 0.f;
  auto load_tile = [&](int i0) {
    tile_load<D, NT>(qreg, Qb, p.q_row, i0, p.Tq, tid);
    tile_load<D, NT>(greg, Gb, p.do_row, i0, p.Tq, tid);
    if (tid < TB) {
      int q = i0 + tid;
      if (q > p.Tq - 1) q = p.Tq - 1;
      lse_r = lse_b[q];
      delta_r = delta_b[q];
    }
  };
  if (q_begin < p.Tq) load_tile(q_begin);
  for (int i0 = q_begin; i0 < p.Tq; i0 += TB) {
    tile_store_rows<D, NT>(qreg, Qs, tid);
    tile_store_rows<D, NT>(greg, Gs, tid);
    if (tid < TB) {
      lse_s[tid] = lse_r;
      delta_s[tid] = delta_r;
    }
    __syncthreads();
    if (i0 + TB < p.Tq) load_tile(i0 + TB);                  // the next tile's rows travel while this tile is multiplied
    // Both 32-query blocks of the tile: S and dP of BOTH first (four independent accumulator chains), then the softmax
    // arithmetic of block 0, then the dV / dK MFMAs of block 0 in one basic block with the arithmetic of block 1 (the wave has
    // its SIMD to itself: independent VALU work is what can run under its own MFMAs), then the MFMAs of block 1.
    float16v sacc[2], dpacc[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[qb][r] = 0.f, dpacc[qb][r] = 0.f;
      const int row = qb * 32 + ql;
      h16x8 qfr[KSTEPS], gfr[KSTEPS];                 // fragments first, MFMAs after (see attn_bwd_dq_kernel)
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        qfr[kk] = frag_rows<D>(Qs, row, kk, hi);
        gfr[kk] = frag_rows<D>(Gs, row, kk, hi);
      }
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        sacc[qb] = G4R_MFMA_32X32X16(qfr[kk], kf[kk], sacc[qb], 0, 0, 0);
        dpacc[qb] = G4R_MFMA_32X32X16(gfr[kk], vf[kk], dpacc[qb], 0, 0, 0);
      }
    }
    h16x8 pfr[2][2], sfr[2][2];
    auto softmax_block = [&](int qb) {
      float pr[16], ds[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int il = qb * 32 + 8 * g + 4 * hi;
        const float4v l4 = *reinterpret_cast<const float4v*>(lse_s + il);
        const float4v d4 = *reinterpret_cast<const float4v*>(delta_s + il);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = g * 4 + e;
          const int qi = i0 + il + e;
          const bool valid = qi < p.Tq && kj < p.Tk && (!p.causal || kj <= qi + off);
          const float ex = __builtin_amdgcn_exp2f(sacc[qb][r] * sc2 - l4[e]);
          pr[r] = valid ? ex : 0.f;
          ds[r] = pr[r] * (dpacc[qb][r] - d4[e]) * p.scale;
        }
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        pfr[qb][hf] = pack8(pr + hf * 8);
        sfr[qb][hf] = pack8(ds + hf * 8);
      }
    };
    auto grad_block = [&](int qb) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int ibase = qb * 32 + hf * 16;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          dvacc[d] = G4R_MFMA_32X32X16(frag_tr<D>(Gs, trl, d, ibase), pfr[qb][hf], dvacc[d], 0, 0, 0);
          dkacc[d] = G4R_MFMA_32X32X16(frag_tr<D>(Qs, trl, d, ibase), sfr[qb][hf], dkacc[d], 0, 0, 0);
        }
      }
    };
    softmax_block(0);
    grad_block(0);
    softmax_block(1);
    grad_block(1);
    __syncthreads();
  }
  if (kj < p.Tk) {
    h16_t* krow = p.dK + (size_t)b * p.dk_batch + (size_t)kj * p.dk_row + (size_t)h * D;
    h16_t* vrow = p.dV + (size_t)b * p.dv_batch + (size_t)kj * p.dv_row + (size_t)h * D;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint2v wk = {pack_h16x2(dkacc[d][g * 4], dkacc[d][g * 4 + 1]),
                           pack_h16x2(dkacc[d][g * 4 + 2], dkacc[d][g * 4 + 3])};
        const uint2v wv = {pack_h16x2(dvacc[d][g * 4], dvacc[d][g * 4 + 1]),
                           pack_h16x2(dvacc[d][g * 4 + 2], dvacc[d][g * 4 + 3])};
        *reinterpret_cast<uint2v*>(krow + d * 32 + g * 8 + 4 * hi) = wk;
        *reinterpret_cast<uint2v*>(vrow + d * 32 + g * 8 + 4 * hi) = wv;
      }
  }
}

template <int D>
int launch_bwd(const AttnBwdArgs& a, int B, hipStream_t stream) {
  {
    const long total = (long)B * a.Tq * a.H * (D / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((attn_delta_kernel<D>), dim3(blocks), dim3(256), 0, stream, a, B);
    G4R_CHECK_LAUNCH("attn_delta");
  }
  // D = 128: 4 waves share the staging of a tile (measured: 2-wave workgroups re-stage twice as much and run
  // 240 us vs 186 us per LLaMA layer at T = 767); D = 64: the staging split needs 16 * D / 8 >= threads
  constexpr int NW = D == 128 ? 4 : 2;
  hipLaunchKernelGGL((attn_bwd_dq_kernel<D, NW>), dim3(g4r_ceil_div(a.Tq, NW * 32), a.H, B), dim3(NW * 64), 0,
                     stream, a);
  G4R_CHECK_LAUNCH("attn_bwd_dq");
  constexpr int DKV_LDS = 2 * TB * D * 2 + 2 * TB * 4;
  static G4rPerDeviceOnce attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<D, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
    if (e != hipSuccess) { attr_set.failed(); return g4r_note_hip_error(e, "attn_bwd_dkv: hipFuncSetAttribute"); }
  }
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, NW>), dim3(g4r_ceil_div(a.Tk, NW * 32), a.H, B), dim3(NW * 64), DKV_LDS,
                     stream, a);
  G4R_CHECK_LAUNCH("attn_bwd_dkv");
  return G4R_OK;
}

}  // namespace

extern "C" {

int g4r_flash_attn_bwd_bf16(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                            const float* lse, float* delta, void* dQ, void* dK, void* dV, int B, int H, int Tq,
                            int Tk, int head_dim, long q_row, long k_row, long v_row, long o_row, long do_row,
                            long dq_row, long dk_row, long dv_row, long q_batch, long k_batch, long v_batch,
                            long o_batch, long do_batch, long dq_batch, long dk_batch, long dv_batch,
                            float scale, int causal, void* stream) {
  G4R_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "flash_attn_bwd: bad shape");
  G4R_REQUIRE(head_dim == 64 || head_dim == 128, "flash_attn_bwd: head_dim must be 64 or 128");
  G4R_REQUIRE(Q && K && V && O && dO && lse && delta && dQ && dK && dV, "flash_attn_bwd: null pointer");
  G4R_REQUIRE(!causal || Tk >= Tq, "flash_attn_bwd: causal needs Tk >= Tq");
  const long strides[] = {q_row,   k_row,   v_row,   o_row,   do_row,   dq_row,   dk_row,   dv_row,
                          q_batch, k_batch, v_batch, o_batch, do_batch, dq_batch, dk_batch, dv_batch};
  for (long s : strides) G4R_REQUIRE(s % 8 == 0, "flash_attn_bwd: strides must keep 16-byte alignment");
  AttnBwdArgs a = {(const h16_t*)Q, (const h16_t*)K, (const h16_t*)V, (const h16_t*)O, (const h16_t*)dO,
                   (h16_t*)dQ, (h16_t*)dK, (h16_t*)dV, lse, delta,
                   q_row, k_row, v_row, o_row, do_row, dq_row, dk_row, dv_row,
                   q_batch, k_batch, v_batch, o_batch, do_batch, dq_batch, dk_batch, dv_batch,
                   Tq, Tk, H, scale, causal};
  if (head_dim == 64) return launch_bwd<64>(a, B, (hipStream_t)stream);
  return launch_bwd<128>(a, B, (hipStream_t)stream);
}

}  // extern "C"
