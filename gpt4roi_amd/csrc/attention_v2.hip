// attention_v2.hip -- the prefill attention forward of the path, second form (round 3).
//
//   CLIP ViT-L/14 self-attention  (16 heads x 64, S = P^2+1, bidirectional)     SURVEY 8a row a4
//   LLaMA-7B self-attention       (32 heads x 128, causal, KV cache)            row a16 (region tokens x KV)
// Reference arithmetic: HF transformers attention (spi_llava.py:66-67, 198-205) / flash-attn
// (llava/train/llama_flash_attn_monkey_patch.py:15-91): softmax(Q K^T * scale [+ causal mask]) V.
//
// What was wrong with the first form (attention.hip, kept for Tq < 32 and as the A/B arm): one wave per SIMD ran the K/V
// staging (global -> registers -> LDS, V transposed by 8-byte stores), two barriers, the softmax and the 32 MFMAs of a
// 64-key tile back to back (~2.4 us per tile against 0.5 us of MFMA), and at T = 767 the longest workgroup walks 12 tiles
// while the shortest walks 2.  Here:
//   * a workgroup = NG key GROUPS x NWG waves.  All groups own the SAME 32 x NWG query rows; group g takes the key tiles
//     g, g + NG, g + 2 NG, ... (flash-decoding inside the workgroup).  The critical path of the longest query block
//     drops from 12 tiles to 12 / NG steps, and every SIMD hosts NG waves whose softmax / MFMA / LDS phases overlap.
//     The NG partial states (m, l, O) meet once, through LDS, at the end.
//   * K and V tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB pieces): no staging registers, no
//     ds_write pass, next tile in flight during the current tile's compute (double buffer, one barrier per step).
//     K image: row-major [64 keys][D] with the 16-byte-slot XOR swizzle applied to the per-lane SOURCE address and to
//     the ds_read_b128 address ("both sides or neither").  V image: [D/16 d-blocks][64 keys][16 d] (32-byte rows,
//     d-blocks 2176 B apart so that the two 16-lane groups of a half-wave hit different banks), read with
//     ds_read_b64_tr_b16: a 16-lane group fetches a [4 keys][16 d] block and each lane receives one COLUMN of it = 4
//     keys of its own d -- the V^T fragment of the MFMA A operand with no transposing store anywhere.
//   * register mapping as in the first form: S^T = K Q^T with v_mfma_f32_32x32x16_bf16 so a lane owns one query
//     column (its 32 scores, its running max / sum, its O^T registers); the key order inside a 16-key MFMA step is
//     permuted to the order the lane already holds P in.
#include "g4r_common.h"

namespace {


struct Attn2Args {
  const h16_t* Q;
  const h16_t* K;
  const h16_t* V;
  h16_t* O;
  long q_row, k_row, v_row, o_row;
  long q_batch, k_batch, v_batch, o_batch;
  int Tq, Tk, H;
  float scale;
  int causal;
  const int* kv_len_dev;
  float* lse;
  long long* probe;   // tools only: s_memtime stamps of step 2 of workgroup (0, 0, 0), waves 0 and NWG ([group][16])
};

constexpr int A2_KVB = 64;        // keys per tile
constexpr int A2_VSUB = 64 * 32 + 128;   // bytes of one [64 keys][16 d] V sub-image + the bank offset pad

// value of the lane 32 away (the other half-wave), without the LDS crossbar: __shfl_xor(x, 32) compiles to ds_bpermute_b32, an
// LDS-queue instruction, and an LDS-queue instruction issued by a wave behind its own LDS-DMA pieces waits until they have
// landed (tools/attn_probe2.py: the row-max exchange right after the piece issue cost 600-1200 cycles per step).
// v_permlane32_swap is VALU: {x_lo, x_lo} / {x_hi, x_hi} from one instruction.
__device__ __forceinline__ float a2_other_half(float x, bool want_max) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const float lo = __uint_as_float(r[0]), hi = __uint_as_float(r[1]);
  return want_max ? fmaxf(lo, hi) : lo + hi;
}

// one 1 KiB LDS-DMA piece as buffer_load_dwordx4 ... lds (a free __device__ function: written inside a lambda of the kernel
// the builtins make hipcc drop the kernel's host handle, see gemm_bf16.hip g4r_buffer_piece)
__device__ __forceinline__ void a2_buffer_piece(const void* base, unsigned bytes, void* lds, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

template <int D>
__device__ __forceinline__ int a2_kswz(int row) { return D == 128 ? (row & 15) : ((row >> 1) & 7); }

template <int D>
constexpr int a2_tile_bytes() { return A2_KVB * D * 2 + (D / 16) * A2_VSUB; }

// PP (ping-pong): a step is two phases with a barrier after each, and odd groups run one phase behind even groups:
//     phase A = the 16 QK^T MFMAs, the LDS-DMA pieces of the group's next tile, then scale / mask / row max (VALU)
//     phase B = exp2 / row sum / P -> bf16 / O rescale (VALU), then the 16 PV MFMAs
// so that on every SIMD one wave's MFMA burst runs beside the other wave's VALU burst and vice versa (the first version
// had every wave in the same phase at the same time: QK, softmax and PV of both waves of a SIMD simply added up, ~2.3 us
// per step against 1 us of MFMA).  All LDS tile buffers are group-private, so the offset adds no hazard: a group's next
// tile is issued in its phase A (its buffer was last read in the group's phase A / B of the previous step, two barriers
// ago) and waited for at the end of its phase B.
template <int D, int NWG, int NG, bool PP>
__global__ __launch_bounds__(NWG* NG * 64) void flash_attn_fwd2_kernel(Attn2Args p) {
  constexpr int QB = NWG * 32;
  constexpr int SLOTS = D / 8, KSTEPS = D / 16, DB = D / 32, NDD = D / 16;
  constexpr int K_BYTES = A2_KVB * D * 2, TILE_BYTES = a2_tile_bytes<D>();
  constexpr int KP = K_BYTES / 1024, VP = NDD * 2;        // 1 KiB pieces of a K / V tile
  constexpr int PPW = (KP + VP) / NWG, KPW = KP / NWG;    // pieces per wave: the first KPW are K pieces
  constexpr int RPP = 1024 / (D * 2);                     // key rows per K piece
  constexpr int NWAVES = NWG * NG;
  constexpr int QP = QB * D * 2 / 1024;                   // 1 KiB pieces of the Q tile
  constexpr int QPW = (QP + NWAVES - 1) / NWAVES;
  static_assert(KP % NWG == 0 && VP % NWG == 0, "piece split");
  static_assert(QB * D * 2 <= TILE_BYTES, "the Q tile is staged in one tile buffer");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NG groups x 2 buffers x TILE_BYTES (one array: no second
                                                                // __shared__ object beside an LDS-DMA pipeline)
  if (p.kv_len_dev) p.Tk = *p.kv_len_dev + p.Tq;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave / NWG, wv = wave % NWG;
  const int hi = lane >> 5, ql = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qblock = ((int)gridDim.x - 1 - (int)blockIdx.x) * QB;     // heaviest (latest) causal blocks first
  const int qw0 = qblock + wv * 32;
  const int qi = qw0 + ql;
  const int off = p.Tk - p.Tq;
  const h16_t* Qb = p.Q + (size_t)b * p.q_batch + (size_t)h * D;
  const h16_t* Kb = p.K + (size_t)b * p.k_batch + (size_t)h * D;
  const h16_t* Vb = p.V + (size_t)b * p.v_batch + (size_t)h * D;
  char* gbuf = smem + grp * (2 * TILE_BYTES);

  int kend = p.Tk;
  if (p.causal) {
    const int last = qblock + QB - 1 + off + 1;
    if (last < kend) kend = last;
  }
  const int ntiles = (kend + A2_KVB - 1) / A2_KVB;
  const int nsteps = (ntiles + NG - 1) / NG;

  // ---- this lane's part of the wave's pieces.  Buffer form (round 3b): a 32-bit per-lane byte offset computed ONCE and a
  //      scalar tile offset per step -- one SALU add (M0) + one VMEM instruction per piece.  The global_load_lds form
  //      rebuilt a clamped 64-bit address per piece per step (two v_mul_lo, a v_mad_u64, ...: ~100 cycles per piece,
  //      800-1100 per step and wave, tools/attn_probe2.py). ----
  const int k_row_b = (int)p.k_row * 2, v_row_b = (int)p.v_row * 2;       // row pitches in bytes
  const long k_span = ((long)(p.Tk - 1) * p.k_row + D) * 2, v_span = ((long)(p.Tk - 1) * p.v_row + D) * 2;
  if (k_span > 0x7fffffffL || v_span > 0x7fffffffL) __builtin_trap();    // the host routes such tensors to the first form
  int pvoff[PPW], pdst[PPW];
  auto piece_row = [&](int j) {                                           // key row (inside the tile) of piece j of this lane
    return j < KPW ? (wv + NWG * j) * RPP + lane / SLOTS : ((wv + NWG * (j - KPW)) & 1) * 32 + (lane >> 1);
  };
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    if (j < KPW) {
      const int pk = wv + NWG * j;
      const int row = pk * RPP + lane / SLOTS;
      pvoff[j] = row * k_row_b + (((lane % SLOTS) ^ a2_kswz<D>(row)) * 8) * 2;
      pdst[j] = pk * 1024;
    } else {
      const int pv = wv + NWG * (j - KPW);
      const int dd = pv >> 1, half = pv & 1;
      pvoff[j] = (half * 32 + (lane >> 1)) * v_row_b + (dd * 16 + (lane & 1) * 8) * 2;
      pdst[j] = K_BYTES + dd * A2_VSUB + half * 1024;
    }
  }
  auto issue = [&](int tile, int buf) {
    const int j0 = tile * A2_KVB;
    char* dst = gbuf + buf * TILE_BYTES;
    if (j0 + A2_KVB <= p.Tk) {                       // whole tile inside the keys (wave-uniform): scalar tile offset
      const int ks = j0 * k_row_b, vs = j0 * v_row_b;
#pragma unroll
      for (int j = 0; j < PPW; ++j)
        a2_buffer_piece(j < KPW ? Kb : Vb, (unsigned)(j < KPW ? k_span : v_span), dst + pdst[j], pvoff[j], j < KPW ? ks : vs);
    } else {                                         // ragged last tile: rows beyond the keys re-read row Tk - 1 (masked later)
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const int r = piece_row(j);
        int key = j0 + r;
        if (key > p.Tk - 1) key = p.Tk - 1;
        a2_buffer_piece(j < KPW ? Kb : Vb, (unsigned)(j < KPW ? k_span : v_span), dst + pdst[j],
                        pvoff[j] + (key - r) * (j < KPW ? k_row_b : v_row_b), 0);
      }
    }
  };

  // per-lane LDS offsets of the fragment reads
  const int k_row_off = ql * (D * 2);
  const int k_sw = a2_kswz<D>(ql);      // rows ql and ql + 32 share the swizzle (32 = 0 mod 16; (32 >> 1) = 0 mod 8)
  const int v_lane_off = K_BYTES + ((lane >> 4) & 1) * A2_VSUB + hi * 128 + (lane & 15) * 8;

  // ---- Q tile: staged once by LDS-DMA (whole 256 / 128-byte rows per 16 / 8 lanes instead of one strided row per lane)
  //      into group 0's second tile buffer, in the K image (row-major, swizzled slots), then read as B fragments ----
  char* qlds = smem + TILE_BYTES;
#pragma unroll
  for (int j = 0; j < QPW; ++j) {
    const int pq = wave + NWAVES * j;
    if (pq < QP) {
      const int row = pq * RPP + lane / SLOTS;
      int qr = qblock + row;
      if (qr > p.Tq - 1) qr = p.Tq - 1;
      const h16_t* src = Qb + (size_t)qr * p.q_row + ((lane % SLOTS) ^ a2_kswz<D>(row)) * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(qlds + pq * 1024), 16, 0, 0);
    }
  }
  if (grp < ntiles) issue(grp, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);                     // vmcnt(0); builtin, not inline asm: the compiler's counter model sees it
  __builtin_amdgcn_s_barrier();
  h16x8 qf[KSTEPS];
  {
    const char* qrow = qlds + (wv * 32) * (D * 2) + k_row_off;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) qf[kk] = *reinterpret_cast<const h16x8*>(qrow + (((kk * 2 + hi) ^ k_sw) << 4));
  }
  float16v oacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sc2 = p.scale * 1.4426950408889634f;
  __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();            // every wave holds its Q fragments: the staging buffer may be overwritten

  // state carried from phase A to phase B of a step
  float16v sacc[2];
  float m_use = 0.f, alpha = 1.f, m_new = -INFINITY, rs_a = 0.f;
  bool active = false;
  // tools only (tools/attn_probe2.py; build with G4R_EXTRA_HIPCC_FLAGS=-DG4R_ATTN2_PROBE): s_memtime stamps of step 2
#ifdef G4R_ATTN2_PROBE
  const bool probing = p.probe != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && wv == 0 && lane == 0;
  long long* stamps = p.probe ? p.probe + grp * 16 : nullptr;
#define A2_STAMP(slot, s_, dep)                                                       \
  if (p.probe) {                                                                      \
    float tmp_;                                                                       \
    asm volatile("v_mov_b32 %0, %1" : "=v"(tmp_) : "v"(dep));                          \
    if (probing && (s_) == 2) stamps[slot] = __builtin_amdgcn_s_memtime();            \
  }
#else
#define A2_STAMP(slot, s_, dep)
#endif
  // Both phases read ALL their LDS fragments before the first MFMA that needs them (a scheduling fence pins the order): the
  // compiler's own order was read -> s_waitcnt lgkmcnt(0) -> MFMA, sixteen times per phase, i.e. an LDS round trip in front
  // of every MFMA of the first chain (QK: 1076 cycles for 16 MFMAs = 512 with the other wave of the SIMD in its VALU phase).
  auto phase_a = [&](int s) {
    A2_STAMP(0, s, m_run);
    const int tile = s * NG + grp;
    const int buf = s & 1;
    const int j0 = tile * A2_KVB;
    // a wave skips tiles that are beyond the keys (ragged tail of the split) or entirely above its causal diagonal
    active = tile < ntiles && !(p.causal && j0 > qw0 + 31 + off);
    const char* kt = gbuf + buf * TILE_BYTES;
    if (active) {
      // The 2 x KSTEPS K fragments go out back to back and every MFMA pair waits for ITS two reads only (counted lgkmcnt):
      // QK = max(LDS time, MFMA time) instead of their sum.  The reads are inline asm because the compiler, given plain
      // loads, puts s_waitcnt lgkmcnt(0) in front of every MFMA that consumes one (never a counted wait on ds_read_b128
      // here), i.e. the first MFMA waited for all sixteen reads: 1076-1108 cycles for 16 MFMAs = 512.  Each wait is tied
      // ("+v") to the fragments it releases, so the MFMA cannot move above it.
      h16x8 kf[2][KSTEPS];
      const unsigned kbase = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(kt + k_row_off);
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        const unsigned ka = kbase + ((unsigned)((kk * 2 + hi) ^ k_sw) << 4);
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0][kk]) : "v"(ka) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[1][kk]) : "v"(ka), "n"(32 * D * 2) : "memory");
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {            // two independent accumulator chains, interleaved
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(kf[0][kk]), "+v"(kf[1][kk]) : "n"(2 * (KSTEPS - 1 - kk)));
        sacc[0] = G4R_MFMA_32X32X16(kf[0][kk], qf[kk], sacc[0], 0, 0, 0);
        sacc[1] = G4R_MFMA_32X32X16(kf[1][kk], qf[kk], sacc[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);             // ... and the next wait cannot move above it
      }
    }
    if (active) { A2_STAMP(1, s, sacc[1][15]); }
    A2_STAMP(2, s, m_run);
    if (active) {
      // online softmax, first half: masks and the row max of the RAW scores; the softmax scale (log2 domain) is applied to
      // the max here and to the scores inside the exp2 argument (one fused multiply-add per score in phase B)
      const bool need_mask = (j0 + A2_KVB > p.Tk) || (p.causal && j0 + A2_KVB - 1 > qw0 + off);
      float mt0 = -INFINITY, mt1 = -INFINITY;
      if (need_mask) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = j0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= p.Tk || (p.causal && key > qi + off)) sacc[kb][r] = -INFINITY;
          }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        mt0 = fmaxf(mt0, sacc[0][r]);
        mt1 = fmaxf(mt1, sacc[1][r]);
      }
      const float mt = a2_other_half(fmaxf(mt0, mt1), true);
      m_new = fmaxf(m_run, mt * sc2);
      m_use = m_new == -INFINITY ? 0.f : m_new;
      alpha = __builtin_amdgcn_exp2f(m_run - m_use);
      // the exponentials of the first 32 keys belong to phase A: with them the two phases carry about the same VALU time
      // (A: QK MFMAs, max, 16 exp; B: 16 exp, sums, rescale, PV MFMAs), and on a SIMD one wave's MFMAs face the other's VALU
      const float nm = -m_use;
      float r0 = 0.f, r1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[0][r], sc2, nm));
        const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[0][r + 1], sc2, nm));
        sacc[0][r] = e0;
        sacc[0][r + 1] = e1;
        r0 += e0;
        r1 += e1;
      }
      rs_a = r0 + r1;
    }
    A2_STAMP(3, s, alpha);
  };
  auto v_frag = [&](const char* vt, int kb, int hf, int d) {
    const char* a = vt + d * 2 * A2_VSUB + (kb * 32 + hf * 16) * 32;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(a));
    const short4v up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(a + 8 * 32));
    const short8 vv = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
    return __builtin_bit_cast(h16x8, vv);
  };
  auto phase_b = [&](int s) {
    A2_STAMP(4, s, alpha);
    const int tile = s * NG + grp;
    const char* vt = gbuf + (s & 1) * TILE_BYTES + v_lane_off;
    // every V^T fragment of the tile goes out BEFORE the exponentials, whose VALU time covers the LDS round trip; the pieces
    // of the group's NEXT tile are issued behind them: the compiler puts s_waitcnt vmcnt(0) in front of any LDS read that
    // follows an LDS-DMA instruction in program order (it cannot see that the DMA writes the other buffer), so the pieces
    // must come after the step's last LDS read.  They land during exp / PV / the barrier (~1400 cycles).
    h16x8 vf[2][2][DB];
    if (active) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int d = 0; d < DB; ++d) vf[kb][hf][d] = v_frag(vt, kb, hf, d);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tile + NG < ntiles) issue(tile + NG, (s & 1) ^ 1);     // every wave carries its share of the pieces
    A2_STAMP(9, s, m_run);
    if (!active) return;
    float rs0 = 0.f, rs1 = 0.f;
    const float nm = -m_use;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[1][r], sc2, nm));
      const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[1][r + 1], sc2, nm));
      sacc[1][r] = e0;
      sacc[1][r + 1] = e1;
      rs0 += e0;
      rs1 += e1;
    }
    const float rs = a2_other_half(rs_a + (rs0 + rs1), false);
    l_run = l_run * alpha + rs;
    m_run = m_new;
    if (!__all(alpha == 1.f)) {          // the running max moved for some row of this wave: rescale O (wave-uniform branch)
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
    }
    A2_STAMP(5, s, oacc[0][0]);
    // O^T += V^T P^T : A operand = 4 + 4 keys of this lane's d through two transpose reads
    h16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        uint4v pw;
        pw.x = pack_h16x2(sacc[kb][hf * 8 + 0], sacc[kb][hf * 8 + 1]);
        pw.y = pack_h16x2(sacc[kb][hf * 8 + 2], sacc[kb][hf * 8 + 3]);
        pw.z = pack_h16x2(sacc[kb][hf * 8 + 4], sacc[kb][hf * 8 + 5]);
        pw.w = pack_h16x2(sacc[kb][hf * 8 + 6], sacc[kb][hf * 8 + 7]);
        pf[kb][hf] = __builtin_bit_cast(h16x8, pw);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int d = 0; d < DB; ++d)
          oacc[d] = G4R_MFMA_32X32X16(vf[kb][hf][d], pf[kb][hf], oacc[d], 0, 0, 0);
    A2_STAMP(6, s, oacc[DB - 1][15]);
  };

  if (PP) {
    if (grp & 1) __builtin_amdgcn_s_barrier();             // odd groups run one phase behind
    for (int s = 0; s < nsteps; ++s) {
      phase_a(s);
      __builtin_amdgcn_s_barrier();
      phase_b(s);
      __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0) lgkmcnt(0): the group's next tile has landed.  The
                                                           // builtin, not inline asm: the compiler's counter model sees it
      A2_STAMP(7, s, m_run);
      __builtin_amdgcn_s_barrier();
      A2_STAMP(8, s, m_run);
    }
    if (!(grp & 1)) __builtin_amdgcn_s_barrier();
  } else {
    for (int s = 0; s < nsteps; ++s) {
      phase_a(s);
      phase_b(s);
      __builtin_amdgcn_s_waitcnt(0x0070);
      A2_STAMP(7, s, m_run);
      __builtin_amdgcn_s_barrier();                        // everyone's pieces landed; everyone is done with this step's buffers
      A2_STAMP(8, s, m_run);
    }
  }

  // ---- merge the NG partial states through LDS (the tile buffers are free: every wave passed the last barrier) ----
  constexpr int REGS = DB * 16 + 2;                        // O^T registers + m + l, per (group - 1, wave, lane)
  constexpr int MERGE_BYTES = (NG - 1) * NWG * REGS * 64 * 4;
  if (NG > 1) {
    float* mo = reinterpret_cast<float*>(smem);
    if (grp > 0) {
      float* dst = mo + ((size_t)((grp - 1) * NWG + wv) * REGS) * 64 + lane;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(d * 16 + r) * 64] = oacc[d][r];
      dst[(DB * 16) * 64] = m_run;
      dst[(DB * 16 + 1) * 64] = l_run;
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < NG; ++g) {
      const float* src = mo + ((size_t)((g - 1) * NWG + wv) * REGS) * 64 + lane;
      const float m_o = src[(DB * 16) * 64], l_o = src[(DB * 16 + 1) * 64];
      const float mm = fmaxf(m_run, m_o);
      const float mu = mm == -INFINITY ? 0.f : mm;
      const float a_me = __builtin_amdgcn_exp2f(m_run - mu), a_o = __builtin_amdgcn_exp2f(m_o - mu);
      l_run = l_run * a_me + l_o * a_o;
      m_run = mm;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = oacc[d][r] * a_me + src[(d * 16 + r) * 64] * a_o;
    }
  }

  // ---- normalise; O through LDS so that a wave instruction stores whole rows (a lane owns ONE query row: stored directly
  //      it wrote 32 rows x 16 bytes per instruction) ----
  if (p.lse && hi == 0 && qi < p.Tq) p.lse[((size_t)b * p.H + h) * p.Tq + qi] = m_run + __log2f(l_run);
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  constexpr int ORS = D * 2 + 16;                          // row stride of the O staging image (bytes)
  char* olds = smem + MERGE_BYTES + (size_t)wv * 32 * ORS;  // this wave's 32 rows (behind the merge area)
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint2v w = {pack_h16x2(oacc[d][g * 4] * inv, oacc[d][g * 4 + 1] * inv),
                        pack_h16x2(oacc[d][g * 4 + 2] * inv, oacc[d][g * 4 + 3] * inv)};
      *reinterpret_cast<uint2v*>(olds + ql * ORS + (d * 32 + g * 8 + 4 * hi) * 2) = w;
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): same wave wrote and reads, in-order LDS, no barrier needed
  constexpr int LPR = D / 8;                               // lanes per output row (16 bytes each)
  constexpr int RPI = 64 / LPR;                            // rows per wave instruction
#pragma unroll
  for (int it = 0; it < 32 / RPI; ++it) {
    const int r = it * RPI + lane / LPR;
    const int qrow = qw0 + r;
    const uint4v w = *reinterpret_cast<const uint4v*>(olds + r * ORS + (lane % LPR) * 16);
    if (qrow < p.Tq)
      *reinterpret_cast<uint4v*>(p.O + (size_t)b * p.o_batch + (size_t)qrow * p.o_row + (size_t)h * D + (lane % LPR) * 8) = w;
  }
}

template <int D, int NWG, int NG, bool PP = true>
int launch_attn2(const Attn2Args& a, int B, hipStream_t st) {
  constexpr int LDS = NG * 2 * a2_tile_bytes<D>();
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(LDS >= (NG - 1) * NWG * (D / 32 * 16 + 2) * 64 * 4 + NWG * 32 * (D * 2 + 16),
                "merge area + O staging fit the tile buffers");
  auto kfn = flash_attn_fwd2_kernel<D, NWG, NG, PP>;
  static G4rPerDeviceOnce attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) { attr_set.failed(); return g4r_note_hip_error(e, "flash_attn_fwd2: hipFuncSetAttribute"); }
  }
  dim3 grid(g4r_ceil_div(a.Tq, NWG * 32), a.H, B);
  hipLaunchKernelGGL(kfn, grid, dim3(NWG * NG * 64), LDS, st, a);
  return G4R_OK;
}


}  // namespace

static long long* g_attn2_probe = nullptr;   // tools: device buffer for the phase stamps (tools/attn_probe2.py)
#ifndef G4R_F16
extern "C" void g4r_attn2_debug_probe(void* ptr) { g_attn2_probe = (long long*)ptr; }
#endif

// variant: 0 = production choice; otherwise NWG * 10 + NG of an instantiated form (tools / tests)
int g4r_attn2_dispatch(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq, int Tk, int head_dim,
                       long q_row, long k_row, long v_row, long o_row, long q_batch, long k_batch, long v_batch,
                       long o_batch, float scale, int causal, const int* kv_len_dev, float* lse, int variant,
                       void* stream) {
  Attn2Args a = {(const h16_t*)Q, (const h16_t*)K, (const h16_t*)V, (h16_t*)O, q_row, k_row, v_row, o_row,
                 q_batch, k_batch, v_batch, o_batch, Tq, Tk, H, scale, causal, kv_len_dev, lse, g_attn2_probe};
  hipStream_t st = (hipStream_t)stream;
  int rc = G4R_OK;
  // variant = NWG * 10 + NG; + 100 = the lock-step form (one barrier per step), which is what production runs: the
  // phase-offset form measures the same or slower (T = 767: 21.2 vs 21.1 us; ViT S = 577: 10.4 vs 9.5 us; T = 2048: 81.4 vs
  // 79.8, profiles/r03_attention_stamps.txt) -- the step is bound by the LDS reads of K and V, which both forms issue alike
  if (head_dim == 128) {
    if (variant == 0) variant = 142;
    if (variant == 42) rc = launch_attn2<128, 4, 2>(a, B, st);
    else if (variant == 142) rc = launch_attn2<128, 4, 2, false>(a, B, st);
    else if (variant == 41) rc = launch_attn2<128, 4, 1, false>(a, B, st);
    else return g4r_note_error(G4R_ERR_INVALID_ARG, "flash_attn_fwd2: unknown variant for head_dim 128");
  } else {
    if (variant == 0) {
      // few workgroups (the batch-1 ViT: 16 heads x 577 rows): 64-row blocks x 4 key groups; otherwise 128-row blocks
      const long wgs128 = (long)g4r_ceil_div(Tq, 128) * H * B;
      variant = wgs128 < 256 ? 124 : 142;
    }
    if (variant == 24) rc = launch_attn2<64, 2, 4>(a, B, st);
    else if (variant == 124) rc = launch_attn2<64, 2, 4, false>(a, B, st);
    else if (variant == 42) rc = launch_attn2<64, 4, 2>(a, B, st);
    else if (variant == 142) rc = launch_attn2<64, 4, 2, false>(a, B, st);
    else if (variant == 41) rc = launch_attn2<64, 4, 1, false>(a, B, st);
    else return g4r_note_error(G4R_ERR_INVALID_ARG, "flash_attn_fwd2: unknown variant for head_dim 64");
  }
  if (rc != G4R_OK) return rc;
  G4R_CHECK_LAUNCH("flash_attn_fwd2");
  return G4R_OK;
}
