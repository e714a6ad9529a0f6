// gemm_bf16.hip -- bf16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (CDNA4).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )          (nn.Linear layout: W is [out, in])
//
// One kernel template serves every dense contraction of the region-feature path
// (SURVEY.md section 8a rows a4, a7, a8, a14, a16 -- the arithmetic the reference delegates to
// cuBLAS/cuDNN through torch.nn.Linear / nn.Conv2d, e.g. gpt4roi/models/layers.py:129-144,
// 257-270 and llava/model/llava.py:52):
//   AMODE 0  dense A [M, lda]
//   AMODE 1  implicit GEMM of a 3x3 / pad 1 / stride 1 convolution over NHWC activations:
//            row m = pixel (b, y, x), K index = ((group * 9 + tap) * Cin + c), walked taps-fastest;
//            out-of-image taps read a zero line.  `groups` > 1 sums several convolutions in one accumulator
//            (the "sum_l pconv_l(roi_feats[l])" of layers.py:321-324).
//
// Machine mapping (MI355X_MICROARCH.md / cdna_hip_programming.md section 5):
//   * v_mfma_f32_32x32x16_bf16, operands swapped (W as the "A" operand) so that a lane's 4
//     consecutive accumulator registers are 4 consecutive N columns -> 8/16-byte stores;
//   * BK = 64; tiles staged global->LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPR round
//     trip), double buffered, one barrier per K tile;
//   * the LDS image is lane-linear for the DMA, so the bank-conflict swizzle
//     (16-B slot ^= (row>>1)&7) is applied on the per-lane SOURCE address and again on the
//     ds_read_b128 address (rule "both sides or neither");
//   * XCD-aware, bijective workgroup -> tile map so neighbouring tiles share panels in one L2.
#include <type_traits>

#include "g4r_common.h"

namespace {


struct GemmArgs {
  const h16_t* A;
  const h16_t* W;
  void* C;               // bf16 or f32 [M, ldc]
  float* ws;             // split-K partials [splits][M][N] (fp32), or null
  const float* bias;     // [N] fp32 or null
  const h16_t* residual;  // [M, ldr] bf16 or null
  const h16_t* zeros;   // >= 128 B of zeros (AMODE 1 padding source)
  long a_group_stride;   // AMODE 1: elements between groups
  int M, N, K;
  int lda, ldw, ldc, ldr;
  int act;               // 0 none, 1 relu, 2 quick_gelu (x*sigmoid(1.702x)), 3 silu,
                         // 4 swiglu: columns are interleaved (gate, up) pairs, C has N/2 columns
  int out_f32;
  int splits, tiles_per_split;  // K tiles (of 64) per z slice
  int H, Wd, Cin, groups;      // AMODE 1 geometry
  int lvl_start[5];            // AMODE 2: first row of each level's maps (lvl_start[n_lvl] = M), levels stacked [level][b][y][x]
  int lvl_h[4], lvl_w[4];      // AMODE 2: map size per level
  int n_lvl;
  int tiles_m, tiles_n;
  int n_fastest;  // tile order: 1 = consecutive workgroups walk N first (share the A / activation tile)
  int group_m;    // > 0: grouped tile order (g4r_tile_coords), row tiles per group
  int dbg;  // ablation probe (tools only): 1 = skip the loads after the first tile, 2 = skip the MFMAs
  unsigned a_bytes, w_bytes;   // extent of the A / W operands in bytes when < 2 GiB (buffer descriptors), else 0
  int defer_reduce;            // split-K: leave the fp32 partials in ws, the CALLER's next kernel combines them
  int stagger_ticks;           // one-wave-per-SIMD kernel: start offset step of the first 256 workgroups in 10 ns ticks (0 = none)
  // persistent form (gemm_bf16_w4k64p_kernel): extents in bytes of the tensors its epilogue addresses through buffer descriptors
  unsigned c_bytes, r_bytes, rq_bytes, rkv_bytes;
  int persist_any_k;           // tile_cfg 36: the persistent form whatever K (tests: one / two / three K tiles per output tile)
  // act == 5 (fused q|k|v projection of a LLaMA layer, ring ping-pong tiles only): RoPE and the KV-cache append happen in
  // the epilogue -- what g4r_rope_qkv_bf16 did in a launch of its own.  Columns [0, HD) -> rotated q rows of rope_q,
  // [HD, 2 HD) -> rotated k into the cache rows pos0 + t, [2 HD, 3 HD) -> v into the cache.  Row m = b * rope_T + t.
  h16_t* rope_q;              // [M][HD]
  h16_t* rope_k;              // cache base of this layer: + b * rope_kbatch + (pos0 + t) * rope_krow
  h16_t* rope_v;
  const float* rope_cos;       // [maxT][D/2] fp32
  const float* rope_sin;
  long rope_krow, rope_kbatch;
  int rope_T, rope_pos0, rope_HD;
};

constexpr int BK = 64;
static int g_gemm_dbg = 0;   // tools only: ablation / A-B probe modes (g4r_gemm_debug_mode)

// Workgroup -> (output tile, K slice).  The dispatcher deals consecutive linear workgroup ids (x fastest, then y) round robin
// to the 8 XCDs; each XCD has its own L2.  The map hands every XCD one CONTIGUOUS run of the virtual ids
// [slice][tile] (cdna_hip_programming.md 5.5 T1, bijective for any count): without K slices that is a run of neighbouring
// tiles; with K slices an XCD works on as few slices as possible -- many tiles of ONE K window share the A rows and the W rows
// of that window in its L2.  (Round 2 mapped tiles only and gave every XCD all the slices of a few tiles: on the LLaMA
// down_proj, 64 tiles x 4 slices, every L2 then streamed the whole A matrix, 8 x 17 MB, and the K tile ran 1.7x the qkv's.)
__device__ __forceinline__ void g4r_workgroup_tile_slice(int nwg, bool tile_major, int& tile, int& slice) {
  if (tile_major) {                        // tools only (debug mode 11): the round-2 map, for A/B runs
    const int wg = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    slice = blockIdx.y;
    return;
  }
  const int total = nwg * (int)gridDim.y;
  const int lin = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
  const int q = total >> 3, r = total & 7, xcd = lin & 7, idx = lin >> 3;
  const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  slice = v / nwg;
  tile = v - slice * nwg;
}

// Tile id -> (row tile, column tile).  group_m == 0: the plain orders (M fastest: consecutive workgroups share a W panel; N
// fastest: they share an A tile).  group_m > 0 (round 4, dense GEMMs with many row tiles): GROUPED order -- ids sweep `group_m`
// row tiles x all column tiles, M fastest inside the group -- so that the 32 workgroups an XCD runs at a time form a compact
// group_m x (32 / group_m) block of tiles instead of one column of 32 row tiles: its L2 then streams group_m A tiles + 32 /
// group_m W panels per wave instead of ALL of A (8 merged requests, 6136 x 4096 x 4096: every XCD re-read the whole 50 MB
// activation matrix for each of its waves, 1.65 GB of HBM-side reads per launch against 134 MB algorithmic,
// profiles/r04_pmc_report.txt).
__device__ __forceinline__ void g4r_tile_coords(int wg, int tiles_m, int tiles_n, int n_fastest, int group_m, int& tile_m, int& tile_n) {
  if (group_m < 0) {
    // groups of -group_m COLUMN tiles x all row tiles, N fastest inside the group (conv: few column tiles, a big W panel each)
    const int gn = -group_m;
    const int per_group = gn * tiles_m;
    const int gid = wg / per_group;
    const int first_n = gid * gn;
    const int gsz = (tiles_n - first_n) < gn ? (tiles_n - first_n) : gn;
    const int in_group = wg - gid * per_group;
    tile_n = first_n + in_group % gsz;
    tile_m = in_group / gsz;
    return;
  }
  if (group_m > 0) {
    const int per_group = group_m * tiles_n;
    const int gid = wg / per_group;
    const int first_m = gid * group_m;
    const int gsz = (tiles_m - first_m) < group_m ? (tiles_m - first_m) : group_m;
    const int in_group = wg - gid * per_group;
    tile_m = first_m + in_group % gsz;
    tile_n = in_group / gsz;
    return;
  }
  tile_m = n_fastest ? wg / tiles_n : wg % tiles_m;
  tile_n = n_fastest ? wg % tiles_n : wg / tiles_m;
}

// waves per SIMD the kernel is allowed to assume = workgroups that fit the 160 KB LDS (<= 3)
constexpr int gemm_waves_per_eu(int bm, int bn, int nw, int stages, int bk) {
  int blocks = (160 * 1024) / (stages * (bm + bn) * bk * 2);
  if (blocks > 3) blocks = 3;
  if (blocks < 1) blocks = 1;
  int w = blocks * nw / 4;
  return w > 4 ? 4 : (w < 1 ? 1 : w);
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return v > 0.f ? v : 0.f;
  if (act == 2) return v / (1.f + __expf(-1.702f * v));
  if (act == 3) return v / (1.f + __expf(-v));
  return v;
}

// Shared epilogue.  acc[i][j] is the 32x32 accumulator of rows mw + 32*i.. and columns nw + 32*j..,
// in the swapped-operand layout D[n][m]: this lane holds m = mw + 32*i and, per register quad q,
// the 4 consecutive columns nw + 32*j + 8*q .. +3 (mw / nw already include the lane offsets).
// J0 / JN: the accumulator columns this call handles (a 4 x 4 wave tile is emitted as two calls of 4 x 2: the fully
// unrolled 16-accumulator nest exceeds the compiler's unroll budget, and a rolled nest would index `acc` dynamically,
// i.e. put the accumulators in scratch).
template <int TM, int TN, int J0 = 0, int JN = TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, float16v (&acc)[TM][TN], int mw, int nw,
                                              int split) {
  const bool vec_ok = (p.N & 3) == 0;  // then every in-range quad is a full, aligned quad
  if (p.splits > 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = J0; j < J0 + JN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = mw + i * 32, n = nw + j * 32 + q * 8;
          if (m < p.M && n < p.N) {
            float* dst = p.ws + ((size_t)split * p.M + m) * p.N + n;
            if (vec_ok) {
              *reinterpret_cast<float4v*>(dst) = float4v{acc[i][j][q * 4], acc[i][j][q * 4 + 1],
                                                         acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (n + r < p.N) dst[r] = acc[i][j][q * 4 + r];
            }
          }
        }
    return;
  }
  // One 32x32 accumulator at a time (keeps the live range of epilogue temporaries short):
  // bias (16-B loads) -> activation (wave-uniform branch) -> residual add -> packed store.
  const bool res_vec = vec_ok && (p.ldr & 3) == 0;
  const bool out_vec = vec_ok && (p.ldc & 3) == 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = J0; j < J0 + JN; ++j) {
      float16v c = acc[i][j];
      const int m = mw + i * 32;
      if (p.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nw + j * 32 + q * 8;
          float4v b = {0.f, 0.f, 0.f, 0.f};
          if (n < p.N) {
            if (vec_ok) {
              b = *reinterpret_cast<const float4v*>(p.bias + n);
            } else {
              b.x = p.bias[n];
              if (n + 1 < p.N) b.y = p.bias[n + 1];
              if (n + 2 < p.N) b.z = p.bias[n + 2];
              if (n + 3 < p.N) b.w = p.bias[n + 3];
            }
          }
          c[q * 4 + 0] += b.x;
          c[q * 4 + 1] += b.y;
          c[q * 4 + 2] += b.z;
          c[q * 4 + 3] += b.w;
        }
      }
      if (p.act == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = fmaxf(c[r], 0.f);
      } else if (p.act == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = c[r] / (1.f + __expf(-1.702f * c[r]));
      } else if (p.act == 3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = c[r] / (1.f + __expf(-c[r]));
      }
      if (m < p.M && p.act == 4) {
        // SwiGLU epilogue (LLaMA MLP, HF LlamaMLP: down(silu(gate(x)) * up(x))): the weight rows were
        // interleaved at prepare() so this lane's quad is (g0, u0, g1, u1); silu is rounded to bf16
        // before the product, as the un-fused reference does.
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nw + j * 32 + q * 8;
          if (n >= p.N) continue;
          const float g0 = c[q * 4], u0 = c[q * 4 + 1], g1 = c[q * 4 + 2], u1 = c[q * 4 + 3];
          const float s0 = h16lo(pack_h16x2(g0 / (1.f + __expf(-g0)), 0.f));
          const float s1 = h16lo(pack_h16x2(g1 / (1.f + __expf(-g1)), 0.f));
          h16_t* dst = reinterpret_cast<h16_t*>(p.C) + (size_t)m * p.ldc + (n >> 1);
          *reinterpret_cast<uint32_t*>(dst) = pack_h16x2(s0 * u0, s1 * u1);
        }
      } else if (m < p.M) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nw + j * 32 + q * 8;
          if (n >= p.N) continue;
          float v0 = c[q * 4], v1 = c[q * 4 + 1], v2 = c[q * 4 + 2], v3 = c[q * 4 + 3];
          if (p.residual) {
            const h16_t* rp = p.residual + (size_t)m * p.ldr + n;
            if (res_vec) {
              const uint2v rr = *reinterpret_cast<const uint2v*>(rp);
              v0 += h16lo(rr.x); v1 += h16hi(rr.x); v2 += h16lo(rr.y); v3 += h16hi(rr.y);
            } else {
              v0 += h16_to_f32(rp[0]);
              if (n + 1 < p.N) v1 += h16_to_f32(rp[1]);
              if (n + 2 < p.N) v2 += h16_to_f32(rp[2]);
              if (n + 3 < p.N) v3 += h16_to_f32(rp[3]);
            }
          }
          if (p.out_f32) {
            float* dst = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
            if (out_vec) {
              *reinterpret_cast<float4v*>(dst) = float4v{v0, v1, v2, v3};
            } else {
              dst[0] = v0;
              if (n + 1 < p.N) dst[1] = v1;
              if (n + 2 < p.N) dst[2] = v2;
              if (n + 3 < p.N) dst[3] = v3;
            }
          } else {
            h16_t* dst = reinterpret_cast<h16_t*>(p.C) + (size_t)m * p.ldc + n;
            if (out_vec) {
              *reinterpret_cast<uint2v*>(dst) = uint2v{pack_h16x2(v0, v1), pack_h16x2(v2, v3)};
            } else {
              dst[0] = f32_to_h16(v0);
              if (n + 1 < p.N) dst[1] = f32_to_h16(v1);
              if (n + 2 < p.N) dst[2] = f32_to_h16(v2);
              if (n + 3 < p.N) dst[3] = f32_to_h16(v3);
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Epilogue of the 256 x 256 kernels, staged through LDS (round 2).  In the swapped-operand accumulator layout a lane
// owns ONE output row m and scattered column quads, so the direct epilogue above issues stores of 64 x 8 bytes on 32
// different rows: every store instruction touches 32-64 cache lines with 8 bytes each.  Measured: ~20 us of a ~105 us
// K = 4096 launch outside the K loop (profiles/r02_gemm_tiles.md).  Here each wave parks its accumulators (fp32, 64 rows at
// a time) in its own slice of the -- now idle -- operand ring and reads them back ROW-major: 8 consecutive columns per
// lane, WTN/8 lanes per row, so every store / bias / residual access is a whole 16-byte (bf16) or 32-byte (fp32) run of
// one row and a wave instruction covers complete 128-byte lines.  The arithmetic and its order are those of
// gemm_epilogue (bias -> activation -> residual -> one rounding), so results are bit-identical to the direct path.
// ---------------------------------------------------------------------------------------------
template <int TN, int PR = 64>
struct EpiLds {
  static constexpr int WTN = 32 * TN;            // wave-tile columns
  static constexpr int RS = WTN * 4 + 16;        // row stride in bytes (fp32 row + 16 B: conflict-free b128 writes)
  static constexpr int WAVE_BYTES = PR * RS;     // PR (32 or 64) rows per pass
};

template <int TM, int TN, int PR = 64>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmArgs& p, float16v (&acc)[TM][TN], char* wave_lds, int m_wave0,
                                                  int n_wave0, int lane, int split) {
  using E = EpiLds<TN, PR>;
  constexpr int LPR = E::WTN / 8;                // lanes per row (8 columns each)
  constexpr int RPI = 64 / LPR;                  // rows per wave instruction (WTN = 96: 5 rows, 4 lanes idle)
  constexpr int BPP = PR / 32;                   // 32-row accumulator blocks per pass
  constexpr int NIT = (PR + RPI - 1) / RPI;      // read-back iterations per pass
  static_assert((TM * 32) % PR == 0, "pass geometry");
  const int wr = lane & 31, wh = lane >> 5;
  const int rrow = lane / LPR, rcol = (lane % LPR) * 8;
  const bool vec_n = (p.N & 7) == 0;
  const bool swiglu = p.act == 4;
#pragma unroll
  for (int half = 0; half < TM / BPP; ++half) {
    // ---- park PR rows: lane writes its 4-column quads (16-byte LDS stores, rows = lanes) ----
#pragma unroll
    for (int i2 = 0; i2 < BPP; ++i2)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float16v& c = acc[half * BPP + i2][j];
          *reinterpret_cast<float4v*>(wave_lds + (i2 * 32 + wr) * E::RS + (j * 32 + q * 8 + wh * 4) * 4) =
              float4v{c[q * 4], c[q * 4 + 1], c[q * 4 + 2], c[q * 4 + 3]};
        }
    if (p.act == 5) {
      // ---- fused RoPE + KV-cache append (head_dim 128: this wave's 64 columns are HALF a head, the other half sits in
      //      the LDS slice of the neighbouring wave, same rows, same lane -> column map) ----
      if constexpr (E::WTN == 64) {
        __syncthreads();                                     // every wave has parked this pass
        const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const char* mate_lds = wave_lds + ((wave_id & 1) ? -E::WAVE_BYTES : E::WAVE_BYTES);
        const int part = n_wave0 / p.rope_HD;                // 0 q, 1 k, 2 v (a 256-column tile never straddles: HD % 256 == 0)
        const bool second = (wave_id & 1) != 0;              // this wave holds d in [64, 128) of its head
#pragma unroll 2
        for (int it = 0; it < NIT; ++it) {
          const int r = it * RPI + rrow;
          if (rrow >= RPI || r >= PR) continue;
          const int m = m_wave0 + half * PR + r;
          if (m >= p.M) continue;
          const float4v a0 = *reinterpret_cast<const float4v*>(wave_lds + r * E::RS + rcol * 4);
          const float4v a1 = *reinterpret_cast<const float4v*>(wave_lds + r * E::RS + rcol * 4 + 16);
          float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          const int b = m / p.rope_T, t = m - b * p.rope_T;
          const int pos = p.rope_pos0 + t;
          const int col = n_wave0 + rcol - part * p.rope_HD;   // column inside q / k / v
          // the unfused path stored the projection as bf16 before rotating it: keep that rounding point
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = h16_to_f32(f32_to_h16(v[k]));
          if (part < 2) {
            const float4v m0 = *reinterpret_cast<const float4v*>(mate_lds + r * E::RS + rcol * 4);
            const float4v m1 = *reinterpret_cast<const float4v*>(mate_lds + r * E::RS + rcol * 4 + 16);
            float u[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
            const float* cp = p.rope_cos + (size_t)pos * 64 + rcol;
            const float* sp = p.rope_sin + (size_t)pos * 64 + rcol;
            const float4v c0 = *reinterpret_cast<const float4v*>(cp), c1 = *reinterpret_cast<const float4v*>(cp + 4);
            const float4v s0 = *reinterpret_cast<const float4v*>(sp), s1 = *reinterpret_cast<const float4v*>(sp + 4);
            const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float mate = h16_to_f32(f32_to_h16(u[k]));
              // rotate_half: first half  a' = a cos - b sin ; second half  b' = b cos + a sin   (a = x[d], b = x[d + 64])
              v[k] = second ? __builtin_fmaf(v[k], cs[k], mate * sn[k]) : __builtin_fmaf(v[k], cs[k], -(mate * sn[k]));
            }
          }
          h16_t* dst = part == 0 ? p.rope_q + (size_t)m * p.rope_HD + col
                                  : (part == 1 ? p.rope_k : p.rope_v) + (size_t)b * p.rope_kbatch + (size_t)pos * p.rope_krow + col;
          *reinterpret_cast<uint4v*>(dst) = uint4v{pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3]), pack_h16x2(v[4], v[5]),
                                                   pack_h16x2(v[6], v[7])};
        }
        __syncthreads();                                     // the neighbour is done with this wave's slice
      }
      continue;
    }
    // ---- read back row-major and finish ----
#pragma unroll 2
    for (int it = 0; it < NIT; ++it) {
      const int r = it * RPI + rrow;
      if (rrow >= RPI || r >= PR) continue;      // idle lanes / the last, partial iteration (no barrier inside the loop)
      const int m = m_wave0 + half * PR + r, n = n_wave0 + rcol;
      const float4v a0 = *reinterpret_cast<const float4v*>(wave_lds + r * E::RS + rcol * 4);
      const float4v a1 = *reinterpret_cast<const float4v*>(wave_lds + r * E::RS + rcol * 4 + 16);
      if (m >= p.M || n >= p.N) continue;
      float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const int nv = p.N - n < 8 ? p.N - n : 8;           // valid columns of this run
      if (p.splits > 1) {
        float* dst = p.ws + ((size_t)split * p.M + m) * p.N + n;
        if (nv == 8 && vec_n) {
          *reinterpret_cast<float4v*>(dst) = a0;
          *reinterpret_cast<float4v*>(dst + 4) = a1;
        } else {
          for (int k = 0; k < nv; ++k) dst[k] = v[k];
        }
        continue;
      }
      if (p.bias) {
        if (nv == 8 && vec_n) {
          const float4v b0 = *reinterpret_cast<const float4v*>(p.bias + n), b1 = *reinterpret_cast<const float4v*>(p.bias + n + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        } else {
          for (int k = 0; k < nv; ++k) v[k] += p.bias[n + k];
        }
      }
      if (p.act == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
      } else if (p.act == 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = v[k] / (1.f + __expf(-1.702f * v[k]));
      } else if (p.act == 3) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = v[k] / (1.f + __expf(-v[k]));
      }
      if (swiglu) {
        // (gate, up) column pairs -> N/2 output columns (the launcher guarantees N % 4 == 0, bf16 out, no bias/residual)
        h16_t* dst = reinterpret_cast<h16_t*>(p.C) + (size_t)m * p.ldc + (n >> 1);
        uint32_t w2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float g0 = v[4 * k], u0 = v[4 * k + 1], g1 = v[4 * k + 2], u1 = v[4 * k + 3];
          const float s0 = h16lo(pack_h16x2(g0 / (1.f + __expf(-g0)), 0.f));
          const float s1 = h16lo(pack_h16x2(g1 / (1.f + __expf(-g1)), 0.f));
          w2[k] = pack_h16x2(s0 * u0, s1 * u1);
        }
        if (nv == 8 && (p.ldc & 3) == 0) {
          *reinterpret_cast<uint2v*>(dst) = uint2v{w2[0], w2[1]};
        } else {
          *reinterpret_cast<uint32_t*>(dst) = w2[0];
          if (nv > 4) *reinterpret_cast<uint32_t*>(dst + 2) = w2[1];
        }
        continue;
      }
      if (p.residual) {
        const h16_t* rp = p.residual + (size_t)m * p.ldr + n;
        if (nv == 8 && vec_n && (p.ldr & 7) == 0) {
          const uint4v rr = *reinterpret_cast<const uint4v*>(rp);
          v[0] += h16lo(rr.x); v[1] += h16hi(rr.x); v[2] += h16lo(rr.y); v[3] += h16hi(rr.y);
          v[4] += h16lo(rr.z); v[5] += h16hi(rr.z); v[6] += h16lo(rr.w); v[7] += h16hi(rr.w);
        } else {
          for (int k = 0; k < nv; ++k) v[k] += h16_to_f32(rp[k]);
        }
      }
      if (p.out_f32) {
        float* dst = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
        if (nv == 8 && vec_n && (p.ldc & 3) == 0) {
          *reinterpret_cast<float4v*>(dst) = float4v{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<float4v*>(dst + 4) = float4v{v[4], v[5], v[6], v[7]};
        } else {
          for (int k = 0; k < nv; ++k) dst[k] = v[k];
        }
      } else {
        h16_t* dst = reinterpret_cast<h16_t*>(p.C) + (size_t)m * p.ldc + n;
        if (nv == 8 && vec_n && (p.ldc & 7) == 0) {
          *reinterpret_cast<uint4v*>(dst) = uint4v{pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3]), pack_h16x2(v[4], v[5]),
                                                   pack_h16x2(v[6], v[7])};
        } else {
          for (int k = 0; k < nv; ++k) dst[k] = f32_to_h16(v[k]);
        }
      }
    }
  }
}

// STAGES == 2: double buffer, one __syncthreads per K tile, latency hidden by 2-3 co-resident
//              workgroups per CU.
// STAGES >= 3: ring of LDS buffers, STAGES-1 tiles of LDS-DMA in flight, counted s_waitcnt vmcnt
//              (never 0 in steady state) + raw s_barrier, fragments double-buffered in registers;
//              one workgroup per CU owns most of the 160 KB LDS.
// BKT = K depth of one staged tile (64, or 32: half the LDS per stage -> more co-resident workgroups
// or a deeper ring at the same footprint).
template <int BM, int BN, int WM, int WN, int AMODE, bool GLDS, int STAGES, int BKT, int STYLE>
__global__ __launch_bounds__(WM* WN * 64, gemm_waves_per_eu(BM, BN, WM * WN, STAGES, BKT))
void gemm_bf16_nt_kernel(GemmArgs p) {
  constexpr int NW = WM * WN;
  constexpr int NT = NW * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;  // 32x32 accumulators per wave
  constexpr int ROWB = BKT * 2;          // bytes per tile row
  constexpr int SPR = BKT / 8;           // 16-B slots per tile row
  constexpr int KK = BKT / 16;           // MFMA k-steps per tile
  constexpr int NA = BM * SPR / NT, NB = BN * SPR / NT;  // 16-B slots staged per thread
  constexpr int A_BYTES = BM * BKT * 2, B_BYTES = BN * BKT * 2;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static_assert(BM * SPR % NT == 0 && BN * SPR % NT == 0, "tile/threads mismatch");
  // bank-conflict swizzle of the 16-B slot index within a row (conflict-free ds_read_b128 lane groups)
  auto swz = [](int row) { return BKT == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  static_assert(STAGES == 2 || GLDS, "the deep pipeline needs LDS-DMA staging");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // STAGES * STAGE_BYTES

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- XCD-aware bijective remap of the 1-D grid (cdna_hip_programming.md 5.5 T1) ----
  const int nwg = p.tiles_m * p.tiles_n;
  int wg, split;
  g4r_workgroup_tile_slice(nwg, p.dbg == 11, wg, split);
  // Tile order = which operand the co-scheduled workgroups of one XCD share through its L2:
  //   M fastest: same W panel (weights are the big operand when M << N: LLaMA / ViT projections);
  //   N fastest: same A tile (activations are the big operand when M >> N: the implicit-GEMM convs,
  //              whose 9-tap gather was re-fetched once per column tile -- 2.5 GB/launch of fabric
  //              traffic at 192^2 against 170 MB algorithmic, profiles/r01_bench_rocprof.md).
  const int tile_m = p.n_fastest ? wg / p.tiles_n : wg % p.tiles_m;
  const int tile_n = p.n_fastest ? wg % p.tiles_n : wg / p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t_begin = split * p.tiles_per_split;
  int t_end = t_begin + p.tiles_per_split;
  const int nt_total = p.K / BKT;
  if (t_end > nt_total) t_end = nt_total;

  // ---- per-thread staging descriptors ----
  const h16_t* a_src[NA];
  int a_y[NA], a_x[NA];
  const h16_t* b_src[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int pslot = (j * NW + wave) * 64 + lane;
    const int row = pslot / SPR, ps = pslot % SPR;
    const int kslot = ps ^ swz(row);
    int gm = m0 + row;
    if (gm > p.M - 1) gm = p.M - 1;
    if (AMODE == 0) {
      a_src[j] = p.A + (size_t)gm * p.lda + kslot * 8;
      a_y[j] = a_x[j] = 0;
    } else {
      const int hw = p.H * p.Wd;
      const int b = gm / hw, rem = gm - b * hw;
      a_y[j] = rem / p.Wd;
      a_x[j] = rem - a_y[j] * p.Wd;
      a_src[j] = p.A + (size_t)gm * p.lda + kslot * 8;  // centre tap of this pixel
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int pslot = (j * NW + wave) * 64 + lane;
    const int row = pslot / SPR, ps = pslot % SPR;
    const int kslot = ps ^ swz(row);
    int gn = n0 + row;
    if (gn > p.N - 1) gn = p.N - 1;
    b_src[j] = p.W + (size_t)gn * p.ldw + kslot * 8;
  }

  auto stage = [&](int t, int buf) {
    char* sa = smem + buf * STAGE_BYTES;
    char* sb = sa + A_BYTES;
    int k0 = t * BKT;
    // A operand
    long a_off;
    int dy = 0, dx = 0;
    if (AMODE == 0) {
      a_off = k0;
    } else {
      // K tiles walk the taps FASTEST (channel slice outer): the nine shifted reads of one channel slice follow each
      // other within ~9 K tiles, so eight of them hit the L2 instead of HBM (the weights keep the [tap][Cin] layout)
      const int n_taps = 9 * p.groups;
      int ct = t / n_taps;
      int tap_lin = t - ct * n_taps;            // group * 9 + tap
      if (p.dbg == 7) {                         // A/B probe (tools only): taps outermost, the round-1 order
        const int per_tap = p.Cin / BKT;
        tap_lin = t / per_tap;
        ct = t - tap_lin * per_tap;
      }
      const int c0 = ct * BKT;
      const int grp = tap_lin / 9, tap = tap_lin - grp * 9;
      dy = tap / 3 - 1;
      dx = tap - (tap / 3) * 3 - 1;
      a_off = (long)grp * p.a_group_stride + ((long)dy * p.Wd + dx) * p.lda + c0;
      k0 = tap_lin * p.Cin + c0;
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      if (p.dbg == 3 && t != t_begin) break;  // ablation: A staged once, W keeps streaming
      const h16_t* src = a_src[j] + a_off;
      if (AMODE == 1) {
        const int yy = a_y[j] + dy, xx = a_x[j] + dx;
        if (yy < 0 || yy >= p.H || xx < 0 || xx >= p.Wd) src = p.zeros + (lane & 7) * 8;
      }
      char* dst_wave = sa + (j * NW + wave) * 1024;
      if (GLDS) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst_wave, 16, 0, 0);
      } else {
        *reinterpret_cast<uint4v*>(dst_wave + lane * 16) = *reinterpret_cast<const uint4v*>(src);
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (p.dbg == 5 && t != t_begin) break;  // ablation: W staged once, A keeps streaming
      const h16_t* src = b_src[j] + k0;
      char* dst_wave = sb + (j * NW + wave) * 1024;
      if (GLDS) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst_wave, 16, 0, 0);
      } else {
        *reinterpret_cast<uint4v*>(dst_wave + lane * 16) = *reinterpret_cast<const uint4v*>(src);
      }
    }
  };

  float16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets: row = base + (lane & 31), 16-B slot = (kk*2 + (lane>>5)) ^ ((row>>1)&7)
  const int frow = lane & 31;
  const int fsw = swz(frow);
  const int fhi = lane >> 5;
  const int a_row_off = (wm * WTM + frow) * ROWB;
  const int b_row_off = (wn * WTN + frow) * ROWB;

  auto compute = [&](int buf) {
    const char* sa = smem + buf * STAGE_BYTES;
    const char* sb = sa + A_BYTES;
    // STYLE 0: by LDS footprint (interleaved when >= 2 workgroups fit a CU), 1: interleaved, 2: burst
    if constexpr (STYLE == 1 || (STYLE == 0 && STAGES * (BM + BN) * BKT * 2 <= 80 * 1024)) {
      // 2-3 co-resident workgroups per CU: let the compiler interleave the 4 reads and 4 MFMAs of
      // each k-step (few VGPRs -> more waves).  Measured on MI355X: faster end to end than
      // issuing all 16 reads first (tools/gemm_ablate.py; DESIGN.md "GEMM experiments").
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int slot = ((kk * 2 + fhi) ^ fsw) << 4;
        h16x8 af[TM], wf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = *reinterpret_cast<const h16x8*>(sa + a_row_off + i * 32 * ROWB + slot);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          wf[j] = *reinterpret_cast<const h16x8*>(sb + b_row_off + j * 32 * ROWB + slot);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = G4R_MFMA_32X32X16(wf[j], af[i], acc[i][j], 0, 0, 0);
      }
    } else if constexpr (STYLE == 3) {
      // every fragment read of the K tile goes out back to back and each k-step's MFMAs wait for THEIR reads only (counted
      // lgkmcnt).  The reads are inline asm: given plain loads the compiler orders them read, read, s_waitcnt lgkmcnt(0),
      // MFMA -- an LDS round trip in front of every k-step (the interleaved style above: ~150 cycles x KK per K tile against
      // 32 x KK of MFMA on a 64 x 64 tile) -- and the burst style below makes the first MFMA wait for ALL the reads.  Each
      // wait is tied ("+v") to the fragments it releases so that their MFMAs cannot move above it.
      h16x8 af[KK][TM], wf[KK][TN];
      const unsigned la = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(sa + a_row_off);
      const unsigned lb = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(sb + b_row_off);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const unsigned slot = (unsigned)(((kk * 2 + fhi) ^ fsw) << 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[kk][i]) : "v"(la + slot), "n"(i * 32 * ROWB) : "memory");
#pragma unroll
        for (int j = 0; j < TN; ++j)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[kk][j]) : "v"(lb + slot), "n"(j * 32 * ROWB) : "memory");
      }
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(af[kk][0]) : "n"((TM + TN) * (KK - 1 - kk)));
#pragma unroll
        for (int i = 1; i < TM; ++i) asm volatile("" : "+v"(af[kk][i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(wf[kk][j]));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = G4R_MFMA_32X32X16(wf[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // one workgroup per CU: request every fragment of the K tile up front, then one MFMA burst
      h16x8 af[KK][TM], wf[KK][TN];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int slot = ((kk * 2 + fhi) ^ fsw) << 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[kk][i] = *reinterpret_cast<const h16x8*>(sa + a_row_off + i * 32 * ROWB + slot);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          wf[kk][j] = *reinterpret_cast<const h16x8*>(sb + b_row_off + j * 32 * ROWB + slot);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the burst
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = G4R_MFMA_32X32X16(wf[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
    }
  };

  if constexpr (STAGES == 2) {
    if (t_begin < t_end) {
      stage(t_begin, 0);
      __syncthreads();  // drains the LDS-DMA queue (vmcnt(0)) and publishes the tile
      int cur = 0;
      for (int t = t_begin; t < t_end - 1; ++t) {
        if (p.dbg != 1) stage(t + 1, cur ^ 1);
        if (p.dbg != 2) compute(cur);
        __syncthreads();
        cur ^= 1;
      }
      compute(cur);
    }
  } else {
    constexpr int NLD = NA + NB;  // LDS-DMA instructions per thread per tile
    const int nt = t_end - t_begin;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < nt) stage(t_begin + s, s);
    int rd = 0, wr = STAGES - 1;
    for (int i = 0; i < nt; ++i) {
      // tile i must have landed; up to STAGES-2 younger tiles stay in flight across the barrier
      const int ahead = nt - 1 - i;
      if (STAGES >= 4 && ahead >= 2)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
      else if (ahead >= 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // everyone's tile i is in LDS; everyone is done reading tile i-1
      asm volatile("" ::: "memory");
      if (i + STAGES - 1 < nt && p.dbg != 1) stage(t_begin + i + STAGES - 1, wr);
      if (p.dbg != 2 || i + 1 == nt) compute(rd);
      rd = rd + 1 == STAGES ? 0 : rd + 1;
      wr = wr + 1 == STAGES ? 0 : wr + 1;
    }
  }

  // ---- epilogue through LDS (row-major, whole-line stores: see gemm_epilogue_lds), 32 rows per pass ----
  __syncthreads();   // every wave has read its last fragments: the staging buffers are free
  gemm_epilogue_lds<TM, TN, 32>(p, acc, smem + wave * EpiLds<TN, 32>::WAVE_BYTES, m0 + wm * WTM, n0 + wn * WTN, lane, split);
}

// split-K combine + epilogue: C = act(sum_s ws[s] + bias) + residual
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs p) {
  const long total = (long)p.M * p.N;
  // vector path: 4 consecutive columns per thread, 16-byte loads of every partial (the scalar loops below moved 4 bytes per
  // lane per load: 10 us average over the 89 reduce launches of a round-2 step).  Same summation order (s = 0, 1, ...) and
  // the same bias -> activation -> residual -> one rounding sequence: bit-identical to the scalar path.
  // (base pointers too: gemm() passes column-sliced views of C / residual / bias -- 16-byte loads of the bias and 8/16-byte
  // accesses of C and the residual need the slice itself aligned, not only its leading dimension)
  const bool ptrs_aligned = ((uintptr_t)p.C & 15) == 0 && ((uintptr_t)p.residual & 7) == 0 && ((uintptr_t)p.bias & 15) == 0 &&
                            ((uintptr_t)p.ws & 15) == 0;
  if (ptrs_aligned && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 && (p.act != 4 || (p.ldc & 1) == 0)) {
    const int n4 = p.N >> 2;
    const long quads = (long)p.M * n4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < quads; i += (long)gridDim.x * 256) {
      const int c4 = (int)(i % n4) * 4;
      const long m = i / n4;
      float4v x = {0.f, 0.f, 0.f, 0.f};
      const float* src = p.ws + m * p.N + c4;
      for (int s0 = 0; s0 < p.splits; s0 += 4) {      // up to 4 partial loads in flight
        float4v v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (s0 + u < p.splits) v[u] = *reinterpret_cast<const float4v*>(src + (size_t)(s0 + u) * total);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (s0 + u < p.splits) x += v[u];
      }
      if (p.act == 4) {
        const float s0v = h16lo(pack_h16x2(x[0] / (1.f + __expf(-x[0])), 0.f));
        const float s1v = h16lo(pack_h16x2(x[2] / (1.f + __expf(-x[2])), 0.f));
        *reinterpret_cast<uint32_t*>(reinterpret_cast<h16_t*>(p.C) + (size_t)m * p.ldc + (c4 >> 1)) =
            pack_h16x2(s0v * x[1], s1v * x[3]);
        continue;
      }
      if (p.bias) {
        const float4v bv = *reinterpret_cast<const float4v*>(p.bias + c4);
        x += bv;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = apply_act(x[e], p.act);
      if (p.residual) {
        const uint2v r = *reinterpret_cast<const uint2v*>(p.residual + (size_t)m * p.ldr + c4);
        x[0] += h16lo(r.x); x[1] += h16hi(r.x); x[2] += h16lo(r.y); x[3] += h16hi(r.y);
      }
      if (p.out_f32) {
        *reinterpret_cast<float4v*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + c4) = x;
      } else {
        const uint2v w = {pack_h16x2(x[0], x[1]), pack_h16x2(x[2], x[3])};
        *reinterpret_cast<uint2v*>(reinterpret_cast<h16_t*>(p.C) + (size_t)m * p.ldc + c4) = w;
      }
    }
    return;
  }
  if (p.act == 4) {  // SwiGLU over interleaved (gate, up) column pairs: C has N/2 columns (bf16)
    const int half = p.N >> 1;
    const long pairs = (long)p.M * half;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (long)gridDim.x * 256) {
      const int c = (int)(i % half);
      const long m = i / half;
      float g = 0.f, u = 0.f;
      for (int s = 0; s < p.splits; ++s) {
        const float2 v = *reinterpret_cast<const float2*>(p.ws + (size_t)s * total + m * p.N + 2 * c);
        g += v.x;
        u += v.y;
      }
      const float sg = h16lo(pack_h16x2(g / (1.f + __expf(-g)), 0.f));
      reinterpret_cast<h16_t*>(p.C)[(size_t)m * p.ldc + c] = f32_to_h16(sg * u);
    }
    return;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i % p.N);
    const long m = i / p.N;
    float x = 0.f;
    for (int s = 0; s < p.splits; ++s) x += p.ws[(size_t)s * total + i];
    if (p.bias) x += p.bias[n];
    x = apply_act(x, p.act);
    if (p.residual) x += h16_to_f32(p.residual[(size_t)m * p.ldr + n]);
    if (p.out_f32)
      reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + n] = x;
    else
      reinterpret_cast<h16_t*>(p.C)[(size_t)m * p.ldc + n] = f32_to_h16(x);
  }
}

// Tiny/irregular contractions (K not a multiple of 64, e.g. the Linear(4,256) of pos_embedd,
// gpt4roi/models/layers.py:260-267): one thread per output, fp32 accumulate.  Not a hot path.
__global__ __launch_bounds__(256) void small_linear_kernel(const h16_t* __restrict__ A,
                                                           const h16_t* __restrict__ W,
                                                           const float* __restrict__ bias,
                                                           void* __restrict__ C, int M, int N, int K,
                                                           int lda, int ldw, int ldc, int act,
                                                           int out_f32) {
  const long total = (long)M * N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i % N);
    const long m = i / N;
    float x = 0.f;
    for (int k = 0; k < K; ++k) x += h16_to_f32(A[m * lda + k]) * h16_to_f32(W[(size_t)n * ldw + k]);
    if (bias) x += bias[n];
    x = apply_act(x, act);
    if (out_f32)
      reinterpret_cast<float*>(C)[m * ldc + n] = x;
    else
      reinterpret_cast<h16_t*>(C)[m * ldc + n] = f32_to_h16(x);
  }
}

// ---------------------------------------------------------------------------------------------
// GEMV for the single-token decode step (M = 1): pure weight streaming, HBM-bound (13.5 GB of bf16
// LLaMA-7B weights per token).  One wave owns 4 consecutive W rows; lanes split K in 16-byte
// pieces (one coalesced 1 KiB load per row per step, 4 rows in flight), fp32 accumulate, shuffle
// reduction.  Same epilogues as the GEMM (bias, residual, SwiGLU over interleaved row pairs, fp32 out).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void unpack8_h16(const uint4v& r, float* f) {
  f[0] = h16lo(r.x); f[1] = h16hi(r.x); f[2] = h16lo(r.y); f[3] = h16hi(r.y);
  f[4] = h16lo(r.z); f[5] = h16hi(r.z); f[6] = h16lo(r.w); f[7] = h16hi(r.w);
}

__device__ __forceinline__ float dot8_bf16(const uint4v& a, const uint4v& b, float acc) {
  acc += h16lo(a.x) * h16lo(b.x); acc += h16hi(a.x) * h16hi(b.x);
  acc += h16lo(a.y) * h16lo(b.y); acc += h16hi(a.y) * h16hi(b.y);
  acc += h16lo(a.z) * h16lo(b.z); acc += h16hi(a.z) * h16hi(b.z);
  acc += h16lo(a.w) * h16lo(b.w); acc += h16hi(a.w) * h16hi(b.w);
  return acc;
}

// x is staged ONCE per workgroup in LDS (the 4 waves of a workgroup would otherwise each re-read it from L2: K*2 B per
// wave against R*K*2 B of weights).  NORM fuses the RMSNorm that precedes the q|k|v, gate|up and lm_head projections
// (HF LlamaRMSNorm; same staging pattern, summation order and roundings as rmsnorm_bf16_kernel in norm.hip, so the
// normalised vector is bit-identical to the separate launch it replaces).  A wave owns R consecutive W rows and keeps
// R*U 16-byte loads in flight per lane (U K-steps of 1 KiB per row); shipped: R = 2, U = 8.
// XMODE 0: x as it is; 1: fused RMSNorm (gamma, eps); 2: x = the attention output assembled from the per-split partials
// the decode attention leaves behind (attention.hip, attn_decode_kernel with defer_merge): `gamma` then points at
// ws [H][S][D + 2] (un-normalised o, running max m, sum l per split), mS = S, mD = D.  Doing the merge in the consumer's
// staging loop replaces a cross-workgroup hand-off inside the attention launch (sc1 write-through, arrival counter, sc1
// read-back: ~6 us of serial memory round trips per layer) by a kernel boundary that is there anyway.
// x is staged in dynamic LDS (2 K bytes) next to a 16-byte static array: both must fit the default 64 KB limit
#define G4R_GEMV_MAX_K 32736
template <int R, int U, int XMODE, int NWV, int MAXV = 4>
__global__ __launch_bounds__(NWV * 64) void gemv_bf16_kernel(const h16_t* __restrict__ x, const float* __restrict__ gamma,
                                                        float eps, int mS, int mD, const h16_t* __restrict__ W,
                                                        void* __restrict__ C,
                                                        const float* __restrict__ bias,
                                                        const h16_t* __restrict__ residual, int N, int K, int ldw,
                                                        int act, int out_f32, int early_on) {
  extern __shared__ __attribute__((aligned(16))) char gemv_smem[];
  uint4v* xs = reinterpret_cast<uint4v*>(gemv_smem);
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nvec = K >> 3;
  const int n0 = (blockIdx.x * NWV + wave) * R;
  const h16_t* wrow[R];
#pragma unroll
  for (int r = 0; r < R; ++r) wrow[r] = W + (size_t)(n0 + r < N ? n0 + r : N - 1) * ldw;
  uint4v wv[U][R];
  auto load_w = [&](int vb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = vb + 64 * u < nvec ? vb + 64 * u : nvec - 1;
#pragma unroll
      for (int r = 0; r < R; ++r)
        wv[u][r] = __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(wrow[r] + (size_t)idx * 8));
    }
  };
  // Round 6: every wave issues its first block of weight loads right AFTER the loads of its share of the staging and BEFORE it
  // waits for them: vector memory returns in order, so the x loads (older) are not delayed and the weight stream runs from the
  // first cycles of the workgroup instead of after the 2-3 memory round trips + 2 barriers of the staging.  Same loads, same
  // arithmetic, same bits.  (The other order -- weights first, then x -- queues x behind R*U HBM loads: 5-8 % slower, round 2.)
  const bool early = early_on && n0 < N;   // (early_on = 0: debug mode 62, the A/B of tools/decode_bench.py)
  bool pre = false;                        // the first block of W is already in flight when the main loop starts
  if (XMODE == 1) {
    // the first 4 waves stage and normalise x exactly as rmsnorm_bf16_kernel's 256 threads do (same element -> thread
    // map and summation order: bit-identical rstd); further waves of a wide workgroup only wait at the barriers
    // MAXV vectors of 8 per staging thread: K <= 2048 * MAXV (the launcher picks 2 for K <= 4096, else 4; K <= 8192 checked there).
    // x stays PACKED in registers and is unpacked twice: with the first weight block (R*U*4 registers) in flight next to x and
    // gamma the kernel must stay under 80 registers, or a CU no longer holds three workgroups (the wide projections launch 3-8 per CU)
    uint4v xr[MAXV];
    // gamma: the waves that do not stage x (4..NWV-1) fetch it next to the x loads of the stagers and park it in LDS behind x, so
    // its round trip no longer follows the two barriers of the reduction and costs the stagers no registers
    constexpr bool GLDS = NWV > 4;
    constexpr int GW = GLDS ? (NWV - 4) * 64 : 1;
    float* gs = reinterpret_cast<float*>(gemv_smem + (size_t)K * 2);
    float4v gq[GLDS ? 2 * MAXV : 1];
    float s2 = 0.f;
    const bool stager = tid < 256;
    if (stager) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int v = tid + i * 256;
        if (v < nvec) xr[i] = *reinterpret_cast<const uint4v*>(x + v * 8);
      }
    } else if (GLDS) {
#pragma unroll
      for (int i = 0; i < 2 * MAXV; ++i) {
        const int v4 = tid - 256 + i * GW;
        if (v4 < (K >> 2)) gq[i] = *reinterpret_cast<const float4v*>(gamma + v4 * 4);
      }
    }
    if (early) { load_w(lane); pre = true; }
    if (stager) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int v = tid + i * 256;
        if (v < nvec) {
          float fx[8];
          unpack8_h16(xr[i], fx);
#pragma unroll
          for (int k = 0; k < 8; ++k) s2 += fx[k] * fx[k];
        }
      }
    } else if (GLDS) {
#pragma unroll
      for (int i = 0; i < 2 * MAXV; ++i) {
        const int v4 = tid - 256 + i * GW;
        if (v4 < (K >> 2)) *reinterpret_cast<float4v*>(gs + v4 * 4) = gq[i];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
    __syncthreads();
    if (lane == 0 && stager) red[wave] = s2;
    __syncthreads();
    const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = tid + i * 256;
      if (stager && v < nvec) {
        const float* gp = GLDS ? gs + v * 8 : gamma + v * 8;
        const float4v g0 = *reinterpret_cast<const float4v*>(gp);
        const float4v g1 = *reinterpret_cast<const float4v*>(gp + 4);
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        float o[8], fx[8];
        unpack8_h16(xr[i], fx);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = h16_to_f32(f32_to_h16(fx[k] * rstd)) * g[k];
        uint4v w;
        w.x = pack_h16x2(o[0], o[1]); w.y = pack_h16x2(o[2], o[3]);
        w.z = pack_h16x2(o[4], o[5]); w.w = pack_h16x2(o[6], o[7]);
        xs[v] = w;
      }
    }
  } else if (XMODE == 2) {
    const int SLD = mD + 2;
    constexpr int MS = 8;            // splits whose partials one thread holds in registers (the decode step uses 8)
    float pm[MS], pl[MS];
    float4v po0[MS], po1[MS];
    auto part_of = [&](int v, int& d0) {
      const int h = (v * 8) / mD;
      d0 = (v * 8) - h * mD;
      return gamma + (size_t)h * mS * SLD;
    };
    auto mload = [&](int v) {        // every load of the merge of one 8-element vector, issued together
      int d0;
      const float* part = part_of(v, d0);
#pragma unroll
      for (int i = 0; i < MS; ++i) {
        if (i < mS) {
          pm[i] = part[i * SLD + mD];
          pl[i] = part[i * SLD + mD + 1];
          po0[i] = *reinterpret_cast<const float4v*>(part + i * SLD + d0);
          po1[i] = *reinterpret_cast<const float4v*>(part + i * SLD + d0 + 4);
        }
      }
    };
    auto mfinish = [&](int v) {      // the arithmetic of the loop form below, term for term
      float mm = -INFINITY;
#pragma unroll
      for (int i = 0; i < MS; ++i)
        if (i < mS) mm = fmaxf(mm, pm[i]);
      float Lt = 0.f, a2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MS; ++i) {
        if (i < mS) {
          const float w = pm[i] == -INFINITY ? 0.f : exp2f(pm[i] - mm);
          Lt += w * pl[i];
          a2[0] += w * po0[i].x; a2[1] += w * po0[i].y; a2[2] += w * po0[i].z; a2[3] += w * po0[i].w;
          a2[4] += w * po1[i].x; a2[5] += w * po1[i].y; a2[6] += w * po1[i].z; a2[7] += w * po1[i].w;
        }
      }
      uint4v w8;
      w8.x = pack_h16x2(a2[0] / Lt, a2[1] / Lt); w8.y = pack_h16x2(a2[2] / Lt, a2[3] / Lt);
      w8.z = pack_h16x2(a2[4] / Lt, a2[5] / Lt); w8.w = pack_h16x2(a2[6] / Lt, a2[7] / Lt);
      xs[v] = w8;
    };
    if (mS <= MS) {
      int v = tid;
      if (v < nvec) mload(v);
      if (early) { load_w(lane); pre = true; }
      if (v < nvec) mfinish(v);
      for (v += NWV * 64; v < nvec; v += NWV * 64) {
        mload(v);
        mfinish(v);
      }
    } else {
      for (int v = tid; v < nvec; v += NWV * 64) {
        int d0;
        const float* part = part_of(v, d0);
        float mm = -INFINITY;
        for (int i = 0; i < mS; ++i) mm = fmaxf(mm, part[i * SLD + mD]);
        float Lt = 0.f, a2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < mS; ++i) {
          const float mi = part[i * SLD + mD];
          const float w = mi == -INFINITY ? 0.f : exp2f(mi - mm);
          Lt += w * part[i * SLD + mD + 1];
          const float4v o0 = *reinterpret_cast<const float4v*>(part + i * SLD + d0);
          const float4v o1 = *reinterpret_cast<const float4v*>(part + i * SLD + d0 + 4);
          a2[0] += w * o0.x; a2[1] += w * o0.y; a2[2] += w * o0.z; a2[3] += w * o0.w;
          a2[4] += w * o1.x; a2[5] += w * o1.y; a2[6] += w * o1.z; a2[7] += w * o1.w;
        }
        uint4v w8;
        w8.x = pack_h16x2(a2[0] / Lt, a2[1] / Lt); w8.y = pack_h16x2(a2[2] / Lt, a2[3] / Lt);
        w8.z = pack_h16x2(a2[4] / Lt, a2[5] / Lt); w8.w = pack_h16x2(a2[6] / Lt, a2[7] / Lt);
        xs[v] = w8;
      }
    }
  } else {
    constexpr int XPRE = 3;          // x vectors per thread held in registers across the early weight loads (K <= 12288 at 8 waves)
    uint4v xr[XPRE];
#pragma unroll
    for (int i = 0; i < XPRE; ++i) {
      const int v = tid + i * NWV * 64;
      if (v < nvec) xr[i] = *reinterpret_cast<const uint4v*>(x + v * 8);
    }
    if (early) { load_w(lane); pre = true; }
#pragma unroll
    for (int i = 0; i < XPRE; ++i) {
      const int v = tid + i * NWV * 64;
      if (v < nvec) xs[v] = xr[i];
    }
    for (int v = tid + XPRE * NWV * 64; v < nvec; v += NWV * 64) xs[v] = *reinterpret_cast<const uint4v*>(x + v * 8);
  }
  __syncthreads();
  if (n0 >= N) return;
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.f;
  for (int vb = lane; vb < nvec; vb += 64 * U) {
    if (!(pre && vb == lane)) load_w(vb);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (vb + 64 * u < nvec) {
        const uint4v xv = xs[vb + 64 * u];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = dot8_bf16(xv, wv[u][r], acc[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o);
  if (lane != 0) return;
  if (act == 4) {  // rows are (gate, up) pairs
#pragma unroll
    for (int r = 0; r < R; r += 2) {
      if (n0 + r + 1 < N) {
        const float g = acc[r], u = acc[r + 1];
        const float sg = h16lo(pack_h16x2(g / (1.f + __expf(-g)), 0.f));
        reinterpret_cast<h16_t*>(C)[(n0 + r) >> 1] = f32_to_h16(sg * u);
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int n = n0 + r;
    if (n >= N) break;
    float vv = acc[r];
    if (bias) vv += bias[n];
    vv = apply_act(vv, act);
    if (residual) vv += h16_to_f32(residual[n]);
    if (out_f32)
      reinterpret_cast<float*>(C)[n] = vv;
    else
      reinterpret_cast<h16_t*>(C)[n] = f32_to_h16(vv);
  }
}

template <int R, int U, int NWV>
static void launch_gemv(int xmode, const h16_t* x, const float* gamma, float eps, int mS, int mD, const h16_t* W, void* C,
                        const float* bias, const h16_t* residual, int N, int K, int ldw, int act, int out_f32,
                        hipStream_t stream) {
  const int waves = g4r_ceil_div(N, R);
  const dim3 grid(g4r_ceil_div(waves, NWV)), block(NWV * 64);
  const size_t lds = (size_t)K * 2 + (xmode == 1 && NWV > 4 ? (size_t)K * 4 : 0);   // x, and gamma behind it (fused norm, wide workgroups)
  if (xmode == 1 && K <= 4096)
    hipLaunchKernelGGL((gemv_bf16_kernel<R, U, 1, NWV, 2>), grid, block, lds, stream, x, gamma, eps, mS, mD, W, C, bias,
                       residual, N, K, ldw, act, out_f32, g_gemm_dbg == 62 ? 0 : 1);
  else if (xmode == 1)
    hipLaunchKernelGGL((gemv_bf16_kernel<R, U, 1, NWV>), grid, block, lds, stream, x, gamma, eps, mS, mD, W, C, bias,
                       residual, N, K, ldw, act, out_f32, g_gemm_dbg == 62 ? 0 : 1);
  else if (xmode == 2)
    hipLaunchKernelGGL((gemv_bf16_kernel<R, U, 2, NWV>), grid, block, lds, stream, x, gamma, eps, mS, mD, W, C, bias,
                       residual, N, K, ldw, act, out_f32, g_gemm_dbg == 62 ? 0 : 1);
  else
    hipLaunchKernelGGL((gemv_bf16_kernel<R, U, 0, NWV>), grid, block, lds, stream, x, gamma, eps, mS, mD, W, C, bias,
                       residual, N, K, ldw, act, out_f32, g_gemm_dbg == 62 ? 0 : 1);
}

// (R rows per wave, U K-steps in flight, waves per workgroup); variant >= 0: A/B probe (tools/gemm_bench.cpp v: cases)
static void gemv_dispatch(int variant, int xmode, const h16_t* x, const float* gamma, float eps, int mS, int mD,
                          const h16_t* W, void* C, const float* bias, const h16_t* residual, int N, int K, int ldw,
                          int act, int out_f32, hipStream_t st) {
  if (variant < 0) variant = 2;   // R = 2, U = 4, 8 waves: best or within noise of the best on all five LLaMA-7B projections (profiles/r02_gemv_variants.txt)
#define G4R_GEMV_CASE(v, R_, U_, NW_) \
  case v: launch_gemv<R_, U_, NW_>(xmode, x, gamma, eps, mS, mD, W, C, bias, residual, N, K, ldw, act, out_f32, st); break;
  switch (variant) {
    G4R_GEMV_CASE(0, 4, 2, 4)
    G4R_GEMV_CASE(1, 2, 4, 4)
    G4R_GEMV_CASE(2, 2, 4, 8)
    G4R_GEMV_CASE(3, 2, 4, 16)
    G4R_GEMV_CASE(4, 2, 8, 4)
    G4R_GEMV_CASE(5, 2, 8, 8)
    G4R_GEMV_CASE(6, 4, 2, 8)
    default: launch_gemv<4, 4, 8>(xmode, x, gamma, eps, mS, mD, W, C, bias, residual, N, K, ldw, act, out_f32, st); break;
  }
#undef G4R_GEMV_CASE
}

// ---------------------------------------------------------------------------------------------
// Ping-pong kernel: 256 x 256 x 64 tile, 8 waves (2 x 4), wave tile 128 x 64, 128 KB LDS (2 K tiles).
// The two waves that share a SIMD (wave w and w+4 = the two row halves) run HALF A K-TILE OUT OF
// PHASE, separated by 4 workgroup barriers per K tile: while one does the 16 MFMAs of a k-half the other
// reads its 12 fragments of the next k-half from LDS, so the matrix pipe of every SIMD always has a
// wave whose operands are already in registers (the two-barrier loop leaves it idle during every
// ds_read round trip: 44 % MFMA utilisation in the ablation).  LDS-DMA for K tile t+1 is issued at the
// first phase of tile t and waited for (vmcnt(0)) at its last phase, four phases later.
//        phase:      P0            P1            P2            P3
//   waves 0-3:   read h0(t)     MFMA h0(t)    read h1(t)    MFMA h1(t)
//   waves 4-7:   MFMA h1(t-1)   read h0(t)    MFMA h0(t)    read h1(t)
// ---------------------------------------------------------------------------------------------
#define G4R_PP_BARRIER()                                   \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

template <int AMODE, bool PROBE = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp_kernel(GemmArgs p) {
  constexpr int BM = 256, BN = 256, NW = 8, NT = 512, BKT = 64;
  constexpr int ROWB = 128, SPR = 8;
  constexpr int TM = 4, TN = 2;
  constexpr int NA = BM * SPR / NT, NB = BN * SPR / NT;
  constexpr int A_BYTES = BM * BKT * 2, STAGE_BYTES = (BM + BN) * BKT * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 * STAGE_BYTES = 128 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int grp = wm;  // waves w and w+4 share a SIMD
  const int nwg = p.tiles_m * p.tiles_n;
  int wg, split;
  g4r_workgroup_tile_slice(nwg, p.dbg == 11, wg, split);
  const int tile_m = p.n_fastest ? wg / p.tiles_n : wg % p.tiles_m;
  const int tile_n = p.n_fastest ? wg % p.tiles_n : wg / p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t_begin = split * p.tiles_per_split;
  int t_end = t_begin + p.tiles_per_split;
  const int nt_total = p.K / BKT;
  if (t_end > nt_total) t_end = nt_total;

  const h16_t* a_src[NA];
  int a_y[NA], a_x[NA];
  const h16_t* b_src[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int pslot = (j * NW + wave) * 64 + lane;
    const int row = pslot / SPR, ps = pslot % SPR;
    const int kslot = ps ^ ((row >> 1) & 7);
    int gm = m0 + row;
    if (gm > p.M - 1) gm = p.M - 1;
    a_src[j] = p.A + (size_t)gm * p.lda + kslot * 8;
    a_y[j] = a_x[j] = 0;
    if (AMODE == 1) {
      const int hw = p.H * p.Wd;
      const int b = gm / hw, rem = gm - b * hw;
      a_y[j] = rem / p.Wd;
      a_x[j] = rem - a_y[j] * p.Wd;
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int pslot = (j * NW + wave) * 64 + lane;
    const int row = pslot / SPR, ps = pslot % SPR;
    const int kslot = ps ^ ((row >> 1) & 7);
    int gn = n0 + row;
    if (gn > p.N - 1) gn = p.N - 1;
    b_src[j] = p.W + (size_t)gn * p.ldw + kslot * 8;
  }
  // One K tile = 8 LDS-DMA pieces per wave (j 0-3: A rows, 4-7: W rows), 1 KiB each.
  struct TileSrc { long a_off; int k0, dy, dx; };
  auto tile_src = [&](int t) {
    TileSrc ts;
    ts.k0 = t * BKT;
    ts.a_off = ts.k0;
    ts.dy = ts.dx = 0;
    if (AMODE == 1) {
      const int n_taps = 9 * p.groups;          // taps fastest, channel slice outer (see gemm_bf16_nt_kernel)
      int ct = t / n_taps;
      int tap_lin = t - ct * n_taps;
      if (p.dbg == 7) {                         // A/B probe (tools only): taps outermost, the round-1 order
        const int per_tap = p.Cin / BKT;
        tap_lin = t / per_tap;
        ct = t - tap_lin * per_tap;
      }
      const int c0 = ct * BKT;
      const int g = tap_lin / 9, tap = tap_lin - g * 9;
      ts.dy = tap / 3 - 1;
      ts.dx = tap - (tap / 3) * 3 - 1;
      ts.a_off = (long)g * p.a_group_stride + ((long)ts.dy * p.Wd + ts.dx) * p.lda + c0;
      ts.k0 = tap_lin * p.Cin + c0;
    }
    return ts;
  };
  auto piece = [&](int j, const TileSrc& ts, int buf) {
    char* sa = smem + buf * STAGE_BYTES;
    if (j < NA) {
      const h16_t* src = a_src[j] + ts.a_off;
      if (AMODE == 1) {
        const int yy = a_y[j] + ts.dy, xx = a_x[j] + ts.dx;
        if (yy < 0 || yy >= p.H || xx < 0 || xx >= p.Wd) src = p.zeros + (lane & 7) * 8;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sa + (j * NW + wave) * 1024), 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[j - NA] + ts.k0),
                                       (__attribute__((address_space(3))) void*)(sa + A_BYTES + ((j - NA) * NW + wave) * 1024), 16, 0, 0);
    }
  };

  float16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fsw = (frow >> 1) & 7, fhi = lane >> 5;
  const int a_row_off = (wm * 128 + frow) * ROWB;
  const int b_row_off = (wn * 64 + frow) * ROWB;
  h16x8 af[2][TM], wf[2][TN];  // the fragments of ONE k-half (2 of the 4 MFMA k-steps)
  auto ldfrag = [&](int buf, int h) {
    const char* sa = smem + buf * STAGE_BYTES;
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int slot = (((h * 2 + k2) * 2 + fhi) ^ fsw) << 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) af[k2][i] = *reinterpret_cast<const h16x8*>(sa + a_row_off + i * 32 * ROWB + slot);
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[k2][j] = *reinterpret_cast<const h16x8*>(sb + b_row_off + j * 32 * ROWB + slot);
    }
  };
  // 16 MFMAs with three issue slots (after the 3rd, 8th and 13th) for LDS-DMA pieces: among MFMAs a piece
  // costs ~60 cycles of issue, against 100+ when all 8 are issued back to back in front of the reads.
  auto mma = [&](auto&& slot) {
    __builtin_amdgcn_s_setprio(1);
    int n = 0;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = G4R_MFMA_32X32X16(wf[k2][j], af[k2][i], acc[i][j], 0, 0, 0);
          ++n;
          if (n == 3 || n == 8 || n == 13) {
            __builtin_amdgcn_sched_barrier(0);
            slot(n == 3 ? 0 : (n == 8 ? 1 : 2));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    __builtin_amdgcn_s_setprio(0);
  };

  // Both groups run the SAME instruction stream; waves 4-7 take one extra barrier before the loop (and
  // waves 0-3 one after it), which skews them by exactly one phase for the whole loop.  The 8 pieces of
  // the next K tile are spread 3/3/2 over three consecutive GLOBAL phases (the buffer they go to is
  // free from global phase 4i and must be full by the end of 4i+3), i.e. local phases 0,1,2 for group 0
  // and 3(previous tile),0,1 for group 1; the wait (vmcnt(0)) is one phase after the last issue.
  const int nt = t_end - t_begin;
  if (nt > 0) {
    {
      const TileSrc ts0 = tile_src(t_begin);
#pragma unroll
      for (int j = 0; j < 8; ++j) piece(j, ts0, 0);
    }
    TileSrc s1 = tile_src(t_begin + 1);
    if (grp == 1 && nt > 1) {
#pragma unroll
      for (int j = 0; j < 3; ++j) piece(j, s1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (grp == 1) G4R_PP_BARRIER();
    // PROBE (tools/pp_probe.py): workgroup 0, waves 0 and 4 stamp s_memtime around every phase of K tile 8
    // and around the whole loop into p.ws (long long [2][8]).
    long long* stamps = reinterpret_cast<long long*>(p.ws) + grp * 8;
    const bool probing = PROBE && blockIdx.x == 0 && blockIdx.y == 0 && (wave & 3) == 0 && lane == 0;
#define G4R_PP_STAMP(slot) \
  if (PROBE) { if (probing && (i == 8 || (slot) >= 6)) stamps[slot] = __builtin_amdgcn_s_memtime(); }
    {
      const int i = 0;
      G4R_PP_STAMP(6);
    }
    for (int i = 0; i < nt; ++i) {
      const int buf = i & 1;
      const bool n1 = i + 1 < nt, n2 = i + 2 < nt;
      const TileSrc s2 = tile_src(t_begin + i + 2);
      G4R_PP_STAMP(0);
      // local phase 0: read k-half 0
      ldfrag(buf, 0);
      if (n1) {
        if (grp == 0) {
#pragma unroll
          for (int j = 0; j < 3; ++j) piece(j, s1, buf ^ 1);
        } else {
#pragma unroll
          for (int j = 3; j < 6; ++j) piece(j, s1, buf ^ 1);
        }
      }
      G4R_PP_STAMP(5);
      G4R_PP_BARRIER();
      G4R_PP_STAMP(1);
      // local phase 1: MFMA k-half 0
      mma([&](int sl) {
        if (n1) {
          if (grp == 0) piece(3 + sl, s1, buf ^ 1);
          else if (sl < 2) piece(6 + sl, s1, buf ^ 1);
        }
      });
      G4R_PP_BARRIER();
      G4R_PP_STAMP(2);
      // local phase 2: read k-half 1
      ldfrag(buf, 1);
      if (grp == 0) {
        if (n1) {
#pragma unroll
          for (int j = 6; j < 8; ++j) piece(j, s1, buf ^ 1);
        }
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // group 1's share of K tile i+1 is in LDS
      }
      G4R_PP_BARRIER();
      G4R_PP_STAMP(3);
      // local phase 3: MFMA k-half 1
      mma([&](int sl) {
        if (n2 && grp == 1) piece(sl, s2, buf);  // K tile i+2 goes where tile i was: both groups are done reading it
      });
      if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // group 0's share of K tile i+1
      G4R_PP_BARRIER();
      G4R_PP_STAMP(4);
      s1 = s2;
    }
    {
      const int i = 0;
      G4R_PP_STAMP(7);
    }
    if (grp == 0) G4R_PP_BARRIER();
  }
  __syncthreads();   // every wave is done with the operand ring: its space now stages the epilogue
  gemm_epilogue_lds<TM, TN>(p, acc, smem + wave * EpiLds<TN>::WAVE_BYTES, m0 + wm * 128, n0 + wn * 64, lane, split);
}

// ---------------------------------------------------------------------------------------------
// Ping-pong kernel, second form: K tiles of 32 in a RING OF FOUR (4 x 32 KB of LDS), one phase per K tile.
// The probe of the first form (tools/pp_probe.py) showed that the LDS-DMA address path is the co-limiter: a CU
// accepts one 1 KiB piece every ~26 cycles, i.e. 1664 cycles of issue per 64 of K against 2048 cycles of MFMA,
// and a wave that is issuing pieces cannot issue MFMAs.  So here ONLY the group that is reading fragments issues
// pieces (its 4 per K tile, for the tile three ahead: 16 pieces = ~416 cycles per phase, under the other group's
// 16 MFMAs), never the group on the matrix pipe, and every wave has five phases of slack for its pieces to land:
//        phase:     2t              2t+1            2t+2
//   waves 0-3:   read(t)+DMA(t+3)   MFMA(t)      read(t+1)+DMA(t+4)
//   waves 4-7:   MFMA(t-1)          read(t)+DMA(t+3)   MFMA(t)
// ---------------------------------------------------------------------------------------------
// BM x BN = 256 x 256 (wave tile 128 x 64) or 128 x 384 (wave tile 64 x 96: 6 x 32 = 192 workgroups for the LLaMA fused-qkv
// shape 767 x 12288, where 256 x 256 tiles give only 144 of the 256 CUs a tile); both stage 32 KB per K tile = 4 LDS-DMA
// pieces per wave, so the ring, the waits and the phase structure are identical.
// (Round-3 probes, since removed: issuing a wave's W pieces -- or all four -- at the HEAD of its MFMA phase instead of the end
// of its read phase changes nothing with global_load_lds pieces (4096^3 1174 / 1141 / 1153 TF/s back to back, conv 192^2
// 860 / 859 / 739; profiles/r03_gemm_bench_a.jsonl) and loses 5 % with the buffer form below (1153 -> 1098; conv 1013 ->
// 895); a piece after every fourth MFMA is 14x slower.  In-kernel stamps with the buffer form: fragment reads 220-300
// cycles + 4 pieces 260-270 (the 16 pieces of a group now go out at the vector-memory path's 16 cycles each) + 92 + 85 of
// wait / barrier = ~670 against 512 + ~60 for the MFMA phase: 1410 cycles per K = 32 tile, was 1481.)
// one 1 KiB LDS-DMA piece as buffer_load_dwordx4 ... lds: descriptor over `base`, per-lane byte offset, scalar byte offset.
// (A free __device__ function: the same builtins written inside a lambda of the kernel make hipcc drop the kernel's host
// handle -- the lambda is implicitly __host__ __device__ and the builtin does not exist on the host.)
__device__ __forceinline__ void g4r_buffer_piece(const void* base, unsigned bytes, void* lds, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// BUF (round 3, the default): the pieces are buffer_load_dwordx4 ... lds through a buffer descriptor -- a 32-bit per-lane
// byte offset computed once + a scalar K / tap offset per tile -- instead of global_load_lds with 64-bit per-lane
// addresses: one SALU add (M0) and one VMEM instruction per piece, no VALU in the read phase for the dense case.  An
// out-of-image conv tap is an offset beyond the descriptor's num_records, which the hardware returns as zeros (no zero
// line, no 64-bit select).  Measured (tools/gemm_bench.cpp, burst arm): 4096^3 1101 -> 1157 TF/s, 8192^2 x 4096
// 1099 -> 1186, the 144-workgroup qkv launch 767 x 12288 x 4096 106.5 -> 88.1 us.  BUF = false (global_load_lds) remains
// for operands of 2 GiB and more, which a 32-bit offset cannot span.
// SCHED 1 (round 3): ONE barrier per K tile and the two groups' loop bodies rotated against each other,
//        interval k:   group 0:  read(k)  pieces(k+3)  MFMA(k)            | vmcnt, barrier
//                      group 1:  MFMA(k-1)  read(k)  pieces(k+3)          | vmcnt, barrier
// instead of the two barriers per tile of SCHED 0, which lock the groups into "one reads while the other multiplies"
// slots of max(read phase, MFMA phase) = ~705 cycles each (1410 per tile; the read phase, ~670, is the longer one).
// Here a wave's tile costs read + MFMA = ~1180 of its own time and the other wave of its SIMD has its MFMAs exactly
// where this one reads (group 1 multiplies tile k-1 during group 0's read of tile k, group 0 multiplies tile k during
// group 1's read of it).  Prefetch distance 2 in the ring of 4: the pieces of tile k+2 overwrite the buffer of tile k-2,
// whose last reads were two barriers ago for BOTH groups; tile k's pieces were waited for (counted vmcnt, leaving only
// tile k+1's in flight) before barrier k by every wave.
// Round 4 (schedules measured and removed again; tools/gemm_bench.cpp, 4096^3 / 8192^2 x 4096 / 3068 x {12288, 4096, 22016}
// x 4096, profiles/r04_gemm_schedules.txt): the wave's four pieces BETWEEN its MFMAs instead of behind its fragment reads
// (unstaggered: 1235 vs 1227 TF/s = equal; staggered over the group's waves through scalar branches: 1084); 3 + 1 and 2 + 2
// splits of the pieces between the two phases (1190 / 1183 vs 1186 = equal); the fragment reads of tile i+1 pipelined into
// the tail of the wave's own MFMA phase (second register set, 236 VGPRs) with the pieces opening the read phase (1120 vs
// 1187: slower).  Every re-arrangement lands within 1 % or loses, because it only moves work between two phases whose SUM per
// wave is fixed: ~260 cycles of fragment reads + ~270 of pieces + 512 of MFMAs + two barrier hand-offs per K tile.  The
// Two further STRUCTURES were written, are correct (tools/gemm_bench.cpp: dense, conv, K slices, ragged shapes) and measure
// the same again -- their text is kept in tools/probe/gemm_pp64_q8_experiment.hip.txt: (a) this kernel's phases over operands
// staged in K = 64 super-tiles, so that every LDS-DMA piece is 8 rows x 128 B = WHOLE cache lines (a CU streams those at 33
// B/clk against 24 for the 16 rows x 64 B pieces that K = 32 tiles force: tools/probe/loadpath_probe.hip) -- 4096^3 1164 vs
// 1146 TF/s, 3068 x 12288 x 4096 1081 vs 1067, 8192^2 1046 vs 1130, conv 192^2 990 vs 1041; (b) the guide's "8-phase" form
// (all eight waves in one role, quadrant phases, K 64 double buffer): 1186 vs 1176 / 1246 vs 1230 at K = 8192 / 1137 vs 1181.
// What all of them share is the CHIP: on ZERO-filled operands this kernel runs 4096^3 at 1537 TF/s (the guide's reference
// template: 1563), on uniform or normal random operands at 1090-1230 on the same box -- the clock under the power cap, not the
// instruction schedule, sets the random-data number (profiles/r04_gemm_dvfs.txt; MFMA-only ablation: 1728 zeros / 1412 random).
// ablations (dbg 21-26, profiles/r04_gemm_ablation.txt) put numbers on it: per K = 32 tile the 32 pieces + reads WITHOUT the
// MFMAs take 0.66 us, the MFMAs + reads WITHOUT the pieces 0.61 us, both together 0.85 us; all 256 workgroups streaming the
// SAME tile's operands (L2-resident) runs within 2 % of the real access pattern -- the limiter is the CU's own load /
// issue budget, not L2 or HBM.  What would cut it is fewer operand bytes and fragment reads per MFMA (a bigger tile than
// 256 x 256 needs one wave per SIMD with 384 accumulator registers), not another order of the same instructions.
template <int AMODE, bool PROBE = false, int BM = 256, int BN = 256, bool BUF = true, int SCHED = 0>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp32_kernel(GemmArgs p) {
  constexpr int NW = 8, NT = 512, BKT = 32, RING = 4;
  constexpr int ROWB = 64, SPR = 4;
  constexpr int WTM = BM / 2, WTN = BN / 4;      // wave tile: 2 row groups (the two ping-pong groups) x 4 column slices
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert((BM + BN == 512 || (BM == 192 && BN == 256)) && WTM % 32 == 0 && WTN % 32 == 0,
                "32 KB (28 KB for 192 x 256) per K tile, 32x32 accumulator blocks");
  // pieces per wave per K tile: 2 + 2; for BM = 192 the A tile has 12 pieces -> waves 0-3 carry two, waves 4-7 one
  constexpr int NA = (BM * SPR + NT - 1) / NT, NB = BN * SPR / NT;
  constexpr bool UNEVEN = (BM * SPR) % NT != 0;
  constexpr int A_BYTES = BM * BKT * 2, STAGE_BYTES = (BM + BN) * BKT * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // RING * STAGE_BYTES = 128 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int grp = wm;
  const int nwg = p.tiles_m * p.tiles_n;
  int wg, split;
  g4r_workgroup_tile_slice(nwg, p.dbg == 11, wg, split);
  int tile_m, tile_n;
  g4r_tile_coords(wg, p.tiles_m, p.tiles_n, p.n_fastest, p.group_m, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  // tools only (ablations of the K loop; results are wrong by construction): dbg 21 = every workgroup streams the operands
  // of tile (0, 0) -- an L2-resident 4 MB working set, so what remains is the CU's own load path; 22 = no pieces after the
  // ring fill (MFMA + fragment reads only); 23 = no MFMAs (loads + fragment reads only); 24 = no fragment reads either
  const int lm0 = p.dbg == 21 ? 0 : m0, ln0 = p.dbg == 21 ? 0 : n0;
  const int t_begin = split * p.tiles_per_split;
  int t_end = t_begin + p.tiles_per_split;
  const int nt_total = p.K / BKT;
  if (t_end > nt_total) t_end = nt_total;
  // PROBE (tools/pp32_probe3.py): every workgroup's wave 0 records entry / loop begin / loop end / exit (s_memtime) and
  // its XCC id at ws + 16 + 8 * blockIdx.x  (int64)
  long long* wg_stamps = reinterpret_cast<long long*>(p.ws) + 16 + 8 * (long)blockIdx.x;
  const bool wg_probe = PROBE && blockIdx.y == 0 && wave == 0 && lane == 0;
  if (PROBE) { if (wg_probe) { wg_stamps[0] = __builtin_amdgcn_s_memtime(); wg_stamps[4] = __builtin_amdgcn_s_getreg(6164); wg_stamps[5] = wall_clock64(); } }

  const h16_t* a_src[NA];
  int a_y[NA], a_x[NA];
  int a_h[NA], a_w[NA];                    // AMODE 2: the row's own map size (rows of several pyramid levels in one GEMM)
  unsigned a_ok[NA];
  int a_pitch[NA];
  const h16_t* b_src[NB];
  int a_voff[NA], b_voff[NB];              // BUF: byte offsets of this lane's 16 bytes from the matrix base (k = 0)
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int pslot = (j * NW + wave) * 64 + lane;
    // tools only, dbg 25 / 26: every piece reads 8 rows x 128 B (whole cache lines) instead of 16 rows x 64 B (half lines)
    const bool full_lines = p.dbg == 25 || p.dbg == 26;
    const int row = full_lines ? pslot / 8 : pslot / SPR, ps = full_lines ? pslot % 8 : pslot % SPR;
    const int kslot = full_lines ? ps : ps ^ ((row >> 2) & 3);
    int gm = lm0 + row;
    if (gm > p.M - 1) gm = p.M - 1;
    a_src[j] = p.A + (size_t)gm * p.lda + kslot * 8;
    a_voff[j] = (gm * p.lda + kslot * 8) * 2;
    a_y[j] = a_x[j] = 0;
    a_h[j] = p.H; a_w[j] = p.Wd;
    if (AMODE == 2) {
      int lv = 0;
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (q < p.n_lvl && gm >= p.lvl_start[q]) lv = q;
      a_h[j] = p.lvl_h[lv]; a_w[j] = p.lvl_w[lv];
      const int local = gm - p.lvl_start[lv];
      const int hw = a_h[j] * a_w[j];
      const int rem = local - (local / hw) * hw;
      a_y[j] = rem / a_w[j];
      a_x[j] = rem - a_y[j] * a_w[j];
    }
    if (AMODE == 1) {
      const int hw = p.H * p.Wd;
      const int b = gm / hw, rem = gm - b * hw;
      a_y[j] = rem / p.Wd;
      a_x[j] = rem - a_y[j] * p.Wd;
    }
    // the nine taps' in-image bits of this lane's pixel and its row pitch in elements: the per-K-tile work of a piece is
    // then one bit test, one multiply-add and a select (the four compares and the 64-bit multiply per piece per K tile of
    // round 2 sat in the READ phase, the longer one of the ping-pong: the conv's K tile ran 1.3x the dense GEMM's)
    a_ok[j] = 0;
    a_pitch[j] = a_w[j] * p.lda;
    if (AMODE >= 1) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int yy = a_y[j] + tp / 3 - 1, xx = a_x[j] + tp % 3 - 1;
        if (yy >= 0 && yy < a_h[j] && xx >= 0 && xx < a_w[j]) a_ok[j] |= 1u << tp;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int pslot = (j * NW + wave) * 64 + lane;
    const bool full_lines = p.dbg == 25 || p.dbg == 26;
    const int row = full_lines ? pslot / 8 : pslot / SPR, ps = full_lines ? pslot % 8 : pslot % SPR;
    const int kslot = full_lines ? ps : ps ^ ((row >> 2) & 3);
    int gn = ln0 + row;
    if (gn > p.N - 1) gn = p.N - 1;
    b_src[j] = p.W + (size_t)gn * p.ldw + kslot * 8;
    b_voff[j] = (gn * p.ldw + kslot * 8) * 2;
  }
  struct TileSrc { long a_off; int k0, dy, dx, tap, shift, soff; };   // shift / soff: the BUF form of a_off (elements / bytes)
  // (channel slice, group, tap) of the NEXT tile to stage: tiles are staged strictly in order, so the walk is a counter
  // (taps fastest, then groups, then 32-channel slices) instead of two runtime divisions per K tile per wave
  int st_ct = 0, st_g = 0, st_tap = 0;
  if (AMODE >= 1) {
    const int n_taps = 9 * p.groups;
    st_ct = t_begin / n_taps;
    const int rem = t_begin - st_ct * n_taps;
    st_g = rem / 9;
    st_tap = rem - st_g * 9;
  }
  auto tile_src = [&](int t) {
    TileSrc ts;
    ts.k0 = t * BKT;
    if (p.dbg == 25 || p.dbg == 26) ts.k0 = ((t * 64) % (p.K - 63)) & ~63;
    ts.a_off = ts.k0;
    ts.dy = ts.dx = ts.tap = ts.shift = 0;
    ts.soff = ts.k0 * 2;
    if (AMODE >= 1) {
      const int c0 = st_ct * BKT;
      ts.tap = st_tap;
      ts.dy = st_tap / 3 - 1;
      ts.dx = st_tap - (st_tap / 3) * 3 - 1;
      ts.a_off = (long)st_g * p.a_group_stride + (long)((ts.dy * p.Wd + ts.dx) * p.lda) + c0;
      if (AMODE == 2) ts.a_off = c0;            // the pixel shift depends on the row's own map width: added per piece
      // buffer form: the (possibly negative) pixel shift goes into the per-lane offset, the scalar offset stays >= 0
      ts.shift = AMODE == 2 ? 0 : (ts.dy * p.Wd + ts.dx) * p.lda;
      ts.soff = (int)(((long)st_g * p.a_group_stride + c0) * 2);
      ts.k0 = (st_g * 9 + st_tap) * p.Cin + c0;
      if (++st_tap == 9) {
        st_tap = 0;
        if (++st_g == p.groups) { st_g = 0; ++st_ct; }
      }
    }
    return ts;
  };
  // piece j of a K tile for this wave: j < NA -> rows of A, else rows of W (1 KiB each)
  auto piece = [&](int j, const TileSrc& ts, int buf) {
    char* sa = smem + buf * STAGE_BYTES;
    if (j < NA) {
      if (UNEVEN && (j * NW + wave) * 16 >= BM) return;        // this wave has no j-th A piece (wave-uniform)
      if (BUF) {
        int voff = a_voff[j];
        if (AMODE == 1) voff += ts.shift * 2;
        if (AMODE == 2) voff += (ts.dy * a_pitch[j] + ts.dx * p.lda) * 2;
        if (AMODE >= 1 && !((a_ok[j] >> ts.tap) & 1u)) voff = (int)0x80000000;      // beyond num_records: reads as zeros
        g4r_buffer_piece(p.A, p.a_bytes, sa + (j * NW + wave) * 1024, voff, ts.soff);
        return;
      }
      const h16_t* src = a_src[j] + ts.a_off;
      if (AMODE == 2) src += ts.dy * a_pitch[j] + ts.dx * p.lda;
      if (AMODE >= 1) {
        if (!((a_ok[j] >> ts.tap) & 1u)) src = p.zeros + (lane & 3) * 8;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sa + (j * NW + wave) * 1024), 16, 0, 0);
    } else {
      if (BUF) {
        g4r_buffer_piece(p.W, p.w_bytes, sa + A_BYTES + ((j - NA) * NW + wave) * 1024, b_voff[j - NA], ts.k0 * 2);
        return;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[j - NA] + ts.k0),
                                       (__attribute__((address_space(3))) void*)(sa + A_BYTES + ((j - NA) * NW + wave) * 1024), 16, 0, 0);
    }
  };
  auto stage = [&](int t, int buf) {
    const TileSrc ts = tile_src(t);
#pragma unroll
    for (int j = 0; j < NA + NB; ++j) piece(j, ts, buf);
  };
  float16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fsw = (frow >> 2) & 3, fhi = lane >> 5;
  const int a_row_off = (wm * WTM + frow) * ROWB;
  const int b_row_off = (wn * WTN + frow) * ROWB;
  h16x8 af[2][TM], wf[2][TN];
  // the 2 x (TM + TN) fragment reads of a K tile in three chunks
  auto ldchunk = [&](int buf, int c) {
    const char* sa = smem + buf * STAGE_BYTES;
    const char* sb = sa + A_BYTES;
    const int s0 = ((0 * 2 + fhi) ^ fsw) << 4, s1 = ((1 * 2 + fhi) ^ fsw) << 4;
    if (c == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const h16x8*>(sa + a_row_off + i * 32 * ROWB + s0);
    } else if (c == 1) {
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[0][j] = *reinterpret_cast<const h16x8*>(sb + b_row_off + j * 32 * ROWB + s0);
#pragma unroll
      for (int i = 0; i < TM / 2; ++i) af[1][i] = *reinterpret_cast<const h16x8*>(sa + a_row_off + i * 32 * ROWB + s1);
    } else {
#pragma unroll
      for (int i = TM / 2; i < TM; ++i) af[1][i] = *reinterpret_cast<const h16x8*>(sa + a_row_off + i * 32 * ROWB + s1);
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[1][j] = *reinterpret_cast<const h16x8*>(sb + b_row_off + j * 32 * ROWB + s1);
    }
  };
  auto mma = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = G4R_MFMA_32X32X16(wf[k2][j], af[k2][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  const int nt = t_end - t_begin;
  if (SCHED == 1 && nt > 0) {
#pragma unroll
    for (int t = 0; t < RING - 1; ++t)
      if (t < nt) stage(t_begin + t, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto tail_wait = [&](int i) {          // end of interval i: tile i+1 has landed (tiles i+2, i+3 may still be in flight)
      if (i + RING - 1 < nt) {
        if (UNEVEN && wave >= 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    if (grp == 0) {
      for (int i = 0; i < nt; ++i) {
        const int buf = i & (RING - 1);
        ldchunk(buf, 0);
        ldchunk(buf, 1);
        ldchunk(buf, 2);
        if (i + RING - 1 < nt) stage(t_begin + i + RING - 1, (i + RING - 1) & (RING - 1));
        mma();
        tail_wait(i);
        G4R_PP_BARRIER();
      }
      G4R_PP_BARRIER();                    // group 1's last interval (its MFMAs of the last tile)
    } else {
      for (int i = 0; i <= nt; ++i) {
        if (i > 0) mma();                  // tile i-1: fragments read in the previous interval
        if (i < nt) {
          const int buf = i & (RING - 1);
          ldchunk(buf, 0);
          ldchunk(buf, 1);
          ldchunk(buf, 2);
          if (i + RING - 1 < nt) stage(t_begin + i + RING - 1, (i + RING - 1) & (RING - 1));
        }
        tail_wait(i);
        G4R_PP_BARRIER();
      }
    }
  }
  if (SCHED == 0 && nt > 0) {
#pragma unroll
    for (int t = 0; t < RING - 1; ++t)
      if (t < nt) stage(t_begin + t, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (grp == 1) G4R_PP_BARRIER();
    long long* stamps = reinterpret_cast<long long*>(p.ws) + grp * 8;
    const bool probing = PROBE && blockIdx.x == 0 && blockIdx.y == 0 && (wave & 3) == 0 && lane == 0;
#define G4R_PP32_STAMP(slot) \
  if (PROBE) { if (probing && (i == 16 || (slot) >= 6)) stamps[slot] = __builtin_amdgcn_s_memtime(); }
    {
      const int i = 0;
      G4R_PP32_STAMP(6);
    }
    if (PROBE) { if (wg_probe) wg_stamps[1] = __builtin_amdgcn_s_memtime(); }
    for (int i = 0; i < nt; ++i) {
      const int buf = i & (RING - 1);
      G4R_PP32_STAMP(0);
      // read phase of K tile i: fragments, then this wave's 4 pieces of tile i+3 into the buffer that tile i-1
      // left (both groups finished reading it: group 1 one phase ago, group 0 two)
      // fragments first, pieces after.  Alternating them inside the wave is much slower (2150-2220 vs 1490 cycles
      // per K tile, with the builtin AND with raw-ISA pieces the compiler cannot see): a ds_read behind an LDS-DMA
      // piece of the same wave waits for it in hardware.
      if (p.dbg != 24 && p.dbg != 25) {
        ldchunk(buf, 0);
        ldchunk(buf, 1);
        ldchunk(buf, 2);
      }
      G4R_PP32_STAMP(1);
      if (i + RING - 1 < nt && p.dbg != 22) {
        stage(t_begin + i + RING - 1, (i + RING - 1) & (RING - 1));
        G4R_PP32_STAMP(2);
        // tile i+1's pieces (issued two read phases ago) have landed: all but the two youngest tiles' pieces are waited for
        if (UNEVEN && wave >= 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      G4R_PP32_STAMP(3);
      G4R_PP_BARRIER();
      G4R_PP32_STAMP(4);
      if (p.dbg != 23 && p.dbg != 24 && p.dbg != 25) mma();
      G4R_PP_BARRIER();
      G4R_PP32_STAMP(5);
    }
    {
      const int i = 0;
      G4R_PP32_STAMP(7);
    }
    if (PROBE) { if (wg_probe) wg_stamps[2] = __builtin_amdgcn_s_memtime(); }
    if (grp == 0) G4R_PP_BARRIER();
  }
  __syncthreads();   // every wave is done with the operand ring: its space now stages the epilogue
  constexpr int EPR = (TN == 2 && TM % 2 == 0) ? 64 : 32;   // rows per epilogue pass: 8 waves x pass must fit the 160 KB of LDS
  gemm_epilogue_lds<TM, TN, EPR>(p, acc, smem + wave * EpiLds<TN, EPR>::WAVE_BYTES, m0 + wm * WTM, n0 + wn * WTN, lane, split);
  if (PROBE) { if (wg_probe) { wg_stamps[3] = __builtin_amdgcn_s_memtime(); wg_stamps[6] = wall_clock64(); } }
}


// ---------------------------------------------------------------------------------------------
// One-wave-per-SIMD kernel: 256 x 256 tile, FOUR waves (2 x 2), wave tile 128 x 128 = 16 accumulators of 32x32
// (all 256 AGPRs of the 512-register budget a lone wave per SIMD has), K tiles of 32 in a ring of four.
// Rationale: the ping-pong kernels above spend two waves per SIMD to overlap one
// wave's operand reads with the other's MFMAs, and pay for it with 24 ds_read_b128 per wave per 64 of K and two
// barriers per phase.  A wave that owns 128 x 128 needs 16 + 16 reads per 64 of K for TWICE the flops (192 KB ->
// 128 KB of LDS reads per CU per 64 of K), keeps next k-step's fragments in flight under its own MFMAs (8 issue
// slots per 32-cycle MFMA, <= 2 of them used), and synchronises once per K tile of 32 (4 waves, not 8).
//   per K tile i (two k-steps of 16, F0/F1 = the two fragment register sets):
//     k-step 0: 16 MFMA on F0 | 8 reads F1 <- (buf i, k1)         | pieces 4-7 of tile i+3
//     k-step 1:  4 MFMA on F1 | lgkmcnt(0), vmcnt(<=16), barrier(i)                            (tile i+1 has landed,
//               12 MFMA on F1 | 8 reads F0 <- (buf i+1, k0)       | pieces 0-3 of tile i+4      buf i is free)
// ---------------------------------------------------------------------------------------------
template <int AMODE>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4_kernel(GemmArgs p) {
  constexpr int BM = 256, BN = 256, NW = 4, NT = 256, BKT = 32, RING = 4;
  constexpr int ROWB = 64, SPR = 4;
  constexpr int TM = 4, TN = 4;
  constexpr int NA = BM * SPR / NT, NB = BN * SPR / NT;  // 4 + 4 pieces per wave per K tile
  constexpr int A_BYTES = BM * BKT * 2, STAGE_BYTES = (BM + BN) * BKT * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // RING * STAGE_BYTES = 128 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = p.tiles_m * p.tiles_n;
  int wg, split;
  g4r_workgroup_tile_slice(nwg, p.dbg == 11, wg, split);
  const int tile_m = p.n_fastest ? wg / p.tiles_n : wg % p.tiles_m;
  const int tile_n = p.n_fastest ? wg % p.tiles_n : wg / p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t_begin = split * p.tiles_per_split;
  int t_end = t_begin + p.tiles_per_split;
  const int nt_total = p.K / BKT;
  if (t_end > nt_total) t_end = nt_total;

  const h16_t* a_src[NA];
  int a_yx[NA];
  const h16_t* b_src[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int pslot = (j * NW + wave) * 64 + lane;
    const int row = pslot / SPR, ps = pslot % SPR;
    const int kslot = ps ^ ((row >> 2) & 3);
    int gm = m0 + row;
    if (gm > p.M - 1) gm = p.M - 1;
    a_src[j] = p.A + (size_t)gm * p.lda + kslot * 8;
    a_yx[j] = 0;
    if (AMODE == 1) {
      const int hw = p.H * p.Wd;
      const int b = gm / hw, rem = gm - b * hw;
      const int y = rem / p.Wd;
      a_yx[j] = (y << 16) | (rem - y * p.Wd);
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int pslot = (j * NW + wave) * 64 + lane;
    const int row = pslot / SPR, ps = pslot % SPR;
    const int kslot = ps ^ ((row >> 2) & 3);
    int gn = n0 + row;
    if (gn > p.N - 1) gn = p.N - 1;
    b_src[j] = p.W + (size_t)gn * p.ldw + kslot * 8;
  }
  struct TileSrc { long a_off; int k0, dy, dx; };
  auto tile_src = [&](int t) {
    TileSrc ts;
    ts.k0 = t * BKT;
    ts.a_off = ts.k0;
    ts.dy = ts.dx = 0;
    if (AMODE == 1) {
      const int n_taps = 9 * p.groups;          // taps fastest, channel slice outer (see gemm_bf16_nt_kernel)
      int ct = t / n_taps;
      int tap_lin = t - ct * n_taps;
      if (p.dbg == 7) {                         // A/B probe (tools only): taps outermost, the round-1 order
        const int per_tap = p.Cin / BKT;
        tap_lin = t / per_tap;
        ct = t - tap_lin * per_tap;
      }
      const int c0 = ct * BKT;
      const int g = tap_lin / 9, tap = tap_lin - g * 9;
      ts.dy = tap / 3 - 1;
      ts.dx = tap - (tap / 3) * 3 - 1;
      ts.a_off = (long)g * p.a_group_stride + ((long)ts.dy * p.Wd + ts.dx) * p.lda + c0;
      ts.k0 = tap_lin * p.Cin + c0;
    }
    return ts;
  };
  // piece j (0-3: rows of A, 4-7: rows of W; 1 KiB each) of K tile `ts` into ring slot `buf`
  auto piece = [&](int j, const TileSrc& ts, int buf) {
    char* sa = smem + buf * STAGE_BYTES;
    if (j < NA) {
      const h16_t* src = a_src[j] + ts.a_off;
      if (AMODE == 1) {
        const int yy = (a_yx[j] >> 16) + ts.dy, xx = (a_yx[j] & 0xffff) + ts.dx;
        if (yy < 0 || yy >= p.H || xx < 0 || xx >= p.Wd) src = p.zeros + (lane & 3) * 8;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sa + (j * NW + wave) * 1024), 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[j - NA] + ts.k0),
                                       (__attribute__((address_space(3))) void*)(sa + A_BYTES + ((j - NA) * NW + wave) * 1024), 16, 0, 0);
    }
  };
  float16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fsw = (frow >> 2) & 3, fhi = lane >> 5;
  const int a_row_off = (wm * 128 + frow) * ROWB;
  const int b_row_off = (wn * 128 + frow) * ROWB;
  const int slot_k0 = ((0 * 2 + fhi) ^ fsw) << 4, slot_k1 = ((1 * 2 + fhi) ^ fsw) << 4;
  h16x8 af0[TM], wf0[TN], af1[TM], wf1[TN];

  auto ldfrag = [&](h16x8 (&AF)[TM], h16x8 (&WF)[TN], int buf, int slot) {
    const char* sa = smem + buf * STAGE_BYTES;
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < TM; ++i) AF[i] = *reinterpret_cast<const h16x8*>(sa + a_row_off + i * 32 * ROWB + slot);
#pragma unroll
    for (int j = 0; j < TN; ++j) WF[j] = *reinterpret_cast<const h16x8*>(sb + b_row_off + j * 32 * ROWB + slot);
  };
  auto mma_row = [&](h16x8 (&AF)[TM], h16x8 (&WF)[TN], auto irow) {
    constexpr int i = decltype(irow)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = G4R_MFMA_32X32X16(WF[j], AF[i], acc[i][j], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  const int nt = t_end - t_begin;
  if (nt > 0) {
    // prologue: tiles 0-2 completely, the first half of tile 3 (its second half is iteration 0's share)
#pragma unroll
    for (int t = 0; t < 3; ++t)
      if (t < nt) {
        const TileSrc ts = tile_src(t_begin + t);
#pragma unroll
        for (int j = 0; j < NA + NB; ++j) piece(j, ts, t);
      }
    if (3 < nt) {
      const TileSrc ts = tile_src(t_begin + 3);
#pragma unroll
      for (int j = 0; j < 4; ++j) piece(j, ts, 3);
    }
    if (nt > 3) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (nt > 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (nt > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ldfrag(af0, wf0, 0, slot_k0);
    // sched_group_barrier masks (LLVM SchedGroupMask): 0x8 MFMA, 0x10 VMEM, 0x100 DS read
    auto body = [&](int i, auto steady_tag) {
      constexpr bool STEADY = decltype(steady_tag)::value;   // tiles i+1 .. i+4 all exist: no branches in the body
      const int buf = i & (RING - 1);
      // ---- k-step 0: 16 MFMA on F0 | F1 <- (buf, k1) one read per MFMA | pieces 4-7 of tile i+3 every 2nd MFMA ----
      ldfrag(af1, wf1, buf, slot_k1);
      if (STEADY || i + 3 < nt) {
        const TileSrc ts = tile_src(t_begin + i + 3);
#pragma unroll
        for (int j = 4; j < 8; ++j) piece(j, ts, (i + 3) & (RING - 1));
      }
      mma_row(af0, wf0, I0{}); mma_row(af0, wf0, I1{}); mma_row(af0, wf0, I2{}); mma_row(af0, wf0, I3{});
      if (STEADY) {
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- k-step 1, first row: the pipe stays busy while the waves meet at the barrier ----
      mma_row(af1, wf1, I0{});
      // every fragment of buf i is in registers (the MFMAs above needed af1[0] and all of wf1; the explicit wait
      // covers af1[1..3]); tile i+1 must have landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (STEADY || i + 3 < nt) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (i + 2 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- k-step 1, rows 1-3: 12 MFMA | F0 <- (buf i+1, k0) | pieces 0-3 of tile i+4 into the slot tile i left ----
      if (STEADY || i + 1 < nt) ldfrag(af0, wf0, (i + 1) & (RING - 1), slot_k0);
      if (STEADY || i + 4 < nt) {
        const TileSrc ts = tile_src(t_begin + i + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) piece(j, ts, buf);
      }
      mma_row(af1, wf1, I1{}); mma_row(af1, wf1, I2{}); mma_row(af1, wf1, I3{});
      if (STEADY) {
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 1);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 1);
          __builtin_amdgcn_sched_group_barrier(0x10, 1, 1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    int i = 0;
    for (; i + 4 < nt; ++i) body(i, std::true_type{});
    for (; i < nt; ++i) body(i, std::false_type{});
  }
  __syncthreads();   // the operand ring is idle: stage the epilogue through it
  gemm_epilogue_lds<TM, TN>(p, acc, smem + wave * EpiLds<TN>::WAVE_BYTES, m0 + wm * 128, n0 + wn * 128, lane, split);
}

template <int AMODE>
int launch_w4(GemmArgs& p, hipStream_t stream) {
  {
    const int nt = p.K / 32;
    int splits = p.splits < 1 ? 1 : p.splits;
    if (splits > nt) splits = nt;
    p.tiles_per_split = g4r_ceil_div(nt, splits);
    p.splits = g4r_ceil_div(nt, p.tiles_per_split);
  }
  p.tiles_m = g4r_ceil_div(p.M, 256);
  p.tiles_n = g4r_ceil_div(p.N, 256);
  const size_t ring = 4 * (256 + 256) * 32 * 2, epi = 4 * (size_t)EpiLds<4>::WAVE_BYTES;
  const size_t lds = ring > epi ? ring : epi;
  auto kern = gemm_bf16_w4_kernel<AMODE>;
  static G4rPerDeviceOnce attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { attr_set.failed(); return g4r_note_hip_error(e, "gemm_w4: hipFuncSetAttribute"); }
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, p.splits), dim3(256), lds, stream, p);
  G4R_CHECK_LAUNCH("gemm_bf16_w4");
  if (p.splits > 1 && !p.defer_reduce) {
    long total = (long)p.M * p.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    G4R_CHECK_LAUNCH("splitk_reduce");
  }
  return G4R_OK;
}

// ---------------------------------------------------------------------------------------------
// Epilogue of the one-wave-per-SIMD kernel (round 5): wave tile 128 x 128, FOUR waves.  In-kernel stamps of the first form
// (tools/wg_timeline.py, profiles/r05_wg_timeline.txt) put the K loop at 2221 cycles per 64 of K (MFMA floor 2048) and the
// shared epilogue above at 24.5 k cycles per tile -- 14 % of a K = 4096 tile: with half the waves of the ring kernel every
// wave runs twice the number of its serial read-back iterations (LDS read -> mode branches -> one store each).
// Here the mode is a TEMPLATE parameter of the kernel (the launcher picks it), so every instantiation carries one
// straight-line epilogue whose LDS reads and global accesses the compiler batches:
//   W4_P16  16-bit park, ONE pass (plain / bias / activation): the values are final BEFORE they are parked, so the wave parks
//           16-bit values (128 rows x 256 B) and reads them back as whole 16-byte runs of a row;
//   W4_SWIGLU  the (gate, up) column pairs sit in one lane: 64 outputs per row, parked as 16-bit;
//   W4_ROPE    the fused RoPE + KV-cache append of q|k|v: the projection is parked at the rounding point the unfused path had,
//              the wave's 128 columns are ONE head, so the rotation partner is column d +- 64 of the same parked row;
//   W4_WIDE    fp32 park, two passes of 64 rows (residual, fp32 output, K-slice partials): arithmetic after the park;
//   W4_GENERIC gemm_epilogue_lds (ragged N, unaligned rows).
// Same arithmetic and order as gemm_epilogue_lds (bias -> activation -> residual -> one rounding): bit-identical results.
// A wave whose 128 columns are not all inside N, or whose tile crosses row M, takes the guarded forms.
// ---------------------------------------------------------------------------------------------
enum { W4_GENERIC = 0, W4_P16 = 1, W4_SWIGLU = 2, W4_ROPE = 3, W4_WIDE = 4 };
#define G4R_W4_STAGGER_TICKS 0     // default start de-phasing step (10 ns ticks); see gemm_bf16_w4k64_kernel
struct W4Epi {
  static constexpr int RS16 = 256 + 16;         // 128 x 16-bit row + 16 B (rows stay 16-byte aligned for the b128 read-back)
  static constexpr int RSW = 128 + 16;          // SwiGLU: 64 x 16-bit outputs per row
  static constexpr int RS32 = 512 + 16;         // 128 x fp32 row + 16 B (EpiLds<4>::RS)
  static constexpr int WAVE_BYTES = 128 * RS16;  // >= 64 * RS32
};
static_assert(64 * W4Epi::RS32 <= W4Epi::WAVE_BYTES && 128 * W4Epi::RSW <= W4Epi::WAVE_BYTES && W4Epi::WAVE_BYTES >= EpiLds<4>::WAVE_BYTES,
              "every mode fits the wave's slice");

// One accumulator register -> VGPR, in program order.  Without it the compiler copies all 256 AGPRs to VGPRs at the head of
// the epilogue (v_accvgpr_read x 256 -> 548 bytes of scratch per lane); the volatile asm keeps every value in its AGPR until
// the park loop reaches it.
__device__ __forceinline__ float w4_acc(const float16v& c, int r) {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(c[r]));
  return v;
}

// read-back of a parked 16-bit tile: NR rows of RB bytes (LPR lanes x 16 B per row), whole rows of the output
template <int NR, int RS, int LPR, bool GUARD>
__device__ __forceinline__ void w4_copy_rows(const char* wave_lds, h16_t* dst0, long ldc, int lane, int rows_left) {
  constexpr int RPI = 64 / LPR;
  const int rrow = lane / LPR, rchunk = lane % LPR;
  const char* src = wave_lds + rrow * RS + rchunk * 16;
  h16_t* dst = dst0 + (long)rrow * ldc + rchunk * 8;
#pragma unroll
  for (int b = 0; b < NR / RPI; b += 8) {
    uint4v d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = *reinterpret_cast<const uint4v*>(src + (b + u) * RPI * RS);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (!GUARD || (b + u) * RPI + rrow < rows_left) *reinterpret_cast<uint4v*>(dst + (long)(b + u) * RPI * ldc) = d[u];
  }
}

template <int MODE, bool PROBE>
__device__ __forceinline__ void w4_epilogue(const GemmArgs& p, float16v (&acc)[4][4], char* wave_lds, int m_wave0, int n_wave0,
                                            int lane, int split, long long& park_t) {
  if (MODE == W4_GENERIC || n_wave0 + 128 > p.N) {         // (wave-uniform; W4_ROPE never sees a ragged tile: heads * 128 % 256 == 0)
    if (MODE != W4_ROPE) gemm_epilogue_lds<4, 4>(p, acc, wave_lds, m_wave0, n_wave0, lane, split);
    return;
  }
  if (m_wave0 >= p.M) return;                               // the whole wave tile is below the last row (wave-uniform)
  const int wr = lane & 31, wh = lane >> 5;
  const int rows_left = p.M - m_wave0;
  const bool full = rows_left >= 128;
  if (MODE == W4_SWIGLU) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float16v& c = acc[i][j];
          const float g0 = w4_acc(c, q * 4), u0 = w4_acc(c, q * 4 + 1), g1 = w4_acc(c, q * 4 + 2), u1 = w4_acc(c, q * 4 + 3);
          const float s0 = h16lo(pack_h16x2(g0 / (1.f + __expf(-g0)), 0.f));
          const float s1 = h16lo(pack_h16x2(g1 / (1.f + __expf(-g1)), 0.f));
          *reinterpret_cast<uint32_t*>(wave_lds + (i * 32 + wr) * W4Epi::RSW + (j * 16 + q * 4 + wh * 2) * 2) = pack_h16x2(s0 * u0, s1 * u1);
        }
    if (PROBE) park_t = __builtin_amdgcn_s_memtime();
    h16_t* dst0 = reinterpret_cast<h16_t*>(p.C) + (size_t)m_wave0 * p.ldc + (n_wave0 >> 1);
    if (full) w4_copy_rows<128, W4Epi::RSW, 8, false>(wave_lds, dst0, p.ldc, lane, rows_left);
    else w4_copy_rows<128, W4Epi::RSW, 8, true>(wave_lds, dst0, p.ldc, lane, rows_left);
    return;
  }
  if (MODE == W4_P16 || MODE == W4_ROPE) {
    // ---- 16-bit park: bias / activation applied in the accumulator layout (RoPE: the projection rounded as the unfused path did) ----
    const bool has_bias = MODE == W4_P16 && p.bias != nullptr;
    auto park16 = [&](auto act_tag) {                     // the activation is a compile-time constant of each copy
      constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4v b = {0.f, 0.f, 0.f, 0.f};
          if (has_bias) b = *reinterpret_cast<const float4v*>(p.bias + n_wave0 + j * 32 + q * 8 + wh * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float16v& c = acc[i][j];
            float v[4] = {w4_acc(c, q * 4) + b.x, w4_acc(c, q * 4 + 1) + b.y, w4_acc(c, q * 4 + 2) + b.z, w4_acc(c, q * 4 + 3) + b.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float x = v[k];
              v[k] = ACT == 0 ? x : (ACT == 1 ? fmaxf(x, 0.f) : (ACT == 2 ? x / (1.f + __expf(-1.702f * x)) : x / (1.f + __expf(-x))));
            }
            *reinterpret_cast<uint2v*>(wave_lds + (i * 32 + wr) * W4Epi::RS16 + (j * 32 + q * 8 + wh * 4) * 2) =
                uint2v{pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3])};
          }
        }
    };
    const int act = MODE == W4_P16 ? p.act : 0;
    if (act == 0) park16(std::integral_constant<int, 0>{});
    else if (act == 1) park16(std::integral_constant<int, 1>{});
    else if (act == 2) park16(std::integral_constant<int, 2>{});
    else park16(std::integral_constant<int, 3>{});
    if (PROBE) park_t = __builtin_amdgcn_s_memtime();
    if (MODE == W4_P16) {
      h16_t* dst0 = reinterpret_cast<h16_t*>(p.C) + (size_t)m_wave0 * p.ldc + n_wave0;
      if (full) w4_copy_rows<128, W4Epi::RS16, 16, false>(wave_lds, dst0, p.ldc, lane, rows_left);
      else w4_copy_rows<128, W4Epi::RS16, 16, true>(wave_lds, dst0, p.ldc, lane, rows_left);
      return;
    }
    // fused RoPE + KV-cache append (gemm_epilogue_lds, act 5, states the arithmetic)
    const int rrow = lane >> 4, rchunk = lane & 15;      // 4 rows x 256 B per wave instruction
    const int part = n_wave0 / p.rope_HD;                // 0 q, 1 k, 2 v
    const int col0 = n_wave0 - part * p.rope_HD + rchunk * 8;
    const bool second = rchunk >= 8;
    const int d0 = (rchunk & 7) * 8;                     // position of this run inside its half of the head
#pragma unroll 4
    for (int it = 0; it < 32; ++it) {
      const int r = it * 4 + rrow;
      int m = m_wave0 + r;
      const bool live = m < p.M;
      if (!live) m = p.M - 1;                            // (reads stay in range; the store is masked)
      const uint4v own = *reinterpret_cast<const uint4v*>(wave_lds + r * W4Epi::RS16 + rchunk * 16);
      const uint4v mate = *reinterpret_cast<const uint4v*>(wave_lds + r * W4Epi::RS16 + (rchunk ^ 8) * 16);
      const int b = m / p.rope_T, t = m - b * p.rope_T;
      const int pos = p.rope_pos0 + t;
      uint4v o = own;
      if (part < 2) {
        const float* cp = p.rope_cos + (size_t)pos * 64 + d0;
        const float* sp = p.rope_sin + (size_t)pos * 64 + d0;
        const float4v c0 = *reinterpret_cast<const float4v*>(cp), c1 = *reinterpret_cast<const float4v*>(cp + 4);
        const float4v s0 = *reinterpret_cast<const float4v*>(sp), s1 = *reinterpret_cast<const float4v*>(sp + 4);
        const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const uint32_t ow[4] = {own.x, own.y, own.z, own.w}, mw[4] = {mate.x, mate.y, mate.z, mate.w};
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a0 = h16lo(ow[k]), a1 = h16hi(ow[k]), b0 = h16lo(mw[k]), b1 = h16hi(mw[k]);
          v[2 * k] = second ? __builtin_fmaf(a0, cs[2 * k], b0 * sn[2 * k]) : __builtin_fmaf(a0, cs[2 * k], -(b0 * sn[2 * k]));
          v[2 * k + 1] = second ? __builtin_fmaf(a1, cs[2 * k + 1], b1 * sn[2 * k + 1])
                                : __builtin_fmaf(a1, cs[2 * k + 1], -(b1 * sn[2 * k + 1]));
        }
        o = uint4v{pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3]), pack_h16x2(v[4], v[5]), pack_h16x2(v[6], v[7])};
      }
      h16_t* dst = part == 0 ? p.rope_q + (size_t)m * p.rope_HD + col0
                              : (part == 1 ? p.rope_k : p.rope_v) + (size_t)b * p.rope_kbatch + (size_t)pos * p.rope_krow + col0;
      if (live) *reinterpret_cast<uint4v*>(dst) = o;
    }
    return;
  }
  // ---- W4_WIDE: fp32 park, two passes of 64 rows: residual / fp32 output / K-slice partials ----
  const int rrow = lane >> 4, rcol = (lane & 15) * 8;
  float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr && p.splits == 1) {
    const float4v b0 = *reinterpret_cast<const float4v*>(p.bias + n_wave0 + rcol), b1 = *reinterpret_cast<const float4v*>(p.bias + n_wave0 + rcol + 4);
    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
  }
  const int n = n_wave0 + rcol;
  auto readback = [&](int mh, auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;        // -1: K-slice partials
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int r = it * 4 + rrow;
      int m = mh + r;
      const bool live = m < p.M;
      if (!live) m = p.M - 1;                            // (reads stay in range; the stores are masked)
      const float4v a0 = *reinterpret_cast<const float4v*>(wave_lds + r * W4Epi::RS32 + rcol * 4);
      const float4v a1 = *reinterpret_cast<const float4v*>(wave_lds + r * W4Epi::RS32 + rcol * 4 + 16);
      if (ACT < 0) {
        float* dst = p.ws + ((size_t)split * p.M + m) * p.N + n;
        if (live) {
          *reinterpret_cast<float4v*>(dst) = a0;
          *reinterpret_cast<float4v*>(dst + 4) = a1;
        }
        continue;
      }
      float v[8] = {a0.x + bias8[0], a0.y + bias8[1], a0.z + bias8[2], a0.w + bias8[3],
                    a1.x + bias8[4], a1.y + bias8[5], a1.z + bias8[6], a1.w + bias8[7]};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float x = v[k];
        v[k] = ACT <= 0 ? x : (ACT == 1 ? fmaxf(x, 0.f) : (ACT == 2 ? x / (1.f + __expf(-1.702f * x)) : x / (1.f + __expf(-x))));
      }
      if (p.residual) {
        const uint4v rr = *reinterpret_cast<const uint4v*>(p.residual + (size_t)m * p.ldr + n);
        v[0] += h16lo(rr.x); v[1] += h16hi(rr.x); v[2] += h16lo(rr.y); v[3] += h16hi(rr.y);
        v[4] += h16lo(rr.z); v[5] += h16hi(rr.z); v[6] += h16lo(rr.w); v[7] += h16hi(rr.w);
      }
      if (p.out_f32) {
        float* dst = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
        if (live) {
          *reinterpret_cast<float4v*>(dst) = float4v{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<float4v*>(dst + 4) = float4v{v[4], v[5], v[6], v[7]};
        }
      } else if (live) {
        *reinterpret_cast<uint4v*>(reinterpret_cast<h16_t*>(p.C) + (size_t)m * p.ldc + n) =
            uint4v{pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3]), pack_h16x2(v[4], v[5]), pack_h16x2(v[6], v[7])};
      }
    }
  };
  const int act = p.splits > 1 ? -1 : p.act;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float16v& c = acc[half * 2 + i2][j];
          *reinterpret_cast<float4v*>(wave_lds + (i2 * 32 + wr) * W4Epi::RS32 + (j * 32 + q * 8 + wh * 4) * 4) =
              float4v{c[q * 4], c[q * 4 + 1], c[q * 4 + 2], c[q * 4 + 3]};       // (ds_write_b128 takes its data straight from AGPRs)
        }
    if (PROBE && half == 0) park_t = __builtin_amdgcn_s_memtime();
    const int mh = m_wave0 + half * 64;
    if (act < 0) readback(mh, std::integral_constant<int, -1>{});
    else if (act == 0) readback(mh, std::integral_constant<int, 0>{});
    else if (act == 1) readback(mh, std::integral_constant<int, 1>{});
    else if (act == 2) readback(mh, std::integral_constant<int, 2>{});
    else readback(mh, std::integral_constant<int, 3>{});
  }
}

// ---------------------------------------------------------------------------------------------
// Round 5 -- one-wave-per-SIMD kernel, second form ("w4k64", tile_cfg 34): 256 x 256 tile, FOUR waves (2 x 2), wave tile
// 128 x 128 (16 accumulators of 32x32 = all 256 AGPRs), K tiles of 64 in TWO LDS buffers that are refilled as soon as they
// are consumed.
// Why (profiles/r05_vendor_ab.txt): a same-process A/B showed the vendor library at 1550 TF/s SUSTAINED on random operands
// where the ring ping-pong kernel above does 1160-1235 -- the "power-cap ceiling" of round 4 was a ceiling of THAT kernel.
// Its 8 waves x (128 x 64) read 192 KB of fragments per CU per 64 of K; 4 waves x (128 x 128) read 128 KB for the same
// flops, every fragment read and every LDS-DMA piece can be hidden under the OWN wave's MFMAs (one wave per SIMD: 7 free
// issue slots per 32-cycle MFMA), and there are two workgroup barriers per 64 MFMAs instead of four hand-offs.
// Layout: a K tile is rows of 128 B (64 elements); a 1 KiB LDS-DMA piece = 8 rows x 128 B = WHOLE cache lines; the 16-byte
// slot of k-chunk c of row r sits at slot c ^ ((r >> 1) & 7) (source-side swizzle, conflict-free for the lane groups of
// ds_read_b128, MI355X_MICROARCH.md LDS table).  Per K tile a wave issues 8 + 8 pieces and 16 + 16 fragment reads.
// Schedule of tile i (buffer X = i & 1; H0 / H1 = the fragments of k-steps 0,1 / 2,3, 64 + 64 VGPRs):
//   A: 20 MFMA on H0           | 16 reads H1 <- X                                   | lgkmcnt(0), barrier  (X is consumed)
//   B: 24 MFMA on H0 / H1      | 12 pieces of tile i+2 -> X                         | vmcnt(12), barrier   (tile i+1 landed)
//   C: 20 MFMA on H1           | 16 reads H0 <- X^1 (tile i+1), last 4 pieces       |
// so a tile's pieces have 1.3 tiles (~2700 cycles) to land.  AMODE 1 / 2 (implicit-GEMM convolution): the tap shift goes
// into the per-lane offset, an out-of-image tap is an offset beyond the descriptor's extent (zeros), K tile = 64 channels
// of one tap.
// ---------------------------------------------------------------------------------------------
template <int AMODE, bool PROBE, int EPI, bool MI16P = false>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4k64_kernel(GemmArgs p) {
  constexpr int BM = 256, BN = 256, NW = 4, BKT = 64, ROWB = 128, NP = 8;
  constexpr int TM = 4, TN = 4;
  constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;      // 32 KB + 32 KB per K tile
  extern __shared__ __attribute__((aligned(16))) char smem[];            // 2 * STAGE_BYTES = 128 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = p.tiles_m * p.tiles_n;
  int wg, split;
  g4r_workgroup_tile_slice(nwg, false, wg, split);
  int tile_m, tile_n;
  g4r_tile_coords(wg, p.tiles_m, p.tiles_n, p.n_fastest, p.group_m, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t_begin = split * p.tiles_per_split;
  int t_end = t_begin + p.tiles_per_split;
  const int nt_total = p.K / BKT;
  if (t_end > nt_total) t_end = nt_total;
  // PROBE (tools/wg_timeline.py, tile 35): wave 0 of every workgroup records entry / loop begin / loop end / exit (s_memtime),
  // its XCC id and the 100 MHz wall clock at ws + 16 + 8 * blockIdx.x (int64) -- the convention of the ring kernel's probe
  long long* wg_stamps = reinterpret_cast<long long*>(p.ws) + 16 + 8 * (long)blockIdx.x;
  const bool wg_probe = PROBE && blockIdx.y == 0 && wave == 0;       // (wave-uniform: all 64 lanes store the same stamp)
  if (PROBE) { if (wg_probe) { wg_stamps[0] = __builtin_amdgcn_s_memtime(); wg_stamps[4] = __builtin_amdgcn_s_getreg(6164); wg_stamps[5] = wall_clock64(); } }

  // De-phasing (round 5): every workgroup of a launch runs the same number of K tiles, so all 256 CUs reach their epilogue at
  // the same moment and the 32 MB of a wave's outputs hit the memory system as ONE burst while every matrix pipe idles.  The
  // first 256 workgroups (one per CU) start `phase x stagger_ticks x 10 ns` late, phase = 0..7 by CU slot inside the XCD; later
  // workgroups inherit the phase of the CU they land on, so at any moment only ~1/8 of the CUs are storing.
  if (p.stagger_ticks > 0) {
    const int lin = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    const int phase = (lin >> 3) & 7;
    if (lin < 256 && phase != 0) {
      const long long t_go = wall_clock64() + (long long)phase * p.stagger_ticks;
      while (wall_clock64() < t_go) __builtin_amdgcn_s_sleep(8);
    }
  }

  // piece j of this wave covers rows 8 * (4 j + wave) + lane / 8; lane % 8 is the 16-byte slot it writes
  int a_voff[NP], b_voff[NP];
  unsigned a_ok[NP];
  int a_pitch[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int row = (j * NW + wave) * 8 + (lane >> 3);
    const int kslot = (lane & 7) ^ ((row >> 1) & 7);
    int gm = m0 + row;
    if (gm > p.M - 1) gm = p.M - 1;
    a_voff[j] = (gm * p.lda + kslot * 8) * 2;
    int gn = n0 + row;
    if (gn > p.N - 1) gn = p.N - 1;
    b_voff[j] = (gn * p.ldw + kslot * 8) * 2;
    a_ok[j] = 0;
    a_pitch[j] = p.Wd * p.lda;
    if (AMODE >= 1) {
      int h = p.H, w = p.Wd, local = gm;
      if (AMODE == 2) {
        int lv = 0;
#pragma unroll
        for (int q = 1; q < 4; ++q)
          if (q < p.n_lvl && gm >= p.lvl_start[q]) lv = q;
        h = p.lvl_h[lv]; w = p.lvl_w[lv];
        local = gm - p.lvl_start[lv];
        a_pitch[j] = w * p.lda;
      }
      const int hw = h * w;
      const int rem = local - (local / hw) * hw;
      const int y = rem / w, x = rem - y * w;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) a_ok[j] |= 1u << tp;
      }
    }
  }
  // the NEXT K tile to stage (tiles are staged strictly in order): dense = a K offset; conv = (64-channel slice, group, tap),
  // taps fastest, walked by a counter
  int st_t = t_begin, st_ct = 0, st_g = 0, st_tap = 0;
  if (AMODE >= 1) {
    const int n_taps = 9 * p.groups;
    st_ct = t_begin / n_taps;
    const int rem = t_begin - st_ct * n_taps;
    st_g = rem / 9;
    st_tap = rem - st_g * 9;
  }
  struct TileSrc { int a_soff, w_soff, tap, dy, dx; };
  auto next_tile = [&]() {
    TileSrc ts;
    ts.tap = ts.dy = ts.dx = 0;
    ts.a_soff = ts.w_soff = st_t * BKT * 2;
    ++st_t;
    if (AMODE >= 1) {
      const int c0 = st_ct * BKT;
      ts.tap = st_tap;
      ts.dy = st_tap / 3 - 1;
      ts.dx = st_tap - (st_tap / 3) * 3 - 1;
      ts.a_soff = (int)(((long)st_g * p.a_group_stride + c0) * 2);
      ts.w_soff = ((st_g * 9 + st_tap) * p.Cin + c0) * 2;
      if (++st_tap == 9) {
        st_tap = 0;
        if (++st_g == p.groups) { st_g = 0; ++st_ct; }
      }
    }
    return ts;
  };
  // piece q of a K tile for this wave: q < 8 -> rows of A, else rows of W
  auto piece = [&](int q, const TileSrc& ts, int buf) {
    char* sa = smem + buf * STAGE_BYTES;
    if (q < NP) {
      int voff = a_voff[q];
      if (AMODE == 1) voff += (ts.dy * p.Wd + ts.dx) * p.lda * 2;
      if (AMODE == 2) voff += (ts.dy * a_pitch[q] + ts.dx * p.lda) * 2;
      if (AMODE >= 1 && !((a_ok[q] >> ts.tap) & 1u)) voff = (int)0x80000000;      // beyond num_records: reads as zeros
      g4r_buffer_piece(p.A, p.a_bytes, sa + (q * NW + wave) * 1024, voff, ts.a_soff);
    } else {
      g4r_buffer_piece(p.W, p.w_bytes, sa + A_BYTES + ((q - NP) * NW + wave) * 1024, b_voff[q - NP], ts.w_soff);
    }
  };
  float16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fsw = (frow >> 1) & 7, fhi = lane >> 5;
  const int a_row_off = (wm * 128 + frow) * ROWB;
  const int b_row_off = A_BYTES + (wn * 128 + frow) * ROWB;
  h16x8 fa[4][TM], fw[4][TN];                  // [k-step][block]: k-steps 0,1 = H0, 2,3 = H1
  auto ldfrag = [&](auto ks_tag, int buf) {
    constexpr int ks = decltype(ks_tag)::value;
    const char* sb = smem + buf * STAGE_BYTES;
    const int slot = ((2 * ks + fhi) ^ fsw) << 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const h16x8*>(sb + a_row_off + i * 32 * ROWB + slot);
#pragma unroll
    for (int j = 0; j < TN; ++j) fw[ks][j] = *reinterpret_cast<const h16x8*>(sb + b_row_off + j * 32 * ROWB + slot);
  };
  // MI16P (tools build, debug mode 65; results are WRONG by construction): the same loads, fragment reads and barriers, but every
  // 32 x 32 x 16 product is replaced by TWO v_mfma_f32_16x16x32 on four-register accumulators -- the same FLOPs per K tile through
  // the other MFMA shape, to measure what that shape buys under the power cap with the real kernel's traffic around it
  float4v acc4[MI16P ? TM : 1][MI16P ? TN : 1][4];
  if (MI16P) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[MI16P ? i : 0][MI16P ? j : 0][r] = float4v{0.f, 0.f, 0.f, 0.f};
  }
  auto mma_rows = [&](auto ks_tag, auto i0_tag, auto i1_tag) {    // rows [i0, i1) of k-step ks
    constexpr int ks = decltype(ks_tag)::value, i0 = decltype(i0_tag)::value, i1 = decltype(i1_tag)::value;
#pragma unroll
    for (int i = i0; i < i1; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (MI16P) {
          constexpr int r0 = 2 * (ks & 1);
          acc4[i][j][r0] = G4R_MFMA_16X16X32(fw[ks][j], fa[ks][i], acc4[i][j][r0], 0, 0, 0);
          acc4[i][j][r0 + 1] = G4R_MFMA_16X16X32(fa[ks][i], fw[ks][j], acc4[i][j][r0 + 1], 0, 0, 0);
        } else {
          acc[i][j] = G4R_MFMA_32X32X16(fw[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
        }
      }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;

  const int nt = t_end - t_begin;
  if (nt > 0) {
    {
      const TileSrc ts = next_tile();
#pragma unroll
      for (int q = 0; q < 2 * NP; ++q) piece(q, ts, 0);
    }
    if (nt > 1) {
      const TileSrc ts = next_tile();
#pragma unroll
      for (int q = 0; q < 2 * NP; ++q) piece(q, ts, 1);
      __builtin_amdgcn_s_waitcnt(0x4f70);                     // vmcnt(16): tile 0 has landed
    } else {
      __builtin_amdgcn_s_waitcnt(0x0f70);                     // vmcnt(0)
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ldfrag(I0{}, 0);
    ldfrag(I1{}, 0);
    // sched_group_barrier masks (LLVM SchedGroupMask): 0x8 MFMA, 0x10 VMEM, 0x100 DS read
    auto body = [&](int i, auto steady_tag) {
      constexpr bool STEADY = decltype(steady_tag)::value;      // tiles i+1, i+2 exist: no branches in the body
      const int buf = i & 1;
      // ---- A: k-step 0 + the first row of k-step 1 | H1 <- X, one read per MFMA ----
      ldfrag(I2{}, buf);
      ldfrag(I3{}, buf);
      mma_rows(I0{}, I0{}, I4{});
      mma_rows(I1{}, I0{}, I1{});
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): every fragment of X is in registers (the builtin,
      asm volatile("" ::: "memory");                           //   not inline asm: the compiler's counter model sees it)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                            // ... in every wave: X may be overwritten
      __builtin_amdgcn_sched_barrier(0);
      // ---- B: rest of k-step 1 + three rows of k-step 2 | pieces 0-11 of tile i+2 -> X, one per two MFMAs ----
      TileSrc ts2 = {};
      const bool more = STEADY || i + 2 < nt;
      if (more) {
        ts2 = next_tile();
#pragma unroll
        for (int q = 0; q < 12; ++q) piece(q, ts2, buf);
      }
      mma_rows(I1{}, I1{}, I4{});
      mma_rows(I2{}, I0{}, I3{});
      if (STEADY) {
#pragma unroll
        for (int n = 0; n < 12; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x8, 2, 1);
          __builtin_amdgcn_sched_group_barrier(0x10, 1, 1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) __builtin_amdgcn_s_waitcnt(0x0f7c);            // vmcnt(12): tile i+1 has landed (only this tile's 12 in flight)
      else __builtin_amdgcn_s_waitcnt(0x0f70);                 // vmcnt(0)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- C: last row of k-step 2 + k-step 3 | H0 <- X^1 (tile i+1), one read per MFMA; the last 4 pieces ----
      if (STEADY || i + 1 < nt) {
        ldfrag(I0{}, buf ^ 1);
        ldfrag(I1{}, buf ^ 1);
      }
      if (more) {
#pragma unroll
        for (int q = 12; q < 16; ++q) piece(q, ts2, buf);
      }
      mma_rows(I2{}, I3{}, I4{});
      mma_rows(I3{}, I0{}, I4{});
      if (STEADY) {
#pragma unroll
        for (int n = 0; n < 16; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 2);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 2);
          __builtin_amdgcn_sched_group_barrier(0x10, 1, 2);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (PROBE) { if (wg_probe) wg_stamps[1] = __builtin_amdgcn_s_memtime(); }
    int i = 0;
    for (; i + 2 < nt; ++i) body(i, std::true_type{});
    for (; i < nt; ++i) body(i, std::false_type{});
    if (PROBE) { if (wg_probe) wg_stamps[2] = __builtin_amdgcn_s_memtime(); }
  }
  if (MI16P) {       // no epilogue: the accumulators only have to stay alive (C is left untouched)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc4[MI16P ? i : 0][MI16P ? j : 0][r >> 2][r & 3];
    if (sum == 12345.678f) reinterpret_cast<float*>(p.C)[lane] = sum;
    return;
  }
  __syncthreads();   // the operand buffers are idle: stage the epilogue through them
  long long park_t = 0;
  w4_epilogue<EPI, PROBE>(p, acc, smem + wave * W4Epi::WAVE_BYTES, m0 + wm * 128, n0 + wn * 128, lane, split, park_t);
  if (PROBE) { if (wg_probe) { wg_stamps[3] = __builtin_amdgcn_s_memtime(); wg_stamps[6] = wall_clock64(); wg_stamps[7] = park_t; } }
}

template <int AMODE, bool PROBE, int EPI, bool MI16P = false>
int launch_w4k64_epi(GemmArgs& p, hipStream_t stream) {
  const size_t ring = 2 * (256 + 256) * 64 * 2, epi = 4 * (size_t)W4Epi::WAVE_BYTES;
  const size_t lds = ring > epi ? ring : epi;
  auto kern = gemm_bf16_w4k64_kernel<AMODE, PROBE, EPI, MI16P>;
  static G4rPerDeviceOnce attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { attr_set.failed(); return g4r_note_hip_error(e, "gemm_w4k64: hipFuncSetAttribute"); }
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, p.splits), dim3(256), lds, stream, p);
  G4R_CHECK_LAUNCH("gemm_bf16_w4k64");
  return G4R_OK;
}

// ---------------------------------------------------------------------------------------------
// Round 6 -- the PERSISTENT form of the one-wave-per-SIMD kernel ("w4k64p"): the same 256 x 256 tile, K loop and schedule as
// gemm_bf16_w4k64_kernel, but ONE workgroup per CU walks the launch's tiles (tile t of CU slot b is the t-th tile the
// hardware dispatcher would have handed that slot: linear id b + 256 t through the same XCD map), and the epilogue goes
// straight from the accumulator registers to memory.
// Why (profiles/r05_wg_timeline.txt): of a K = 4096 tile's lifetime the K loop is 0.88; the rest is the
// prologue (~3 k cycles: nothing to multiply until the first two K tiles have landed), the LDS-parked epilogue (~16 k) and
// the gap between a workgroup's exit and its successor's start (the launch of 9 waves of tiles takes ~5 us per wave longer
// than 9 workgroup lifetimes).  Here the first two K tiles of the NEXT tile are requested (LDS-DMA into the idle operand
// buffers) before the epilogue of the finished one starts, so they land while it stores; nothing is parked in LDS:
//   * in the swapped-operand accumulator layout a lane holds, of output row m, the column quads {8 q + 4 wh .. + 3}.  After
//     rounding to 16 bit a quad is two dwords; ONE v_permlane32_swap per dword between the quads q = 2 p of the upper
//     half-wave and q = 2 p + 1 of the lower one leaves every lane with 8 CONSECUTIVE columns = one 16-byte store (lane
//     wh = 0: columns 16 p .. + 7, wh = 1: 16 p + 8 .. + 15), so a store instruction writes 32 contiguous bytes of 32 rows:
//     4 096 line requests per tile against 32 768 for the 8-byte direct epilogue of round 1 and ~1 000 + two LDS passes for
//     the parked one;
//   * stores are buffer stores: a row beyond M gets an offset beyond the descriptor's extent (dropped by the hardware), so
//     the number of vector-memory instructions of an epilogue is a compile-time constant and the wait for the next tile's
//     first K tile can be COUNTED (vmcnt = 16 pieces of its second K tile + the epilogue's stores) instead of draining the
//     store queue -- vector-memory operations retire in issue order on gfx9 (loads and stores share vmcnt);
//   * SwiGLU (two levels of the swap), the fused RoPE + KV-cache append (the rotation partner d +- 64 is accumulator
//     [i][j +- 2] of the SAME lane and register: no exchange at all) and bias / activation / residual are done in the
//     accumulator layout with the arithmetic and rounding points of w4_epilogue: bit-identical results.
// Serves launches of more than one wave of tiles with N % 256 == 0, no K slices and 16-bit outputs; everything else stays on
// gemm_bf16_w4k64_kernel (launch_w4k64).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t w4p_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// two quads of 16-bit values (x = quad 2 p, y = quad 2 p + 1; two dwords each) -> this lane's 8 consecutive columns
__device__ __forceinline__ uint4v w4p_gather8(uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1) {
  const auto ra = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
  const auto rb = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
  return uint4v{ra[0], rb[0], ra[1], rb[1]};
}

constexpr int w4p_stores(int epi) { return epi == W4_SWIGLU ? 16 : 32; }     // buffer stores per wave and tile

// Epilogue of the persistent form, phase 1: accumulators -> final 16-bit values (bias / activation / residual / SwiGLU / RoPE in
// the accumulator layout, the arithmetic and rounding points of w4_epilogue), transposed 32 rows at a time through the wave's
// 8 KB of spare LDS into ROW-major 16-byte chunks: chunk (lane % LPR) of row 32 i + RPI u + lane / LPR of the wave tile (LPR = 16
// lanes per 256-byte row, RPI = 4 rows per instruction; SwiGLU: 8 and 8) goes back into the accumulator registers of pass i,
// which are dead by then -- acc[i][u / 4][4 (u % 4) .. + 3] -- so the finished tile waits for its stores in AGPRs and costs no
// VGPRs while the next tile's offsets are computed and its pieces issued.
// Every ordinary load and every LDS operation of the epilogue happens HERE, i.e. before the next tile's LDS-DMA pieces are
// issued: an LDS instruction behind a wave's own pieces waits until they have landed, and hipcc waits vmcnt(0) for an ordinary
// load next to an LDS-DMA in flight.  A parked row is 32 (SwiGLU: 16) slots of 8 B; slot s of row r sits at s ^ (r & 15): the
// 16 lanes of a ds_write_b64 group (16 rows, one slot) hit 16 different slots, the 16 lanes of a ds_read_b128 group (one row) 16
// different chunks.
template <int EPI, bool LIGHT = false>      // LIGHT (the convolutions): no bias, activation none / ReLU only -- two copies instead of eight
__device__ __forceinline__ void w4p_epilogue_compute(const GemmArgs& p, float16v (&acc)[4][4], char* wave_lds, int m_wave0, int n_wave0,
                                                     int lane) {
  const int wr = lane & 31, wh = lane >> 5;
  constexpr int OOR = (int)0x80000000;                    // beyond any extent: a load returns zeros
  constexpr int RB = EPI == W4_SWIGLU ? 128 : 256;        // parked row bytes
  constexpr int LPR = RB / 16, RPI = 64 / LPR, NU = 32 / RPI;
  const int rrow = lane / LPR, rch = lane % LPR;
  auto read_back = [&](int i) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int r = u * RPI + rrow;
      // 16-byte chunk rch of row r = slots 2 rch, 2 rch + 1 -> chunk rch ^ ((r & 15) >> 1), halves swapped when r is odd
      uint4v v = *reinterpret_cast<const uint4v*>(wave_lds + r * RB + ((rch ^ ((r & 15) >> 1)) << 4));
      if (r & 1) v = uint4v{v.z, v.w, v.x, v.y};
      float16v& dst = acc[i][u >> 2];
      dst[(u & 3) * 4 + 0] = __uint_as_float(v.x);
      dst[(u & 3) * 4 + 1] = __uint_as_float(v.y);
      dst[(u & 3) * 4 + 2] = __uint_as_float(v.z);
      dst[(u & 3) * 4 + 3] = __uint_as_float(v.w);
    }
  };
  if (EPI == W4_SWIGLU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float16v& c = acc[i][j];
        uint32_t d[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float g0 = w4_acc(c, q * 4), u0 = w4_acc(c, q * 4 + 1), g1 = w4_acc(c, q * 4 + 2), u1 = w4_acc(c, q * 4 + 3);
          const float s0 = h16lo(pack_h16x2(g0 / (1.f + __expf(-g0)), 0.f));
          const float s1 = h16lo(pack_h16x2(g1 / (1.f + __expf(-g1)), 0.f));
          d[q] = pack_h16x2(s0 * u0, s1 * u1);             // output columns 16 j + 4 q + 2 wh + {0, 1}
        }
        // quads q = 0,1 -> the 8-byte slot 4 j + wh of this row (4 consecutive outputs); q = 2,3 -> slot 4 j + 2 + wh
        const auto c0 = __builtin_amdgcn_permlane32_swap(d[0], d[1], false, false);
        const auto c1 = __builtin_amdgcn_permlane32_swap(d[2], d[3], false, false);
        *reinterpret_cast<uint2v*>(wave_lds + wr * RB + (((4 * j + wh) ^ (wr & 15)) << 3)) = uint2v{c0[0], c0[1]};
        *reinterpret_cast<uint2v*>(wave_lds + wr * RB + (((4 * j + 2 + wh) ^ (wr & 15)) << 3)) = uint2v{c1[0], c1[1]};
      }
      read_back(i);
    }
    return;
  }
  if (EPI == W4_ROPE) {
    // d = 32 jj + 8 q + 4 wh + k (< 64) is accumulator block jj, its rotation partner d + 64 block jj + 2, SAME lane and register
    const int part = n_wave0 / p.rope_HD;                // 0 q, 1 k, 2 v (wave-uniform; the wave's 128 columns are one head)
    const float inv_T = 1.0f / (float)p.rope_T;          // (m + 0.5) / T never lands within 0.5 / T of an integer: exact for M < 2^20
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (part < 2) {
        int m = m_wave0 + i * 32 + wr;
        if (m > p.M - 1) m = p.M - 1;                    // (table reads stay in range; the row is not stored)
        const int b = (int)(((float)m + 0.5f) * inv_T);
        const int tab = (p.rope_pos0 + m - b * p.rope_T) * 64 + wh * 4;
        float4v cs[8], sn[8];                            // this pass's table entries, all in flight together
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          cs[e] = *reinterpret_cast<const float4v*>(p.rope_cos + tab + e * 8);
          sn[e] = *reinterpret_cast<const float4v*>(p.rope_sin + tab + e * 8);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4v c_ = cs[jj * 4 + q], s_ = sn[jj * 4 + q];
            const float c4[4] = {c_.x, c_.y, c_.z, c_.w}, s4[4] = {s_.x, s_.y, s_.z, s_.w};
            float lo[4], hi[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              // the unfused path stored the projection in 16 bit before rotating it: keep that rounding point
              const float a = h16lo(pack_h16x2(w4_acc(acc[i][jj], q * 4 + k), 0.f));
              const float bb = h16lo(pack_h16x2(w4_acc(acc[i][jj + 2], q * 4 + k), 0.f));
              lo[k] = __builtin_fmaf(a, c4[k], -(bb * s4[k]));      // rotate_half: a' = a cos - b sin
              hi[k] = __builtin_fmaf(bb, c4[k], a * s4[k]);         //              b' = b cos + a sin
            }
            *reinterpret_cast<uint2v*>(wave_lds + wr * RB + (((8 * jj + 2 * q + wh) ^ (wr & 15)) << 3)) =
                uint2v{pack_h16x2(lo[0], lo[1]), pack_h16x2(lo[2], lo[3])};
            *reinterpret_cast<uint2v*>(wave_lds + wr * RB + (((8 * (jj + 2) + 2 * q + wh) ^ (wr & 15)) << 3)) =
                uint2v{pack_h16x2(hi[0], hi[1]), pack_h16x2(hi[2], hi[3])};
          }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float16v& c = acc[i][j];
            *reinterpret_cast<uint2v*>(wave_lds + wr * RB + (((8 * j + 2 * q + wh) ^ (wr & 15)) << 3)) =
                uint2v{pack_h16x2(w4_acc(c, q * 4), w4_acc(c, q * 4 + 1)), pack_h16x2(w4_acc(c, q * 4 + 2), w4_acc(c, q * 4 + 3))};
          }
      }
      read_back(i);
    }
    return;
  }
  // ---- W4_P16 (bias -> activation -> one rounding) and W4_WIDE (... -> + residual -> one rounding): 16-bit output ----
  const bool has_res = EPI == W4_WIDE && p.residual != nullptr;
  const __amdgpu_buffer_rsrc_t rr = w4p_rsrc(has_res ? (const void*)p.residual : (const void*)p.C, has_res ? p.r_bytes : 0u);
  const bool has_bias = p.bias != nullptr;
  auto run = [&](auto act_tag, auto bias_tag) {           // activation and bias are compile-time constants of each copy:
    constexpr int ACT = decltype(act_tag)::value;         // straight-line code (a branch per quad costs more than the quad)
    constexpr bool BIAS = decltype(bias_tag)::value;
    // The residual tile is loaded ROW-major -- 32 loads of 4 rows x 256 B per wave tile, all in flight together (128 VGPRs: the
    // operand fragments are dead) -- and transposed into the accumulator layout through the same 8 KB of LDS, one 32-row pass
    // ahead of the values it is added to.  (Loading it in the accumulator layout -- 8 B per lane on 32 different rows per
    // instruction -- took ~400 cycles per load instruction: profiles/r06_w4p_timeline.txt.)
    uint4v resrow[EPI == W4_WIDE ? 32 : 1];
    if (EPI == W4_WIDE) {
#pragma unroll
      for (int e = 0; e < 32; ++e) {
        const int m = m_wave0 + (e >> 3) * 32 + (e & 7) * 4 + rrow;
        const int roff = (m < p.M && has_res) ? (m * p.ldr + n_wave0 + rch * 8) * 2 : OOR;          // (absent / beyond M: zeros)
        resrow[e] = __builtin_amdgcn_raw_buffer_load_b128(rr, roff, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint2v res[16];                                     // this lane's residual quads of pass i (slot 8 j + 2 q + wh of row wr)
      if (EPI == W4_WIDE) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {                     // row-major chunk -> the parked-row layout (see read_back)
          const int r = u * 4 + rrow;
          uint4v v = resrow[i * 8 + u];
          if (r & 1) v = uint4v{v.z, v.w, v.x, v.y};
          *reinterpret_cast<uint4v*>(wave_lds + r * RB + ((rch ^ ((r & 15) >> 1)) << 4)) = v;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)
          res[e] = *reinterpret_cast<const uint2v*>(wave_lds + wr * RB + (((2 * e + wh) ^ (wr & 15)) << 3));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float16v& c = acc[i][j];
          // (+ 0.f: the per-tile form always adds its bias quad, zeros when absent -- keeps -0 -> +0 bit-identical)
          float v[4] = {w4_acc(c, q * 4) + 0.f, w4_acc(c, q * 4 + 1) + 0.f, w4_acc(c, q * 4 + 2) + 0.f, w4_acc(c, q * 4 + 3) + 0.f};
          if (BIAS) {
            const float4v b = *reinterpret_cast<const float4v*>(p.bias + n_wave0 + j * 32 + q * 8 + wh * 4);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float x = v[k];
            v[k] = ACT == 0 ? x : (ACT == 1 ? fmaxf(x, 0.f) : (ACT == 2 ? x / (1.f + __expf(-1.702f * x)) : x / (1.f + __expf(-x))));
          }
          if (EPI == W4_WIDE) {
            const uint2v r0 = res[j * 4 + q];
            v[0] += h16lo(r0.x); v[1] += h16hi(r0.x); v[2] += h16lo(r0.y); v[3] += h16hi(r0.y);
          }
          // columns 32 j + 8 q + 4 wh .. + 3 = slot 8 j + 2 q + wh of the row
          *reinterpret_cast<uint2v*>(wave_lds + wr * RB + (((8 * j + 2 * q + wh) ^ (wr & 15)) << 3)) =
              uint2v{pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3])};
        }
      read_back(i);
    }
  };
  if (LIGHT) {
    if (p.act == 0) run(std::integral_constant<int, 0>{}, std::false_type{});
    else run(std::integral_constant<int, 1>{}, std::false_type{});
  } else if (has_bias) {
    if (p.act == 0) run(std::integral_constant<int, 0>{}, std::true_type{});
    else if (p.act == 1) run(std::integral_constant<int, 1>{}, std::true_type{});
    else if (p.act == 2) run(std::integral_constant<int, 2>{}, std::true_type{});
    else run(std::integral_constant<int, 3>{}, std::true_type{});
  } else {
    if (p.act == 0) run(std::integral_constant<int, 0>{}, std::false_type{});
    else if (p.act == 1) run(std::integral_constant<int, 1>{}, std::false_type{});
    else if (p.act == 2) run(std::integral_constant<int, 2>{}, std::false_type{});
    else run(std::integral_constant<int, 3>{}, std::false_type{});
  }
}

// Phase 2: the 32 (SwiGLU: 16) whole-row buffer stores of a wave tile.  A row beyond M gets an offset beyond the descriptor's
// extent (dropped by the hardware), so the count is a compile-time constant.
template <int EPI>
__device__ __forceinline__ void w4p_epilogue_store(const GemmArgs& p, float16v (&acc)[4][4], int m_wave0, int n_wave0, int lane) {
  constexpr int OOR = (int)0x80000000;
  constexpr int LPR = EPI == W4_SWIGLU ? 8 : 16, RPI = 64 / LPR, NU = 32 / RPI;
  const int rrow = lane / LPR, rch = lane % LPR;
  auto chunk = [&](int i, int u) {
    const float16v& c = acc[i][u >> 2];
    return uint4v{__float_as_uint(w4_acc(c, (u & 3) * 4)), __float_as_uint(w4_acc(c, (u & 3) * 4 + 1)),
                  __float_as_uint(w4_acc(c, (u & 3) * 4 + 2)), __float_as_uint(w4_acc(c, (u & 3) * 4 + 3))};
  };
  if (EPI == W4_ROPE) {
    const int part = n_wave0 / p.rope_HD;
    const int col0 = n_wave0 - part * p.rope_HD + rch * 8;
    const __amdgpu_buffer_rsrc_t rc = part == 0 ? w4p_rsrc(p.rope_q, p.rq_bytes) : w4p_rsrc(part == 1 ? p.rope_k : p.rope_v, p.rkv_bytes);
    const float inv_T = 1.0f / (float)p.rope_T;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int m = m_wave0 + i * 32 + u * RPI + rrow;
        const int b = (int)(((float)m + 0.5f) * inv_T);
        const int pos = p.rope_pos0 + m - b * p.rope_T;
        const long o = part == 0 ? (long)m * p.rope_HD + col0 : (long)b * p.rope_kbatch + (long)pos * p.rope_krow + col0;
        __builtin_amdgcn_raw_buffer_store_b128(chunk(i, u), rc, m < p.M ? (int)(o * 2) : OOR, 0, 0);
      }
    return;
  }
  const __amdgpu_buffer_rsrc_t rc = w4p_rsrc(p.C, p.c_bytes);
  const int ncol = (EPI == W4_SWIGLU ? (n_wave0 >> 1) : n_wave0) + rch * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int m = m_wave0 + i * 32 + u * RPI + rrow;
      __builtin_amdgcn_raw_buffer_store_b128(chunk(i, u), rc, m < p.M ? (m * p.ldc + ncol) * 2 : OOR, 0, 0);
    }
}

template <int AMODE, int EPI>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4k64p_kernel(GemmArgs p) {
  constexpr int BM = 256, BN = 256, NW = 4, BKT = 64, ROWB = 128, NP = 8;
  constexpr int TM = 4, TN = 4;
  constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;      // 32 KB + 32 KB per K tile
  extern __shared__ __attribute__((aligned(16))) char smem[];            // 2 * STAGE_BYTES = 128 KB + 4 x 8 KB of epilogue staging
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int total = p.tiles_m * p.tiles_n;                                // (no K slices in this form)
  const int nt = p.K / BKT;

  // per-lane piece state.  Dense: one byte offset per piece (rows clamped at M - 1 / N - 1).  AMODE 2 (the convolution over all
  // pyramid levels, LEAN form): the launcher guarantees that every level starts on a tile boundary, so a tile lies in ONE level
  // (map size and row pitch are wave-uniform) and the rows of piece j are the rows of piece 0 + 32 j: ONE offset for A and one for
  // W (the 32 j rows go into the instruction's scalar offset), the tap masks of the 8 piece rows, and no clamping (a row beyond M
  // is beyond the descriptor's extent: zeros) -- 10 registers instead of 32, which is what lets this form fit beside the 128
  // fragment registers without moving an accumulator block to VGPRs.
  constexpr bool LEAN = AMODE == 2;
  int a_voff[LEAN ? 1 : NP], b_voff[LEAN ? 1 : NP];
  unsigned a_ok[NP];
  int a_pitch[LEAN ? 1 : NP];
  int lean_pitch = 0;                         // LEAN: bytes between map rows of the tile's level (wave-uniform)
  int m0 = 0, n0 = 0;
  // work item -> tile, and the per-lane piece offsets of that tile (piece j of this wave covers rows 8 (4 j + wave) + lane / 8;
  // lane % 8 is the 16-byte slot it writes: gemm_bf16_w4k64_kernel)
  auto setup = [&](int item) {
    const int q = total >> 3, r = total & 7, xcd = item & 7, idx = item >> 3;     // g4r_workgroup_tile_slice on a linear id
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    int tile_m, tile_n;
    g4r_tile_coords(wg, p.tiles_m, p.tiles_n, p.n_fastest, p.group_m, tile_m, tile_n);
    m0 = tile_m * BM;
    n0 = tile_n * BN;
    // (opaque copy of the lane id: without it hipcc hoists the tile-invariant parts of the offsets out of the persistent loop
    //  and keeps them alive through the K loop, where the convolution forms then spill)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    if constexpr (LEAN) {
      int lv = 0;
#pragma unroll
      for (int q2 = 1; q2 < 4; ++q2)
        if (q2 < p.n_lvl && m0 >= p.lvl_start[q2]) lv = q2;                 // (scalar: the tile's level)
      const int h = p.lvl_h[lv], w = p.lvl_w[lv], hw = h * w;
      lean_pitch = w * p.lda * 2;
      const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)w;       // (x + 0.5) / d is never within 0.5 / d of an integer: exact below 2^22
      const int row0 = wave * 8 + (ln >> 3);
      const int kslot = (ln & 7) ^ ((row0 >> 1) & 7);                       // (the same for all 8 pieces: (32 j) >> 1 is a multiple of 8)
      a_voff[0] = ((m0 + row0) * p.lda + kslot * 8) * 2;
      b_voff[0] = ((n0 + row0) * p.ldw + kslot * 8) * 2;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int gm = m0 + row0 + j * 32;
        const int local = gm - p.lvl_start[lv];
        const int img = (int)(((float)local + 0.5f) * inv_hw);
        const int rem = local - img * hw;
        const int y = (int)(((float)rem + 0.5f) * inv_w), x = rem - y * w;
        unsigned ok = 0;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
          if (yy >= 0 && yy < h && xx >= 0 && xx < w) ok |= 1u << tp;
        }
        a_ok[j] = gm < p.M ? ok : 0u;                                       // (rows beyond M: every tap reads zeros)
      }
    } else {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int row = (j * NW + wave) * 8 + (ln >> 3);
        const int kslot = (ln & 7) ^ ((row >> 1) & 7);
        int gm = m0 + row;
        if (gm > p.M - 1) gm = p.M - 1;
        a_voff[j] = (gm * p.lda + kslot * 8) * 2;
        int gn = n0 + row;
        if (gn > p.N - 1) gn = p.N - 1;
        b_voff[j] = (gn * p.ldw + kslot * 8) * 2;
        a_ok[j] = 0;
        a_pitch[j] = p.Wd * p.lda;
        if (AMODE >= 1) {
          const int h = p.H, w = p.Wd, local = gm;
          const int hw = h * w;
          const int rem = local - (local / hw) * hw;
          const int y = rem / w, x = rem - y * w;
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) a_ok[j] |= 1u << tp;
          }
        }
      }
    }
  };
  // the NEXT K tile to stage (tiles are staged strictly in order): dense = a K offset; conv = (64-channel slice, group, tap),
  // taps fastest, walked by a counter
  int st_t = 0, st_ct = 0, st_g = 0, st_tap = 0;
  struct TileSrc { int a_soff, w_soff, tap, dy, dx; };
  auto next_tile = [&]() {
    TileSrc ts;
    ts.tap = ts.dy = ts.dx = 0;
    ts.a_soff = ts.w_soff = st_t * BKT * 2;
    ++st_t;
    if (AMODE >= 1) {
      const int c0 = st_ct * BKT;
      ts.tap = st_tap;
      ts.dy = st_tap / 3 - 1;
      ts.dx = st_tap - (st_tap / 3) * 3 - 1;
      ts.a_soff = (int)(((long)st_g * p.a_group_stride + c0) * 2);
      ts.w_soff = ((st_g * 9 + st_tap) * p.Cin + c0) * 2;
      if (++st_tap == 9) {
        st_tap = 0;
        if (++st_g == p.groups) { st_g = 0; ++st_ct; }
      }
    }
    return ts;
  };
  auto piece = [&](int q, const TileSrc& ts, int buf) {
    char* sa = smem + buf * STAGE_BYTES;
    if constexpr (LEAN) {
      if (q < NP) {
        // the tap shift and the 32 q rows of piece q are wave-uniform (one level per tile): ONE scalar added to the lane's base
        // offset.  (They cannot ride in the instruction's scalar offset: the descriptor's range check sees the VECTOR offset alone,
        // and base + a negative tap shift is "out of range" before the scalar offset brings it back -- zeros instead of texels.)
        int voff = a_voff[0] + (q * 32 * p.lda * 2 + ts.dy * lean_pitch + ts.dx * p.lda * 2);
        if (!((a_ok[q] >> ts.tap) & 1u)) voff = (int)0x80000000;            // beyond num_records: reads as zeros
        g4r_buffer_piece(p.A, p.a_bytes, sa + (q * NW + wave) * 1024, voff, ts.a_soff);
      } else {
        g4r_buffer_piece(p.W, p.w_bytes, sa + A_BYTES + ((q - NP) * NW + wave) * 1024, b_voff[0], ts.w_soff + (q - NP) * 32 * p.ldw * 2);
      }
      return;
    }
    if (q < NP) {
      int voff = a_voff[q];
      if (AMODE == 1) voff += (ts.dy * p.Wd + ts.dx) * p.lda * 2;
      if (AMODE == 2) voff += (ts.dy * a_pitch[q] + ts.dx * p.lda) * 2;
      if (AMODE >= 1 && !((a_ok[q] >> ts.tap) & 1u)) voff = (int)0x80000000;      // beyond num_records: reads as zeros
      g4r_buffer_piece(p.A, p.a_bytes, sa + (q * NW + wave) * 1024, voff, ts.a_soff);
    } else {
      g4r_buffer_piece(p.W, p.w_bytes, sa + A_BYTES + ((q - NP) * NW + wave) * 1024, b_voff[q - NP], ts.w_soff);
    }
  };
  // the first two K tiles of the tile `setup` prepared: 16 + 16 pieces into buffers 0 / 1
  auto prologue = [&]() {
    st_t = st_ct = st_g = st_tap = 0;
    {
      const TileSrc ts = next_tile();
#pragma unroll
      for (int q = 0; q < 2 * NP; ++q) piece(q, ts, 0);
    }
    if (nt > 1) {
      const TileSrc ts = next_tile();
#pragma unroll
      for (int q = 0; q < 2 * NP; ++q) piece(q, ts, 1);
    }
  };
  float16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fsw = (frow >> 1) & 7, fhi = lane >> 5;
  const int a_row_off = (wm * 128 + frow) * ROWB;
  const int b_row_off = A_BYTES + (wn * 128 + frow) * ROWB;
  h16x8 fa[4][TM], fw[4][TN];                  // [k-step][block]: k-steps 0,1 = H0, 2,3 = H1
  auto ldfrag = [&](auto ks_tag, int buf) {
    constexpr int ks = decltype(ks_tag)::value;
    const char* sb = smem + buf * STAGE_BYTES;
    const int slot = ((2 * ks + fhi) ^ fsw) << 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const h16x8*>(sb + a_row_off + i * 32 * ROWB + slot);
#pragma unroll
    for (int j = 0; j < TN; ++j) fw[ks][j] = *reinterpret_cast<const h16x8*>(sb + b_row_off + j * 32 * ROWB + slot);
  };
  auto mma_rows = [&](auto ks_tag, auto i0_tag, auto i1_tag, auto zero_tag) {    // rows [i0, i1) of k-step ks
    constexpr int ks = decltype(ks_tag)::value, i0 = decltype(i0_tag)::value, i1 = decltype(i1_tag)::value;
    constexpr bool ZERO = decltype(zero_tag)::value;              // the first products of an output tile: C = 0 (an inline constant)
    const float16v z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = i0; i < i1; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = G4R_MFMA_32X32X16(fw[ks][j], fa[ks][i], ZERO ? z16 : acc[i][j], 0, 0, 0);
  };
  using NoZ = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  // the K-tile body of gemm_bf16_w4k64_kernel (phases A / B / C, see there), unchanged
  // RELAX: the epilogue's stores of the previous output tile sit between K tile 1's pieces and K tile 2's in the (in-order)
  // vector-memory queue; the wait for K tile 1 at the end of phase B of K tile 0 may leave them in flight (RELAX = their
  // number), which gives them two K tiles (~4.4 k cycles) to drain behind the matrix pipe instead of one phase
  auto body = [&](int i, auto steady_tag, auto relax_tag) {
    constexpr bool STEADY = decltype(steady_tag)::value;      // tiles i+1, i+2 exist: no branches in the body
    constexpr int RELAX = decltype(relax_tag)::value;
    const int buf = i & 1;
    ldfrag(I2{}, buf);
    ldfrag(I3{}, buf);
    mma_rows(I0{}, I0{}, I4{}, NoZ{});      // (C = 0 for the first K tile after an epilogue instead of zeroing the accumulators
                                            //  was tried: hipcc then spills 50-200 VGPRs in every epilogue mode)
    mma_rows(I1{}, I0{}, I1{}, NoZ{});
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): every fragment of X is in registers
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                            // ... in every wave: X may be overwritten
    __builtin_amdgcn_sched_barrier(0);
    TileSrc ts2 = {};
    const bool more = STEADY || i + 2 < nt;
    if (more) {
      ts2 = next_tile();
#pragma unroll
      for (int q = 0; q < 12; ++q) piece(q, ts2, buf);
    }
    mma_rows(I1{}, I1{}, I4{}, NoZ{});
    mma_rows(I2{}, I0{}, I3{}, NoZ{});
    if (STEADY) {
#pragma unroll
      for (int n = 0; n < 12; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x8, 2, 1);
        __builtin_amdgcn_sched_group_barrier(0x10, 1, 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      if (RELAX == 32) __builtin_amdgcn_s_waitcnt(0x8f7c);   // vmcnt(44): tile i+1 has landed; 32 stores + this tile's 12 pieces may be in flight
      else if (RELAX == 16) __builtin_amdgcn_s_waitcnt(0x4f7c);   // vmcnt(28)
      else __builtin_amdgcn_s_waitcnt(0x0f7c);               // vmcnt(12): tile i+1 has landed (and every older store has retired)
    } else {
      __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0)
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (STEADY || i + 1 < nt) {
      ldfrag(I0{}, buf ^ 1);
      ldfrag(I1{}, buf ^ 1);
    }
    if (more) {
#pragma unroll
      for (int q = 12; q < 16; ++q) piece(q, ts2, buf);
    }
    mma_rows(I2{}, I3{}, I4{}, NoZ{});
    mma_rows(I3{}, I0{}, I4{}, NoZ{});
    if (STEADY) {
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 2);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 2);
        __builtin_amdgcn_sched_group_barrier(0x10, 1, 2);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // Order at a tile boundary: epilogue phase 1 (every load and LDS operation of the epilogue; results in registers) -> the next
  // tile's offsets and its first two K tiles' pieces -> epilogue phase 2 (the stores).  The pieces land while the stores issue,
  // and the wait for them below leaves the stores in flight.  (The epilogue sits at the loop's END: with it at the head hipcc
  // moves all 256 accumulators to VGPRs and spills.)
  int item = blockIdx.x;
  setup(item);
  prologue();
  // vector-memory operations issued AFTER the first K tile's 16 pieces when the loop is entered: the second K tile's 16
  // pieces and, when the epilogue ran behind the pieces, its stores (a compile-time count: a row beyond M is a dropped
  // store, not a skipped one).  They retire in issue order (loads and stores share vmcnt on gfx9), so at most that many
  // outstanding = the first K tile has landed.
  int behind = nt > 1 ? 16 : 0;
#ifdef G4R_W4P_PROBE
  // tools/w4p_timeline.py (build with G4R_EXTRA_HIPCC_FLAGS=-DG4R_W4P_PROBE, pass a workspace): wave 0 of every workgroup records
  // s_memtime at [tile top, K loop begin, K loop end, epilogue end] of each of its tiles at ws + (blockIdx.x * 16 + tile) * 4
  long long* stamps = reinterpret_cast<long long*>(p.ws) + (long)blockIdx.x * 64;
  int tile_no = 0;
#define G4R_W4P_STAMP(slot) do { if (p.ws != nullptr && wave == 0 && tile_no < 16) stamps[tile_no * 4 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G4R_W4P_STAMP(slot) do { } while (0)
#endif
  for (;;) {
    G4R_W4P_STAMP(0);
    if (behind >= 48) __builtin_amdgcn_s_waitcnt(0xcf70);        // vmcnt(48)
    else if (behind >= 32) __builtin_amdgcn_s_waitcnt(0x8f70);   // vmcnt(32)
    else if (behind >= 16) __builtin_amdgcn_s_waitcnt(0x4f70);   // vmcnt(16)
    else __builtin_amdgcn_s_waitcnt(0x0f70);                     // vmcnt(0)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ldfrag(I0{}, 0);
    ldfrag(I1{}, 0);
    G4R_W4P_STAMP(1);
    int i = 0;
    if (behind > 16 && nt > 2) {                                 // (behind > 16: an epilogue's stores are in the queue)
      body(0, std::true_type{}, std::integral_constant<int, w4p_stores(EPI)>{});
      i = 1;
    }
    for (; i + 2 < nt; ++i) body(i, std::true_type{}, I0{});
    for (; i < nt; ++i) body(i, std::false_type{}, I0{});
    // every wave has read its last fragments (phase A of the last K tile ends in a barrier) and no piece is in flight: the
    // operand buffers are idle
    G4R_W4P_STAMP(2);
    const int em = m0 + wm * 128, en = n0 + wn * 128;
    const int next = item + (int)gridDim.x;
    const bool has_next = next < total;
    w4p_epilogue_compute<EPI, (AMODE != 0)>(p, acc, smem + 2 * STAGE_BYTES + wave * 8192, em, en, lane);
    if (has_next) {
      setup(next);
      prologue();
    }
    w4p_epilogue_store<EPI>(p, acc, em, en, lane);
    G4R_W4P_STAMP(3);
#ifdef G4R_W4P_PROBE
    ++tile_no;
#endif
    if (!has_next) break;
#pragma unroll
    for (int i2 = 0; i2 < TM; ++i2)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i2][j][r] = 0.f;
    behind = (nt > 1 ? 16 : 0) + w4p_stores(EPI);
    item = next;
  }
}

template <int AMODE, int EPI>
int launch_w4k64p_epi(GemmArgs& p, int grid, hipStream_t stream) {
  const size_t lds = 2 * (256 + 256) * 64 * 2 + 4 * 8192;        // the operand ring + 8 KB of epilogue staging per wave = all 160 KB
  auto kern = gemm_bf16_w4k64p_kernel<AMODE, EPI>;
  static G4rPerDeviceOnce attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { attr_set.failed(); return g4r_note_hip_error(e, "gemm_w4k64p: hipFuncSetAttribute"); }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
  G4R_CHECK_LAUNCH("gemm_bf16_w4k64p");
  return G4R_OK;
}

// workgroups of the persistent form = CUs of the device (one workgroup of 256 AGPRs + 128 KB LDS per CU)
static int g4r_cu_count() {
  static int n[64] = {};
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return 256;
  if (n[d] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
    n[d] = v;
  }
  return n[d];
}

template <int AMODE, bool PROBE = false, int BM = 256, int BN = 256, bool BUF = true, int SCHED = 0>
int launch_pp32(GemmArgs& p, hipStream_t stream);      // (defined below: the fallback for operands beyond a buffer descriptor)

template <int AMODE, bool PROBE = false>
int launch_w4k64(GemmArgs& p, hipStream_t stream) {
  {
    // extents for the buffer descriptors: the last byte a clamped row / in-image tap can touch
    size_t ab = (size_t)p.M * p.lda * 2, wb = (size_t)p.N * p.ldw * 2;
    if (AMODE == 1) ab = ((size_t)(p.groups - 1) * p.a_group_stride + (size_t)p.M * p.lda) * 2;
    // operands of 2 GiB and more do not fit a buffer descriptor's 32-bit extent: the ring ping-pong tile with plain
    // global_load_lds pieces (64-bit addresses) serves them, as it did before this kernel became the default (e.g. the
    // stage-2 data gradient dgu[tokens, 22016] x W^T above ~48.7 k tokens)
    if (ab >= 0x7fffffffu || wb >= 0x7fffffffu) {
      if (PROBE) return g4r_note_error(G4R_ERR_UNSUPPORTED, "gemm_w4k64 (probe): operands of 2 GiB and more");
      return launch_pp32<AMODE, false, 256, 256, false, (AMODE >= 1 ? 1 : 0)>(p, stream);
    }
    p.a_bytes = (unsigned)ab;
    p.w_bytes = (unsigned)wb;
  }
  if (p.K % 64 != 0 || (AMODE >= 1 && p.Cin % 64 != 0)) return g4r_note_error(G4R_ERR_INVALID_ARG, "gemm_w4k64: K (Cin) must be a multiple of 64");
  {
    const int nt = p.K / 64;
    int splits = p.splits < 1 ? 1 : p.splits;
    if (splits > nt) splits = nt;
    p.tiles_per_split = g4r_ceil_div(nt, splits);
    p.splits = g4r_ceil_div(nt, p.tiles_per_split);
  }
  p.tiles_m = g4r_ceil_div(p.M, 256);
  p.tiles_n = g4r_ceil_div(p.N, 256);
  // start de-phasing of launches of several waves of tiles (tools: debug modes 70 + n = n x 0.25 us per phase, 79 = off)
  p.stagger_ticks = 0;
  if ((long)p.tiles_m * p.tiles_n * p.splits > 2 * 256) p.stagger_ticks = G4R_W4_STAGGER_TICKS;
  if (g_gemm_dbg >= 70 && g_gemm_dbg < 79) p.stagger_ticks = (g_gemm_dbg - 70) * 25;
  if (g_gemm_dbg == 79) p.stagger_ticks = 0;
  // the epilogue mode (W4Epi): the straight-line forms need 16-byte rows on every operand they touch
  const bool wide = p.splits > 1 || p.out_f32 || p.residual != nullptr;
  int mode = W4_GENERIC;
  if ((p.N & 7) == 0) {
    if (p.act == 5) mode = W4_ROPE;
    else if (p.splits > 1) mode = W4_WIDE;
    else if ((p.ldc & 7) == 0 && (p.residual == nullptr || (p.ldr & 7) == 0) && (p.act != 4 || ((p.ldc & 7) == 0 && !wide)))
      mode = wide ? W4_WIDE : (p.act == 4 ? W4_SWIGLU : W4_P16);
  }
  if (p.act == 5 && mode != W4_ROPE) return g4r_note_error(G4R_ERR_INVALID_ARG, "gemm_w4k64: the fused RoPE epilogue needs N % 8 == 0");
  // Round 6: launches of more than one wave of tiles take the PERSISTENT form (gemm_bf16_w4k64p_kernel: one workgroup per CU
  // walks the tiles, the next tile's first K tiles land during the epilogue, which stores straight from the accumulators).
  // tools: debug mode 61 = off (A/B arm; the bit-identity tests compare the two forms).
  if (!PROBE && g_gemm_dbg != 61 && p.splits == 1 && (p.N % 256) == 0 && !p.out_f32 && (p.ldc & 7) == 0) {
    const int ncu = g4r_cu_count();
    const long tiles = (long)p.tiles_m * p.tiles_n;
    const bool mode_ok = mode == W4_P16 || mode == W4_SWIGLU || mode == W4_ROPE || (mode == W4_WIDE && p.residual != nullptr);
    size_t cb = (size_t)p.M * p.ldc * 2, rb = p.residual ? (size_t)p.M * p.ldr * 2 : 0, rq = 0, rkv = 0;
    if (mode == W4_ROPE) {
      const long nb = p.M / p.rope_T;
      rq = (size_t)p.M * p.rope_HD * 2;
      rkv = (size_t)((nb - 1) * p.rope_kbatch + (long)(p.rope_pos0 + p.rope_T - 1) * p.rope_krow + p.rope_HD) * 2;
      cb = rq;
    }
    // (K < 2048 -- the ViT block GEMMs -- stays per tile: with 16 K tiles per output tile the epilogue is a fifth of a tile and the
    //  per-tile form hides its store drain behind the next workgroup's start: 6-9 % faster there, profiles/r06_persist_ab.jsonl)
    if (mode_ok && tiles > ncu && (p.K >= 2048 || p.persist_any_k) && p.M < (1 << 20) && cb < 0x7fffffffu && rb < 0x7fffffffu && rq < 0x7fffffffu && rkv < 0x7fffffffu) {
      p.c_bytes = (unsigned)cb; p.r_bytes = (unsigned)rb; p.rq_bytes = (unsigned)rq; p.rkv_bytes = (unsigned)rkv;
      if (AMODE != 0) {
        // The convolution over all pyramid levels has a persistent form too (LEAN addressing, see the kernel: every level starts on a
        // tile boundary -- true for the pyramids of the path at any batch: 64 P^2, 16 P^2 and 4 P^2 rows per image are multiples of
        // 256 at P = 16 and 24).  It is bit-identical to the per-tile form and 4-6 % SLOWER (1282 vs 1358 TF/s at batch 16,
        // profiles/r06_conv_persist_ab.txt: with 144 K tiles per output tile the boundary is 5 % of a tile and hipcc spills around it),
        // so it is compiled into the tools build only (debug mode 63, tools/conv_persist_ab.py).
#ifdef G4R_TOOLS_BUILD
        if constexpr (AMODE == 2) {
          bool aligned = p.groups == 1;
          for (int l = 1; l < p.n_lvl; ++l) aligned = aligned && (p.lvl_start[l] % 256) == 0;
          if (g_gemm_dbg == 63 && aligned && mode == W4_P16 && p.bias == nullptr && p.act <= 1) return launch_w4k64p_epi<2, W4_P16>(p, ncu, stream);
        }
#endif
      } else {
        switch (mode) {
          case W4_P16: return launch_w4k64p_epi<0, W4_P16>(p, ncu, stream);
          case W4_SWIGLU: return launch_w4k64p_epi<0, W4_SWIGLU>(p, ncu, stream);
          case W4_ROPE: return launch_w4k64p_epi<0, W4_ROPE>(p, ncu, stream);
          default: return launch_w4k64p_epi<0, W4_WIDE>(p, ncu, stream);
        }
      }
    }
  }
#ifdef G4R_TOOLS_BUILD
  if (AMODE == 0 && !PROBE && g_gemm_dbg == 65 && mode == W4_P16) return launch_w4k64_epi<0, false, W4_P16, true>(p, stream);   // MFMA-shape probe
#endif
  int rc = G4R_OK;
  if (AMODE != 0) {                       // the convolutions: bias / ReLU only
    rc = (mode == W4_P16) ? launch_w4k64_epi<AMODE, false, W4_P16>(p, stream)
                          : (mode == W4_WIDE ? launch_w4k64_epi<AMODE, false, W4_WIDE>(p, stream) : launch_w4k64_epi<AMODE, false, W4_GENERIC>(p, stream));
  } else if (PROBE) {
    rc = mode == W4_P16 ? launch_w4k64_epi<0, PROBE, W4_P16>(p, stream) : launch_w4k64_epi<0, PROBE, W4_WIDE>(p, stream);
  } else {
    switch (mode) {
      case W4_P16: rc = launch_w4k64_epi<0, false, W4_P16>(p, stream); break;
      case W4_SWIGLU: rc = launch_w4k64_epi<0, false, W4_SWIGLU>(p, stream); break;
      case W4_ROPE: rc = launch_w4k64_epi<0, false, W4_ROPE>(p, stream); break;
      case W4_WIDE: rc = launch_w4k64_epi<0, false, W4_WIDE>(p, stream); break;
      default: rc = launch_w4k64_epi<0, false, W4_GENERIC>(p, stream); break;
    }
  }
  if (rc != G4R_OK) return rc;
  if (p.splits > 1 && !p.defer_reduce) {
    long total = (long)p.M * p.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    G4R_CHECK_LAUNCH("splitk_reduce");
  }
  return G4R_OK;
}

template <int AMODE, bool PROBE, int BM, int BN, bool BUF, int SCHED>
int launch_pp32(GemmArgs& p, hipStream_t stream) {
  if (BUF) {
    // extents for the buffer descriptors: the last byte a clamped row / in-image tap can touch
    size_t ab = (size_t)p.M * p.lda * 2, wb = (size_t)p.N * p.ldw * 2;
    if (AMODE == 1) ab = ((size_t)(p.groups - 1) * p.a_group_stride + (size_t)p.M * p.lda) * 2;
    if (ab >= 0x7fffffffu || wb >= 0x7fffffffu || g_gemm_dbg == 8) return launch_pp32<AMODE, PROBE, BM, BN, false, SCHED>(p, stream);
    p.a_bytes = (unsigned)ab;
    p.w_bytes = (unsigned)wb;
  }
  {
    const int nt = p.K / 32;
    int splits = p.splits < 1 ? 1 : p.splits;
    if (splits > nt) splits = nt;
    p.tiles_per_split = g4r_ceil_div(nt, splits);
    p.splits = g4r_ceil_div(nt, p.tiles_per_split);
  }
  p.tiles_m = g4r_ceil_div(p.M, BM);
  p.tiles_n = g4r_ceil_div(p.N, BN);
  constexpr int TN = BN / 4 / 32, TM = BM / 2 / 32;
  const size_t ring = 4 * (BM + BN) * 32 * 2, epi = 8 * (size_t)EpiLds<TN, ((TN == 2 && TM % 2 == 0) ? 64 : 32)>::WAVE_BYTES;
  const size_t lds = ring > epi ? ring : epi;
  auto kern = gemm_bf16_pp32_kernel<AMODE, PROBE, BM, BN, BUF, SCHED>;
  static G4rPerDeviceOnce attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { attr_set.failed(); return g4r_note_hip_error(e, "gemm_pp32: hipFuncSetAttribute"); }
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, p.splits), dim3(512), lds, stream, p);
  G4R_CHECK_LAUNCH("gemm_bf16_pp32");
  if (p.splits > 1 && !p.defer_reduce) {
    long total = (long)p.M * p.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    G4R_CHECK_LAUNCH("splitk_reduce");
  }
  return G4R_OK;
}

template <int AMODE, bool PROBE = false>
int launch_pp(GemmArgs& p, hipStream_t stream) {
  {
    const int nt = p.K / 64;
    int splits = p.splits < 1 ? 1 : p.splits;
    if (splits > nt) splits = nt;
    p.tiles_per_split = g4r_ceil_div(nt, splits);
    p.splits = g4r_ceil_div(nt, p.tiles_per_split);
  }
  p.tiles_m = g4r_ceil_div(p.M, 256);
  p.tiles_n = g4r_ceil_div(p.N, 256);
  const size_t ring = 2 * (256 + 256) * 64 * 2, epi = 8 * (size_t)EpiLds<2>::WAVE_BYTES;
  const size_t lds = ring > epi ? ring : epi;
  auto kern = gemm_bf16_pp_kernel<AMODE, PROBE>;
  static G4rPerDeviceOnce attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { attr_set.failed(); return g4r_note_hip_error(e, "gemm_pp: hipFuncSetAttribute"); }
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, p.splits), dim3(512), lds, stream, p);
  G4R_CHECK_LAUNCH("gemm_bf16_pp");
  if (p.splits > 1 && !p.defer_reduce) {
    long total = (long)p.M * p.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    G4R_CHECK_LAUNCH("splitk_reduce");
  }
  return G4R_OK;
}

template <int BM, int BN, int WM, int WN, int AMODE, bool GLDS, int STAGES = 2, int BKT = 64, int STYLE = 0>
int launch_tile(GemmArgs& p, hipStream_t stream) {
  {  // split-K geometry in units of this kernel's K tile
    const int nt = p.K / BKT;
    int splits = p.splits < 1 ? 1 : p.splits;
    if (splits > nt) splits = nt;
    p.tiles_per_split = g4r_ceil_div(nt, splits);
    p.splits = g4r_ceil_div(nt, p.tiles_per_split);
  }
  p.tiles_m = g4r_ceil_div(p.M, BM);
  p.tiles_n = g4r_ceil_div(p.N, BN);
  const size_t ring = (size_t)STAGES * (BM + BN) * BKT * 2, epi = (size_t)WM * WN * EpiLds<BN / WN / 32, 32>::WAVE_BYTES;
  const size_t lds = ring > epi ? ring : epi;
  auto kern = gemm_bf16_nt_kernel<BM, BN, WM, WN, AMODE, GLDS, STAGES, BKT, STYLE>;
  static G4rPerDeviceOnce attr_set;
  if (attr_set.first()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { attr_set.failed(); return g4r_note_hip_error(e, "gemm: hipFuncSetAttribute"); }
  }
  dim3 grid(p.tiles_m * p.tiles_n, p.splits);
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, stream, p);
  G4R_CHECK_LAUNCH("gemm_bf16_nt");
  if (p.splits > 1 && !p.defer_reduce) {
    long total = (long)p.M * p.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    G4R_CHECK_LAUNCH("splitk_reduce");
  }
  return G4R_OK;
}

template <int AMODE>
int launch_gemm(GemmArgs& p, int tile_cfg, hipStream_t stream) {
  if (g_gemm_dbg == 9) p.n_fastest = 0;     // tools: force the tile order (A/B of the L2 sharing pattern)
  if (g_gemm_dbg == 10) p.n_fastest = 1;
  // grouped tile order for dense launches with many row tiles (ring ping-pong kernel): tools modes 31 / 32 / 33 force a group of
  // 8 / 4 / 16 row tiles, 30 forces the plain order
  p.group_m = 0;
  if (AMODE == 0 && (tile_cfg == 24 || tile_cfg == 28 || tile_cfg == 34 || tile_cfg == 35 || tile_cfg == 36)) {
    const int tm = g4r_ceil_div(p.M, tile_cfg == 28 ? 192 : 256);
    if (tm >= 12) p.group_m = 8;
    if (g_gemm_dbg == 30) p.group_m = 0;
    if (g_gemm_dbg == 31) p.group_m = 8;
    if (g_gemm_dbg == 32) p.group_m = 4;
    if (g_gemm_dbg == 33) p.group_m = 16;
  }
  if (AMODE >= 1 && g_gemm_dbg == 41) p.group_m = -2;   // tools: conv, an XCD's wave = 16 pixel tiles x 2 weight panels
  if (AMODE >= 1 && g_gemm_dbg == 42) p.group_m = -1;   // tools: 32 pixel tiles x 1 weight panel
  switch (tile_cfg) {
    // ---- the tiles kernels.py dispatches (pick_tile / pick_conv_tile / long_k_plan / partial_wave_plan / layers.py) ----
    case 0: return launch_tile<128, 128, 2, 2, AMODE, true>(p, stream);
    case 4: return launch_tile<64, 128, 1, 4, AMODE, true>(p, stream);
    case 7: return launch_tile<128, 128, 2, 4, AMODE, true, 4, 64, 3>(p, stream);   // 8 waves, 64x32 wave tiles; counted-wait fragment pipeline
    // Small-M shapes (CLIP ViT, M = 577; K = 1024 is only 16 K tiles): fewer workgroups than CUs, so what matters is ONE
    // workgroup's latency -> deep LDS-DMA rings instead of co-resident workgroups.
    case 13: return launch_tile<64, 128, 1, 4, AMODE, true, 3, 64, 3>(p, stream);   // 72 KB ring of 3: 2 wg/CU; counted-wait fragment pipeline
    case 14: return launch_tile<64, 64, 2, 2, AMODE, true, 4, 64, 3>(p, stream);    // 64 KB ring of 4: 2 wg/CU; counted-wait fragment pipeline
    case 34: return launch_w4k64<AMODE>(p, stream);                              // 256x256, 4 waves x (128x128), K 64 x 2 buffers refilled as consumed (round 5)
    case 36: p.persist_any_k = 1; return launch_w4k64<AMODE>(p, stream);         // tile 34 with its persistent form offered at every K (round 6; tests)
    case 24:                                                                     // 256x256 ping-pong, K 32 ring of 4 (G4R_BIG_TILE=24; operands >= 2 GiB)
      // the implicit-GEMM convs take the rotated single-barrier schedule (192^2 conv 693 -> 668 us, tools/gemm_bench.cpp
      // tile 31); the dense GEMMs lose 8-12 % on it (4096^3 1172 -> 1076 TF/s) and keep the two-barrier form
      if constexpr (AMODE >= 1) return launch_pp32<AMODE, false, 256, 256, true, 1>(p, stream);
      else return launch_pp32<AMODE>(p, stream);
    case 28: return launch_pp32<AMODE, false, 192, 256>(p, stream);              // 192x256 ring ping-pong (767 x 12288: 4 x 48 = 192 workgroups)
#ifdef G4R_TOOLS_BUILD
    // ---- superseded forms and A/B arms: compiled only into the tools build (G4R_EXTRA_HIPCC_FLAGS=-DG4R_TOOLS_BUILD python -m
    //      gpt4roi_amd.build --force); the shipped library holds the dispatched kernels only ----
    case 1: return launch_tile<256, 128, 4, 2, AMODE, true>(p, stream);
    case 2: return launch_tile<128, 128, 2, 2, AMODE, false>(p, stream);  // register-staged A/B probe
    case 5: return launch_tile<128, 128, 2, 2, AMODE, true, 3>(p, stream);   // 96 KB ring
    case 6: return launch_tile<128, 128, 2, 2, AMODE, true, 4>(p, stream);   // 128 KB ring
    case 8: return launch_tile<256, 128, 4, 2, AMODE, true, 3>(p, stream);   // 144 KB ring, 8 waves
    case 9: return launch_tile<256, 256, 2, 4, AMODE, true, 2>(p, stream);   // 128 KB, wave tile 128x64
    case 10: return launch_tile<128, 128, 2, 4, AMODE, true, 2>(p, stream);  // 8 waves x (64x32), 2 wg/CU
    case 11: return launch_tile<128, 64, 2, 2, AMODE, true, 2>(p, stream);   // 48 KB: 3 wg/CU
    // (12-21 of round 1 were the BK = 32 / interleaved-read ring experiments of DESIGN.md section 3; they lost and
    // were removed.)
    case 12: return launch_tile<64, 128, 1, 4, AMODE, true, 4>(p, stream);   // 96 KB ring of 4
    case 15: return launch_tile<128, 64, 2, 2, AMODE, true, 4>(p, stream);   // 96 KB ring of 4
    // A/B arms of round 3c (dense GEMM only): tiles 13 / 14 / 7 with the compiler's own read order (STYLE 0; the counted-wait
    // fragment pipeline, STYLE 3, is their production form: ViT qkv 10.9 -> 10.8 us, o 8.2 -> 7.9, fc1 15.5 -> 14.6, fc2 21.1 ->
    // 20.4, LLaMA o_proj 41.7 -> 40.2), and the two-stage 128 x 128 tile WITH the pipeline (no gain: 68.3 vs 69.6 us)
    case 40: if constexpr (AMODE == 0) return launch_tile<128, 128, 2, 2, AMODE, true, 2, 64, 3>(p, stream); else break;
    case 43: if constexpr (AMODE == 0) return launch_tile<64, 128, 1, 4, AMODE, true, 3>(p, stream); else break;
    case 44: if constexpr (AMODE == 0) return launch_tile<64, 64, 2, 2, AMODE, true, 4>(p, stream); else break;
    case 47: if constexpr (AMODE == 0) return launch_tile<128, 128, 2, 4, AMODE, true, 4>(p, stream); else break;
    case 35: if constexpr (AMODE == 0) return launch_w4k64<AMODE, true>(p, stream); else break;   // tile 34 + s_memtime stamps
    case 26: return launch_w4<AMODE>(p, stream);                                 // 256x256, 4 waves x (128x128): one wave per SIMD, K 32 ring of 4
    case 22: return launch_pp<AMODE>(p, stream);                                 // 256x256 ping-pong (4 barriers / K tile)
    case 27: return launch_pp32<AMODE, false, 128, 384>(p, stream);              // 128x384 ring ping-pong (767 x 12288: 192 workgroups)
    case 25: return launch_pp32<AMODE, true>(p, stream);                         // tile 24 + s_memtime stamps
    case 30: return launch_pp32<AMODE, false, 256, 256, false>(p, stream);       // A/B arm: pieces by global_load_lds (the round-2 form)
    case 31: return launch_pp32<AMODE, false, 256, 256, true, 1>(p, stream);     // rotated single-barrier schedule (A/B arm for dense)
    case 33: return launch_pp32<AMODE, false, 256, 256, true, 0>(p, stream);     // two-barrier schedule (A/B arm for the convs)
    case 23: return launch_pp<AMODE, true>(p, stream);                           // tile 22 + s_memtime stamps into ws
#endif
    default: break;
  }
  return g4r_note_error(G4R_ERR_INVALID_ARG, "gemm: unknown tile_cfg (superseded forms / A-B arms exist in the tools build only: -DG4R_TOOLS_BUILD)");
}

}  // namespace


extern "C" {

// tools/ only: ablation switch for the GEMM main loop (0 = normal).  Not declared in include/.
//   1 / 2: skip the staging / the compute of the generic kernel's loop;  7: taps-outermost K order of the conv (A/B arm);
//   8: pieces by global_load_lds instead of buffer loads;  9 / 10: force the M-fastest / N-fastest tile order;
//   11: the round-2 workgroup -> tile map (tiles only, every XCD sees all K slices);  100 + v: GEMV variant v.
#ifndef G4R_F16
void g4r_gemm_debug_mode(int mode) { g_gemm_dbg = mode; }
#endif

// See include/g4r_kernels.h for the contract.
// C = A W^T as `*splits_out` fp32 K-slice partials [slice][M][N] in `workspace` (>= splits * M * N floats), WITHOUT the reduce
// launch: the consumer combines them (g4r_rmsnorm_splitk_bf16: the reduce of the LLaMA down_proj folded into the next
// RMSNorm).  splits >= 2; the number of slices actually written (<= splits) is returned through splits_out.
int g4r_gemm_bf16_nt_partials(const void* A, const void* W, float* workspace, int M, int N, int K, int lda, int ldw,
                              int splits, int tile_cfg, int* splits_out, void* stream) {
  G4R_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0, "gemm_partials: K must be a positive multiple of 64");
  G4R_REQUIRE(A && W && workspace && splits_out, "gemm_partials: null pointer");
  G4R_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0 && splits >= 2, "gemm_partials: 16-byte rows, splits >= 2");
  {
    // the slices the tile will really cut (its K tile is 32 for the ring ping-pong / one-wave-per-SIMD tiles, 64 otherwise),
    // checked BEFORE the launch: a tile that clamps to one slice would run its normal epilogue into `workspace`
    const bool k32 = tile_cfg == 24 || tile_cfg == 25 || tile_cfg == 26 || tile_cfg == 27 || tile_cfg == 28 || tile_cfg == 30 ||
                     tile_cfg == 31 || tile_cfg == 33;
    const int nt = K / (k32 ? 32 : 64);
    const int s = splits > nt ? nt : splits;
    G4R_REQUIRE(s >= 1 && g4r_ceil_div(nt, g4r_ceil_div(nt, s)) >= 2,
                "gemm_partials: the tile's K slices collapse to one (K too short for `splits`)");
  }
  GemmArgs p = {};
  p.A = (const h16_t*)A; p.W = (const h16_t*)W; p.C = workspace; p.ws = workspace;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = N; p.ldr = 0;
  p.act = 0; p.out_f32 = 1; p.dbg = g_gemm_dbg;
  p.n_fastest = (long)M * 1 > (long)N * 2;
  p.splits = splits;
  p.defer_reduce = 1;
  const int rc = launch_gemm<0>(p, tile_cfg, (hipStream_t)stream);
  *splits_out = p.splits;
  G4R_REQUIRE(rc != G4R_OK || p.splits >= 2, "gemm_partials: the tile's K slices collapsed to one (K too short for `splits`)");
  return rc;
}

// The fused q|k|v projection of a LLaMA layer with RoPE and the KV-cache append in the GEMM epilogue: A [B*T, K] hidden rows,
// W [3*heads*128, K] (q | k | v rows).  q_out [B*T, heads*128] receives the rotated queries; k_cache / v_cache (this layer's
// cache, rows `cache_row` elements apart, sequences `cache_batch` apart) receive the rotated keys / the values at rows
// pos0 + t.  cos / sin: [max_pos][64] fp32.  Same arithmetic and rounding points as g4r_gemm_bf16_nt + g4r_rope_qkv_bf16
// (the projection is rounded to bf16, rotated in fp32, rounded once more).  tile_cfg: 24 or 28 (0 = 28).
int g4r_gemm_qkv_rope_bf16(const void* A, const void* W, int B, int T, int K, int lda, int ldw, int heads, int head_dim,
                           void* q_out, void* k_cache, void* v_cache, long cache_row, long cache_batch,
                           const float* cos_tab, const float* sin_tab, int pos0, int tile_cfg, void* stream) {
  G4R_REQUIRE(B > 0 && T > 0 && K > 0 && K % BK == 0 && heads > 0, "gemm_qkv_rope: bad shape");
  G4R_REQUIRE(head_dim == 128 && (heads * head_dim) % 256 == 0, "gemm_qkv_rope: head_dim 128, heads * 128 a multiple of 256");
  G4R_REQUIRE(A && W && q_out && k_cache && v_cache && cos_tab && sin_tab, "gemm_qkv_rope: null pointer");
  G4R_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0 && (cache_row % 8) == 0 && (cache_batch % 8) == 0, "gemm_qkv_rope: 16-byte rows");
  if (tile_cfg == 0) tile_cfg = 28;
  G4R_REQUIRE(tile_cfg == 24 || tile_cfg == 28 || tile_cfg == 34 || tile_cfg == 36, "gemm_qkv_rope: 256-wide tiles only (24 / 28 / 34 / 36)");
  GemmArgs p = {};
  p.A = (const h16_t*)A; p.W = (const h16_t*)W; p.C = q_out;
  p.M = B * T; p.N = 3 * heads * head_dim; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = heads * head_dim;
  p.act = 5; p.dbg = g_gemm_dbg;
  p.n_fastest = 0;
  p.splits = 1;
  p.rope_q = (h16_t*)q_out; p.rope_k = (h16_t*)k_cache; p.rope_v = (h16_t*)v_cache;
  p.rope_cos = cos_tab; p.rope_sin = sin_tab; p.rope_krow = cache_row; p.rope_kbatch = cache_batch;
  p.rope_T = T; p.rope_pos0 = pos0; p.rope_HD = heads * head_dim;
  return launch_gemm<0>(p, tile_cfg, (hipStream_t)stream);
}

int g4r_gemm_bf16_nt(const void* A, const void* W, void* C, const float* bias, const void* residual,
                     float* workspace, int M, int N, int K, int lda, int ldw, int ldc, int ldr,
                     int act, int out_f32, int splits, int tile_cfg, void* stream) {
  G4R_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: negative shape");
  if (M == 0 || N == 0) return G4R_OK;
  G4R_REQUIRE(A && W && C, "gemm: null pointer");
  G4R_REQUIRE(act >= 0 && act <= 4, "gemm: act must be 0..4 (5 = the fused RoPE epilogue: g4r_gemm_qkv_rope_bf16)");
  G4R_REQUIRE(act != 4 || (N % 4 == 0 && ldc % 2 == 0 && !residual && !bias && !out_f32 && K % BK == 0),
              "gemm: swiglu epilogue needs N % 4 == 0, bf16 output, no bias/residual");
  if (M == 1 && K % 8 == 0 && (ldw % 8) == 0 && splits == 1 && K >= 512 && K <= G4R_GEMV_MAX_K && (act != 4 || N % 4 == 0)) {
    // single-token decode: weight-streaming GEMV
    gemv_dispatch(g_gemm_dbg >= 100 ? g_gemm_dbg - 100 : -1, 0, (const h16_t*)A, nullptr, 0.f, 0, 0, (const h16_t*)W, C,
                  bias, (const h16_t*)residual, N, K, ldw, act, out_f32, (hipStream_t)stream);
    G4R_CHECK_LAUNCH("gemv_bf16");
    return G4R_OK;
  }
  if (K % BK != 0 || K == 0) {
    G4R_REQUIRE(residual == nullptr, "gemm: residual unsupported on the small-K path");
    long total = (long)M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(small_linear_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const h16_t*)A, (const h16_t*)W, bias, C, M, N, K, lda, ldw, ldc, act, out_f32);
    G4R_CHECK_LAUNCH("small_linear");
    return G4R_OK;
  }
  G4R_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0, "gemm: lda/ldw must be multiples of 8 (16-byte rows)");
  G4R_REQUIRE(splits >= 1, "gemm: splits >= 1");
  G4R_REQUIRE(splits == 1 || workspace, "gemm: split-K needs a workspace of splits*M*N floats");
  GemmArgs p = {};
  p.A = (const h16_t*)A; p.W = (const h16_t*)W; p.C = C; p.ws = workspace; p.bias = bias;
  p.residual = (const h16_t*)residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
  p.act = act; p.out_f32 = out_f32; p.dbg = g_gemm_dbg;
  p.n_fastest = (long)M * 1 > (long)N * 2;
  p.splits = splits;
  return launch_gemm<0>(p, tile_cfg, (hipStream_t)stream);
}

// y [N] = act(W [N, K] . rmsnorm(x; gamma, eps) + bias) + residual for ONE activation row: the projections of the decode
// step with the RMSNorm in front of them fused in (gamma null = no norm).  See g4r_kernels.h.
int g4r_gemv_rmsnorm_bf16(const void* x, const float* gamma, float eps, const void* W, void* C, const float* bias,
                          const void* residual, int N, int K, int ldw, int act, int out_f32, void* stream) {
  G4R_REQUIRE(N > 0 && K >= 512 && K % 8 == 0 && ldw % 8 == 0 && ldw >= K, "gemv: K >= 512, K and ldw multiples of 8");
  G4R_REQUIRE(x && W && C, "gemv: null pointer");
  G4R_REQUIRE(act >= 0 && act <= 4, "gemv: act must be 0..4");
  G4R_REQUIRE(act != 4 || (N % 4 == 0 && !residual && !bias && !out_f32), "gemv: swiglu needs N % 4 == 0, bf16 output");
  G4R_REQUIRE(K <= (gamma ? 8192 : G4R_GEMV_MAX_K), "gemv: K <= 8192 with the fused norm, <= 32736 without (x is staged in 64 KB of LDS)");
  gemv_dispatch(g_gemm_dbg >= 100 ? g_gemm_dbg - 100 : -1, gamma ? 1 : 0, (const h16_t*)x, gamma, eps, 0, 0,
                (const h16_t*)W, C, bias, (const h16_t*)residual, N, K, ldw, act, out_f32, (hipStream_t)stream);
  G4R_CHECK_LAUNCH("gemv_rmsnorm_bf16");
  return G4R_OK;
}

// C [N] = W [N, K] . a + bias + residual, where a [K = H * head_dim] is the decode-attention output assembled on the fly
// from the per-split partials g4r_attn_decode_bf16 wrote with defer_merge (partials [H][splits][head_dim + 2] fp32).
int g4r_gemv_attn_merge_bf16(const float* partials, int splits, int head_dim, const void* W, void* C, const float* bias,
                             const void* residual, int N, int K, int ldw, int out_f32, void* stream) {
  G4R_REQUIRE(N > 0 && K >= 512 && K % 8 == 0 && ldw % 8 == 0 && ldw >= K && K <= G4R_GEMV_MAX_K, "gemv_attn_merge: bad shape");
  G4R_REQUIRE(partials && W && C, "gemv_attn_merge: null pointer");
  G4R_REQUIRE(splits >= 1 && splits <= 64 && (head_dim == 64 || head_dim == 128) && K % head_dim == 0,
              "gemv_attn_merge: splits in [1, 64], head_dim 64 or 128 dividing K");
  gemv_dispatch(-1, 2, nullptr, partials, 0.f, splits, head_dim, (const h16_t*)W, C, bias, (const h16_t*)residual, N, K,
                ldw, 0, out_f32, (hipStream_t)stream);
  G4R_CHECK_LAUNCH("gemv_attn_merge_bf16");
  return G4R_OK;
}

int g4r_conv3x3_nhwc_bf16(const void* X, const void* W, void* Y, const float* bias, const void* zeros,
                          float* workspace, int batch, int H, int Wd, int Cin, int Cout, int groups,
                          long x_group_stride, int act, int out_f32, int splits, int tile_cfg,
                          void* stream) {
  G4R_REQUIRE(batch >= 0 && H > 0 && Wd > 0 && Cin > 0 && Cout > 0 && groups >= 1, "conv3x3: bad shape");
  if (batch == 0) return G4R_OK;
  G4R_REQUIRE(X && W && Y && zeros, "conv3x3: null pointer");
  G4R_REQUIRE((Cin % BK) == 0, "conv3x3: Cin must be a multiple of 64");
  G4R_REQUIRE(act >= 0 && act <= 3, "conv3x3: act must be 0..3");
  G4R_REQUIRE(splits >= 1 && (splits == 1 || workspace), "conv3x3: split-K needs a workspace");
  GemmArgs p = {};
  p.A = (const h16_t*)X; p.W = (const h16_t*)W; p.C = Y; p.ws = workspace; p.bias = bias;
  p.zeros = (const h16_t*)zeros;
  p.M = batch * H * Wd; p.N = Cout; p.K = groups * 9 * Cin;
  p.lda = Cin; p.ldw = p.K; p.ldc = Cout; p.ldr = 0;
  p.act = act; p.out_f32 = out_f32;
  p.H = H; p.Wd = Wd; p.Cin = Cin; p.groups = groups; p.a_group_stride = x_group_stride;
  p.n_fastest = 1; p.dbg = g_gemm_dbg;
  p.splits = splits;
  return launch_gemm<1>(p, tile_cfg, (hipStream_t)stream);
}

// The 3x3 convolutions of ONE fuse round over all pyramid levels as a single implicit GEMM (gpt4roi/models/layers.py:218-236
// applies the SAME ConvModule to every level): X / Y hold the levels' NHWC maps stacked [level][b][y][x][C]; row m decodes
// to (level, b, y, x) and shifts inside its own map.  For the 336^2 pyramid (192^2 + 96^2 + 48^2 + 24^2 = 48960 rows) the
// launch has 192 x 4 = 768 tiles of 256 x 256 = exactly three waves of the 256 CUs, where the four separate launches leave
// the chip partly idle in each of their tails (and the two small maps latency-bound on split-K tiles).
int g4r_conv3x3_mlvl_nhwc_bf16(const void* X, const void* W, void* Y, const float* bias, const void* zeros, int n_levels,
                               const int* level_h, const int* level_w, int batch, int Cin, int Cout, int act,
                               void* stream) {
  G4R_REQUIRE(n_levels >= 1 && n_levels <= 4 && batch >= 1 && Cin > 0 && Cout > 0, "conv3x3_mlvl: 1..4 levels");
  G4R_REQUIRE(X && W && Y && zeros && level_h && level_w, "conv3x3_mlvl: null pointer");
  G4R_REQUIRE((Cin % BK) == 0, "conv3x3_mlvl: Cin must be a multiple of 64");
  G4R_REQUIRE(act >= 0 && act <= 3, "conv3x3_mlvl: act must be 0..3");
  GemmArgs p = {};
  long rows = 0;
  for (int l = 0; l < n_levels; ++l) {
    G4R_REQUIRE(level_h[l] > 0 && level_w[l] > 0, "conv3x3_mlvl: bad map size");
    p.lvl_start[l] = (int)rows;
    p.lvl_h[l] = level_h[l]; p.lvl_w[l] = level_w[l];
    rows += (long)batch * level_h[l] * level_w[l];
  }
  G4R_REQUIRE(rows < (1L << 31), "conv3x3_mlvl: too many rows");
  for (int l = n_levels; l <= 4; ++l) p.lvl_start[l] = (int)rows;
  p.n_lvl = n_levels;
  p.A = (const h16_t*)X; p.W = (const h16_t*)W; p.C = Y; p.bias = bias; p.zeros = (const h16_t*)zeros;
  p.M = (int)rows; p.N = Cout; p.K = 9 * Cin;
  p.lda = Cin; p.ldw = p.K; p.ldc = Cout;
  p.act = act; p.H = level_h[0]; p.Wd = level_w[0]; p.Cin = Cin; p.groups = 1;
  p.n_fastest = 1; p.dbg = g_gemm_dbg; p.splits = 1;
  // tile order: an XCD's wave of 32 workgroups = 16 pixel tiles x 2 weight panels (round 5, tools/order_ab.py: 1356-1360 vs
  // 1321-1345 TF/s for the N-fastest 8 x 4 order on the one-wave-per-SIMD kernel; half the weight-panel bytes per wave).
  // tools: debug mode 40 = the 8 x 4 order, 42 = 32 pixel tiles x 1 weight panel
  p.group_m = -2;
  if (g_gemm_dbg == 40 || g_gemm_dbg == 60) p.group_m = 0;
  if (g_gemm_dbg == 42) p.group_m = -1;
  // round 5: the one-wave-per-SIMD K 64 kernel (1383-1393 vs 1286-1292 TF/s, profiles/r05_w4k64_epilogue.txt); debug mode 60 = the
  // ring ping-pong kernel (tools: A/B arm), also the fallback for maps of 2 GiB and more
  if (g_gemm_dbg != 60 && (size_t)rows * Cin * 2 < 0x7fffffffu) return launch_w4k64<2>(p, (hipStream_t)stream);
  return launch_pp32<2, false, 256, 256, true, 1>(p, (hipStream_t)stream);
}

}  // extern "C"
