// "TN" GEMM: C = A^T B with the reduction over the ROWS of both operands -- the weight gradients of the training rows,
// read from the operands as they lie in memory.  g4r_gemm_tn_bf16 is the plain form (a Linear's dW = dY^T X); the description
// below is the 3x3 convolution WEIGHT gradient read straight from NHWC operands (training rows; the convs are
// gpt4roi/models/layers.py:129-144,191-195,321-325, their weight gradients what torch autograd computes for them in the
// reference's backward).
//
//   dW[co][ci][ky][kx] = sum over pixels p of dY[p][co] * X[p + (ky-1, kx-1)][ci]          (zero outside the image)
//
// is nine GEMMs C_tap = A^T B_tap whose REDUCTION index is the pixel: both operands are NHWC, i.e. [pixel][channel] with the
// channel contiguous -- "TN" GEMMs.  Rounds 2-3 fed them to the NT kernel through channel-major, zero-bordered copies
// (g4r_nhwc_to_cm_padded: three shifted copies of X + one of dY per conv, 13.5 ms of layout passes per step) whose rows are
// 620 KB apart, so that a K tile touched 512 different DRAM pages 64 bytes at a time (857 TF/s at 192^2, 730 at 96^2).
// Here:
//   * operands stay [pixel][channel]; the only copy is a zero-BORDERED NHWC image (g4r_nhwc_pad_bf16: [B][H+2][W+2][C], a
//     streaming copy).  On the bordered grid a tap is a pure row offset, dY's zero border kills the products of the border
//     positions and X's supplies the zeros outside the image -- no masks, no per-tap copies, one A operand for all nine taps;
//   * a K tile is 32 pixel rows x 256 channels = 32 contiguous 512-byte rows per operand, staged by LDS-DMA
//     (buffer_load_dwordx4 ... lds, 1 KiB = two whole rows per wave instruction) into a row-major [32][256] image whose
//     16-byte slots are XOR-swizzled on the SOURCE side (slot ^ 4 * (row & 3));
//   * MFMA fragments need 8 consecutive k per lane for a fixed channel -- the TRANSPOSE of that image: ds_read_b64_tr_b16
//     (a 16-lane group fetches a [4 pixels][16 channels] block, every lane receives one channel's 4 pixels), two per
//     fragment; with the swizzle the 32 lanes served together hit 32 different bank pairs.  Both operands use the same
//     loader, so the permutation of k inside a 16-k step is the same on both sides and drops out of the sum;
//   * 256 x 256 tile, 8 waves (2 x 4, wave tile 128 x 64), K tiles of 32 in a ring of four, the two wave groups one phase
//     apart (one reads fragments + issues its pieces while the other multiplies) -- the loop of gemm_bf16_pp32_kernel;
//   * grid = 16 tiles x 9 taps x S pixel slices, slice-major over the XCDs (the 32 workgroups an XCD runs at a time are two
//     taps x all tiles of ONE slice: they share the A slice and, up to a one-pixel / one-row shift, the B slice in its L2);
//     fp32 partials [S][9][Cout][Cin], reduced straight into the torch layout [Cout][Cin][3][3].
#include "g4r_common.h"

namespace {

struct TnLevel {        // one map geometry (pyramid level); the levels of a fuse round share the weight, hence the output
  const h16_t* A;      // dY on the bordered grid: [32 * nk rows][M], row 0 = bordered pixel 0
  const h16_t* B;      // X on the bordered grid with a guard band in front: tap z reads rows tap_row[z] + k
  int nk, tiles_per_slice, slice0, n_slices;
  int tap_row[9];
  unsigned a_bytes, b_bytes;
  int lda, ldb;
};
struct TnArgs {
  TnLevel lv[4];
  int n_lvl, slices;   // pixel slices of all levels together
  int xcd_first[9];    // XCD x runs work items [xcd_first[x], xcd_first[x + 1]): ranges of equal WORK (items differ in length)
  float* P;            // partials [slices][ntaps][M][N] -- or, `direct`, the result itself [M][ldc] (one slice, one tap)
  int M, N;
  int tiles_m, tiles_n;
  int ntaps, direct, ldc;
};

__device__ __forceinline__ void tn_piece(const void* base, unsigned bytes, void* lds, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

#define TN_BARRIER()                                       \
  do {                                                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

__global__ __launch_bounds__(512) void gemm_tn_kernel(TnArgs p) {
  constexpr int NW = 8, BM = 256, BN = 256, BKT = 32, RING = 4;
  constexpr int ROWB = 512;                       // bytes of one image row (256 channels)
  constexpr int A_BYTES = BKT * ROWB, STAGE_BYTES = 2 * A_BYTES;
  constexpr int TM = 4, TN = 2;                   // wave tile 128 x 64 = 4 x 2 accumulator blocks of 32 x 32
  extern __shared__ __attribute__((aligned(16))) char smem[];   // RING * 32 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int grp = wm;
  // work item: slice-major over the XCDs (workgroup id -> XCD id % 8 in hardware; XCD x owns a contiguous range of items,
  // cut by the host so that the ranges carry equal work)
  const int tiles = p.tiles_m * p.tiles_n;
  const int per_slice = tiles * p.ntaps;
  const int lin = blockIdx.x;
  const int xcd = lin & 7, idx = lin >> 3;
  const int v = p.xcd_first[xcd] + idx;
  if (v >= p.xcd_first[xcd + 1]) return;           // (the grid is 8 x the longest range)
  const int slice = v / per_slice;
  const int rest = v - slice * per_slice;
  const int tap = rest / tiles;
  const int tile = rest - tap * tiles;
  const int tile_m = tile % p.tiles_m, tile_n = tile / p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int li = 0;
#pragma unroll
  for (int l = 1; l < 4; ++l)
    if (l < p.n_lvl && slice >= p.lv[l].slice0) li = l;
  const TnLevel& L = p.lv[li];
  const int t_begin = (slice - L.slice0) * L.tiles_per_slice;
  int t_end = t_begin + L.tiles_per_slice;
  if (t_end > L.nk) t_end = L.nk;
  const int nt = t_end - t_begin;
  const h16_t* const Ap = L.A;
  const h16_t* const Bp = L.B;
  const unsigned a_bytes = L.a_bytes, b_bytes = L.b_bytes;
  const int lda = L.lda, ldb = L.ldb;

  // pieces of a K tile: 16 of A + 16 of B, 1 KiB = image rows 2q, 2q + 1 each; wave w carries A pieces w, w + 8 and B
  // pieces w, w + 8.  Lane l writes LDS bytes [16 l, 16 l + 16) of the piece = row 2q + (l >> 5), slot l & 31, and reads the
  // source slot that the swizzle maps there.
  int a_voff[2], b_voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int prow = 2 * (wave + 8 * j) + (lane >> 5);
    const int slot = (lane & 31) ^ (4 * (prow & 3));
    // columns beyond M / N (edge tiles): an offset past the descriptor's num_records, which the hardware reads as zeros
    a_voff[j] = m0 + slot * 8 < p.M ? (prow * lda + m0 + slot * 8) * 2 : (int)0x80000000;
    b_voff[j] = n0 + slot * 8 < p.N ? (prow * ldb + n0 + slot * 8) * 2 : (int)0x80000000;
  }
  const int b_row0 = L.tap_row[tap];
  auto stage = [&](int t, int buf) {
    char* sa = smem + buf * STAGE_BYTES;
    const int k0 = t * BKT;
    const int sa_off = k0 * lda * 2, sb_off = (b_row0 + k0) * ldb * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) tn_piece(Ap, a_bytes, sa + (wave + 8 * j) * 1024, a_voff[j], sa_off);
#pragma unroll
    for (int j = 0; j < 2; ++j) tn_piece(Bp, b_bytes, sa + A_BYTES + (wave + 8 * j) * 1024, b_voff[j], sb_off);
  };

  // fragment reads: lane (hi = lane >> 5, g = bit 4, c = lane & 15) supplies the 8 bytes at image row
  // 16 ks + 8 rd + 4 hi + (c >> 2), channels 16 * block + 4 * (c & 3) ..+3 and receives channel 16 * block + c of the four
  // rows 16 ks + 8 rd + 4 hi + 0..3 (rd = 0, 1: the two halves of the lane's 8 k values).
  const int c = lane & 15, g = (lane >> 4) & 1, hi = lane >> 5;
  const int r4 = c >> 2;
  const int lane_off = (hi * 4 + r4) * ROWB + (c & 1) * 8;
  unsigned a_off[TM], b_off[TN];
#pragma unroll
  for (int f = 0; f < TM; ++f) {
    const int slot = wm * 16 + f * 4 + g * 2 + ((c >> 1) & 1);
    a_off[f] = (unsigned)(lane_off + ((slot ^ (4 * r4)) << 4));
  }
#pragma unroll
  for (int h = 0; h < TN; ++h) {
    const int slot = wn * 8 + h * 4 + g * 2 + ((c >> 1) & 1);
    b_off[h] = (unsigned)(A_BYTES + lane_off + ((slot ^ (4 * r4)) << 4));
  }
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;

  float16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  uint2v af[2][TM][2], bf[2][TN][2];          // [k-step][fragment][half]
  auto read_frags = [&](int buf) {
    const unsigned sb = lds0 + (unsigned)(buf * STAGE_BYTES);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int f = 0; f < TM; ++f) {
        const unsigned a = sb + a_off[f];
        if (ks == 0) {
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(af[0][f][0]) : "v"(a) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(af[0][f][1]) : "v"(a) : "memory");
        } else {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(af[1][f][0]) : "v"(a) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:12288" : "=v"(af[1][f][1]) : "v"(a) : "memory");
        }
      }
#pragma unroll
      for (int h = 0; h < TN; ++h) {
        const unsigned a = sb + b_off[h];
        if (ks == 0) {
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bf[0][h][0]) : "v"(a) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(bf[0][h][1]) : "v"(a) : "memory");
        } else {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(bf[1][h][0]) : "v"(a) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:12288" : "=v"(bf[1][h][1]) : "v"(a) : "memory");
        }
      }
    }
  };
  // the reads above are asynchronous and invisible to the compiler's counter model: this is their wait, tied to every
  // fragment register so that no MFMA can be scheduled above it
  auto wait_frags = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(af[0][0][0]), "+v"(af[0][1][0]), "+v"(af[0][2][0]), "+v"(af[0][3][0]), "+v"(af[0][0][1]),
                   "+v"(af[0][1][1]), "+v"(af[0][2][1]), "+v"(af[0][3][1])
                 :
                 : "memory");
    // (volatile asm statements keep their order: everything tied below is behind the wait as well)
    asm volatile("" : "+v"(af[1][0][0]), "+v"(af[1][1][0]), "+v"(af[1][2][0]), "+v"(af[1][3][0]), "+v"(af[1][0][1]),
                      "+v"(af[1][1][1]), "+v"(af[1][2][1]), "+v"(af[1][3][1])
                 :
                 : "memory");
    asm volatile("" : "+v"(bf[0][0][0]), "+v"(bf[0][1][0]), "+v"(bf[0][0][1]), "+v"(bf[0][1][1]), "+v"(bf[1][0][0]),
                      "+v"(bf[1][1][0]), "+v"(bf[1][0][1]), "+v"(bf[1][1][1])
                 :
                 : "memory");
  };
  auto frag = [](const uint2v& lo, const uint2v& up) {
    const uint4v w = {lo.x, lo.y, up.x, up.y};
    return __builtin_bit_cast(h16x8, w);
  };
  auto mma = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = G4R_MFMA_32X32X16(frag(af[ks][i][0], af[ks][i][1]), frag(bf[ks][j][0], bf[ks][j][1]), acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  if (nt > 0) {
#pragma unroll
    for (int t = 0; t < RING - 1; ++t)
      if (t < nt) stage(t_begin + t, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (grp == 1) TN_BARRIER();
    for (int i = 0; i < nt; ++i) {
      const int buf = i & (RING - 1);
      // read phase of K tile i: its fragments, then this wave's four pieces of tile i + 3 into the buffer tile i - 1 left
      // (both groups finished reading it: the other group one phase ago, this one two)
      read_frags(buf);
      if (i + RING - 1 < nt) {
        stage(t_begin + i + RING - 1, (i + RING - 1) & (RING - 1));
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // tile i + 1 has landed (tiles i + 2, i + 3 may be in flight)
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      wait_frags();
      TN_BARRIER();
      mma();
      TN_BARRIER();
    }
    if (grp == 0) TN_BARRIER();
  }
  // accumulator register r of block (i, j) is row (r & 3) + 8 (r >> 2) + 4 hi, column lane & 31
  const int row0 = m0 + wm * 128, col0 = n0 + wn * 64 + (lane & 31);
  float* out;
  long ld;
  if (p.direct) {
    out = p.P;
    ld = p.ldc;
  } else {
    out = p.P + (size_t)(slice * p.ntaps + tap) * p.M * p.N;
    ld = p.N;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, col = col0 + j * 32;
        if (row < p.M && col < p.N) out[(size_t)row * ld + col] = acc[i][j][r];
      }
}

// dW[co][ci][tap] = sum_s P[s][tap][co][ci]: one thread per (co, 4 ci); the taps of an output element are contiguous
// (NTAPS = 9: the torch conv layout [Cout][Cin][3][3]; NTAPS = 1: a plain [M][N] matrix)
template <int NTAPS>
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ P, float* __restrict__ dw, int M, int N,
                                                        int splits, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long n4 = N / 4;
  if (i >= (long)M * n4) return;
  const long co = i / n4, c4 = (i - co * n4) * 4;
  float4v s[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) s[t] = float4v{0.f, 0.f, 0.f, 0.f};
  const size_t plane = (size_t)M * N;
  for (int sp = 0; sp < splits; ++sp) {
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
      const float4v v = *reinterpret_cast<const float4v*>(P + ((size_t)sp * NTAPS + t) * plane + co * N + c4);
      s[t] += v;
    }
  }
  float* o = dw + (co * N + c4) * NTAPS;
  if (NTAPS == 1) {
    float4v r = s[0];
    if (accumulate) r += *reinterpret_cast<const float4v*>(o);
    *reinterpret_cast<float4v*>(o) = r;
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) o[e * NTAPS + t] = accumulate ? o[e * NTAPS + t] + s[t][e] : s[t][e];
}

// NHWC [B][H][W][C] -> the bordered grid [B][H+2][W+2][C] starting at row `row0` of dst (interior only: the border and the
// guard rows were zeroed once when the buffer was made and are never written)
__global__ __launch_bounds__(256) void nhwc_pad_kernel(const h16_t* __restrict__ src, h16_t* __restrict__ dst, long pixels,
                                                       int H, int W, int C8, long row0) {
  const long total = pixels * C8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long pix = i / C8;
    const int v = (int)(i - pix * C8);
    const long b = pix / ((long)H * W);
    const int rem = (int)(pix - b * (long)H * W);
    const int y = rem / W, x = rem - y * W;
    const long prow = row0 + (b * (H + 2) + y + 1) * (long)(W + 2) + x + 1;
    reinterpret_cast<uint4v*>(dst)[prow * C8 + v] = reinterpret_cast<const uint4v*>(src)[i];
  }
}

G4rPerDeviceOnce g_tn_lds;

}  // namespace

extern "C" {

// src [B][H][W][C] bf16 -> dst rows row0 + ((b (H+2) + y + 1) (W+2) + x + 1), C channels each (see nhwc_pad_kernel).
int g4r_nhwc_pad_bf16(const void* src, void* dst, int B, int H, int W, int C, long row0, void* stream) {
  G4R_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && row0 >= 0, "nhwc_pad: bad shape");
  G4R_REQUIRE(src && dst, "nhwc_pad: null pointer");
  const long total = (long)B * H * W * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(nhwc_pad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const h16_t*)src,
                     (h16_t*)dst, (long)B * H * W, H, W, C / 8, row0);
  G4R_CHECK_LAUNCH("nhwc_pad");
  return G4R_OK;
}

// dW [Cout][Cin][3][3] fp32 of a 3x3 / stride 1 / pad 1 convolution from bordered NHWC operands, summed over the
// n_levels map geometries that share the weight (the levels of a fuse round; 1 for a plain conv).  Per level l:
//   dy_pads[l]  [krows_l][Cout]         row = bordered pixel index (b (H+2) + y') (W+2) + x', zero on the border and beyond
//                                       B (H+2) (W+2); krows_l = that count rounded up to a multiple of 32
//   x_pads[l]   [guard_l + krows_l + guard_l][Cin], guard_l = W_l + 3 rows of zeros in front of bordered pixel 0 / behind
// The pixel axis of every level is cut into slices of about `slice_tiles` K tiles (32 pixels each); partials
// [total slices][9][Cout][Cin] fp32 -- g4r_conv3x3_wgrad_nhwc_slices() returns the count for the same arguments.
// Cout and Cin multiples of 256; accumulate: dw += .
// items in (slice, tap, tile) order cut into eight ranges of equal work; an item costs its K tiles + ~48 tiles' worth of
// ring fill, epilogue and launch
static void tn_balance(TnArgs& a) {
  const int per_slice = a.tiles_m * a.tiles_n * a.ntaps;
  long total_work = 0;
  for (int l = 0; l < a.n_lvl; ++l) {
    const TnLevel& L = a.lv[l];
    for (int s = 0; s < L.n_slices; ++s) {
      int len = L.nk - s * L.tiles_per_slice;
      if (len > L.tiles_per_slice) len = L.tiles_per_slice;
      total_work += (long)(len + 48) * per_slice;
    }
  }
  long done = 0;
  int item = 0, x = 1;
  a.xcd_first[0] = 0;
  for (int l = 0; l < a.n_lvl; ++l) {
    const TnLevel& L = a.lv[l];
    for (int s = 0; s < L.n_slices; ++s) {
      int len = L.nk - s * L.tiles_per_slice;
      if (len > L.tiles_per_slice) len = L.tiles_per_slice;
      for (int i = 0; i < per_slice; ++i) {
        while (x < 8 && done * 8 >= total_work * x) a.xcd_first[x++] = item;
        done += len + 48;
        ++item;
      }
    }
  }
  while (x <= 8) a.xcd_first[x++] = item;
}

static int tn_launch(TnArgs& a, void* stream) {
  const int lds = 4 * 32768;
  if (g_tn_lds.first()) {
    const hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { g_tn_lds.failed(); return g4r_note_hip_error(e, "gemm_tn: hipFuncSetAttribute"); }
  }
  int longest = 0;
  for (int x = 0; x < 8; ++x)
    if (a.xcd_first[x + 1] - a.xcd_first[x] > longest) longest = a.xcd_first[x + 1] - a.xcd_first[x];
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(8 * longest), dim3(512), lds, (hipStream_t)stream, a);
  G4R_CHECK_LAUNCH("gemm_tn");
  return G4R_OK;
}

static int tn_plan(TnArgs& a, int n_levels, const int* heights, const int* widths, int B, int Cin, int Cout, int slice_tiles) {
  a.n_lvl = n_levels;
  a.M = Cout; a.N = Cin;
  a.tiles_m = Cout / 256; a.tiles_n = Cin / 256;
  a.ntaps = 9; a.direct = 0; a.ldc = Cin;
  int slices = 0;
  for (int l = 0; l < n_levels; ++l) {
    const int H = heights[l], W = widths[l];
    const long kp = (long)B * (H + 2) * (W + 2);
    const long krows = (kp + 31) / 32 * 32;
    const long guard = W + 3;
    const long a_bytes = krows * Cout * 2, b_bytes = (guard + krows + guard) * Cin * 2;
    if (a_bytes >= 0x7fffffffL || b_bytes >= 0x7fffffffL) return -1;
    TnLevel& L = a.lv[l];
    L.nk = (int)(krows / 32);
    L.n_slices = (L.nk + slice_tiles - 1) / slice_tiles;
    if (L.n_slices < 1) L.n_slices = 1;
    L.tiles_per_slice = (L.nk + L.n_slices - 1) / L.n_slices;
    L.slice0 = slices;
    slices += L.n_slices;
    for (int t = 0; t < 9; ++t) L.tap_row[t] = (int)(guard + (t / 3 - 1) * (W + 2) + (t % 3 - 1));
    L.a_bytes = (unsigned)a_bytes; L.b_bytes = (unsigned)b_bytes;
    L.lda = Cout; L.ldb = Cin;
  }
  a.slices = slices;
  tn_balance(a);
  return slices;
}

static bool tn_shape_ok(int n_levels, const int* heights, const int* widths, int B, int Cin, int Cout, int slice_tiles) {
  if (n_levels < 1 || n_levels > 4 || !heights || !widths || B <= 0 || slice_tiles < 8) return false;
  if (Cin <= 0 || Cout <= 0 || Cin % 256 || Cout % 256) return false;
  for (int l = 0; l < n_levels; ++l)
    if (heights[l] <= 0 || widths[l] <= 0) return false;
  return true;
}

int g4r_conv3x3_wgrad_nhwc_slices(int n_levels, const int* heights, const int* widths, int B, int Cin, int Cout,
                                  int slice_tiles) {
  if (!tn_shape_ok(n_levels, heights, widths, B, Cin, Cout, slice_tiles)) return -1;
  TnArgs a;
  return tn_plan(a, n_levels, heights, widths, B, Cin, Cout, slice_tiles);
}

int g4r_conv3x3_wgrad_nhwc_bf16(const void* const* dy_pads, const void* const* x_pads, int n_levels, const int* heights,
                                const int* widths, int B, int Cin, int Cout, int slice_tiles, float* partials, float* dw,
                                int accumulate, void* stream) {
  G4R_REQUIRE(tn_shape_ok(n_levels, heights, widths, B, Cin, Cout, slice_tiles),
              "conv3x3_wgrad_nhwc: 1-4 levels, channels multiples of 256, slices of >= 8 K tiles");
  G4R_REQUIRE(dy_pads && x_pads && partials && dw, "conv3x3_wgrad_nhwc: null pointer");
  TnArgs a;
  const int slices = tn_plan(a, n_levels, heights, widths, B, Cin, Cout, slice_tiles);
  G4R_REQUIRE(slices > 0, "conv3x3_wgrad_nhwc: operand beyond the 2 GiB of a 32-bit offset");
  for (int l = 0; l < n_levels; ++l) {
    G4R_REQUIRE(dy_pads[l] && x_pads[l], "conv3x3_wgrad_nhwc: null level pointer");
    a.lv[l].A = (const h16_t*)dy_pads[l];
    a.lv[l].B = (const h16_t*)x_pads[l];
  }
  a.P = partials;
  {
    const int rc = tn_launch(a, stream);
    if (rc != G4R_OK) return rc;
  }
  const long n = (long)Cout * (Cin / 4);
  hipLaunchKernelGGL((tn_reduce_kernel<9>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)partials, dw, Cout, Cin, slices, accumulate);
  G4R_CHECK_LAUNCH("conv3x3_wgrad_reduce");
  return G4R_OK;
}

// C [M][N] fp32 (+)= A^T B with A [K][lda] (element (k, m)), B [K][ldb] (element (k, n)) bf16: the reduction index is the
// ROW of both operands -- the weight gradient of a Linear, dW [N_out][K_in] = dY^T X with dY [tokens][N_out], X [tokens][K_in]
// (torch autograd's grad_weight for the nn.Linear layers of the decoder, the projector and the 1x1 input convs), read as it
// lies: no transposed copies.  M, N, lda, ldb multiples of 8; K any (the rows beyond K read as zeros).
// slices = 1 and accumulate = 0: stored directly (ldc = row stride of C).  Otherwise the K axis is cut into `slices`
// ranges, partials [slices][M][N] fp32, and a reduce writes (or adds to) C, which must then be dense (ldc == N).
int g4r_gemm_tn_bf16(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, float* partials,
                     int slices, int accumulate, void* stream) {
  G4R_REQUIRE(M > 0 && N > 0 && K > 0 && M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N,
              "gemm_tn: M, N and the row strides must be multiples of 8");
  G4R_REQUIRE(A && B && C && slices >= 1 && slices <= 256, "gemm_tn: bad arguments");
  const bool direct = slices == 1 && !accumulate;
  G4R_REQUIRE(direct || (partials && ldc == N), "gemm_tn: sliced / accumulating form needs the workspace and a dense C");
  G4R_REQUIRE(ldc >= N, "gemm_tn: ldc < N");
  const long a_bytes = (long)K * lda * 2, b_bytes = (long)K * ldb * 2;
  G4R_REQUIRE(a_bytes < 0x7fffffffL && b_bytes < 0x7fffffffL, "gemm_tn: operand beyond the 2 GiB of a 32-bit offset");
  TnArgs a;
  a.n_lvl = 1;
  a.M = M; a.N = N;
  a.tiles_m = (M + 255) / 256; a.tiles_n = (N + 255) / 256;
  a.ntaps = 1; a.direct = direct ? 1 : 0; a.ldc = ldc;
  TnLevel& L = a.lv[0];
  L.A = (const h16_t*)A; L.B = (const h16_t*)B;
  L.nk = (K + 31) / 32;
  L.n_slices = slices > L.nk ? L.nk : slices;
  L.tiles_per_slice = (L.nk + L.n_slices - 1) / L.n_slices;
  L.n_slices = (L.nk + L.tiles_per_slice - 1) / L.tiles_per_slice;
  L.slice0 = 0;
  for (int t = 0; t < 9; ++t) L.tap_row[t] = 0;
  L.a_bytes = (unsigned)a_bytes; L.b_bytes = (unsigned)b_bytes;
  L.lda = lda; L.ldb = ldb;
  a.slices = L.n_slices;
  a.P = direct ? C : partials;
  tn_balance(a);
  {
    const int rc = tn_launch(a, stream);
    if (rc != G4R_OK) return rc;
  }
  if (!direct) {
    const long n = (long)M * (N / 4);
    hipLaunchKernelGGL((tn_reduce_kernel<1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)partials, C, M, N, a.slices, accumulate);
    G4R_CHECK_LAUNCH("gemm_tn_reduce");
  }
  return G4R_OK;
}

}  // extern "C"
