// tools/probe/loadpath_probe.hip -- what bounds the operand stream of a 256 x 256 GEMM tile on one CU?
// Every workgroup (512 threads, one per CU) streams the operands of its own output tile exactly like the ring ping-pong GEMM:
// per "K tile" it loads 32 KB = 512 rows (256 of A + 256 of W, row pitch = K * 2 bytes) x 64 bytes, advancing 64 bytes along
// K per iteration; no MFMAs, no fragment reads.  Modes:
//   0  LDS-DMA (buffer_load_dwordx4 ... lds), 16 rows x 64 B per wave instruction  (half cache lines: the production pattern)
//   1  LDS-DMA, 8 rows x 128 B per instruction, 128 B of K per iteration over 256 rows (whole lines, same bytes)
//   2  buffer_load_dwordx4 into VGPRs (kept live by an empty asm), 16 rows x 64 B
//   3  buffer_load_dwordx4 into VGPRs, 8 rows x 128 B
//   4  mode 2 + ds_write_b128 of what was loaded one iteration earlier (register staging)
// A/W are laid out as M x K and N x K bf16 with M = N = 4096, K = 4096 (the 4096^3 case: 256 tiles); prints us per K tile per
// workgroup and bytes / clock / CU at the measured shader clock (s_memtime ticks are 100 MHz: we use wall time and the
// nominal 2.4 GHz only for the B/clk column).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/loadpath_probe tools/probe/loadpath_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void piece(const void* base, unsigned bytes, void* lds, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ u4 vload(const void* base, unsigned bytes, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void stream(const char* A, const char* W, unsigned bytes, int K, int ktiles, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // 4 x 32 KB ring
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_m = blockIdx.x % 16, tile_n = blockIdx.x / 16;
  constexpr bool FULL = MODE == 1 || MODE == 3;
  int voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pslot = (j * 8 + wave) * 64 + lane;                  // 0 .. 2047
    int row, slot;
    if (FULL) { row = (pslot >> 3) & 127; slot = pslot & 7; }      // 8 rows x 8 slots per piece, 128 rows per operand
    else { row = (pslot >> 2) & 255; slot = pslot & 3; }           // 16 rows x 4 slots per piece, 256 rows per operand
    const bool isw = FULL ? (pslot >> 10) : (pslot >> 10);         // pieces 0-15 -> A, 16-31 -> W
    const int base_row = (isw ? tile_n : tile_m) * 256 + row;
    voff[j] = base_row * K * 2 + slot * 16;
  }
  u4 keep[4] = {};
  u4 prev[4] = {};
  float s = 0.f;
  for (int t = 0; t < ktiles; ++t) {
    const int buf = (t & 3) * 32768;
    const int soff = FULL ? ((t * 128) & (K * 2 - 128)) : ((t * 64) & (K * 2 - 64));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const char* base = (j >= 2) ? W : A;
      if (MODE <= 1) piece(base, bytes, smem + buf + (j * 8 + wave) * 1024, voff[j], soff);
      else keep[j] = vload(base, bytes, voff[j], soff);
    }
    if (MODE <= 1) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // two tiles in flight, as the GEMM keeps them
    } else {
      if (MODE == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u4*>(smem + buf + (j * 8 + wave) * 1024 + lane * 16) = prev[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("" : "+v"(keep[j]));
        prev[j] = keep[j];
      }
    }
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  s += reinterpret_cast<float*>(smem)[tid] + __uint_as_float(prev[0][0] & 0x3fffffffu);
  out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
void run(const char* name, const char* A, const char* W, float* out, int wgs) {
  const int K = 4096, ktiles = 4096;                              // 4096 "K tiles" of 64 bytes: wraps inside the row
  hipFuncSetAttribute((const void*)stream<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  stream<MODE><<<wgs, 512, 131072>>>(A, W, 4096u * 4096u * 2u, K, 64, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  stream<MODE><<<wgs, 512, 131072>>>(A, W, 4096u * 4096u * 2u, K, ktiles, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double us_tile = ms * 1e3 / ktiles;
  printf("%-58s %3d wgs: %7.3f us per 32 KB K tile per CU = %5.1f GB/s per CU = %5.1f B/clk at 2.4 GHz; chip %5.2f TB/s\n", name, wgs, us_tile,
         32768.0 / us_tile / 1e3, 32768.0 / us_tile / 1e3 / 2.4, wgs * 32768.0 / us_tile / 1e6);
}

int main() {
  char *A, *W;
  float* out;
  hipMalloc(&A, 4096ul * 4096 * 2);
  hipMalloc(&W, 4096ul * 4096 * 2);
  hipMalloc(&out, 256 * 512 * 4);
  hipMemset(A, 1, 4096ul * 4096 * 2);
  hipMemset(W, 1, 4096ul * 4096 * 2);
  for (int wgs : {256, 128, 32}) {
    run<0>("0 LDS-DMA, 16 rows x 64 B per instruction (production)", A, W, out, wgs);
    run<1>("1 LDS-DMA, 8 rows x 128 B per instruction", A, W, out, wgs);
    run<2>("2 VGPR loads, 16 rows x 64 B", A, W, out, wgs);
    run<3>("3 VGPR loads, 8 rows x 128 B", A, W, out, wgs);
    run<4>("4 VGPR loads 16 x 64 B + ds_write_b128 (register staging)", A, W, out, wgs);
  }
  return 0;
}
