// tools/probe/mfma_probe.hip -- decomposes the GEMM inner loop on the real chip:
//   K1 pure MFMA | K2 + ds_read_b128 fragments | K3 + one barrier per 16 MFMAs | K4 + LDS-DMA loads
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_probe tools/probe/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256, 2) void probe(const char* g, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // init LDS
  for (int i = tid; i < 65536 / 16; i += 256) ((u4*)smem)[i] = u4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  __syncthreads();
  f16v acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  u4 ra = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  bf16x8 fa = __builtin_bit_cast(bf16x8, ra), fb = fa;
  const char* gp = g + ((size_t)blockIdx.x * 256 + tid) * 16;
  for (int it = 0; it < iters; ++it) {
    const int buf = (it & 1) * 32768;
    if (MODE >= 4) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + (size_t)j * 4096 * 16 + (size_t)(it & 63) * 524288),
                                         (__attribute__((address_space(3))) void*)(smem + (buf ^ 32768) + (j * 4 + wave) * 1024), 16, 0, 0);
    }
    bf16x8 f[16];
    if (MODE >= 2) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        f[k] = *reinterpret_cast<const bf16x8*>(smem + buf + ((k * 2048 + lane * 16 + wave * 1024) & 32767));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (MODE >= 2)
        acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[k], f[(k + 1) & 15], acc[k % NACC], 0, 0, 0);
      else
        acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[k % NACC], 0, 0, 0);
    }
    if (MODE >= 3) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int NACC>
void run(const char* name, const char* g, float* out, int blocks) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)probe<MODE, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, NACC><<<blocks, 256, 65536>>>(g, out, 10);
  hipEventRecord(e0);
  probe<MODE, NACC><<<blocks, 256, 65536>>>(g, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 16 * 32768.0;
  printf("%-46s blocks=%4d  %8.3f ms  %8.1f TF/s  (%.1f cycles/MFMA/SIMD @2.1GHz, %d blk/CU)\n", name, blocks, ms,
         flops / ms / 1e9, ms * 1e-3 * 2.1e9 / ((double)iters * 16 * ((blocks + 255) / 256)), (blocks + 255) / 256);
}

int main() {
  char* g; float* out;
  hipMalloc(&g, (size_t)64 * 524288 + (1 << 26));
  hipMemset(g, 0, (size_t)64 * 524288 + (1 << 26));
  hipMalloc(&out, 4096 * 256 * 4);
  for (int blocks : {256, 512}) {
    run<1, 4>("K1 pure MFMA, 4 accumulators", g, out, blocks);
    run<1, 2>("K1 pure MFMA, 2 accumulators", g, out, blocks);
    run<1, 8>("K1 pure MFMA, 8 accumulators", g, out, blocks);
    run<2, 4>("K2 + 16 ds_read_b128 per 16 MFMA", g, out, blocks);
    run<3, 4>("K3 + barrier per 16 MFMA", g, out, blocks);
    run<4, 4>("K4 + 8 global_load_lds per 16 MFMA", g, out, blocks);
  }
  return 0;
}
