// tools/probe/mfma_power_probe.hip -- round 6: does the MFMA SHAPE set the throughput under the power cap?
// One wave per SIMD (256 threads per workgroup, one workgroup per CU), all 256 accumulator registers in use, operands held in
// registers (random bf16, different registers for every product so the operand buses toggle), no memory traffic in the loop:
//   mode 0: 16 accumulators of v_mfma_f32_32x32x16_bf16 (the hand kernels' shape): 16 products per pass
//   mode 1: 64 accumulators of v_mfma_f32_16x16x32_bf16 (the vendor library's shape): 64 products per pass = TWICE mode 0's FLOPs
// Host: tools/mfma_power.py launches each for a few seconds and samples socket power / shader clock.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o mfma_power_probe.so mfma_power_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 rnd8(uint32_t& s) {
  union { bf16x8 v; uint16_t u[8]; } r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    // random sign / mantissa, exponent near 1.0 (values in [-2, 2)): what normal-ish activations look like to the multipliers
    r.u[i] = (uint16_t)(((s >> 16) & 0x807fu) | 0x3f80u | (((s >> 9) & 1u) << 6));
  }
  return r.v;
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void mfma_probe(float* out, int iters) {
  uint32_t seed = 1234567u + threadIdx.x * 7919u + blockIdx.x * 104729u;
  bf16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = rnd8(seed); b[i] = rnd8(seed); }
  if (MODE == 0) {
    float16v acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 7], b[(i >> 1) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
  } else {
    float4v acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = float4v{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 64; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 7], b[(i >> 3) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678f) out[threadIdx.x] = s;
  }
}

extern "C" int mfma_probe_launch(int mode, float* out, int iters, int blocks, void* stream) {
  if (mode == 0) hipLaunchKernelGGL(mfma_probe<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  else hipLaunchKernelGGL(mfma_probe<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  return (int)hipGetLastError();
}
