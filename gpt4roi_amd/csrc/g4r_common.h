// g4r_common.h -- shared device/host helpers for the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define G4R_OK 0
#define G4R_ERR_INVALID_ARG 1
#define G4R_ERR_LAUNCH 2
#define G4R_ERR_UNSUPPORTED 3

// Records the text of a failing HIP call for g4r_last_error(); returns G4R_ERR_LAUNCH.
int g4r_note_hip_error(hipError_t e, const char* where);
int g4r_note_error(int code, const char* what);

// Every launcher ends with this: surface launch-configuration errors synchronously, as the
// reference does with AT_CUDA_CHECK(cudaGetLastError()) (roi_align_cuda.cu:29,57).
#define G4R_CHECK_LAUNCH(where)                                   \
  do {                                                            \
    hipError_t _e = hipGetLastError();                            \
    if (_e != hipSuccess) return g4r_note_hip_error(_e, where);   \
  } while (0)

#define G4R_REQUIRE(cond, what)                                        \
  do {                                                                 \
    if (!(cond)) return g4r_note_error(G4R_ERR_INVALID_ARG, what);     \
  } while (0)

typedef uint16_t bf16_t;  // raw bfloat16 bit pattern

typedef short short8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_to_f32(bf16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
// float -> bf16, round-to-nearest-even: gfx950 has the conversion in hardware
// (v_cvt_pk_bf16_f32); the vector cast below compiles to exactly one such instruction per pair.
typedef __bf16 g4r_bf16x2 __attribute__((ext_vector_type(2)));
typedef float g4r_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const g4r_f32x2 f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, g4r_bf16x2));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

static inline int g4r_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
