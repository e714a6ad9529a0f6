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

// ---- the 16-bit storage type ------------------------------------------------------------------------------------
// Every kernel is written against `h16_t` (raw 16-bit pattern in HBM / LDS), the MFMA operand vector `h16x8` and the five
// conversion helpers below; fp32 accumulates everywhere.  The library is compiled TWICE from the same sources
// (gpt4roi_amd/build.py): the default instantiation stores bfloat16 (the reference's training dtype, train_stage1.sh:19),
// -DG4R_F16 stores IEEE half (the reference's inference dtype: app.py:74-98 loads the model, the boxes :271 and the images
// :296 as fp16).  Same MFMA rate (v_mfma_f32_32x32x16_{bf16,f16}), same byte layout, three more mantissa bits.  The entry
// points of the second instantiation carry `f16` where the first carries `bf16` (include/g4r_f16_names.h).
typedef uint16_t h16_t;

typedef short short8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
typedef float g4r_f32x2 __attribute__((ext_vector_type(2)));

#ifdef G4R_F16
typedef _Float16 g4r_h16_native;
#define G4R_DTYPE_NAME "f16"
#define G4R_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define G4R_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
typedef __bf16 g4r_h16_native;
#define G4R_DTYPE_NAME "bf16"
#define G4R_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define G4R_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif
typedef g4r_h16_native h16x8 __attribute__((ext_vector_type(8)));
typedef g4r_h16_native h16x4 __attribute__((ext_vector_type(4)));
typedef g4r_h16_native g4r_h16x2 __attribute__((ext_vector_type(2)));

// float -> 16 bit, round-to-nearest-even: gfx950 has both conversions in hardware (v_cvt_pk_bf16_f32 / v_cvt_f16_f32);
// the vector cast below compiles to one packed instruction (bf16) or two converts and a pack (f16).
__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi) {
  const g4r_f32x2 f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, g4r_h16x2));
}
__device__ __forceinline__ h16_t f32_to_h16(float f) { return (h16_t)(pack_h16x2(f, 0.f) & 0xffffu); }
#ifdef G4R_F16
__device__ __forceinline__ float h16_to_f32(h16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float h16lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float h16hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
#else
__device__ __forceinline__ float h16_to_f32(h16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ float h16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#endif

// hipFuncSetAttribute(hipFuncAttributeMaxDynamicSharedMemorySize) applies to the CURRENT device only: launchers remember it
// per (kernel, device), not in one process-wide flag (a second GPU or a device reset would otherwise launch the
// > 64 KB LDS kernels without it).  A benign race between host threads sets the attribute twice.
// A failing hipFuncSetAttribute must NOT leave the device marked: the launcher calls failed() on its error path,
// so the next call retries the attribute instead of launching a > 64 KB dynamic-LDS kernel without the raised limit.
struct G4rPerDeviceOnce {
  bool done[64] = {};
  bool first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
    if (done[d]) return false;
    done[d] = true;
    return true;
  }
  void failed() {
    int d = 0;
    if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) done[d] = false;
  }
};

static inline int g4r_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
