// runtime.hip -- error bookkeeping shared by every launcher (no device code).
#include <stdio.h>
#include <string.h>

#include "g4r_common.h"

namespace {
thread_local char g_last_error[512] = "";
}

int g4r_note_hip_error(hipError_t e, const char* where) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s", where, hipGetErrorString(e));
  return G4R_ERR_LAUNCH;
}

int g4r_note_error(int code, const char* what) {
  snprintf(g_last_error, sizeof(g_last_error), "%s", what);
  return code;
}

extern "C" {
int g4r_abi_version(void) { return 5; }   // 5 (round 4): + the f16 instantiation, ragged decode, TN GEMM / NHWC weight gradients, weight layouts
const char* g4r_last_error(void) { return g_last_error; }
}
