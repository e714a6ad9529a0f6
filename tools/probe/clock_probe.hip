// s_memtime tick rate: spin for N ticks of __builtin_amdgcn_s_memtime(), time the launch with HIP events.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(long long ticks, long long* out) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  long long t = t0;
  while (t - t0 < ticks) t = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t - t0; out[1] = wall_clock64(); }
}
int main() {
  long long* d; hipMalloc(&d, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (long long ticks : {100000000LL, 400000000LL}) {
    for (int grid : {1, 256}) {
      spin<<<grid, 64>>>(1000, d); hipDeviceSynchronize();
      hipEventRecord(e0); spin<<<grid, 64>>>(ticks, d); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("grid %d: %lld s_memtime ticks in %.3f ms -> %.1f MHz\n", grid, h[0], ms, h[0] / ms / 1e3);
    }
  }
  return 0;
}
