// tr_probe.hip -- what ds_read_b64_tr_b16 returns for (a) lane address = lane * 8 bytes and (b) the V-fragment address
// pattern of csrc/attention_v2.hip.  LDS holds its own element index; the expectation is printed beside the result.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/tr_probe.hip -o tools/probe/tr_probe && tools/probe/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((ext_vector_type(4)));
constexpr int VSUB = 64 * 32 + 128;
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x;
  short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(lds + lane * 4));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = a[j];
  // pattern (b): byte offset = gq * VSUB + hi * 128 + (lane & 15) * 8 (+ kbase * 32 with kbase = 16: immediate part)
  const int off = ((lane >> 4) & 1) * VSUB + (lane >> 5) * 128 + (lane & 15) * 8 + 16 * 32;
  short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)((char*)lds + off));
  for (int j = 0; j < 4; ++j) out[256 + lane * 4 + j] = b[j];
}
int main() {
  unsigned short* d;
  if (hipMalloc(&d, 512 * 2) != hipSuccess) return 2;
  k<<<1, 64>>>(d);
  unsigned short h[512];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  int bad_a = 0, bad_b = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int want_a = (l & 15) + j * 16 + (l >> 4) * 64;
      // (b): the image is [dd][key][16 d] with 32-byte rows: element index of V[key][d] in sub-image dd = dd*VSUB/2 + key*16 + d
      const int dd = (l >> 4) & 1, hi = l >> 5, key = 16 + 4 * hi + j, dcol = l & 15;
      const int want_b = dd * (VSUB / 2) + key * 16 + dcol;
      bad_a += h[l * 4 + j] != want_a;
      bad_b += h[256 + l * 4 + j] != want_b;
    }
  for (int l = 0; l < 64; l += 7)
    printf("lane %2d: A %5d %5d %5d %5d   B %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3],
           h[256+l*4], h[256+l*4+1], h[256+l*4+2], h[256+l*4+3]);
  printf("pattern A mismatches %d, pattern B (attention_v2 V fragment) mismatches %d\n", bad_a, bad_b);
  return (bad_a || bad_b) ? 1 : 0;
}
