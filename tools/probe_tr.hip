// Probe of ds_read_b64_tr_b16 semantics (developer tool; run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(unsigned short* out, int pitch) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;  // value = row*pitch + col with pitch 64
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  // group g reads the 4x16 block at rows 4g..4g+3, cols 16g..16g+15 (distinct per group to see the mapping)
  const unsigned short* addr = lds + (4 * g + q) * 64 + 16 * g + 4 * p;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 64);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int g = l >> 4, i = l & 15;
    for (int e = 0; e < 4; ++e) {
      int expect = (4 * g + e) * 64 + 16 * g + i;  // lane i of the group gets column i, row e in element e
      if (h[l * 4 + e] != expect) { if (bad < 8) printf("lane %d e %d got %d (row %d col %d) expect %d\n", l, e, h[l*4+e], h[l*4+e]/64, h[l*4+e]%64, expect); ++bad; }
    }
  }
  printf("tr16_b64 probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "matches the guide (lane i <- column i, element e <- row e)", bad);
  return 0;
}
