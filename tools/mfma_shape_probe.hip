// Developer probe (run on the GPU box): does the bf16 MFMA shape matter for the halo convolution's inner loop?
// Both kernels run the loop's instruction MIX per 64 x 64 x 32 wave-tile step on random LDS data - 2 activation fragment reads
// (ds_read_b128) + 2 weight fragments as transposing reads (ds_read_b64_tr_b16) per k16, one wait per step, then the step's MFMAs
// (hipcc may slide the next step's reads under them; with two workgroups per CU the other one fills the gaps anyway) - with v_mfma_f32_32x32x16_bf16 (2 x 2 tiles, 4 MFMAs per k16) or
// v_mfma_f32_16x16x32_bf16 (4 x 4 tiles, 16 MFMAs per k32).  Same FLOPs, same LDS bytes, same accumulator registers (64).
// No global traffic inside the loop, 4 waves per workgroup, 1 or 2 workgroups per CU.  The addresses are bank-friendly but NOT a
// valid GEMM fragment layout: this measures issue / clock behaviour, not numerics.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_probe tools/mfma_shape_probe.hip && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_BYTES (72 * 1024)  // like the 256 x 64 halo tile: two workgroups fit a CU

__device__ __forceinline__ bf16x8 join(s16x4 lo, s16x4 hi) {
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

template <bool SMALL>
__global__ void __launch_bounds__(256, 2) probe(const unsigned* __restrict__ seed, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < LDS_BYTES / 4; i += 256) {  // random bf16 pairs with bounded exponents
    unsigned r = seed[(i * 7 + blockIdx.x) & 65535];
    reinterpret_cast<unsigned*>(smem)[i] = (r & 0x807F807Fu) | 0x3F003F00u;
  }
  __syncthreads();
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // activation rows: 128-byte rows, 16-byte chunk swizzled by the row; weight image behind them
  const int fr = SMALL ? (lane & 15) : (lane & 31), fk = SMALL ? (lane >> 4) : (lane >> 5);
  const unsigned a_row0 = lds0 + (wave * 64 + fr) * 128;
  const unsigned b_img = lds0 + 40 * 1024;
  f32x16 acc32[2][2];
  f32x4 acc16[4][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc16[i][j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const unsigned step = (unsigned)(it & 7) * 4096;  // walk through the images like taps do
    if (!SMALL) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {  // two k16 steps = the FLOPs of one k32 step below
        bf16x8 a[2];
        s16x4 bl[2], bh[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned addr = a_row0 + i * 32 * 128 + ((((fk + 2 * s) ^ (fr >> 1)) & 7) << 4) + (step & 0x3000);
          asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"(addr) : "memory");
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const unsigned addr = b_img + ((16 * s + 8 * (lane >> 5) + ((lane & 15) >> 2)) * 128) + (((j * 4 + ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1)) ^ (lane & 4)) << 4) + (lane & 1) * 8 + (step & 0x3000);
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bl[j]) : "v"(addr) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bh[j]) : "v"(addr + 512) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1])::"memory");
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join(bl[j], bh[j]), a[i], acc32[i][j], 0, 0, 0);
      }
    } else {
      bf16x8 a[4];
      s16x4 bl[4], bh[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // 16 rows x 32 k per fragment: 4 k-groups of 8 per row
        const unsigned addr = a_row0 + i * 16 * 128 + (((fk ^ (fr >> 1)) & 7) << 4) + (step & 0x3000);
        asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"(addr) : "memory");
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned addr = b_img + ((8 * (lane >> 4) + ((lane & 15) >> 2)) * 128) + (((j * 2 + ((lane & 3) >> 1)) ^ (lane & 4)) << 4) + (lane & 1) * 8 + (step & 0x3000);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bl[j]) : "v"(addr) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bh[j]) : "v"(addr + 512) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]),
                   "+v"(bl[2]), "+v"(bh[2]), "+v"(bl[3]), "+v"(bh[3])::"memory");
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(bl[j], bh[j]), a[i], acc16[i][j], 0, 0, 0);
    }
  }
  float t = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) t += acc32[i][j][e];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) t += acc16[i][j][e];
  if (t == 123.456f) out[0] = t;  // keep the accumulators alive
}

template <bool SMALL>
static double run(const unsigned* seed, float* out, int wgs, int iters) {
  hipFuncSetAttribute((const void*)probe<SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<SMALL>, dim3(wgs), dim3(256), LDS_BYTES, 0, seed, out, iters);
  hipEventRecord(e0, 0);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe<SMALL>, dim3(wgs), dim3(256), LDS_BYTES, 0, seed, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)wgs * 4 /*waves*/ * iters * 2.0 * 64 * 64 * 32;
  return flop * reps / (ms * 1e-3) / 1e12;
}

int main() {
  unsigned* h = (unsigned*)malloc(65536 * 4);
  srand(1);
  for (int i = 0; i < 65536; ++i) h[i] = ((unsigned)rand() << 16) ^ (unsigned)rand();
  unsigned* seed; float* out;
  hipMalloc(&seed, 65536 * 4); hipMalloc(&out, 64);
  hipMemcpy(seed, h, 65536 * 4, hipMemcpyHostToDevice);
  const int iters = 20000;
  for (int pass = 0; pass < 2; ++pass)
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
      const double t32 = run<false>(seed, out, 256 * per_cu, iters);
      const double t16 = run<true>(seed, out, 256 * per_cu, iters);
      printf("%d workgroup(s) per CU: 32x32x16 %7.1f TFLOP/s   16x16x32 %7.1f TFLOP/s   ratio %.3f\n", per_cu, t32, t16, t16 / t32);
      fflush(stdout);
    }
  return 0;
}
