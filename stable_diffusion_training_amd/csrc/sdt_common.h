// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libsdtrain_hip.so.
// Wave = 64 lanes everywhere. bf16 is carried as raw uint16 in memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdt.h"

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (8 bf16)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define SDT_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(unsigned short, b);
}
typedef __attribute__((ext_vector_type(2))) float sdt_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 sdt_bf16x2_t;
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  // the vector form is what makes hipcc emit ONE v_cvt_pk_bf16_f32 lo, hi (two scalar casts cost cvt + cvt + shift + or)
  const sdt_f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, sdt_bf16x2_t));
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); scratch: >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += scratch[i];
  return r;
}

// ---- ordered cross-workgroup reductions (no float atomics on data) -------------------------------------------------------
// Workgroups that feed one sum each store a PARTIAL to their own slot (plain stores), then call sdt_arrive_last on the sum's
// counter: it returns true (workgroup-uniform) in the workgroup that arrived last, which then adds all the partials in a fixed
// order and writes the result - the same bits whichever workgroup happens to be last.  The counter is zero on entry and is reset
// here, so a zeroed counter area is reusable launch after launch (include/sdt.h: "split workspace" contract - the first 64 KiB
// of the workspace are counters, the rest is scratch that needs no initialisation).
// Release / acquire as MI355X_MICROARCH.md prescribes for a placement-independent hand-off: every storing wave drains its
// stores, the workgroup meets, ONE lane releases at agent scope in front of the relaxed ticket; the last arriver acquires at agent
// scope and the workgroup meets again before any thread loads other workgroups' partials.
#define SDT_WS_COUNTER_BYTES 65536
// WT = true: the caller has stored EVERY partial with write-through agent-scope stores (sdt_store_wt) - then the drain + barrier in
// front of the relaxed ticket publishes them and no L2 write-back (release fence, ~2 us per workgroup, contended when thousands of
// workgroups end together) is needed: the form split_reduce in gemm.hip uses.  The last arriver acquires either way and must read
// the partials of a WT producer with sdt_load_wt.
template <bool WT>
__device__ __forceinline__ bool sdt_arrive_last(int* counter, int expected, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!WT) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == expected - 1;
    if (last) {
      __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}
template <typename T>
__device__ __forceinline__ void sdt_store_wt(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }  // global_store ... sc1
template <typename T>
__device__ __forceinline__ T sdt_load_wt(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }  // global_load ... sc1

// flax nn.gelu(approximate=True) and its derivative (GEGLU: elementwise.hip's kernels and gemm.hip's fused feed-forward epilogues).
// No floating-point contraction inside: the two translation units must round identically (the fused epilogues are tested bit for bit
// against the separate kernels), and which multiply-adds get fused would otherwise depend on the surrounding code.
__device__ __forceinline__ float gelu_tanh_f(float x) {
#pragma clang fp contract(off)
  const float k = 0.7978845608028654f;
  return 0.5f * x * (1.f + tanhf(k * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float gelu_tanh_grad(float x) {
#pragma clang fp contract(off)
  const float k = 0.7978845608028654f;
  float u = k * (x + 0.044715f * x * x * x);
  float th = tanhf(u);
  float du = k * (1.f + 3.f * 0.044715f * x * x);
  return 0.5f * (1.f + th) + 0.5f * x * (1.f - th * th) * du;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }

// ---- host-side error plumbing (thread-local message, negative return codes) ----
void sdt_set_error(const char* fmt, ...);
#define SDT_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      sdt_set_error(__VA_ARGS__);           \
      return SDT_ERR_INVALID_ARG;           \
    }                                       \
  } while (0)
#define SDT_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      sdt_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return SDT_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

static inline int sdt_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline int sdt_grid_1d(long work_items, int per_block, int cap = 8192) {
  long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}
