// HBM-bound optimizer sweep: global-norm reduction, fused clip + Lion (8-bit blockwise or fp32
// momentum) + weight decay + parameter update + EMA + bf16 re-cast, one pass over the flat
// parameter buffer.  Replaces, for the reference:
//   optax.clip_by_global_norm(1)                      training_utils.py:380, :417
//   lion_quant.py:52-64 (_quantize/_dequantize), :66-92 (block codec), :133-154 (update_fn)
//   add_decayed_weights / -lr / apply_updates         lion_quant.py:201-211, training_utils.py:732-733
//   compute_model_ema                                 training_utils.py:537-544
// Algorithmic bytes per parameter: g 4r + p 4r/4w + code 1r/1w + scale (4r/4w)/block (+ema 4r/4w, +bf16 2w).
// This translation unit is compiled with -ffp-contract=off so that every multiply/add rounds
// separately, as the NumPy float32 oracle (and XLA elementwise f32) does.
#include "sdt_common.h"

#define LION_OFFSET 3.7398995e-09f  // lion_quant.py:49

__device__ __forceinline__ float lion_deq(int code) {  // lion_quant.py:61-64
  float t = (float)code / 127.0f;
  float t2 = t * t;
  float t4 = t2 * t2;
  return t4 * t - LION_OFFSET;
}
// _quantize (lion_quant.py:52-59): code = rint(sign(x + offset) * |x + offset|^(1/5) * 127).
// The same integer without powf: _quantize is a monotone step function of a = |x + offset|, described exactly by the 127
// float32 thresholds thr[c] = smallest a whose code is >= c (thr[0] = 0, thr[128] = +inf; built on the host with the float32
// power / multiply / rint of the definition, lion_codec.quantization_thresholds).  v_log_f32 / v_exp_f32 give the code to
// well within one unit; two threshold reads settle it: thr[c] <= a < thr[c + 1].  Bit-exact against the host definition at
// every rounding boundary, and no device/host pow ulp disagreement (the HBM-bound sweep was VALU-bound on powf).
__device__ __forceinline__ int lion_quant_tab(float x, const float* __restrict__ thr) {
  const float xo = x + LION_OFFSET;
  const float a = fabsf(xo);
  const float q = __builtin_amdgcn_exp2f(0.2f * __builtin_amdgcn_logf(a)) * 127.0f;
  const float r = rintf(q);
  int c = (int)fminf(fmaxf(r, 0.f), 127.f);
  // the estimate is good to ~2e-5 code units (1-ulp v_log / v_exp): only values within 2.5e-4 of a rounding boundary can
  // land on the wrong side and need the table (about one element in 2000; the others skip the two LDS reads)
  if (fabsf(q - r) > 0.5f - 2.5e-4f) c += (a >= thr[c + 1] ? 1 : 0) - (a < thr[c] ? 1 : 0);
  return xo < 0.f ? -c : c;
}
__device__ __forceinline__ void lion_load_tables(float* deq_tab, float* thr_tab, const float* __restrict__ thr) {
  // exact lion_deq() of every int8 code (the /127 and the 5th power done once) and the codec thresholds, in LDS
  for (int i = threadIdx.x; i < 256; i += blockDim.x) deq_tab[i] = lion_deq(i - 128);
  for (int i = threadIdx.x; i < 129; i += blockDim.x) thr_tab[i] = i < 128 ? thr[i] : __builtin_inff();
  __syncthreads();
}

// Sum of squares in double: every product of two floats is exact in double and the running sum carries ~1e-16 relative
// error, so the float32 norm the optimizer kernels derive from it is the float32 rounding of the true norm - the value
// optax.global_norm rounds to - and the clipped gradients (g / norm) match the host definition bit for bit.  The pass is
// HBM-bound (4 B per parameter); four double FMAs per 16 bytes are far below the fp64 vector rate.
// Workgroups store their double partial sums to `part`; the one that arrives last adds them in workgroup order into *out (no
// float / double atomics: the norm, and with it every clipped gradient, is bitwise reproducible).
// G16: the buffer holds bf16 values (the kernel leaves' gradients [r4]): four per 8 bytes, widened exactly.
template <bool G16>
__global__ void __launch_bounds__(256) sqnorm_kernel(const void* __restrict__ gv, long n, double* __restrict__ out, int* counter,
                                                     double* __restrict__ part) {
  const long nv = n >> 2;
  double d0 = 0.0, d1 = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    float4 v;
    if (G16) {
      const uint2 h = reinterpret_cast<const uint2*>(gv)[i];
      v.x = __uint_as_float(h.x << 16); v.y = __uint_as_float(h.x & 0xffff0000u);
      v.z = __uint_as_float(h.y << 16); v.w = __uint_as_float(h.y & 0xffff0000u);
    } else {
      v = reinterpret_cast<const float4*>(gv)[i];
    }
    d0 = fma((double)v.x, (double)v.x, d0);
    d1 = fma((double)v.y, (double)v.y, d1);
    d0 = fma((double)v.z, (double)v.z, d0);
    d1 = fma((double)v.w, (double)v.w, d1);
  }
  double dacc = d0 + d1;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = G16 ? bf2f(reinterpret_cast<const bf16_t*>(gv)[(nv << 2) + threadIdx.x]) : reinterpret_cast<const float*>(gv)[(nv << 2) + threadIdx.x];
    dacc += (double)v * (double)v;
  }
  for (int o = 32; o > 0; o >>= 1) dacc += __shfl_xor(dacc, o, 64);
  __shared__ double dsc[16];
  __shared__ int s_last;
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) dsc[w] = dacc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += dsc[i];
    sdt_store_wt(part + blockIdx.x, t);
  }
  if (!sdt_arrive_last<true>(counter, (int)gridDim.x, &s_last)) return;
  // last arriver: 256 threads x strided partials, then a fixed tree
  double t = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += sdt_load_wt(part + i);
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) dsc[w] = t;
  __syncthreads();
  if (threadIdx.x == 0) *out += (dsc[0] + dsc[1]) + (dsc[2] + dsc[3]);
}

// *out += sum of n doubles, added in a fixed order (contiguous chunk per workgroup, strided over the threads, the sqnorm_kernel's
// tree): the squared-norm partials the weight-gradient kernels wrote to their slots (include/sdt.h sdt_gemm_tn_wgrad sq_slots).
__global__ void __launch_bounds__(256) sum_f64_kernel(const double* __restrict__ x, long n, double* __restrict__ out, int* counter,
                                                      double* __restrict__ part) {
  const long per = (n + gridDim.x - 1) / gridDim.x;
  const long lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
  double d0 = 0.0, d1 = 0.0;
  long i = lo + threadIdx.x;
  for (; i + 256 < hi; i += 512) {
    d0 += x[i];
    d1 += x[i + 256];
  }
  if (i < hi) d0 += x[i];
  double dacc = d0 + d1;
  for (int o = 32; o > 0; o >>= 1) dacc += __shfl_xor(dacc, o, 64);
  __shared__ double dsc[16];
  __shared__ int s_last;
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) dsc[w] = dacc;
  __syncthreads();
  if (threadIdx.x == 0) sdt_store_wt(part + blockIdx.x, (dsc[0] + dsc[1]) + (dsc[2] + dsc[3]));
  if (!sdt_arrive_last<true>(counter, (int)gridDim.x, &s_last)) return;
  double t = 0.0;
  for (int j = threadIdx.x; j < (int)gridDim.x; j += 256) t += sdt_load_wt(part + j);
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) dsc[w] = t;
  __syncthreads();
  if (threadIdx.x == 0) *out += (dsc[0] + dsc[1]) + (dsc[2] + dsc[3]);
}

// clip factor semantics of optax.clip_by_global_norm: g if norm < max else (g / norm) * max
__device__ __forceinline__ float clip_grad(float g, float gnorm, float max_norm, bool do_clip) {
  return do_clip ? (g / gnorm) * max_norm : g;
}

// LPB lanes cooperate on one quantisation block of BS = 4*LPB elements; each lane owns a float4.
#define LION_SLICES 4
template <int LPB, bool G16>
__global__ void __launch_bounds__(256) lion8_kernel(float* __restrict__ p, const void* __restrict__ g,
                                                    int8_t* __restrict__ codes, float* __restrict__ inv_scale,
                                                    float* __restrict__ ema, bf16_t* __restrict__ w_bf16, long n4,
                                                    const double* __restrict__ sqnorm, const float* __restrict__ thr,
                                                    float max_norm, float neg_lr, float wd, float c1, float c1m, float c2,
                                                    float c2m, float ema_r, float ema_rm) {
  __shared__ float deq_tab[256];
  __shared__ float thr_tab[132];
  lion_load_tables(deq_tab, thr_tab, thr);
  float gnorm = 0.f;
  bool do_clip = false;
  if (sqnorm) {
    gnorm = (float)sqrt(*sqnorm);
    do_clip = !(gnorm < max_norm);
  }
  // n4 is a multiple of LPB and consecutive lanes hold consecutive float4s, so the LPB lanes of a block stay together.
  // A workgroup sweeps LION_SLICES consecutive slices of 256 float4s and the grid covers the buffer once: the resident
  // workgroups then work on ONE contiguous window of each of the seven streams (a capped grid striding over the whole buffers
  // ran the sweep at 4.1 instead of 5.9 TB/s), and every byte is touched once per step, so all of it moves non-temporally.
  typedef float f4v __attribute__((ext_vector_type(4)));
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  const long i_end = min(n4, ((long)blockIdx.x + 1) * (LION_SLICES * 256));
  for (long i = (long)blockIdx.x * (LION_SLICES * 256) + threadIdx.x; i < i_end; i += 256) {
    f4v gv;
    if (G16) {  // bf16 gradient (8 bytes per float4 of parameters), widened exactly: what optax sees of a bf16 cotangent
      const u2v h = __builtin_nontemporal_load(&reinterpret_cast<const u2v*>(g)[i]);
      gv.x = __uint_as_float(h.x << 16); gv.y = __uint_as_float(h.x & 0xffff0000u);
      gv.z = __uint_as_float(h.y << 16); gv.w = __uint_as_float(h.y & 0xffff0000u);
    } else {
      gv = __builtin_nontemporal_load(&reinterpret_cast<const f4v*>(g)[i]);
    }
    const f4v pv = __builtin_nontemporal_load(&reinterpret_cast<const f4v*>(p)[i]);
    const unsigned cw = __builtin_nontemporal_load(&reinterpret_cast<const unsigned*>(codes)[i]);
    const long blk = i / LPB;
    const float inv = inv_scale[blk];
    float gg[4] = {gv.x, gv.y, gv.z, gv.w};
    float pp[4] = {pv.x, pv.y, pv.z, pv.w};
    float mn[4];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int c = (int)(int8_t)((cw >> (8 * j)) & 0xff);
      float mf = deq_tab[c + 128] / inv;                    // lion_quant.py:88-91
      float gc = clip_grad(gg[j], gnorm, max_norm, do_clip);
      float cc = c1m * gc + c1 * mf;                        // lion_quant.py:141-143
      float u = (cc > 0.f) ? 1.f : ((cc < 0.f) ? -1.f : 0.f);
      mn[j] = c2m * gc + c2 * mf;                           // lion_quant.py:105-107
      amax = fmaxf(amax, fabsf(mn[j]));
      if (wd != 0.f) u = u + wd * pp[j];                    // add_decayed_weights
      u = neg_lr * u;                                       // _scale_by_learning_rate
      pp[j] = pp[j] + u;                                    // apply_updates
    }
#pragma unroll
    for (int o = 1; o < LPB; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float ninv = 1.0f / ((amax <= 0.f) ? 1.0f : amax);  // lion_quant.py:72-76
    unsigned ncw = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int q = lion_quant_tab(mn[j] * ninv, thr_tab);
      ncw |= ((unsigned)(q & 0xff)) << (8 * j);
    }
    __builtin_nontemporal_store(ncw, &reinterpret_cast<unsigned*>(codes)[i]);
    if ((i % LPB) == 0) inv_scale[blk] = ninv;
    {
      const f4v o = {pp[0], pp[1], pp[2], pp[3]};
      __builtin_nontemporal_store(o, &reinterpret_cast<f4v*>(p)[i]);
    }
    if (ema) {
      f4v ev = __builtin_nontemporal_load(&reinterpret_cast<const f4v*>(ema)[i]);
      ev.x = ema_r * ev.x + ema_rm * pp[0];
      ev.y = ema_r * ev.y + ema_rm * pp[1];
      ev.z = ema_r * ev.z + ema_rm * pp[2];
      ev.w = ema_r * ev.w + ema_rm * pp[3];
      __builtin_nontemporal_store(ev, &reinterpret_cast<f4v*>(ema)[i]);
    }
    if (w_bf16) {
      const u2v o = {pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3])};
      __builtin_nontemporal_store(o, &reinterpret_cast<u2v*>(w_bf16)[i]);
    }
  }
}

__global__ void __launch_bounds__(256) lion32_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ mom, float* __restrict__ ema,
                                                     bf16_t* __restrict__ w_bf16, long n,
                                                     const double* __restrict__ sqnorm, float max_norm, float neg_lr,
                                                     float wd, float c1, float c1m, float c2, float c2m, float ema_r,
                                                     float ema_rm) {
  float gnorm = 0.f;
  bool do_clip = false;
  if (sqnorm) {
    gnorm = (float)sqrt(*sqnorm);
    do_clip = !(gnorm < max_norm);
  }
  const long i_end = min(n, ((long)blockIdx.x + 1) * 1024);  // a contiguous 1024-element slice per workgroup (see lion8_kernel)
  for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < i_end; i += 256) {
    float gc = clip_grad(g[i], gnorm, max_norm, do_clip);
    float mf = mom[i];
    float pv = p[i];
    float cc = c1m * gc + c1 * mf;
    float u = (cc > 0.f) ? 1.f : ((cc < 0.f) ? -1.f : 0.f);
    mom[i] = c2m * gc + c2 * mf;
    if (wd != 0.f) u = u + wd * pv;
    u = neg_lr * u;
    pv = pv + u;
    p[i] = pv;
    if (ema) ema[i] = ema_r * ema[i] + ema_rm * pv;
    if (w_bf16) w_bf16[i] = f2bf(pv);
  }
}

__global__ void __launch_bounds__(256) lion8_quantize_kernel(const float* __restrict__ x, int8_t* __restrict__ codes,
                                                             float* __restrict__ inv_scale, long nblocks, int bs,
                                                             const float* __restrict__ thr) {
  __shared__ float deq_tab[256];
  __shared__ float thr_tab[132];
  lion_load_tables(deq_tab, thr_tab, thr);
  for (long b = (long)blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += (long)gridDim.x * blockDim.x) {
    const float* xb = x + b * bs;
    float amax = 0.f;
    for (int j = 0; j < bs; ++j) amax = fmaxf(amax, fabsf(xb[j]));
    float inv = 1.0f / ((amax <= 0.f) ? 1.0f : amax);
    for (int j = 0; j < bs; ++j) codes[b * bs + j] = (int8_t)lion_quant_tab(xb[j] * inv, thr_tab);
    inv_scale[b] = inv;
  }
}

__global__ void __launch_bounds__(256) lion8_dequantize_kernel(const int8_t* __restrict__ codes,
                                                               const float* __restrict__ inv_scale,
                                                               float* __restrict__ x, long n, int bs) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    x[i] = lion_deq((int)codes[i]) / inv_scale[i / bs];
}

extern "C" {

#define SQNORM_MAX_BLOCKS 2048
int64_t sdt_sqnorm_workspace_bytes(void) { return SDT_WS_COUNTER_BYTES + SQNORM_MAX_BLOCKS * (int64_t)sizeof(double); }

/* *out_sq += sum g^2 (double).  workspace: sdt_sqnorm_workspace_bytes() bytes under the split-workspace contract (first 64 KiB
 * zero when enqueued, zero again afterwards; include/sdt.h). */
int sdt_sqnorm_accumulate(const float* g, int64_t n, double* out_sq, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  SDT_CHECK_ARG(g && out_sq && n >= 0, "sdt_sqnorm_accumulate: null pointer or negative n");
  SDT_CHECK_ARG(((uintptr_t)g & 15) == 0, "sdt_sqnorm_accumulate: g must be 16-byte aligned");
  SDT_CHECK_ARG(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= sdt_sqnorm_workspace_bytes(),
                "sdt_sqnorm_accumulate: workspace of sdt_sqnorm_workspace_bytes() needed");
  if (n == 0) return SDT_OK;
  hipLaunchKernelGGL(sqnorm_kernel<false>, dim3(sdt_grid_1d(n >> 2, 256 * 8, SQNORM_MAX_BLOCKS)), dim3(256), 0, stream, (const void*)g, (long)n, out_sq,
                     reinterpret_cast<int*>(workspace), reinterpret_cast<double*>((unsigned char*)workspace + SDT_WS_COUNTER_BYTES));
  SDT_LAUNCH_CHECK("sdt_sqnorm_accumulate");
  return SDT_OK;
}

int sdt_sqnorm_accumulate_bf16(const uint16_t* g, int64_t n, double* out_sq, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  SDT_CHECK_ARG(g && out_sq && n >= 0, "sdt_sqnorm_accumulate_bf16: null pointer or negative n");
  SDT_CHECK_ARG(((uintptr_t)g & 7) == 0, "sdt_sqnorm_accumulate_bf16: g must be 8-byte aligned");
  SDT_CHECK_ARG(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= sdt_sqnorm_workspace_bytes(),
                "sdt_sqnorm_accumulate_bf16: workspace of sdt_sqnorm_workspace_bytes() needed");
  if (n == 0) return SDT_OK;
  hipLaunchKernelGGL(sqnorm_kernel<true>, dim3(sdt_grid_1d(n >> 2, 256 * 8, SQNORM_MAX_BLOCKS)), dim3(256), 0, stream, (const void*)g, (long)n, out_sq,
                     reinterpret_cast<int*>(workspace), reinterpret_cast<double*>((unsigned char*)workspace + SDT_WS_COUNTER_BYTES));
  SDT_LAUNCH_CHECK("sdt_sqnorm_accumulate_bf16");
  return SDT_OK;
}

int sdt_sum_f64_accumulate(const double* x, int64_t n, double* out, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  SDT_CHECK_ARG(x && out && n >= 0, "sdt_sum_f64_accumulate: null pointer or negative n");
  SDT_CHECK_ARG(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= sdt_sqnorm_workspace_bytes(),
                "sdt_sum_f64_accumulate: workspace of sdt_sqnorm_workspace_bytes() needed");
  if (n == 0) return SDT_OK;
  hipLaunchKernelGGL(sum_f64_kernel, dim3(sdt_grid_1d(n, 256 * 16, SQNORM_MAX_BLOCKS)), dim3(256), 0, stream, x, (long)n, out,
                     reinterpret_cast<int*>(workspace), reinterpret_cast<double*>((unsigned char*)workspace + SDT_WS_COUNTER_BYTES));
  SDT_LAUNCH_CHECK("sdt_sum_f64_accumulate");
  return SDT_OK;
}

int sdt_lion8_step(float* p, const void* g, int g_bf16, int8_t* codes, float* inv_scale, float* ema, uint16_t* w_bf16, int64_t n,
                   int block_size, const double* sqnorm, const float* thresholds, double max_norm, double lr, double wd,
                   double b1, double b2, double ema_rate, hipStream_t stream) {
  SDT_CHECK_ARG(p && g && codes && inv_scale && thresholds, "sdt_lion8_step: null pointer");
  SDT_CHECK_ARG(n >= 0 && block_size >= 4 && block_size <= 256 && (block_size & (block_size - 1)) == 0,
                "sdt_lion8_step: block_size must be a power of two in [4,256] (got %d)", block_size);
  SDT_CHECK_ARG(n % block_size == 0, "sdt_lion8_step: n=%ld not a multiple of block_size=%d (lion_quant.py:70 reshape)",
                (long)n, block_size);
  SDT_CHECK_ARG((((uintptr_t)p | (uintptr_t)ema) & 15) == 0 && ((uintptr_t)g & (g_bf16 ? 7 : 15)) == 0 && ((uintptr_t)codes & 3) == 0 &&
                    ((uintptr_t)w_bf16 & 7) == 0,
                "sdt_lion8_step: misaligned buffer");
  if (n == 0) return SDT_OK;
  const long n4 = n >> 2;
  const int lpb = block_size >> 2;
  const float c1 = (float)b1, c1m = (float)(1.0 - b1), c2 = (float)b2, c2m = (float)(1.0 - b2);
  const float er = (float)ema_rate, erm = (float)(1.0 - ema_rate);
  dim3 grid(sdt_grid_1d(n4, 256 * LION_SLICES, 1 << 30)), block(256);
#define LAUNCH_L8B(L, H)                                                                                             \
  hipLaunchKernelGGL((lion8_kernel<L, H>), grid, block, 0, stream, p, g, codes, inv_scale, ema, (bf16_t*)w_bf16, \
                     n4, sqnorm, thresholds, (float)max_norm, (float)(-lr), (float)wd, c1, c1m, c2, c2m, er, erm)
#define LAUNCH_L8(L)      \
  do {                    \
    if (g_bf16)           \
      LAUNCH_L8B(L, true); \
    else                  \
      LAUNCH_L8B(L, false); \
  } while (0)
  switch (lpb) {
    case 1: LAUNCH_L8(1); break;
    case 2: LAUNCH_L8(2); break;
    case 4: LAUNCH_L8(4); break;
    case 8: LAUNCH_L8(8); break;
    case 16: LAUNCH_L8(16); break;
    case 32: LAUNCH_L8(32); break;
    default: LAUNCH_L8(64); break;
  }
#undef LAUNCH_L8
#undef LAUNCH_L8B
  SDT_LAUNCH_CHECK("sdt_lion8_step");
  return SDT_OK;
}

int sdt_lion32_step(float* p, const float* g, float* mom, float* ema, uint16_t* w_bf16, int64_t n,
                    const double* sqnorm, double max_norm, double lr, double wd, double b1, double b2, double ema_rate,
                    hipStream_t stream) {
  SDT_CHECK_ARG(p && g && mom && n >= 0, "sdt_lion32_step: null pointer or negative n");
  if (n == 0) return SDT_OK;
  const float c1 = (float)b1, c1m = (float)(1.0 - b1), c2 = (float)b2, c2m = (float)(1.0 - b2);
  const float er = (float)ema_rate, erm = (float)(1.0 - ema_rate);
  hipLaunchKernelGGL(lion32_kernel, dim3(sdt_grid_1d(n, 1024, 1 << 30)), dim3(256), 0, stream, p, g, mom, ema,
                     (bf16_t*)w_bf16, (long)n, sqnorm, (float)max_norm, (float)(-lr), (float)wd, c1, c1m, c2, c2m, er, erm);
  SDT_LAUNCH_CHECK("sdt_lion32_step");
  return SDT_OK;
}

int sdt_lion8_quantize(const float* x, int8_t* codes, float* inv_scale, int64_t n, int block_size,
                       const float* thresholds, hipStream_t stream) {
  SDT_CHECK_ARG(x && codes && inv_scale && thresholds && block_size > 0 && n % block_size == 0, "sdt_lion8_quantize: bad args");
  if (n == 0) return SDT_OK;
  long nb = n / block_size;
  hipLaunchKernelGGL(lion8_quantize_kernel, dim3(sdt_grid_1d(nb, 256, 4096)), dim3(256), 0, stream, x, codes,
                     inv_scale, nb, block_size, thresholds);
  SDT_LAUNCH_CHECK("sdt_lion8_quantize");
  return SDT_OK;
}

int sdt_lion8_dequantize(const int8_t* codes, const float* inv_scale, float* x, int64_t n, int block_size,
                         hipStream_t stream) {
  SDT_CHECK_ARG(x && codes && inv_scale && block_size > 0 && n % block_size == 0, "sdt_lion8_dequantize: bad args");
  if (n == 0) return SDT_OK;
  hipLaunchKernelGGL(lion8_dequantize_kernel, dim3(sdt_grid_1d(n, 256, 4096)), dim3(256), 0, stream, codes, inv_scale,
                     x, (long)n, block_size);
  SDT_LAUNCH_CHECK("sdt_lion8_dequantize");
  return SDT_OK;
}

}  // extern "C"
