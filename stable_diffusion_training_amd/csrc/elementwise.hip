// HBM-bound elementwise / data-movement kernels of the train_step path (all bf16 I/O is
// 16 bytes per lane, fp32 math).  Reference call sites:
//   add_noise / get_velocity   schedulers/scheduling_utils_flax.py:316-343 (via training_utils.py:628-633, 691-696)
//   posterior sample + scale   training_utils.py:582-586
//   MSE (+min-SNR weight)      training_utils.py:546-568, 704-709
//   timestep embedding         diffusers embeddings_flax.get_sinusoidal_embeddings (training_utils.py:678-684)
//   GEGLU / SiLU / quick-GELU / nearest-2x / concat: diffusers attention_flax.py, resnet_flax.py,
//   unet_2d_blocks_flax.py; transformers modeling_flax_clip.py (third-party, restated in oracle/nets.py)
#include "sdt_common.h"

// ------------------------------------------------------------------ noise add (+ velocity target)
// latents/noise: f32 NCHW (B,C,H,W).  Outputs: noisy bf16 NHWC with channels padded to cpad (zeros),
// optional noisy f32 NCHW, optional velocity target f32 NCHW.
__global__ void __launch_bounds__(256) add_noise_kernel(const float* __restrict__ lat, const float* __restrict__ noise,
                                                        const int* __restrict__ t, const float* __restrict__ acp,
                                                        bf16_t* __restrict__ noisy_nhwc, float* __restrict__ noisy_nchw,
                                                        float* __restrict__ vel_nchw, int B, int C, int HW, int cpad) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW), p = (int)(i % HW);
    const float a = acp[t[b]];
    const float sa = sqrtf(a), so = sqrtf(1.0f - a);  // ** 0.5 in the reference
    for (int c = 0; c < cpad; ++c) {
      float nz = 0.f;
      if (c < C) {
        const long off = ((long)b * C + c) * HW + p;
        const float x0 = lat[off], e = noise[off];
        nz = sa * x0 + so * e;
        if (noisy_nchw) noisy_nchw[off] = nz;
        if (vel_nchw) vel_nchw[off] = sa * e - so * x0;
      }
      noisy_nhwc[i * cpad + c] = f2bf(nz);
    }
  }
}

// One sampling step of the reference pipeline (models/pipeline_flax_stable_diffusion.py:222-232) in one launch:
// classifier-free guidance  m = un + g*(tx - un)  over the doubled UNet batch (pred rows [0,B) unconditional, [B,2B) text),
// the eta = 0 DDIM update of diffusers' scheduling_ddim_flax.py step(), and the doubled bf16 NHWC UNet input of the next
// step.  lat f32 NCHW (B,C,h,w) in place; pred / x_next bf16 NHWC (2B,h,w,cpad), padding channels written as zero.
// ptype: 0 epsilon, 1 sample, 2 v_prediction.
__global__ void __launch_bounds__(256) ddim_cfg_step_kernel(const bf16_t* __restrict__ pred, float* __restrict__ lat,
                                                            bf16_t* __restrict__ x_next, int B, int C, int HW, int cpad,
                                                            float guidance, float sa, float sb, float sa_prev, float sb_prev,
                                                            int ptype) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW), p = (int)(i % HW);
    for (int c = 0; c < cpad; ++c) {
      float nx = 0.f;
      if (c < C) {
        const long off = ((long)b * C + c) * HW + p;
        const float un = bf2f(pred[i * cpad + c]), tx = bf2f(pred[(i + total) * cpad + c]);
        const float m = un + guidance * (tx - un);
        const float x = lat[off];
        float x0, eps;
        if (ptype == 0) { x0 = (x - sb * m) / sa; eps = m; }
        else if (ptype == 1) { x0 = m; eps = (x - sa * x0) / sb; }
        else { x0 = sa * x - sb * m; eps = sa * m + sb * x; }
        nx = sa_prev * x0 + sb_prev * eps;
        lat[off] = nx;
      }
      const bf16_t v = f2bf(nx);
      x_next[i * cpad + c] = v;
      x_next[(i + total) * cpad + c] = v;
    }
  }
}

// moments bf16 NHWC (B,h,w,mstride) with mean = ch [0,L), logvar = ch [L,2L); eps f32 NHWC (B,h,w,L)
// -> latents f32 NCHW (B,L,h,w) = (mean + exp(0.5*clip(logvar,-30,20))*eps) * scale
__global__ void __launch_bounds__(256) posterior_sample_kernel(const bf16_t* __restrict__ mom, const float* __restrict__ eps,
                                                               float* __restrict__ lat, int B, int L, int HW, int mstride,
                                                               float scale) {
  const long total = (long)B * HW * L;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % L);
    const long bp = i / L;
    const int b = (int)(bp / HW), p = (int)(bp % HW);
    const float mean = bf2f(mom[bp * mstride + c]);
    float lv = bf2f(mom[bp * mstride + L + c]);
    lv = fminf(fmaxf(lv, -30.f), 20.f);
    const float v = (mean + __expf(0.5f * lv) * eps[i]) * scale;
    lat[((long)b * L + c) * HW + p] = v;
  }
}

// pred bf16 NHWC (B,h,w,cpad), target f32 NCHW (B,C,h,w), w f32 (B) or null.
// loss_sum += sum w*(t-p)^2 * inv_count ; dpred (bf16 NHWC cpad) = -2*w*(t-p)*inv_count
__global__ void __launch_bounds__(256) mse_kernel(const bf16_t* __restrict__ pred, const float* __restrict__ target,
                                                  const float* __restrict__ w, float* __restrict__ loss,
                                                  bf16_t* __restrict__ dpred, int B, int C, int HW, int cpad,
                                                  float inv_count, int* counter, float* __restrict__ part) {
  __shared__ float scratch[16];
  __shared__ int s_last;
  const long total = (long)B * HW;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW), p = (int)(i % HW);
    const float wt = w ? w[b] : 1.f;
    for (int c = 0; c < cpad; ++c) {
      float d = 0.f;
      if (c < C) {
        const float tv = target[((long)b * C + c) * HW + p];
        const float diff = tv - bf2f(pred[i * cpad + c]);
        acc += wt * diff * diff;
        d = -2.f * wt * diff * inv_count;
      }
      if (dpred) dpred[i * cpad + c] = f2bf(d);
    }
  }
  float s = block_sum(acc, scratch);
  if (threadIdx.x == 0) sdt_store_wt(part + blockIdx.x, s);
  // the workgroup that arrives last adds the per-workgroup sums in workgroup order (no float atomics: reproducible loss)
  if (!sdt_arrive_last<true>(counter, (int)gridDim.x, &s_last)) return;
  float t = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += sdt_load_wt(part + i);
  t = block_sum(t, scratch);
  if (threadIdx.x == 0) *loss += t * inv_count;
}

// get_sinusoidal_embeddings: out[b] = [cos(t*inv_i) | sin(t*inv_i)] (flip) or [sin|cos]
__global__ void timestep_embed_kernel(const int* __restrict__ t, bf16_t* __restrict__ out, int B, int dim,
                                      int flip_sin_to_cos, float freq_shift, float max_period) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i % half;
  const float inc = logf(max_period) / ((float)half - freq_shift);
  const float e = (float)t[b] * expf((float)k * -inc);
  const float s = sinf(e), c = cosf(e);
  out[(long)b * dim + k] = f2bf(flip_sin_to_cos ? c : s);
  out[(long)b * dim + half + k] = f2bf(flip_sin_to_cos ? s : c);
}

// ------------------------------------------------------------------ activations (vectors of 8 bf16)
enum { ACT_SILU = 0, ACT_QUICK_GELU = 1, ACT_GELU_ERF = 2 };

template <int ACT>
__device__ __forceinline__ float act_fwd(float x) {
  if (ACT == ACT_SILU) return siluf_(x);
  if (ACT == ACT_QUICK_GELU) return x * sigmoidf_(1.702f * x);
  return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
}
template <int ACT>
__device__ __forceinline__ float act_bwd(float x, float dy) {
  if (ACT == ACT_SILU) {
    float s = sigmoidf_(x);
    return dy * s * (1.f + x * (1.f - s));
  }
  if (ACT == ACT_QUICK_GELU) {
    float s = sigmoidf_(1.702f * x);
    return dy * s * (1.f + 1.702f * x * (1.f - s));
  }
  float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return dy * (cdf + x * pdf);
}

template <int ACT>
__global__ void __launch_bounds__(256) act_fwd_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, long nv) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    float f[8];
    unpack8(x[i], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = act_fwd<ACT>(f[j]);
    y[i] = pack8(f);
  }
}
template <int ACT>
__global__ void __launch_bounds__(256) act_bwd_kernel(const uint4* __restrict__ x, const uint4* __restrict__ dy,
                                                      uint4* __restrict__ dx, long nv) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    float f[8], g[8];
    unpack8(x[i], f);
    unpack8(dy[i], g);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = act_bwd<ACT>(f[j], g[j]);
    dx[i] = pack8(f);
  }
}

// GEGLU: h (M, 2F) -> out (M, F) = h[:, :F] * gelu_tanh(h[:, F:])   (flax nn.gelu approximate=True)
// (gelu_tanh_f / gelu_tanh_grad: sdt_common.h - the fused feed-forward epilogues of gemm.hip evaluate the same functions)
__global__ void __launch_bounds__(256) geglu_fwd_kernel(const uint4* __restrict__ h, uint4* __restrict__ out, long M,
                                                        int Fv) {
  const long total = M * Fv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / Fv;
    const int c = (int)(i % Fv);
    float a[8], g[8];
    unpack8(h[m * 2 * Fv + c], a);
    unpack8(h[m * 2 * Fv + Fv + c], g);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = a[j] * gelu_tanh_f(g[j]);
    out[i] = pack8(a);
  }
}
__global__ void __launch_bounds__(256) geglu_bwd_kernel(const uint4* __restrict__ h, const uint4* __restrict__ dout,
                                                        uint4* __restrict__ dh, long M, int Fv) {
  const long total = M * Fv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / Fv;
    const int c = (int)(i % Fv);
    float a[8], g[8], d[8], da[8], dg[8];
    unpack8(h[m * 2 * Fv + c], a);
    unpack8(h[m * 2 * Fv + Fv + c], g);
    unpack8(dout[i], d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      da[j] = d[j] * gelu_tanh_f(g[j]);
      dg[j] = d[j] * a[j] * gelu_tanh_grad(g[j]);
    }
    dh[m * 2 * Fv + c] = pack8(da);
    dh[m * 2 * Fv + Fv + c] = pack8(dg);
  }
}

// ------------------------------------------------------------------ data movement
// 2-D strided copy of bf16 rows (concat / split along channels): cols % 8 == 0, strides % 8 == 0
__global__ void __launch_bounds__(256) copy2d_kernel(uint4* __restrict__ dst, long dstride_v, const uint4* __restrict__ src,
                                                     long sstride_v, long rows, int cols_v) {
  const long total = rows * cols_v;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols_v;
    const int c = (int)(i % cols_v);
    dst[r * dstride_v + c] = src[r * sstride_v + c];
  }
}

// n column segments of a wide matrix <-> n narrow matrices, one launch (blockIdx.y = segment): channel concat / split and the
// gradient gather of column slices used to be one copy2d launch per segment
struct ColSeg {
  uint4* part[32];     // narrow matrix of segment k (nullptr when gathering: the segment is zero-filled)
  long ld_v[32];       // its row pitch (16-byte vectors)
  int col0_v[32];      // first vector column of the segment inside the wide matrix
  int cols_v[32];      // vectors per row of the segment
};
__global__ void __launch_bounds__(256) copy_cols_kernel(uint4* __restrict__ wide, long ld_wide_v, const ColSeg seg, long rows, int to_wide) {
  const int k = blockIdx.y;
  const int cv = seg.cols_v[k];
  uint4* part = seg.part[k];
  const long ldp = seg.ld_v[k];
  uint4* w0 = wide + seg.col0_v[k];
  const long total = rows * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cv;
    const int c = (int)(i - r * cv);
    if (to_wide) w0[r * ld_wide_v + c] = part ? part[r * ldp + c] : make_uint4(0, 0, 0, 0);
    else part[r * ldp + c] = w0[r * ld_wide_v + c];
  }
}

__global__ void __launch_bounds__(256) add_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                  uint4* __restrict__ y, long nv) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    float f[8], g[8];
    unpack8(a[i], f);
    unpack8(b[i], g);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] += g[j];
    y[i] = pack8(f);
  }
}

// nearest 2x upsample NHWC: out (B,2H,2W,C) ; backward: din (B,H,W,C) = sum of the 2x2 outputs
__global__ void __launch_bounds__(256) upsample2x_fwd_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int B,
                                                             int H, int W, int Cv) {
  const long total = (long)B * 2 * H * 2 * W * Cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cv);
    long r = i / Cv;
    const int ox = (int)(r % (2 * W));
    r /= (2 * W);
    const int oy = (int)(r % (2 * H));
    const int b = (int)(r / (2 * H));
    y[i] = x[(((long)b * H + (oy >> 1)) * W + (ox >> 1)) * Cv + c];
  }
}
__global__ void __launch_bounds__(256) upsample2x_bwd_kernel(const uint4* __restrict__ dy, uint4* __restrict__ dx, int B,
                                                             int H, int W, int Cv) {
  const long total = (long)B * H * W * Cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cv);
    long r = i / Cv;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, f[8];
#pragma unroll
    for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
      for (int dxx = 0; dxx < 2; ++dxx) {
        unpack8(dy[(((long)b * 2 * H + 2 * y + dyy) * 2 * W + 2 * x + dxx) * Cv + c], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += f[j];
      }
    dx[i] = pack8(acc);
  }
}

// f32 NCHW (B,C,H,W) -> bf16 NHWC (B,H,W,cpad) zero padded
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int B,
                                                           int C, int HW, int cpad) {
  const long total = (long)B * HW * cpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpad);
    const long bp = i / cpad;
    const int b = (int)(bp / HW), p = (int)(bp % HW);
    y[i] = (c < C) ? f2bf(x[((long)b * C + c) * HW + p]) : (bf16_t)0;
  }
}
// bf16 NHWC (B,H,W,cpad) -> f32 NCHW (B,C,H,W)
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int B,
                                                           int C, int HW, int cpad) {
  const long total = (long)B * C * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const long bc = i / HW;
    const int b = (int)(bc / C), c = (int)(bc % C);
    y[i] = bf2f(x[((long)b * HW + p) * cpad + c]);
  }
}

__global__ void __launch_bounds__(256) cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = f2bf(x[i]);
}

// batched bf16 transpose (batch, R, C) -> (batch, C, R) through a 64x64 LDS tile
__global__ void __launch_bounds__(256) transpose_bf16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int R,
                                                             int C) {
  __shared__ bf16_t tile[64][66];
  const long base = (long)blockIdx.z * R * C;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? x[base + (long)(r0 + r) * C + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < R && c0 + c < C) y[base + (long)(c0 + c) * R + r0 + r] = tile[r][c];
  }
}

// Parameter preparation: one launch per model.  For every matrix-shaped leaf ([batch][R][C] in Flax layout: Dense batch=1
// R=in C=out; conv HWIO batch=kh*kw R=Cin C=Cout) produce the bf16 compute copy W (same layout, zero-padded to Rp x Cp) from
// the fp32 master.  Every contraction reads W as it stands (forward: k-major operand through transposing LDS reads; input
// gradient: row-major operand), so no transposed copy exists; Wt != NULL additionally writes [batch][Cp][Rp] (kept for
// callers that want it).  Per step only the few zero-padded leaves pass through here: the optimizer sweep itself mirrors
// the master into W (sdt_lion8_step / sdt_lion32_step w_bf16) for all the others.
struct SdtPrepDesc {
  long src_off, w_off, wt_off;  // element offsets into master / W / Wt flat buffers
  int batch, R, C, Rp, Cp;      // logical and padded dims
  int tile0;                    // first 64x64 tile index of this leaf in the launch
  int flags;                    // reserved (0)
};
__global__ void __launch_bounds__(256) param_prepare_kernel(const float* __restrict__ master, bf16_t* __restrict__ W,
                                                            bf16_t* __restrict__ Wt, const SdtPrepDesc* __restrict__ descs,
                                                            int ndesc) {
  __shared__ bf16_t tile[64][66];
  const int tile_id = blockIdx.x;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {  // last desc with tile0 <= tile_id
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile0 <= tile_id) lo = mid; else hi = mid - 1;
  }
  const SdtPrepDesc d = descs[lo];
  int tl = tile_id - d.tile0;
  const int tc = (d.Cp + 63) >> 6, tr = (d.Rp + 63) >> 6;
  const int bz = tl / (tc * tr);
  tl -= bz * tc * tr;
  const int r0 = (tl / tc) * 64, c0 = (tl % tc) * 64;
  const float* src = master + d.src_off + (long)bz * d.R * d.C;
  bf16_t* w = W + d.w_off + (long)bz * d.Rp * d.Cp;
  bf16_t* wt = Wt ? Wt + d.wt_off + (long)bz * d.Rp * d.Cp : nullptr;
  // interior tiles of leaves whose dims are multiples of 4 (every large kernel): 16-byte loads, 8-byte stores
  const bool vec = r0 + 64 <= d.R && c0 + 64 <= d.C && (d.C & 3) == 0 && (d.Cp & 3) == 0 && (d.Rp & 3) == 0 && (d.R & 3) == 0 &&
                   ((d.src_off | d.w_off | d.wt_off) & 3) == 0;
  if (vec) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int r = ps * 16 + (threadIdx.x >> 4), c = (threadIdx.x & 15) * 4;
      const float4 f = *reinterpret_cast<const float4*>(src + (long)(r0 + r) * d.C + c0 + c);
      uint2 pk;
      pk.x = pack2bf(f.x, f.y);
      pk.y = pack2bf(f.z, f.w);
      *reinterpret_cast<uint2*>(w + (long)(r0 + r) * d.Cp + c0 + c) = pk;
      tile[r][c] = (bf16_t)(pk.x & 0xffffu); tile[r][c + 1] = (bf16_t)(pk.x >> 16);
      tile[r][c + 2] = (bf16_t)(pk.y & 0xffffu); tile[r][c + 3] = (bf16_t)(pk.y >> 16);
    }
    if (!wt) return;  // (wave-uniform: a kernel argument)
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int c = ps * 16 + (threadIdx.x >> 4), r = (threadIdx.x & 15) * 4;
      uint2 pk;
      pk.x = (unsigned)(unsigned short)tile[r][c] | ((unsigned)(unsigned short)tile[r + 1][c] << 16);
      pk.y = (unsigned)(unsigned short)tile[r + 2][c] | ((unsigned)(unsigned short)tile[r + 3][c] << 16);
      *reinterpret_cast<uint2*>(wt + (long)(c0 + c) * d.Rp + r0 + r) = pk;
    }
    return;
  }
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    bf16_t v = 0;
    if (r0 + r < d.R && c0 + c < d.C) v = f2bf(src[(long)(r0 + r) * d.C + c0 + c]);
    tile[r][c] = v;
    if (r0 + r < d.Rp && c0 + c < d.Cp) w[(long)(r0 + r) * d.Cp + c0 + c] = v;
  }
  if (!wt) return;
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < d.Rp && c0 + c < d.Cp) wt[(long)(c0 + c) * d.Rp + r0 + r] = tile[r][c];
  }
}

// CLIP embeddings: out[row] = tok[ids[row]] + pos[row % S]  (bf16 out, fp32 tables)
__global__ void __launch_bounds__(256) embedding_fwd_kernel(const int* __restrict__ ids, const float* __restrict__ tok,
                                                            const float* __restrict__ pos, bf16_t* __restrict__ out,
                                                            long rows, int S, int D) {
  const long total = rows * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D;
    const int c = (int)(i % D);
    out[i] = f2bf(tok[(long)ids[r] * D + c] + pos[(long)(r % S) * D + c]);
  }
}
// Gather form, one writer per element (no atomics): workgroup r < rows owns token row ids[r] IF r is the first row carrying that
// id, and adds the gradients of every row with the same id in row order; workgroup rows + s owns position row s and adds the
// rows r = s, s + S, ... in order.
__global__ void __launch_bounds__(256) embedding_bwd_kernel(const int* __restrict__ ids, const bf16_t* __restrict__ dout,
                                                            float* __restrict__ dtok, float* __restrict__ dpos, long rows,
                                                            int S, int D) {
  const long blk = blockIdx.x;
  if (blk >= rows) {
    const long s0 = blk - rows;
    for (int c = threadIdx.x; c < D; c += 256) {
      float g = 0.f;
      for (long r = s0; r < rows; r += S) g += bf2f(dout[r * D + c]);
      dpos[s0 * D + c] += g;
    }
    return;
  }
  const int id = ids[blk];
  int dup = 0;
  for (long j = threadIdx.x; j < blk; j += 256) dup |= ids[j] == id;
  if (__syncthreads_or(dup)) return;  // an earlier row owns this token
  for (int c = threadIdx.x; c < D; c += 256) {
    float g = 0.f;
    for (long j = blk; j < rows; ++j)
      if (ids[j] == id) g += bf2f(dout[j * D + c]);
    dtok[(long)id * D + c] += g;
  }
}

// column sums: out[b][n] (+)= sum over the rows of batch slice b of dy[m][n].  Workgroup (x, y, z) sums rows_per_block rows of 256
// columns of slice z into its own partial row; the workgroup that arrives last at (x, z) adds the partial rows in y order (no
// float atomics) and writes the result: += into the fp32 db, or rounded into the bf16 out_bf (the per-image row-bias gradient).
__global__ void __launch_bounds__(256) colsum_kernel(const bf16_t* __restrict__ dy, float* __restrict__ db, bf16_t* __restrict__ out_bf,
                                                     long M, int N, int ld, int rows_per_block, long batch_stride_dy,
                                                     int batch_stride_db, int* counters, float* __restrict__ part) {
  dy += (long)blockIdx.z * batch_stride_dy;
  // thread (tx = column-vector of 8, ty = row lane); blockDim = 256 = 32 x 8
  __shared__ float red[8][32][8];
  __shared__ int s_last;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int cv = blockIdx.x * 32 + tx;
  const long m0 = (long)blockIdx.y * rows_per_block;
  const long m1 = (m0 + rows_per_block < M) ? m0 + rows_per_block : M;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (cv * 8 < N) {
    for (long m = m0 + ty; m < m1; m += 8) {
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(dy + m * ld + cv * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ty][tx][j] = acc[j];
  __syncthreads();
  const int group = blockIdx.z * gridDim.x + blockIdx.x;
  float* grp = part + (long)group * gridDim.y * 256;
  {
    const int col = threadIdx.x;  // 256 columns of this workgroup
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += red[k][col >> 3][col & 7];
    sdt_store_wt(grp + (long)blockIdx.y * 256 + col, sum);
  }
  if (!sdt_arrive_last<true>(counters + group, (int)gridDim.y, &s_last)) return;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < N) {
    float t = 0.f;
    for (int y = 0; y < (int)gridDim.y; ++y) t += sdt_load_wt(grp + (long)y * 256 + threadIdx.x);
    if (out_bf) out_bf[(long)blockIdx.z * batch_stride_db + n] = f2bf(t);
    else db[(long)blockIdx.z * batch_stride_db + n] += t;
  }
}

// row bias gradient for the temb broadcast add: dt[b][n] = sum_{m in batch b} dy[m][n]  -> bf16 (B, N)
// (done with colsum per batch slice on the host side)

// in-place row softmax over bf16 rows of length n (fp32 math); one block per row
__global__ void __launch_bounds__(256) softmax_rows_kernel(bf16_t* __restrict__ x, int n, float scale) {
  __shared__ float scratch[16];
  bf16_t* row = x + (long)blockIdx.x * n;
  float mx = -3.0e38f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) mx = fmaxf(mx, bf2f(row[i]) * scale);
  mx = wave_max(mx);
  {
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) scratch[w] = mx;
    __syncthreads();
    mx = scratch[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) mx = fmaxf(mx, scratch[i]);
  }
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += __expf(bf2f(row[i]) * scale - mx);
  s = block_sum(s, scratch);
  const float inv = 1.f / s;
  for (int i = threadIdx.x; i < n; i += blockDim.x) row[i] = f2bf(__expf(bf2f(row[i]) * scale - mx) * inv);
}

// ================================================================== C ABI
// out = sum of n (<= 32) bf16 tensors, accumulated in fp32 in argument order (the autograd engine's chain of
// binary adds rounds to bf16 after every add; one pass keeps fp32 until the end)
struct SumPtrs { const uint4* p[32]; };
__global__ void __launch_bounds__(256) sum_n_kernel(const SumPtrs ptrs, uint4* __restrict__ out, int n, long nvec) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    float acc[8], f[8];
    unpack8(ptrs.p[0][i], acc);
    for (int k = 1; k < n; ++k) {
      unpack8(ptrs.p[k][i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e];
    }
    out[i] = pack8(acc);
  }
}


// Zero a list of float ranges of one buffer in one launch: the gradient leaves that are ACCUMULATED into (norm scales and
// biases, embeddings); weight and bias gradients of Dense / conv layers are written whole by sdt_gemm_tn_wgrad and need none.
// ranges: device int64 pairs (first float4 index, float4 count), each at most ZR_CHUNK float4s (the host splits longer ones).
#define ZR_CHUNK 4096
__global__ void __launch_bounds__(256) zero_ranges_kernel(float4* __restrict__ base, const long* __restrict__ ranges) {
  const long first = ranges[2 * blockIdx.x], count = ranges[2 * blockIdx.x + 1];
  for (long i = threadIdx.x; i < count; i += 256) base[first + i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

extern "C" {

int sdt_add_noise_velocity(const float* latents, const float* noise, const int32_t* timesteps,
                           const float* alphas_cumprod, uint16_t* noisy_nhwc_bf16, float* noisy_nchw,
                           float* velocity_nchw, int B, int C, int H, int W, int cpad, hipStream_t stream) {
  SDT_CHECK_ARG(latents && noise && timesteps && alphas_cumprod && noisy_nhwc_bf16, "sdt_add_noise_velocity: null pointer");
  SDT_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && cpad >= C, "sdt_add_noise_velocity: bad shape B=%d C=%d H=%d W=%d cpad=%d", B, C, H, W, cpad);
  hipLaunchKernelGGL(add_noise_kernel, dim3(sdt_grid_1d((long)B * H * W, 256)), dim3(256), 0, stream, latents, noise,
                     timesteps, alphas_cumprod, (bf16_t*)noisy_nhwc_bf16, noisy_nchw, velocity_nchw, B, C, H * W, cpad);
  SDT_LAUNCH_CHECK("sdt_add_noise_velocity");
  return SDT_OK;
}

int sdt_ddim_cfg_step(const uint16_t* pred_nhwc, float* latents_nchw, uint16_t* next_input_nhwc, int B, int C, int H, int W,
                      int cpad, float guidance_scale, float alpha_prod_t, float alpha_prod_prev, int prediction_type,
                      hipStream_t stream) {
  SDT_CHECK_ARG(pred_nhwc && latents_nchw && next_input_nhwc, "sdt_ddim_cfg_step: null pointer");
  SDT_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && cpad >= C, "sdt_ddim_cfg_step: bad shape B=%d C=%d H=%d W=%d cpad=%d", B, C, H, W, cpad);
  SDT_CHECK_ARG(prediction_type >= 0 && prediction_type <= 2, "sdt_ddim_cfg_step: prediction_type %d (0 epsilon, 1 sample, 2 v_prediction)", prediction_type);
  SDT_CHECK_ARG(alpha_prod_t > 0.f && alpha_prod_t <= 1.f && alpha_prod_prev > 0.f && alpha_prod_prev <= 1.f,
                "sdt_ddim_cfg_step: alpha products must lie in (0, 1]");
  hipLaunchKernelGGL(ddim_cfg_step_kernel, dim3(sdt_grid_1d((long)B * H * W, 256)), dim3(256), 0, stream, (const bf16_t*)pred_nhwc,
                     latents_nchw, (bf16_t*)next_input_nhwc, B, C, H * W, cpad, guidance_scale, sqrtf(alpha_prod_t),
                     sqrtf(1.0f - alpha_prod_t), sqrtf(alpha_prod_prev), sqrtf(1.0f - alpha_prod_prev), prediction_type);
  SDT_LAUNCH_CHECK("sdt_ddim_cfg_step");
  return SDT_OK;
}

int sdt_vae_posterior_sample(const uint16_t* moments_nhwc, const float* eps_nhwc, float* latents_nchw, int B, int L,
                             int H, int W, int moment_stride, float scale, hipStream_t stream) {
  SDT_CHECK_ARG(moments_nhwc && eps_nhwc && latents_nchw && B > 0 && L > 0 && moment_stride >= 2 * L,
                "sdt_vae_posterior_sample: bad args");
  hipLaunchKernelGGL(posterior_sample_kernel, dim3(sdt_grid_1d((long)B * H * W * L, 256)), dim3(256), 0, stream,
                     (const bf16_t*)moments_nhwc, eps_nhwc, latents_nchw, B, L, H * W, moment_stride, scale);
  SDT_LAUNCH_CHECK("sdt_vae_posterior_sample");
  return SDT_OK;
}

#define MSE_MAX_BLOCKS 512
int64_t sdt_reduce_workspace_bytes(void) { return SDT_WS_COUNTER_BYTES + 65536; }

/* workspace: sdt_reduce_workspace_bytes() bytes under the split-workspace contract (first 64 KiB zero when enqueued, zero again
 * afterwards): the per-workgroup partial losses, added in order by the workgroup that arrives last */
int sdt_mse_loss_fwd_bwd(const uint16_t* pred_nhwc, const float* target_nchw, const float* weight, float* loss_accum,
                         uint16_t* dpred_nhwc, int B, int C, int H, int W, int cpad, void* workspace, int64_t workspace_bytes,
                         hipStream_t stream) {
  SDT_CHECK_ARG(pred_nhwc && target_nchw && loss_accum && B > 0 && C > 0 && cpad >= C, "sdt_mse_loss_fwd_bwd: bad args");
  SDT_CHECK_ARG(workspace && workspace_bytes >= sdt_reduce_workspace_bytes(), "sdt_mse_loss_fwd_bwd: workspace of sdt_reduce_workspace_bytes() needed");
  const float inv_count = 1.0f / ((float)B * C * H * W);
  hipLaunchKernelGGL(mse_kernel, dim3(sdt_grid_1d((long)B * H * W, 256, MSE_MAX_BLOCKS)), dim3(256), 0, stream,
                     (const bf16_t*)pred_nhwc, target_nchw, weight, loss_accum, (bf16_t*)dpred_nhwc, B, C, H * W, cpad,
                     inv_count, reinterpret_cast<int*>(workspace), reinterpret_cast<float*>((unsigned char*)workspace + SDT_WS_COUNTER_BYTES));
  SDT_LAUNCH_CHECK("sdt_mse_loss_fwd_bwd");
  return SDT_OK;
}

int sdt_timestep_embedding(const int32_t* timesteps, uint16_t* out, int B, int dim, int flip_sin_to_cos,
                           float freq_shift, hipStream_t stream) {
  SDT_CHECK_ARG(timesteps && out && B > 0 && dim > 0 && dim % 2 == 0, "sdt_timestep_embedding: bad args");
  const int n = B * (dim / 2);
  hipLaunchKernelGGL(timestep_embed_kernel, dim3(sdt_ceil_div(n, 256)), dim3(256), 0, stream, timesteps, (bf16_t*)out, B,
                     dim, flip_sin_to_cos, freq_shift, 10000.0f);
  SDT_LAUNCH_CHECK("sdt_timestep_embedding");
  return SDT_OK;
}

static int check_vec(const void* a, const void* b, const void* c, int64_t n, const char* name) {
  SDT_CHECK_ARG(n >= 0 && n % 8 == 0, "%s: element count %ld must be a multiple of 8", name, (long)n);
  SDT_CHECK_ARG((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0, "%s: pointers must be 16-byte aligned", name);
  return SDT_OK;
}

int sdt_act_fwd(const uint16_t* x, uint16_t* y, int64_t n, int act, hipStream_t stream) {
  SDT_CHECK_ARG(x && y, "sdt_act_fwd: null pointer");
  int rc = check_vec(x, y, nullptr, n, "sdt_act_fwd");
  if (rc) return rc;
  if (n == 0) return SDT_OK;
  dim3 g(sdt_grid_1d(n / 8, 256)), b(256);
  switch (act) {
    case ACT_SILU: hipLaunchKernelGGL(act_fwd_kernel<ACT_SILU>, g, b, 0, stream, (const uint4*)x, (uint4*)y, (long)(n / 8)); break;
    case ACT_QUICK_GELU: hipLaunchKernelGGL(act_fwd_kernel<ACT_QUICK_GELU>, g, b, 0, stream, (const uint4*)x, (uint4*)y, (long)(n / 8)); break;
    case ACT_GELU_ERF: hipLaunchKernelGGL(act_fwd_kernel<ACT_GELU_ERF>, g, b, 0, stream, (const uint4*)x, (uint4*)y, (long)(n / 8)); break;
    default: sdt_set_error("sdt_act_fwd: unknown activation %d", act); return SDT_ERR_INVALID_ARG;
  }
  SDT_LAUNCH_CHECK("sdt_act_fwd");
  return SDT_OK;
}

int sdt_act_bwd(const uint16_t* x, const uint16_t* dy, uint16_t* dx, int64_t n, int act, hipStream_t stream) {
  SDT_CHECK_ARG(x && dy && dx, "sdt_act_bwd: null pointer");
  int rc = check_vec(x, dy, dx, n, "sdt_act_bwd");
  if (rc) return rc;
  if (n == 0) return SDT_OK;
  dim3 g(sdt_grid_1d(n / 8, 256)), b(256);
  switch (act) {
    case ACT_SILU: hipLaunchKernelGGL(act_bwd_kernel<ACT_SILU>, g, b, 0, stream, (const uint4*)x, (const uint4*)dy, (uint4*)dx, (long)(n / 8)); break;
    case ACT_QUICK_GELU: hipLaunchKernelGGL(act_bwd_kernel<ACT_QUICK_GELU>, g, b, 0, stream, (const uint4*)x, (const uint4*)dy, (uint4*)dx, (long)(n / 8)); break;
    case ACT_GELU_ERF: hipLaunchKernelGGL(act_bwd_kernel<ACT_GELU_ERF>, g, b, 0, stream, (const uint4*)x, (const uint4*)dy, (uint4*)dx, (long)(n / 8)); break;
    default: sdt_set_error("sdt_act_bwd: unknown activation %d", act); return SDT_ERR_INVALID_ARG;
  }
  SDT_LAUNCH_CHECK("sdt_act_bwd");
  return SDT_OK;
}

int sdt_geglu_fwd(const uint16_t* h, uint16_t* out, int64_t M, int F, hipStream_t stream) {
  SDT_CHECK_ARG(h && out && M >= 0 && F > 0 && F % 8 == 0, "sdt_geglu_fwd: bad args (F=%d must be a multiple of 8)", F);
  if (M == 0) return SDT_OK;
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(sdt_grid_1d(M * (F / 8), 256)), dim3(256), 0, stream, (const uint4*)h,
                     (uint4*)out, (long)M, F / 8);
  SDT_LAUNCH_CHECK("sdt_geglu_fwd");
  return SDT_OK;
}

int sdt_geglu_bwd(const uint16_t* h, const uint16_t* dout, uint16_t* dh, int64_t M, int F, hipStream_t stream) {
  SDT_CHECK_ARG(h && dout && dh && M >= 0 && F > 0 && F % 8 == 0, "sdt_geglu_bwd: bad args");
  if (M == 0) return SDT_OK;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(sdt_grid_1d(M * (F / 8), 256)), dim3(256), 0, stream, (const uint4*)h,
                     (const uint4*)dout, (uint4*)dh, (long)M, F / 8);
  SDT_LAUNCH_CHECK("sdt_geglu_bwd");
  return SDT_OK;
}

int sdt_copy2d_bf16(uint16_t* dst, int64_t dst_stride, const uint16_t* src, int64_t src_stride, int64_t rows,
                    int cols, hipStream_t stream) {
  SDT_CHECK_ARG(dst && src && rows >= 0 && cols > 0, "sdt_copy2d_bf16: bad args");
  SDT_CHECK_ARG(cols % 8 == 0 && dst_stride % 8 == 0 && src_stride % 8 == 0 &&
                    (((uintptr_t)dst | (uintptr_t)src) & 15) == 0,
                "sdt_copy2d_bf16: cols/strides must be multiples of 8 and pointers 16-byte aligned");
  if (rows == 0) return SDT_OK;
  hipLaunchKernelGGL(copy2d_kernel, dim3(sdt_grid_1d(rows * (cols / 8), 256)), dim3(256), 0, stream, (uint4*)dst,
                     (long)(dst_stride / 8), (const uint4*)src, (long)(src_stride / 8), (long)rows, cols / 8);
  SDT_LAUNCH_CHECK("sdt_copy2d_bf16");
  return SDT_OK;
}

/* n (<= 32) column segments: wide[r][col0_k .. col0_k + cols_k) <-> parts[k][r][0 .. cols_k) (row pitch ld_parts[k]), segments laid
 * side by side in `wide` in index order.  to_wide = 1 gathers (parts[k] == NULL zero-fills the segment), 0 scatters. */
int sdt_copy_cols_bf16(uint16_t* wide, int64_t ld_wide, void* const* parts, const int64_t* ld_parts, const int* cols, int n,
                       int64_t rows, int to_wide, hipStream_t stream) {
  SDT_CHECK_ARG(wide && parts && ld_parts && cols && n > 0 && n <= 32 && rows >= 0, "sdt_copy_cols_bf16: bad args (1..32 segments)");
  SDT_CHECK_ARG(ld_wide % 8 == 0 && ((uintptr_t)wide & 15) == 0, "sdt_copy_cols_bf16: wide matrix must be 16-byte aligned with a pitch that is a multiple of 8");
  ColSeg seg;
  int col0 = 0, maxc = 0;
  for (int k = 0; k < n; ++k) {
    SDT_CHECK_ARG(cols[k] > 0 && cols[k] % 8 == 0 && ld_parts[k] % 8 == 0 && ld_parts[k] >= cols[k] && ((uintptr_t)parts[k] & 15) == 0,
                  "sdt_copy_cols_bf16: segment %d: widths / pitches must be multiples of 8, pointers 16-byte aligned", k);
    SDT_CHECK_ARG(parts[k] || to_wide, "sdt_copy_cols_bf16: segment %d: null destination", k);
    seg.part[k] = (uint4*)parts[k];
    seg.ld_v[k] = ld_parts[k] / 8;
    seg.col0_v[k] = col0 / 8;
    seg.cols_v[k] = cols[k] / 8;
    col0 += cols[k];
    if (cols[k] > maxc) maxc = cols[k];
  }
  SDT_CHECK_ARG(col0 <= ld_wide, "sdt_copy_cols_bf16: segments (%d columns) exceed the wide pitch %ld", col0, (long)ld_wide);
  if (rows == 0) return SDT_OK;
  hipLaunchKernelGGL(copy_cols_kernel, dim3(sdt_grid_1d(rows * (maxc / 8), 256, 4096), n), dim3(256), 0, stream, (uint4*)wide,
                     (long)(ld_wide / 8), seg, (long)rows, to_wide);
  SDT_LAUNCH_CHECK("sdt_copy_cols_bf16");
  return SDT_OK;
}

int sdt_add_bf16(const uint16_t* a, const uint16_t* b, uint16_t* y, int64_t n, hipStream_t stream) {
  SDT_CHECK_ARG(a && b && y, "sdt_add_bf16: null pointer");
  int rc = check_vec(a, b, y, n, "sdt_add_bf16");
  if (rc) return rc;
  if (n == 0) return SDT_OK;
  hipLaunchKernelGGL(add_kernel, dim3(sdt_grid_1d(n / 8, 256)), dim3(256), 0, stream, (const uint4*)a, (const uint4*)b,
                     (uint4*)y, (long)(n / 8));
  SDT_LAUNCH_CHECK("sdt_add_bf16");
  return SDT_OK;
}

int sdt_upsample2x_fwd(const uint16_t* x, uint16_t* y, int B, int H, int W, int C, hipStream_t stream) {
  SDT_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "sdt_upsample2x_fwd: bad args");
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(sdt_grid_1d((long)B * 4 * H * W * (C / 8), 256)), dim3(256), 0, stream,
                     (const uint4*)x, (uint4*)y, B, H, W, C / 8);
  SDT_LAUNCH_CHECK("sdt_upsample2x_fwd");
  return SDT_OK;
}

int sdt_upsample2x_bwd(const uint16_t* dy, uint16_t* dx, int B, int H, int W, int C, hipStream_t stream) {
  SDT_CHECK_ARG(dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "sdt_upsample2x_bwd: bad args");
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(sdt_grid_1d((long)B * H * W * (C / 8), 256)), dim3(256), 0, stream,
                     (const uint4*)dy, (uint4*)dx, B, H, W, C / 8);
  SDT_LAUNCH_CHECK("sdt_upsample2x_bwd");
  return SDT_OK;
}

int sdt_nchw_f32_to_nhwc_bf16(const float* x, uint16_t* y, int B, int C, int H, int W, int cpad, hipStream_t stream) {
  SDT_CHECK_ARG(x && y && B > 0 && C > 0 && cpad >= C, "sdt_nchw_f32_to_nhwc_bf16: bad args");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(sdt_grid_1d((long)B * H * W * cpad, 256)), dim3(256), 0, stream, x,
                     (bf16_t*)y, B, C, H * W, cpad);
  SDT_LAUNCH_CHECK("sdt_nchw_f32_to_nhwc_bf16");
  return SDT_OK;
}

int sdt_nhwc_bf16_to_nchw_f32(const uint16_t* x, float* y, int B, int C, int H, int W, int cpad, hipStream_t stream) {
  SDT_CHECK_ARG(x && y && B > 0 && C > 0 && cpad >= C, "sdt_nhwc_bf16_to_nchw_f32: bad args");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(sdt_grid_1d((long)B * H * W * C, 256)), dim3(256), 0, stream,
                     (const bf16_t*)x, y, B, C, H * W, cpad);
  SDT_LAUNCH_CHECK("sdt_nhwc_bf16_to_nchw_f32");
  return SDT_OK;
}

int sdt_cast_f32_to_bf16(const float* x, uint16_t* y, int64_t n, hipStream_t stream) {
  SDT_CHECK_ARG(x && y && n >= 0, "sdt_cast_f32_to_bf16: bad args");
  if (n == 0) return SDT_OK;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(sdt_grid_1d(n, 256 * 4)), dim3(256), 0, stream, x, (bf16_t*)y, (long)n);
  SDT_LAUNCH_CHECK("sdt_cast_f32_to_bf16");
  return SDT_OK;
}

int sdt_transpose_bf16(const uint16_t* x, uint16_t* y, int batch, int R, int C, hipStream_t stream) {
  SDT_CHECK_ARG(x && y && batch > 0 && R > 0 && C > 0 && batch < 65536, "sdt_transpose_bf16: bad args");
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(sdt_ceil_div(C, 64), sdt_ceil_div(R, 64), batch), dim3(256), 0, stream,
                     (const bf16_t*)x, (bf16_t*)y, R, C);
  SDT_LAUNCH_CHECK("sdt_transpose_bf16");
  return SDT_OK;
}

int sdt_param_prepare(const float* master, uint16_t* w_bf16, uint16_t* wt_bf16, const void* descs_device, int ndesc,
                      int total_tiles, hipStream_t stream) {
  SDT_CHECK_ARG(master && w_bf16 && descs_device && ndesc > 0 && total_tiles > 0, "sdt_param_prepare: bad args");
  hipLaunchKernelGGL(param_prepare_kernel, dim3(total_tiles), dim3(256), 0, stream, master, (bf16_t*)w_bf16,
                     (bf16_t*)wt_bf16, (const SdtPrepDesc*)descs_device, ndesc);
  SDT_LAUNCH_CHECK("sdt_param_prepare");
  return SDT_OK;
}

int sdt_zero_ranges_chunk(void) { return ZR_CHUNK; }

int sdt_zero_ranges(float* base, const int64_t* ranges_device, int nranges, hipStream_t stream) {
  SDT_CHECK_ARG(base && ranges_device && nranges > 0 && ((uintptr_t)base & 15) == 0, "sdt_zero_ranges: bad args");
  hipLaunchKernelGGL(zero_ranges_kernel, dim3(nranges), dim3(256), 0, stream, reinterpret_cast<float4*>(base),
                     reinterpret_cast<const long*>(ranges_device));
  SDT_LAUNCH_CHECK("sdt_zero_ranges");
  return SDT_OK;
}

int sdt_param_prepare_desc_size(void) { return (int)sizeof(SdtPrepDesc); }

int sdt_embedding_fwd(const int32_t* ids, const float* tok, const float* pos, uint16_t* out, int64_t rows, int S, int D,
                      hipStream_t stream) {
  SDT_CHECK_ARG(ids && tok && pos && out && rows > 0 && S > 0 && D > 0, "sdt_embedding_fwd: bad args");
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(sdt_grid_1d(rows * D, 256)), dim3(256), 0, stream, ids, tok, pos,
                     (bf16_t*)out, (long)rows, S, D);
  SDT_LAUNCH_CHECK("sdt_embedding_fwd");
  return SDT_OK;
}

int sdt_embedding_bwd(const int32_t* ids, const uint16_t* dout, float* dtok, float* dpos, int64_t rows, int S, int D,
                      hipStream_t stream) {
  SDT_CHECK_ARG(ids && dout && dtok && dpos && rows > 0 && S > 0 && D > 0, "sdt_embedding_bwd: bad args");
  SDT_CHECK_ARG(rows + S < (1L << 31), "sdt_embedding_bwd: too many rows");
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)(rows + S)), dim3(256), 0, stream, ids,
                     (const bf16_t*)dout, dtok, dpos, (long)rows, S, D);
  SDT_LAUNCH_CHECK("sdt_embedding_bwd");
  return SDT_OK;
}

static int64_t colsum_ws_need(int groups, int nby) { return SDT_WS_COUNTER_BYTES + (int64_t)groups * nby * 256 * (int64_t)sizeof(float); }

/* scratch of the column-sum calls (split-workspace contract: first 64 KiB zero when enqueued, zero again afterwards) */
int64_t sdt_colsum_workspace_bytes(int batch, int64_t rows_per_batch, int N) {
  if (batch <= 0 || rows_per_batch <= 0 || N <= 0) return 0;
  const int ncb = sdt_ceil_div(sdt_ceil_div(N, 8), 32);
  return colsum_ws_need(ncb * batch, (int)((rows_per_batch + 63) / 64));
}

static int colsum_launch(const uint16_t* dy, float* db, uint16_t* out_bf, int batch, int64_t rows, int N, int ld, int want_blocks,
                         void* workspace, int64_t workspace_bytes, const char* name, hipStream_t stream) {
  const int ncb = sdt_ceil_div(sdt_ceil_div(N, 8), 32);
  int nby = (int)((rows + 63) / 64);
  const int want = (want_blocks + ncb * batch - 1) / (ncb * batch);
  if (nby > want) nby = want;
  const int rpb = (int)((rows + nby - 1) / nby);
  nby = sdt_ceil_div(rows, rpb);
  SDT_CHECK_ARG((int64_t)ncb * batch * (int64_t)sizeof(int) <= SDT_WS_COUNTER_BYTES, "%s: too many column groups", name);
  SDT_CHECK_ARG(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= colsum_ws_need(ncb * batch, nby),
                "%s: workspace of sdt_colsum_workspace_bytes() needed", name);
  hipLaunchKernelGGL(colsum_kernel, dim3(ncb, nby, batch), dim3(256), 0, stream, (const bf16_t*)dy, db, (bf16_t*)out_bf, (long)rows, N, ld,
                     rpb, batch > 1 ? (long)rows * ld : 0L, N, reinterpret_cast<int*>(workspace),
                     reinterpret_cast<float*>((unsigned char*)workspace + SDT_WS_COUNTER_BYTES));
  return SDT_OK;
}

/* db[n] += sum_m dy[m][n] */
int sdt_colsum_accumulate(const uint16_t* dy, float* db, int64_t M, int N, int ld, void* workspace, int64_t workspace_bytes,
                          hipStream_t stream) {
  SDT_CHECK_ARG(dy && db && M >= 0 && N > 0 && ld >= N && ld % 8 == 0 && ((uintptr_t)dy & 15) == 0,
                "sdt_colsum_accumulate: bad args (ld=%d must be a multiple of 8)", ld);
  if (M == 0) return SDT_OK;
  int rc = colsum_launch(dy, db, nullptr, 1, M, N, ld, 256, workspace, workspace_bytes, "sdt_colsum_accumulate", stream);
  if (rc) return rc;
  SDT_LAUNCH_CHECK("sdt_colsum_accumulate");
  return SDT_OK;
}

/* out[b][n] = bf16(sum over the rows of batch b (rows_per_batch consecutive rows each) of dy[m][n]); out is (batch, N) bf16:
 * the gradient of a per-image row bias (the time-embedding add of a ResBlock's first convolution) */
int sdt_colsum_batched_bf16(const uint16_t* dy, uint16_t* out, int batch, int64_t rows_per_batch, int N, int ld, void* workspace,
                            int64_t workspace_bytes, hipStream_t stream) {
  SDT_CHECK_ARG(dy && out && batch > 0 && batch < 65536 && rows_per_batch > 0 && N > 0 && ld >= N && ld % 8 == 0 &&
                    ((uintptr_t)dy & 15) == 0, "sdt_colsum_batched_bf16: bad args");
  int rc = colsum_launch(dy, nullptr, out, batch, rows_per_batch, N, ld, 512, workspace, workspace_bytes, "sdt_colsum_batched_bf16", stream);
  if (rc) return rc;
  SDT_LAUNCH_CHECK("sdt_colsum_batched_bf16");
  return SDT_OK;
}

int sdt_softmax_rows_inplace(uint16_t* x, int64_t rows, int n, float scale, hipStream_t stream) {
  SDT_CHECK_ARG(x && rows > 0 && n > 0 && rows < 2147483647L, "sdt_softmax_rows_inplace: bad args");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, stream, (bf16_t*)x, n, scale);
  SDT_LAUNCH_CHECK("sdt_softmax_rows_inplace");
  return SDT_OK;
}

int sdt_sum_n_bf16(const uint16_t* const* inputs, int n, uint16_t* out, int64_t numel, hipStream_t stream) {
  SDT_CHECK_ARG(inputs && out && n >= 1 && n <= 32 && numel >= 0 && numel % 8 == 0, "sdt_sum_n_bf16: bad arguments (n=%d numel=%ld)", n, (long)numel);
  SumPtrs sp;
  for (int k = 0; k < n; ++k) {
    SDT_CHECK_ARG(inputs[k] && ((uintptr_t)inputs[k] & 15) == 0, "sdt_sum_n_bf16: input %d null or misaligned", k);
    sp.p[k] = reinterpret_cast<const uint4*>(inputs[k]);
  }
  for (int k = n; k < 32; ++k) sp.p[k] = nullptr;
  if (numel == 0) return SDT_OK;
  hipLaunchKernelGGL(sum_n_kernel, dim3(sdt_grid_1d(numel / 8, 256, 2048)), dim3(256), 0, stream, sp, reinterpret_cast<uint4*>(out), n, numel / 8);
  SDT_LAUNCH_CHECK("sdt_sum_n_bf16");
  return SDT_OK;
}

}  // extern "C"
