// GroupNorm(+SiLU) and LayerNorm, forward and backward, NHWC / row-major bf16 activations, fp32 statistics.
// HBM-bound: wavefront + LDS reductions, 16-byte loads.  Replaces flax nn.GroupNorm / nn.LayerNorm as used by
// diffusers 0.21.4 FlaxResnetBlock2D / FlaxTransformer2DModel / vae_flax and transformers FlaxCLIP
// (third-party; reached from training_utils.py:574-579, 635-640, 678-684).  Variance = E[x^2]-E[x]^2 in fp32,
// like flax's use_fast_variance default.
#include "sdt_common.h"

#define GN_MAXJ 2  // channel vectors per thread: supports C <= 8*256*2 = 4096

struct GnLayout {
  int Cv, TX, TY, J;
};
__device__ __forceinline__ GnLayout gn_layout(int C) {
  GnLayout L;
  L.Cv = C >> 3;
  L.TX = L.Cv < 256 ? L.Cv : 256;
  L.J = (L.Cv + L.TX - 1) / L.TX;
  L.TY = 256 / L.TX;
  return L;
}

// ---------------------------------------------------------------- GroupNorm: statistics as ordered partial sums
// No float atomics anywhere in this file: every reduction that crosses threads or workgroups is a set of partial sums written by
// exactly one writer each and added up in a FIXED order, so the same inputs give the same bits on every launch (the reference's
// jitted step is deterministic).
//
// Statistics of one image travel as `nparts` partial rows part[b][i][g] = {sum, sum of squares} of group g (i < nparts); they come
// from gn_stats_kernel (one row per workgroup) or from the epilogue of the GEMM / convolution that wrote the tensor
// (sdt_gemm_nt_bf16 gn_stats: two rows per output row tile).  A consumer adds the rows in index order.
#define GN_MAX_INLINE_PARTS 128  // up to this many rows the apply kernels add them up in their own prologue
#define GN_PR_COLS 16            // columns per workgroup of the dgamma / dbeta partial-row sums riding in the backward apply launch

// part[b][blockIdx.x][g] = {sum, sumsq} over this block's pixels
__global__ void __launch_bounds__(256) gn_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, int HW, int C, int G,
                                                       int pix_per_block) {
  extern __shared__ __attribute__((aligned(16))) float chs[];  // [TY][2][C]: per-thread-row channel sums of this block
  const GnLayout L = gn_layout(C);
  const int b = blockIdx.y;
  const int tx = threadIdx.x % L.TX, ty = threadIdx.x / L.TX;
  const int cpg = C / G;
  float s[GN_MAXJ][8], q[GN_MAXJ][8];
#pragma unroll
  for (int j = 0; j < GN_MAXJ; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[j][e] = 0.f; q[j][e] = 0.f; }
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, HW);
  if (ty < L.TY) {
    const bf16_t* xb = x + (long)b * HW * C;
#pragma unroll 4
    for (int p = p0 + ty; p < p1; p += L.TY) {
#pragma unroll
      for (int j = 0; j < GN_MAXJ; ++j) {
        const int cv = tx + j * L.TX;
        if (j < L.J && cv < L.Cv) {
          float f[8];
          unpack8(*reinterpret_cast<const uint4*>(xb + (long)p * C + cv * 8), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) { s[j][e] += f[e]; q[j][e] += f[e] * f[e]; }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < GN_MAXJ; ++j) {
      const int cv = tx + j * L.TX;
      if (j < L.J && cv < L.Cv) {  // private slots: no LDS atomics
        float* r1 = chs + (long)ty * 2 * C + cv * 8;
        float* r2 = r1 + C;
        *reinterpret_cast<float4*>(r1) = make_float4(s[j][0], s[j][1], s[j][2], s[j][3]);
        *reinterpret_cast<float4*>(r1 + 4) = make_float4(s[j][4], s[j][5], s[j][6], s[j][7]);
        *reinterpret_cast<float4*>(r2) = make_float4(q[j][0], q[j][1], q[j][2], q[j][3]);
        *reinterpret_cast<float4*>(r2 + 4) = make_float4(q[j][4], q[j][5], q[j][6], q[j][7]);
      }
    }
  }
  __syncthreads();
  // 2G threads: (group, which sum) over the group's channels x TY thread rows, fixed order
  if (threadIdx.x < 2 * G) {
    const int g = threadIdx.x >> 1, w = threadIdx.x & 1;
    float a = 0.f;
    for (int r = 0; r < L.TY; ++r) {
      const float* row = chs + (long)r * 2 * C + w * C + g * cpg;
      for (int c = 0; c < cpg; ++c) a += row[c];
    }
    part[(((long)b * gridDim.x + blockIdx.x) * G + g) * 2 + w] = a;
  }
}

// Many partial rows per image (the VAE's 512x512 levels leave 2048): first level of the sum.  out[b][y][i] = sum over the rows
// [y*rows_per, (y+1)*rows_per) of image b of part[b][row][i], i < n2 = 2G (<= 128); gridDim = (GN_L2_PARTS, B).  The apply kernel
// then adds the GN_L2_PARTS rows of `out` in its prologue like any other partial rows.
#define GN_L2_PARTS 32
__global__ void __launch_bounds__(256) gn_group_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk,
                                                              int n2, int rows_per) {
  __shared__ float red[256];
  const int b = blockIdx.y;
  const int slices = 256 / n2;
  const int item = threadIdx.x % n2, sl = threadIdx.x / n2;
  const int r0 = blockIdx.x * rows_per, r1 = min(r0 + rows_per, nblk);
  float acc = 0.f;
  if (sl < slices) {
    const float* pb = part + (long)b * nblk * n2 + item;
#pragma unroll 4
    for (int i = r0 + sl; i < r1; i += slices) acc += pb[(long)i * n2];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < n2) {
    float t = 0.f;
    for (int k = 0; k < slices; ++k) t += red[k * n2 + threadIdx.x];
    out[((long)b * gridDim.x + blockIdx.x) * n2 + threadIdx.x] = t;
  }
}

// Workgroup-wide: tot[i] = sum over the nparts rows of part[row][i], i < n2 = 2G (<= 128), in a fixed order (same split of the
// rows over 256 / n2 thread slices as gn_group_reduce_kernel, so a tensor gets the same sums whichever of the two adds them).
__device__ __forceinline__ void gn_sum_parts(const float* __restrict__ part, int nparts, int n2, float* red /*[256]*/, float* tot /*[128]*/) {
  const int slices = 256 / n2;
  const int item = threadIdx.x % n2, sl = threadIdx.x / n2;
  float acc = 0.f;
  if (sl < slices)
    for (int i = sl; i < nparts; i += slices) acc += part[(long)i * n2 + item];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < n2) {
    float t = 0.f;
    for (int k = 0; k < slices; ++k) t += red[k * n2 + threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
}

// y = (silu)(gamma * (x - mean) * rstd + beta).  part: nparts rows of statistics per image (added up here, in the prologue of
// every workgroup: <= 64 rows x 2G floats from L2); the workgroups of blockIdx.x == 0 also store the totals to stats[b][G][2]
// (what the backward reads).  part == stats with nparts == 1 is the "already reduced" form.
template <bool SILU>
__global__ void __launch_bounds__(256) gn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ part, int nparts,
                                                       float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, bf16_t* __restrict__ y, int HW, int C, int G,
                                                       int pix_per_block, float eps) {
  __shared__ float red[256], tot[128], gmean[64], grstd[64];
  const GnLayout L = gn_layout(C);
  const int b = blockIdx.y;
  const int tx = threadIdx.x % L.TX, ty = threadIdx.x / L.TX;
  const int cpg = C / G;
  const float inv_cnt = 1.0f / ((float)HW * cpg);
  gn_sum_parts(part + (long)b * nparts * 2 * G, nparts, 2 * G, red, tot);
  if (threadIdx.x < G) {
    const float mean = tot[2 * threadIdx.x] * inv_cnt;
    const float var = fmaxf(tot[2 * threadIdx.x + 1] * inv_cnt - mean * mean, 0.f);
    gmean[threadIdx.x] = mean;
    grstd[threadIdx.x] = rsqrtf(var + eps);
  }
  if (blockIdx.x == 0 && threadIdx.x < 2 * G && stats != part) stats[(long)b * 2 * G + threadIdx.x] = tot[threadIdx.x];
  __syncthreads();
  if (ty >= L.TY) return;
  float a[GN_MAXJ][8], sh[GN_MAXJ][8];
#pragma unroll
  for (int j = 0; j < GN_MAXJ; ++j) {
    const int cv = tx + j * L.TX;
    float gm[8], bt[8];
    if (j < L.J && cv < L.Cv) {
      *reinterpret_cast<float4*>(gm) = *reinterpret_cast<const float4*>(gamma + cv * 8);
      *reinterpret_cast<float4*>(gm + 4) = *reinterpret_cast<const float4*>(gamma + cv * 8 + 4);
      *reinterpret_cast<float4*>(bt) = *reinterpret_cast<const float4*>(beta + cv * 8);
      *reinterpret_cast<float4*>(bt + 4) = *reinterpret_cast<const float4*>(beta + cv * 8 + 4);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[j][e] = 0.f; sh[j][e] = 0.f;
      if (j < L.J && cv < L.Cv) {
        const int g = (cv * 8 + e) / cpg;
        a[j][e] = grstd[g] * gm[e];
        sh[j][e] = bt[e] - gmean[g] * a[j][e];
      }
    }
  }
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, HW);
  const bf16_t* xb = x + (long)b * HW * C;
  bf16_t* yb = y + (long)b * HW * C;
#pragma unroll 4
  for (int p = p0 + ty; p < p1; p += L.TY) {
#pragma unroll
    for (int j = 0; j < GN_MAXJ; ++j) {
      const int cv = tx + j * L.TX;
      if (j < L.J && cv < L.Cv) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(xb + (long)p * C + cv * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float z = a[j][e] * f[e] + sh[j][e];
          f[e] = SILU ? siluf_(z) : z;
        }
        *reinterpret_cast<uint4*>(yb + (long)p * C + cv * 8) = pack8(f);
      }
    }
  }
}

// ---------------------------------------------------------------- GroupNorm backward
// pass 1: per-channel sums of dz and dz*xhat of this block's pixels -> partial[blk][dgamma C | dbeta C] and the per-group
// sums gpart[b][blk][g] = {S1, S2} (gamma-weighted); both are added up by pass 2 in a fixed order
template <bool SILU>
__global__ void __launch_bounds__(256) gn_bwd_stats_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ partial,
                                                           float* __restrict__ gpart, int HW, int C, int G, int pix_per_block,
                                                           float eps) {
  extern __shared__ __attribute__((aligned(16))) float chs[];  // [TY][2][C]: per-thread-row channel sums of this block
  const GnLayout L = gn_layout(C);
  const int b = blockIdx.y;
  const int tx = threadIdx.x % L.TX, ty = threadIdx.x / L.TX;
  const int cpg = C / G;
  const float inv_cnt = 1.0f / ((float)HW * cpg);
  if (ty < L.TY) {
    float mean[GN_MAXJ][8], rstd[GN_MAXJ][8], gam[GN_MAXJ][8], bet[GN_MAXJ][8], s1[GN_MAXJ][8], s2[GN_MAXJ][8];
#pragma unroll
    for (int j = 0; j < GN_MAXJ; ++j) {
      const int cv = tx + j * L.TX;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        mean[j][e] = 0.f; rstd[j][e] = 0.f; gam[j][e] = 0.f; bet[j][e] = 0.f; s1[j][e] = 0.f; s2[j][e] = 0.f;
        if (j < L.J && cv < L.Cv) {
          const int ch = cv * 8 + e, g = ch / cpg;
          const float m = stats[((long)b * G + g) * 2] * inv_cnt;
          const float var = fmaxf(stats[((long)b * G + g) * 2 + 1] * inv_cnt - m * m, 0.f);
          mean[j][e] = m;
          rstd[j][e] = rsqrtf(var + eps);
          gam[j][e] = gamma[ch];
          bet[j][e] = beta[ch];
        }
      }
    }
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(p0 + pix_per_block, HW);
    const bf16_t* xb = x + (long)b * HW * C;
    const bf16_t* db = dy + (long)b * HW * C;
#pragma unroll 4
    for (int p = p0 + ty; p < p1; p += L.TY) {
#pragma unroll
      for (int j = 0; j < GN_MAXJ; ++j) {
        const int cv = tx + j * L.TX;
        if (j < L.J && cv < L.Cv) {
          float f[8], d[8];
          unpack8(*reinterpret_cast<const uint4*>(xb + (long)p * C + cv * 8), f);
          unpack8(*reinterpret_cast<const uint4*>(db + (long)p * C + cv * 8), d);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = (f[e] - mean[j][e]) * rstd[j][e];
            float dz = d[e];
            if (SILU) {
              const float z = gam[j][e] * xh + bet[j][e];
              const float sg = sigmoidf_(z);
              dz = dz * sg * (1.f + z * (1.f - sg));
            }
            s1[j][e] += dz;
            s2[j][e] += dz * xh;
          }
        }
      }
    }
    // private slots red[ty][{s1,s2}][C]: no LDS atomics (every thread of a block would hit the same few addresses)
#pragma unroll
    for (int j = 0; j < GN_MAXJ; ++j) {
      const int cv = tx + j * L.TX;
      if (j < L.J && cv < L.Cv) {
        float* r1 = chs + (long)ty * 2 * C + cv * 8;
        float* r2 = r1 + C;
        *reinterpret_cast<float4*>(r1) = make_float4(s1[j][0], s1[j][1], s1[j][2], s1[j][3]);
        *reinterpret_cast<float4*>(r1 + 4) = make_float4(s1[j][4], s1[j][5], s1[j][6], s1[j][7]);
        *reinterpret_cast<float4*>(r2) = make_float4(s2[j][0], s2[j][1], s2[j][2], s2[j][3]);
        *reinterpret_cast<float4*>(r2 + 4) = make_float4(s2[j][4], s2[j][5], s2[j][6], s2[j][7]);
      }
    }
  }
  __syncthreads();
  // per-channel sums of this block -> its private partial row [dgamma | dbeta] (frozen norms pass partial = nullptr)
  float* pp = partial ? partial + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C : nullptr;
  for (int ch = threadIdx.x; ch < C; ch += 256) {
    float a1 = 0.f, a2 = 0.f;
    for (int r = 0; r < L.TY; ++r) {
      a1 += chs[(long)r * 2 * C + ch];
      a2 += chs[(long)r * 2 * C + C + ch];
    }
    if (pp) {
      pp[ch] = a2;
      pp[C + ch] = a1;
    }
    const float gm = gamma[ch];
    chs[ch] = gm * a1;  // row 0 of this channel was read by this thread only
    chs[C + ch] = gm * a2;
  }
  __syncthreads();
  if (threadIdx.x < G) {
    float g1 = 0.f, g2 = 0.f;
    for (int ch = threadIdx.x * cpg; ch < (threadIdx.x + 1) * cpg; ++ch) {
      g1 += chs[ch];
      g2 += chs[C + ch];
    }
    float* o = gpart + (((long)b * gridDim.x + blockIdx.x) * G + threadIdx.x) * 2;
    o[0] = g1;
    o[1] = g2;
  }
}

// column sums of `nrows` partial rows [2C] into dgamma / dbeta (+=, ONE writer per element, rows added in a fixed order):
// workgroup `blk` owns COLS of the 2C columns; 256 threads = COLS columns x 256 / COLS row slices (slice s adds rows s, s + S, ...;
// the slices are then added in slice order)
template <int COLS>
__device__ __forceinline__ void partial_rows_reduce(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                    float* __restrict__ dbeta, int nrows, int C, int blk, float* red /*[256]*/) {
  constexpr int SL = 256 / COLS;
  const int cx = threadIdx.x % COLS, sy = threadIdx.x / COLS;
  const int ch = blk * COLS + cx;
  float s = 0.f;
  if (ch < 2 * C) {
#pragma unroll 8
    for (int r = sy; r < nrows; r += SL) s += partial[(long)r * 2 * C + ch];
  }
  red[sy * COLS + cx] = s;
  __syncthreads();
  if (sy == 0 && ch < 2 * C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < SL; ++k) t += red[k * COLS + cx];
    if (ch < C) dgamma[ch] += t; else dbeta[ch - C] += t;
  }
}

// pass 2: dx = rstd * (gamma*dz - (S1 + xhat*S2)/cnt); {S1, S2} = the gpart rows of the image added up in the prologue.
// Workgroups blockIdx.x >= nch of image 0 add the partial rows of pass 1 into dgamma / dbeta (they run beside the apply work).
template <bool SILU>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           const float* __restrict__ stats, const float* __restrict__ gpart,
                                                           int nparts, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           bf16_t* __restrict__ dx, const bf16_t* __restrict__ dres,
                                                           const float* __restrict__ partial, int partial_rows,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int nch, int HW,
                                                           int C, int G, int pix_per_block, float eps) {
  __shared__ float red[256], tot[128], gmean[64], grstd[64];
  if ((int)blockIdx.x >= nch) {
    if (blockIdx.y == 0 && partial) partial_rows_reduce<GN_PR_COLS>(partial, dgamma, dbeta, partial_rows, C, (int)blockIdx.x - nch, red);
    return;
  }
  const GnLayout L = gn_layout(C);
  const int b = blockIdx.y;
  const int tx = threadIdx.x % L.TX, ty = threadIdx.x / L.TX;
  const int cpg = C / G;
  const float inv_cnt = 1.0f / ((float)HW * cpg);
  gn_sum_parts(gpart + (long)b * nparts * 2 * G, nparts, 2 * G, red, tot);
  if (threadIdx.x < G) {
    const float m = stats[((long)b * G + threadIdx.x) * 2] * inv_cnt;
    const float var = fmaxf(stats[((long)b * G + threadIdx.x) * 2 + 1] * inv_cnt - m * m, 0.f);
    gmean[threadIdx.x] = m;
    grstd[threadIdx.x] = rsqrtf(var + eps);
  }
  __syncthreads();
  if (ty >= L.TY) return;
  float mean[GN_MAXJ][8], rstd[GN_MAXJ][8], gam[GN_MAXJ][8], bet[GN_MAXJ][8], k1[GN_MAXJ][8], k2[GN_MAXJ][8];
#pragma unroll
  for (int j = 0; j < GN_MAXJ; ++j) {
    const int cv = tx + j * L.TX;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mean[j][e] = 0.f; rstd[j][e] = 0.f; gam[j][e] = 0.f; bet[j][e] = 0.f; k1[j][e] = 0.f; k2[j][e] = 0.f;
      if (j < L.J && cv < L.Cv) {
        const int ch = cv * 8 + e, g = ch / cpg;
        const float r = grstd[g];
        mean[j][e] = gmean[g]; rstd[j][e] = r; gam[j][e] = gamma[ch]; bet[j][e] = beta[ch];
        k1[j][e] = r * tot[2 * g] * inv_cnt;
        k2[j][e] = r * tot[2 * g + 1] * inv_cnt;
      }
    }
  }
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, HW);
  const bf16_t* xb = x + (long)b * HW * C;
  const bf16_t* db = dy + (long)b * HW * C;
  bf16_t* ob = dx + (long)b * HW * C;
#pragma unroll 4
  for (int p = p0 + ty; p < p1; p += L.TY) {
#pragma unroll
    for (int j = 0; j < GN_MAXJ; ++j) {
      const int cv = tx + j * L.TX;
      if (j < L.J && cv < L.Cv) {
        float f[8], d[8];
        unpack8(*reinterpret_cast<const uint4*>(xb + (long)p * C + cv * 8), f);
        unpack8(*reinterpret_cast<const uint4*>(db + (long)p * C + cv * 8), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (f[e] - mean[j][e]) * rstd[j][e];
          float dz = d[e];
          if (SILU) {
            const float z = gam[j][e] * xh + bet[j][e];
            const float sg = sigmoidf_(z);
            dz = dz * sg * (1.f + z * (1.f - sg));
          }
          f[e] = rstd[j][e] * gam[j][e] * dz - k1[j][e] - xh * k2[j][e];
        }
        if (dres) {  // gradient arriving over the skip branch that forked off x (rounded like a separate add)
          unpack8(pack8(f), f);
          unpack8(*reinterpret_cast<const uint4*>(dres + ((long)b * HW + p) * C + cv * 8), d);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += d[e];
        }
        *reinterpret_cast<uint4*>(ob + (long)p * C + cv * 8) = pack8(f);
      }
    }
  }
}

// ---------------------------------------------------------------- LayerNorm (one wave per row)
#define LN_MAXV 4  // C <= 8*64*4 = 2048
__global__ void __launch_bounds__(256) ln_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean_rstd, long M, int C, float eps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int Cv = C >> 3;
  const long wave = (long)blockIdx.x * 4 + wid, nwaves = (long)gridDim.x * 4;
  for (long r = wave; r < M; r += nwaves) {
    float f[LN_MAXV][8];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int cv = lane + 64 * j;
      if (cv < Cv) {
        unpack8(*reinterpret_cast<const uint4*>(x + r * C + cv * 8), f[j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s += f[j][e]; q += f[j][e] * f[j][e]; }
      }
    }
    s = wave_sum(s);
    q = wave_sum(q);
    const float mean = s / C;
    const float var = fmaxf(q / C - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    if (lane == 0 && mean_rstd) { mean_rstd[r * 2] = mean; mean_rstd[r * 2 + 1] = rstd; }
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int cv = lane + 64 * j;
      if (cv < Cv) {
        float gm[8], bt[8];
        *reinterpret_cast<float4*>(gm) = *reinterpret_cast<const float4*>(gamma + cv * 8);
        *reinterpret_cast<float4*>(gm + 4) = *reinterpret_cast<const float4*>(gamma + cv * 8 + 4);
        *reinterpret_cast<float4*>(bt) = *reinterpret_cast<const float4*>(beta + cv * 8);
        *reinterpret_cast<float4*>(bt + 4) = *reinterpret_cast<const float4*>(beta + cv * 8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[j][e] = (f[j][e] - mean) * rstd * gm[e] + bt[e];
        *reinterpret_cast<uint4*>(y + r * C + cv * 8) = pack8(f[j]);
      }
    }
  }
}

// NR rows per wave are processed together (their loads issued back to back): the row loop is latency-serial otherwise.
template <int MAXV, int NR>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                     bf16_t* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ partial,
                                                     const bf16_t* __restrict__ dres, long M, int C) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int Cv = C >> 3;
  const long wave = (long)blockIdx.x * 4 + wid, nwaves = (long)gridDim.x * 4;
  float gm[MAXV][8], ag[MAXV][8], ab[MAXV][8];
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int cv = lane + 64 * j;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[j][e] = 0.f; ab[j][e] = 0.f; gm[j][e] = 0.f; }
    if (cv < Cv) {
      *reinterpret_cast<float4*>(gm[j]) = *reinterpret_cast<const float4*>(gamma + cv * 8);
      *reinterpret_cast<float4*>(gm[j] + 4) = *reinterpret_cast<const float4*>(gamma + cv * 8 + 4);
    }
  }
  for (long r0 = wave * NR; r0 < M; r0 += nwaves * NR) {
    uint4 xv[NR][MAXV], dv[NR][MAXV];
    float mean[NR], rstd[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const long r = (r0 + k < M) ? r0 + k : M - 1;  // clamp: loads stay unconditional, the store is guarded
      mean[k] = mean_rstd[r * 2];
      rstd[k] = mean_rstd[r * 2 + 1];
#pragma unroll
      for (int j = 0; j < MAXV; ++j) {
        const int cv = (lane + 64 * j < Cv) ? lane + 64 * j : 0;
        xv[k][j] = *reinterpret_cast<const uint4*>(x + r * C + cv * 8);
        dv[k][j] = *reinterpret_cast<const uint4*>(dy + r * C + cv * 8);
      }
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const bool rowok = r0 + k < M;
      float xh[MAXV][8], g[MAXV][8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < MAXV; ++j) {
        const bool ok = rowok && (lane + 64 * j < Cv);
        float f[8], d[8];
        unpack8(xv[k][j], f);
        unpack8(dv[k][j], d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[j][e] = ok ? (f[e] - mean[k]) * rstd[k] : 0.f;
          const float de = ok ? d[e] : 0.f;
          g[j][e] = de * gm[j][e];
          s1 += g[j][e];
          s2 += g[j][e] * xh[j][e];
          ag[j][e] += de * xh[j][e];
          ab[j][e] += de;
        }
      }
      s1 = wave_sum(s1) / C;
      s2 = wave_sum(s2) / C;
      if (rowok) {
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
          const int cv = lane + 64 * j;
          if (cv < Cv) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rstd[k] * (g[j][e] - s1 - xh[j][e] * s2);
            if (dres) {  // gradient arriving over the residual branch that forked off x (rounded like a separate add)
              float rr[8];
              unpack8(pack8(o), o);
              unpack8(*reinterpret_cast<const uint4*>(dres + (r0 + k) * C + cv * 8), rr);
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] += rr[e];
            }
            *reinterpret_cast<uint4*>(dx + (r0 + k) * C + cv * 8) = pack8(o);
          }
        }
      }
    }
  }
  if (dgamma) {
    // block sums of the per-lane channel partials: every wave stores its [2][C] slice (plain 16-byte stores), then 256 threads add
    // the four slices.  (LDS float atomics here - 16 to 64 per lane at a 32-byte lane stride, i.e. 16 lanes per bank - cost 15-30 us
    // per launch: four times the rest of the kernel.)
    extern __shared__ float lred[];  // [4 waves][2][C]
    float* mine = lred + wid * 2 * C;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int cv = lane + 64 * j;
      if (cv < Cv) {
        *reinterpret_cast<float4*>(mine + cv * 8) = make_float4(ag[j][0], ag[j][1], ag[j][2], ag[j][3]);
        *reinterpret_cast<float4*>(mine + cv * 8 + 4) = make_float4(ag[j][4], ag[j][5], ag[j][6], ag[j][7]);
        *reinterpret_cast<float4*>(mine + C + cv * 8) = make_float4(ab[j][0], ab[j][1], ab[j][2], ab[j][3]);
        *reinterpret_cast<float4*>(mine + C + cv * 8 + 4) = make_float4(ab[j][4], ab[j][5], ab[j][6], ab[j][7]);
      }
    }
    __syncthreads();
    float* pp = partial + (long)blockIdx.x * 2 * C;
    for (int ch = threadIdx.x; ch < 2 * C; ch += 256)  // per-block partial row, added up by partial_reduce_kernel (fixed order)
      pp[ch] = (lred[ch] + lred[2 * C + ch]) + (lred[4 * C + ch] + lred[6 * C + ch]);
  }
}

// dgamma[c] += sum_b partial[b][c] ; dbeta[c] += sum_b partial[b][C + c]: workgroup x owns 8 of the 2C columns and adds ALL the
// rows in a fixed order (one writer per element: deterministic); 32 row slices per workgroup keep the chain of dependent loads
// short (up to 1024 partial rows from sdt_layernorm_bwd)
#define LN_PR_COLS 8
__global__ void __launch_bounds__(256) partial_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int nblk, int C) {
  __shared__ float red[256];
  partial_rows_reduce<LN_PR_COLS>(partial, dgamma, dbeta, nblk, C, (int)blockIdx.x, red);
}

static void launch_partial_reduce(const float* partial, float* dgamma, float* dbeta, int nblk, int C, hipStream_t stream) {
  hipLaunchKernelGGL(partial_reduce_kernel, dim3(sdt_ceil_div(2 * C, LN_PR_COLS)), dim3(256), 0, stream, partial, dgamma, dbeta, nblk, C);
}

// The same sums for MANY norms in one launch (sdt_norm_param_grads_group): a transformer block's three LayerNorms each end their
// backward with a ~5 us partial_reduce launch that nothing downstream waits for; held back, a step's worth of them is one launch.
#define PR_GROUP_MAX 96
struct PrGroupParams {
  int n;
  int wg_end[PR_GROUP_MAX];
  int nrows[PR_GROUP_MAX];
  int C[PR_GROUP_MAX];
  const float* partial[PR_GROUP_MAX];
  float* dgamma[PR_GROUP_MAX];
  float* dbeta[PR_GROUP_MAX];
};
static_assert(sizeof(PrGroupParams) <= 4096, "the grouped launch passes its table as kernel arguments");
__global__ void __launch_bounds__(256) partial_reduce_group_kernel(const PrGroupParams gp) {
  __shared__ float red[256];
  const int b = blockIdx.x;
  int lo = 0, hi = gp.n - 1;  // wave-uniform binary search over <= 96 entries
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (b >= gp.wg_end[mid]) lo = mid + 1; else hi = mid;
  }
  const int local = b - (lo ? gp.wg_end[lo - 1] : 0);
  partial_rows_reduce<LN_PR_COLS>(gp.partial[lo], gp.dgamma[lo], gp.dbeta[lo], gp.nrows[lo], gp.C[lo], local, red);
}

// ================================================================== C ABI
// pixel rows per block for `total_blocks` blocks over the batch, but at least two row-iterations per thread (TY rows are in flight
// per iteration); low-resolution, wide-channel tensors (8x8x1280) are latency-bound, so they get many small blocks
static int gn_chunks(int B, int HW, int C, int* pix_per_block, int total_blocks, int max_per_image) {
  int target = total_blocks / (B > 0 ? B : 1);
  if (target < 1) target = 1;
  if (max_per_image > 0 && target > max_per_image) target = max_per_image;
  const int cv = C >> 3;
  const int ty = 256 / (cv < 256 ? cv : 256);
  int ppb = (HW + target - 1) / target;
  if (ppb < 4 * ty) ppb = 4 * ty;
  *pix_per_block = ppb;
  return (HW + ppb - 1) / ppb;
}
#define GN_FINE_BLOCKS 1024
// statistics passes: at most GN_MAX_INLINE_PARTS partial rows per image, which the apply kernels add up in their prologues
static int gn_stat_chunks(int B, int HW, int C, int* ppb) { return gn_chunks(B, HW, C, ppb, GN_FINE_BLOCKS, GN_MAX_INLINE_PARTS); }
static size_t gn_chs_bytes(int C) {
  const int cvh = C >> 3;
  return sizeof(float) * 2 * C * (256 / (cvh < 256 ? cvh : 256));  // [TY][2][C]
}
static int gn_check(const void* x, int B, int HW, int C, int G, const char* name) {
  SDT_CHECK_ARG(x && B > 0 && HW > 0 && C > 0 && G > 0 && G <= 64, "%s: bad shape B=%d HW=%d C=%d G=%d", name, B, HW, C, G);
  SDT_CHECK_ARG(C % 8 == 0 && C % G == 0 && C <= 8 * 256 * GN_MAXJ, "%s: C=%d must be a multiple of 8 and of G=%d, <= %d", name, C, G, 8 * 256 * GN_MAXJ);
  SDT_CHECK_ARG(((uintptr_t)x & 15) == 0, "%s: x must be 16-byte aligned", name);
  SDT_CHECK_ARG(B <= 65535, "%s: B too large", name);
  return SDT_OK;
}

extern "C" {

/* scratch of sdt_groupnorm_fwd: the partial rows of its own statistics pass (parts == NULL), or the first-level sums of a
 * producer's partial rows when there are more than 128 of them per image */
int64_t sdt_groupnorm_fwd_workspace_bytes(int B, int HW, int C, int G) {
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0) return 0;
  int ppb;
  int nch = gn_stat_chunks(B, HW, C, &ppb);
  if (nch < GN_L2_PARTS) nch = GN_L2_PARTS;  // (also covers the first-level sums of a producer's many partial rows)
  return (int64_t)nch * B * 2 * G * (int64_t)sizeof(float);
}

int sdt_groupnorm_fwd(const uint16_t* x, const float* gamma, const float* beta, uint16_t* y, float* stats, int B, int HW,
                      int C, int G, float eps, int fuse_silu, const float* parts, int nparts, void* workspace,
                      int64_t workspace_bytes, hipStream_t stream) {
  int rc = gn_check(x, B, HW, C, G, "sdt_groupnorm_fwd");
  if (rc) return rc;
  SDT_CHECK_ARG(gamma && beta && y && stats, "sdt_groupnorm_fwd: null pointer");
  SDT_CHECK_ARG((parts == nullptr) == (nparts == 0) && nparts >= 0, "sdt_groupnorm_fwd: parts / nparts mismatch");
  SDT_CHECK_ARG((((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "sdt_groupnorm_fwd: gamma / beta must be 16-byte aligned");
  int ppb;
  const int nch = gn_chunks(B, HW, C, &ppb, GN_FINE_BLOCKS, 0);
  const float* src = parts;
  int n = nparts;
  if (!parts) {  // statistics pass of our own: one partial row per workgroup
    SDT_CHECK_ARG(workspace && workspace_bytes >= sdt_groupnorm_fwd_workspace_bytes(B, HW, C, G),
                  "sdt_groupnorm_fwd: workspace of sdt_groupnorm_fwd_workspace_bytes() needed when no statistics are passed");
    int ppb_s;
    const int nch_s = gn_stat_chunks(B, HW, C, &ppb_s);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nch_s, B), dim3(256), gn_chs_bytes(C), stream, (const bf16_t*)x, (float*)workspace, HW, C, G, ppb_s);
    src = (const float*)workspace;
    n = nch_s;
  } else if (nparts > GN_MAX_INLINE_PARTS) {  // many producer tiles per image (the VAE's large levels): a first-level sum
    SDT_CHECK_ARG(workspace && workspace_bytes >= (int64_t)sizeof(float) * B * GN_L2_PARTS * 2 * G,
                  "sdt_groupnorm_fwd: workspace of sdt_groupnorm_fwd_workspace_bytes() needed for %d partial rows", nparts);
    hipLaunchKernelGGL(gn_group_reduce_kernel, dim3(GN_L2_PARTS, B), dim3(256), 0, stream, parts, (float*)workspace, nparts, 2 * G,
                       sdt_ceil_div(nparts, GN_L2_PARTS));
    src = (const float*)workspace;
    n = GN_L2_PARTS;
  }
  if (fuse_silu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(nch, B), dim3(256), 0, stream, (const bf16_t*)x, src, n, stats, gamma, beta, (bf16_t*)y, HW, C, G, ppb, eps);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(nch, B), dim3(256), 0, stream, (const bf16_t*)x, src, n, stats, gamma, beta, (bf16_t*)y, HW, C, G, ppb, eps);
  SDT_LAUNCH_CHECK("sdt_groupnorm_fwd");
  return SDT_OK;
}

/* scratch of sdt_groupnorm_bwd (required): per workgroup of its first pass the channel sums [2C] and the group sums [2G <= 128] */
int64_t sdt_groupnorm_bwd_workspace_bytes(int B, int HW, int C) {
  if (B <= 0 || HW <= 0 || C <= 0) return 0;
  int ppb;
  const int nch = gn_stat_chunks(B, HW, C, &ppb);
  return (int64_t)nch * B * (2 * C + 2 * 64) * (int64_t)sizeof(float);
}

// dgamma/dbeta may be null (frozen norm); otherwise accumulated (+=) by one writer per element.
int sdt_groupnorm_bwd(const uint16_t* x, const uint16_t* dy, const float* stats, const float* gamma, const float* beta,
                      uint16_t* dx, float* dgamma, float* dbeta, const uint16_t* dres, int B, int HW, int C, int G, float eps,
                      int fuse_silu, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  int rc = gn_check(x, B, HW, C, G, "sdt_groupnorm_bwd");
  if (rc) return rc;
  SDT_CHECK_ARG(dy && stats && gamma && beta && dx && ((dgamma == nullptr) == (dbeta == nullptr)), "sdt_groupnorm_bwd: null pointer");
  SDT_CHECK_ARG(workspace && workspace_bytes >= sdt_groupnorm_bwd_workspace_bytes(B, HW, C),
                "sdt_groupnorm_bwd: workspace of sdt_groupnorm_bwd_workspace_bytes() needed");
  int ppb, ppb_s;
  const int nch = gn_chunks(B, HW, C, &ppb, GN_FINE_BLOCKS, 0);
  const int nch_s = gn_stat_chunks(B, HW, C, &ppb_s);
  float* gpart = (float*)workspace;                                      // [B][nch_s][G][2]
  float* part = dgamma ? gpart + (size_t)nch_s * B * 2 * 64 : nullptr;   // [B * nch_s][2C]
  const int extra = dgamma ? sdt_ceil_div(2 * C, GN_PR_COLS) : 0;        // workgroups of the apply launch that add up `part`
  if (fuse_silu) {
    hipLaunchKernelGGL(gn_bwd_stats_kernel<true>, dim3(nch_s, B), dim3(256), gn_chs_bytes(C), stream, (const bf16_t*)x, (const bf16_t*)dy, stats, gamma, beta, part, gpart, HW, C, G, ppb_s, eps);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3(nch + extra, B), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, stats, (const float*)gpart, nch_s, gamma, beta, (bf16_t*)dx, (const bf16_t*)dres, (const float*)part, nch_s * B, dgamma, dbeta, nch, HW, C, G, ppb, eps);
  } else {
    hipLaunchKernelGGL(gn_bwd_stats_kernel<false>, dim3(nch_s, B), dim3(256), gn_chs_bytes(C), stream, (const bf16_t*)x, (const bf16_t*)dy, stats, gamma, beta, part, gpart, HW, C, G, ppb_s, eps);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3(nch + extra, B), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, stats, (const float*)gpart, nch_s, gamma, beta, (bf16_t*)dx, (const bf16_t*)dres, (const float*)part, nch_s * B, dgamma, dbeta, nch, HW, C, G, ppb, eps);
  }
  SDT_LAUNCH_CHECK("sdt_groupnorm_bwd");
  return SDT_OK;
}

int sdt_layernorm_fwd(const uint16_t* x, const float* gamma, const float* beta, uint16_t* y, float* mean_rstd, int64_t M,
                      int C, float eps, hipStream_t stream) {
  SDT_CHECK_ARG(x && gamma && beta && y && M >= 0, "sdt_layernorm_fwd: null pointer");
  SDT_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 8 * 64 * LN_MAXV, "sdt_layernorm_fwd: C=%d must be a multiple of 8 and <= %d", C, 8 * 64 * LN_MAXV);
  SDT_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "sdt_layernorm_fwd: misaligned pointer");
  if (M == 0) return SDT_OK;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(sdt_grid_1d(M, 4, 4096)), dim3(256), 0, stream, (const bf16_t*)x, gamma, beta,
                     (bf16_t*)y, mean_rstd, (long)M, C, eps);
  SDT_LAUNCH_CHECK("sdt_layernorm_fwd");
  return SDT_OK;
}

// rows per block: one pass of the block's four waves (NR rows each in flight)
static int ln_rows_per_block(int C) { return C <= 512 ? 16 : (C <= 1024 ? 8 : 4); }

/* scratch of sdt_layernorm_bwd (required when dgamma / dbeta are wanted): one partial row [2C] per workgroup */
int64_t sdt_layernorm_bwd_workspace_bytes(int64_t M, int C) {
  if (M <= 0 || C <= 0) return 0;
  const int rpb = ln_rows_per_block(C);
  int64_t nblk = (M + rpb - 1) / rpb;
  if (nblk > 1024) nblk = 1024;
  return nblk * 2 * C * (int64_t)sizeof(float);
}

int sdt_layernorm_bwd(const uint16_t* x, const uint16_t* dy, const float* gamma, const float* mean_rstd, uint16_t* dx,
                      float* dgamma, float* dbeta, const uint16_t* dres, int64_t M, int C, int defer_param_grads, void* workspace,
                      int64_t workspace_bytes, hipStream_t stream) {
  SDT_CHECK_ARG(x && dy && gamma && mean_rstd && dx && M >= 0 && ((dgamma == nullptr) == (dbeta == nullptr)),
                "sdt_layernorm_bwd: null pointer");
  SDT_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 8 * 64 * LN_MAXV, "sdt_layernorm_bwd: C=%d unsupported", C);
  if (M == 0) return SDT_OK;
  SDT_CHECK_ARG(!dgamma || (workspace && workspace_bytes >= sdt_layernorm_bwd_workspace_bytes(M, C)),
                "sdt_layernorm_bwd: workspace of sdt_layernorm_bwd_workspace_bytes() needed for dgamma / dbeta");
  const int rpb = ln_rows_per_block(C);
  int nblk = (int)((M + rpb - 1) / rpb);  // one pass per wave: the row loop is latency-serial, so spread it wide
  if (nblk > 1024) nblk = 1024;
  float* part = dgamma ? (float*)workspace : nullptr;
  const size_t lds = sizeof(float) * 2 * C * 4;  // [4 waves][2][C]: <= 64 KiB at the largest supported C (2048)
  if (C <= 512)
    hipLaunchKernelGGL((ln_bwd_kernel<1, 4>), dim3(nblk), dim3(256), lds, stream, (const bf16_t*)x, (const bf16_t*)dy, gamma, mean_rstd, (bf16_t*)dx, dgamma, dbeta, part, (const bf16_t*)dres, (long)M, C);
  else if (C <= 1024)
    hipLaunchKernelGGL((ln_bwd_kernel<2, 2>), dim3(nblk), dim3(256), lds, stream, (const bf16_t*)x, (const bf16_t*)dy, gamma, mean_rstd, (bf16_t*)dx, dgamma, dbeta, part, (const bf16_t*)dres, (long)M, C);
  else
    hipLaunchKernelGGL((ln_bwd_kernel<4, 1>), dim3(nblk), dim3(256), lds, stream, (const bf16_t*)x, (const bf16_t*)dy, gamma, mean_rstd, (bf16_t*)dx, dgamma, dbeta, part, (const bf16_t*)dres, (long)M, C);
  if (dgamma && !defer_param_grads) launch_partial_reduce(part, dgamma, dbeta, nblk, C, stream);
  SDT_LAUNCH_CHECK("sdt_layernorm_bwd");
  return SDT_OK;
}

/* rows of partial sums sdt_layernorm_bwd(defer_param_grads = 1) leaves at the start of its workspace */
int64_t sdt_layernorm_bwd_partial_rows(int64_t M, int C) {
  if (M <= 0 || C <= 0) return 0;
  const int rpb = ln_rows_per_block(C);
  const int64_t nblk = (M + rpb - 1) / rpb;
  return nblk > 1024 ? 1024 : nblk;
}

int sdt_norm_param_grads_group_max(void) { return PR_GROUP_MAX; }

/* dgamma[i][c] += sum over the nrows[i] rows of partial[i][row][c], dbeta[i][c] += ... partial[i][row][C + c], for n norms in ONE
 * launch (the deferred tail of sdt_layernorm_bwd(defer_param_grads = 1)); one writer per element, rows added in a fixed order */
int sdt_norm_param_grads_group(const SdtNormGradJob* jobs, int n, hipStream_t stream) {
  SDT_CHECK_ARG(jobs && n > 0 && n <= PR_GROUP_MAX, "sdt_norm_param_grads_group: 1..%d jobs", PR_GROUP_MAX);
  PrGroupParams gp;
  gp.n = n;
  int wg = 0;
  for (int i = 0; i < n; ++i) {
    SDT_CHECK_ARG(jobs[i].partial && jobs[i].dgamma && jobs[i].dbeta && jobs[i].nrows > 0 && jobs[i].C > 0, "sdt_norm_param_grads_group: job %d: bad arguments", i);
    wg += sdt_ceil_div(2 * jobs[i].C, LN_PR_COLS);
    gp.wg_end[i] = wg;
    gp.nrows[i] = jobs[i].nrows;
    gp.C[i] = jobs[i].C;
    gp.partial[i] = jobs[i].partial;
    gp.dgamma[i] = jobs[i].dgamma;
    gp.dbeta[i] = jobs[i].dbeta;
  }
  hipLaunchKernelGGL(partial_reduce_group_kernel, dim3(wg), dim3(256), 0, stream, gp);
  SDT_LAUNCH_CHECK("sdt_norm_param_grads_group");
  return SDT_OK;
}

}  // extern "C"
