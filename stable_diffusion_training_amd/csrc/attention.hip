// Exact softmax attention (flash-style, no N x N materialisation) for gfx950, forward and backward.
// Replaces diffusers 0.21.4 attention_flax.jax_memory_efficient_attention as patched by
// key_chunk_patch.patch:1-9 (key chunk == all keys => plain exact softmax(q k^T / sqrt(d)) v, computed
// query-block by query-block), its transpose under jax.value_and_grad (training_utils.py:719-729), and the
// causal self-attention of transformers FlaxCLIPTextModel (training_utils.py:635-640).
//
// Layout: q/k/v/o are (B, N, H*D) row-major bf16 with explicit row strides; a head is the D-wide column
// slice h*D.. (no head transposes are ever materialised).  lse is (B, H, Nq) fp32 in the log2 domain:
// lse2 = max2 + log2(sum exp2(s2 - max2)), s2 = q.k * scale * log2(e).
//
// MFMA mapping (v_mfma_f32_32x32x16_bf16, one wave = 32 queries or 32 keys on the lanes):
//   fwd / dq : S^T = K Q^T   (keys on accumulator rows, query on the lane => row max / sum / LSE are per-lane
//              scalars, and the bf16-converted accumulator is directly the B operand of the next product)
//              O^T += V^T P^T,   dP^T = V dO^T,   dQ^T += K^T dS^T
//   dkv      : S = Q K^T, dP = dO V^T (key on the lane), dV^T += dO^T P, dK^T += Q^T dS
// The k-order of an accumulator-as-operand step is permuted: element j of lane half h is accumulator row
// 16s + 8(j>>2) + 4h + (j&3), so the LDS-side operand is fetched as two 8-byte reads at 16s+4h and 16s+8+4h
// from a [feature][token] (transposed) LDS image with a 136-byte pitch (conflict-free ds_read_b64).
#include "sdt_common.h"

#define KT 64            // keys (or queries) staged per LDS tile
#define TPITCH (KT + 4)  // pitch (elements) of transposed [feature][token] images: 136 B
#define NEG_BIG -1.0e30f

template <int DP16>
struct RowImg {  // row-major [token][feature] image, pitch DP16+8 elements (odd multiple of 16 B => conflict-free b128)
  static constexpr int PITCH = DP16 + 8;
};

__device__ __forceinline__ bf16x8_t cvt_frag(const float* p) {
  uint4 u;
  u.x = pack2bf(p[0], p[1]); u.y = pack2bf(p[2], p[3]); u.z = pack2bf(p[4], p[5]); u.w = pack2bf(p[6], p[7]);
  return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* img_row, int tok0, int fh) {
  // two 8-byte reads: tokens tok0+4h..+3 and tok0+8+4h..+3 of one feature row
  uint2 a = *reinterpret_cast<const uint2*>(img_row + tok0 + 4 * fh);
  uint2 b = *reinterpret_cast<const uint2*>(img_row + tok0 + 8 + 4 * fh);
  uint4 u = make_uint4(a.x, a.y, b.x, b.y);
  return __builtin_bit_cast(bf16x8_t, u);
}

__device__ __forceinline__ uint4 keep16(uint4 v, bool k) {
  v.x = k ? v.x : 0u; v.y = k ? v.y : 0u; v.z = k ? v.z : 0u; v.w = k ? v.w : 0u;
  return v;
}

// stage KT rows x D features of src (row stride ld) into a row-major image [KT][PITCH], zero padded.
// All global loads are issued first from clamped (always valid) addresses, then zero-selected and stored: a branch
// around each load would make hipcc wait per element.
template <int DP16>
__device__ __forceinline__ void stage_rows(bf16_t* img, const bf16_t* src, long ld, int tok_base, int ntok_total, int D) {
  constexpr int PITCH = DP16 + 8, CH = DP16 / 8, ITEMS = KT * CH, ITERS = (ITEMS + 255) / 256;
  uint4 v[ITERS];
  bool ok[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = threadIdx.x + it * 256;
    const int tok = idx / CH, ch = idx - tok * CH;
    ok[it] = idx < ITEMS && tok_base + tok < ntok_total && ch * 8 < D;
    v[it] = *reinterpret_cast<const uint4*>(ok[it] ? src + (long)(tok_base + tok) * ld + ch * 8 : src);
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = threadIdx.x + it * 256;
    const int tok = idx / CH, ch = idx - tok * CH;
    if (idx < ITEMS) *reinterpret_cast<uint4*>(img + tok * PITCH + ch * 8) = keep16(v[it], ok[it]);
  }
}
// stage transposed: image [DP32 features][TPITCH tokens]; each work item = 4 tokens x 8 features
template <int DP32>
__device__ __forceinline__ void stage_transposed(bf16_t* img, const bf16_t* src, long ld, int tok_base, int ntok_total, int D) {
  constexpr int CH = DP32 / 8, ITEMS = (KT / 4) * CH, ITERS = (ITEMS + 255) / 256;
  uint4 v[ITERS][4];
  bool ok[ITERS][4];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = threadIdx.x + it * 256;
    const int tg = idx % (KT / 4), ch = idx / (KT / 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tok = tok_base + 4 * tg + i;
      ok[it][i] = idx < ITEMS && tok < ntok_total && ch * 8 < D;
      v[it][i] = *reinterpret_cast<const uint4*>(ok[it][i] ? src + (long)tok * ld + ch * 8 : src);
    }
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = threadIdx.x + it * 256;
    const int tg = idx % (KT / 4), ch = idx / (KT / 4);
    if (idx < ITEMS) {
      unsigned w[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint4 t = keep16(v[it][i], ok[it][i]);
        w[i][0] = t.x; w[i][1] = t.y; w[i][2] = t.z; w[i][3] = t.w;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint2 lo, hi;
        lo.x = (w[0][q] & 0xffffu) | (w[1][q] << 16); lo.y = (w[2][q] & 0xffffu) | (w[3][q] << 16);
        hi.x = (w[0][q] >> 16) | (w[1][q] & 0xffff0000u); hi.y = (w[2][q] >> 16) | (w[3][q] & 0xffff0000u);
        *reinterpret_cast<uint2*>(img + (ch * 8 + 2 * q) * TPITCH + 4 * tg) = lo;
        *reinterpret_cast<uint2*>(img + (ch * 8 + 2 * q + 1) * TPITCH + 4 * tg) = hi;
      }
    }
  }
}

struct AttnParams {
  const bf16_t *q, *k, *v, *o, *dout;
  bf16_t *out, *dq, *dk, *dv;
  float* lse;          // (B,H,Nq) log2 domain
  const float* delta;  // (B,H,Nq)
  int B, H, Nq, Nk, D;
  long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;  // row strides (elements)
  long bsq, bsk, bsv, bso, bsdo, bsdq, bsdk, bsdv;  // batch strides (elements)
  float scale2;  // scale * log2(e)
  float scale;
  int causal;
};

// ------------------------------------------------------------------------------------------ forward
template <int DP16, int DP32>
__global__ void __launch_bounds__(256) attn_fwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int PITCH = DP16 + 8, NS = DP16 / 16, NB = DP32 / 32;
  bf16_t* k_img = reinterpret_cast<bf16_t*>(smem_raw);   // [KT][PITCH]
  bf16_t* vt_img = k_img + KT * PITCH;                   // [DP32][TPITCH]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 31, fh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128;
  const int qi = q0 + wave * 32 + fr;  // this lane's query
  const bf16_t* qb = p.q + (long)b * p.bsq + h * p.D;
  const bf16_t* kb = p.k + (long)b * p.bsk + h * p.D;
  const bf16_t* vb = p.v + (long)b * p.bsv + h * p.D;

  bf16x8_t qf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    uint4 v = make_uint4(0, 0, 0, 0);
    const int d0 = 16 * s + 8 * fh;
    if (qi < p.Nq && d0 < p.D) v = *reinterpret_cast<const uint4*>(qb + (long)qi * p.ldq + d0);
    qf[s] = __builtin_bit_cast(bf16x8_t, v);
  }
  f32x16_t o_acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[i][e] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;

  int kend = p.Nk;
  if (p.causal) kend = min(p.Nk, q0 + 128);  // keys beyond the block's last query are fully masked
  for (int kbase = 0; kbase < kend; kbase += KT) {
    __syncthreads();
    stage_rows<DP16>(k_img, kb, p.ldk, kbase, p.Nk, p.D);
    stage_transposed<DP32>(vt_img, vb, p.ldv, kbase, p.Nk, p.D);
    __syncthreads();
    f32x16_t st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) st[kt][e] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(k_img + (kt * 32 + fr) * PITCH + 16 * s + 8 * fh);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[kt], 0, 0, 0);
      }
    }
    // running max on the RAW scores (scale2 > 0), scale folded into the exp2 argument; masks only where needed
    const bool need_mask = (kbase + KT > p.Nk) || p.causal;  // wave-uniform
    float mx = NEG_BIG;
    if (need_mask) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kbase + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
          if (key >= p.Nk || (p.causal && key > qi)) st[kt][e] = NEG_BIG;
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[kt][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale2);
    m_run = m_new;
    const float mneg = -m_new * p.scale2;
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(st[kt][e], p.scale2, mneg));
        st[kt][e] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) o_acc[i][e] *= alpha;
    // O^T += V^T P^T
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float tmp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tmp[j] = st[kt][8 * s + j];
        const bf16x8_t pf = cvt_frag(tmp);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const bf16x8_t vf = tr_frag(vt_img + (i * 32 + fr) * TPITCH, kt * 32 + 16 * s, fh);
          o_acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o_acc[i], 0, 0, 0);
        }
      }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv_l = 1.f / l_tot;
  if (qi < p.Nq) {
    if (fh == 0 && p.lse) p.lse[((long)b * p.H + h) * p.Nq + qi] = m_run * p.scale2 + log2f(l_tot);
    bf16_t* ob = p.out + (long)b * p.bso + (long)qi * p.ldo + h * p.D;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d = i * 32 + 8 * g4 + 4 * fh;
        if (d < p.D) {
          uint2 pk;
          pk.x = pack2bf(o_acc[i][4 * g4] * inv_l, o_acc[i][4 * g4 + 1] * inv_l);
          pk.y = pack2bf(o_acc[i][4 * g4 + 2] * inv_l, o_acc[i][4 * g4 + 3] * inv_l);
          *reinterpret_cast<uint2*>(ob + d) = pk;
        }
      }
  }
}

// delta[b][h][q] = sum_d dO[q][d] * O[q][d]
__global__ void __launch_bounds__(256) attn_delta_kernel(const AttnParams p) {
  const long total = (long)p.B * p.H * p.Nq;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % p.Nq);
    const long bh = i / p.Nq;
    const int h = (int)(bh % p.H), b = (int)(bh / p.H);
    const bf16_t* o = p.o + (long)b * p.bso + (long)q * p.ldo + h * p.D;
    const bf16_t* d = p.dout + (long)b * p.bsdo + (long)q * p.lddo + h * p.D;
    float acc = 0.f;
    for (int c = 0; c < p.D; c += 8) {
      float f[8], g[8];
      unpack8(*reinterpret_cast<const uint4*>(o + c), f);
      unpack8(*reinterpret_cast<const uint4*>(d + c), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += f[e] * g[e];
    }
    const_cast<float*>(p.delta)[i] = acc;
  }
}

// ------------------------------------------------------------------------------------------ backward: dQ
template <int DP16, int DP32>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int PITCH = DP16 + 8, NS = DP16 / 16, NB = DP32 / 32;
  bf16_t* k_img = reinterpret_cast<bf16_t*>(smem_raw);  // [KT][PITCH]
  bf16_t* v_img = k_img + KT * PITCH;                   // [KT][PITCH]
  bf16_t* kt_img = v_img + KT * PITCH;                  // [DP32][TPITCH]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 31, fh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128;
  const int qi = q0 + wave * 32 + fr;
  const bf16_t* qb = p.q + (long)b * p.bsq + h * p.D;
  const bf16_t* dob = p.dout + (long)b * p.bsdo + h * p.D;
  const bf16_t* kb = p.k + (long)b * p.bsk + h * p.D;
  const bf16_t* vb = p.v + (long)b * p.bsv + h * p.D;

  bf16x8_t qf[NS], dof[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    uint4 v = make_uint4(0, 0, 0, 0), w = make_uint4(0, 0, 0, 0);
    const int d0 = 16 * s + 8 * fh;
    if (qi < p.Nq && d0 < p.D) {
      v = *reinterpret_cast<const uint4*>(qb + (long)qi * p.ldq + d0);
      w = *reinterpret_cast<const uint4*>(dob + (long)qi * p.lddo + d0);
    }
    qf[s] = __builtin_bit_cast(bf16x8_t, v);
    dof[s] = __builtin_bit_cast(bf16x8_t, w);
  }
  float lse2 = 0.f, dlt = 0.f;
  if (qi < p.Nq) {
    lse2 = p.lse[((long)b * p.H + h) * p.Nq + qi];
    dlt = p.delta[((long)b * p.H + h) * p.Nq + qi];
  }
  f32x16_t dq_acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq_acc[i][e] = 0.f;

  int kend = p.Nk;
  if (p.causal) kend = min(p.Nk, q0 + 128);
  for (int kbase = 0; kbase < kend; kbase += KT) {
    __syncthreads();
    stage_rows<DP16>(k_img, kb, p.ldk, kbase, p.Nk, p.D);
    stage_rows<DP16>(v_img, vb, p.ldv, kbase, p.Nk, p.D);
    stage_transposed<DP32>(kt_img, kb, p.ldk, kbase, p.Nk, p.D);
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x16_t st, dpt;
#pragma unroll
      for (int e = 0; e < 16; ++e) { st[e] = 0.f; dpt[e] = 0.f; }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(k_img + (kt * 32 + fr) * PITCH + 16 * s + 8 * fh);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st, 0, 0, 0);
        bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(v_img + (kt * 32 + fr) * PITCH + 16 * s + 8 * fh);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[s], dpt, 0, 0, 0);
      }
      float ds[16];
      const bool need_mask = (kbase + KT > p.Nk) || p.causal || (q0 + 128 > p.Nq);  // wave-uniform
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float pv = __builtin_amdgcn_exp2f(fmaf(st[e], p.scale2, -lse2));
        if (need_mask) {
          const int key = kbase + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
          if (key >= p.Nk || (p.causal && key > qi) || qi >= p.Nq) pv = 0.f;
        }
        ds[e] = pv * (dpt[e] - dlt);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8_t dsf = cvt_frag(ds + 8 * s);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const bf16x8_t ktf = tr_frag(kt_img + (i * 32 + fr) * TPITCH, kt * 32 + 16 * s, fh);
          dq_acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq_acc[i], 0, 0, 0);
        }
      }
    }
  }
  if (qi < p.Nq) {
    bf16_t* ob = p.dq + (long)b * p.bsdq + (long)qi * p.lddq + h * p.D;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d = i * 32 + 8 * g4 + 4 * fh;
        if (d < p.D) {
          uint2 pk;
          pk.x = pack2bf(dq_acc[i][4 * g4] * p.scale, dq_acc[i][4 * g4 + 1] * p.scale);
          pk.y = pack2bf(dq_acc[i][4 * g4 + 2] * p.scale, dq_acc[i][4 * g4 + 3] * p.scale);
          *reinterpret_cast<uint2*>(ob + d) = pk;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
template <int DP16, int DP32>
__global__ void __launch_bounds__(256) attn_bwd_dkv_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int PITCH = DP16 + 8, NS = DP16 / 16, NB = DP32 / 32;
  bf16_t* q_img = reinterpret_cast<bf16_t*>(smem_raw);  // [KT][PITCH]
  bf16_t* do_img = q_img + KT * PITCH;                  // [KT][PITCH]
  bf16_t* qt_img = do_img + KT * PITCH;                 // [DP32][TPITCH]
  bf16_t* dot_img = qt_img + DP32 * TPITCH;             // [DP32][TPITCH]
  float* lse_s = reinterpret_cast<float*>(dot_img + DP32 * TPITCH);  // [KT]
  float* dlt_s = lse_s + KT;                                         // [KT]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 31, fh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int k0 = blockIdx.x * 128;
  const int ki = k0 + wave * 32 + fr;  // this lane's key
  const bf16_t* qb = p.q + (long)b * p.bsq + h * p.D;
  const bf16_t* dob = p.dout + (long)b * p.bsdo + h * p.D;
  const bf16_t* kb = p.k + (long)b * p.bsk + h * p.D;
  const bf16_t* vb = p.v + (long)b * p.bsv + h * p.D;
  const float* lse_g = p.lse + ((long)b * p.H + h) * p.Nq;
  const float* dlt_g = p.delta + ((long)b * p.H + h) * p.Nq;

  bf16x8_t kf[NS], vf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    uint4 v = make_uint4(0, 0, 0, 0), w = make_uint4(0, 0, 0, 0);
    const int d0 = 16 * s + 8 * fh;
    if (ki < p.Nk && d0 < p.D) {
      v = *reinterpret_cast<const uint4*>(kb + (long)ki * p.ldk + d0);
      w = *reinterpret_cast<const uint4*>(vb + (long)ki * p.ldv + d0);
    }
    kf[s] = __builtin_bit_cast(bf16x8_t, v);
    vf[s] = __builtin_bit_cast(bf16x8_t, w);
  }
  f32x16_t dk_acc[NB], dv_acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk_acc[i][e] = 0.f; dv_acc[i][e] = 0.f; }

  int qstart = 0;
  if (p.causal) qstart = (k0 / KT) * KT;  // queries before the block's first key see none of its keys
  for (int qbase = qstart; qbase < p.Nq; qbase += KT) {
    __syncthreads();
    stage_rows<DP16>(q_img, qb, p.ldq, qbase, p.Nq, p.D);
    stage_rows<DP16>(do_img, dob, p.lddo, qbase, p.Nq, p.D);
    stage_transposed<DP32>(qt_img, qb, p.ldq, qbase, p.Nq, p.D);
    stage_transposed<DP32>(dot_img, dob, p.lddo, qbase, p.Nq, p.D);
    if (threadIdx.x < KT) {
      const int q = qbase + threadIdx.x;
      lse_s[threadIdx.x] = (q < p.Nq) ? lse_g[q] : 0.f;
      dlt_s[threadIdx.x] = (q < p.Nq) ? dlt_g[q] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x16_t sa, dpa;
#pragma unroll
      for (int e = 0; e < 16; ++e) { sa[e] = 0.f; dpa[e] = 0.f; }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(q_img + (qt * 32 + fr) * PITCH + 16 * s + 8 * fh);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[s], sa, 0, 0, 0);
        bf16x8_t df = *reinterpret_cast<const bf16x8_t*>(do_img + (qt * 32 + fr) * PITCH + 16 * s + 8 * fh);
        dpa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, vf[s], dpa, 0, 0, 0);
      }
      float pr[16], ds[16];
      const bool need_mask = (qbase + KT > p.Nq) || (k0 + 128 > p.Nk) || p.causal;  // wave-uniform
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ql = qt * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        float pv = __builtin_amdgcn_exp2f(fmaf(sa[e], p.scale2, -lse_s[ql]));
        if (need_mask) {
          const int q = qbase + ql;
          if (q >= p.Nq || ki >= p.Nk || (p.causal && ki > q)) pv = 0.f;
        }
        pr[e] = pv;
        ds[e] = pv * (dpa[e] - dlt_s[ql]);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8_t pf = cvt_frag(pr + 8 * s);
        const bf16x8_t dsf = cvt_frag(ds + 8 * s);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const bf16x8_t dotf = tr_frag(dot_img + (i * 32 + fr) * TPITCH, qt * 32 + 16 * s, fh);
          dv_acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf, pf, dv_acc[i], 0, 0, 0);
          const bf16x8_t qtf = tr_frag(qt_img + (i * 32 + fr) * TPITCH, qt * 32 + 16 * s, fh);
          dk_acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dk_acc[i], 0, 0, 0);
        }
      }
    }
  }
  if (ki < p.Nk) {
    bf16_t* dkb = p.dk + (long)b * p.bsdk + (long)ki * p.lddk + h * p.D;
    bf16_t* dvb = p.dv + (long)b * p.bsdv + (long)ki * p.lddv + h * p.D;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d = i * 32 + 8 * g4 + 4 * fh;
        if (d < p.D) {
          uint2 pk;
          pk.x = pack2bf(dk_acc[i][4 * g4] * p.scale, dk_acc[i][4 * g4 + 1] * p.scale);
          pk.y = pack2bf(dk_acc[i][4 * g4 + 2] * p.scale, dk_acc[i][4 * g4 + 3] * p.scale);
          *reinterpret_cast<uint2*>(dkb + d) = pk;
          pk.x = pack2bf(dv_acc[i][4 * g4], dv_acc[i][4 * g4 + 1]);
          pk.y = pack2bf(dv_acc[i][4 * g4 + 2], dv_acc[i][4 * g4 + 3]);
          *reinterpret_cast<uint2*>(dvb + d) = pk;
        }
      }
  }
}

// ================================================================== C ABI
static int attn_fill(AttnParams* p, const SdtAttnDesc* d, const char* name) {
  SDT_CHECK_ARG(d, "%s: null descriptor", name);
  SDT_CHECK_ARG(d->B > 0 && d->H > 0 && d->Nq > 0 && d->Nk > 0 && d->D > 0 && d->B <= 65535 && d->H <= 65535,
                "%s: bad shape B=%d H=%d Nq=%d Nk=%d D=%d", name, d->B, d->H, d->Nq, d->Nk, d->D);
  SDT_CHECK_ARG(d->D % 8 == 0 && d->D <= 160, "%s: head dim %d must be a multiple of 8 and <= 160", name, d->D);
  SDT_CHECK_ARG(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldo % 8 == 0, "%s: row strides must be multiples of 8", name);
  p->B = d->B; p->H = d->H; p->Nq = d->Nq; p->Nk = d->Nk; p->D = d->D;
  p->ldq = d->ldq; p->ldk = d->ldk; p->ldv = d->ldv; p->ldo = d->ldo;
  p->bsq = (long)d->Nq * d->ldq; p->bsk = (long)d->Nk * d->ldk; p->bsv = (long)d->Nk * d->ldv; p->bso = (long)d->Nq * d->ldo;
  p->lddo = d->ldo; p->bsdo = p->bso;
  p->lddq = d->ldq; p->bsdq = p->bsq; p->lddk = d->ldk; p->bsdk = p->bsk; p->lddv = d->ldv; p->bsdv = p->bsv;
  p->scale = d->scale;
  p->scale2 = d->scale * 1.4426950408889634f;
  p->causal = d->causal;
  return SDT_OK;
}

template <int DP16, int DP32>
static void launch_fwd(const AttnParams& p, hipStream_t stream) {
  const size_t lds = (size_t)(KT * (DP16 + 8) + DP32 * TPITCH) * 2;
  hipFuncSetAttribute((const void*)attn_fwd_kernel<DP16, DP32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((attn_fwd_kernel<DP16, DP32>), dim3(sdt_ceil_div(p.Nq, 128), p.H, p.B), dim3(256), lds, stream, p);
}
template <int DP16, int DP32>
static void launch_bwd(const AttnParams& p, hipStream_t stream) {
  const size_t lds_dq = (size_t)(2 * KT * (DP16 + 8) + DP32 * TPITCH) * 2;
  const size_t lds_dkv = (size_t)(2 * KT * (DP16 + 8) + 2 * DP32 * TPITCH) * 2 + 2 * KT * sizeof(float);
  hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<DP16, DP32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq);
  hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<DP16, DP32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dkv);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<DP16, DP32>), dim3(sdt_ceil_div(p.Nq, 128), p.H, p.B), dim3(256), lds_dq, stream, p);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<DP16, DP32>), dim3(sdt_ceil_div(p.Nk, 128), p.H, p.B), dim3(256), lds_dkv, stream, p);
}

extern "C" {

int sdt_attention_fwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out, float* lse,
                      const SdtAttnDesc* desc, hipStream_t stream) {
  SDT_CHECK_ARG(q && k && v && out, "sdt_attention_fwd: null pointer");
  AttnParams p = {};
  int rc = attn_fill(&p, desc, "sdt_attention_fwd");
  if (rc) return rc;
  SDT_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, "sdt_attention_fwd: pointers must be 16-byte aligned");
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.out = (bf16_t*)out; p.lse = lse;
  const int D = p.D;
  if (D <= 48) launch_fwd<48, 64>(p, stream);
  else if (D <= 64) launch_fwd<64, 64>(p, stream);
  else if (D <= 80) launch_fwd<80, 96>(p, stream);
  else if (D <= 96) launch_fwd<96, 96>(p, stream);
  else if (D <= 128) launch_fwd<128, 128>(p, stream);
  else launch_fwd<160, 160>(p, stream);
  SDT_LAUNCH_CHECK("sdt_attention_fwd");
  return SDT_OK;
}

// delta_ws: workspace of B*H*Nq floats
int sdt_attention_bwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* out, const uint16_t* dout,
                      const float* lse, uint16_t* dq, uint16_t* dk, uint16_t* dv, float* delta_ws, const SdtAttnDesc* desc,
                      hipStream_t stream) {
  SDT_CHECK_ARG(q && k && v && out && dout && lse && dq && dk && dv && delta_ws, "sdt_attention_bwd: null pointer");
  AttnParams p = {};
  int rc = attn_fill(&p, desc, "sdt_attention_bwd");
  if (rc) return rc;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (const bf16_t*)out;
  p.dout = (const bf16_t*)dout; p.lse = const_cast<float*>(lse); p.delta = delta_ws;
  p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
  if (desc->ldgrad_q) { p.lddq = desc->ldgrad_q; p.bsdq = (long)p.Nq * p.lddq; }
  if (desc->ldgrad_k) { p.lddk = desc->ldgrad_k; p.bsdk = (long)p.Nk * p.lddk; }
  if (desc->ldgrad_v) { p.lddv = desc->ldgrad_v; p.bsdv = (long)p.Nk * p.lddv; }
  if (desc->ld_dout) { p.lddo = desc->ld_dout; p.bsdo = (long)p.Nq * p.lddo; }
  hipLaunchKernelGGL(attn_delta_kernel, dim3(sdt_grid_1d((long)p.B * p.H * p.Nq, 256)), dim3(256), 0, stream, p);
  const int D = p.D;
  if (D <= 48) launch_bwd<48, 64>(p, stream);
  else if (D <= 64) launch_bwd<64, 64>(p, stream);
  else if (D <= 80) launch_bwd<80, 96>(p, stream);
  else if (D <= 96) launch_bwd<96, 96>(p, stream);
  else if (D <= 128) launch_bwd<128, 128>(p, stream);
  else launch_bwd<160, 160>(p, stream);
  SDT_LAUNCH_CHECK("sdt_attention_bwd");
  return SDT_OK;
}

}  // extern "C"
