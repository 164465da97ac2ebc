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
//
// Staging: every 64-token tile of K/V (fwd, dq) or Q/dO (dkv) is copied global -> LDS by the LDS-DMA path
// (global_load_lds, 16 B per lane, no VGPR round trip) into ONE row-major image [64 tokens][DPP features] per
// tensor, double buffered so tile t+1 lands while tile t is in the MFMAs (one barrier per tile).  Both operand
// shapes are read from that image: [token][feature] fragments with ds_read_b128, and the transposed
// [feature][token] fragments with ds_read_b64_tr_b16.  The k-order of an accumulator-as-operand step is permuted
// (element j of lane half h is accumulator row 16s + 8(j>>2) + 4h + (j&3)), so the transposed fragment is two
// tr reads at token rows 16s+4h.. and 16s+8+4h...  Padding (features >= D, tokens >= N) is DMA'd from a zero page.
// The 16-byte chunks of a row are XOR-swizzled on the SOURCE side (LDS-DMA writes lane-linear) so that both read
// shapes are bank-conflict free.
#include <type_traits>

#include "sdt_common.h"

#define KT 64  // keys (or queries) staged per LDS tile
#define NEG_BIG -1.0e30f

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __attribute__((aligned(16))) unsigned int g_attn_zero16[4] = {0u, 0u, 0u, 0u};
// bf16 {1, 0, 0, 0, 0, 0, 0, 0}: the chunk that turns one padded V feature into a column of ones, so that the P.V
// MFMAs also deliver the softmax row sum (forward kernel, head dims with spare padding)
__device__ __attribute__((aligned(16))) unsigned int g_attn_one16[4] = {0x3f80u, 0u, 0u, 0u};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

__device__ __forceinline__ bf16x8_t cvt_frag(const float* p) {
  uint4 u;
  u.x = pack2bf(p[0], p[1]); u.y = pack2bf(p[2], p[3]); u.z = pack2bf(p[4], p[5]); u.w = pack2bf(p[6], p[7]);
  return __builtin_bit_cast(bf16x8_t, u);
}

// One staged image: [KT rows][DPP features] bf16, DPP in {64, 128, 256} (row = 128 / 256 / 512 bytes).
template <int DPP>
struct Img {
  static constexpr int RB = DPP * 2;            // bytes per row
  static constexpr int CPR = DPP / 8;           // 16-byte chunks per row
  static constexpr int BYTES = KT * RB;
  static constexpr int IPW = BYTES / 1024 / 4;  // LDS-DMA instructions per wave per image
  // chunk XOR of a row: 16 consecutive rows at one chunk (b128 fragment reads) and 4 consecutive rows x 4 chunks
  // (tr reads of one 32-lane half) both cover all 16 slots of the 256-byte bank window exactly once
  static __device__ __forceinline__ int swz(int row) {
    if (CPR == 8) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    return ((row & 3) << 2) | ((row >> 2) & 3);
  }
};

// per-lane source map of the image DMA: instruction i of wave w fills LDS bytes [(w*IPW+i)*1024, +1024)
template <int DPP>
struct TileDma {
  static constexpr int IPW = Img<DPP>::IPW, CPR = Img<DPP>::CPR;
  int row[IPW];  // token row inside the tile
  int col[IPW];  // first feature of the lane's chunk, -1 when it lies in the zero padding, -2 for the ones chunk
  __device__ __forceinline__ void init(int wave, int lane, int D, int ones_chunk = -1) {
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int L = (wave * IPW + i) * 64 + lane;
      const int r = L / CPR, slot = L % CPR;
      const int c = slot ^ Img<DPP>::swz(r);
      row[i] = r;
      col[i] = (c * 8 < D) ? c * 8 : (c == ones_chunk ? -2 : -1);
    }
  }
  // Tiles are issued in token order, KT apart: the lane's source pointers are kept and ADVANCED (one 64-bit add per piece per
  // tile) instead of being rebuilt from (token, stride, column) with a 64-bit multiply-add each time - the attention loops are
  // VALU-issue bound and the address arithmetic was a fifth of their vector instructions.
  const bf16_t* ptr[IPW];
  long step;
  // src: the (batch, head) base of the tensor; tok0: first token of the first tile that will be issued
  __device__ __forceinline__ void bind(const bf16_t* src, long ld, int tok0) {
    step = (long)KT * ld;
#pragma unroll
    for (int i = 0; i < IPW; ++i) ptr[i] = src + ((long)(tok0 + row[i]) * ld + (col[i] > 0 ? col[i] : 0));
  }
  // issues the tile the pointers stand at (its first token is tok_base: only the bounds test uses it) and advances them;
  // img: wave-uniform image base
  template <bool ONES = false>
  __device__ __forceinline__ void issue(int tok_base, int ntok, unsigned char* img, int wave_u) {
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_attn_zero16);
    const bf16_t* one = reinterpret_cast<const bf16_t*>(g_attn_one16);
    const bool full = tok_base + KT <= ntok;  // wave-uniform: only the last tile of a tensor tests tokens
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const bool tok_ok = full || tok_base + row[i] < ntok;
      const bool ok = col[i] >= 0 && tok_ok;
      const bf16_t* pad = (ONES && col[i] == -2 && tok_ok) ? one : zero;
      glds16(ok ? ptr[i] : pad, img + (wave_u * IPW + i) * 1024);
      ptr[i] += step;
    }
  }
};

// [token][feature] fragment: 8 features (chunk) of one token row
template <int DPP>
__device__ __forceinline__ bf16x8_t row_frag(const unsigned char* img, int row, int chunk) {
  return *reinterpret_cast<const bf16x8_t*>(img + row * Img<DPP>::RB + ((chunk ^ Img<DPP>::swz(row)) << 4));
}

// per-lane constants of the transposed fragment reads
template <int DPP>
struct TrLane {
  int off1, off2;  // byte offset of the lane's row (tile-local 4h+q, +8) with the `within` part folded in
  int cx1, cx2;    // (lane chunk) ^ swz(row)
  __device__ __forceinline__ void init(int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int r1 = 4 * (g >> 1) + q, r2 = r1 + 8;
    const int cl = 2 * (g & 1) + (pp >> 1), within = (pp & 1) * 8;
    off1 = r1 * Img<DPP>::RB + within;
    off2 = r2 * Img<DPP>::RB + within;
    cx1 = cl ^ Img<DPP>::swz(r1);
    cx2 = cl ^ Img<DPP>::swz(r2);
  }
  // A operand [32 features fb*32.. on the lanes][16 tokens tok0.. as k, accumulator-permuted order]; tok0 % 16 == 0
  __device__ __forceinline__ bf16x8_t frag(const unsigned char* img, int tok0, int fb) const {
    const unsigned char* a1 = img + tok0 * Img<DPP>::RB + off1 + (((fb * 4) ^ cx1) << 4);
    const unsigned char* a2 = img + tok0 * Img<DPP>::RB + off2 + (((fb * 4) ^ cx2) << 4);
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)a1);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)a2);
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
};

// all of this wave's LDS-DMA has landed, then every wave's (the barrier); also closes the reads of the previous tile
__device__ __forceinline__ void dma_join() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
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
  int dbg;  // developer ablation bits, honoured only in -DSDT_ATTN_DBG builds
  // dK/dV with the query range split over workgroups (few keys, many queries: cross-attention): each query chunk writes its partial
  // sums to its own fp32 slab [2][B][Nk][H*D] and attn_kv_finish_kernel adds the slabs in order and rounds them into dk / dv
  float* kv_ws;
  int qchunk, kblocks;
  // optional per-key weights w[Nk] > 0: P = softmax(s + ln w).  Emulates the key chunks of diffusers' memory-efficient attention,
  // whose last chunk is a clamped (overlapping) slice when the key count is not a multiple of the chunk: overlapped keys count twice
  const float* key_w;
};
#ifdef SDT_ATTN_DBG
#define ATTN_DBG(bit) (p.dbg & (bit))
#else
#define ATTN_DBG(bit) false
#endif

// ------------------------------------------------------------------------------------------ forward
// DPP: image pitch (features); NS = ceil(D/16) k-steps of q.k; NB = ceil(D/32) feature blocks of the output.
// MSUM: D <= NB*32 - 8, so V's padded feature NB*32-8 is staged as a column of ones and the softmax denominator is
// accumulated by the P.V MFMAs themselves (accumulator row 24 of the last block = register 12 of lanes 0..31).
// The running max is only raised when some row's maximum grew by more than 2^RESCALE_LOG2 (probabilities then stay below
// that bound instead of 1, which fp32 accumulation and the scale-free bf16 rounding of P do not notice), so most tiles skip
// the rescale of the output accumulators.
#define RESCALE_LOG2 8.0f
template <int DPP, int NS, int NB, bool MSUM>
__global__ void __launch_bounds__(256) attn_fwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using I = Img<DPP>;
  constexpr int STAGE = 2 * I::BYTES;  // K image | V image
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 31, fh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128;
  const int qi = q0 + wave * 32 + fr;  // this lane's query
  const bf16_t* qb = p.q + (long)b * p.bsq + h * p.D;
  const bf16_t* kb = p.k + (long)b * p.bsk + h * p.D;
  const bf16_t* vb = p.v + (long)b * p.bsv + h * p.D;

  TileDma<DPP> dma, dmav;
  dma.init(wave, lane, p.D);
  dmav.init(wave, lane, p.D, MSUM ? NB * 4 - 1 : -1);
  TrLane<DPP> tr;
  tr.init(lane);
  int kend = p.Nk;
  if (p.causal) kend = min(p.Nk, q0 + 128);  // keys beyond the block's last query are fully masked
  dma.bind(kb, p.ldk, 0);
  dmav.bind(vb, p.ldv, 0);
  dma.issue(0, p.Nk, smem, wave_u);
  dmav.template issue<MSUM>(0, p.Nk, smem + I::BYTES, wave_u);

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
  int cur = 0;
  for (int kbase = 0; kbase < kend; kbase += KT, cur ^= 1) {
    if (!ATTN_DBG(16)) dma_join();
    if (kbase + KT < kend && !ATTN_DBG(1)) {  // next tile flies under this tile's math
      dma.issue(kbase + KT, p.Nk, smem + (cur ^ 1) * STAGE, wave_u);
      dmav.template issue<MSUM>(kbase + KT, p.Nk, smem + (cur ^ 1) * STAGE + I::BYTES, wave_u);
    }
    const unsigned char* k_img = smem + cur * STAGE;
    const unsigned char* v_img = k_img + I::BYTES;
    f32x16_t st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) st[kt][e] = 0.f;
      if (!ATTN_DBG(8)) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const bf16x8_t kf = row_frag<DPP>(k_img, kt * 32 + fr, 2 * s + fh);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[kt], 0, 0, 0);
      }
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
    if (__any((mx - m_run) * p.scale2 > RESCALE_LOG2)) {  // wave-uniform; always taken on the first tile (m_run = -big)
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale2);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o_acc[i][e] *= alpha;
    }
    const float mneg = -m_run * p.scale2;
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float arg = fmaf(st[kt][e], p.scale2, mneg);
        const float pv = ATTN_DBG(2) ? arg : __builtin_amdgcn_exp2f(arg);
        st[kt][e] = pv;
      }
    if (p.key_w) {  // wave-uniform; the running max stays on the unweighted scores (weights are O(1))
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kbase + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
          st[kt][e] *= p.key_w[min(key, p.Nk - 1)];
        }
    }
    if (!MSUM) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) psum += st[kt][e];
      l_run += psum;
    }
    // O^T += V^T P^T
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float tmp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tmp[j] = st[kt][8 * s + j];
        const bf16x8_t pf = cvt_frag(tmp);
        if (!ATTN_DBG(4)) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const bf16x8_t vf = tr.frag(v_img, kt * 32 + 16 * s, i);
          o_acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o_acc[i], 0, 0, 0);
        }
        }
      }
  }
  float l_tot;
  if (MSUM) l_tot = __shfl(o_acc[NB - 1][12], fr, 64);  // ones-column row sum: accumulator row 24 lives in lanes 0..31
  else l_tot = l_run + __shfl_xor(l_run, 32, 64);
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

// ------------------------------------------------------------------------------------------ backward: dQ
template <int DPP, int NS, int NB>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using I = Img<DPP>;
  constexpr int STAGE = 2 * I::BYTES;  // K image | V image
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 31, fh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128;
  const int qi = q0 + wave * 32 + fr;
  const bf16_t* qb = p.q + (long)b * p.bsq + h * p.D;
  const bf16_t* dob = p.dout + (long)b * p.bsdo + h * p.D;
  const bf16_t* kb = p.k + (long)b * p.bsk + h * p.D;
  const bf16_t* vb = p.v + (long)b * p.bsv + h * p.D;

  TileDma<DPP> dma, dmav;
  dma.init(wave, lane, p.D);
  dmav.init(wave, lane, p.D);
  TrLane<DPP> tr;
  tr.init(lane);
  int kend = p.Nk;
  if (p.causal) kend = min(p.Nk, q0 + 128);
  dma.bind(kb, p.ldk, 0);
  dmav.bind(vb, p.ldv, 0);
  dma.issue(0, p.Nk, smem, wave_u);
  dmav.issue(0, p.Nk, smem + I::BYTES, wave_u);

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
  // delta[q] = sum_d dO[q][d] * O[q][d]: each lane already holds half of its query's dO row, so the dq kernel computes the
  // row dot itself and publishes it for the dK/dV kernel launched behind it (no separate pass over dO and O)
  float lse2 = 0.f, dlt = 0.f;
  if (qi < p.Nq) {
    lse2 = p.lse[((long)b * p.H + h) * p.Nq + qi];
    const bf16_t* orow = p.o + (long)b * p.bso + (long)qi * p.ldo + h * p.D;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int d0 = 16 * s + 8 * fh;
      if (d0 < p.D) {
        float fo[8], fd[8];
        unpack8(*reinterpret_cast<const uint4*>(orow + d0), fo);
        unpack8(__builtin_bit_cast(uint4, dof[s]), fd);
#pragma unroll
        for (int e = 0; e < 8; ++e) dlt += fo[e] * fd[e];
      }
    }
  }
  dlt += __shfl_xor(dlt, 32, 64);
  if (qi < p.Nq && fh == 0) const_cast<float*>(p.delta)[((long)b * p.H + h) * p.Nq + qi] = dlt;
  f32x16_t dq_acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq_acc[i][e] = 0.f;

  int cur = 0;
  for (int kbase = 0; kbase < kend; kbase += KT, cur ^= 1) {
    dma_join();
    if (kbase + KT < kend) {
      dma.issue(kbase + KT, p.Nk, smem + (cur ^ 1) * STAGE, wave_u);
      dmav.issue(kbase + KT, p.Nk, smem + (cur ^ 1) * STAGE + I::BYTES, wave_u);
    }
    const unsigned char* k_img = smem + cur * STAGE;
    const unsigned char* v_img = k_img + I::BYTES;
    const bool need_mask = (kbase + KT > p.Nk) || p.causal || (q0 + 128 > p.Nq);  // wave-uniform
    // two straight-line tile bodies, the wave-uniform choice made once per tile (see attn_bwd_dkv_kernel)
    auto tile = [&](auto masked, auto weighted) {
      constexpr bool MASK = decltype(masked)::value, KW = decltype(weighted)::value;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        f32x16_t st, dpt;
#pragma unroll
        for (int e = 0; e < 16; ++e) { st[e] = 0.f; dpt[e] = -dlt; }  // dP - delta comes out of the MFMA chain
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const bf16x8_t kf = row_frag<DPP>(k_img, kt * 32 + fr, 2 * s + fh);
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st, 0, 0, 0);
          const bf16x8_t vf = row_frag<DPP>(v_img, kt * 32 + fr, 2 * s + fh);
          dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[s], dpt, 0, 0, 0);
        }
        float ds[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float pv = __builtin_amdgcn_exp2f(fmaf(st[e], p.scale2, -lse2));
          const int key = kbase + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
          if (MASK) pv = (key >= p.Nk || (p.causal && key > qi) || qi >= p.Nq) ? 0.f : pv;
          if (KW) pv *= p.key_w[min(key, p.Nk - 1)];
          ds[e] = pv * dpt[e];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8_t dsf = cvt_frag(ds + 8 * s);
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            const bf16x8_t ktf = tr.frag(k_img, kt * 32 + 16 * s, i);
            dq_acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq_acc[i], 0, 0, 0);
          }
        }
      }
    };
    if (p.key_w) tile(std::true_type{}, std::true_type{});  // (cross-attention with repeated text keys: few, small tiles)
    else if (need_mask) tile(std::true_type{}, std::false_type{});
    else tile(std::false_type{}, std::false_type{});
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
template <int DPP, int NS, int NB, bool QSPLIT>
__global__ void __launch_bounds__(256) attn_bwd_dkv_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using I = Img<DPP>;
  constexpr int STAGE = 2 * I::BYTES + 2 * KT * 4;  // Q image | dO image | lse[KT] | delta[KT]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 31, fh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int b = blockIdx.z, h = blockIdx.y;
  const int kblk = QSPLIT ? (int)blockIdx.x % p.kblocks : (int)blockIdx.x;
  const int k0 = kblk * 128;
  const int ki = k0 + wave * 32 + fr;  // this lane's key
  const float kwt = p.key_w ? p.key_w[min(ki, p.Nk - 1)] : 1.f;
  const bf16_t* qb = p.q + (long)b * p.bsq + h * p.D;
  const bf16_t* dob = p.dout + (long)b * p.bsdo + h * p.D;
  const bf16_t* kb = p.k + (long)b * p.bsk + h * p.D;
  const bf16_t* vb = p.v + (long)b * p.bsv + h * p.D;
  const float* lse_g = p.lse + ((long)b * p.H + h) * p.Nq;
  const float* dlt_g = p.delta + ((long)b * p.H + h) * p.Nq;

  TileDma<DPP> dma, dmad;
  dma.init(wave, lane, p.D);
  dmad.init(wave, lane, p.D);
  TrLane<DPP> tr;
  tr.init(lane);
  int qstart = 0, qend = p.Nq;
  if (p.causal) qstart = (k0 / KT) * KT;  // queries before the block's first key see none of its keys
  if (QSPLIT) {
    qstart = ((int)blockIdx.x / p.kblocks) * p.qchunk;
    qend = min(p.Nq, qstart + p.qchunk);
  }
  dma.bind(qb, p.ldq, qstart);
  dmad.bind(dob, p.lddo, qstart);
  auto stage = [&](int qbase, unsigned char* st) {  // (called for qstart, qstart + KT, ... in order)
    dma.issue(qbase, p.Nq, st, wave_u);
    dmad.issue(qbase, p.Nq, st + I::BYTES, wave_u);
    // per-query softmax statistics: one 4-byte DMA per lane (wave 0: lse, wave 1: delta); rows past Nq read row Nq-1 (masked)
    if (wave_u < 2) {
      const int q = min(qbase + lane, p.Nq - 1);
      glds4((wave_u == 0 ? lse_g : dlt_g) + q, st + 2 * I::BYTES + wave_u * (KT * 4));
    }
  };
  if (qstart < qend) stage(qstart, smem);

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

  int cur = 0;
  for (int qbase = qstart; qbase < qend; qbase += KT, cur ^= 1) {
    dma_join();
    if (qbase + KT < qend) stage(qbase + KT, smem + (cur ^ 1) * STAGE);
    const unsigned char* q_img = smem + cur * STAGE;
    const unsigned char* do_img = q_img + I::BYTES;
    const float* lse_s = reinterpret_cast<const float*>(q_img + 2 * I::BYTES);
    const float* dlt_s = lse_s + KT;
    const bool need_mask = (qbase + KT > p.Nq) || (k0 + 128 > p.Nk) || p.causal;  // wave-uniform
    // The tile body exists twice - with and without the boundary / causal mask - and the wave-uniform choice is made ONCE per
    // tile: a test inside the per-element loop cuts the unrolled body into 32 three-instruction basic blocks (fma - exp - mul
    // with a branch each), which nothing can be scheduled across (every exp latency exposed, no MFMA beside the VALU work).
    auto tile = [&](auto masked, auto weighted) {
      constexpr bool MASK = decltype(masked)::value, KW = decltype(weighted)::value;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        f32x16_t sa, dpa;
        float lrow[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          // accumulator rows 8*g4 + 4*fh + 0..3 are four consecutive queries: one 16-byte read each of lse / delta;
          // -delta seeds the dP accumulator so that dP - delta comes out of the MFMA chain
          const float4 l4 = *reinterpret_cast<const float4*>(lse_s + qt * 32 + 8 * g4 + 4 * fh);
          const float4 d4 = *reinterpret_cast<const float4*>(dlt_s + qt * 32 + 8 * g4 + 4 * fh);
          lrow[4 * g4] = l4.x; lrow[4 * g4 + 1] = l4.y; lrow[4 * g4 + 2] = l4.z; lrow[4 * g4 + 3] = l4.w;
          dpa[4 * g4] = -d4.x; dpa[4 * g4 + 1] = -d4.y; dpa[4 * g4 + 2] = -d4.z; dpa[4 * g4 + 3] = -d4.w;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) sa[e] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const bf16x8_t qf = row_frag<DPP>(q_img, qt * 32 + fr, 2 * s + fh);
          sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[s], sa, 0, 0, 0);
          const bf16x8_t df = row_frag<DPP>(do_img, qt * 32 + fr, 2 * s + fh);
          dpa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, vf[s], dpa, 0, 0, 0);
        }
        float pr[16], ds[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float pv = __builtin_amdgcn_exp2f(fmaf(sa[e], p.scale2, -lrow[e]));
          if (KW) pv *= kwt;
          if (MASK) {
            const int q = qbase + qt * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
            pv = (q >= p.Nq || ki >= p.Nk || (p.causal && ki > q)) ? 0.f : pv;
          }
          pr[e] = pv;
          ds[e] = pv * dpa[e];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8_t pf = cvt_frag(pr + 8 * s);
          const bf16x8_t dsf = cvt_frag(ds + 8 * s);
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            const bf16x8_t dotf = tr.frag(do_img, qt * 32 + 16 * s, i);
            dv_acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf, pf, dv_acc[i], 0, 0, 0);
            const bf16x8_t qtf = tr.frag(q_img, qt * 32 + 16 * s, i);
            dk_acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dk_acc[i], 0, 0, 0);
          }
        }
      }
    };
    if (p.key_w) tile(std::true_type{}, std::true_type{});  // (cross-attention with repeated text keys: few, small tiles)
    else if (need_mask) tile(std::true_type{}, std::false_type{});
    else tile(std::false_type{}, std::false_type{});
  }
  if (QSPLIT) {  // this query chunk's partial sums go to its OWN fp32 slab (plain stores, one writer per element: no atomics, no
    // zero fill); attn_kv_finish_kernel adds the slabs in chunk order
    if (ki < p.Nk) {
      const long C = (long)p.H * p.D;
      const long slab = 2 * (long)p.B * p.Nk * C;
      float* wk = p.kv_ws + ((int)blockIdx.x / p.kblocks) * slab + ((long)b * p.Nk + ki) * C + h * p.D;
      float* wv = wk + (long)p.B * p.Nk * C;
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d = i * 32 + 8 * g4 + 4 * fh;
          if (d < p.D) {
            *reinterpret_cast<float4*>(wk + d) = make_float4(dk_acc[i][4 * g4] * p.scale, dk_acc[i][4 * g4 + 1] * p.scale,
                                                             dk_acc[i][4 * g4 + 2] * p.scale, dk_acc[i][4 * g4 + 3] * p.scale);
            *reinterpret_cast<float4*>(wv + d) = make_float4(dv_acc[i][4 * g4], dv_acc[i][4 * g4 + 1], dv_acc[i][4 * g4 + 2], dv_acc[i][4 * g4 + 3]);
          }
        }
    }
    return;
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

// fp32 slabs [nsplit][2][B][Nk][C] (one per query chunk) -> bf16 dk / dv (row strides lddk / lddv), 8 columns per thread; the
// slabs are added in chunk order: bitwise reproducible
__global__ void __launch_bounds__(256) attn_kv_finish_kernel(const AttnParams p, int nsplit) {
  const int C = p.H * p.D, cv = C >> 3;
  const long rows = (long)p.B * p.Nk, total = rows * cv;
  const long slab = 2 * rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cv;
    const int c = (int)(i - r * cv) * 8;
    const int b = (int)(r / p.Nk), ki = (int)(r - (long)b * p.Nk);
    float fk[8], fv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { fk[e] = 0.f; fv[e] = 0.f; }
    for (int sp = 0; sp < nsplit; ++sp) {
      const float* wk = p.kv_ws + sp * slab + r * C + c;
      const float* wv = wk + rows * C;
      const float4 a0 = *reinterpret_cast<const float4*>(wk), a1 = *reinterpret_cast<const float4*>(wk + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(wv), b1 = *reinterpret_cast<const float4*>(wv + 4);
      fk[0] += a0.x; fk[1] += a0.y; fk[2] += a0.z; fk[3] += a0.w; fk[4] += a1.x; fk[5] += a1.y; fk[6] += a1.z; fk[7] += a1.w;
      fv[0] += b0.x; fv[1] += b0.y; fv[2] += b0.z; fv[3] += b0.w; fv[4] += b1.x; fv[5] += b1.y; fv[6] += b1.z; fv[7] += b1.w;
    }
    *reinterpret_cast<uint4*>(p.dk + (long)b * p.bsdk + (long)ki * p.lddk + c) = pack8(fk);
    *reinterpret_cast<uint4*>(p.dv + (long)b * p.bsdv + (long)ki * p.lddv + c) = pack8(fv);
  }
}

// query split of the dK/dV pass: used when the key blocks alone leave most of the chip idle
static int attn_qsplits(int B, int H, int Nq, int Nk, int causal) {
  if (causal) return 1;
  const long base = (long)sdt_ceil_div(Nk, 128) * H * B;
  if (base >= 128 || Nq < 1024) return 1;
  int s = (int)((256 + base - 1) / base);
  const int max_s = Nq / 256;
  if (s > max_s) s = max_s;
  return s < 2 ? 1 : s;
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
  p->key_w = d->key_weight;
#ifdef SDT_ATTN_DBG
  p->dbg = getenv("SDT_ATTN_DBG") ? atoi(getenv("SDT_ATTN_DBG")) : 0;
#endif
  return SDT_OK;
}

template <typename K>
static void ensure_lds(K kernel, size_t lds, bool* done) {  // once per instantiation (also keeps it out of graph captures)
  if (!*done) {
    hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    *done = true;
  }
}
template <int DPP, int NS, int NB>
static void launch_fwd(const AttnParams& p, hipStream_t stream) {
  static_assert(NB * 32 <= DPP && NS * 16 <= DPP, "image pitch too small");
  static bool set_m = false, set_a = false;
  const size_t lds = (size_t)2 * 2 * Img<DPP>::BYTES;
  const dim3 grid(sdt_ceil_div(p.Nq, 128), p.H, p.B);
  if (p.D <= NB * 32 - 8) {  // a spare padded feature carries the ones column
    ensure_lds(attn_fwd_kernel<DPP, NS, NB, true>, lds, &set_m);
    hipLaunchKernelGGL((attn_fwd_kernel<DPP, NS, NB, true>), grid, dim3(256), lds, stream, p);
  } else {
    ensure_lds(attn_fwd_kernel<DPP, NS, NB, false>, lds, &set_a);
    hipLaunchKernelGGL((attn_fwd_kernel<DPP, NS, NB, false>), grid, dim3(256), lds, stream, p);
  }
}
template <int DPP, int NS, int NB>
static void launch_bwd(const AttnParams& p, hipStream_t stream) {
  static_assert(NB * 32 <= DPP && NS * 16 <= DPP, "image pitch too small");
  static bool set_q = false, set_kv = false;
  const size_t lds_dq = (size_t)2 * 2 * Img<DPP>::BYTES;
  const size_t lds_dkv = (size_t)2 * (2 * Img<DPP>::BYTES + 2 * KT * sizeof(float));
  static bool set_kvs = false;
  ensure_lds(attn_bwd_dq_kernel<DPP, NS, NB>, lds_dq, &set_q);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<DPP, NS, NB>), dim3(sdt_ceil_div(p.Nq, 128), p.H, p.B), dim3(256), lds_dq, stream, p);
  if (p.kv_ws) {
    AttnParams ps = p;
    const int splits = attn_qsplits(p.B, p.H, p.Nq, p.Nk, p.causal);
    ps.kblocks = sdt_ceil_div(p.Nk, 128);
    ps.qchunk = sdt_ceil_div(sdt_ceil_div(p.Nq, splits), KT) * KT;
    const int nsplit = sdt_ceil_div(p.Nq, ps.qchunk);
    ensure_lds(attn_bwd_dkv_kernel<DPP, NS, NB, true>, lds_dkv, &set_kvs);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DPP, NS, NB, true>), dim3(ps.kblocks * nsplit, p.H, p.B), dim3(256), lds_dkv, stream, ps);
    hipLaunchKernelGGL(attn_kv_finish_kernel, dim3(sdt_grid_1d((long)p.B * p.Nk * p.H * p.D / 8, 256, 1024)), dim3(256), 0, stream, ps, nsplit);
  } else {
    ensure_lds(attn_bwd_dkv_kernel<DPP, NS, NB, false>, lds_dkv, &set_kv);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DPP, NS, NB, false>), dim3(sdt_ceil_div(p.Nk, 128), p.H, p.B), dim3(256), lds_dkv, stream, p);
  }
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
  if (D <= 48) launch_fwd<64, 3, 2>(p, stream);
  else if (D <= 64) launch_fwd<64, 4, 2>(p, stream);
  else if (D <= 80) launch_fwd<128, 5, 3>(p, stream);
  else if (D <= 96) launch_fwd<128, 6, 3>(p, stream);
  else if (D <= 128) launch_fwd<128, 8, 4>(p, stream);
  else launch_fwd<256, 10, 5>(p, stream);
  SDT_LAUNCH_CHECK("sdt_attention_fwd");
  return SDT_OK;
}

/* scratch for sdt_attention_bwd: B*H*Nq floats (delta) + one 2*B*Nk*H*D-float slab per query chunk when the dK/dV pass splits the
 * query range (each chunk writes its own slab, a finishing pass adds them in order: no atomics) */
int64_t sdt_attention_bwd_workspace_bytes(const SdtAttnDesc* d) {
  if (!d || d->B <= 0 || d->H <= 0 || d->Nq <= 0 || d->Nk <= 0 || d->D <= 0) return 0;
  int64_t n = (int64_t)d->B * d->H * d->Nq;
  n = (n + 3) / 4 * 4;
  const int qs = attn_qsplits(d->B, d->H, d->Nq, d->Nk, d->causal);
  if (qs > 1) n += (int64_t)qs * 2 * d->B * d->Nk * d->H * d->D;
  return n * (int64_t)sizeof(float);
}

// workspace: sdt_attention_bwd_workspace_bytes(desc) bytes (at least B*H*Nq floats: without the rest the query split is not used)
int sdt_attention_bwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* out, const uint16_t* dout,
                      const float* lse, uint16_t* dq, uint16_t* dk, uint16_t* dv, float* delta_ws, int64_t workspace_bytes,
                      const SdtAttnDesc* desc, hipStream_t stream) {
  SDT_CHECK_ARG(q && k && v && out && dout && lse && dq && dk && dv && delta_ws, "sdt_attention_bwd: null pointer");
  SDT_CHECK_ARG(desc && workspace_bytes >= (int64_t)sizeof(float) * desc->B * desc->H * desc->Nq, "sdt_attention_bwd: workspace too small");
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
  if (attn_qsplits(p.B, p.H, p.Nq, p.Nk, p.causal) > 1 && workspace_bytes >= sdt_attention_bwd_workspace_bytes(desc))
    p.kv_ws = delta_ws + ((long)p.B * p.H * p.Nq + 3) / 4 * 4;
  const int D = p.D;
  if (D <= 48) launch_bwd<64, 3, 2>(p, stream);
  else if (D <= 64) launch_bwd<64, 4, 2>(p, stream);
  else if (D <= 80) launch_bwd<128, 5, 3>(p, stream);
  else if (D <= 96) launch_bwd<128, 6, 3>(p, stream);
  else if (D <= 128) launch_bwd<128, 8, 4>(p, stream);
  else launch_bwd<256, 10, 5>(p, stream);
  SDT_LAUNCH_CHECK("sdt_attention_bwd");
  return SDT_OK;
}

}  // extern "C"
