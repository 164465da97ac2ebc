// MFMA (v_mfma_f32_32x32x16_bf16) GEMM family for gfx950: every dense contraction of the UNet / VAE / CLIP
// graphs outside the attention core goes through the kernels in this file.
//
//  gemm_nt_kernel      : C[M,N] (bf16) = A_g[M,K] * Bt[N,K]^T  (+bias[N]) (+rowbias[m/rpb][N], row pitch ld_rowbias) (+residual[M,N])
//      A_g is a plain row-major matrix (Linear fwd / dgrad, 1x1 conv; several reduction segments for Dense layers that share
//      an input) or an im2col view gathered on the fly from an NHWC tensor (strided / asymmetric-pad convs, strided dgrad).
//  conv3x3_halo_kernel : the same contraction for 3x3 / stride 1 / pad 1 convolutions (fprop and dgrad): a 256-pixel x
//      128-channel tile whose input halo is staged ONCE per 64-channel chunk for all nine taps.
//  gemm_tn_kernel      : dW[tap][K1][N] (fp32, atomically accumulated, split over M) += A_g[M,K1]^T * dY[M,N]
//      (Linear / conv weight gradients + bias gradients, written straight into the Flax [in,out] / HWIO gradient layout).
//  conv_wgrad3_kernel  : the weight gradient of a 3x3 / stride 1 / pad 1 convolution, three taps per workgroup.
//
// Common ground: BK = 64; operands are staged global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction,
// bank swizzle on the SOURCE address, padding from a 16-byte zero page) into rings whose later stages stay in flight across
// the barrier (counted s_waitcnt vmcnt + raw s_barrier); row-major images are read with ds_read_b128 ([row][k] fragments) or
// ds_read_b64_tr_b16 (k-major operands); where hipcc would drain the ring in front of an LDS read it can see, the reads are
// inline asm with hand-counted lgkmcnt.  Split-K: fp32 atomics into a caller workspace, the last-arriving split finishes the
// tile and leaves the workspace zero.  Epilogues can accumulate the GroupNorm statistics of what they store (gn_stats).
//
// Reference call sites these replace (all lowered by XLA in the reference): flax nn.Conv / nn.Dense inside
// diffusers 0.21.4 unet_2d_condition_flax.py, unet_2d_blocks_flax.py, attention_flax.py, resnet_flax.py,
// vae_flax.py and transformers modeling_flax_clip.py, reached from training_utils.py:574-579, 635-640, 678-684,
// and their transposes under jax.value_and_grad (training_utils.py:719-729).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "sdt_common.h"

#define BK 64
#define LDS_ROW_BYTES 128

struct FastDiv {
  unsigned mp, l, d;
};
static FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.l = l;
  f.mp = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ unsigned fd_div(unsigned n, const FastDiv& f) {
  unsigned t = __umulhi(f.mp, n);
  return (unsigned)(((unsigned long long)t + n) >> f.l);
}

enum { GATHER_PLAIN = 0, GATHER_FPROP = 1, GATHER_DGRAD = 2 };

struct GatherDesc {
  int mode;
  int IH, IW;  // spatial dims of the tensor rows are gathered from
  int OH, OW;  // spatial grid the GEMM rows enumerate (M = B*OH*OW)
  int KH, KW, stride, pad_t, pad_l;
  FastDiv div_ohw, div_ow;
};

struct GemmNtParams {
  int nst;  // LDS ring depth of this launch (>= NtCfg::NST; nt_ring_depth)
  const bf16_t* A;
  const bf16_t* Bt;
  bf16_t* C;
  const float* bias;
  const bf16_t* rowbias;
  const bf16_t* residual;
  int M, N, Kc, taps;
  int lda, ldb, ldc, ldres;
  int ldrb;  // row pitch of rowbias (elements)
  long b_tap_stride;
  int rows_per_batch;
  int tiles_m, tiles_n;
  int nfast;             // tile order: 0 = consecutive tiles walk M (one weight column tile stays hot in L2, the rows stream),
                         // 1 = consecutive tiles walk N (a row panel is fetched once and meets every weight column tile): nt_tile_order
  int ksteps_per_split;  // split-K: blockIdx.y owns K-steps [y*ksteps_per_split, ...)
  unsigned char* slab;   // split-K: per-workgroup fp32 partial tiles, [tile][split][TnSlab bytes] (scratch, needs no initialisation)
  int* tile_cnt;         // split-K: per-tile arrival counters (zero on entry, zero on exit)
  float* gn_stats;       // optional: [batch][gn_parts][gn_groups][2] partial {sum, sum of squares} of the bf16 outputs (gn_tile_flush)
  int gn_groups, gn_parts;
  // 3x3 / stride 1 / pad 1 halo kernel: a tile is NI images x TH rows x TW columns = 256 output pixels
  int cv_ni, cv_th, cv_tw, cv_ltw, cv_lth;  // (log2 of TW, TH)
  int cv_tiles_x, cv_tiles_y, cv_chunks_per_split;
  FastDiv cv_div_w2, cv_div_himg, cv_div_tx, cv_div_ty;  // / (TW+2), / ((TH+2)(TW+2)), / tiles_x, / tiles_y
  int dbg;               // developer ablation bits: honoured only by -DSDT_NT_DBG builds (SDT_HIPCC_EXTRA), see NT_DBG below
  // B_KMAJOR kernels: B is [taps][Kc][ldb] (k-major: the Flax kernel layout itself, read through transposing LDS reads) instead of
  // Bt [N][ldb]; b_nseg > 0: its N columns are b_nseg-wide segments, segment s at B + s*b_seg_stride with row pitch ldb
  // (Dense layers that share an input, whose kernels are separate leaves)
  int b_nseg;
  long b_seg_stride;
  // GEGLU inside the epilogue of the transformer feed-forward's first Dense layer (gemm_nt_kernel EPI = 1, sdt_ff_geglu_fwd): F = 4C gated
  // features; the N = 2F output columns are processed as 128-column tiles of 64 VALUE columns + the 64 GATE columns that gate them
  // (virtual column 128 b + w  <->  real column 64 b + (w & 63) + (w >= 64 ? F : 0)); the epilogue stores h (both halves, the backward
  // needs them) to C and value * gelu(gate) to C2: geglu_fwd's launch and its re-read of h are gone.  Same bf16 roundings: bit-identical.
  int geglu_f;
  bf16_t* C2;
  GatherDesc g;
};

// Ablation switches (they produce WRONG results: no waits / no math / no traffic / no epilogue) exist only in developer builds
// (SDT_HIPCC_EXTRA=-DSDT_NT_DBG, then the SDT_NT_DBG environment variable selects the bits); the shipped library compiles them out
// and reads no environment variable that can change a result.
#ifdef SDT_NT_DBG
#define NT_DBG(bit) (p.dbg & (bit))
#else
#define NT_DBG(bit) false
#endif

// 16 zero bytes every lane may DMA from (padding rows, conv halo, K tail)
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

// one lane's 16 bytes of a direct global -> LDS load; lds_wave_base must be wave-uniform (lane l lands at base + 16*l)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform n known only at run time (the immediate must be a constant)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define SDT_VMC(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    SDT_VMC(1) SDT_VMC(2) SDT_VMC(3) SDT_VMC(4) SDT_VMC(5) SDT_VMC(6) SDT_VMC(7) SDT_VMC(8) SDT_VMC(9) SDT_VMC(10) SDT_VMC(11)
    SDT_VMC(12) SDT_VMC(13) SDT_VMC(14) SDT_VMC(15) SDT_VMC(16) SDT_VMC(17) SDT_VMC(18) SDT_VMC(19) SDT_VMC(20) SDT_VMC(21)
    SDT_VMC(22) SDT_VMC(23) SDT_VMC(24) SDT_VMC(25) SDT_VMC(26) SDT_VMC(27) SDT_VMC(28) SDT_VMC(29) SDT_VMC(30) SDT_VMC(31)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef SDT_VMC
}

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * LDS_ROW_BYTES + (((chunk ^ ((row >> 1) ^ (row >> 4))) & 7) << 4);
}

// bijective XCD-aware remap: blocks that share an XCD (bid % 8) get a contiguous run of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---------------------------------------------------------------------------------------------------------------
// Reductions split across workgroups (split-K of the forward / input-gradient GEMMs, split-M of the weight gradients) use no
// atomics on the data: every split publishes its fp32 accumulators to a slab with write-through (sc1) 16-byte stores, drains
// them, and takes a ticket; the split that arrives LAST reads all slabs of the tile back (sc1 loads behind an agent-scope
// acquire), adds them IN SPLIT ORDER - so the sums are bitwise reproducible whichever split happens to be last - and goes on
// to the epilogue with the complete sums in its registers.  It also resets the ticket: counters are zero between launches
// and the slabs need no initialisation.  Placement-independent: nothing is assumed about which XCD / CU a split runs on, and
// no workgroup waits for another.  (fp32 atomics execute at the memory side at ~1.3 TB/s chip-wide and forced a zeroed
// accumulator: the weight gradients of the 1280-channel layers, tens of MB each, were bound by them.)
// Weight gradients are therefore WRITTEN, never accumulated: every element of dW (and of the fused bias gradient) has exactly
// one writer per launch, so the gradient buffer needs no zero fill.
#define TN_BIAS_SLOTS 128  // floats at the end of a workgroup slab for the fused bias-gradient partial sums
#define SPLIT_CNT_BYTES 65536  // arrival counters: the first 64 KiB of every split workspace (16384 tile groups), then the slabs
template <int NV, int NT = 256>
struct TnSlab {
  static constexpr int BIAS = TN_BIAS_SLOTS * (NT / 256);            // bias-gradient partial sums: one per tile column
  static constexpr int BYTES = NV * 16 * NT * 4 + BIAS * 4;          // NV accumulators of 16 registers x NT threads
};
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

// v: the workgroup's accumulators; bv: this lane's bias partial sums (lanes with bias_lane set own slots bias_slot0 + 32*j).
// Returns true in the workgroup that now holds the complete sums and has to write them out.
// S splits, this workgroup is split `me`; slab / tile_cnt: the launch's scratch, group: the (tile, tap) this workgroup adds to.
template <int NV, int NB, int NT = 256>
__device__ __forceinline__ bool split_reduce(unsigned char* slab, int* tile_cnt, int S, int me, f32x16_t (&v)[NV], float (&bv)[NB],
                                             bool bias_lane, int bias_slot0, int group, unsigned char* smem, int tid, int dbg = 0) {
  if (S == 1) return true;
  constexpr int BYTES = TnSlab<NV, NT>::BYTES;
  unsigned char* base = slab + (size_t)group * S * BYTES;  // wave-uniform: kernel argument + blockIdx arithmetic
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, S * BYTES, 0x00020000);
  const int mine = me * BYTES;
#ifdef SDT_NT_DBG
  if (!(dbg & 512))
#endif
#pragma unroll
  for (int r = 0; r < NV; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4_t f = {v[r][4 * q], v[r][4 * q + 1], v[r][4 * q + 2], v[r][4 * q + 3]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f), rs, mine + ((r * 4 + q) * NT + tid) * 16, 0, 16);
    }
  if (bias_lane) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(bv[j]), rs, mine + NV * 64 * NT + (bias_slot0 + 32 * j) * 4, 0, 16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its write-through stores have been performed
  int* s_last = reinterpret_cast<int*>(smem);       // the staging ring is dead by now (callers drained their LDS reads)
  __syncthreads();
  if (tid == 0) {
    // Publication: every byte of the slab left this CU as a write-through (sc1) 16-byte store, every storing wave has drained
    // its stores (s_waitcnt vmcnt(0) above) and the barrier puts all of that in front of this lane's ticket.  That is the recipe the
    // CDNA4 guide gives for exactly this case (cdna_hip_programming.md, "In-launch split-K reduction": "sc1 (write-through) slab stores
    // ..., which need no release fence -> every wave s_waitcnt vmcnt(0) -> __syncthreads() -> lane 0 relaxed agent fetch_add; the
    // reducer then reads the slabs with sc1 loads ... or with an acquire fence"; Guideline 16 R1; MI355X_MICROARCH.md "publish-large" /
    // "splitk-seam" and its hand-off table, row 1) - with BOTH of the consumer's options kept (acquire fence and sc1 loads, below), so
    // nothing rests on the one-workgroup-per-CU cell of that table.  An agent-scope RELEASE fence here (buffer_wbl2 sc1) is what the abstract memory
    // model would ask for on top; it writes back every dirty line of the XCD's L2 - the previous kernels' activations, nothing of
    // this slab, which is not dirty anywhere - and costs 1.1 ms per SD1.5 step (41.3 -> 40.3 ms same-box: (1024, 1280, 5120)
    // 43.7 -> 37.3 us per launch).  -DSDT_SPLIT_RELEASE_FENCE builds it in; the default is the guide's form, held by
    // test_split_reduction_handoff_under_uneven_load (every word, hundreds of launches beside a second stream's traffic).
#ifdef SDT_SPLIT_RELEASE_FENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    const int old = __hip_atomic_fetch_add(tile_cnt + group, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == S - 1;
    if (last) __hip_atomic_store(tile_cnt + group, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = last;
  }
  __syncthreads();
  if (!*s_last) return false;
#ifdef SDT_NT_DBG
  if (dbg & 256) return true;  // developer ablation: the last arriver goes on with its own sums (wrong results)
#endif
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave, ahead of its own loads
#pragma unroll
  for (int r = 0; r < NV; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) v[r][e] = 0.f;
#pragma unroll
  for (int j = 0; j < NB; ++j) bv[j] = 0.f;
  for (int s = 0; s < S; ++s) {  // fixed order (own slab included): the same sums whoever arrives last
    const int off = s * BYTES;
#ifndef SPLIT_G
#define SPLIT_G 8
#endif
    constexpr int G = NV * 4 < SPLIT_G ? NV * 4 : SPLIT_G;  // 16-byte loads in flight per thread (the accumulators fill most of the file)
#pragma unroll
    for (int i0 = 0; i0 < NV * 4; i0 += G) {
      u32x4_t w[G];
#pragma unroll
      for (int i = 0; i < G; ++i) w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + ((i0 + i) * NT + tid) * 16, 0, 16);
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const f32x4_t f = __builtin_bit_cast(f32x4_t, w[i]);
        const int r = (i0 + i) >> 2, q = (i0 + i) & 3;
        v[r][4 * q + 0] += f[0]; v[r][4 * q + 1] += f[1]; v[r][4 * q + 2] += f[2]; v[r][4 * q + 3] += f[3];
      }
    }
    if (bias_lane) {
#pragma unroll
      for (int j = 0; j < NB; ++j)
        bv[j] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off + NV * 64 * NT + (bias_slot0 + 32 * j) * 4, 0, 16));
    }
  }
  return true;
}


// GroupNorm statistics of the tile just written, for the GroupNorm that consumes this output (fused so that it needs no
// statistics pass of its own).  Every thread has summed its 8 columns over its rows (all of one image: the launcher only
// passes gn_stats when a tile never straddles images); lanes sharing a column chunk are CPR apart.  No atomics: the per-wave
// column sums go to LDS slots, one thread per group adds the group's columns of this tile (waves, then columns, in a fixed
// order), and the result is WRITTEN as this tile's partial row: gn_stats[b][2*rt + side][g], side 0 for a group that starts
// inside the tile's columns, side 1 for the one that started in the column tile to the left (a group is at most one tile
// wide: sdt_gemm_nt_gn_parts).  Every slot has exactly one writer, so the buffer needs no initialisation and the consumer's
// sum over the rows (sdt_groupnorm_fwd) is bitwise reproducible.
template <int CPR>
__device__ __forceinline__ void gn_tile_flush(const GemmNtParams& p, float (&s)[8], float (&q)[8], int n0, long b, int rt, float* scratch,
                                              int tid) {
  constexpr int W = CPR * 8;  // tile width in columns
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int off = CPR; off < 64; off <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[e] += __shfl_xor(s[e], off, 64);
      q[e] += __shfl_xor(q[e], off, 64);
    }
  float* ws = scratch;           // [4 waves][W] column sums
  float* wq = scratch + 4 * W;   // [4 waves][W] column sums of squares
  if (lane < CPR) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ws[wave * W + lane * 8 + e] = s[e];
      wq[wave * W + lane * 8 + e] = q[e];
    }
  }
  __syncthreads();
  const int G = p.gn_groups, cpg = p.N / G;
  const int g = n0 / cpg + tid;  // thread t: the t-th group that intersects this tile's columns
  if (tid < 64 && g < G) {
    const int c_lo = g * cpg, c_hi = c_lo + cpg;
    const int lo = max(c_lo, n0), hi = min(c_hi, min(n0 + W, p.N));
    if (lo < hi) {
      float a = 0.f, c = 0.f;
      for (int col = lo - n0; col < hi - n0; ++col)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a += ws[w * W + col];
          c += wq[w * W + col];
        }
      float* o = p.gn_stats + ((((long)b * p.gn_parts + 2 * rt) * G) + g) * 2;
      if (c_lo >= n0) {
        o[0] = a; o[1] = c;
        if (c_hi <= n0 + W) { o[2 * G] = 0.f; o[2 * G + 1] = 0.f; }  // complete in this tile: nobody else writes its side-1 slot
      } else {
        o[2 * G] = a; o[2 * G + 1] = c;
      }
    }
  }
}
__device__ __forceinline__ void gn_accum(float (&s)[8], float (&q)[8], uint4 v) {
  float f[8];
  unpack8(v, f);
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
}

// source element offset (or -1) of GEMM row (b, oy, ox) for tap (kh, kw)
__device__ __forceinline__ long gather_src(const GatherDesc& g, int b, int oy, int ox, int kh, int kw, int ld) {
  int sy, sx;
  if (g.mode == GATHER_FPROP) {
    sy = oy * g.stride + kh - g.pad_t;
    sx = ox * g.stride + kw - g.pad_l;
    if (sy < 0 || sx < 0 || sy >= g.IH || sx >= g.IW) return -1;
  } else {  // DGRAD: source = dY of the forward conv, rows enumerate the forward conv's input grid
    const int ty = oy + g.pad_t - kh, tx = ox + g.pad_l - kw;
    if (ty < 0 || tx < 0) return -1;
    if (g.stride == 1) {
      sy = ty; sx = tx;
    } else {
      if ((ty % g.stride) | (tx % g.stride)) return -1;
      sy = ty / g.stride; sx = tx / g.stride;
    }
    if (sy >= g.IH || sx >= g.IW) return -1;
  }
  return ((long)(b * g.IH + sy) * g.IW + sx) * ld;
}


typedef __attribute__((ext_vector_type(4))) short s16x4_t;

#ifndef TN_KB2
#define TN_KB2 64   // rows of a 128-tile K-step
#endif
#ifndef TN_NST2
#define TN_NST2 2   // ... and ring stages
#endif
#ifndef TN_WPS
#define TN_WPS 2    // waves per SIMD the weight-gradient kernels are built for (= workgroups per CU)
#endif
template <int TM>
struct TnCfg {
  static constexpr int EDGE = 64 * TM;
  static constexpr int KB = (TM == 2) ? TN_KB2 : 64;  // rows (reduction indices) per staged tile
  static constexpr int RB = EDGE * 2;               // bytes per staged row
  static constexpr int CPR = EDGE / 8;              // 16-byte chunks per row
  static constexpr int RPI = 1024 / RB;             // rows written by one wave-instruction of LDS-DMA
  static constexpr int IPW = KB / (RPI * 4);        // DMA instructions per wave per operand per tile
  static constexpr int TILE_BYTES = KB * RB;
  static constexpr int NST = (TM == 2) ? TN_NST2 : 4;     // ring stages (A + B each)
  static constexpr int LDS_BYTES = NST * 2 * TILE_BYTES;
};

// chunk swizzle making the tr reads (4 rows x 64 B per 32-lane half) conflict-free
template <int TM>
__device__ __forceinline__ int tn_swz(int row) {
  if (TM == 2) return ((row & 3) << 2) | ((row >> 2) & 3);
  return ((row >> 1) & 1) << 2;
}

// asm-owned transposing reads.  hipcc treats an in-flight LDS-DMA as a pending LDS write and drains it (s_waitcnt vmcnt(0))
// in front of the first LDS read it can see, which would serialise the DMA ring on every K-step; reads it cannot see are
// ordered by hand instead: counted s_waitcnt lgkmcnt on the fragment registers (TR_WAIT*), data readiness by the ring's own
// vmcnt + barrier.  A fragment is only assembled from its two halves AFTER its wait (any copy the compiler adds is then safe).
struct TrFrag {
  s16x4_t lo, hi;
};
__device__ __forceinline__ unsigned lds_offset_of(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}
template <int TM>
__device__ __forceinline__ void tn_frag_issue(TrFrag& f, unsigned img, int col_base, int s, int lane, int row_off = 0) {
  constexpr int RB = TnCfg<TM>::RB;
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int col = col_base + 16 * (g & 1) + 4 * pp;
  const int chunk = col >> 3, within = (pp & 1) * 8;
  const int r1 = 16 * s + 8 * (g >> 1) + q + row_off, r2 = r1 + 4;
  const unsigned a1 = img + r1 * RB + ((chunk ^ tn_swz<TM>(r1)) << 4) + within;
  const unsigned a2 = img + r2 * RB + ((chunk ^ tn_swz<TM>(r2)) << 4) + within;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a1) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a2) : "memory");
}
// the same for v_mfma_f32_16x16x32_bf16: lane l holds B[k = 8 (l >> 4) + j][col_base + (l & 15)], j = 0..7, of the k32-step s
template <int TM>
__device__ __forceinline__ void tn_frag_issue16(TrFrag& f, unsigned img, int col_base, int s, int lane) {
  constexpr int RB = TnCfg<TM>::RB;
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int col = col_base + 4 * pp;
  const int chunk = col >> 3, within = (pp & 1) * 8;
  const int r1 = 32 * s + 8 * g + q, r2 = r1 + 4;
  const unsigned a1 = img + r1 * RB + ((chunk ^ tn_swz<TM>(r1)) << 4) + within;
  const unsigned a2 = img + r2 * RB + ((chunk ^ tn_swz<TM>(r2)) << 4) + within;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a1) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a2) : "memory");
}
__device__ __forceinline__ bf16x8_t tr_value(const TrFrag& t) {
  bf16x8_t f;
  f[0] = t.lo[0]; f[1] = t.lo[1]; f[2] = t.lo[2]; f[3] = t.lo[3]; f[4] = t.hi[0]; f[5] = t.hi[1]; f[6] = t.hi[2]; f[7] = t.hi[3];
  return f;
}
// [r4] The same reads as a lane-dependent base + an immediate.  tn_swz uses only bits of the row that a multiple of 16 leaves alone,
// so the fragment of K16-step s sits 16 * s rows (16 * s * RB bytes) behind the fragment of step 0: one address register pair per
// fragment and stage (base + stage offset) instead of an address computed per read (the weight-gradient loops issued 2 - 5 vector
// ALU instructions per transposing read: 54 v_add per 16 MFMAs in the Dense kernel, ~250 per 28 in the 3x3 one - beside two waves'
// MFMAs the vector issue slots were as scarce as the matrix pipe).
struct TrBase {
  unsigned lo, hi;
};
template <int TM>
__device__ __forceinline__ TrBase tn_frag_base(int col_base, int lane, int row_off = 0) {
  constexpr int RB = TnCfg<TM>::RB;
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int col = col_base + 16 * (g & 1) + 4 * pp;
  const int chunk = col >> 3, within = (pp & 1) * 8;
  const int r1 = 8 * (g >> 1) + q + row_off, r2 = r1 + 4;
  TrBase b;
  b.lo = r1 * RB + ((chunk ^ tn_swz<TM>(r1)) << 4) + within;
  b.hi = r2 * RB + ((chunk ^ tn_swz<TM>(r2)) << 4) + within;
  return b;
}
// an asm-owned 16-byte fragment read (waited for by a counted s_waitcnt that names the register)
__device__ __forceinline__ void lds_read128(bf16x8_t& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
}
template <int OFF>
__device__ __forceinline__ void tr_read_at(TrFrag& f, unsigned lo, unsigned hi) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(lo), "n"(OFF) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(hi), "n"(OFF) : "memory");
}
template <int V>
struct IntC {
  static constexpr int value = V;
};
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {  // f(IntC<I>{}) ... f(IntC<N - 1>{}): loop indices usable as immediates
  if constexpr (I < N) {
    f(IntC<I>{});
    static_for<I + 1, N>(f);
  }
}
#define TR_OPS1(f) "+v"((f).lo), "+v"((f).hi)
#define TR_WAIT2(N, a, b) asm volatile("s_waitcnt lgkmcnt(" #N ")" : TR_OPS1(a), TR_OPS1(b)::"memory")
#define TR_WAIT4(N, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(" #N ")" : TR_OPS1(a), TR_OPS1(b), TR_OPS1(c), TR_OPS1(d)::"memory")
#define TR_WAIT7(N, a, b, c, d, e, f, g) \
  asm volatile("s_waitcnt lgkmcnt(" #N ")" : TR_OPS1(a), TR_OPS1(b), TR_OPS1(c), TR_OPS1(d), TR_OPS1(e), TR_OPS1(f), TR_OPS1(g)::"memory")


template <int TM>  // TM x TM MFMA tiles per wave: tile edge = 64 * TM
struct TileCfg {
  static constexpr int EDGE = 64 * TM;
  static constexpr int NL = EDGE / 32;                    // 16-byte loads per thread per operand (NT kernel)
  static constexpr int TILE_BYTES = EDGE * LDS_ROW_BYTES;  // one operand tile
  static constexpr int LDS_BYTES = 4 * TILE_BYTES;         // TN kernel: 2 buffers x (A + B)
#ifndef NT_NST1
#define NT_NST1 3  // 64-tile ring depth: 3 stages = 48 KB, three workgroups per CU (4 stages / two per CU: +0.18 ms per step, 2 stages / five: +0.45)
#endif
  static constexpr int NST = (TM == 2) ? 2 : NT_NST1;      // NT kernel: LDS-DMA ring depth; 128-tile: 2 stages so that two blocks share a CU (measured faster than 3 stages alone)
  static constexpr int GN_SCRATCH = (TM == 2) ? 49152 : 16384;  // epilogue: GroupNorm partial-sum scratch behind the C tile
  static constexpr int LDS_BYTES_NT = NST * 2 * TILE_BYTES;
  static constexpr int R_NT = (TM == 1) ? 3 : 2;           // K-tiles of global loads kept in flight (register ring)
  static constexpr int R_TN = (TM == 1) ? 4 : 2;
};

// The NT kernel's tile: EDGE x EDGE outputs, K-steps of KB = 64 or 32 elements.  KB = 32 (128-tiles only): half-size stages, three
// of them (48 KB), so THREE workgroups share a CU instead of two: for short reductions (K <= 1280: 5 - 20 steps of 64) a tile is
// mostly first-fetch latency + epilogue, and what hides those is another workgroup's main loop, not a deeper ring of one's own
// (measured on (16384, 2560, 320): staging alone 33 us, epilogue alone 31 us, MFMAs ~10 us, the whole launch 69 us at two per CU).
template <int TM, int KB>
struct NtCfg {
  static constexpr int EDGE = 64 * TM;
  static constexpr int ROWB = KB * 2;                      // bytes of one staged row
  static constexpr int CH = KB / 8;                        // 16-byte chunks per row
  static constexpr int RPP = 256 / CH;                     // rows one pass of the 256 threads stages
  static constexpr int NL = EDGE / RPP;                    // 16-byte loads per thread per operand
  static constexpr int TILE_BYTES = EDGE * ROWB;
#ifndef NT_K32_NST
#define NT_K32_NST 3
#endif
  static constexpr int NST = (KB == 32) ? NT_K32_NST : TileCfg<TM>::NST;
  static constexpr int LDS_BYTES = NST * 2 * TILE_BYTES;
  static constexpr int GN_SCRATCH = (TM == 2) ? 36864 : 16384;  // epilogue: GroupNorm partial-sum scratch behind the C tile
  static_assert(LDS_BYTES >= GN_SCRATCH + 8 * EDGE * 4, "epilogue scratch must fit the ring");
};
// LDS byte offset of chunk `chunk` of staged row `row` (row-major operand tiles): the XOR keeps every ds_read_b128 lane group on
// 16 different 16-byte bank groups (128-byte rows: any 8 rows x 2 k-halves; 64-byte rows: rows r, r+12, r+20, r+24 of a group share
// a base slot and differ in (row >> 2) & 3)
template <int KB>
__device__ __forceinline__ int nt_lds_off(int row, int chunk) {
  if (KB == 64) return row * 128 + (((chunk ^ ((row >> 1) ^ (row >> 4))) & 7) << 4);
  return row * 64 + (((chunk ^ (row >> 2)) & 3) << 4);
}

// GENERIC = false: every tap's source offset is  base(row) + tapoff(tap)  with a per-row validity bit mask, all hoisted
// out of the K loop (plain rows, conv fprop at any stride, conv dgrad at stride 1).  GENERIC = true keeps the
// per-load decomposition (conv dgrad at stride > 1: the three UNet downsamplers).  Offsets are 32-bit elements.
// PACK8 (3x3 forward convolutions of an 8-channel input - the VAE's and the UNet's conv_in - with k-major weights): a pixel's 8
// channels are ONE 16-byte chunk, so a 64-wide K-step holds eight TAPS instead of one tap's 8 channels and 56 zeros: the
// launcher passes taps = 1, Kc = 72 (the HWIO kernel [9][8][N] is a plain [72][N] k-major matrix) and the lane that stages chunk c
// of K-step s gathers tap 8 s + c.  2 K-steps instead of 9 (-0.08 ms per SD1.5 step same-box: the VAE's conv_in is bound by its 268 MB of output, not by the MFMAs).
template <int TM, bool SPLITK, bool GENERIC, bool BKM, bool PACK8 = false, int KB = 64, int EPI = 0>
__global__ void __launch_bounds__(256, 2) gemm_nt_kernel(const GemmNtParams p) {
  static_assert(EPI == 0 || (EPI == 1 && TM == 2 && !SPLITK && !GENERIC && !PACK8 && BKM), "GEGLU epilogue: unsplit 128-tiles, k-major weights");
  using Cfg = NtCfg<TM, KB>;
  constexpr int EDGE = Cfg::EDGE, NL = Cfg::NL, TILE_BYTES = Cfg::TILE_BYTES, CH = Cfg::CH, RPP = Cfg::RPP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  const int m0 = (p.nfast ? tile / p.tiles_n : tile % p.tiles_m) * EDGE, n0 = (p.nfast ? tile % p.tiles_n : tile / p.tiles_m) * EDGE;

  // ---- per-thread load plan: chunk c (16 B of the KB-wide K slab), rows r + RPP*i
  const int c = tid % CH, r = tid / CH;
  int a_b[NL], a_y[NL], a_x[NL];   // GENERIC: row decomposition
  int a_base[NL];                  // fast path: element offset of tap (0,0) (only dereferenced when the mask bit is set)
  unsigned a_mask[NL];             // fast path: bits [0,8) = kh valid, bits [8,16) = kw valid
  int b_row[NL];                   // n*ldb (or -1); k-major B: element offset of the lane's chunk inside a K-step (or -1)
  int csw_b[NL];                   // k-major B: the lane's row inside the K-step
  const bool dgrad = p.g.mode == GATHER_DGRAD;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int m = m0 + r + RPP * i;
    a_b[i] = -1; a_y[i] = 0; a_x[i] = 0; a_base[i] = 0; a_mask[i] = 0;
    if (m < p.M) {
      if (p.g.mode == GATHER_PLAIN) {
        a_base[i] = m * p.lda;
        a_mask[i] = 0x80000000u;  // row valid; plain rows have no per-tap validity (tap t = column block t of A)
      } else {
        const unsigned b = fd_div((unsigned)m, p.g.div_ohw);
        const unsigned rem = (unsigned)m - b * p.g.div_ohw.d;
        const unsigned oy = fd_div(rem, p.g.div_ow);
        const int ox = (int)(rem - oy * p.g.div_ow.d);
        a_b[i] = (int)b; a_y[i] = (int)oy; a_x[i] = ox;
        if (!GENERIC) {
          const int y0 = dgrad ? (int)oy + p.g.pad_t : (int)oy * p.g.stride - p.g.pad_t;
          const int x0 = dgrad ? ox + p.g.pad_l : ox * p.g.stride - p.g.pad_l;
          a_base[i] = (((int)b * p.g.IH + y0) * p.g.IW + x0) * p.lda;
          unsigned mk = 0;
          for (int k = 0; k < p.g.KH; ++k) {
            const int sy = dgrad ? y0 - k : y0 + k;
            if (sy >= 0 && sy < p.g.IH) mk |= 1u << k;
          }
          for (int k = 0; k < p.g.KW; ++k) {
            const int sx = dgrad ? x0 - k : x0 + k;
            if (sx >= 0 && sx < p.g.IW) mk |= 0x100u << k;
          }
          a_mask[i] = mk;
        }
      }
    }
    const int n = n0 + r + RPP * i;
    b_row[i] = (n < p.N) ? n * p.ldb : -1;
    if (BKM) {  // k-major B: piece (i*4 + wave) of the [64 k][EDGE n] tile = RPI rows; this lane's row and 8-column chunk
      constexpr int CPRB = EDGE / 8, RPI = 1024 / (EDGE * 2);
      const int rloc = (i * 4 + wave) * RPI + lane / CPRB;
      const int ncol = n0 + (((lane % CPRB) ^ tn_swz<TM>(rloc)) << 3);
      int off = ncol;
      if (EPI == 1) {  // virtual -> real column: 64 value columns, then their 64 gate columns
        const int w = ncol & 127;
        off = (n0 >> 1) + (w & 63) + (w >= 64 ? p.geglu_f : 0);
      } else if (p.b_nseg > 0) {
        const int seg = ncol / p.b_nseg;
        off = (int)(seg * p.b_seg_stride) + (ncol - seg * p.b_nseg);
      }
      b_row[i] = (ncol < p.N) ? rloc * p.ldb + off : -1;
      csw_b[i] = rloc;  // the reduction row this lane stages (validity against the K tail)
    }
  }

  const int ksteps_per_tap = (p.Kc + KB - 1) / KB;
  const int Ttot = p.taps * ksteps_per_tap;
  int t_beg = 0, t_end = Ttot;
  if (SPLITK) {
    t_beg = blockIdx.y * p.ksteps_per_split;
    t_end = min(t_beg + p.ksteps_per_split, Ttot);  // (the launcher creates no empty split: every split takes its ticket)
  }

  // Staging is direct global -> LDS DMA (global_load_lds_dwordx4): one wave-instruction writes 64 lanes x 16 B = 8 rows
  // x 128 B contiguously at a wave-uniform LDS base, so the bank swizzle lives on the SOURCE side: the lane that owns
  // LDS slot c of row q fetches global chunk c ^ f(q).  Out-of-range rows / taps / K-tail fetch 16 zero bytes.
  int csw[NL];  // swizzled K offset (elements) of this lane's chunk, per row pass
  const bf16_t* pa[NL];  // per-row source pointers with the lane's chunk folded in (fast path)
  const bf16_t* pb[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    csw[i] = (nt_lds_off<KB>(r + RPP * i, c) - (r + RPP * i) * Cfg::ROWB) >> 1;  // the global chunk (in elements) that lands in this lane's slot
    pa[i] = p.A + (a_base[i] + csw[i]);
    pb[i] = p.Bt + ((b_row[i] >= 0 ? b_row[i] : 0) + (BKM ? 0 : csw[i]));
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(g_zero16);
  const bool ktail = (p.Kc & (KB - 1)) != 0;  // only then can a chunk fall past the end of the reduction
  // scalar K-step cursor (tap, kh, kw, k offset inside the tap), advanced once per staged tile: no divisions in the loop
  int s_tap = t_beg / ksteps_per_tap;
  int s_kc = (t_beg - s_tap * ksteps_per_tap) * KB;
  int s_kh = s_tap / p.g.KW, s_kw = s_tap - s_kh * p.g.KW;
  auto stage = [&](int buf) {
    const int kc0 = s_kc, kh = s_kh, kw = s_kw;
    const bool plain = p.g.mode == GATHER_PLAIN;
    const long soff_a = plain ? (long)s_tap * p.Kc + kc0 : (long)(dgrad ? -(kh * p.g.IW + kw) : (kh * p.g.IW + kw)) * p.lda + kc0;
    const long soff_b = (long)s_tap * p.b_tap_stride + (BKM ? (long)kc0 * p.ldb : (long)kc0);
    const unsigned tapbit = plain ? 0x80000000u : ((1u << kh) | (0x100u << kw));
    unsigned char* sa = smem + buf * 2 * TILE_BYTES + wave_u * 1024;  // buf = ring stage; a wave-instruction fills 1 KiB of whole rows
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const bool kvalid = !ktail || (kc0 + csw[i] < p.Kc);
      const bf16_t* srca;
      if (GENERIC) {
        long off = -1;
        if (a_b[i] >= 0) off = gather_src(p.g, a_b[i], a_y[i], a_x[i], kh, kw, p.lda);
        srca = (kvalid && off >= 0) ? p.A + (off + kc0 + csw[i]) : zero_src;
      } else if (PACK8) {
        const int tap = (kc0 + csw[i]) >> 3;            // 0 .. 15, taps 9 .. 15 are the K tail
        const int tkh = (tap * 11) >> 5, tkw = tap - 3 * tkh;  // tap / 3, tap % 3 for tap < 16
        const unsigned tb = (1u << tkh) | (0x100u << tkw);
        srca = (kvalid && (a_mask[i] & tb) == tb) ? p.A + (a_base[i] + (tkh * p.g.IW + tkw) * p.lda) : zero_src;
      } else {
        srca = (kvalid && (a_mask[i] & tapbit) == tapbit) ? pa[i] + soff_a : zero_src;
      }
      const bool kvalid_b = BKM ? (!ktail || kc0 + csw_b[i] < p.Kc) : kvalid;
      const bf16_t* srcb = (kvalid_b && b_row[i] >= 0) ? pb[i] + soff_b : zero_src;
      glds16(srca, sa + i * 4096);
      glds16(srcb, sa + TILE_BYTES + i * 4096);
    }
    s_kc += KB;
    if (s_kc >= p.Kc) {
      s_kc = 0; ++s_tap; ++s_kw;
      if (s_kw == p.g.KW) { s_kw = 0; ++s_kh; }
    }
  };

  f32x16_t acc[TM][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  constexpr int WE = 32 * TM;  // wave tile edge

  // fragment offsets inside a stage: A (row-major tile) per K16-step; B per K16-step (row-major) or one base pair (k-major tile,
  // transposing reads: K16-step q sits q * 16 rows further, tr_read_at)
  constexpr int NS = KB / 16;
  const unsigned lds0 = lds_offset_of(smem);
  unsigned ra[TM][NS], rb[TM][BKM ? 2 : NS];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      ra[i][q] = nt_lds_off<KB>(wm * WE + i * 32 + fr, 2 * q + fh);
      if (!BKM) rb[i][q] = nt_lds_off<KB>(wn * WE + i * 32 + fr, 2 * q + fh);
    }
    if (BKM) {
      const TrBase tbase = tn_frag_base<TM>(wn * WE + i * 32, lane);
      rb[i][0] = tbase.lo; rb[i][1] = tbase.hi;
    }
  }

  // NST-stage LDS ring: the DMA of tiles t+1 .. t+NST-2 stays in flight ACROSS the barrier (counted vmcnt, raw
  // s_barrier - a __syncthreads() would drain it), so only throughput, not the issue->landed latency, is exposed.
  // [r4] The depth is a launch parameter (p.nst >= Cfg::NST): a grid that leaves CUs with one workgroup (the text tower, the
  // time-embedding and 8 x 8 / 16 x 16-level projections: 60 - 250 tiles) gets the LDS the absent neighbours would have used as a
  // deeper ring, so most of its short reduction is in flight from the prologue on instead of two tiles at a time.
  const int NST = p.nst;
  constexpr int LPT = 2 * NL;  // LDS-DMA instructions each wave issues per tile
  for (int s = 0; s < NST - 1; ++s)
    if (t_beg + s < t_end) stage(s);
  int rd = 0, wr = NST - 1;  // ring cursors: the stage tile t is read from, the stage the next DMA fills
  for (int t = t_beg; t < t_end; ++t) {
    const int idx = rd;
    const int ahead = min(NST - 2, t_end - 1 - t);  // younger tiles already issued
    if (!NT_DBG(1)) wait_vmcnt(ahead * LPT);
    __builtin_amdgcn_s_barrier();  // tile t landed for every wave; everyone is done reading the stage the next DMA fills
    if (t + NST - 1 < t_end && !NT_DBG(4)) {
      stage(wr);
      if (++wr == NST) wr = 0;
    }
    if (++rd == NST) rd = 0;
    if (NT_DBG(2)) continue;
    // [r4] every fragment of the K-step goes out at once (asm-owned reads at precomputed addresses), then the MFMAs of K16-step s
    // wait for exactly their own reads (counted lgkmcnt: LDS reads return in order).  Round 3 read, waited lgkmcnt(0) and
    // multiplied once per K16-step - the LDS latency four times per tile beside 32-cycle MFMAs - and rebuilt every address with
    // vector ALU work in the loop (64-tiles: ~250 VALU instructions around 4 MFMAs).  The MFMA chain over k is unchanged.
    const unsigned st_off = lds0 + idx * 2 * TILE_BYTES;
    unsigned ca[TM][NS], cb[TM][BKM ? 2 : NS];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int q = 0; q < NS; ++q) ca[i][q] = st_off + ra[i][q];
#pragma unroll
      for (int q = 0; q < (BKM ? 2 : NS); ++q) cb[i][q] = st_off + TILE_BYTES + rb[i][q];
    }
    bf16x8_t af[NS][TM], bq[NS][TM];
    TrFrag tb[NS][TM];
    constexpr int RPS = (BKM ? 3 : 2) * TM;                      // reads per K16-step
    constexpr int DEPTH = NS < 15 / RPS + 1 ? NS : 15 / RPS + 1;  // K16-steps of reads in flight (lgkmcnt counts to 15)
    auto issue = [&](auto S) {
      constexpr int q = decltype(S)::value;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        lds_read128(af[q][i], ca[i][q]);
        if (BKM) tr_read_at<q * 16 * TnCfg<TM>::RB>(tb[q][i], cb[i][0], cb[i][1]);
        else lds_read128(bq[q][i], cb[i][BKM ? 0 : q]);
      }
    };
    static_for<0, DEPTH>(issue);
    static_for<0, NS>([&](auto S) {
      constexpr int q = decltype(S)::value;
      constexpr int YOUNGER = ((NS - 1 - q) < (DEPTH - 1) ? (NS - 1 - q) : (DEPTH - 1)) * RPS;  // reads issued behind this K16-step's
      if (BKM) {
        if constexpr (TM == 1) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(af[q][0]), TR_OPS1(tb[q][0]) : "n"(YOUNGER) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(af[q][0]), "+v"(af[q][TM - 1]), TR_OPS1(tb[q][0]), TR_OPS1(tb[q][TM - 1]) : "n"(YOUNGER) : "memory");
      } else {
        if constexpr (TM == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(af[q][0]), "+v"(bq[q][0]) : "n"(YOUNGER) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(af[q][0]), "+v"(af[q][TM - 1]), "+v"(bq[q][0]), "+v"(bq[q][TM - 1]) : "n"(YOUNGER) : "memory");
      }
      bf16x8_t bfr[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) bfr[i] = BKM ? tr_value(tb[q][i]) : bq[q][i];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          // swapped: D[row = n_local][col = m_local]: each lane owns 4 consecutive n of one output row
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[q][i], acc[i][j], 0, 0, 0);
        }
      if constexpr (q + DEPTH < NS) issue(IntC<q + DEPTH>{});
    });
  }
  __syncthreads();  // all waves done with the ring before the epilogue reuses it
  if (NT_DBG(64)) return;  // developer ablation (-DSDT_NT_DBG builds only): no epilogue

  if (SPLITK) {  // only the split that arrives last at this tile goes on, with the complete sums (split_reduce)
    float nob[1] = {0.f};
    if (!split_reduce<TM * TM, 1>(p.slab, p.tile_cnt, (int)gridDim.y, (int)blockIdx.y, reinterpret_cast<f32x16_t(&)[TM * TM]>(acc), nob,
                                  false, 0, tile, smem, tid, p.dbg))
      return;
    __syncthreads();  // (the ticket word in LDS is about to be overwritten by the C tile)
  }

  // ---- epilogue: (+bias) -> bf16 -> LDS C tile [EDGE][EDGE+8] -> coalesced 16-byte stores (+rowbias, +residual)
  constexpr int CP = EDGE + 8;  // pitch in elements (16-byte aligned rows)
  bf16_t* sc = reinterpret_cast<bf16_t*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int ml = wm * WE + i * 32 + fr;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int nl = wn * WE + j * 32 + 8 * g4 + 4 * fh;
        float v0 = acc[i][j][4 * g4 + 0], v1 = acc[i][j][4 * g4 + 1], v2 = acc[i][j][4 * g4 + 2], v3 = acc[i][j][4 * g4 + 3];
        if (p.bias && n0 + nl < p.N) {
          const int nb = EPI == 1 ? (n0 >> 1) + (nl & 63) + (nl >= 64 ? p.geglu_f : 0) : n0 + nl;
          const float4 bv = *reinterpret_cast<const float4*>(p.bias + nb);
          v0 += bv.x; v1 += bv.y; v2 += bv.z; v3 += bv.w;
        }
        uint2 pk;
        pk.x = pack2bf(v0, v1);
        pk.y = pack2bf(v2, v3);
        *reinterpret_cast<uint2*>(sc + ml * CP + nl) = pk;
      }
    }
  __syncthreads();
  if (EPI == 1) {  // FF1 + GEGLU: 128 rows x 8 chunk pairs (value chunk c, gate chunk c + 8)
    const int F = p.geglu_f, nv = (n0 >> 1);
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int idx = tid + 256 * ps, ml = idx >> 3, cc = idx & 7;
      const int m = m0 + ml, n = nv + cc * 8;
      if (m < p.M && n < F) {
        const uint4 va = *reinterpret_cast<const uint4*>(sc + ml * CP + cc * 8);
        const uint4 vg = *reinterpret_cast<const uint4*>(sc + ml * CP + 64 + cc * 8);
        float a[8], g[8];
        unpack8(va, a);
        unpack8(vg, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = a[e] * gelu_tanh_f(g[e]);
        *reinterpret_cast<uint4*>(p.C + (long)m * p.ldc + n) = va;
        *reinterpret_cast<uint4*>(p.C + (long)m * p.ldc + F + n) = vg;
        *reinterpret_cast<uint4*>(p.C2 + (long)m * F + n) = pack8(a);
      }
    }
    return;
  }
  {
    constexpr int CPR = EDGE / 8;      // 16-byte chunks per row
    constexpr int RPP = 256 / CPR;     // rows per pass
    const int cc = tid % CPR, rr = tid / CPR;
    const int n = n0 + cc * 8;
    float gns[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gnq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ps = 0; ps < EDGE / RPP; ++ps) {
      const int ml = rr + RPP * ps;
      const int m = m0 + ml;
      if (m < p.M && n < p.N) {
        uint4 v = *reinterpret_cast<const uint4*>(sc + ml * CP + cc * 8);
        if (p.rowbias || p.residual) {
          float f[8], g[8];
          unpack8(v, f);
          if (p.rowbias) {
            unpack8(*reinterpret_cast<const uint4*>(p.rowbias + (long)(m / p.rows_per_batch) * p.ldrb + n), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += g[e];
          }
          if (p.residual) {
            unpack8(*reinterpret_cast<const uint4*>(p.residual + (long)m * p.ldres + n), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += g[e];
          }
          v = pack8(f);
        }
        if (!NT_DBG(128)) *reinterpret_cast<uint4*>(p.C + (long)m * p.ldc + n) = v;
        if (p.gn_stats) gn_accum(gns, gnq, v);
      }
    }
    if (p.gn_stats) {
      __syncthreads();  // every wave has read its rows of the C tile: the scratch may overlap it
      gn_tile_flush<CPR>(p, gns, gnq, n0, m0 / p.rows_per_batch, (m0 % p.rows_per_batch) / EDGE, reinterpret_cast<float*>(smem + Cfg::GN_SCRATCH), tid);
    }
  }
}

// =====================================================================================================
// 3x3, stride 1, pad 1 convolution (fprop, and dgrad = the same sweep with mirrored taps over the transposed weights) as
// an implicit GEMM that stages the INPUT ONCE per 64-channel chunk: a tile is 256 output pixels (NI images x TH rows x TW
// columns) x 128 output channels, and its (TH+2) x (TW+2) input halo lands in LDS by LDS-DMA once and serves all nine taps
// (the generic kernel re-stages a 128 x 64 activation tile per tap: 9x the activation traffic, and LDS-DMA issue is what
// bounds that kernel).  Per tap only the 128 x 64 weight tile streams through a 3-stage ring, issued two taps ahead.  Per 64-channel chunk a
// workgroup moves ~51 KB of halo + 9 x 16 KB of weights for 9 x 32 MFMAs per wave: ~21 B per MFMA-cycle per CU, under
// the ~33 B/clk/CU the LDS-DMA path sustains, so the loop is MFMA-paced.
//   LDS: halo image [<=416 px][64 ch] x 2 buffers (next chunk lands during this chunk's taps, 2 pieces per tap per wave)
//        + weight ring 3 x [128][64]; 16-byte chunks XOR-swizzled by (row >> 1) & 7 on the DMA source side: any 16
//        consecutive rows at one chunk index are bank-conflict free, whatever the tap shift.
//   4 waves as 2 (pixels) x 2 (channels): 128 x 64 per wave = 4 x 2 MFMA tiles (128 accumulator registers), 1 WG / CU.
//   BN = 64 (conv_halo_bn): the tile is 256 pixels x 64 channels, 4 waves of 64 x 64, ONE halo buffer + 3 x 8 KB of weights =
//   76 KB of LDS and <= 256 registers per wave, so TWO workgroups share a CU: while one waits (chunk seam: its next halo is
//   fetched after the last tap, nothing of its own overlaps that; tap barriers; the store-issue-bound epilogue) the other
//   keeps the matrix pipe busy.  Halo traffic doubles (each 64-channel tile fetches it), weights per MFMA stay: ~27 B per
//   MFMA-cycle per CU.
#define CV_BM 256
#define CV_HALO_PIECES 13                                  // 1-KiB LDS-DMA pieces per wave per halo (4 x 13 x 8 = 416 pixels)
#define CV_HALO_BYTES (4 * CV_HALO_PIECES * 1024)
#define CV_NSTB 3
template <int BN>
struct CvCfg {
  static constexpr int HB = BN == 128 ? 2 : 1;             // halo buffers
  static constexpr int WM = BN == 128 ? 2 : 4;             // waves along the pixel dimension (x 4 / WM along channels)
  static constexpr int NI = CV_BM / WM / 32;               // 32-pixel MFMA tiles per wave (x 2 channel tiles: 64 channels per wave)
  static constexpr int B_BYTES = BN * LDS_ROW_BYTES;       // one weight tile [BN][64] / [64][BN]
  static constexpr int BPW = B_BYTES / 4096;               // weight DMA pieces per wave per tap
  static constexpr int LDS_BYTES = HB * CV_HALO_BYTES + CV_NSTB * B_BYTES;
  static constexpr int WG_PER_CU = BN == 128 ? 1 : 2;
};

__device__ __forceinline__ int cv_off(int row, int chunk) { return row * LDS_ROW_BYTES + (((chunk ^ (row >> 1)) & 7) << 4); }

// MF16 (BN = 64 only): the MFMAs are v_mfma_f32_16x16x32_bf16 - a wave's 64 x 64 tile as 4 x 4 tiles, two k32-steps per tap - instead
// of v_mfma_f32_32x32x16_bf16 (2 x 2 tiles, four k16-steps): same fragments bytes, same accumulator registers; with two workgroups per
// CU the loop's read / MFMA mix runs 1.17x faster in that shape (tools/mfma_shape_probe.hip: 1667 vs 1434 TFLOP/s).
template <bool SPLITK, bool BKM, int BN, bool MF16>
__global__ void __launch_bounds__(256, CvCfg<BN>::WG_PER_CU) conv3x3_halo_kernel(const GemmNtParams p) {
  static_assert(!MF16 || BN == 64, "the 16x16x32 form is built for the 64-channel tile");
  using Cfg = CvCfg<BN>;
  constexpr int HB = Cfg::HB, BPW = Cfg::BPW;
  constexpr int RT = MF16 ? 16 : 32;                    // rows / columns of one MFMA tile
  constexpr int NI = (CV_BM / Cfg::WM) / RT;            // pixel tiles per wave
  constexpr int NJ = 64 / RT;                           // channel tiles per wave (64 channels)
  constexpr int KS = MF16 ? 2 : 4;                      // MFMA k-steps per 64-channel tap
  constexpr int KG = MF16 ? 4 : 2;                      // 16-byte k-chunks per k-step
  constexpr int TMB = BN / 64;  // row width of a k-major weight tile in 128-byte units (tn_swz / tn_frag_issue)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* halo_base = smem;
  unsigned char* bring = smem + HB * CV_HALO_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int fr = lane & (RT - 1), fh = lane / RT;  // the lane's row inside a tile, its k-chunk inside a k-step
  const int wm = BN == 128 ? wave >> 1 : wave, wn = BN == 128 ? wave & 1 : 0;
  const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  const int tm_i = p.nfast ? tile / p.tiles_n : tile % p.tiles_m, n0 = (p.nfast ? tile % p.tiles_n : tile / p.tiles_m) * BN;
  const int TW = p.cv_tw, TH = p.cv_th, W2 = TW + 2, HIMG = (TH + 2) * W2;
  const int H = p.g.OH, W = p.g.OW;
  // tile origin: image group, top-left pixel
  const unsigned tyx = fd_div((unsigned)tm_i, p.cv_div_tx);            // tm_i / tiles_x
  const int txi = tm_i - (int)tyx * p.cv_tiles_x;
  const unsigned tg = fd_div(tyx, p.cv_div_ty);                        // image-group index
  const int tyi = (int)tyx - (int)tg * p.cv_tiles_y;
  const int img0 = (int)tg * p.cv_ni, y0 = tyi * TH, x0 = txi * TW;
  const bool flip = p.g.mode == GATHER_DGRAD;
  const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(g_zero16);

  // ---- halo DMA plan: piece j of this wave = LDS bytes [(4j + wave) KiB, +1 KiB) = 8 halo pixels x 128 B
  int a_off[CV_HALO_PIECES];  // element offset of the lane's 16-byte chunk at channel 0, or -1 (padding / outside)
  const int nbatch = p.M / (H * W);
#pragma unroll
  for (int j = 0; j < CV_HALO_PIECES; ++j) {
    const int hp = (4 * j + wave) * 8 + (lane >> 3);
    const int slot = lane & 7;
    const int chunk = (slot ^ (hp >> 1)) & 7;
    const unsigned il = fd_div((unsigned)hp, p.cv_div_himg);
    const unsigned rem = (unsigned)hp - il * (unsigned)HIMG;
    const unsigned hr = fd_div(rem, p.cv_div_w2);
    const int hc = (int)(rem - hr * (unsigned)W2);
    const int iy = y0 + (int)hr - 1, ix = x0 + hc - 1, img = img0 + (int)il;
    const bool ok = (int)il < p.cv_ni && img < nbatch && iy >= 0 && iy < H && ix >= 0 && ix < W;
    a_off[j] = ok ? ((img * H + iy) * W + ix) * p.lda + chunk * 8 : -1;
  }
  auto issue_halo = [&](int j, int c0, int buf) {  // one piece
    const bf16_t* src = a_off[j] >= 0 ? p.A + (a_off[j] + c0) : zero_src;
    glds16(src, halo_base + buf * CV_HALO_BYTES + (4 * j + wave_u) * 1024);
  };
  // ---- weight DMA plan: tile [BN n][64 k]; wave w fills rows 8*BPW*w .. (BPW pieces of 8 rows).  BKM (forward: the Flax kernel
  // itself, [tap][Cin][Cout]): tile [64 k][BN n] in 2*BN-byte rows, wave w fills k rows 16w .. 16w+15 (BPW pieces), chunks
  // swizzled for the transposing reads (tn_swz)
  int b_off[BPW];
#pragma unroll
  for (int i = 0; i < BPW; ++i) {
    if (BKM) {
      constexpr int RPP = 1024 / (2 * BN);  // k rows per piece (4 at BN = 128, 8 at BN = 64), BN / 8 chunks per row
      const int rloc = (wave * BPW + i) * RPP + lane / (BN / 8);
      const int n = n0 + (((lane % (BN / 8)) ^ tn_swz<TMB>(rloc)) << 3);
      b_off[i] = n < p.N ? rloc * p.ldb + n : -1;
    } else {
      const int row = (wave * BPW + i) * 8 + (lane >> 3);
      const int chunk = ((lane & 7) ^ (row >> 1)) & 7;
      const int n = n0 + row;
      b_off[i] = n < p.N ? n * p.ldb + chunk * 8 : -1;
    }
  }
  auto issue_b = [&](int tap, int c0, int stage) {
    const bf16_t* bt = p.Bt + (long)tap * p.b_tap_stride + (BKM ? (long)c0 * p.ldb : (long)c0);
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
      const bf16_t* src = b_off[i] >= 0 ? bt + b_off[i] : zero_src;
      glds16(src, bring + stage * Cfg::B_BYTES + (wave_u * BPW + i) * 1024);
    }
  };

  // ---- fragment rows: tile pixel -> halo pixel at tap (0,0)
  int hp0[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int pix = wm * (NI * RT) + i * RT + fr;
    const int il = pix >> (p.cv_ltw + p.cv_lth);
    const int r = (pix >> p.cv_ltw) & (TH - 1), c = pix & (TW - 1);
    hp0[i] = il * HIMG + r * W2 + c;
  }

  constexpr int AE = MF16 ? 4 : 16;  // accumulator registers per MFMA tile
  typedef __attribute__((ext_vector_type(AE))) float acc_t;
  acc_t acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < AE; ++e) acc[i][j][e] = 0.f;

  const int nchunks = p.Kc / BK;
  int ch_beg = 0, ch_end = nchunks;
  if (SPLITK) {
    ch_beg = blockIdx.y * p.cv_chunks_per_split;
    ch_end = min(ch_beg + p.cv_chunks_per_split, nchunks);  // (never empty: conv_halo_plan)
  }
  // One k16-step = NI + 2 fragment reads + 2 NI MFMAs per wave.  The reads of step g+1 are issued in front of the MFMAs of step g
  // (one wave per SIMD: nothing else hides LDS latency), also across taps: the workgroup barrier that publishes tap+1's
  // weights sits in front of the LAST step of tap, where the next tap's DMA is issued as well.
  // The fragment reads and their waits are asm-owned: hipcc otherwise waits lgkmcnt(0) right behind the reads it has just
  // issued (it cannot count while scalar loads share the counter), which exposes the LDS latency on every step.
  bf16x8_t fa[2][NI], fb[2][NJ];
  TrFrag tfb[2][NJ];  // BKM: the weight fragments arrive as two transposing reads each
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned brow[NJ];  // byte offset of this lane's weight rows inside a ring stage, swizzle key folded in below
  int bkey[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int row = wn * 64 + j * RT + fr;
    brow[j] = row * LDS_ROW_BYTES;
    bkey[j] = (fh ^ (row >> 1)) & 7;
  }
  // byte offsets of the lane's activation rows inside a halo buffer at the current tap, k-step 0 (step s flips chunk bits:
  // ^ (KG * s << 4)).  Recomputed per tap behind an optimisation barrier: left alone, hipcc hoists all 9 x 4 x NI addresses out of
  // the chunk loop and parks them in AGPRs (a v_accvgpr_read in front of every fragment read).
  unsigned abase[NI];
  auto set_tap = [&](int tapoff) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = hp0[i] + tapoff;
      unsigned v = row * LDS_ROW_BYTES + (((fh ^ (row >> 1)) & 7) << 4);
      asm volatile("" : "+v"(v));
      abase[i] = v;
    }
  };
  auto load_frags = [&](unsigned ha_off, unsigned hb_off, int s, bf16x8_t* a, bf16x8_t* b, TrFrag* tb) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned addr = lds0 + ha_off + (abase[i] ^ (unsigned)((KG * s) << 4));
      asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"(addr) : "memory");
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (BKM) {
        if (MF16) tn_frag_issue16<TMB>(tb[j], lds0 + hb_off, wn * 64 + j * RT, s, lane);
        else tn_frag_issue<TMB>(tb[j], lds0 + hb_off, wn * 64 + j * RT, s, lane);
      } else {
        const unsigned addr = lds0 + hb_off + brow[j] + ((bkey[j] ^ (KG * s)) << 4);
        asm volatile("ds_read_b128 %0, %1" : "=v"(b[j]) : "v"(addr) : "memory");
      }
    }
  };
  // (all of a step's reads are waited for together: lgkmcnt(0); the next step's reads are only issued behind this step's MFMAs)
#define CV_FRAG_WAIT(N, a, b, tb)                                                                                              \
  do {                                                                                                                         \
    if (BKM) {                                                                                                                 \
      asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a[0]), "+v"(a[1]), TR_OPS1(tb[0]), TR_OPS1(tb[1])::"memory");           \
      if (NJ == 4) asm volatile("" : TR_OPS1(tb[NJ - 2]), TR_OPS1(tb[NJ - 1]));                                                \
      _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_) b[j_] = tr_value(tb[j_]);                                              \
    } else {                                                                                                                   \
      asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1])::"memory");                    \
      if (NJ == 4) asm volatile("" : "+v"(b[NJ - 2]), "+v"(b[NJ - 1]));                                                        \
    }                                                                                                                          \
    if (NI == 4) asm volatile("" : "+v"(a[NI - 2]), "+v"(a[NI - 1]));  /* the other fragments: defined behind the wait too (volatile asms keep their order) */ \
  } while (0)
  auto mfma_step = [&](const bf16x8_t* a, const bf16x8_t* b) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if constexpr (MF16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
      }
  };
  auto tap_off = [&](int tap) {
    const int kh = tap / 3, kw = tap - 3 * (tap / 3);
    return (flip ? 2 - kh : kh) * W2 + (flip ? 2 - kw : kw);
  };
  constexpr unsigned BRING = HB * CV_HALO_BYTES;
  // prologue: whole first halo and the weights of tap 0; then the weights of taps 1 and 2 go out and the first fragments come in
#pragma unroll
  for (int j = 0; j < CV_HALO_PIECES; ++j) issue_halo(j, ch_beg * BK, 0);
  issue_b(0, ch_beg * BK, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  issue_b(1, ch_beg * BK, 1);
  issue_b(2, ch_beg * BK, 2);
  set_tap(tap_off(0));
  load_frags(0, BRING, 0, fa[0], fb[0], tfb[0]);
  int hbuf = 0;
  for (int chunk = ch_beg; chunk < ch_end; ++chunk, hbuf ^= (HB - 1)) {
    const bool more = chunk + 1 < ch_end;
    const unsigned ha = hbuf * CV_HALO_BYTES;
    const unsigned ha_next = (hbuf ^ (HB - 1)) * CV_HALO_BYTES;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {  // unrolled: piece indices and ring stages (9 % 3 == 0) are compile-time
      const unsigned hb = BRING + (tap % CV_NSTB) * Cfg::B_BYTES;
#pragma unroll
      for (int s = 0; s < KS - 1; ++s) {  // MFMAs first: the next step's reads and (below) the DMA are issued in their shadow
        CV_FRAG_WAIT(0, fa[s & 1], fb[s & 1], tfb[s & 1]);
        mfma_step(fa[s & 1], fb[s & 1]);
        load_frags(ha, hb, s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1], tfb[(s + 1) & 1]);
      }
      // publish tap+1: its weights were issued TWO taps ago (a tap lasts ~0.7 us; a weight tile that is not L2-hot takes longer
      // than that to land: in the training step the longer lead is worth 0.27 ms, back-to-back runs of one layer, whose weights
      // stay in L2, do not care).  DMA loads land in issue order, so "all but the newest BPW" = everything except the weight
      // tile issued a tap ago: tap+1's weights and every halo piece are in.
      // The stage that takes the new tile (tap+3 -> stage tap % 3) is the one this tap reads: every wave finishes its last
      // fragment reads of it BEFORE the barrier (lgkmcnt(0) first), so nobody's read is still in flight when the DMA is issued.
      CV_FRAG_WAIT(0, fa[1], fb[1], tfb[1]);
      if (tap <= 6 || more) {
        if (BPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last chunk's taps 6..8 issue nothing: the newest tile IS tap+1's)
      }
      __builtin_amdgcn_s_barrier();
      mfma_step(fa[1], fb[1]);
      set_tap(tap_off(tap + 1 < 9 ? tap + 1 : 0));
      if (tap + 1 < 9) load_frags(ha, BRING + ((tap + 1) % CV_NSTB) * Cfg::B_BYTES, 0, fa[0], fb[0], tfb[0]);
      else if (HB == 2 && more) load_frags(ha_next, BRING, 0, fa[0], fb[0], tfb[0]);
      if (HB == 2 && tap < 7 && more && !NT_DBG(32)) {  // next chunk's halo: two pieces per tap (the 14th slot repeats piece 12)
        issue_halo(2 * tap < CV_HALO_PIECES ? 2 * tap : CV_HALO_PIECES - 1, (chunk + 1) * BK, hbuf ^ 1);
        issue_halo(2 * tap + 1 < CV_HALO_PIECES ? 2 * tap + 1 : CV_HALO_PIECES - 1, (chunk + 1) * BK, hbuf ^ 1);
      }
      if (HB == 1 && tap == 8 && more) {
        // one halo buffer: every wave's reads of it finished before this tap's barrier, so the next chunk's halo goes out now,
        // in front of the weights of its tap 2; it must have landed (all but the newest weight tile) before anyone reads it.
        // Nothing of THIS workgroup overlaps the fetch: the workgroup sharing the CU does.
#pragma unroll
        for (int j = 0; j < CV_HALO_PIECES; ++j) issue_halo(j, (chunk + 1) * BK, 0);
      }
      if (NT_DBG(16)) continue;  // developer ablation (SDT_NT_DBG, also bit 32 above): no weight / halo traffic in the loop, wrong results
      if (tap + 3 < 9) issue_b(tap + 3, chunk * BK, tap % CV_NSTB);
      else if (more) issue_b(tap + 3 - 9, (chunk + 1) * BK, tap % CV_NSTB);
      if (HB == 1 && tap == 8 && more) {
        if (BPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        load_frags(0, BRING, 0, fa[0], fb[0], tfb[0]);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef CV_FRAG_WAIT
  __syncthreads();  // ring and halo are free: the epilogue reuses the LDS
  if (NT_DBG(64)) return;  // developer ablation: no epilogue (wrong results): what the store path costs per tile

  // output row (GEMM m) of tile pixel `pix`
  auto out_row = [&](int pix) -> long {
    const int il = pix >> (p.cv_ltw + p.cv_lth);
    const int r = (pix >> p.cv_ltw) & (TH - 1), c = pix & (TW - 1);
    const int img = img0 + il;
    return img < nbatch ? ((long)img * H + (y0 + r)) * W + (x0 + c) : -1;
  };

  if (SPLITK) {  // only the split that arrives last at this tile goes on, with the complete sums (split_reduce)
    float nob[1] = {0.f};
    constexpr int NV = NI * NJ * AE / 16;  // the accumulators as 16-register groups (the slab is a per-thread register dump)
    if (!split_reduce<NV, 1>(p.slab, p.tile_cnt, (int)gridDim.y, (int)blockIdx.y, reinterpret_cast<f32x16_t(&)[NV]>(acc), nob, false, 0,
                                 tile, smem, tid))
      return;
    __syncthreads();
  }

  // ---- epilogue: (+bias) -> bf16 -> LDS C tile [256][BN+8] -> coalesced 16-byte stores (+rowbias, +residual)
  constexpr int CP = BN + 8;
  constexpr int CCH = BN / 8;        // 16-byte chunks per output row
  constexpr int RPP = 256 / CCH;     // rows stored per pass of the workgroup
  static_assert(CV_BM * CP * 2 + 8 * BN * 4 <= CV_HALO_BYTES * HB, "C tile and statistics scratch fit the halo buffers");
  bf16_t* sc = reinterpret_cast<bf16_t*>(smem);
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ml = wm * (NI * RT) + i * RT + fr;
#pragma unroll
      for (int g4 = 0; g4 < AE / 4; ++g4) {  // (D[row = channel][col = pixel]: 32x32: rows 8 g4 + 4 fh + e; 16x16: rows 4 fh + e)
        const int nl = wn * 64 + j * RT + 8 * g4 + 4 * fh;
        float v0 = acc[i][j][4 * g4 + 0], v1 = acc[i][j][4 * g4 + 1], v2 = acc[i][j][4 * g4 + 2], v3 = acc[i][j][4 * g4 + 3];
        if (p.bias && n0 + nl < p.N) {
          const float4 bv = *reinterpret_cast<const float4*>(p.bias + n0 + nl);
          v0 += bv.x; v1 += bv.y; v2 += bv.z; v3 += bv.w;
        }
        uint2 pk;
        pk.x = pack2bf(v0, v1);
        pk.y = pack2bf(v2, v3);
        *reinterpret_cast<uint2*>(sc + ml * CP + nl) = pk;
      }
    }
  __syncthreads();
  {
    const int cc = tid % CCH, rr = tid / CCH;
    const int n = n0 + cc * 8;
    float gns[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gnq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int ps = 0; ps < CV_BM / RPP; ++ps) {
      const int ml = rr + RPP * ps;
      const long m = out_row(ml);
      if (m >= 0 && n < p.N) {
        uint4 v = *reinterpret_cast<const uint4*>(sc + ml * CP + cc * 8);
        if (p.rowbias || p.residual) {
          float f[8], g[8];
          unpack8(v, f);
          if (p.rowbias) {
            unpack8(*reinterpret_cast<const uint4*>(p.rowbias + (long)(img0 + (ml >> (p.cv_ltw + p.cv_lth))) * p.ldrb + n), g);  // rows_per_batch = H*W: the row's image
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += g[e];
          }
          if (p.residual) {
            unpack8(*reinterpret_cast<const uint4*>(p.residual + m * p.ldres + n), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += g[e];
          }
          v = pack8(f);
        }
        *reinterpret_cast<uint4*>(p.C + m * p.ldc + n) = v;
        if (p.gn_stats) gn_accum(gns, gnq, v);
      }
    }
    if (p.gn_stats) gn_tile_flush<CCH>(p, gns, gnq, n0, img0, tyi * p.cv_tiles_x + txi, reinterpret_cast<float*>(smem + CV_BM * CP * 2), tid);
  }
}

// =====================================================================================================
// weight gradients are written once per step and read much later (global norm, optimizer): streaming stores keep them out of
// the L2 the GEMMs' operands live in (-0.38 ms per SD1.5 step, same-box)
#define WG_STORE(ptr, v) __builtin_nontemporal_store((v), (ptr))
struct GemmTnParams {
  const bf16_t* A;   // gathered operand (activations x)
  const bf16_t* B;   // dY [M][ldb]
  float* dW;         // [taps][K1_out][ldw] fp32, accumulated atomically
  int M, K1, N;      // K1 = padded channel count read from A; N = padded column count read from B
  int K1_valid, N_valid;  // logical dims of dW actually written
  int lda, ldb, ldw;
  long w_tap_stride;
  int tiles_k1, tiles_n, rows_per_split;
  int splits, taps;  // the reduction over M is cut into `splits` row ranges; workgroups = splits x taps x tiles (see tn_place)
  int out_bf16;      // dW is a bf16 buffer [r4]: the fp32 sums are rounded ONCE (RNE) when they are stored - the precision the reference's
                     // kernel cotangents have (flax Dense / Conv with dtype=bfloat16 hand back a bf16 value widened to fp32); offsets,
                     // pitches and strides stay in elements.  The bias gradient and the squared-norm slots stay fp32 / double.
  int n_seg;         // > 0: output columns are cut into segments of n_seg, segment s starts at dW + s*seg_stride (merged q/k/v weights)
  long seg_stride;
  float* dbias;      // optional: db[n] = sum_m dY[m][n], done by the k1-tile-0 / tap-0 workgroups from the dY tiles they stage
  unsigned char* slab;  // reduction split over M (gridDim.z > 1): per-workgroup fp32 partial tiles, [tile group][split][TnSlab bytes]
  int* tile_cnt;        // ... and one arrival counter per tile group (zero on entry, zero on exit)
  // optional: sum of squares of the gradient values this launch stores, for clip_by_global_norm without a pass over the finished
  // gradient buffer: the wave that stores a block adds up its squares in double (every product of two floats is exact there) and
  // WRITES the sum to the block's own slot - sq[(tap * sq_nk + k1 / 32) * sq_nn + n / 32], one writer per slot, slots of 32 x 32
  // blocks no wave starts at stay as the caller zeroed them; the caller adds the slots in index order (sdt_sum_f64_accumulate)
  double* sq;
  int sq_nk, sq_nn;
  GatherDesc g;
};
// slot geometry of the fused squared-norm partials (GemmTnParams.sq): 32 x 32 blocks over whole 128-tiles, whatever tile the plan picks
static void tn_set_sq(GemmTnParams* p, double* slots) {
  p->sq = slots;
  p->sq_nk = 4 * sdt_ceil_div(p->K1, 128);
  p->sq_nn = 4 * sdt_ceil_div(p->N, 128);
}
// one wave's double sum -> its slot (lane 0 writes)
__device__ __forceinline__ void wg_sq_flush(double v, double* slot, int lane) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (lane == 0) *slot = v;
}

// ---- bf16 gradient stores [r4]: one 2-byte store per element (lanes walk n: 64 contiguous bytes per half wave and register).  Lane
// pairs exchanging values for 4-byte stores - half the store instructions - measured SLOWER (4.89 vs 4.81 ms over the replayed launches
// of a step: two cross-lane moves per pair cost more than the store they save); `sq` takes the squares of the values AS STORED.
// ---------------------------------------------------------------------------------------------------------------
// Weight-gradient kernel.  Both operands are reduction-major in memory (A_g[m][k1], dY[m][n]), i.e. the MFMA k index
// is the ROW of the staged tile.  Tiles are staged row-major by LDS-DMA ([64 m][EDGE cols], 16-byte chunks XOR-swizzled
// on the source side) and the k-contiguous fragments are produced by the hardware transposing read
// ds_read_b64_tr_b16 (a 16-lane group reads a 4 row x 16 column block; lane i receives column i of the 4 rows),
// so no register transposes and no VGPR staging are needed.  NST-stage ring with counted vmcnt as in the NT kernel.

// One wave's TM x TM accumulator blocks -> dW (rows krow0 .., columns ncol0 ..) + its squared-norm slot.
// D[row = k1_local][col = n_local]: lanes walk n; single-writer stores, float32 or (out_bf16) rounded once to bf16.
template <int TM>
__device__ __forceinline__ void tn_store_tiles(const GemmTnParams& p, const int tap, const int krow0, const int ncol0, f32x16_t (&acc)[TM][TM],
                                               const int lane) {
  const int fr = lane & 31, fh = lane >> 5;
  float* wbase = p.dW + (long)tap * p.w_tap_stride;
  double sq = 0.0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int n = ncol0 + j * 32 + fr;
      long ncol = n;
      if (p.n_seg > 0) {
        const int seg = n / p.n_seg;
        ncol = (long)seg * p.seg_stride + (n - seg * p.n_seg);
      }
      if (p.out_bf16) {
        bf16_t* wb = reinterpret_cast<bf16_t*>(p.dW) + (long)tap * p.w_tap_stride;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int k1 = krow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
          if (k1 < p.K1_valid && n < p.N_valid) {
            const bf16_t h = f2bf(acc[i][j][e]);
            __builtin_nontemporal_store(h, wb + (long)k1 * p.ldw + ncol);
            sq = fma((double)bf2f(h), (double)bf2f(h), sq);
          }
        }
        continue;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k1 = krow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (k1 < p.K1_valid && n < p.N_valid) {
          WG_STORE(&wbase[(long)k1 * p.ldw + ncol], acc[i][j][e]);
          sq = fma((double)acc[i][j][e], (double)acc[i][j][e], sq);
        }
      }
    }
  if (p.sq) wg_sq_flush(sq, p.sq + ((long)tap * p.sq_nk + ((krow0) >> 5)) * p.sq_nn + ((ncol0) >> 5), lane);
}

// gemm_tn_body(tile, tap, split `me` of `nsplit`, ntaps): what blockIdx carries in the one-problem launch and what the grouped
// launch (gemm_tn_group_kernel) derives from its tile table.
// MODE 0: plain rows (Dense layers, 1x1 convolutions); 1: convolution gather with the rows walked incrementally; 2: the gather
// recomputed per load (images of fewer than 64 pixels).
#define TN_PLAIN 0
#define TN_WALK 1
#define TN_GENERIC 2
template <int TM, int MODE>
__device__ __forceinline__ void gemm_tn_body(const GemmTnParams& p, const int tile, const int tap, const int me, const int nsplit,
                                             const int ntaps, unsigned char* smem) {
  using Cfg = TnCfg<TM>;
  constexpr int EDGE = Cfg::EDGE, RB = Cfg::RB, CPR = Cfg::CPR, RPI = Cfg::RPI, IPW = Cfg::IPW;
  constexpr int TILE_BYTES = Cfg::TILE_BYTES, NST = Cfg::NST, KB = Cfg::KB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds0 = lds_offset_of(smem);
  const int k0 = (tile % p.tiles_k1) * EDGE, n0 = (tile / p.tiles_k1) * EDGE;
  const int kh = tap / p.g.KW, kw = tap - kh * p.g.KW;
  const int mbeg = me * p.rows_per_split;
  const int mend = min(mbeg + p.rows_per_split, p.M);
  const int T = mbeg < mend ? (mend - mbeg + KB - 1) / KB : 0;  // (the launcher never creates an empty split; it would add zeros)

  // ---- DMA plan: instruction j of this wave covers tile rows (j*4 + wave)*RPI .. +RPI-1; lane -> (row, LDS slot)
  const int lrow = lane / CPR, slot = lane % CPR;
  const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(g_zero16);
  int r_m[IPW], r_y[IPW], r_x[IPW], r_pix[IPW], gch[IPW];
  // plain rows [r4]: running source pointers and column-validity bits, everything the issue path needs in registers (the grouped
  // launch reads its problem from a table in the kernel arguments: left to the compiler, every DMA of the loop re-fetched lda / K1
  // / the base pointers through scalar loads and waited for them - 67 s_load + 50 s_waitcnt lgkmcnt(0) in the kernel)
  const bf16_t* pa[IPW];
  const bf16_t* pb[IPW];
  bool ca[IPW], cb[IPW];
  const long a_step = (long)KB * p.lda, b_step = (long)KB * p.ldb;
  const int stepY = KB / p.g.OW, stepX = KB - stepY * p.g.OW;
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int rloc = (j * 4 + wave) * RPI + lrow;
    gch[j] = (slot ^ tn_swz<TM>(rloc)) << 3;  // element offset of the global chunk this lane fetches
    const int m = mbeg + rloc;
    r_m[j] = m; r_y[j] = 0; r_x[j] = 0; r_pix[j] = 0;
    if (MODE == TN_PLAIN) {
      ca[j] = k0 + gch[j] < p.K1;
      cb[j] = n0 + gch[j] < p.N;
      pa[j] = p.A + ((long)m * p.lda + k0 + gch[j]);
      pb[j] = p.B + ((long)m * p.ldb + n0 + gch[j]);
    }
    if (MODE == TN_WALK) {
      const unsigned b = fd_div((unsigned)m, p.g.div_ohw);
      const unsigned rem = (unsigned)m - b * p.g.div_ohw.d;
      const unsigned oy = fd_div(rem, p.g.div_ow);
      r_y[j] = (int)oy; r_x[j] = (int)(rem - oy * p.g.div_ow.d); r_pix[j] = (int)b * p.g.IH * p.g.IW;
    }
  }
  auto stage = [&](int st) {  // issues the DMA of the NEXT tile in sequence (row cursors advance by KB)
    unsigned char* sa = smem + st * 2 * TILE_BYTES + wave_u * 1024;
    if (MODE == TN_PLAIN) {
#pragma unroll
      for (int j = 0; j < IPW; ++j) {
        const bool vm = r_m[j] < mend;
        const bf16_t* srca = vm ? pa[j] : zero_src;
        const bf16_t* srcb = vm ? pb[j] : zero_src;
        // (columns past K1 / N - the half-empty edge tiles of 320-wide layers - are staged as zeros from the zero page; letting those
        //  lanes sit the DMA out instead measured 5 - 7 % SLOWER on the level-0 groups: 368 -> 394 us, 662 -> 700 us)
        glds16(ca[j] ? srca : zero_src, sa + j * 4096);
        glds16(cb[j] ? srcb : zero_src, sa + TILE_BYTES + j * 4096);
        r_m[j] += KB;
        pa[j] += a_step;
        pb[j] += b_step;
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const int m = r_m[j];
      const bool vm = m < mend;
      const int acol = k0 + gch[j], bcol = n0 + gch[j];
      bool va = vm && acol < p.K1;
      const bool vb = vm && bcol < p.N;
      int aoff;
      if (MODE == TN_GENERIC) {
        const unsigned mm = vm ? (unsigned)m : 0u;
        const unsigned b = fd_div(mm, p.g.div_ohw);
        const unsigned rem = mm - b * p.g.div_ohw.d;
        const unsigned oy = fd_div(rem, p.g.div_ow);
        const long off = gather_src(p.g, (int)b, (int)oy, (int)(rem - oy * p.g.div_ow.d), kh, kw, p.lda);
        va = va && off >= 0;
        aoff = (int)off;
      } else {
        const int sy = r_y[j] * p.g.stride + kh - p.g.pad_t, sx = r_x[j] * p.g.stride + kw - p.g.pad_l;
        va = va && sy >= 0 && sx >= 0 && sy < p.g.IH && sx < p.g.IW;
        aoff = (r_pix[j] + sy * p.g.IW + sx) * p.lda;
      }
      const bf16_t* srca = va ? p.A + (aoff + acol) : zero_src;
      const bf16_t* srcb = vb ? p.B + (m * p.ldb + bcol) : zero_src;
      glds16(srca, sa + j * 4096);
      glds16(srcb, sa + TILE_BYTES + j * 4096);
      r_m[j] += KB;
      if (MODE == TN_WALK) {
        r_x[j] += stepX; r_y[j] += stepY;
        while (r_x[j] >= p.g.OW) { r_x[j] -= p.g.OW; ++r_y[j]; }
        while (r_y[j] >= p.g.OH) { r_y[j] -= p.g.OH; r_pix[j] += p.g.IH * p.g.IW; }
      }
    }
  };

  f32x16_t acc[TM][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // fused bias gradient: db[n] = sum_m 1 * dY[m][n]  == one more MFMA row block with an all-ones A operand
  const bool do_bias = p.dbias != nullptr && k0 == 0 && tap == 0 && (wave >> 1) == 0;  // wave-uniform
  f32x16_t bacc[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) bacc[j][e] = 0.f;
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;

  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  constexpr int WE = 32 * TM;
  constexpr int LPT = 2 * IPW;  // DMA instructions each wave issues per tile
  TrBase ba[TM], bb[TM];        // fragment bases inside a staged tile (tn_frag_base)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    ba[i] = tn_frag_base<TM>(wm * WE + i * 32, lane);
    bb[i] = tn_frag_base<TM>(wn * WE + i * 32, lane);
  }

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < T) stage(s);
  for (int t = 0; t < T; ++t) {
    // tile t has landed; younger than its DMA and allowed to stay in flight: the staging DMAs of the tiles behind it
    const int ahead = min(NST - 2, T - 1 - t);
    wait_vmcnt(ahead * LPT);
    __builtin_amdgcn_s_barrier();
#if defined(TN_ABL) && (TN_ABL & 1)  // developer timing ablation (compile time only, wrong results): no staging behind the prologue
    if (t + NST - 1 < T && t < 0) stage((t + NST - 1) % NST);
#else
    if (t + NST - 1 < T) stage((t + NST - 1) % NST);
#endif
#if defined(TN_ABL) && (TN_ABL & 2)  // ... no fragment reads, no MFMAs
    continue;
#endif
    const unsigned sa = lds0 + (t % NST) * 2 * TILE_BYTES;
    const unsigned sb = sa + TILE_BYTES;
    unsigned aa[TM][2], ab[TM][2];  // this stage's fragment addresses (K16-step 0)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      aa[i][0] = sa + ba[i].lo; aa[i][1] = sa + ba[i].hi;
      ab[i][0] = sb + bb[i].lo; ab[i][1] = sb + bb[i].hi;
    }
    TrFrag fa[2][TM], fb[2][TM];
    auto issue = [&](auto S, TrFrag* a, TrFrag* b) {
      constexpr int OFF = decltype(S)::value * 16 * RB;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        tr_read_at<OFF>(a[i], aa[i][0], aa[i][1]);
        tr_read_at<OFF>(b[i], ab[i][0], ab[i][1]);
      }
    };
    issue(IntC<0>{}, fa[0], fb[0]);
    static_for<0, KB / 16>([&](auto S) {
      constexpr int s = decltype(S)::value;
      TrFrag* ca = fa[s & 1];
      TrFrag* cb = fb[s & 1];
      if constexpr (s + 1 < KB / 16) {  // next step's reads go out before this step's MFMAs; waits count them as "younger"
        issue(IntC<s + 1>{}, fa[(s + 1) & 1], fb[(s + 1) & 1]);
        if (TM == 1) TR_WAIT2(4, ca[0], cb[0]); else TR_WAIT4(8, ca[0], ca[TM - 1], cb[0], cb[TM - 1]);
      } else {
        if (TM == 1) TR_WAIT2(0, ca[0], cb[0]); else TR_WAIT4(0, ca[0], ca[TM - 1], cb[0], cb[TM - 1]);
      }
      bf16x8_t af[TM], bfr[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        af[i] = tr_value(ca[i]);
        bfr[i] = tr_value(cb[i]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int j = 0; j < TM; ++j) bacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bfr[j], bacc[j], 0, 0, 0);
      }
    });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // reduction split over M: only the last-arriving split of this (tile, tap) goes on, holding the complete sums
  float bv[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) bv[j] = bacc[j][0];  // every accumulator row holds the column sum: row 0 = register 0 of lane half 0
  const bool bias_lane = do_bias && fh == 0;
  __syncthreads();  // all waves have left the staging ring (split_reduce reuses its first word)
#if defined(TN_ABL) && (TN_ABL & 4)  // ... no slab publish, no output
  if (acc[0][0][0] != 12345.678f) return;
#endif
  if (!split_reduce<TM * TM, TM>(p.slab, p.tile_cnt, nsplit, me, reinterpret_cast<f32x16_t(&)[TM * TM]>(acc), bv, bias_lane, wn * WE + fr,
                                     tile * ntaps + tap, smem, tid))
    return;
  tn_store_tiles<TM>(p, tap, k0 + wm * WE, n0 + wn * WE, acc, lane);
  if (bias_lane) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int n = n0 + wn * WE + j * 32 + fr;
      if (n < p.N_valid) p.dbias[n] = bv[j];
    }
  }
}

// Placement of the weight-gradient workgroups [r4].  All tiles (and taps) of one split read the SAME rows of A and dY, so the
// workgroups are numbered (split, tap, tile) with the tile fastest.  Plain-row problems (Dense layers): every XCD takes a
// CONTIGUOUS run of that order (blocks b and b + 8 share an XCD and are dispatched one after the other), so a split's row panels
// are fetched into ONE L2 and met there by all its tiles, which walk the rows in step, instead of being dealt tile by tile over
// the eight L2s (round 3: every XCD fetched every panel - 1,021 MB of fabric reads per grouped launch, 367 MB now).  Measured on
// the replayed launches of a step (tools/tn_group_micro.py): the Dense groups 2,505 -> 2,305 us; the 3x3 kernels the other way
// (2,606 -> 2,675 us with 757 -> 399 MB of reads per launch: their re-reads are served by the Infinity Cache and were never what
// bound them), so those keep the round-robin deal.  A speed hint only: the slab hand-off (split_reduce) assumes nothing about placement.
template <int TM, int MODE>
__global__ void __launch_bounds__(256, TN_WPS) gemm_tn_kernel(const GemmTnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tiles = p.tiles_k1 * p.tiles_n;
  const int L = MODE == TN_PLAIN ? xcd_remap(blockIdx.x, tiles * p.taps * p.splits) : (int)blockIdx.x;
  const int tile = L % tiles, rest = L / tiles;
  gemm_tn_body<TM, MODE>(p, tile, rest % p.taps, rest / p.taps, p.splits, p.taps, smem);
}

// Several Dense-layer weight gradients as ONE launch (sdt_gemm_tn_wgrad_group): the weight gradients of a transformer block are
// independent of the input-gradient chain, individually small (25 - 100 tiles, 10 - 25 us of mostly prologue, slab traffic and
// tail), and each used to be a launch of its own.  Workgroup b serves problem i where wg_end[i-1] <= b < wg_end[i]; inside a
// problem the workgroups are (split, tile) with the tile fastest.  Problems keep their own split plan, slabs and counters.

// [r4] The workgroups of the launch in order (problem, split, tile) are cut into eight contiguous runs of equal WORK (K-steps +
// a fixed cost per workgroup: problems differ in rows per split), one per XCD: xcd_begin[x] .. xcd_begin[x + 1].  Block b is
// the (b >> 3)-th workgroup of run b & 7; the grid is 8 x the longest run and the surplus blocks of shorter runs leave at once.
#define TN_GROUP_MAX 16
struct GemmTnGroupParams {
  int n;
  int wg_end[TN_GROUP_MAX];
  int xcd_begin[9];
  GemmTnParams prob[TN_GROUP_MAX];
};
template <bool CONTIGUOUS>
__device__ __forceinline__ int tn_group_place(const GemmTnGroupParams& gp) {
#ifdef TN_PLACE_RR  // developer A/B: round 3's placement everywhere
  return (int)blockIdx.x < gp.xcd_begin[8] ? (int)blockIdx.x : -1;
#endif
  if (!CONTIGUOUS) return (int)blockIdx.x < gp.xcd_begin[8] ? (int)blockIdx.x : -1;  // consecutive workgroups dealt over the XCDs
  const int x = blockIdx.x & 7;
  const int L = gp.xcd_begin[x] + (int)(blockIdx.x >> 3);
  return L < gp.xcd_begin[x + 1] ? L : -1;
}
template <int TM>
__global__ void __launch_bounds__(256, TN_WPS) gemm_tn_group_kernel(const GemmTnGroupParams gp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = tn_group_place<true>(gp);
  if (b < 0) return;
  int i = 0;
  while (i + 1 < gp.n && b >= gp.wg_end[i]) ++i;  // wave-uniform scalar walk over <= 16 entries
  const int local = b - (i ? gp.wg_end[i - 1] : 0);
  const GemmTnParams p = gp.prob[i];  // by value: the fields the body uses are fetched once, into scalar registers
  const int tiles = p.tiles_k1 * p.tiles_n;
  const int me = local / tiles, tile = local - me * tiles;
  gemm_tn_body<TM, TN_PLAIN>(p, tile, 0, me, p.splits, 1, smem);
}


// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of a 3x3 / stride 1 / pad 1 convolution, three taps (one kernel row kh, kw = 0..2) per workgroup:
// the 64-pixel dY tile is staged once for the three taps, and so is the input: pixel m of tap kw reads input pixel
// m + (kh-1)*W + (kw-1), i.e. the SAME row-major image [68 pixels][128 channels] shifted by kw rows, fetched by the
// transposing reads at a row offset.  What a shifted linear index gets wrong is masked: rows of dY whose y+kh-1 leaves
// the image are staged as zeros (kh is fixed per workgroup), and the one fragment element per image row whose x+kw-1
// leaves the image is zeroed in the register.  Tile: 128 input channels x 64 output channels, 3 x 32 accumulator
// registers per wave, two workgroups per CU; staged bytes per FLOP are ~half the one-tap kernel's.
#define W3_AROWS 68  // 66 needed (64 + one pixel either side), staged as 17 one-KiB pieces (wave 0 issues the odd one)
#define W3_A_BYTES (W3_AROWS * 256)
#define W3_B_BYTES (64 * 128)
#define W3_STAGE (W3_A_BYTES + W3_B_BYTES)
#define W3_NST 3     // two chunks of DMA in flight: a 64-pixel chunk is ~0.4 us of MFMAs against a 1-2 us global latency
#define W3_LDS_BYTES (W3_NST * W3_STAGE)

// (tile, kernel row kh, split `me` of `nsplit`): blockIdx in the one-problem launch, the tile table in the grouped one
__device__ __forceinline__ void conv_wgrad3_body(const GemmTnParams& p, const int tile, const int kh, const int me, const int nsplit,
                                                 unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds0 = lds_offset_of(smem);
  const int k0 = (tile % p.tiles_k1) * 128, n0 = (tile / p.tiles_k1) * 64;
  const int mbeg = me * p.rows_per_split;
  const int mend = min(mbeg + p.rows_per_split, p.M);
  const int T = mbeg < mend ? (mend - mbeg + BK - 1) / BK : 0;
  const int W = p.g.OW, H = p.g.OH;
  const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(g_zero16);

  // ---- DMA plan.  A: piece j of this wave = image rows (4j + wave)*4 .. +3 (256-byte rows, 16 chunks); row r holds input
  // pixel (chunk base) + (kh-1)*W - 1 + r.  B: piece j = rows (4j + wave)*8 .. +7 of the [64][64] dY tile.
  // running source pointers (advanced by one chunk per stage) + the scalars their validity depends on: no 64-bit multiplies,
  // no branches in the issue path (a select between two valid-to-form pointers compiles to v_cndmask)
  int a_q[5];            // linear input pixel of the lane's row, or far negative when the row / channel chunk is padding
  const bf16_t* a_ptr[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int rloc = (j * 4 + wave) * 4 + (lane >> 4);
    const int col = k0 + (((lane & 15) ^ tn_swz<2>(rloc)) << 3);
    const int q = mbeg + (kh - 1) * W - 1 + rloc;
    a_q[j] = (rloc < 66 && col < p.K1) ? q : -(1 << 30);  // rows past the halo and channels past K1 stay zero
    a_ptr[j] = p.A + ((long)q * p.lda + col);
  }
  int b_m[2], b_y[2], b_x[2];
  const bf16_t* b_ptr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rloc = (j * 4 + wave) * 8 + (lane >> 3);
    const int col = n0 + (((lane & 7) ^ tn_swz<1>(rloc)) << 3);
    const int m = mbeg + rloc;
    b_m[j] = col < p.N ? m : (1 << 30);
    const unsigned b = fd_div((unsigned)m, p.g.div_ohw);
    const unsigned rem = (unsigned)m - b * p.g.div_ohw.d;
    b_y[j] = (int)fd_div(rem, p.g.div_ow);
    b_x[j] = (int)rem - b_y[j] * W;
    b_ptr[j] = p.B + ((long)m * p.ldb + col);
  }
  // [r4] any width that is a multiple of 8 (round 3: powers of two only - the 96 / 48 / 24-wide levels of a 768 x 768 image fell back
  // to the nine-tap kernel): a 64-pixel chunk advances the row cursor by 64 / W rows and 64 % W pixels
  const int stepY = BK / W, stepX = BK - stepY * W;
  const long a_step = (long)BK * p.lda, b_step = (long)BK * p.ldb;
  auto stage = [&](int st) {
    unsigned char* sa = smem + st * W3_STAGE;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j == 4 && wave_u != 0) break;  // 17 pieces: the last one is wave 0's
      const bool va = (unsigned)a_q[j] < (unsigned)p.M;
      const bf16_t* src = va ? a_ptr[j] : zero_src;
      glds16(src, sa + (j * 4 + wave_u) * 1024);
      a_q[j] += BK;
      a_ptr[j] += a_step;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool vb = b_m[j] < mend && (unsigned)(b_y[j] + kh - 1) < (unsigned)H;
      const bf16_t* src = vb ? b_ptr[j] : zero_src;
      glds16(src, sa + W3_A_BYTES + (j * 4 + wave_u) * 1024);
      b_m[j] += BK;
      b_ptr[j] += b_step;
      b_x[j] += stepX;
      const int carry = b_x[j] >= W ? 1 : 0;
      b_x[j] -= carry * W;
      b_y[j] += stepY + carry;
      if (b_y[j] >= H) b_y[j] -= H;
      if (b_y[j] >= H) b_y[j] -= H;
    }
  };

  f32x16_t acc[3][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][i][e] = 0.f;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  // bias gradient: kh == 1 masks no dY row, so those workgroups (first channel tile, wm == 0 waves) sum dY's columns
  const bool do_bias = p.dbias != nullptr && k0 == 0 && kh == 1 && wm == 0;
  f32x16_t bacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) bacc[e] = 0.f;
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;

  TrBase ba3[6], bb3 = tn_frag_base<1>(wn * 32, lane);  // fragment bases inside a staged chunk: A by (kw, i), B
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int i = 0; i < 2; ++i) ba3[kw * 2 + i] = tn_frag_base<2>(wm * 64 + i * 32, lane, kw);
  int x_chunk = (int)((unsigned)mbeg - fd_div((unsigned)mbeg, p.g.div_ow) * (unsigned)W);  // mbeg mod W, then advanced chunk by chunk
  if (T > 0) stage(0);
  if (T > 1) stage(1);
  for (int t = 0; t < T; ++t) {
    // chunk t has landed; the chunk issued one iteration ago may stay in flight (7 pieces from wave 0, 6 from the others)
    if (t + 1 < T) {
      if (wave_u == 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // ... for every wave; everyone finished chunk t-1, whose stage takes chunk t+2
#if defined(TN_ABL) && (TN_ABL & 1)  // developer timing ablations (compile time only, wrong results): see gemm_tn_body
    if (t + 2 < T && t < 0) stage((t + 2) % W3_NST);
#else
    if (t + 2 < T) stage((t + 2) % W3_NST);
#endif
#if defined(TN_ABL) && (TN_ABL & 2)
    continue;
#endif
    const unsigned sa = lds0 + (t % W3_NST) * W3_STAGE;
    const unsigned sb = sa + W3_A_BYTES;
    const int x0 = x_chunk;  // x of the chunk's first pixel
    x_chunk += stepX;
    if (x_chunk >= W) x_chunk -= W;
    unsigned aa[6][2], ab[2];  // this stage's fragment addresses (K16-step 0)
#pragma unroll
    for (int f = 0; f < 6; ++f) { aa[f][0] = sa + ba3[f].lo; aa[f][1] = sa + ba3[f].hi; }
    ab[0] = sb + bb3.lo; ab[1] = sb + bb3.hi;
    TrFrag fa[2][6], fb[2];
    auto issue = [&](auto S, TrFrag* a, TrFrag& b) {
      constexpr int s = decltype(S)::value;
      tr_read_at<s * 16 * 128>(b, ab[0], ab[1]);
#pragma unroll
      for (int f = 0; f < 6; ++f) tr_read_at<s * 16 * 256>(a[f], aa[f][0], aa[f][1]);
    };
    issue(IntC<0>{}, fa[0], fb[0]);
    static_for<0, BK / 16>([&](auto S) {
      constexpr int s = decltype(S)::value;
      TrFrag* ca = fa[s & 1];
      TrFrag& cb = fb[s & 1];
      if constexpr (s + 1 < BK / 16) {
        issue(IntC<s + 1>{}, fa[(s + 1) & 1], fb[(s + 1) & 1]);
        TR_WAIT7(14, cb, ca[0], ca[1], ca[2], ca[3], ca[4], ca[5]);
      } else {
        TR_WAIT7(0, cb, ca[0], ca[1], ca[2], ca[3], ca[4], ca[5]);
      }
      const bf16x8_t bfr = tr_value(cb);
      // x of this lane's fragment element 0 (element j: x + j), mod W.  W and the offsets are multiples of 8, so an image row can
      // only begin at element 0 and only end behind element 7 of a lane's eight pixels
      const unsigned xl = (unsigned)(x0 + 16 * s + 8 * fh);
      const int xs = (int)(xl - fd_div(xl, p.g.div_ow) * (unsigned)W);
      const bool edge_l = xs == 0;                         // element 0 is the first pixel of an image row
      const bool edge_r = xs + 8 == W;                     // element 7 is the last pixel of an image row
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          bf16x8_t af = tr_value(ca[kw * 2 + i]);
          if (kw == 0) af[0] = edge_l ? (short)0 : af[0];
          if (kw == 2) af[7] = edge_r ? (short)0 : af[7];
          acc[kw][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[kw][i], 0, 0, 0);
        }
      }
      if (do_bias) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bfr, bacc, 0, 0, 0);
    });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  float bv[1] = {bacc[0]};
  const bool bias_lane = do_bias && fh == 0;
  __syncthreads();
#if defined(TN_ABL) && (TN_ABL & 4)
  if (acc[0][0][0] != 12345.678f) return;
#endif
  if (!split_reduce<6, 1>(p.slab, p.tile_cnt, nsplit, me, reinterpret_cast<f32x16_t(&)[6]>(acc), bv, bias_lane, wn * 32 + fr, tile * 3 + kh, smem, tid)) return;
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    float* wbase = p.dW + (long)(kh * 3 + kw) * p.w_tap_stride;
    const int n = n0 + wn * 32 + fr;
    double sq = 0.0;
    if (p.out_bf16) {  // (see tn_store_tiles)
      bf16_t* wb = reinterpret_cast<bf16_t*>(p.dW) + (long)(kh * 3 + kw) * p.w_tap_stride;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int k1 = k0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
          if (k1 < p.K1_valid && n < p.N_valid) {
            const bf16_t h = f2bf(acc[kw][i][e]);
            __builtin_nontemporal_store(h, wb + (long)k1 * p.ldw + n);
            sq = fma((double)bf2f(h), (double)bf2f(h), sq);
          }
        }
    } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k1 = k0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (k1 < p.K1_valid && n < p.N_valid) {
          WG_STORE(&wbase[(long)k1 * p.ldw + n], acc[kw][i][e]);
          sq = fma((double)acc[kw][i][e], (double)acc[kw][i][e], sq);
        }
      }
    }
    if (p.sq) wg_sq_flush(sq, p.sq + ((long)(kh * 3 + kw) * p.sq_nk + ((k0 + wm * 64) >> 5)) * p.sq_nn + ((n0 + wn * 32) >> 5), lane);
  }
  if (bias_lane) {
    const int n = n0 + wn * 32 + fr;
    if (n < p.N_valid) p.dbias[n] = bv[0];
  }
}

__global__ void __launch_bounds__(256, 2) conv_wgrad3_kernel(const GemmTnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tiles = p.tiles_k1 * p.tiles_n;
  const int L = blockIdx.x;  // (split, kernel row, tile), dealt round-robin over the XCDs: see gemm_tn_kernel
  const int tile = L % tiles, rest = L / tiles;
  conv_wgrad3_body(p, tile, rest % 3, rest / 3, p.splits, smem);
}

// Several 3x3 convolution weight gradients as one launch (sdt_conv_wgrad3_group), like gemm_tn_group_kernel for the Dense layers:
// inside a problem the workgroups are (split, kernel row, tile) with the tile fastest.
__global__ void __launch_bounds__(256, 2) conv_wgrad3_group_kernel(const GemmTnGroupParams gp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = tn_group_place<false>(gp);
  if (b < 0) return;
  int i = 0;
  while (i + 1 < gp.n && b >= gp.wg_end[i]) ++i;
  const int local = b - (i ? gp.wg_end[i - 1] : 0);
  const GemmTnParams p = gp.prob[i];  // by value (see gemm_tn_group_kernel)
  const int tiles = p.tiles_k1 * p.tiles_n;
  const int tile = local % tiles, rest = local / tiles;
  conv_wgrad3_body(p, tile, rest % 3, rest / 3, p.splits, smem);
}

// ================================================================== C ABI
static int fill_gather(GatherDesc* g, const SdtConvGeom* geom, int mode, const char* name) {
  g->mode = mode;
  g->KH = 1; g->KW = 1; g->stride = 1; g->pad_t = 0; g->pad_l = 0;
  g->IH = g->IW = g->OH = g->OW = 1;
  g->div_ohw = make_fastdiv(1);
  g->div_ow = make_fastdiv(1);
  if (mode == GATHER_PLAIN) return SDT_OK;
  SDT_CHECK_ARG(geom, "%s: conv geometry required", name);
  SDT_CHECK_ARG(geom->in_h > 0 && geom->in_w > 0 && geom->out_h > 0 && geom->out_w > 0 && geom->kh > 0 && geom->kw > 0 &&
                    geom->stride > 0 && geom->batch > 0,
                "%s: bad conv geometry", name);
  g->KH = geom->kh; g->KW = geom->kw; g->stride = geom->stride; g->pad_t = geom->pad_top; g->pad_l = geom->pad_left;
  if (mode == GATHER_DGRAD) {  // rows enumerate the conv input grid, source is dY on the output grid
    g->OH = geom->in_h; g->OW = geom->in_w; g->IH = geom->out_h; g->IW = geom->out_w;
  } else {
    g->OH = geom->out_h; g->OW = geom->out_w; g->IH = geom->in_h; g->IW = geom->in_w;
  }
  g->div_ohw = make_fastdiv((unsigned)(g->OH * g->OW));
  g->div_ow = make_fastdiv((unsigned)g->OW);
  return SDT_OK;
}

#ifdef SDT_NT_DBG
static int g_nt_dbg = getenv("SDT_NT_DBG") ? atoi(getenv("SDT_NT_DBG")) : 0;
extern "C" void sdt_dbg_set_nt(int bits) { g_nt_dbg = bits; }  // developer builds only: ablation bits per launch (tools/phase_overlap_probe.py)
static int nt_dbg_bits() { return g_nt_dbg; }
#else
static int nt_dbg_bits() { return 0; }
#endif
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// Tile order of an NT launch.  Workgroups that share an XCD (and its 4 MB L2) take a contiguous run of tiles (xcd_remap), so the
// order decides which operand that L2 keeps and which one streams in from the fabric once per tile column / row:
//   walking M: one column tile of the weights stays hot; the rows are fetched again for every column tile unless all of them fit;
//   walking N: a row panel is fetched once and meets every column tile; the weights are fetched once per XCD if they fit, else per panel.
// Bytes through the fabric under either order, smaller wins (level-0 Dense layers: 16384 x 320 rows against 0.2 - 1.6 MB of weights:
// 30 - 210 MB walking M, 12 - 24 MB walking N).  Same arithmetic per tile either way: results are identical bit for bit.
static int nt_tile_order(double a_bytes, double b_bytes, int tiles_m, int tiles_n) {
  static const int force = env_int("SDT_NT_NFAST", -1);  // developer A/B: 0 / 1
  if (force == 0 || force == 1) return force;
  const double cap = 2.5 * 1024 * 1024;  // what one XCD's L2 holds of an operand beside the other's stream
  const double walk_m = b_bytes + (a_bytes <= cap ? 8.0 * a_bytes : (double)tiles_n * a_bytes);
  const double walk_n = a_bytes + (b_bytes <= cap ? 8.0 * b_bytes : (double)tiles_m * b_bytes);
  return walk_n < walk_m ? 1 : 0;
}
// tile / split-K plan for the NT GEMM (shared by the workspace query and the launcher)
struct NtPlan {
  int tm;       // 2 -> 128x128 tiles, 1 -> 64x64
  int splits;   // >1 -> split-K through the fp32 workspace
  int ksteps_per_split;
};
static NtPlan plan_nt(int64_t M, int N, int Kc, int taps) {
  NtPlan pl;
  const long t128 = (long)sdt_ceil_div(M, 128) * sdt_ceil_div(N, 128);
  const long t64 = (long)sdt_ceil_div(M, 64) * sdt_ceil_div(N, 64);
  const int T = taps * sdt_ceil_div(Kc, BK);
  pl.splits = 1;
  pl.ksteps_per_split = T;
  int s = 1;
  if (t128 >= 256) {
    pl.tm = 2;  // >= one 128x128 tile per CU: the big tile (2x the MFMA work per staged byte) wins
  } else if (T >= 48 && t128 >= 16) {
    pl.tm = 2;  // deep reduction, few tiles (16x16 / 8x8 UNet levels): big tiles + split-K beat many small tiles
    static const int tgt2 = env_int("SDT_NT_SPLIT_WG2", 400), minsteps2 = env_int("SDT_NT_SPLIT_STEPS2", 12);
    s = (int)((tgt2 + t128 - 1) / t128);
    if (s > T / minsteps2) s = T / minsteps2;
  } else {
    pl.tm = 1;
    static const int tgt1 = env_int("SDT_NT_SPLIT_WG1", 480), minsteps1 = env_int("SDT_NT_SPLIT_STEPS1", 8);
    static const int mint1 = env_int("SDT_NT_SPLIT_MINT1", 32), maxt1 = env_int("SDT_NT_SPLIT_MAXT1", 160);
    if (t64 < maxt1 && T >= mint1) {
      s = (int)((tgt1 + t64 - 1) / t64);
      if (s > T / minsteps1) s = T / minsteps1;
    } else if (t64 <= 32 && T >= 8) {
      // a handful of tiles (time-embedding projections, M = batch): the K loop IS the kernel, so cut it short even though every
      // split costs an atomic round trip (measured (4,1280,1280): 13.8 -> 8.5 us; (308,768,768) with 60 tiles gets slower)
      s = T / 4;
    }
  }
  {  // developer sweeps: force the tile size (splits then follow the other tile's rule only roughly)
    static const int force_tm = env_int("SDT_NT_TM", 0);
    if (force_tm == 1 || force_tm == 2) pl.tm = force_tm;
  }
  if (s > 32) s = 32;
  if (s >= 2) {
    pl.ksteps_per_split = (T + s - 1) / s;
    pl.splits = (T + pl.ksteps_per_split - 1) / pl.ksteps_per_split;
  }
  return pl;
}

// Ring depth of an NT launch [r4]: the configuration's own (NtCfg::NST: sized so that two or three workgroups share a CU).  Developer
// switch SDT_NT_DEEP_RING=1: a grid that leaves every CU fewer workgroups than that gets the LDS the absent neighbours would have used
// as a deeper ring (<= 8 stages, <= 128 KB, no deeper than the reduction is long) - built to test whether the small launches (text
// tower, 8 x 8 / 16 x 16 levels) wait on issue->landed latency; they do not (a CU streams ~24 GB/s from HBM however much is in
// flight), the step did not move, and the default stays the configured depth.  ksteps: 64-wide K-steps of one workgroup's reduction.
template <int TM, int KB>
static int nt_ring_depth(long wgs, int ksteps) {
  using Cfg = NtCfg<TM, KB>;
  static const int on = env_int("SDT_NT_DEEP_RING", 0);  // measured: no gain on the step (39.48 vs 39.60 ms same-box), so off
  const int stage = 2 * Cfg::TILE_BYTES, steps = ksteps * (64 / KB);
  const long per_cu = (wgs + 255) / 256;
  int nst = (int)(128 * 1024 / per_cu / stage);
  if (nst > 8) nst = 8;
  if (nst > steps + 1) nst = steps + 1;
  if (!on || nst < Cfg::NST) nst = Cfg::NST;
  return nst;
}
#define NT_MAX_LDS (128 * 1024)

template <int TM, bool SPLITK, bool GENERIC, bool BKM>
static void launch_nt2(const GemmNtParams& p, int splits, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_kernel<TM, SPLITK, GENERIC, BKM>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_MAX_LDS);
    attr_set = true;
  }
  GemmNtParams q = p;
  q.nst = nt_ring_depth<TM, 64>((long)p.tiles_m * p.tiles_n * splits, p.ksteps_per_split);
  hipLaunchKernelGGL((gemm_nt_kernel<TM, SPLITK, GENERIC, BKM>), dim3(p.tiles_m * p.tiles_n, splits), dim3(256), (q.nst * 2 * NtCfg<TM, 64>::TILE_BYTES), stream, q);
}
// longest reduction (in 64-wide steps) that still runs the 32-wide-step 128-tile kernel (developer sweep: SDT_NT_K32_STEPS, 0 = off)
static int nt_k32_max_steps() {
  static const int v = env_int("SDT_NT_K32_STEPS", 20);
  return v;
}
template <int TM>
static void launch_nt_pack8(const GemmNtParams& p, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_kernel<TM, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_MAX_LDS);
    attr_set = true;
  }
  GemmNtParams q = p;
  q.nst = nt_ring_depth<TM, 64>((long)p.tiles_m * p.tiles_n, p.ksteps_per_split);
  hipLaunchKernelGGL((gemm_nt_kernel<TM, false, false, true, true>), dim3(p.tiles_m * p.tiles_n, 1), dim3(256), (q.nst * 2 * NtCfg<TM, 64>::TILE_BYTES), stream, q);
}
// 128-tiles with 32-wide K-steps, three workgroups per CU (NtCfg): unsplit launches with a short reduction
template <bool BKM>
static void launch_nt_k32(const GemmNtParams& p, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_kernel<2, false, false, BKM, false, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_MAX_LDS);
    attr_set = true;
  }
  GemmNtParams q = p;
  q.nst = nt_ring_depth<2, 32>((long)p.tiles_m * p.tiles_n, p.ksteps_per_split);
  hipLaunchKernelGGL((gemm_nt_kernel<2, false, false, BKM, false, 32>), dim3(p.tiles_m * p.tiles_n, 1), dim3(256), (q.nst * 2 * NtCfg<2, 32>::TILE_BYTES), stream, q);
}
template <int TM, bool SPLITK>
static void launch_nt(const GemmNtParams& p, int splits, bool b_kmajor, hipStream_t stream) {
  const bool generic = p.g.mode == GATHER_DGRAD && p.g.stride != 1;  // (input gradients only: never with a k-major B)
  if (generic) launch_nt2<TM, SPLITK, true, false>(p, splits, stream);
  else if (b_kmajor) launch_nt2<TM, SPLITK, false, true>(p, splits, stream);
  else launch_nt2<TM, SPLITK, false, false>(p, splits, stream);
}
template <int TM, int MODE>
static void launch_tn2(const GemmTnParams& p, int taps, int splits, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_tn_kernel<TM, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, TnCfg<TM>::LDS_BYTES);
    attr_set = true;
  }
  GemmTnParams q = p;
  q.taps = taps; q.splits = splits;
  hipLaunchKernelGGL((gemm_tn_kernel<TM, MODE>), dim3(p.tiles_k1 * p.tiles_n * taps * splits), dim3(256), TnCfg<TM>::LDS_BYTES, stream, q);
}
template <int TM>
static void launch_tn(const GemmTnParams& p, int taps, int splits, hipStream_t stream) {
  if (p.g.mode == GATHER_PLAIN) launch_tn2<TM, TN_PLAIN>(p, taps, splits, stream);
  else if (p.g.OH * p.g.OW < BK) launch_tn2<TM, TN_GENERIC>(p, taps, splits, stream);  // the incremental row walk needs >= 64 rows per image
  else launch_tn2<TM, TN_WALK>(p, taps, splits, stream);
}

// ---- weight-gradient plan: kernel, tile, split of the reduction over M, scratch (shared by the workspace query and the launcher)
struct TnPlan {
  bool w3;       // conv_wgrad3_kernel (3x3 / stride 1 / pad 1, width a multiple of 8): three taps per workgroup
  int tm;        // gemm_tn_kernel tile: 64 * tm
  int tiles_k1, tiles_n, groups;  // groups = tile x tap(-row) pairs = arrival counters
  int splits, rows_per_split;
  int64_t cnt_bytes, ws_bytes;    // counters, then splits x groups slabs (0 when the reduction is not split)
};
static TnPlan plan_tn(const GatherDesc& g, int gather_mode, int64_t M, int K1, int N, int taps, int n_seg, int target_override = 0) {
  TnPlan pl;
  static const int w3 = env_int("SDT_WGRAD3", 1);
  // [r4] from 256 output pixels (four 64-pixel chunks) on: the 8 x 8 level at batch 4 used to fall to the nine-tap kernel (900 workgroups
  // of four K-steps each, one tap per workgroup): its launches 90.7 -> 75.8 us (four problems) and 219.5 -> 150.9 us (eight)
  static const int w3_min_m = env_int("SDT_WGRAD3_MIN_M", 256);
  const int W = g.OW;
  const bool wok = W >= 8 && W % 8 == 0;
  pl.w3 = w3 && gather_mode == GATHER_FPROP && taps == 9 && g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad_t == 1 &&
          g.pad_l == 1 && g.IH == g.OH && g.IW == g.OW && wok && n_seg == 0 && M % BK == 0 && M >= w3_min_m;
  long base_wg;
  int target, min_rows, slab_bytes;
  if (pl.w3) {
    pl.tm = 0;
    pl.tiles_k1 = sdt_ceil_div(K1, 128); pl.tiles_n = sdt_ceil_div(N, 64);
    pl.groups = pl.tiles_k1 * pl.tiles_n * 3;
    base_wg = pl.groups;
    static const int t3 = env_int("SDT_WGRAD3_WG", 384);  // (same-box sweep: 192 / 256 / 384 -> 50.3 / 50.1 / 49.9 ms per step)
    static const int r3 = env_int("SDT_WGRAD3_MIN_ROWS", 512);
    target = t3; min_rows = r3; slab_bytes = TnSlab<6>::BYTES;
  } else {
    const long wg128 = (long)sdt_ceil_div(K1, 128) * sdt_ceil_div(N, 128) * taps;
    // 128-tiles stage half the bytes per FLOP: worth their tile-quantisation waste once there are enough of them
    // (measured: (16384,320,2560) 105 -> 68 us, (16384,320,320)x9 126 -> 100 us; small-M weights stay on 64-tiles)
    static const int force_tm = env_int("SDT_TN_TM", 0);  // developer sweeps
    // [r3] 128-tiles wherever both dimensions fill one: since the Dense weight gradients are issued in groups (sdt_gemm_tn_wgrad_group)
    // a launch no longer depends on ONE problem's tile count to fill the chip, and half the staged bytes per FLOP wins everywhere
    // (same-box: 43.5 -> 42.7 ms per SD1.5 step against the round-2 rule wg128 >= 512 || (wg128 >= 48 && M >= 4096))
    (void)wg128;
    pl.tm = force_tm ? force_tm : ((K1 >= 128 && N >= 128) ? 2 : 1);
    const int edge = 64 * pl.tm;
    pl.tiles_k1 = sdt_ceil_div(K1, edge); pl.tiles_n = sdt_ceil_div(N, edge);
    pl.groups = pl.tiles_k1 * pl.tiles_n * taps;
    base_wg = pl.groups;
    static const int tt = env_int("SDT_TN_TARGET_WG", 384);
    static const int tr = env_int("SDT_TN_MIN_ROWS", 1024);
    target = tt > 0 ? tt : 384; min_rows = tr >= BK ? tr : 1024;
    slab_bytes = pl.tm == 2 ? TnSlab<4>::BYTES : TnSlab<1>::BYTES;
  }
  if (target_override > 0) target = target_override;
  // Splits add workgroups, but every split writes its whole partial tile and the last arriver reads them all back: a problem that
  // already has three quarters of the target in tiles stays unsplit ([r4]: the one-problem launch of the (1024, 1920, 1280) x 9
  // gradient - 900 workgroups against a target of 1,024 - was cut in two and spent half its 116 us on 354 MB of slabs)
  int splits = base_wg * 4 >= (long)target * 3 ? 1 : (int)((target + base_wg - 1) / base_wg);
  const int max_splits = (int)((M + min_rows - 1) / min_rows);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits > 64) splits = 64;
  int rps = (int)((M + splits - 1) / splits);
  rps = ((rps + BK - 1) / BK) * BK;
  splits = (int)((M + rps - 1) / rps);  // no empty split
  pl.splits = splits;
  pl.rows_per_split = rps;
  // The arrival counters live in a FIXED area at the start of the workspace whatever the launch (launches of a stream share
  // the buffer: a counter area sized per launch would overlap the slab bytes of a launch with fewer tile groups)
  pl.cnt_bytes = SPLIT_CNT_BYTES;
  if ((int64_t)pl.groups * (int64_t)sizeof(int) > SPLIT_CNT_BYTES) {  // (never at the model's sizes) more groups than counters: no split
    pl.splits = 1;
    pl.rows_per_split = (int)(((M + BK - 1) / BK) * BK);
  }
  pl.ws_bytes = pl.splits > 1 ? pl.cnt_bytes + (int64_t)pl.groups * pl.splits * slab_bytes : 0;
  return pl;
}

// ---- 3x3 halo convolution: eligibility, tile shape, split plan
struct ConvHaloPlan {
  int ni, th, tw, tiles_x, tiles_y, tiles_m, tiles_n, splits, chunks_per_split;
};
// output channels per halo-convolution tile: 64 (two workgroups per CU; the default: 5-17 % less device time per launch,
// -1.05 ms per SD1.5 step same-box) or 128 (one per CU; SDT_HALO_BN=128, kept as the reference of the bitwise parity test)
static int conv_halo_bn() {  // (read per call: the parity test runs both in one process)
  return env_int("SDT_HALO_BN", 64) == 128 ? 128 : 64;
}
static int conv_halo_splits(long tiles, int chunks) {
  // one workgroup per CU (152 KB of LDS; two at BN = 64): as many channel-chunk splits as still fit the 256 CUs in ONE round (a 257th
  // workgroup would wait for a whole tile time); measured: 240 workgroups beat 160 by 10-14 %, 280 lose 20 %
  static const int cus1 = env_int("SDT_CONV_HALO_WG", 256);
  const int cus = cus1 * (conv_halo_bn() == 64 ? 2 : 1);  // (two 64-channel workgroups share a CU)
  if (chunks < 2 || tiles * 2 > cus) return 1;
  int s = (int)(cus / tiles);
  if (s > chunks) s = chunks;
  if (s > 16) s = 16;
  return s < 1 ? 1 : s;
}
static bool conv_halo_plan(const GatherDesc& g, int64_t M, int N, int Kc, int taps, int batch, ConvHaloPlan* pl) {
  static const int enabled = env_int("SDT_CONV_HALO", 1);
  // (round 2 kept the 8x8 levels, 256 pixels = one tile, on the generic split-K path; with the slab hand-off back to the write-through
  //  form, round 3, the halo kernel's split over channel chunks wins there too: -0.3 ms per step same-box)
  static const int min_px = env_int("SDT_CONV_HALO_MINPX", 256);
  if (!enabled || M < min_px) return false;
  if (!(g.mode == GATHER_FPROP || g.mode == GATHER_DGRAD) || taps != 9 || g.KH != 3 || g.KW != 3 || g.stride != 1 ||
      g.pad_t != 1 || g.pad_l != 1 || g.IH != g.OH || g.IW != g.OW || Kc % BK != 0)
    return false;
  const int H = g.OH, W = g.OW;
  if (W % 64 == 0 && H % 4 == 0) { pl->ni = 1; pl->th = 4; pl->tw = 64; }
  else if (W == 32 && H % 8 == 0) { pl->ni = 1; pl->th = 8; pl->tw = 32; }
  else if (W % 16 == 0 && H % 16 == 0) { pl->ni = 1; pl->th = 16; pl->tw = 16; }  // aspect buckets: 96, 48, 80 ... wide levels
  else if (W % 8 == 0 && H % 8 == 0) { pl->ni = 4; pl->th = 8; pl->tw = 8; }      // 72 x 56, 24 x 40, ...; 8 x 8: four images per tile
  else return false;
  pl->tiles_x = W / pl->tw;
  pl->tiles_y = H / pl->th;
  pl->tiles_m = sdt_ceil_div(batch, pl->ni) * pl->tiles_x * pl->tiles_y;
  pl->tiles_n = sdt_ceil_div(N, conv_halo_bn());
  const int chunks = Kc / BK;
  pl->splits = conv_halo_splits((long)pl->tiles_m * pl->tiles_n, chunks);
  pl->chunks_per_split = sdt_ceil_div(chunks, pl->splits);
  pl->splits = sdt_ceil_div(chunks, pl->chunks_per_split);
  return true;
}
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
template <bool SPLITK, bool BKM, int BN, bool MF16>
static void launch_conv_halo3(const GemmNtParams& p, int splits, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)conv3x3_halo_kernel<SPLITK, BKM, BN, MF16>, hipFuncAttributeMaxDynamicSharedMemorySize, CvCfg<BN>::LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_halo_kernel<SPLITK, BKM, BN, MF16>), dim3(p.tiles_m * p.tiles_n, splits), dim3(256), CvCfg<BN>::LDS_BYTES, stream, p);
}
// MFMA shape of the 64-channel tiling: v_mfma_f32_16x16x32_bf16 (SDT_HALO_MFMA=16) or v_mfma_f32_32x32x16_bf16 (32); a larger value
// is a threshold: the small shape for row-major weights (input gradients) with at least that many input channels
// (the measured-slower 16x16x32 instantiations are built only with -DSDT_HALO_MF16: tools/mfma_shape_probe.hip)
#ifdef SDT_HALO_MF16
static bool conv_halo_mf16(bool b_kmajor, int Kc) {
  const int m = env_int("SDT_HALO_MFMA", 32);
  if (m == 16) return true;
  if (m <= 32) return false;
  return !b_kmajor && Kc >= m;
}
#endif
template <bool SPLITK>
static void launch_conv_halo(const GemmNtParams& p, int splits, bool b_kmajor, hipStream_t stream) {
  if (conv_halo_bn() == 64) {
#ifdef SDT_HALO_MF16
    if (conv_halo_mf16(b_kmajor, p.Kc)) {
      if (b_kmajor) launch_conv_halo3<SPLITK, true, 64, true>(p, splits, stream); else launch_conv_halo3<SPLITK, false, 64, true>(p, splits, stream);
      return;
    }
#endif
    if (b_kmajor) launch_conv_halo3<SPLITK, true, 64, false>(p, splits, stream); else launch_conv_halo3<SPLITK, false, 64, false>(p, splits, stream);
  } else {
    if (b_kmajor) launch_conv_halo3<SPLITK, true, 128, false>(p, splits, stream); else launch_conv_halo3<SPLITK, false, 128, false>(p, splits, stream);
  }
}
static int conv_halo_slab_bytes() { return conv_halo_bn() == 64 ? TnSlab<4>::BYTES : TnSlab<8>::BYTES; }

// split-K workspace: one arrival counter per output tile (a whole number of KiB), then tiles x splits partial-sum slabs
static int64_t nt_ws_counter_bytes(int64_t) { return SPLIT_CNT_BYTES; }  // fixed counter area (see plan_tn)
static int64_t nt_workspace_need(int64_t tiles, int splits, int slab_bytes) {
  if (tiles * (int64_t)sizeof(int) > SPLIT_CNT_BYTES) return INT64_MAX / 2;  // more tiles than counters: never offered -> unsplit
  return nt_ws_counter_bytes(tiles) + tiles * splits * (int64_t)slab_bytes;
}
static int64_t nt_plan_need(int64_t M, int N, const NtPlan& pl) {
  if (pl.splits <= 1) return 0;
  const int edge = 64 * pl.tm;
  const int64_t tiles = (int64_t)sdt_ceil_div(M, edge) * sdt_ceil_div(N, edge);
  return nt_workspace_need(tiles, pl.splits, pl.tm == 2 ? TnSlab<4>::BYTES : TnSlab<1>::BYTES);
}

extern "C" {

int64_t sdt_gemm_nt_workspace_bytes(int64_t M, int N, int Kc, int taps) {
  if (M <= 0 || N <= 0 || Kc <= 0 || taps <= 0) return 0;
  const NtPlan pl = plan_nt(M, N, Kc, taps);
  int64_t need = nt_plan_need(M, N, pl);
  if (taps == 9 && Kc % BK == 0) {  // may run as a halo convolution (decided at launch from the geometry): cover that plan too.
    // Its tile count depends on the tile shape (four small images share a tile; a last group may be part empty): bound it
    const long tiles = (4L * sdt_ceil_div(M, CV_BM) + 4) * sdt_ceil_div(N, conv_halo_bn());
    const int hs = conv_halo_splits((long)sdt_ceil_div(M, CV_BM) * sdt_ceil_div(N, conv_halo_bn()), Kc / BK);
    if (hs > 1) need = std::max<int64_t>(need, nt_workspace_need(tiles, hs, conv_halo_slab_bytes()));
  }
  return need;
}

/* Number of partial statistics rows per image that sdt_gemm_nt_bf16 writes to gn_stats for this problem (include/sdt.h), or 0
 * when it cannot: an output tile would straddle two images, or a channel group is wider than a column tile. */
int sdt_gemm_nt_gn_parts(int64_t M, int N, int Kc, int taps, int rows_per_batch, int gn_groups, int gather_mode,
                         const SdtConvGeom* geom) {
  if (M <= 0 || N <= 0 || Kc <= 0 || taps <= 0 || rows_per_batch <= 0 || gn_groups <= 0 || gn_groups > 64 || N % gn_groups) return 0;
  const int cpg = N / gn_groups;
  if (cpg < 2) return 0;  // a 128-column tile must span at most 64 groups
  if (gather_mode != GATHER_PLAIN && geom) {
    GatherDesc g;
    if (fill_gather(&g, geom, gather_mode, "sdt_gemm_nt_gn_parts") == SDT_OK) {
      ConvHaloPlan hp;
      if (conv_halo_plan(g, M, N, Kc, taps, geom->batch, &hp)) {
        if (hp.ni != 1 || cpg > conv_halo_bn() || rows_per_batch != g.OH * g.OW) return 0;
        return 2 * hp.tiles_x * hp.tiles_y;
      }
    }
  }
  const NtPlan pl = plan_nt(M, N, Kc, taps);
  const int edge = 64 * pl.tm;
  if (rows_per_batch % edge != 0 || cpg > edge) return 0;
  return 2 * (rows_per_batch / edge);
}

int sdt_gemm_nt_bf16(const uint16_t* A, const uint16_t* Bt, uint16_t* C, const float* bias, const uint16_t* rowbias,
                     const uint16_t* residual, int64_t M, int N, int Kc, int taps, int lda, int ldb,
                     int64_t b_tap_stride, int ldc, int ldres, int rows_per_batch, int gather_mode,
                     const SdtConvGeom* geom, void* workspace, int64_t workspace_bytes, float* gn_stats, int gn_groups,
                     int b_kmajor, int b_nseg, int64_t b_seg_stride, int ld_rowbias, hipStream_t stream) {
  SDT_CHECK_ARG(A && Bt && C, "sdt_gemm_nt_bf16: null pointer");
  SDT_CHECK_ARG(!gn_stats || (gn_groups > 0 && gn_groups <= 64 && N % gn_groups == 0 && rows_per_batch > 0),
                "sdt_gemm_nt_bf16: gn_stats needs rows_per_batch and N divisible by gn_groups <= 64");
  const int gn_parts = gn_stats ? sdt_gemm_nt_gn_parts(M, N, Kc, taps, rows_per_batch, gn_groups, gather_mode, geom) : 0;
  SDT_CHECK_ARG(!gn_stats || gn_parts > 0, "sdt_gemm_nt_bf16: gn_stats not available for this shape (ask sdt_gemm_nt_gn_parts)");
  SDT_CHECK_ARG(M > 0 && M < (1L << 31) && N > 0 && Kc > 0 && taps > 0, "sdt_gemm_nt_bf16: bad dims M=%ld N=%d Kc=%d taps=%d", (long)M, N, Kc, taps);
  SDT_CHECK_ARG(N % 8 == 0 && Kc % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && b_tap_stride % 8 == 0,
                "sdt_gemm_nt_bf16: N, Kc and all leading dims must be multiples of 8 (N=%d Kc=%d lda=%d ldb=%d ldc=%d)", N, Kc, lda, ldb, ldc);
  SDT_CHECK_ARG((((uintptr_t)A | (uintptr_t)Bt | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)rowbias | (uintptr_t)residual |
                  (uintptr_t)workspace) & 15) == 0,
                "sdt_gemm_nt_bf16: pointers must be 16-byte aligned");
  SDT_CHECK_ARG(!rowbias || rows_per_batch > 0, "sdt_gemm_nt_bf16: rowbias needs rows_per_batch");
  SDT_CHECK_ARG(ld_rowbias == 0 || (rowbias && ld_rowbias >= N && ld_rowbias % 8 == 0), "sdt_gemm_nt_bf16: bad ld_rowbias %d", ld_rowbias);
  SDT_CHECK_ARG(!residual || (ldres % 8 == 0 && ldres >= N), "sdt_gemm_nt_bf16: bad ldres");
  SDT_CHECK_ARG(b_kmajor == 0 || b_kmajor == 1, "sdt_gemm_nt_bf16: b_kmajor must be 0 or 1");
  SDT_CHECK_ARG(b_nseg == 0 || (b_kmajor && b_nseg > 0 && b_nseg % 8 == 0 && N % b_nseg == 0 && b_seg_stride % 8 == 0 && ldb >= b_nseg),
                "sdt_gemm_nt_bf16: column segments need a k-major B, b_nseg | N, multiples of 8");
  SDT_CHECK_ARG(!b_kmajor || gather_mode != GATHER_DGRAD, "sdt_gemm_nt_bf16: a k-major B is for forward contractions");
  GemmNtParams p;
  p.b_nseg = b_nseg; p.b_seg_stride = b_seg_stride;
  int rc = fill_gather(&p.g, geom, gather_mode, "sdt_gemm_nt_bf16");
  if (rc) return rc;
  {  // kernels index A and Bt with 32-bit element offsets
    const int64_t a_elems = gather_mode == GATHER_PLAIN ? M * (int64_t)lda : (int64_t)geom->batch * p.g.IH * p.g.IW * lda;
    const int64_t b_elems = b_kmajor ? (int64_t)taps * (b_tap_stride > 0 ? b_tap_stride : (int64_t)Kc * ldb) + (int64_t)Kc * ldb + (b_nseg ? (N / b_nseg) * b_seg_stride : 0)
                                     : (int64_t)N * ldb;
    SDT_CHECK_ARG(a_elems < (1LL << 31) - (1 << 20) && b_elems < (1LL << 31), "sdt_gemm_nt_bf16: operand exceeds 2^31 elements");
  }
  if (gather_mode != GATHER_PLAIN) {
    SDT_CHECK_ARG(taps == p.g.KH * p.g.KW, "sdt_gemm_nt_bf16: taps=%d != kh*kw", taps);
    SDT_CHECK_ARG(M == (int64_t)geom->batch * p.g.OH * p.g.OW, "sdt_gemm_nt_bf16: M=%ld does not match conv geometry", (long)M);
  } else {  // plain: A is [M][taps*Kc]; column block t contracts with the B segment at Bt + t*b_tap_stride
    SDT_CHECK_ARG(lda >= taps * Kc, "sdt_gemm_nt_bf16: plain mode needs lda >= taps*Kc");
  }
  p.A = (const bf16_t*)A; p.Bt = (const bf16_t*)Bt; p.C = (bf16_t*)C; p.bias = bias;
  p.rowbias = (const bf16_t*)rowbias; p.residual = (const bf16_t*)residual;
  p.M = (int)M; p.N = N; p.Kc = Kc; p.taps = taps; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldres = ldres;
  p.ldrb = ld_rowbias ? ld_rowbias : N;
  p.b_tap_stride = b_tap_stride; p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
  p.gn_stats = gn_stats; p.gn_groups = gn_groups; p.gn_parts = gn_parts;
  ConvHaloPlan hp;
  // (the halo kernel takes a row's image as its row-bias row: only when the bias is per image, rows_per_batch = OH*OW)
  if (gather_mode != GATHER_PLAIN && (!rowbias || rows_per_batch == p.g.OH * p.g.OW) && conv_halo_plan(p.g, M, N, Kc, taps, geom->batch, &hp)) {
    p.cv_ni = hp.ni; p.cv_th = hp.th; p.cv_tw = hp.tw; p.cv_ltw = ilog2(hp.tw); p.cv_lth = ilog2(hp.th);
    p.cv_tiles_x = hp.tiles_x; p.cv_tiles_y = hp.tiles_y;
    p.cv_div_w2 = make_fastdiv((unsigned)(hp.tw + 2));
    p.cv_div_himg = make_fastdiv((unsigned)((hp.th + 2) * (hp.tw + 2)));
    p.cv_div_tx = make_fastdiv((unsigned)hp.tiles_x);
    p.cv_div_ty = make_fastdiv((unsigned)hp.tiles_y);
    p.tiles_m = hp.tiles_m; p.tiles_n = hp.tiles_n;
    p.nfast = nt_tile_order(2.0 * geom->batch * p.g.IH * p.g.IW * Kc, 2.0 * taps * (double)Kc * N, hp.tiles_m, hp.tiles_n);
    p.dbg = nt_dbg_bits();
    const int64_t htiles = (int64_t)hp.tiles_m * hp.tiles_n;
    if (hp.splits > 1 && workspace && workspace_bytes >= nt_workspace_need(htiles, hp.splits, conv_halo_slab_bytes())) {
      p.cv_chunks_per_split = hp.chunks_per_split;
      p.tile_cnt = reinterpret_cast<int*>(workspace);
      p.slab = (unsigned char*)workspace + nt_ws_counter_bytes(htiles);
      launch_conv_halo<true>(p, hp.splits, b_kmajor != 0, stream);
    } else {
      p.cv_chunks_per_split = Kc / BK;
      launch_conv_halo<false>(p, 1, b_kmajor != 0, stream);
    }
    SDT_LAUNCH_CHECK("sdt_gemm_nt_bf16");
    return SDT_OK;
  }
  NtPlan pl = plan_nt(M, N, Kc, taps);
  const int64_t need = nt_plan_need(M, N, pl);
  if (need > 0 && (!workspace || workspace_bytes < need)) {  // no workspace offered: run unsplit (slower, same result path)
    pl.splits = 1;
    pl.ksteps_per_split = taps * sdt_ceil_div(Kc, BK);
  }
  const int edge = 64 * pl.tm;
  p.tiles_m = sdt_ceil_div(M, edge); p.tiles_n = sdt_ceil_div(N, edge);
  p.nfast = nt_tile_order(gather_mode == GATHER_PLAIN ? 2.0 * M * taps * Kc : 2.0 * geom->batch * p.g.IH * p.g.IW * Kc, 2.0 * taps * (double)Kc * N,
                          p.tiles_m, p.tiles_n);
  p.ksteps_per_split = pl.ksteps_per_split;
  p.dbg = nt_dbg_bits();
  // 3x3 forward convolution of an 8-channel input (conv_in): eight taps per K-step (gemm_nt_kernel PACK8); the tile plan is the
  // unpacked shape's, so sdt_gemm_nt_gn_parts and the workspace query need not know
  if (pl.splits == 1 && gather_mode == GATHER_FPROP && b_kmajor && Kc == 8 && lda == 8 && taps == 9 && p.g.KH == 3 && p.g.KW == 3 &&
      b_nseg == 0 && b_tap_stride == (int64_t)8 * ldb) {
    p.taps = 1; p.Kc = 72; p.ksteps_per_split = 2;
    if (pl.tm == 2) launch_nt_pack8<2>(p, stream); else launch_nt_pack8<1>(p, stream);
    SDT_LAUNCH_CHECK("sdt_gemm_nt_bf16");
    return SDT_OK;
  }
  if (pl.splits > 1) {
    p.tile_cnt = reinterpret_cast<int*>(workspace);
    p.slab = (unsigned char*)workspace + nt_ws_counter_bytes((int64_t)p.tiles_m * p.tiles_n);
    if (pl.tm == 2) launch_nt<2, true>(p, pl.splits, b_kmajor != 0, stream); else launch_nt<1, true>(p, pl.splits, b_kmajor != 0, stream);
  } else if (pl.tm == 2 && taps * sdt_ceil_div(Kc, BK) <= nt_k32_max_steps() && !(gather_mode == GATHER_DGRAD && p.g.stride != 1)) {
    if (b_kmajor) launch_nt_k32<true>(p, stream); else launch_nt_k32<false>(p, stream);
  } else {
    if (pl.tm == 2) launch_nt<2, false>(p, 1, b_kmajor != 0, stream); else launch_nt<1, false>(p, 1, b_kmajor != 0, stream);
  }
  SDT_LAUNCH_CHECK("sdt_gemm_nt_bf16");
  return SDT_OK;
}

}  // extern "C"
// ---- transformer feed-forward with the GEGLU fused into the GEMM epilogues (GemmNtParams.geglu_f) -----------------------------
static bool ff_geglu_plan_ok(int64_t M, int N, int K) {
  const NtPlan pl = plan_nt(M, N, K, 1);
  return pl.tm == 2 && pl.splits == 1 && sdt_ceil_div(K, BK) <= nt_k32_max_steps();
}
static bool ff_geglu_ok(int64_t M, int F, int K) {
  if (M <= 0 || M >= (1L << 31) || F <= 0 || K <= 0 || F % 64 || K % 8 || (int64_t)M * 2 * F >= (1LL << 31)) return 0;
  return ff_geglu_plan_ok(M, 2 * F, K);
}
static void launch_ff_geglu(const GemmNtParams& p, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_kernel<2, false, false, true, false, 32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_MAX_LDS);
    attr_set = true;
  }
  GemmNtParams q = p;
  q.nst = nt_ring_depth<2, 32>((long)p.tiles_m * p.tiles_n, p.ksteps_per_split);
  hipLaunchKernelGGL((gemm_nt_kernel<2, false, false, true, false, 32, 1>), dim3(p.tiles_m * p.tiles_n, 1), dim3(256), (q.nst * 2 * NtCfg<2, 32>::TILE_BYTES), stream, q);
}
static void ff_geglu_fill(GemmNtParams* p, int64_t M, int N, int K) {
  memset(p, 0, sizeof(*p));
  p->M = (int)M; p->N = N; p->Kc = K; p->taps = 1;
  p->rows_per_batch = 1;
  p->g.mode = GATHER_PLAIN; p->g.KH = p->g.KW = 1; p->g.stride = 1;
  p->tiles_m = sdt_ceil_div(M, 128); p->tiles_n = sdt_ceil_div(N, 128);
  p->nfast = nt_tile_order(2.0 * M * K, 2.0 * (double)K * N, p->tiles_m, p->tiles_n);
  p->ksteps_per_split = sdt_ceil_div(K, BK);
}
extern "C" {
int sdt_ff_geglu_supported(int64_t M, int F, int K) { return ff_geglu_ok(M, F, K) ? 1 : 0; }
int sdt_ff_geglu_fwd(const uint16_t* x, const uint16_t* W1, const float* bias, uint16_t* h, uint16_t* out, int64_t M, int F, int K,
                     hipStream_t stream) {
  SDT_CHECK_ARG(x && W1 && h && out, "sdt_ff_geglu_fwd: null pointer");
  SDT_CHECK_ARG(sdt_ff_geglu_supported(M, F, K), "sdt_ff_geglu_fwd: shape M=%ld F=%d K=%d not served by the fused kernel (ask sdt_ff_geglu_supported)", (long)M, F, K);
  SDT_CHECK_ARG((((uintptr_t)x | (uintptr_t)W1 | (uintptr_t)bias | (uintptr_t)h | (uintptr_t)out) & 15) == 0, "sdt_ff_geglu_fwd: pointers must be 16-byte aligned");
  GemmNtParams p;
  ff_geglu_fill(&p, M, 2 * F, K);
  p.A = (const bf16_t*)x; p.Bt = (const bf16_t*)W1; p.C = (bf16_t*)h; p.C2 = (bf16_t*)out; p.bias = bias;
  p.lda = K; p.ldb = 2 * F; p.ldc = 2 * F; p.geglu_f = F;
  launch_ff_geglu(p, stream);
  SDT_LAUNCH_CHECK("sdt_ff_geglu_fwd");
  return SDT_OK;
}
int64_t sdt_wgrad_sq_slots(int K1, int N, int taps) {
  if (K1 <= 0 || N <= 0 || taps <= 0) return 0;
  return (int64_t)taps * (4 * sdt_ceil_div(K1, 128)) * (4 * sdt_ceil_div(N, 128));
}
int64_t sdt_gemm_tn_workspace_bytes(int64_t M, int K1, int N, int taps, int n_seg, int gather_mode, const SdtConvGeom* geom) {
  if (M <= 0 || K1 <= 0 || N <= 0 || taps <= 0) return 0;
  GatherDesc g;
  if (fill_gather(&g, geom, gather_mode, "sdt_gemm_tn_workspace_bytes") != SDT_OK) return 0;
  return plan_tn(g, gather_mode, M, K1, N, taps, n_seg).ws_bytes;
}

int sdt_gemm_tn_wgrad(const uint16_t* A, const uint16_t* dY, void* dW, int dw_bf16, float* dbias, int64_t M, int K1, int N, int K1_valid,
                      int N_valid, int taps, int lda, int ldb, int ldw, int64_t w_tap_stride, int n_seg, int64_t seg_stride,
                      int gather_mode, const SdtConvGeom* geom, void* workspace, int64_t workspace_bytes, double* sq_slots,
                      hipStream_t stream) {
  SDT_CHECK_ARG(A && dY && dW, "sdt_gemm_tn_wgrad: null pointer");
  SDT_CHECK_ARG(M > 0 && M < (1L << 31) && K1 > 0 && N > 0 && taps > 0 && taps < 65536, "sdt_gemm_tn_wgrad: bad dims");
  SDT_CHECK_ARG(K1 % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "sdt_gemm_tn_wgrad: K1, N, lda, ldb must be multiples of 8");
  SDT_CHECK_ARG(K1_valid > 0 && K1_valid <= K1 && N_valid > 0 && N_valid <= N && ldw >= (n_seg > 0 ? n_seg : N_valid), "sdt_gemm_tn_wgrad: bad valid dims");
  SDT_CHECK_ARG(n_seg >= 0 && (n_seg == 0 || N_valid % n_seg == 0), "sdt_gemm_tn_wgrad: N_valid must be a whole number of segments");
  SDT_CHECK_ARG((((uintptr_t)A | (uintptr_t)dY | (uintptr_t)workspace) & 15) == 0, "sdt_gemm_tn_wgrad: pointers must be 16-byte aligned");
  GemmTnParams p;
  int rc = fill_gather(&p.g, geom, gather_mode, "sdt_gemm_tn_wgrad");
  if (rc) return rc;
  {
    const int64_t a_elems = gather_mode == GATHER_PLAIN ? M * (int64_t)lda : (int64_t)geom->batch * p.g.IH * p.g.IW * lda;
    SDT_CHECK_ARG(a_elems < (1LL << 31) - (1 << 20) && M * (int64_t)ldb < (1LL << 31), "sdt_gemm_tn_wgrad: operand exceeds 2^31 elements");
  }
  if (gather_mode != GATHER_PLAIN) {
    SDT_CHECK_ARG(gather_mode == GATHER_FPROP, "sdt_gemm_tn_wgrad: gather must be plain or fprop");
    SDT_CHECK_ARG(taps == p.g.KH * p.g.KW, "sdt_gemm_tn_wgrad: taps=%d != kh*kw", taps);
    SDT_CHECK_ARG(M == (int64_t)geom->batch * p.g.OH * p.g.OW, "sdt_gemm_tn_wgrad: M does not match conv geometry");
  } else {
    SDT_CHECK_ARG(taps == 1, "sdt_gemm_tn_wgrad: plain mode needs taps == 1");
  }
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)dY; p.dW = (float*)dW; p.out_bf16 = dw_bf16 ? 1 : 0; p.dbias = dbias;
  p.M = (int)M; p.K1 = K1; p.N = N;
  tn_set_sq(&p, sq_slots); p.K1_valid = K1_valid; p.N_valid = N_valid;
  p.lda = lda; p.ldb = ldb; p.ldw = ldw; p.w_tap_stride = w_tap_stride; p.n_seg = n_seg; p.seg_stride = seg_stride;
  TnPlan pl = plan_tn(p.g, gather_mode, M, K1, N, taps, n_seg);
  if (pl.splits > 1 && (!workspace || workspace_bytes < pl.ws_bytes)) {  // no scratch offered: one workgroup reduces all of M
    pl.splits = 1;
    pl.rows_per_split = (int)(((M + BK - 1) / BK) * BK);
  }
  p.tiles_k1 = pl.tiles_k1; p.tiles_n = pl.tiles_n; p.rows_per_split = pl.rows_per_split;
  p.tile_cnt = reinterpret_cast<int*>(workspace);
  p.slab = pl.splits > 1 ? (unsigned char*)workspace + pl.cnt_bytes : nullptr;
  if (pl.w3) {
    static bool attr_set = false;
    if (!attr_set) {
      hipFuncSetAttribute((const void*)conv_wgrad3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS_BYTES);
      attr_set = true;
    }
    p.taps = 9; p.splits = pl.splits;
    hipLaunchKernelGGL(conv_wgrad3_kernel, dim3(p.tiles_k1 * p.tiles_n * 3 * pl.splits), dim3(256), W3_LDS_BYTES, stream, p);
  } else if (pl.tm == 2) {
    launch_tn<2>(p, taps, pl.splits, stream);
  } else {
    launch_tn<1>(p, taps, pl.splits, stream);
  }
  SDT_LAUNCH_CHECK("sdt_gemm_tn_wgrad");
  return SDT_OK;
}

}  // extern "C"

// ---- grouped Dense weight gradients (include/sdt.h sdt_gemm_tn_wgrad_group) ----
static_assert(sizeof(GemmTnGroupParams) <= 4096, "the grouped weight-gradient launch passes its problem table as kernel arguments");
struct TnGroupItem {
  GemmTnParams p;
  TnPlan pl;
};
// workgroups the problems of one grouped launch aim at together: a few rounds of the chip's 512 resident workgroups, shared by
// the problems in proportion (each problem's reduction is split less than it would be alone: less slab traffic per result)
static int tn_group_target(int n) {
  static const int total = env_int("SDT_TN_GROUP_WG", 1024);
  int t = total / (n > 0 ? n : 1);
  return t < 48 ? 48 : t;
}
static int64_t tn_slab_bytes(const TnPlan& pl) { return pl.tm == 2 ? TnSlab<4>::BYTES : TnSlab<1>::BYTES; }
static int tn_group_fill(const SdtTnProblem* q, int n, TnGroupItem* items, const char* name) {
  for (int i = 0; i < n; ++i) {
    const SdtTnProblem& a = q[i];
    SDT_CHECK_ARG(a.A && a.dY && a.dW, "%s: problem %d: null pointer", name, i);
    SDT_CHECK_ARG(a.M > 0 && a.M < (1L << 31) && a.K1 > 0 && a.N > 0, "%s: problem %d: bad dims", name, i);
    SDT_CHECK_ARG(a.K1 % 8 == 0 && a.N % 8 == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0, "%s: problem %d: K1, N, lda, ldb must be multiples of 8", name, i);
    SDT_CHECK_ARG(a.K1_valid > 0 && a.K1_valid <= a.K1 && a.N_valid > 0 && a.N_valid <= a.N && a.ldw >= (a.n_seg > 0 ? a.n_seg : a.N_valid),
                  "%s: problem %d: bad valid dims", name, i);
    SDT_CHECK_ARG(a.n_seg >= 0 && (a.n_seg == 0 || a.N_valid % a.n_seg == 0), "%s: problem %d: N_valid must be a whole number of segments", name, i);
    SDT_CHECK_ARG((((uintptr_t)a.A | (uintptr_t)a.dY) & 15) == 0, "%s: problem %d: pointers must be 16-byte aligned", name, i);
    SDT_CHECK_ARG(a.M * (int64_t)a.lda < (1LL << 31) - (1 << 20) && a.M * (int64_t)a.ldb < (1LL << 31), "%s: problem %d: operand exceeds 2^31 elements", name, i);
    GemmTnParams& p = items[i].p;
    fill_gather(&p.g, nullptr, GATHER_PLAIN, name);
    p.A = (const bf16_t*)a.A; p.B = (const bf16_t*)a.dY; p.dW = (float*)a.dW; p.out_bf16 = a.dw_bf16 ? 1 : 0; p.dbias = a.dbias;
    p.M = (int)a.M; p.K1 = a.K1; p.N = a.N; p.K1_valid = a.K1_valid; p.N_valid = a.N_valid;
    p.lda = a.lda; p.ldb = a.ldb; p.ldw = a.ldw; p.w_tap_stride = (long)a.K1_valid * a.N_valid; p.n_seg = a.n_seg; p.seg_stride = a.seg_stride;
    items[i].pl = plan_tn(p.g, GATHER_PLAIN, a.M, a.K1, a.N, 1, a.n_seg, tn_group_target(n));
    p.tiles_k1 = items[i].pl.tiles_k1; p.tiles_n = items[i].pl.tiles_n; p.rows_per_split = items[i].pl.rows_per_split;
    tn_set_sq(&p, a.sq_slots);
  }
  return SDT_OK;
}
#define TN_GROUP_ABI_MAX 64
// Cut the workgroups of a grouped launch (wg_end / prob filled, order (problem, split, [kernel row,] tile)) into eight contiguous
// runs of equal work, one per XCD (GemmTnGroupParams).  Work of a workgroup = its K-steps + a fixed cost (prologue latency, slab
// publish, epilogue) expressed in K-steps.  Returns the grid size: 8 x the longest run.
static int tn_group_cut(GemmTnGroupParams* gp) {
  static const int fixed = env_int("SDT_TN_WG_FIXED_STEPS", 6);
  double w[TN_GROUP_MAX], total = 0.0;
  for (int i = 0; i < gp->n; ++i) {
    const int cnt = gp->wg_end[i] - (i ? gp->wg_end[i - 1] : 0);
    w[i] = (double)sdt_ceil_div(std::min(gp->prob[i].rows_per_split, gp->prob[i].M), BK) + fixed;
    total += w[i] * cnt;
  }
  const int wgs = gp->wg_end[gp->n - 1];
  gp->xcd_begin[0] = 0;
  gp->xcd_begin[8] = wgs;
  int i = 0;
  double before = 0.0;  // work of the problems in front of problem i
  for (int x = 1; x < 8; ++x) {
    const double want = total * x / 8.0;
    while (i + 1 < gp->n && before + w[i] * (gp->wg_end[i] - (i ? gp->wg_end[i - 1] : 0)) < want) {
      before += w[i] * (gp->wg_end[i] - (i ? gp->wg_end[i - 1] : 0));
      ++i;
    }
    const int first = i ? gp->wg_end[i - 1] : 0;
    int L = first + (int)((want - before) / w[i] + 0.5);
    L = std::max(L, gp->xcd_begin[x - 1]);
    gp->xcd_begin[x] = std::min(L, wgs);
  }
  int longest = 0;
  for (int x = 0; x < 8; ++x) longest = std::max(longest, gp->xcd_begin[x + 1] - gp->xcd_begin[x]);
  return 8 * longest;
}

extern "C" {

int sdt_gemm_tn_wgrad_group_max(void) { return TN_GROUP_ABI_MAX; }

int64_t sdt_gemm_tn_wgrad_group_workspace_bytes(const SdtTnProblem* problems, int n) {
  if (!problems || n <= 0 || n > TN_GROUP_ABI_MAX) return 0;
  static thread_local TnGroupItem items[TN_GROUP_ABI_MAX];
  if (tn_group_fill(problems, n, items, "sdt_gemm_tn_wgrad_group_workspace_bytes") != SDT_OK) return 0;
  int64_t need = SPLIT_CNT_BYTES;
  for (int i = 0; i < n; ++i)
    if (items[i].pl.splits > 1) need += (int64_t)items[i].pl.groups * items[i].pl.splits * tn_slab_bytes(items[i].pl);
  return need;
}

int sdt_gemm_tn_wgrad_group(const SdtTnProblem* problems, int n, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  SDT_CHECK_ARG(problems && n > 0 && n <= TN_GROUP_ABI_MAX, "sdt_gemm_tn_wgrad_group: 1..%d problems", TN_GROUP_ABI_MAX);
  SDT_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "sdt_gemm_tn_wgrad_group: workspace must be 16-byte aligned");
  static thread_local TnGroupItem items[TN_GROUP_ABI_MAX];
  int rc = tn_group_fill(problems, n, items, "sdt_gemm_tn_wgrad_group");
  if (rc) return rc;
  // scratch: counters of all problems side by side in the fixed 64 KiB area, then their slabs; a problem whose slabs do not fit
  // (or with no workspace at all) runs unsplit - slower, same path
  int64_t cnt_used = 0, slab_used = SPLIT_CNT_BYTES;
  for (int i = 0; i < n; ++i) {
    TnGroupItem& it = items[i];
    const int64_t sb = tn_slab_bytes(it.pl);
    const int64_t want = (int64_t)it.pl.groups * it.pl.splits * sb;
    if (it.pl.splits > 1 && (!workspace || slab_used + want > workspace_bytes || (cnt_used + it.pl.groups) * (int64_t)sizeof(int) > SPLIT_CNT_BYTES)) {
      it.pl.splits = 1;
      it.pl.rows_per_split = (int)(((it.p.M + BK - 1) / BK) * BK);
      it.p.rows_per_split = it.pl.rows_per_split;
    }
    it.p.tile_cnt = workspace ? reinterpret_cast<int*>(workspace) + cnt_used : nullptr;
    it.p.slab = it.pl.splits > 1 ? (unsigned char*)workspace + slab_used : nullptr;
    if (it.pl.splits > 1) { cnt_used += it.pl.groups; slab_used += want; }
  }
  static bool attr1 = false, attr2 = false;
  for (int tm = 1; tm <= 2; ++tm) {
    GemmTnGroupParams gp;
    gp.n = 0;
    int wg = 0;
    auto flush = [&]() {
      if (gp.n == 0) return;
      if (tm == 1) {
        if (!attr1) { hipFuncSetAttribute((const void*)gemm_tn_group_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, TnCfg<1>::LDS_BYTES); attr1 = true; }
        hipLaunchKernelGGL((gemm_tn_group_kernel<1>), dim3(tn_group_cut(&gp)), dim3(256), TnCfg<1>::LDS_BYTES, stream, gp);
      } else {
        if (!attr2) { hipFuncSetAttribute((const void*)gemm_tn_group_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, TnCfg<2>::LDS_BYTES); attr2 = true; }
        hipLaunchKernelGGL((gemm_tn_group_kernel<2>), dim3(tn_group_cut(&gp)), dim3(256), TnCfg<2>::LDS_BYTES, stream, gp);
      }
      gp.n = 0;
      wg = 0;
    };
    for (int i = 0; i < n; ++i) {
      if (items[i].pl.tm != tm) continue;
      wg += items[i].pl.groups * items[i].pl.splits;
      gp.prob[gp.n] = items[i].p;
      gp.prob[gp.n].splits = items[i].pl.splits;
      gp.prob[gp.n].taps = 1;
      gp.wg_end[gp.n] = wg;
      if (++gp.n == TN_GROUP_MAX) flush();
    }
    flush();
  }
  SDT_LAUNCH_CHECK("sdt_gemm_tn_wgrad_group");
  return SDT_OK;
}

/* scratch for sdt_conv_wgrad_group (split-workspace contract) */
int64_t sdt_conv_wgrad_group_workspace_bytes(const SdtConvWgradProblem* q, int n) {
  if (!q || n <= 0 || n > TN_GROUP_ABI_MAX) return 0;
  int64_t grouped = SPLIT_CNT_BYTES, single = 0;
  for (int i = 0; i < n; ++i) {
    GatherDesc g;
    if (fill_gather(&g, &q[i].geom, GATHER_FPROP, "sdt_conv_wgrad_group_workspace_bytes") != SDT_OK) return 0;
    const int64_t M = (int64_t)q[i].geom.batch * g.OH * g.OW;
    const int taps = g.KH * g.KW;
    const TnPlan pg = plan_tn(g, GATHER_FPROP, M, q[i].K1, q[i].N, taps, 0, tn_group_target(n));
    if (pg.w3) {
      if (pg.splits > 1) grouped += (int64_t)pg.groups * pg.splits * TnSlab<6>::BYTES;
    } else {
      single = std::max<int64_t>(single, plan_tn(g, GATHER_FPROP, M, q[i].K1, q[i].N, taps, 0).ws_bytes);
    }
  }
  return std::max(grouped, single);
}

/* The weight gradients of n convolutions (the arguments of sdt_gemm_tn_wgrad in fprop-gather mode) issued together: those the
 * three-taps-per-workgroup kernel serves (3x3, stride 1, pad 1, width a multiple of 8) share grouped launches of up to 16 problems,
 * the others are launched one by one behind them. */
int sdt_conv_wgrad_group(const SdtConvWgradProblem* q, int n, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  SDT_CHECK_ARG(q && n > 0 && n <= TN_GROUP_ABI_MAX, "sdt_conv_wgrad_group: 1..%d problems", TN_GROUP_ABI_MAX);
  SDT_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "sdt_conv_wgrad_group: workspace must be 16-byte aligned");
  static thread_local TnGroupItem items[TN_GROUP_ABI_MAX];
  static thread_local bool grouped[TN_GROUP_ABI_MAX];
  int64_t cnt_used = 0, slab_used = SPLIT_CNT_BYTES;
  for (int i = 0; i < n; ++i) {
    const SdtConvWgradProblem& a = q[i];
    SDT_CHECK_ARG(a.A && a.dY && a.dW, "sdt_conv_wgrad_group: problem %d: null pointer", i);
    SDT_CHECK_ARG(a.K1 > 0 && a.N > 0 && a.K1 % 8 == 0 && a.N % 8 == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 && a.K1_valid > 0 && a.K1_valid <= a.K1 &&
                      a.N_valid > 0 && a.N_valid <= a.N && (((uintptr_t)a.A | (uintptr_t)a.dY) & 15) == 0,
                  "sdt_conv_wgrad_group: problem %d: bad dims / alignment", i);
    GemmTnParams& p = items[i].p;
    int rc = fill_gather(&p.g, &a.geom, GATHER_FPROP, "sdt_conv_wgrad_group");
    if (rc) return rc;
    const int64_t M = (int64_t)a.geom.batch * p.g.OH * p.g.OW;
    const int taps = p.g.KH * p.g.KW;
    SDT_CHECK_ARG((int64_t)a.geom.batch * p.g.IH * p.g.IW * a.lda < (1LL << 31) - (1 << 20) && M * (int64_t)a.ldb < (1LL << 31),
                  "sdt_conv_wgrad_group: problem %d: operand exceeds 2^31 elements", i);
    TnPlan pl = plan_tn(p.g, GATHER_FPROP, M, a.K1, a.N, taps, 0, tn_group_target(n));
    grouped[i] = pl.w3;
    if (!pl.w3) continue;
    p.A = (const bf16_t*)a.A; p.B = (const bf16_t*)a.dY; p.dW = (float*)a.dW; p.out_bf16 = a.dw_bf16 ? 1 : 0; p.dbias = a.dbias;
    p.M = (int)M; p.K1 = a.K1; p.N = a.N; p.K1_valid = a.K1_valid; p.N_valid = a.N_valid;
    p.lda = a.lda; p.ldb = a.ldb; p.ldw = a.N_valid; p.w_tap_stride = (long)a.K1_valid * a.N_valid; p.n_seg = 0; p.seg_stride = 0;
    tn_set_sq(&p, a.sq_slots);
    const int64_t want = (int64_t)pl.groups * pl.splits * TnSlab<6>::BYTES;
    if (pl.splits > 1 && (!workspace || slab_used + want > workspace_bytes || (cnt_used + pl.groups) * (int64_t)sizeof(int) > SPLIT_CNT_BYTES)) {
      pl.splits = 1;
      pl.rows_per_split = (int)(((M + BK - 1) / BK) * BK);
    }
    p.tiles_k1 = pl.tiles_k1; p.tiles_n = pl.tiles_n; p.rows_per_split = pl.rows_per_split;
    p.tile_cnt = workspace ? reinterpret_cast<int*>(workspace) + cnt_used : nullptr;
    p.slab = pl.splits > 1 ? (unsigned char*)workspace + slab_used : nullptr;
    if (pl.splits > 1) { cnt_used += pl.groups; slab_used += want; }
    items[i].pl = pl;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)conv_wgrad3_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS_BYTES);
    attr_set = true;
  }
  GemmTnGroupParams gp;
  gp.n = 0;
  int wg = 0;
  for (int i = 0; i <= n; ++i) {
    if (i < n && grouped[i]) {
      wg += items[i].pl.groups * items[i].pl.splits;  // groups = tiles x 3 kernel rows
      gp.prob[gp.n] = items[i].p;
      gp.prob[gp.n].splits = items[i].pl.splits;
      gp.prob[gp.n].taps = 9;
      gp.wg_end[gp.n] = wg;
      ++gp.n;
    }
    if (gp.n == TN_GROUP_MAX || (i == n && gp.n > 0)) {
      gp.xcd_begin[8] = wg;  // (dealt round-robin: only the total is used)
      hipLaunchKernelGGL(conv_wgrad3_group_kernel, dim3(wg), dim3(256), W3_LDS_BYTES, stream, gp);
      gp.n = 0;
      wg = 0;
    }
  }
  SDT_LAUNCH_CHECK("sdt_conv_wgrad_group");
  for (int i = 0; i < n; ++i) {  // the rest (strided / odd widths / small images): their own launches, behind the grouped ones
    if (grouped[i]) continue;
    const SdtConvWgradProblem& a = q[i];
    const int64_t M = (int64_t)a.geom.batch * a.geom.out_h * a.geom.out_w;
    int rc = sdt_gemm_tn_wgrad(a.A, a.dY, a.dW, a.dw_bf16, a.dbias, M, a.K1, a.N, a.K1_valid, a.N_valid, a.geom.kh * a.geom.kw, a.lda, a.ldb, a.N_valid,
                               (int64_t)a.K1_valid * a.N_valid, 0, 0, GATHER_FPROP, &a.geom, workspace, workspace_bytes, a.sq_slots, stream);
    if (rc) return rc;
  }
  return SDT_OK;
}

}  // extern "C"
