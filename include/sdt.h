/* libsdtrain_hip.so - C ABI of the MI355X-native Stable Diffusion train_step hot path.
 *
 * The reference (lodestone-rock/stable_diffusion_training) has no FFI boundary of its own: its hot path is one
 * jitted Python function, training_utils.py:504-762 (train_step), whose arithmetic XLA lowers for the TPU.  These
 * entry points are what a binding for that path calls instead of XLA; each one cites the reference call site (or
 * the third-party module reached from it) whose computation it replaces.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated; the caller owns all memory;
 *     kernels never allocate (scratch is passed in); every call is asynchronous on `stream` and re-entrant.
 *   - return 0 (SDT_OK) or a negative code; sdt_last_error() returns a thread-local message.
 *   - bf16 tensors are uint16_t bit patterns; activations are NHWC / row-major, channel counts multiples of 8;
 *     16-byte aligned base pointers.
 */
#ifndef SDT_H_
#define SDT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t; /* same opaque handle type as <hip/hip_runtime_api.h> */
#endif

enum { SDT_OK = 0, SDT_ERR_INVALID_ARG = -1, SDT_ERR_UNSUPPORTED = -2, SDT_ERR_LAUNCH = -3 };

const char* sdt_last_error(void);
int sdt_abi_version(void);
/* number of HIP devices visible (0 when none / no driver): lets the host fail loudly instead of falling back */
int sdt_device_count(void);

/* ---- hand-off events between a captured step and the gradient exchange (no reference counterpart: under GSPMD the
 * all-reduce lives inside the jitted step, training_utils.py:35-37, 709).  A step captured into a HIP graph marks "this
 * gradient bucket is complete" with sdt_event_record(ev, external = 1, capturing_stream), which becomes an event-record
 * NODE of the graph (a plain record when the stream is not capturing); after every launch of that graph the host calls sdt_stream_wait_event(comm_stream, ev) in front of
 * the bucket's RCCL all-reduce, which stays outside the graph.  external = 0 is an ordinary hipEventRecord.
 * sdt_stream_wait_event_external is the reverse hand-off: on a capturing stream it adds an event-wait NODE, so that every launch
 * of the graph waits THERE for the event's most recent record (the sharded optimizer's all-gather of the weight mirrors, enqueued
 * on the communication stream between two launches and needed only when the next step reaches its text encoder / UNet: it runs
 * beside the next step's VAE encode); on a stream that is not capturing it is sdt_stream_wait_event. */
int sdt_event_create(void** event);
int sdt_event_destroy(void* event);
int sdt_event_record(void* event, int external, hipStream_t stream);
int sdt_stream_wait_event(hipStream_t stream, void* event);
int sdt_stream_wait_event_external(hipStream_t stream, void* event);

/* ---- geometry descriptors (host memory) ---- */
typedef struct SdtConvGeom {
  int batch, in_h, in_w, out_h, out_w, kh, kw, stride, pad_top, pad_left;
} SdtConvGeom;

typedef struct SdtAttnDesc {
  int B, H, Nq, Nk, D;       /* D = head dim; q/k/v/o are (B, N, >=H*D) with a head = columns [h*D, h*D+D) */
  int ldq, ldk, ldv, ldo;    /* row strides in elements */
  float scale;               /* logits scale (1/sqrt(D)) */
  int causal;
  int ldgrad_q, ldgrad_k, ldgrad_v, ld_dout; /* backward only; 0 = same as ldq/ldk/ldv/ldo */
  /* optional device array w[Nk] > 0 (NULL = all ones): P = softmax(scale*q.k + ln w).  diffusers' memory-efficient attention walks
   * the keys in chunks of min(Nq, Nk) (key_chunk_patch.patch sets the chunk to the query count) and takes the last chunk with a
   * clamped jax.lax.dynamic_slice, so when Nk is not a multiple of the chunk the overlapped keys are summed twice: w = 2 there. */
  const float* key_weight;
} SdtAttnDesc;

enum { SDT_GATHER_PLAIN = 0, SDT_GATHER_CONV_FPROP = 1, SDT_GATHER_CONV_DGRAD = 2 };
enum { SDT_ACT_SILU = 0, SDT_ACT_QUICK_GELU = 1, SDT_ACT_GELU_ERF = 2 };

/* ================= scheduler / loss (schedulers/scheduling_utils_flax.py:316-343; training_utils.py:582-586, 704-709) */
/* noisy = sqrt(acp[t])*x0 + sqrt(1-acp[t])*eps ; velocity = sqrt(acp[t])*eps - sqrt(1-acp[t])*x0.
 * latents/noise/noisy_nchw/velocity_nchw: f32 (B,C,H,W); noisy_nhwc_bf16: (B,H,W,cpad) zero padded. */
int sdt_add_noise_velocity(const float* latents, const float* noise, const int32_t* timesteps,
                           const float* alphas_cumprod, uint16_t* noisy_nhwc_bf16, float* noisy_nchw,
                           float* velocity_nchw, int B, int C, int H, int W, int cpad, hipStream_t stream);
/* One sampling step (models/pipeline_flax_stable_diffusion.py:222-232 + diffusers scheduling_ddim_flax.py step(), eta = 0):
 * m = pred[0:B] + guidance_scale * (pred[B:2B] - pred[0:B]); x0 / eps from m by prediction_type (0 epsilon, 1 sample,
 * 2 v_prediction) with alpha_prod_t; latents <- sqrt(alpha_prod_prev)*x0 + sqrt(1-alpha_prod_prev)*eps (in place, f32 NCHW);
 * next_input_nhwc (2B,H,W,cpad) bf16 = the new latents twice (the doubled UNet batch of the next step), padding zero. */
int sdt_ddim_cfg_step(const uint16_t* pred_nhwc, float* latents_nchw, uint16_t* next_input_nhwc, int B, int C, int H, int W,
                      int cpad, float guidance_scale, float alpha_prod_t, float alpha_prod_prev, int prediction_type,
                      hipStream_t stream);
/* latents (B,L,H,W) f32 = (mean + exp(0.5*clip(logvar,-30,20))*eps)*scale from moments bf16 (B,H,W,moment_stride) */
int sdt_vae_posterior_sample(const uint16_t* moments_nhwc, const float* eps_nhwc, float* latents_nchw, int B, int L,
                             int H, int W, int moment_stride, float scale, hipStream_t stream);
/* loss_accum += mean(w_b*(target-pred)^2); dpred = d loss / d pred (bf16 NHWC, cpad channels).
 * REDUCTION WORKSPACES (this call, sdt_sqnorm_accumulate, sdt_colsum_*): sums that cross workgroups use no float atomics - each
 * workgroup stores a partial, the one that arrives last adds them in a fixed order, so results are bitwise reproducible.  They
 * follow the split-workspace contract of sdt_gemm_nt_bf16: the first 64 KiB are arrival counters that must be ZERO when the call
 * is enqueued and are zero again when it completes, the rest is scratch; one buffer (zeroed once) serves every call of a stream. */
int sdt_mse_loss_fwd_bwd(const uint16_t* pred_nhwc, const float* target_nchw, const float* weight, float* loss_accum,
                         uint16_t* dpred_nhwc, int B, int C, int H, int W, int cpad, void* workspace, int64_t workspace_bytes,
                         hipStream_t stream);
int64_t sdt_reduce_workspace_bytes(void);
/* diffusers embeddings_flax.get_sinusoidal_embeddings -> bf16 (B, dim) */
int sdt_timestep_embedding(const int32_t* timesteps, uint16_t* out, int B, int dim, int flip_sin_to_cos,
                           float freq_shift, hipStream_t stream);

/* ================= optimizer (lion_quant.py:20-211; training_utils.py:355-387, 537-544, 732-746; optax clip/lion) */
/* *out_sq += sum g^2 in double (optax.global_norm; the float32 norm the sweeps derive from it is the rounding of the true norm) */
int sdt_sqnorm_accumulate(const float* g, int64_t n, double* out_sq, void* workspace, int64_t workspace_bytes, hipStream_t stream);
/* the same over a bf16 gradient buffer (the kernel leaves' gradients, ABI 5): squares of the stored values, exact in double */
int sdt_sqnorm_accumulate_bf16(const uint16_t* g, int64_t n, double* out_sq, void* workspace, int64_t workspace_bytes, hipStream_t stream);
int64_t sdt_sqnorm_workspace_bytes(void);
/* fused clip(by *sqnorm, may be NULL) + 8-bit blockwise Lion + decay + update (+EMA) (+bf16 mirror of the new parameters,
 * w_bf16[i] = bf16(p[i]), the compute copy the next forward reads; NULL to skip); in place.
 * thresholds: device float[128], the decision thresholds of _quantize (lion_quant.py:52-59): thresholds[c] = the smallest
 * float32 a >= 0 with rint(a^(1/5) * 127) >= c (thresholds[0] = 0).  The caller builds them once with float32 host arithmetic
 * (stable_diffusion_training_amd/lion_codec.py), which makes the device codes bit-identical to the host definition. */
/* g: the gradient, float32 (g_bf16 = 0) or bf16 (g_bf16 = 1, ABI 5: the kernel leaves' gradients are stored with the precision the
 * reference's kernel cotangents have - flax Dense / Conv with dtype=bfloat16 - and widened here, where optax widens them). */
int sdt_lion8_step(float* p, const void* g, int g_bf16, int8_t* codes, float* inv_scale, float* ema, uint16_t* w_bf16, int64_t n,
                   int block_size, const double* sqnorm, const float* thresholds, double max_norm, double lr, double wd,
                   double b1, double b2, double ema_rate, hipStream_t stream);
int sdt_lion32_step(float* p, const float* g, float* mom, float* ema, uint16_t* w_bf16, int64_t n,
                    const double* sqnorm, double max_norm, double lr, double wd, double b1, double b2, double ema_rate,
                    hipStream_t stream);
int sdt_lion8_quantize(const float* x, int8_t* codes, float* inv_scale, int64_t n, int block_size, const float* thresholds,
                       hipStream_t stream);
int sdt_lion8_dequantize(const int8_t* codes, const float* inv_scale, float* x, int64_t n, int block_size,
                         hipStream_t stream);

/* ================= norms (flax nn.GroupNorm / nn.LayerNorm inside diffusers / transformers modules) */
/* No float atomics: every sum that crosses threads or workgroups is a set of single-writer partial sums added in a fixed
   order, so a launch is bitwise reproducible (the reference's jitted step is deterministic).
   stats: (B,G,2) f32 {sum,sumsq} written by fwd and consumed by bwd.
   parts / nparts (fwd, optional): statistics of x as nparts partial rows (B,nparts,G,2), produced by the contraction that wrote
   x (sdt_gemm_nt_bf16 gn_stats); NULL / 0: a statistics pass of its own, through `workspace` (then required).
   dres (bwd, optional): gradient of the branch that forked off x before the norm (residual / skip); dx = norm_bwd(dy) + dres
   in the same pass, replacing the add the reverse-mode sweep of x -> {norm(x), x} would otherwise launch.
   dgamma / dbeta (bwd): += by ONE writer per element (NULL for a frozen norm).  bwd workspaces are required. */
int sdt_groupnorm_fwd(const uint16_t* x, const float* gamma, const float* beta, uint16_t* y, float* stats, int B, int HW,
                      int C, int G, float eps, int fuse_silu, const float* parts, int nparts, void* workspace,
                      int64_t workspace_bytes, hipStream_t stream);
int64_t sdt_groupnorm_fwd_workspace_bytes(int B, int HW, int C, int G);
int sdt_groupnorm_bwd(const uint16_t* x, const uint16_t* dy, const float* stats, const float* gamma, const float* beta,
                      uint16_t* dx, float* dgamma, float* dbeta, const uint16_t* dres, int B, int HW, int C, int G, float eps,
                      int fuse_silu, void* workspace, int64_t workspace_bytes, hipStream_t stream);
int64_t sdt_groupnorm_bwd_workspace_bytes(int B, int HW, int C);
int sdt_layernorm_fwd(const uint16_t* x, const float* gamma, const float* beta, uint16_t* y, float* mean_rstd, int64_t M,
                      int C, float eps, hipStream_t stream);
/* defer_param_grads = 1: dgamma / dbeta are NOT touched; the per-workgroup partial rows [sdt_layernorm_bwd_partial_rows][2C] stay at
 * the start of `workspace` for a later sdt_norm_param_grads_group call (the sums of many norms in one launch). */
int sdt_layernorm_bwd(const uint16_t* x, const uint16_t* dy, const float* gamma, const float* mean_rstd, uint16_t* dx,
                      float* dgamma, float* dbeta, const uint16_t* dres, int64_t M, int C, int defer_param_grads, void* workspace,
                      int64_t workspace_bytes, hipStream_t stream);
int64_t sdt_layernorm_bwd_workspace_bytes(int64_t M, int C);
int64_t sdt_layernorm_bwd_partial_rows(int64_t M, int C);
typedef struct SdtNormGradJob {
  const float* partial;  /* [nrows][2C]: rows of {dgamma | dbeta} partial sums */
  float* dgamma;         /* [C], += */
  float* dbeta;          /* [C], += */
  int nrows, C;
} SdtNormGradJob;
int sdt_norm_param_grads_group(const SdtNormGradJob* jobs, int n, hipStream_t stream);
int sdt_norm_param_grads_group_max(void);

/* ================= dense contractions (flax nn.Dense / nn.Conv and their transposes) */
/* C[M,N] = A_g[M, taps*Kc] * Bt[N, taps*Kc]^T (+bias[N] f32) (+rowbias[m/rows_per_batch][N] bf16) (+residual).
 * Reduction segment t reads B rows at Bt + t*b_tap_stride; with gather_mode 0 (plain) and taps > 1, A is [M][taps*Kc] and
 * column block t contracts with segment t (the dgrad of Dense layers sharing an input, one launch).
 * b_kmajor = 1: the second operand is B[taps][Kc][ldb] instead (reduction index outermost, N columns contiguous) - the Flax
 * kernel layout itself ([in,out] Dense, HWIO conv), read through transposing LDS reads: the forward contractions consume the
 * bf16 mirror of the parameters as it stands and no transposed copy of the weights exists; the input gradients (which
 * contract over `out`) read the same buffer as Bt.  b_nseg > 0 (with b_kmajor): the N columns are b_nseg-wide segments, segment
 * s is the matrix at B + s*b_seg_stride with row pitch ldb (Dense layers that share an input, whose kernels are separate
 * leaves, as ONE forward GEMM).  ld_rowbias: row pitch of rowbias (0 = N): the per-image row bias of a layer may be a column
 * slice of a wider matrix (the time-embedding projections of all ResBlocks of one width come out of one GEMM). */
int sdt_gemm_nt_bf16(const uint16_t* A, const uint16_t* Bt, uint16_t* C, const float* bias, const uint16_t* rowbias,
                     const uint16_t* residual, int64_t M, int N, int Kc, int taps, int lda, int ldb,
                     int64_t b_tap_stride, int ldc, int ldres, int rows_per_batch, int gather_mode,
                     const SdtConvGeom* geom, void* workspace, int64_t workspace_bytes, float* gn_stats, int gn_groups,
                     int b_kmajor, int b_nseg, int64_t b_seg_stride, int ld_rowbias, hipStream_t stream);
/* gn_stats (optional, [batch][nparts][gn_groups][2] f32): the statistics {sum, sum of squares} of the bf16 outputs per image and
 * channel group, for the flax nn.GroupNorm that consumes this output, WRITTEN (not accumulated: no atomics, no zero fill) by
 * the epilogues as partial rows - output row tile r of an image writes rows 2r (groups that start inside the tile's columns)
 * and 2r+1 (the group that began in the column tile to its left) - which sdt_groupnorm_fwd(parts = gn_stats, nparts) adds up
 * in row order.  nparts = sdt_gemm_nt_gn_parts(...) for this problem; 0 means this shape cannot produce them (an output tile
 * would straddle two images, or a group is wider than a tile). */
int sdt_gemm_nt_gn_parts(int64_t M, int N, int Kc, int taps, int rows_per_batch, int gn_groups, int gather_mode,
                         const SdtConvGeom* geom);
/* bytes of scratch sdt_gemm_nt_bf16 wants for this shape (0 = none; split-K is used only when it is provided).
 * CONTRACT: its first 64 KiB (arrival counters) must be ZERO when the call is enqueued and are zero again when the launch completes
 * (every split of an output tile publishes its fp32 partial sums to its own slab, write-through; the split that arrives last
 * sums the slabs in split order, finishes the tile in the same launch and resets the counter); the slabs themselves need no
 * initialisation, so one buffer zeroed once serves every call issued on one stream.  No atomics touch the data: results are
 * bitwise reproducible from launch to launch. */
int64_t sdt_gemm_nt_workspace_bytes(int64_t M, int N, int Kc, int taps);
/* ---- transformer feed-forward with the GEGLU inside the first Dense layer's epilogue (diffusers FlaxFeedForward / FlaxGEGLU via
 * FlaxBasicTransformerBlock, reference training_utils.py:678-684): out = h[:, :F] * gelu_tanh(h[:, F:]) with h = x @ W1 + b1.
 * sdt_ff_geglu_fwd: x [M][K], W1 the Flax kernel [K][2F], bias [2F] fp32 -> h [M][2F] (kept for the backward) and out [M][F], one launch
 * (the epilogue pairs value and gate columns: no geglu launch, no re-read of h).  Same bf16 rounding points as sdt_gemm_nt_bf16 +
 * sdt_geglu_fwd: bit-identical results.  sdt_ff_geglu_supported: 1 when (M, F, K) is served (unsplit 128-tiles, F % 64 == 0); callers
 * fall back to the separate ops otherwise. */
int sdt_ff_geglu_supported(int64_t M, int F, int K);
int sdt_ff_geglu_fwd(const uint16_t* x, const uint16_t* W1, const float* bias, uint16_t* h, uint16_t* out, int64_t M, int F, int K,
                     hipStream_t stream);


/* dW[tap][K1_valid][N_valid] (f32) = A_g[M,K1]^T * dY[M,N]; optional fused bias gradient dbias[n] = sum_m dY[m][n]
 * (n < N_valid), NULL to skip.  Both are WRITTEN (plain stores, exactly one writer per element), not accumulated: the
 * destination needs no zero fill and no atomics are issued; the sums are bitwise reproducible from launch to launch.
 * n_seg > 0: the N columns are n_seg-wide segments and segment s is written at dW + s*seg_stride (row pitch ldw): one
 * launch for Dense layers that share their input (attention to_q/to_k/to_v), whose gradients are separate leaves.
 * workspace (optional, sdt_gemm_tn_workspace_bytes): lets the reduction over M be split across workgroups when the weight
 * is small (per-split fp32 partial tiles, summed in split order by the split that arrives last).  CONTRACT: its first
 * 64 KiB (arrival counters) must be ZERO when the call is enqueued and are zero again when the launch completes; the rest
 * is scratch, so one buffer zeroed once serves every call issued on one stream.  Without it one workgroup per output tile
 * reduces all of M (same results up to fp32 summation order). */
/* dw_bf16 (ABI 5; here and in the problem tables below): dW is a bf16 buffer - the fp32 sums are rounded once (RNE) when stored, which
 * is the precision of the reference's kernel cotangents (flax modules with dtype=bfloat16); ldw / w_tap_stride / seg_stride stay in
 * elements, sq_slots then hold the squares of the ROUNDED values.  dbias stays float32. */
int sdt_gemm_tn_wgrad(const uint16_t* A, const uint16_t* dY, void* dW, int dw_bf16, float* dbias, int64_t M, int K1, int N, int K1_valid,
                      int N_valid, int taps, int lda, int ldb, int ldw, int64_t w_tap_stride, int n_seg, int64_t seg_stride,
                      int gather_mode, const SdtConvGeom* geom, void* workspace, int64_t workspace_bytes, double* sq_slots,
                      hipStream_t stream);
/* sq_slots (here and in the problem tables below; NULL: off): sdt_wgrad_sq_slots(K1, N, taps) doubles the CALLER HAS ZEROED.  The waves
 * that store dW also add up the squares of what they store (in double: exact products) and WRITE each sum to a slot of their own, so
 * that optax.clip_by_global_norm's norm (reference training_utils.py:379, :732) needs no pass over the finished gradient buffer: the
 * caller adds all slots of all launches in index order (sdt_sum_f64_accumulate) - the float32 norm is the same rounding of the same sum
 * as sdt_sqnorm_accumulate's.  Only meaningful without a gradient exchange (the norm is that of the REDUCED gradient otherwise). */
int64_t sdt_wgrad_sq_slots(int K1, int N, int taps);
int sdt_sum_f64_accumulate(const double* x, int64_t n, double* out, void* workspace, int64_t workspace_bytes, hipStream_t stream);
int64_t sdt_gemm_tn_workspace_bytes(int64_t M, int K1, int N, int taps, int n_seg, int gather_mode, const SdtConvGeom* geom);
/* Several Dense-layer weight gradients (plain rows, taps = 1: the arguments of sdt_gemm_tn_wgrad with the same meaning) as ONE or
 * two launches (one per tile size).  The weight gradients of a transformer block (reverse of the flax nn.Dense layers inside diffusers
 * FlaxBasicTransformerBlock / transformers FlaxCLIPEncoderLayer) do not feed the input-gradient chain, so a caller may hold them
 * back and issue them together: each alone is 25 - 100 tiles and mostly prologue / tail.  Same arithmetic per problem as the single
 * call with the same split plan (bitwise reproducible; the split of the reduction over M, and so the fp32 summation order, may
 * differ from the single call's).  workspace: sdt_gemm_tn_wgrad_group_workspace_bytes under the split-workspace contract. */
typedef struct SdtTnProblem {
  const uint16_t* A;   /* x  [M][lda] */
  const uint16_t* dY;  /* dY [M][ldb] */
  void* dW;            /* [K1_valid][ldw] (or n_seg-wide column segments seg_stride apart), written: float32, or bf16 when dw_bf16 */
  float* dbias;        /* [N_valid] or NULL, written */
  int64_t M;
  int K1, N, K1_valid, N_valid, lda, ldb, ldw, n_seg;
  int64_t seg_stride;
  double* sq_slots;    /* or NULL: see sdt_gemm_tn_wgrad */
  int dw_bf16;
} SdtTnProblem;
int sdt_gemm_tn_wgrad_group(const SdtTnProblem* problems, int n, void* workspace, int64_t workspace_bytes, hipStream_t stream);
int64_t sdt_gemm_tn_wgrad_group_workspace_bytes(const SdtTnProblem* problems, int n);
int sdt_gemm_tn_wgrad_group_max(void);
/* The same for convolution weight gradients (sdt_gemm_tn_wgrad in fprop-gather mode, dW [kh*kw][K1_valid][N_valid] written): the
 * 3x3 / stride 1 / pad 1 ones the three-taps-per-workgroup kernel serves share grouped launches, the rest run one by one. */
typedef struct SdtConvWgradProblem {
  const uint16_t* A;   /* conv input x, NHWC, channel pitch lda */
  const uint16_t* dY;  /* [M][ldb], M = batch * out_h * out_w */
  void* dW;            /* float32, or bf16 when dw_bf16 */
  float* dbias;        /* or NULL */
  SdtConvGeom geom;
  int K1, N, K1_valid, N_valid, lda, ldb;
  double* sq_slots;    /* or NULL: see sdt_gemm_tn_wgrad */
  int dw_bf16;
} SdtConvWgradProblem;
int sdt_conv_wgrad_group(const SdtConvWgradProblem* problems, int n, void* workspace, int64_t workspace_bytes, hipStream_t stream);
int64_t sdt_conv_wgrad_group_workspace_bytes(const SdtConvWgradProblem* problems, int n);
/* db[n] += sum_m dy[m][n] (one writer per element) */
int sdt_colsum_accumulate(const uint16_t* dy, float* db, int64_t M, int N, int ld, void* workspace, int64_t workspace_bytes,
                          hipStream_t stream);
/* out[b][n] = bf16(sum of dy rows of batch b): gradient of the per-image time-embedding bias added by the conv epilogue
 * (diffusers FlaxResnetBlock2D: hidden_states + temb[:, None, None, :]) */
int sdt_colsum_batched_bf16(const uint16_t* dy, uint16_t* out, int batch, int64_t rows_per_batch, int N, int ld, void* workspace,
                            int64_t workspace_bytes, hipStream_t stream);
int64_t sdt_colsum_workspace_bytes(int batch, int64_t rows_per_batch, int N);

/* ================= attention (diffusers attention_flax.py + key_chunk_patch.patch; FlaxCLIPAttention) */
int sdt_attention_fwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out, float* lse,
                      const SdtAttnDesc* desc, hipStream_t stream);
/* workspace: sdt_attention_bwd_workspace_bytes(desc) bytes = B*H*Nq floats (delta = rowsum(dO*O)) plus, for few keys and
 * many queries (cross-attention), fp32 dK/dV partial sums: the dK/dV pass then splits the query range over workgroups. */
int sdt_attention_bwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* out, const uint16_t* dout,
                      const float* lse, uint16_t* dq, uint16_t* dk, uint16_t* dv, float* workspace, int64_t workspace_bytes,
                      const SdtAttnDesc* desc, hipStream_t stream);
int64_t sdt_attention_bwd_workspace_bytes(const SdtAttnDesc* desc);
int sdt_softmax_rows_inplace(uint16_t* x, int64_t rows, int n, float scale, hipStream_t stream);

/* ================= elementwise / data movement */
int sdt_act_fwd(const uint16_t* x, uint16_t* y, int64_t n, int act, hipStream_t stream);
int sdt_act_bwd(const uint16_t* x, const uint16_t* dy, uint16_t* dx, int64_t n, int act, hipStream_t stream);
int sdt_geglu_fwd(const uint16_t* h, uint16_t* out, int64_t M, int F, hipStream_t stream);
int sdt_geglu_bwd(const uint16_t* h, const uint16_t* dout, uint16_t* dh, int64_t M, int F, hipStream_t stream);
/* out = sum of n <= 32 bf16 tensors of numel elements (fp32 accumulation): the fan-in of a tensor consumed n times */
int sdt_sum_n_bf16(const uint16_t* const* inputs, int n, uint16_t* out, int64_t numel, hipStream_t stream);
int sdt_copy2d_bf16(uint16_t* dst, int64_t dst_stride, const uint16_t* src, int64_t src_stride, int64_t rows, int cols,
                    hipStream_t stream);
/* n (<= 32) column segments of a wide bf16 matrix <-> n narrow matrices in ONE launch: channel concatenation / its split (the UNet's
 * skip connections, diffusers unet_2d_blocks_flax.py jnp.concatenate(..., axis=-1)) and the gradient gather of column slices.
 * Segments sit side by side in `wide` in index order.  to_wide = 1: gather (parts[k] == NULL zero-fills segment k); 0: scatter. */
int sdt_copy_cols_bf16(uint16_t* wide, int64_t ld_wide, void* const* parts, const int64_t* ld_parts, const int* cols, int n,
                       int64_t rows, int to_wide, hipStream_t stream);
int sdt_add_bf16(const uint16_t* a, const uint16_t* b, uint16_t* y, int64_t n, hipStream_t stream);
int sdt_upsample2x_fwd(const uint16_t* x, uint16_t* y, int B, int H, int W, int C, hipStream_t stream);
int sdt_upsample2x_bwd(const uint16_t* dy, uint16_t* dx, int B, int H, int W, int C, hipStream_t stream);
int sdt_nchw_f32_to_nhwc_bf16(const float* x, uint16_t* y, int B, int C, int H, int W, int cpad, hipStream_t stream);
int sdt_nhwc_bf16_to_nchw_f32(const uint16_t* x, float* y, int B, int C, int H, int W, int cpad, hipStream_t stream);
int sdt_cast_f32_to_bf16(const float* x, uint16_t* y, int64_t n, hipStream_t stream);
int sdt_transpose_bf16(const uint16_t* x, uint16_t* y, int batch, int R, int C, hipStream_t stream);
/* fp32 master (Flax layout) -> bf16 compute copy W ([batch][Rp][Cp], zero padded) for every matrix leaf listed;
 * wt_bf16 (optional, may be NULL): additionally the per-tap transposes [batch][Cp][Rp] - the train path does not use them
 * (sdt_gemm_nt_bf16 b_kmajor reads W itself).  descs_device: array of {int64 src_off,w_off,wt_off; int32
 * batch,R,C,Rp,Cp,tile0,flags} (sdt_param_prepare_desc_size bytes each; flags reserved, 0) */
int sdt_param_prepare(const float* master, uint16_t* w_bf16, uint16_t* wt_bf16, const void* descs_device, int ndesc,
                      int total_tiles, hipStream_t stream);
int sdt_param_prepare_desc_size(void);
/* base[4*first .. 4*(first+count)) = 0 for every (first, count) pair of ranges_device (int64 pairs in float4 units, count <=
 * sdt_zero_ranges_chunk()): one launch clears the gradient leaves that are accumulated into (norm parameters, embeddings) -
 * the counterpart of jax.grad starting every leaf at zero (training_utils.py:719-729) for the leaves that need it */
int sdt_zero_ranges(float* base, const int64_t* ranges_device, int nranges, hipStream_t stream);
int sdt_zero_ranges_chunk(void);
int sdt_embedding_fwd(const int32_t* ids, const float* tok, const float* pos, uint16_t* out, int64_t rows, int S, int D,
                      hipStream_t stream);
int sdt_embedding_bwd(const int32_t* ids, const uint16_t* dout, float* dtok, float* dpos, int64_t rows, int S, int D,
                      hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SDT_H_ */
