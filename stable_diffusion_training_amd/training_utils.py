"""MI355X-native drop-in for the reference's train-step assembly layer (reference training_utils.py).

Same public names and argument meaning as the reference for the hot path only:
  TrainingConfig (:52-113), create_mask (:116-131), calculate_resolution_array (:134-174), FrozenModel (:40-49),
  create_lion_optimizer_states (:281-427), on_device_model_training_state (:430-501), train_step (:504-762),
  dp_compile_all_unique_resolution (:765-983).
What differs by design: there is no XLA program to compile - kernels are shape-generic - so the "compiled" table
maps pixel_values.shape -> a bound step callable; data parallelism is one process per GPU with a bucketed RCCL
all-reduce of the flat gradient buffer overlapped with backward (dp.GradReducer) instead of GSPMD; parameters,
optimizer state and EMA live in flat HBM buffers (params.ParamStore) updated in place (the reference donates them).
"""
from dataclasses import dataclass, field
from typing import Any, Callable

import numpy as np
import torch

from . import _lib, nets, ops, trace
from .checkpoint import load_models, load_training_state, save_model, save_training_state  # noqa: F401  (reference names)
from .params import EmaView, ParamStore, create_mask  # noqa: F401  (create_mask re-exported, reference name)
from .schedulers import DDPMScheduler


@dataclass
class TrainingConfig:
    """The 28 keys of model_properties.json that the reference's TrainingConfig consumes (training_utils.py:86-113)."""
    model_path: str
    batch_size: int
    learning_rate: float
    unet_learning_rate: float
    text_encoder_learning_rate: float
    lr_scheduler: str
    adam_to_lion_scale_factor: float
    compilation_cache_path: str
    keep_compiled_fn_in_cache: bool
    text_encoder_context_window: int
    context_window_concatenation_count: int
    aot_compile: bool
    strip_bos_eos_token: bool
    offset_noise_magnitude: float
    min_snr_gamma_magnitude: float
    perturbation_noise_magnitude: float
    image_area_root: list
    minimum_axis_length: list
    beta_scheduler: str
    prediction_type: str
    excluded_layer_pattern_from_weight_decay: list
    excluded_layer_from_quantization: list
    quant_block_size: int
    quantize_unet_state: bool
    quantize_text_encoder_state: bool
    accumulate_unet_ema: bool
    accumulate_text_encoder_ema: bool
    ema_rate: float

    @classmethod
    def from_dict(cls, config_dict):
        """training.py:38-40: pick the dataclass fields by name out of the full JSON dict."""
        return cls(**{k: config_dict[k] for k in cls.__dataclass_fields__})


def calculate_resolution_array(max_res_area=512 ** 2, bucket_lower_bound_res=256, rounding=64):
    """Aspect-ratio buckets (width, height) with area <= max_res_area, both sides multiples of `rounding`,
    minor axis >= bucket_lower_bound_res; mirrored around the square (training_utils.py:134-174)."""
    centroid = int(max_res_area ** 0.5)
    lo = bucket_lower_bound_res // rounding * rounding
    hi = centroid // rounding * rounding
    minor = np.arange(lo, hi + rounding, rounding)
    major = ((max_res_area / minor) // rounding * rounding).astype(int)
    n = len(minor) - 1 if minor[-1] == major[-1] else len(minor)  # do not repeat the square bucket
    w = np.concatenate([minor, major[:n][::-1]])
    h = np.concatenate([major, minor[:n][::-1]])
    return np.stack([w, h]).T


@dataclass
class FrozenModel:
    """(callable/config, params) bundle for the frozen VAE and the scheduler (training_utils.py:40-49)."""
    call: Any
    params: Any


@dataclass
class TrainState:
    """Stand-in for flax TrainState (training_utils.py:383-387): apply_fn + params + optimizer, all in `store`."""
    apply_fn: Callable
    store: ParamStore
    config: dict
    hyper: dict = field(default_factory=dict)

    @property
    def step(self):
        return self.store.count

    @property
    def params(self):
        return self.store


def create_lion_optimizer_states(models, train_unet=True, train_text_encoder=True, adam_to_lion_scale_factor=7,
                                 u_net_learning_rate=1e-6, text_encoder_learning_rate=1e-6,
                                 excluded_layer_pattern_from_weight_decay=(), excluded_layer_from_quantization=(),
                                 lion_8bit_block_size=None, quantize_unet_state=False, quantize_text_encoder_state=False,
                                 with_unet_ema=False, with_text_encoder_ema=False, device="cuda"):
    """training_utils.py:281-427.  lr = learning_rate / adam_to_lion_scale_factor, wd = 1e-2 * scale, b1=.9, b2=.99,
    chain(clip_by_global_norm(1), lion_8bit | lion).  Builds the flat HBM stores and loads the weights."""
    out = {"unet_state": None, "text_encoder_state": None}

    def make(spec, weights, cfg, fn, lr, quant, ema):
        store = ParamStore(spec, device=device, quantise=quant, quant_excluded=tuple(excluded_layer_from_quantization),
                           wd_excluded=tuple(excluded_layer_pattern_from_weight_decay),
                           block_size=lion_8bit_block_size or 16, with_ema=ema)
        store.load(weights)
        hyper = dict(lr=lr / adam_to_lion_scale_factor, wd=1e-2 * adam_to_lion_scale_factor, b1=0.9, b2=0.99, max_norm=1.0)
        return TrainState(fn, store, cfg, hyper)

    if train_unet:
        m = models["unet"]
        out["unet_state"] = make(nets.unet_spec(m["config"]), m["unet_params"], m["config"], nets.unet_forward,
                                 u_net_learning_rate, quantize_unet_state, with_unet_ema)
    if train_text_encoder:
        m = models["text_encoder"]
        out["text_encoder_state"] = make(nets.clip_text_spec(m["config"]), m["text_encoder_params"], m["config"],
                                         nets.clip_text_forward, text_encoder_learning_rate,
                                         quantize_text_encoder_state, with_text_encoder_ema)
    return out


def on_device_model_training_state(training_config: TrainingConfig, models=None, device="cuda"):
    """training_utils.py:430-501.  `models`: load_models' result - host weight trees + configs
    ({"unet": {"unet_params", "config"}, "vae": {"vae_params", "config"}, "text_encoder": {...}}); None reads the
    diffusers directory at training_config.model_path (checkpoint.load_models).  Note the reference passes NEITHER learning
    rate from the config (:432-442) - the effective lr is the 1e-6 default / 7 - which is mirrored here."""
    _lib.require_device()
    if models is None:
        models = load_models(training_config)
    states = create_lion_optimizer_states(
        models, train_text_encoder=True, train_unet=True, adam_to_lion_scale_factor=7,
        excluded_layer_pattern_from_weight_decay=training_config.excluded_layer_pattern_from_weight_decay,
        excluded_layer_from_quantization=training_config.excluded_layer_from_quantization,
        lion_8bit_block_size=training_config.quant_block_size,
        quantize_unet_state=training_config.quantize_unet_state,
        quantize_text_encoder_state=training_config.quantize_text_encoder_state,
        with_unet_ema=training_config.accumulate_unet_ema, with_text_encoder_ema=training_config.accumulate_text_encoder_ema,
        device=device)
    vae_cfg = models["vae"]["config"]
    vae_store = ParamStore(nets.vae_encoder_spec(vae_cfg), device=device, trainable=False)
    vae_store.load(models["vae"]["vae_params"])
    vae_store.prepare()
    # the train path holds the encoder only; save_model(vae_params=frozen_vae.params) (training.py:151-158) must still write
    # the whole frozen VAE, so the store keeps a reference to the host tree it was loaded from
    vae_store.full_tree = models["vae"]["vae_params"]
    frozen_vae = FrozenModel(call=vae_cfg, params=vae_store)
    sched = DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule=training_config.beta_scheduler,
                          num_train_timesteps=1000, prediction_type=training_config.prediction_type)  # :223-230
    frozen_sched = FrozenModel(call=sched, params=sched.create_state(device))
    unet_state, te_state = states["unet_state"], states["text_encoder_state"]
    unet_ema = EmaView(unet_state.store) if training_config.accumulate_unet_ema else None
    te_ema = EmaView(te_state.store) if training_config.accumulate_text_encoder_ema else None
    model_object_dict = {"unet": unet_state.config, "vae": vae_cfg, "text_encoder": te_state.config, "schedulers": sched}
    return unet_state, te_state, unet_ema, te_ema, frozen_vae, frozen_sched, model_object_dict


def _min_snr_weights(sched_state, timesteps, gamma, prediction_type):
    """training_utils.py:546-568 (tiny gather on (B,) values)."""
    ac = sched_state.alphas_cumprod
    snr = (ac / (1 - ac))[timesteps.long()]
    m = torch.minimum(snr, torch.full_like(snr, gamma))
    return (m / (snr + 1) if prediction_type == "v_prediction" else m / snr).to(torch.float32).contiguous()


def assemble_context(hs, batch, strip_bos_eos_token):
    """training_utils.py:643-673: (B*k,77,D) -> (B,k,77,D) -> (B,L,D).  k=1 without stripping is a free view."""
    d = hs.shape[-1]
    e = hs.view(batch, -1, 77, d)
    if strip_bos_eos_token:
        return torch.cat([e[:, 0, :-1, :], e[:, 1:-1, 1:-1, :].reshape(batch, -1, d), e[:, -1, 1:, :]], dim=1).contiguous()
    return e.view(batch, -1, d)


_FUSED_NORM = __import__("os").environ.get("SDT_FUSED_NORM", "1") != "0"  # developer A/B: 0 = always the pass over the gradient buffer


def train_step(unet_state, text_encoder_state, unet_ema_params, text_encoder_ema_params, batch, train_rng,
               frozen_vae_state, frozen_noise_scheduler_state, strip_bos_eos_token=True, offset_noise_magnitude=0.0,
               min_snr_gamma_magnitude=0.0, perturbation_noise_magnitude=0.0, ema_rate=0.0, *, rand=None, reducer=None,
               vae_scale=0.18215, aux=None):
    """One DDPM training step on this rank's shard of the batch (training_utils.py:504-762), in place.

    batch: {"pixel_values": f32 (B,3,H,W) NCHW device tensor, "input_ids": i32 (B*k,77), "attention_mask": unused}.
    train_rng: a torch.Generator on the device (the reference threads a JAX key; joint distribution only matters).
    rand: optional dict of explicit draws for parity tests (posterior_eps NHWC, noise NCHW, timesteps[, offset_noise,
    perturb_noise]) - the reference's threefry stream is not reproducible outside JAX.
    Returns the reference's 6-tuple; metrics["loss"] is a device scalar (read it to synchronise, as training.py:238-245)."""
    us, ts = unet_state.store, text_encoder_state.store
    vae_store, vae_cfg = frozen_vae_state.params, frozen_vae_state.call
    sched, sched_state = frozen_noise_scheduler_state.call, frozen_noise_scheduler_state.params
    dev = us.device
    stream = torch.cuda.current_stream().cuda_stream
    rand = rand or {}
    px = batch["pixel_values"]
    B, C_in, H, W = px.shape
    L = vae_cfg["latent_channels"]

    ops.gn_arena_begin(dev)  # GroupNorm statistics accumulated by producer epilogues: one memset per step
    if reducer is not None:
        reducer.begin_step()

    # VAE encode -> posterior sample -> NCHW * 0.18215           (training_utils.py:574-586)
    pix = torch.empty(B, H, W, 8, dtype=torch.bfloat16, device=dev)
    _lib.call("sdt_nchw_f32_to_nhwc_bf16", px.data_ptr(), pix.data_ptr(), B, C_in, H, W, 8, stream)
    with trace.phase("vae_encode"):
        moments = nets.vae_encode_moments(vae_store, vae_cfg, pix)
    h, w = moments.shape[1], moments.shape[2]
    eps = rand.get("posterior_eps")
    if eps is None:
        eps = torch.randn(B, h, w, L, device=dev, generator=train_rng)
    latents = torch.empty(B, L, h, w, dtype=torch.float32, device=dev)
    _lib.call("sdt_vae_posterior_sample", moments.data_ptr(), eps.data_ptr(), latents.data_ptr(), B, L, h, w,
              moments.shape[3], vae_scale, stream)

    # The frozen VAE is all the step has read so far: the trained weights are first touched here.  With the sharded optimizer the
    # all-gather of the bf16 mirrors the previous step's owners wrote is still running beside the VAE encode (dp.GradReducer.wait_gathered)
    if reducer is not None:
        reducer.wait_gathered()
    with trace.phase("prepare_weights"):
        us.prepare()
        ts.prepare()
        us.zero_grad()
        ts.zero_grad()

    # noise, timesteps                                            (training_utils.py:590-624)
    noise = rand.get("noise")
    if noise is None:
        noise = torch.randn(B, L, h, w, device=dev, generator=train_rng)
    if offset_noise_magnitude:
        off = rand.get("offset_noise")
        if off is None:
            off = torch.randn(B, L, 1, 1, device=dev, generator=train_rng)
        noise = noise + off * offset_noise_magnitude
    if perturbation_noise_magnitude:
        pn = rand.get("perturb_noise")
        if pn is None:
            pn = torch.randn(B, L, h, w, device=dev, generator=train_rng)
        noise = noise + perturbation_noise_magnitude * pn
    noise = noise.contiguous()
    timesteps = rand.get("timesteps")
    if timesteps is None:
        timesteps = torch.randint(0, sched.num_train_timesteps, (B,), device=dev, generator=train_rng)
    timesteps = timesteps.to(torch.int32).contiguous()

    # forward diffusion (+ v target)                              (training_utils.py:628-633, 688-701)
    noisy, target, noisy_nchw = sched.add_noise_and_target(sched_state, latents, noise, timesteps, cpad=8,
                                                           want_noisy_nchw=aux is not None)

    # text encoder + context assembly                             (training_utils.py:635-674)
    ids = batch["input_ids"]
    with trace.phase("text_encoder_forward"):
        hs = text_encoder_state.apply_fn(ts, text_encoder_state.config, ids if ids.dtype == torch.int32 else ids.to(torch.int32))
    ctx = assemble_context(hs, B, strip_bos_eos_token)

    # UNet                                                        (training_utils.py:678-684)
    added = None
    if unet_state.config.get("addition_embed_type") == "text_time":
        # SDXL micro-conditioning.  Beyond the reference (its call passes no added_cond_kwargs): the batch carries the pooled text
        # embedding and the six size / crop ids as explicit inputs (SURVEY.md §8(d) note on configs[4])
        added = {"text_embeds": batch["text_embeds"], "time_ids": batch["time_ids"]}
    with trace.phase("unet_forward"):
        pred = unet_state.apply_fn(us, unet_state.config, noisy, timesteps, ctx, added)

    # MSE (+ min-SNR), forward and d loss / d pred in one launch  (training_utils.py:704-709)
    wts = None
    if min_snr_gamma_magnitude:
        wts = _min_snr_weights(sched_state, timesteps, min_snr_gamma_magnitude, sched.prediction_type)
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    dpred = torch.empty_like(pred)
    C_out = unet_state.config["out_channels"]
    rws = ops.reduce_workspace(_lib.load().sdt_reduce_workspace_bytes(), dev)
    _lib.call("sdt_mse_loss_fwd_bwd", pred.data_ptr(), target.data_ptr(), None if wts is None else wts.data_ptr(),
              loss.data_ptr(), dpred.data_ptr(), B, C_out, h, w, pred.shape[3], rws.data_ptr(), rws.numel(), stream)
    if aux is not None:
        aux.update(latents=latents, noisy=noisy_nchw, ctx=ctx.detach(), pred=pred.detach(), target=target, moments=moments)

    # reverse mode through UNet and text encoder                  (training_utils.py:719-729)
    # one process: the norm clip_by_global_norm needs is that of the gradients as the weight-gradient kernels write them - they leave
    # its partial sums behind (ops.sq_begin / sq_end), and the 4-byte-per-parameter pass over the finished buffer is not needed
    fused_norm = reducer is None and _FUSED_NORM and dev.type == "cuda"
    if fused_norm:
        ops.sq_begin(us)
        ops.sq_begin(ts)
    with trace.phase("backward_unet_text"), ops.wgrad_grouping():  # Dense weight gradients are issued a dozen per launch
        pred.backward(dpred)
    sq_u = ops.sq_end(us) if fused_norm else None
    sq_t = ops.sq_end(ts) if fused_norm else None

    # data-parallel mean of the gradients (implicit all-reduce under GSPMD in the reference)
    if reducer is not None:
        with trace.phase("grad_exchange_finish"):
            reducer.finish()
            loss = reducer.mean_scalar(loss)

    # clip -> Lion(8-bit) -> decay -> -lr -> apply -> EMA         (training_utils.py:732-746)
    ur = ema_rate if (ema_rate and unet_ema_params is not None) else 0.0
    tr = ema_rate if (ema_rate and text_encoder_ema_params is not None) else 0.0
    with trace.phase("optimizer_clip_lion8_ema"):
        us.optimizer_step(ema_rate=ur, shard=None if reducer is None else reducer.shard_pieces(us), sq_partials=sq_u, **unet_state.hyper)
        ts.optimizer_step(ema_rate=tr, shard=None if reducer is None else reducer.shard_pieces(ts), sq_partials=sq_t, **text_encoder_state.hyper)
        if reducer is not None:
            reducer.after_optimizer()  # sharded optimizer: all-gather the bf16 weight mirrors the owners have just written

    ops.gn_arena_end(dev)
    new_unet_ema = unet_ema_params if ur else None
    new_te_ema = text_encoder_ema_params if tr else None
    return unet_state, text_encoder_state, new_unet_ema, new_te_ema, {"loss": loss[0]}, train_rng


# Other threads keep making HIP calls while a step is captured (RCCL's watchdog polls events, loaders pin memory): only the
# capturing thread's own unsafe calls should fail the capture.
_CAPTURE_MODE = "thread_local"


class _GraphedStep:
    """One resolution's train_step, captured once into a HIP graph and replayed (the MI355X counterpart of the
    reference jit-compiling train_step per bucket shape, training_utils.py:765-983).

    A step is ~1,250 kernel launches; issued from Python the device idles ~15 % of the step waiting for the host, so
    after `warmup` eager calls (which size every workspace and set kernel attributes) the whole step - VAE encode, CLIP,
    UNet forward/backward, clip + Lion-8bit + EMA - is captured on a side stream and replayed with the batch copied into
    static buffers.  Calls that pass aux= taps stay eager.  With a data-parallel reducer the step is captured as two graphs around
    the gradient exchange (_capture_split)."""

    _pool = None  # graphs of different resolutions are never replayed concurrently: they share one memory pool

    def __init__(self, fn, warmup=2, reducer=None):
        self.fn, self.warmup, self.calls = fn, warmup, 0
        self.graph = self.static = self.static_rand = self.out = self.sig = None
        # multi-rank: graph A (forward + backward) | eager bucketed all-reduce behind per-bucket events | graph B (optimizer)
        self.reducer = reducer if (reducer is not None and reducer.active) else None
        self.graph_b = self.plan = None
        self.disabled = False

    def _capture(self, us, ts, ue, te, batch, rng, vae, sched, rand):
        self.static = {k: v.clone() for k, v in batch.items() if torch.is_tensor(v)}
        self.static_rand = None if rand is None else {k: v.clone() for k, v in rand.items()}
        self.sig = self._sig(us, ts, ue, te, rng, vae, sched, rand)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        if rng is not None and hasattr(g, "register_generator_state"):
            g.register_generator_state(rng)
        if _GraphedStep._pool is None:
            _GraphedStep._pool = torch.cuda.graph_pool_handle()
        if self.reducer is None:
            with torch.cuda.graph(g, pool=_GraphedStep._pool, capture_error_mode=_CAPTURE_MODE):
                self.out = self.fn(us, ts, ue, te, self.static, rng, vae, sched, rand=self.static_rand)
        else:
            self._capture_split(g, us, ts, ue, te, rng, vae, sched)
            if self.disabled:
                return
        # capturing executed nothing on the device, but the host-side step counters moved: undo, replay() re-applies
        us.store.count -= 1
        ts.store.count -= 1
        self.graph = g

    def _capture_split(self, g, us, ts, ue, te, rng, vae, sched):
        """Two graphs around the gradient exchange: reducer.finish() (called by train_step after the backward) ends graph A
        and begins graph B; the buckets' completion points are event-record nodes of graph A (dp.ExchangePlan)."""
        import gc
        import sys
        from . import dp
        red, pool = self.reducer, _GraphedStep._pool
        gb = torch.cuda.CUDAGraph()
        plan = dp.ExchangePlan(us.store.device)
        state = {"cur": None}
        counts = (us.store.count, ts.store.count)

        def split():
            g.capture_end()
            state["cur"] = None
            gb.capture_begin(pool=pool, capture_error_mode=_CAPTURE_MODE)
            state["cur"] = gb

        plan.split = split
        gc.collect()
        torch.cuda.empty_cache()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                g.capture_begin(pool=pool, capture_error_mode=_CAPTURE_MODE)
                state["cur"] = g
                red.capture = plan
                self.out = self.fn(us, ts, ue, te, self.static, rng, vae, sched, rand=self.static_rand)
                if state["cur"] is not gb:
                    raise RuntimeError("train_step never reached reducer.finish()")
                gb.capture_end()
                state["cur"] = None
        except Exception as e:  # leave the stream usable and fall back to eager steps (every rank takes the same branch)
            if state["cur"] is not None:
                try:
                    state["cur"].capture_end()
                except Exception:
                    pass
            us.store.count, ts.store.count = counts
            plan.close()
            self.disabled = True
            print(f"[sdt] step graph capture failed ({type(e).__name__}: {e}); this shape runs eagerly", file=sys.stderr)
            return
        finally:
            red.capture = None
        torch.cuda.current_stream().wait_stream(side)
        self.graph_b, self.plan = gb, plan

    @staticmethod
    def _sig(us, ts, ue, te, rng, vae, sched, rand):
        return (id(us), id(ts), id(ue), id(te), id(rng), id(vae), id(sched), None if rand is None else tuple(sorted(rand)))

    def __call__(self, us, ts, ue, te, batch, rng, vae, sched, rand=None, **extra):
        if extra:  # aux= taps and per-call overrides: eager
            return self.fn(us, ts, ue, te, batch, rng, vae, sched, rand=rand, **extra)
        if self.graph is None:
            if self.calls < self.warmup or self.disabled:
                self.calls += 1
                return self.fn(us, ts, ue, te, batch, rng, vae, sched, rand=rand)
            self._capture(us, ts, ue, te, batch, rng, vae, sched, rand)
            if self.disabled:
                return self.fn(us, ts, ue, te, batch, rng, vae, sched, rand=rand)
        if self.sig != self._sig(us, ts, ue, te, rng, vae, sched, rand):
            raise ValueError("a captured train_step is bound to the state objects (and rand= keys) it was captured with")
        for k, v in self.static.items():
            v.copy_(batch[k], non_blocking=True)
        if rand is not None:
            for k, v in self.static_rand.items():
                v.copy_(rand[k], non_blocking=True)
        self.graph.replay()
        if self.graph_b is not None:
            self.reducer.run_exchange(self.plan)  # overlaps the rest of graph A bucket by bucket
            self.graph_b.replay()
            self.reducer.run_post(self.plan)      # sharded optimizer: all-gather of the weight mirrors
            if self.reducer.shard:
                # what optimizer_step(shard=...) does when it runs as Python: a replayed sharded sweep leaves fp32 master / EMA / momentum
                # current only on the owners of the slices, so exports must raise until GradReducer.gather_state() (ParamStore._gather)
                us.store.state_whole = False
                ts.store.state_whole = False
        us.store.count += 1
        ts.store.count += 1
        o = self.out
        return o[0], o[1], o[2], o[3], {"loss": o[4]["loss"].clone()}, o[5]


def dp_compile_all_unique_resolution(unet_state, text_encoder_state, unet_ema_params, text_encoder_ema_params,
                                     frozen_vae, frozen_schedulers, training_config: TrainingConfig, reducer=None,
                                     per_device_batch=None, use_graph=None, step_overrides=None):
    """training_utils.py:765-983: table {pixel_values.shape: step callable}.  Keys are the bucket shapes
    (B, 3, bucket[0], bucket[1]) of every (image_area_root, minimum_axis_length) pair.  Nothing is compiled up front:
    with use_graph (the default; SDT_GRAPH=0 turns it off) each shape captures its step into HIP graphs on its third call:
    one graph for a single process; with an active reducer, graph A (forward + backward) and graph B (optimizer) around
    the bucketed all-reduce, which stays outside the graphs and overlaps graph A bucket by bucket (dp.ExchangePlan)."""
    import os
    B = per_device_batch or training_config.batch_size
    kw = dict(strip_bos_eos_token=training_config.strip_bos_eos_token,
              offset_noise_magnitude=training_config.offset_noise_magnitude,
              min_snr_gamma_magnitude=training_config.min_snr_gamma_magnitude,
              perturbation_noise_magnitude=training_config.perturbation_noise_magnitude,
              ema_rate=training_config.ema_rate)
    kw.update(step_overrides or {})  # e.g. vae_scale=0.13025 for the SDXL VAE (the reference hard-codes 0.18215, :586)
    if use_graph is None:
        env = os.environ.get("SDT_GRAPH")
        use_graph = env != "0" and unet_state.store.device.type == "cuda"

    def bound(us, ts, ue, te, batch, rng, vae, sched, **extra):
        return train_step(us, ts, ue, te, batch, rng, vae, sched, reducer=reducer, **kw, **extra)

    table = {}
    for area_root, min_axis in zip(training_config.image_area_root, training_config.minimum_axis_length):
        for bucket in calculate_resolution_array(area_root ** 2, min_axis, 64):
            table[(B, 3, int(bucket[0]), int(bucket[1]))] = _GraphedStep(bound, reducer=reducer) if use_graph else bound
    return table
