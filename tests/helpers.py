"""Shared builders for model-level parity tests: identical synthetic weights / inputs for the oracle (CPU fp32)
and the HIP path (no checkpoints or datasets exist offline, so weights are seeded random)."""
import torch

from oracle import nets as onets
from oracle import schedulers as osched


def make_case(size="tiny", B=2, image=64, seed=0, k=1, sched="scaled_linear"):
    """Returns dict(cfgs, weights (host fp32 trees), batch, rand, sched_state)."""
    if size == "sd21_reduced":
        # BASELINE configs[3] structure (SD2.1 UNet: head dim 64, linear projections, 1024-wide context, erf-GELU text tower)
        # with a 2-layer text tower, for a parity run at a resolution the CPU oracle finishes in seconds
        cfgs = dict(unet=onets.unet_config("sd21"), vae=onets.vae_config("sd"),
                    clip=dict(vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=2,
                              num_attention_heads=16, max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5))
    elif size == "sd21":  # BASELINE configs[3]: SD2.1 UNet + OpenCLIP-H text tower
        cfgs = dict(unet=onets.unet_config("sd21"), vae=onets.vae_config("sd"), clip=onets.clip_config("openclip_h"))
    elif size == "sdxl":  # BASELINE configs[4]: SDXL UNet + CLIP-L and OpenCLIP-bigG text towers (one store)
        cfgs = dict(unet=onets.unet_config("sdxl"), vae=onets.vae_config("sd"), clip=onets.dual_clip_config())
    else:
        unet_name, vae_name, clip_name = {"tiny": ("tiny", "tiny", "tiny"), "sd15": ("sd15", "sd", "clip_l")}[size]
        cfgs = dict(unet=onets.unet_config(unet_name), vae=onets.vae_config(vae_name), clip=onets.clip_config(clip_name))
    w = dict(unet=onets.init_params(onets.unet_param_shapes(cfgs["unet"]), seed + 1),
             vae=onets.init_params(onets.vae_encoder_param_shapes(cfgs["vae"]), seed + 2),
             clip=onets.init_params(onets.clip_param_shapes(cfgs["clip"]), seed + 3))
    g = torch.Generator().manual_seed(seed + 10)
    ih, iw = (image, image) if isinstance(image, int) else image  # non-square aspect buckets: image=(height, width)
    lh, lw = ih // 8, iw // 8
    dual = "towers" in cfgs["clip"]
    vocab = cfgs["clip"]["towers"][0]["vocab_size"] if dual else cfgs["clip"]["vocab_size"]
    ids = torch.randint(0, vocab - 2, (B * k, 2, 77) if dual else (B * k, 77), generator=g)
    ids[..., 0] = vocab - 2
    ids[..., -1] = vocab - 1
    batch = dict(pixel_values=torch.rand(B, 3, ih, iw, generator=g) * 2 - 1, input_ids=ids)
    if dual:  # SDXL micro-conditioning: pooled text embedding (1280) and (orig h, w, crop top, left, target h, w)
        batch["text_embeds"] = torch.randn(B, 1280, generator=g)
        batch["time_ids"] = torch.tensor([[ih, iw, 0, 0, ih, iw]] * B, dtype=torch.int32)
    rand = dict(posterior_eps=torch.randn(B, lh, lw, 4, generator=g), noise=torch.randn(B, 4, lh, lw, generator=g),
                timesteps=torch.randint(0, 1000, (B,), generator=g))
    return dict(cfgs=cfgs, weights=w, batch=batch, rand=rand, sched_state=osched.create_state(sched), sched=sched)


def to_dev(d, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}


def build_hip_states(case, dev, *, prediction_type="epsilon", quantize=True, ema=False, block_size=16,
                     quant_excluded=None, wd_excluded=None):
    from stable_diffusion_training_amd import training_utils as tu
    tc = tu.TrainingConfig(
        model_path="synthetic", batch_size=case["batch"]["pixel_values"].shape[0], learning_rate=1e-6, unet_learning_rate=1e-6,
        text_encoder_learning_rate=1e-6, lr_scheduler="constant", adam_to_lion_scale_factor=7.0, compilation_cache_path="",
        keep_compiled_fn_in_cache=False, text_encoder_context_window=77, context_window_concatenation_count=1,
        aot_compile=True, strip_bos_eos_token=False, offset_noise_magnitude=0.0, min_snr_gamma_magnitude=0.0,
        perturbation_noise_magnitude=0.0, image_area_root=[512], minimum_axis_length=[512], beta_scheduler=case["sched"],
        prediction_type=prediction_type,
        excluded_layer_pattern_from_weight_decay=wd_excluded if wd_excluded is not None else ["bias", "scale", "embedding"],
        excluded_layer_from_quantization=quant_excluded if quant_excluded is not None else
        ["bias", "scale", "embedding", "conv_in", "conv_out", "time_embedding", "embeddings", "time_emb_proj"],
        quant_block_size=block_size, quantize_unet_state=quantize, quantize_text_encoder_state=quantize,
        accumulate_unet_ema=ema, accumulate_text_encoder_ema=ema, ema_rate=0.999)
    models = {"unet": {"unet_params": case["weights"]["unet"], "config": case["cfgs"]["unet"]},
              "vae": {"vae_params": case["weights"]["vae"], "config": case["cfgs"]["vae"]},
              "text_encoder": {"text_encoder_params": case["weights"]["clip"], "config": case["cfgs"]["clip"]}}
    return tc, tu.on_device_model_training_state(tc, models, device=dev)


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()
