"""Oracle (test infrastructure): the reference's whole train_step, on the CPU, in fp32.

Follows training_utils.py:504-762 step by step.  The reference draws its randomness
from JAX threefry keys (training_utils.py:530, 590-624) which cannot be reproduced
outside JAX, so every random tensor is an explicit input here:
posterior eps (VAE sample), noise, timesteps, and the optional offset / perturbation
draws.  Gradients come from torch autograd over the fp32 graph (jax.value_and_grad,
training_utils.py:719-729, argnums=[0,1] = UNet and text-encoder params only).
"""
import numpy as np
import torch

from . import lion8, nets, schedulers


def _to_np(d):
    return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in d.items()}


def compute_loss(unet_p, te_p, vae_p, sched_state, cfgs, batch, rand, *, prediction_type="epsilon",
                 strip_bos_eos_token=False, offset_noise_magnitude=0.0, min_snr_gamma_magnitude=0.0,
                 perturbation_noise_magnitude=0.0, vae_scale=0.18215, return_aux=False):
    """training_utils.py:570-710.  cfgs = dict(unet=, vae=, clip=).  batch: pixel_values f32
    (B,3,H,W) NCHW, input_ids (B*k,77).  rand: posterior_eps (B,h,w,4) NHWC, noise (B,4,h,w),
    timesteps (B,) [+ offset_noise (B,4,1,1), perturb_noise (B,4,h,w)]."""
    with torch.no_grad():  # VAE is frozen: argnums=[0,1] (training_utils.py:574-586, :720)
        moments = nets.vae_encode_moments(vae_p, cfgs["vae"], batch["pixel_values"])
        latents = nets.vae_sample_latents(moments, rand["posterior_eps"], vae_scale).contiguous()
    noise = rand["noise"].to(torch.float32)
    if offset_noise_magnitude:  # :594-606
        noise = noise + rand["offset_noise"] * offset_noise_magnitude
    if perturbation_noise_magnitude:  # :608-615
        noise = noise + perturbation_noise_magnitude * rand["perturb_noise"]
    t = rand["timesteps"]
    t_np = t.cpu().numpy()
    noisy = torch.from_numpy(schedulers.add_noise(sched_state, latents.numpy(), noise.numpy(), t_np))  # :628-633
    hs = nets.clip_text_forward(te_p, cfgs["clip"], batch["input_ids"])  # :635-640
    ctx = nets.assemble_context(hs, latents.shape[0], strip_bos_eos_token)  # :643-673
    added = None
    if cfgs["unet"].get("addition_embed_type") == "text_time":  # SDXL (beyond the reference: explicit micro-conditioning inputs)
        added = dict(text_embeds=batch["text_embeds"], time_ids=batch["time_ids"])
    pred = nets.unet_forward(unet_p, cfgs["unet"], noisy, t, ctx, added)  # :678-684
    if prediction_type == "epsilon":  # :688-701
        target = noise
    elif prediction_type == "v_prediction":
        target = torch.from_numpy(schedulers.get_velocity(sched_state, latents.numpy(), noise.numpy(), t_np))
    else:
        raise ValueError(f"Unknown prediction type {prediction_type}")
    loss = (target - pred) ** 2  # :704
    if min_snr_gamma_magnitude:  # :706-708
        w = schedulers.min_snr_weight(sched_state, t_np, min_snr_gamma_magnitude, prediction_type)
        loss = loss * torch.from_numpy(w)[:, None, None, None]
    loss = loss.mean()  # :709
    if return_aux:
        return loss, dict(latents=latents, noisy=noisy, ctx=ctx, pred=pred, target=target, moments=moments)
    return loss


def train_step(unet_p, te_p, vae_p, sched_state, cfgs, batch, rand, opt, *, unet_state=None, te_state=None,
               unet_ema=None, te_ema=None, ema_rate=0.0, **loss_kw):
    """training_utils.py:504-762.  ``opt`` = dict(lr, wd, b1, b2, block_size, quantize_unet,
    quantize_te, wd_excluded, quant_excluded) mirroring create_lion_optimizer_states
    (training_utils.py:281-427; effective lr = 1e-6/7, wd = 0.07 - SURVEY.md §5 config quirks).
    Params are dicts of torch fp32 tensors; returns a dict with new params / states / loss / aux."""
    up = {k: v.detach().clone().requires_grad_(True) for k, v in unet_p.items()}
    tp = {k: v.detach().clone().requires_grad_(True) for k, v in te_p.items()}
    loss, aux = compute_loss(up, tp, vae_p, sched_state, cfgs, batch, rand, return_aux=True, **loss_kw)
    leaves = list(up.values()) + list(tp.values())
    grads = torch.autograd.grad(loss, leaves, allow_unused=True)
    gu = {k: (g if g is not None else torch.zeros_like(v)) for (k, v), g in zip(up.items(), grads[: len(up)])}
    gt = {k: (g if g is not None else torch.zeros_like(v)) for (k, v), g in zip(tp.items(), grads[len(up):])}
    out = dict(loss=float(loss.detach()), aux={k: v.detach() for k, v in aux.items()}, unet_grads=gu, te_grads=gt)

    def apply(params, grads_, state, quantize):
        pn, gn = _to_np(params), _to_np(grads_)
        qmask = lion8.create_mask(pn.keys(), opt.get("quant_excluded", [])) if quantize else None
        if state is None:
            state = lion8.init_state(pn, qmask, opt["block_size"])
        wdx = opt.get("wd_excluded", [])
        dmask = lion8.create_mask(pn.keys(), wdx) if wdx else None
        return lion8.lion_step(pn, gn, state, lr=opt["lr"], wd=opt["wd"], b1=opt.get("b1", 0.9),
                               b2=opt.get("b2", 0.99), block_size=opt["block_size"], decay_mask=dmask, clip=1.0)

    out["unet_params"], out["unet_state"], out["unet_gnorm"] = apply(unet_p, gu, unet_state, opt["quantize_unet"])
    out["te_params"], out["te_state"], out["te_gnorm"] = apply(te_p, gt, te_state, opt["quantize_te"])
    if ema_rate and unet_ema is not None:  # :735-746
        out["unet_ema"] = lion8.ema_update(unet_ema, out["unet_params"], ema_rate)
    if ema_rate and te_ema is not None:
        out["te_ema"] = lion8.ema_update(te_ema, out["te_params"], ema_rate)
    return out


DEFAULT_OPT = dict(lr=1e-6 / 7, wd=1e-2 * 7, b1=0.9, b2=0.99, block_size=16, quantize_unet=True,
                   quantize_te=True, wd_excluded=["bias", "scale", "embedding"],
                   quant_excluded=["bias", "scale", "embedding", "conv_in", "conv_out", "time_embedding",
                                   "embeddings", "time_emb_proj"])
