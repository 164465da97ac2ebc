"""Sampling / validation path: the reference's FlaxStableDiffusionPipeline._generate
(models/pipeline_flax_stable_diffusion.py:160-254) on the same HIP operators train_step uses - CLIP text encoder, UNet
forward, the fused classifier-free-guidance + DDIM update (`sdt_ddim_cfg_step`), VAE decoder.  The reference uses this
class during training only as the checkpoint container (training_utils.py:1007-1023); sampling is how a run is eyeballed.

No CPU fallback: every tensor op here is a libsdtrain_hip.so launch or torch device plumbing."""
import torch

from . import _lib, nets, ops
from .params import EmaView, ParamStore
from .schedulers import DDIMScheduler


def _store_of(x):
    if isinstance(x, EmaView):
        raise TypeError("sample from EMA weights by loading them into a store (load_models on the -EMA directory)")
    return x.store if hasattr(x, "store") else x


class StableDiffusionPipeline:
    """unet / text_encoder: TrainState or ParamStore (the live training parameters are sampled in place, no copy);
    vae_params: host tree of the full VAE (decoder + post_quant_conv are loaded into a frozen store); configs as in load_models."""

    def __init__(self, unet, text_encoder, vae_params, unet_config, text_encoder_config, vae_config, scheduler=None,
                 scaling_factor=0.18215, device=None):
        _lib.require_device()
        self.unet, self.text_encoder = _store_of(unet), _store_of(text_encoder)
        self.unet_config, self.text_encoder_config, self.vae_config = unet_config, text_encoder_config, vae_config
        self.device = torch.device(device) if device is not None else self.unet.device
        self.vae_decoder = ParamStore(nets.vae_decoder_spec(vae_config), device=self.device, trainable=False)
        if any(isinstance(v, dict) for v in vae_params.values()):  # nested Flax tree -> "a/b/kernel" paths
            from .checkpoint import flatten_tree
            vae_params = flatten_tree(vae_params)
        self.vae_decoder.load(vae_params)
        self.vae_decoder.prepare()
        # the reference's placeholder (training_utils.py:998-1004); pass a DDIMScheduler built for the trained schedule instead
        self.scheduler = scheduler or DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                                    num_train_timesteps=1000, prediction_type="v_prediction")
        self.scaling_factor = scaling_factor
        self.vae_scale_factor = 2 ** (len(vae_config["block_out_channels"]) - 1)

    def _uncond_ids(self, batch, length):
        """tokenizer([""] * batch, padding="max_length") for CLIP: <|startoftext|>, then <|endoftext|> (also the pad token)."""
        v = self.text_encoder_config["vocab_size"]
        ids = torch.full((batch, length), v - 1, dtype=torch.int32, device=self.device)
        ids[:, 0] = v - 2
        return ids

    @torch.no_grad()
    def generate(self, prompt_ids, num_inference_steps=50, height=512, width=512, guidance_scale=7.5, latents=None,
                 neg_prompt_ids=None, generator=None, return_latents=False):
        """_generate (:160-254).  prompt_ids int (B,77) device tensor; latents optional f32 (B,C,h,w) initial noise; returns the
        image (B,H,W,3) float32 in [0,1] on the device (and the final latents when return_latents)."""
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        dev = self.device
        stream = torch.cuda.current_stream().cuda_stream
        B = prompt_ids.shape[0]
        C = self.unet_config["in_channels"]
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        if latents is None:
            latents = torch.randn(B, C, h, w, device=dev, dtype=torch.float32, generator=generator)
        elif tuple(latents.shape) != (B, C, h, w):
            raise ValueError(f"Unexpected latents shape, got {tuple(latents.shape)}, expected {(B, C, h, w)}")
        if neg_prompt_ids is None:
            neg_prompt_ids = self._uncond_ids(B, prompt_ids.shape[-1])
        self.unet.prepare()
        self.text_encoder.prepare()
        ids = torch.cat([neg_prompt_ids.to(device=dev, dtype=torch.int32), prompt_ids.to(device=dev, dtype=torch.int32)])
        context = nets.clip_text_forward(self.text_encoder, self.text_encoder_config, ids).detach()  # [negative | prompt] (:191)

        lat = (latents.to(device=dev, dtype=torch.float32) * self.scheduler.init_noise_sigma).contiguous().clone()
        cpad = (C + 7) // 8 * 8
        x_in = torch.empty(2 * B, h, w, cpad, dtype=torch.bfloat16, device=dev)
        for half in (x_in[:B], x_in[B:]):
            _lib.call("sdt_nchw_f32_to_nhwc_bf16", lat.data_ptr(), half.data_ptr(), B, C, h, w, cpad, stream)
        t_dev = torch.empty(2 * B, dtype=torch.int32, device=dev)
        for t in self.scheduler.set_timesteps(num_inference_steps):
            t_dev.fill_(int(t))
            ops.gn_arena_begin(dev)
            pred = nets.unet_forward(self.unet, self.unet_config, x_in, t_dev, context)
            ops.gn_arena_end(dev)
            self.scheduler.cfg_step(pred, lat, x_in, t, guidance_scale)  # guidance + x_t -> x_{t-1} + next UNet input

        z = torch.empty(B, h, w, cpad, dtype=torch.bfloat16, device=dev)
        scaled = lat * (1.0 / self.scaling_factor)
        _lib.call("sdt_nchw_f32_to_nhwc_bf16", scaled.data_ptr(), z.data_ptr(), B, C, h, w, cpad, stream)
        ops.gn_arena_begin(dev)
        img = nets.vae_decode(self.vae_decoder, self.vae_config, z)
        ops.gn_arena_end(dev)
        image = (img[..., : self.vae_config["in_channels"]].float() / 2 + 0.5).clamp(0, 1)
        return (image, lat) if return_latents else image
