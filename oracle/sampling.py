"""Oracle (test infrastructure): fp32 CPU restatement of the reference's sampling loop,
models/pipeline_flax_stable_diffusion.py:160-254 (`_generate`): text embeddings for the prompt and the negative prompt,
classifier-free guidance over a doubled batch (:212-228), DDIM steps (:231), latents / scaling_factor -> VAE decode ->
(image / 2 + 0.5).clip(0, 1) in NHWC (:245-252).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this package; the product path never does."""
import numpy as np
import torch

from . import nets
from . import schedulers as osched


def generate(unet_p, te_p, vae_p, cfgs, sched_state, prompt_ids, neg_prompt_ids, latents_nchw, num_inference_steps,
             guidance_scale, prediction_type="epsilon", scaling_factor=0.18215):
    """prompt_ids / neg_prompt_ids int (B,77); latents_nchw float32 (B,4,h,w) initial noise (init_noise_sigma = 1).
    Returns (image NHWC float32 in [0,1], final latents NCHW float32)."""
    with torch.no_grad():
        pe = nets.clip_text_forward(te_p, cfgs["clip"], torch.as_tensor(prompt_ids).long())
        ne = nets.clip_text_forward(te_p, cfgs["clip"], torch.as_tensor(neg_prompt_ids).long())
        ctx = torch.cat([ne, pe])  # :191
        lat = torch.as_tensor(latents_nchw, dtype=torch.float32).clone()
        for t in osched.ddim_timesteps(num_inference_steps):
            x2 = torch.cat([lat, lat])
            ts = torch.full((x2.shape[0],), int(t), dtype=torch.int64)
            out = nets.unet_forward(unet_p, cfgs["unet"], x2, ts, ctx)
            un, tx = out.chunk(2)
            guided = un + guidance_scale * (tx - un)
            lat = torch.from_numpy(osched.ddim_step(sched_state, guided.numpy(), int(t), lat.numpy(), num_inference_steps,
                                                    prediction_type))
        img = nets.vae_decode(vae_p, cfgs["vae"], (lat / scaling_factor).permute(0, 2, 3, 1))
        return (img / 2 + 0.5).clamp(0, 1).numpy().astype(np.float32), lat.numpy()
