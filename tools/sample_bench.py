"""Developer tool: wall time of the sampling path at SD1.5 size (512x512, classifier-free guidance, DDIM), random-init weights."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from stable_diffusion_training_amd import nets
from stable_diffusion_training_amd.pipeline import StableDiffusionPipeline
from stable_diffusion_training_amd.schedulers import DDIMScheduler

dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tc, cfgs, weights, (us, ts, ue, te, vae, sched, objs) = bench.build_states(dev, 4, ema=False)
w_vae = dict(weights["vae"])
w_vae.update(nets.init_params(nets.vae_decoder_spec(cfgs["vae"]), 5))
pipe = StableDiffusionPipeline(us, ts, w_vae, cfgs["unet"], cfgs["clip"], cfgs["vae"],
                               scheduler=DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear"))
for B in (1, 4):
    ids = bench.synthetic_batch(dev, B, 0)["input_ids"]
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    pipe.generate(ids, num_inference_steps=2, generator=g)  # warm-up: workspaces, kernel attributes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = pipe.generate(ids, num_inference_steps=steps, generator=g)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"batch {B}: {steps} DDIM steps + decode  {dt:.3f} s  ({dt / B:.3f} s/image, {1e3 * dt / steps:.1f} ms/step incl. decode)  "
          f"image {tuple(img.shape)} mean {float(img.mean()):.3f}", flush=True)
