"""Developer tool: ms/step of the SD1.5 train step at another aspect bucket than the headline 512x512 (batch 4, graph replay).
usage: python tools/bucket_bench.py HEIGHT WIDTH [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from stable_diffusion_training_amd import training_utils as tu

H, W = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda", 0)
tc, cfgs, weights, (us, ts, ue, te, vae, sched, _) = bench.build_states(dev, 4)
kw = dict(strip_bos_eos_token=False, ema_rate=tc.ema_rate)
step = tu._GraphedStep(lambda *a, **k: tu.train_step(*a, **kw, **k))
batch = bench.synthetic_batch(dev, 4, 0)
g = torch.Generator().manual_seed(5)
batch["pixel_values"] = (torch.rand(4, 3, H, W, generator=g) * 2 - 1).to(dev)
rng = torch.Generator(device=dev)
rng.manual_seed(2)
for _ in range(4):
    out = step(us, ts, ue, te, batch, rng, vae, sched)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = step(us, ts, ue, te, batch, rng, vae, sched)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{H}x{W}: {1e3 * dt:.2f} ms/step, {4 / dt:.1f} images/sec, loss {float(out[4]['loss']):.4f}  (SDT_CONV_HALO={os.environ.get('SDT_CONV_HALO', '1')})", flush=True)
