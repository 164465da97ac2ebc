"""Developer tool: host (enqueue) time per EAGER train_step vs GPU time, plus a cProfile of one step's host side."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from stable_diffusion_training_amd import training_utils as tu
dev = torch.device("cuda:0")
tc, cfgs, weights, (us, ts, ue, te, vae, sched, _) = bench.build_states(dev, 4)
table = tu.dp_compile_all_unique_resolution(us, ts, ue, te, vae, sched, tc, per_device_batch=4, use_graph=False)
batch = bench.synthetic_batch(dev, 4, 0)
fn = table[tuple(batch["pixel_values"].shape)]
rng = torch.Generator(device=dev); rng.manual_seed(0)
for _ in range(2):
    fn(us, ts, ue, te, batch, rng, vae, sched)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    fn(us, ts, ue, te, batch, rng, vae, sched)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
fn(us, ts, ue, te, batch, rng, vae, sched)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
