"""Developer probe: which host-side torch ops issue device-to-device copies / fills inside one eager train_step (SD1.5 size)."""
import collections
import sys
import traceback

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

dev = torch.device("cuda:0")
tc, cfgs, weights, (us, ts, ue, te, vae, sched, _) = bench.build_states(dev, 4)
from stable_diffusion_training_amd import training_utils as tu  # noqa: E402
batch = bench.synthetic_batch(dev, 4, 0)
rng = torch.Generator(device=dev)
kw = dict(strip_bos_eos_token=False, ema_rate=0.99998)
tu.train_step(us, ts, ue, te, batch, rng, vae, sched, **kw)
torch.cuda.synchronize()
counts = collections.Counter()
orig = {}


def wrap(name):
    f = getattr(torch.Tensor, name)
    orig[name] = f

    def g(self, *a, **k):
        r = f(self, *a, **k)
        if self.is_cuda and (name != "contiguous" or r is not self):
            fr = [x for x in traceback.extract_stack(limit=8) if "stable_diffusion_training_amd" in x.filename or "bench" in x.filename]
            where = f"{fr[-1].filename.split('/')[-1]}:{fr[-1].lineno}" if fr else "autograd/torch"
            counts[(name, where, tuple(self.shape)[-2:] if self.dim() >= 2 else tuple(self.shape))] += 1
        return r
    setattr(torch.Tensor, name, g)


for n in ("copy_", "clone", "contiguous", "zero_", "fill_", "to"):
    wrap(n)
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tu.train_step(us, ts, ue, te, batch, rng, vae, sched, **kw)
    torch.cuda.synchronize()
for n, f in orig.items():
    setattr(torch.Tensor, n, f)
print("python-level tensor copies / fills in one step:")
for k, v in counts.most_common(25):
    print("  ", v, k)
ev = collections.Counter()
for e in prof.events():
    if "Memcpy" in e.name or "Memset" in e.name or e.name.startswith("aten::copy_") or e.name.startswith("aten::fill_") or e.name.startswith("aten::zero_") or e.name.startswith("aten::add") or e.name.startswith("aten::cat") or e.name.startswith("aten::contiguous") or e.name.startswith("aten::clone"):
        ev[e.name] += 1
print("profiler event counts:", dict(ev))
