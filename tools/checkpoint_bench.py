"""Developer tool: wall time and GB/s of the checkpoint paths at SD1.5 size (UNet 860M + CLIP-L 123M parameters + VAE encoder):
save_model / load_models (diffusers Flax layout, fp32 msgpack) and save_training_state / load_training_state (safetensors)."""
import os
import shutil
import sys
import tempfile
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from stable_diffusion_training_amd import training_utils as tu

dev = torch.device("cuda", 0)
tc, cfgs, weights, (us, ts, ue, te, vae, sched, objs) = bench.build_states(dev, 4)
root = tempfile.mkdtemp(prefix="sdt_ckpt_", dir=os.environ.get("SDT_CKPT_DIR", "/tmp"))


def du(path):
    if os.path.isfile(path):
        return os.path.getsize(path)
    return sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(path) for f in fs)


def timed(name, fn, path):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gb = du(path) / 1e9
    print(f"{name:22s} {dt:7.2f} s  {gb:6.2f} GB  {gb / dt:5.2f} GB/s", flush=True)
    return r


try:
    out = os.path.join(root, "model@1")
    state = os.path.join(root, "state.safetensors")
    timed("save_model", lambda: tu.save_model(objs, None, us.params, ts.params, weights["vae"], out), out)
    timed("save_training_state", lambda: tu.save_training_state(state, us, ts, torch.Generator(device=dev)), state)
    m = timed("load_models", lambda: tu.load_models(types.SimpleNamespace(model_path=out)), out)
    assert torch.equal(m["unet"]["unet_params"]["conv_in/kernel"].to(dev), us.store.p("conv_in/kernel"))
    timed("load_training_state", lambda: tu.load_training_state(state, us, ts, torch.Generator(device=dev)), state)
finally:
    shutil.rmtree(root, ignore_errors=True)
