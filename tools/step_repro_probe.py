"""Developer probe: fresh states per run (as a new process / checkpoint load would have), identical inputs: report the first
operator of the train_step whose output differs between runs by more than rounding noise."""
import sys

import torch

sys.path.insert(0, ".")
from tests.helpers import build_hip_states, make_case, rel_l2, to_dev  # noqa: E402
from stable_diffusion_training_amd import ops  # noqa: E402
from stable_diffusion_training_amd import training_utils as tu  # noqa: E402

dev = torch.device("cuda:0")
size = sys.argv[1] if len(sys.argv) > 1 else "tiny"
case = make_case(size, B=2, image=64)
log = []
names = ["conv2d", "group_norm", "linear", "gemm_nt", "layer_norm", "attention_packed", "attention"]
orig = {n: getattr(ops, n) for n in names}


def wrap(n):
    def f(*a, **k):
        r = orig[n](*a, **k)
        if n == "gemm_nt":
            log.append((n + str(a[3:7]), a[2].detach().clone()))
        else:
            t = r[0] if isinstance(r, tuple) else r
            tag = a[2] if len(a) > 2 and isinstance(a[2], str) else ""
            log.append((n + ":" + tag, t.detach().clone()))
            if isinstance(r, tuple) and r[1] is not None and n in ("conv2d", "linear"):
                log.append((n + ":" + tag + ":stats", r[1].detach().clone()))
        return r
    return f


for n in names:
    setattr(ops, n, wrap(n))
runs = []
for it in range(4):
    log.clear()
    tc, (us, ts, ue, te, vae, sc, _) = build_hip_states(case, dev)
    tu.train_step(us, ts, None, None, to_dev(case["batch"], dev), torch.Generator(device=dev), vae, sc,
                  strip_bos_eos_token=False, rand=to_dev(case["rand"], dev))
    torch.cuda.synchronize()
    runs.append(list(log))
    if it:
        n_bad = 0
        for (n0, t0), (n1, t1) in zip(runs[0], runs[it]):
            e = rel_l2(t1, t0) if float(t0.float().norm()) > 0 else float(t1.float().norm())
            if e > 1e-4:
                print(f"run {it}: op #{n_bad} differing by > 1e-4: {n1}: rel {e:.3e} shape {tuple(t0.shape)}", flush=True)
                n_bad += 1
                if n_bad >= 3:
                    break
        if not n_bad:
            print(f"run {it}: identical to 1e-4 ({len(log)} ops)", flush=True)
