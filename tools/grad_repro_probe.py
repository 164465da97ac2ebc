"""Developer probe: run the same tiny train_step several times from fresh states and report, per gradient leaf, how much the
gradients differ from the first run (run-to-run reproducibility; poison = what the gradient buffer held before the step)."""
import sys

import torch

sys.path.insert(0, ".")
from tests.helpers import build_hip_states, make_case, rel_l2, to_dev  # noqa: E402
from stable_diffusion_training_amd import training_utils as tu  # noqa: E402

dev = torch.device("cuda:0")
size = sys.argv[1] if len(sys.argv) > 1 else "tiny"
case = make_case(size, B=2, image=64)
ref = None
for poison in (0.0, 0.0, float("nan"), 1e30, 0.0):
    tc, (us, ts, ue, te, vae, sc, _) = build_hip_states(case, dev)
    us.store.grad.fill_(poison)
    ts.store.grad.fill_(poison)
    aux = {}
    tu.train_step(us, ts, None, None, to_dev(case["batch"], dev), torch.Generator(device=dev), vae, sc,
                  strip_bos_eos_token=False, rand=to_dev(case["rand"], dev), aux=aux)
    torch.cuda.synchronize()
    g = {p: us.store.grad[lf.offset: lf.offset + lf.numel].clone() for p, lf in us.store.leaves.items()}
    pred = aux["pred"].clone()
    keep = {k: aux[k].clone() for k in ("moments", "latents", "noisy", "ctx", "target")}
    if ref is None:
        ref = (g, pred, keep)
        continue
    print("   stage diffs vs run 0: " + ", ".join(f"{k} {rel_l2(keep[k], ref[2][k]):.2e}" for k in keep), flush=True)
    worst = sorted(((rel_l2(g[p], ref[0][p]), p) for p in g if float(ref[0][p].norm()) > 0), reverse=True)[:4]
    print(f"poison {poison}: pred diff {rel_l2(pred, ref[1]):.3e}; worst leaves: " + ", ".join(f"{p} {e:.2e}" for e, p in worst), flush=True)
