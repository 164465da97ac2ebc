"""Developer probe: eps-prediction error against the fp32 oracle and run-to-run spread of the tiny train_step with the fused
GroupNorm statistics (GEMM epilogue atomics) on and off."""
import sys

import torch

sys.path.insert(0, ".")
from oracle import train_step as ots  # noqa: E402
from tests.helpers import build_hip_states, make_case, rel_l2, to_dev  # noqa: E402
from stable_diffusion_training_amd import ops  # noqa: E402
from stable_diffusion_training_amd import training_utils as tu  # noqa: E402

dev = torch.device("cuda:0")
size = sys.argv[1] if len(sys.argv) > 1 else "tiny"
case = make_case(size, B=2, image=64)
ref = ots.train_step(case["weights"]["unet"], case["weights"]["clip"], case["weights"]["vae"], case["sched_state"],
                     case["cfgs"], case["batch"], case["rand"], dict(ots.DEFAULT_OPT))
orig = ops._gn_parts
for fused in (True, False, True, False):
    ops._gn_parts = orig if fused else (lambda *a, **k: 0)
    preds = []
    for it in range(3):
        tc, (us, ts, ue, te, vae, sc, _) = build_hip_states(case, dev)
        aux = {}
        tu.train_step(us, ts, None, None, to_dev(case["batch"], dev), torch.Generator(device=dev), vae, sc,
                      strip_bos_eos_token=False, rand=to_dev(case["rand"], dev), aux=aux)
        preds.append((aux["pred"][..., :4].permute(0, 3, 1, 2).clone(), aux["moments"].clone()))
    e = [rel_l2(p, ref["aux"]["pred"]) for p, _ in preds]
    em = [rel_l2(m, ref["aux"]["moments"]) for _, m in preds]
    spread = [rel_l2(preds[i][0], preds[0][0]) for i in (1, 2)]
    print(f"fused={fused}: pred vs oracle {['%.2e' % x for x in e]}  moments vs oracle {['%.2e' % x for x in em]}  run-to-run {['%.2e' % x for x in spread]}", flush=True)
