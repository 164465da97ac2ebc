"""Generates tests/golden/tiny_step_hip.npz ON THE GPU BOX: outputs of the HIP path itself for the seeded tiny step (tests/helpers.
make_case).  Since round 3 the step is bitwise reproducible (no float atomics), so a frozen HIP output is a meaningful regression
fixture: a later build may differ from it only by bf16 re-rounding where a kernel's summation order legitimately changed (tile
shape, split plan) - orders of magnitude below what a wrong epsilon, a dropped term or a mis-indexed tile produces - and the gate
can sit 6x tighter than the one against the fp32 oracle, whose distance is the network's own bf16 rounding noise.
The script also prints the distances that justify the gates (HIP vs fp32 oracle, bf16-points oracle vs fp32 oracle).
Run (GPU box):  python tests/golden/make_hip_regression.py gpurun_out/tiny_step_hip.npz   then copy the file to tests/golden/."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import nets as onets  # noqa: E402
from oracle import train_step as ots  # noqa: E402
from stable_diffusion_training_amd import training_utils as tu  # noqa: E402
from tests.helpers import build_hip_states, make_case, rel_l2, to_dev  # noqa: E402

LEAVES = ["mid_block/resnets_0/conv1/kernel", "down_blocks_0/attentions_0/transformer_blocks_0/attn1/to_q/kernel",
          "down_blocks_0/attentions_0/transformer_blocks_0/ff/net_0/proj/kernel", "conv_in/kernel", "conv_norm_out/scale",
          "down_blocks_0/attentions_0/transformer_blocks_0/norm2/bias", "time_embedding/linear_2/bias"]
TEXT_LEAVES = ["text_model/encoder/layers/0/self_attn/q_proj/kernel", "text_model/final_layer_norm/scale",
               "text_model/embeddings/position_embedding/embedding"]


def main(out_path):
    dev = torch.device("cuda:0")
    out = {}
    for tag, pred, sched in (("eps", "epsilon", "scaled_linear"), ("v", "v_prediction", "zero_snr_scaled_linear")):
        case = make_case("tiny", B=2, image=64, sched=sched)
        tc, (us, ts, ue, te, vae, sc, _) = build_hip_states(case, dev, prediction_type=pred)
        aux = {}
        res = tu.train_step(us, ts, None, None, to_dev(case["batch"], dev), torch.Generator(device=dev), vae, sc,
                            strip_bos_eos_token=False, rand=to_dev(case["rand"], dev), aux=aux)
        torch.cuda.synchronize()
        out[f"{tag}_loss"] = np.float64(res[4]["loss"].item())
        out[f"{tag}_pred"] = aux["pred"][..., :4].permute(0, 3, 1, 2).float().cpu().numpy()
        out[f"{tag}_moments"] = aux["moments"].float().cpu().numpy()
        out[f"{tag}_ctx"] = aux["ctx"].float().cpu().numpy()
        out[f"{tag}_unet_gnorm"] = np.float64(us.store.grad_norm())
        out[f"{tag}_te_gnorm"] = np.float64(ts.store.grad_norm())
        g, gt = us.store.export("grad"), ts.store.export("grad")
        for i, k in enumerate(LEAVES):
            out[f"{tag}_grad{i}"] = g[k].cpu().numpy()
        for i, k in enumerate(TEXT_LEAVES):
            out[f"{tag}_tgrad{i}"] = gt[k].cpu().numpy()
        # the distances behind the gates
        with torch.no_grad():
            l32, a32 = ots.compute_loss(case["weights"]["unet"], case["weights"]["clip"], case["weights"]["vae"], case["sched_state"],
                                        case["cfgs"], case["batch"], case["rand"], prediction_type=pred, return_aux=True)
            with onets.bf16_points():
                l16, a16 = ots.compute_loss(case["weights"]["unet"], case["weights"]["clip"], case["weights"]["vae"], case["sched_state"],
                                            case["cfgs"], case["batch"], case["rand"], prediction_type=pred, return_aux=True)
        hip = torch.from_numpy(out[f"{tag}_pred"])
        print(f"{tag}: pred rel-L2  HIP vs fp32 oracle {rel_l2(hip, a32['pred']):.3e}   bf16-points oracle vs fp32 oracle "
              f"{rel_l2(a16['pred'], a32['pred']):.3e}   HIP vs bf16-points oracle {rel_l2(hip, a16['pred']):.3e};  loss HIP {out[f'{tag}_loss']:.6f} "
              f"fp32 {float(l32):.6f} bf16pts {float(l16):.6f}")
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, {k: getattr(v, "shape", v) for k, v in out.items() if "grad" not in k})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "tiny_step_hip.npz"))
