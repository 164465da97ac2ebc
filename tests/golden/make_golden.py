"""Generates tests/golden/tiny_step.npz: inputs are seeded (tests/helpers.make_case) and the expected outputs come
from the CPU oracle (oracle/train_step.py).  The reference ships no golden vectors, so this fixture
pins the ORACLE against regressions and gives the GPU tests a second, frozen comparator; it is not reference output.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import train_step as ots  # noqa: E402
from tests.helpers import make_case  # noqa: E402

LEAVES = ["mid_block/resnets_0/conv1/kernel", "down_blocks_0/attentions_0/transformer_blocks_0/attn1/to_q/kernel",
          "conv_in/kernel", "conv_norm_out/scale"]


def main():
    out = {}
    for tag, pred, sched in (("eps", "epsilon", "scaled_linear"), ("v", "v_prediction", "zero_snr_scaled_linear")):
        case = make_case("tiny", B=2, image=64, sched=sched)
        r = ots.train_step(case["weights"]["unet"], case["weights"]["clip"], case["weights"]["vae"], case["sched_state"],
                           case["cfgs"], case["batch"], case["rand"], dict(ots.DEFAULT_OPT), prediction_type=pred)
        out[f"{tag}_loss"] = np.float64(r["loss"])
        out[f"{tag}_pred"] = r["aux"]["pred"].numpy()
        out[f"{tag}_latents"] = r["aux"]["latents"].numpy()
        out[f"{tag}_unet_gnorm"] = np.float64(r["unet_gnorm"])
        out[f"{tag}_te_gnorm"] = np.float64(r["te_gnorm"])
        for i, k in enumerate(LEAVES):
            out[f"{tag}_grad{i}"] = r["unet_grads"][k].numpy()
            out[f"{tag}_param{i}"] = r["unet_params"][k]
            m = r["unet_state"]["mu"][k]
            if isinstance(m, tuple):
                out[f"{tag}_codes{i}"], out[f"{tag}_inv{i}"] = m
            else:
                out[f"{tag}_mom{i}"] = m
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tiny_step.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tiny_sample.npz"), **sampling_case()[1])


def sampling_inputs():
    """Seeded inputs of the sampling fixture: (case, full VAE weights, prompt ids, negative ids, initial latents)."""
    import torch
    from oracle import nets as onets
    case = make_case("tiny", B=2, image=64)
    w_vae = dict(case["weights"]["vae"])
    w_vae.update(onets.init_params(onets.vae_decoder_param_shapes(case["cfgs"]["vae"]), 9))
    ids = case["batch"]["input_ids"]
    vocab = case["cfgs"]["clip"]["vocab_size"]
    neg = torch.full_like(ids, vocab - 1)
    neg[:, 0] = vocab - 2
    lat0 = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(4))
    return case, w_vae, ids, neg, lat0


def sampling_case(ptype="epsilon"):
    """Inputs and oracle outputs of a 4-step classifier-free-guidance DDIM run on the tiny configuration (sampling path)."""
    from oracle import sampling as osamp
    case, w_vae, ids, neg, lat0 = sampling_inputs()
    img, lat = osamp.generate(case["weights"]["unet"], case["weights"]["clip"], w_vae, case["cfgs"], case["sched_state"], ids, neg,
                              lat0, 4, 3.0, ptype)
    return (case, w_vae, ids, neg, lat0), {"image": img, "latents": lat}


if __name__ == "__main__":
    main()
