"""Generates tests/golden/clip_pin_*.npz: outputs of the `transformers` PyTorch CLIPTextModel, a THIRD-PARTY implementation of the
text tower the reference calls (transformers FlaxCLIPTextModel, reference training_utils.py:14, 215-217, 635-640; same architecture
and same checkpoints as the PyTorch class).  These vectors pin oracle.nets.clip_text_forward to something the build did not write
(tests/test_oracle_clip_pin.py).  Runs in the BUILD container only (transformers 5.15.0 is installed there); nothing of
`transformers` is imported by tests, by the package or on the GPU box - only the .npz files travel.

Cases
  small_quick_gelu / small_gelu : 2-layer, 32-wide towers; every weight comes from transformers' own initialiser and is stored in
                                  the fixture under the Flax names / layouts (Dense kernel [in,out] = weight.T); the output and
                                  the gradient of EVERY leaf under a fixed cotangent are stored.
  clip_l / openclip_h           : the full CLIP-L (SD1.5) and OpenCLIP-H (SD2.1, 23 layers, erf-GELU) configurations; the weights are
                                  oracle.nets.init_params(seed) (seeded, not stored), loaded into the transformers model; the
                                  output for one 77-token sequence and the gradients of a few small leaves are stored.
Run:  python tests/golden/make_clip_pin.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import nets as onets  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SMALL = dict(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4,
             max_position_embeddings=77, layer_norm_eps=1e-5)
FULL_GRAD_LEAVES = ["text_model/final_layer_norm/scale", "text_model/final_layer_norm/bias",
                    "text_model/encoder/layers/0/self_attn/q_proj/bias", "text_model/encoder/layers/0/layer_norm1/scale",
                    "text_model/encoder/layers/5/mlp/fc2/bias", "text_model/encoder/layers/11/self_attn/out_proj/bias", "text_model/embeddings/position_embedding/embedding"]


def hf_model(cfg):
    from transformers import CLIPTextConfig, CLIPTextModel
    c = CLIPTextConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                       num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
                       max_position_embeddings=cfg["max_position_embeddings"], hidden_act=cfg["hidden_act"],
                       layer_norm_eps=cfg["layer_norm_eps"], bos_token_id=cfg["vocab_size"] - 2, eos_token_id=cfg["vocab_size"] - 1,
                       pad_token_id=cfg["vocab_size"] - 1, projection_dim=cfg["hidden_size"])
    return CLIPTextModel(c).float().eval()


def hf_name(flax_path):
    """Flax leaf path -> (state_dict key of the PyTorch model, transpose?)."""
    p = flax_path.replace("/", ".")
    if p.endswith(".embedding"):
        return p[: -len("embedding")] + "weight", False
    if p.endswith(".kernel"):
        return p[: -len("kernel")] + "weight", True
    if p.endswith(".scale"):
        return p[: -len("scale")] + "weight", False
    return p, False


def _key(d, key):
    """transformers 5.x drops the `text_model.` level of CLIPTextModel's parameter names; 4.x keeps it."""
    return key if key in d else key[len("text_model."):]


def flax_tree_of(model, cfg):
    sd = model.state_dict()
    out = {}
    for path in onets.clip_param_shapes(cfg):
        key, tr = hf_name(path)
        key = _key(sd, key)
        w = sd[key].detach().clone()
        out[path] = w.t().contiguous() if tr else w
    return out


def load_flax_tree(model, tree):
    sd = model.state_dict()
    for path, w in tree.items():
        key, tr = hf_name(path)
        sd[_key(sd, key)].copy_(w.t() if tr else w)


def run(model, ids, cot):
    model.zero_grad()
    hs = model(input_ids=ids).last_hidden_state
    (hs * cot).sum().backward()
    return hs.detach(), None, None


def grads_of(model, cfg, leaves):
    named = dict(model.named_parameters())
    out = {}
    for path in leaves:
        key, tr = hf_name(path)
        g = named[_key(named, key)].grad
        out[path] = (g.t() if tr else g).contiguous().numpy()
    return out


def main():
    torch.manual_seed(20261004)
    for act in ("quick_gelu", "gelu"):
        cfg = dict(SMALL, hidden_act=act)
        m = hf_model(cfg)
        with torch.no_grad():  # transformers initialises biases to zero and norm scales to one: perturb so that they are pinned too
            g = torch.Generator().manual_seed(11)
            for n, p in m.named_parameters():
                if n.endswith("bias"):
                    p.add_(0.05 * torch.randn(p.shape, generator=g))
                elif "layer_norm" in n and n.endswith("weight"):
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
                elif n.endswith("weight") and p.dim() == 2 and "embedding" not in n:
                    p.mul_(3.0)  # default std 0.02-ish leaves the attention logits ~0: make the softmax matter
        tree = flax_tree_of(m, cfg)
        g = torch.Generator().manual_seed(5)
        ids = torch.randint(0, cfg["vocab_size"] - 2, (3, 77), generator=g)
        ids[:, 0] = cfg["vocab_size"] - 2
        ids[:, -1] = cfg["vocab_size"] - 1
        ids[1, 40:] = cfg["vocab_size"] - 1  # a padded caption: the step passes no attention_mask (training_utils.py:635-640)
        cot = torch.randn(3, 77, cfg["hidden_size"], generator=g)
        hs, _, _ = run(m, ids, cot)
        out = {"ids": ids.numpy().astype(np.int32), "cot": cot.numpy(), "last_hidden_state": hs.numpy(),
               "cfg_hidden_act": np.array(act)}
        for k, v in tree.items():
            out["w:" + k] = v.numpy()
        for k, v in grads_of(m, cfg, list(tree)).items():
            out["g:" + k] = v
        np.savez_compressed(os.path.join(OUT, f"clip_pin_small_{act}.npz"), **out)
        print(f"small {act}: |hs| {hs.norm():.4f}, {len(tree)} leaves")

    for name, seed in (("clip_l", 3), ("openclip_h", 4)):
        cfg = onets.clip_config(name)
        tree = onets.init_params(onets.clip_param_shapes(cfg), seed)
        m = hf_model(cfg)
        with torch.no_grad():
            load_flax_tree(m, tree)
        g = torch.Generator().manual_seed(6)
        ids = torch.randint(0, 49406, (1, 77), generator=g)
        ids[:, 0] = 49406
        ids[:, -1] = 49407
        cot = torch.randn(1, 77, cfg["hidden_size"], generator=g)
        hs, _, _ = run(m, ids, cot)
        out = {"ids": ids.numpy().astype(np.int32), "cot_seed": np.int64(6), "seed": np.int64(seed),
               "last_hidden_state": hs.numpy()}
        for k, v in grads_of(m, cfg, FULL_GRAD_LEAVES).items():
            out["g:" + k] = v
        np.savez_compressed(os.path.join(OUT, f"clip_pin_{name}.npz"), **out)
        print(f"{name}: |hs| {hs.norm():.4f}")


if __name__ == "__main__":
    main()
