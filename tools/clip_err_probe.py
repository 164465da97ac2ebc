"""Developer probe: CLIP text tower error against the fp32 oracle by depth (0 = embeddings + final norm only)."""
import sys

import torch

sys.path.insert(0, ".")
from oracle import nets as onets  # noqa: E402
from tests.helpers import rel_l2  # noqa: E402
from stable_diffusion_training_amd import nets, params  # noqa: E402

dev = torch.device("cuda:0")
for L in (0, 1, 2, 4, 12):
    for scale in (1.0, 10.0):
        cfg = dict(onets.clip_config("clip_l"), num_hidden_layers=L, vocab_size=1000)
        w = onets.init_params(onets.clip_param_shapes(cfg), 7)
        for k in w:
            if k.endswith("embedding"):
                w[k] = w[k] * scale
        ids = torch.randint(0, 1000, (3, 77), generator=torch.Generator().manual_seed(1))
        st = params.ParamStore(nets.clip_text_spec(cfg), device=dev, quantise=False, trainable=True)
        st.load(w)
        y = nets.clip_text_forward(st, cfg, ids.to(dev).to(torch.int32))
        with torch.no_grad():
            yr = onets.clip_text_forward(w, cfg, ids)
            with onets.bf16_points():
                yb = onets.clip_text_forward(w, cfg, ids)
        print(f"layers {L:2d} embedding scale {scale:4.1f}: HIP vs fp32 {rel_l2(y, yr):.2e}  HIP vs bf16-points {rel_l2(y, yb):.2e}  bf16-points vs fp32 {rel_l2(yb, yr):.2e}", flush=True)
