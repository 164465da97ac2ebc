"""Developer tool: forward-attention ablations in ONE process (needs a -DSDT_ATTN_DBG build of the library).
bits: 1 no next-tile DMA, 2 no exp, 4 no P.V MFMAs, 8 no Q.K MFMAs, 16 no barrier/wait."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stable_diffusion_training_amd import ops
dev = torch.device("cuda:0")
B, H, N, D = 4, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 40
q, k, v = (torch.randn(B, N, H * D, device=dev).bfloat16() for _ in range(3))
def ev(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for f in (0, 1, 2, 4, 8, 16, 6, 12, 14, 15, 17, 31, 0):
    os.environ["SDT_ATTN_DBG"] = str(f)
    with torch.no_grad():
        t = ev(lambda: ops.attention(q, k, v, H, D ** -0.5))
    print(f"dbg={f:2d}  fwd {t:8.1f} us", flush=True)
