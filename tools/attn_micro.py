"""Developer micro-benchmark for the attention kernels (device-side times via events over many launches)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stable_diffusion_training_amd import ops
dev = torch.device("cuda:0")
B, H, N, D = 4, 8, 4096, 40
if len(sys.argv) > 1: N = int(sys.argv[1])
if len(sys.argv) > 2: D = int(sys.argv[2])
scale_in = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
C = H * D
q = (torch.randn(B, N, C, device=dev) * scale_in).bfloat16().requires_grad_(True)
k = (torch.randn(B, N, C, device=dev) * scale_in).bfloat16().requires_grad_(True)
v = torch.randn(B, N, C, device=dev).bfloat16().requires_grad_(True)
do = torch.randn(B, N, C, device=dev).bfloat16()
def ev(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
fl = 4.0 * B * H * N * N * D
with torch.no_grad():
    t = ev(lambda: ops.attention(q, k, v, H, D ** -0.5))
print(f"fwd  {t:8.1f} us  {fl/t/1e6:7.1f} TF (algorithmic)")
def fb():
    o = ops.attention(q, k, v, H, D ** -0.5)
    o.backward(do)
t2 = ev(fb)
print(f"fwd+bwd {t2:8.1f} us  bwd {t2-t:8.1f} us  {2.5*fl/(t2-t)/1e6:7.1f} TF (algorithmic 2.5x)")
