"""Developer tool: device time and HBM rate of sdt_lion8_step on a UNet-sized flat buffer (HIP events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ctypes

from stable_diffusion_training_amd import _lib

_lib.require_device()
if len(sys.argv) > 1:  # A/B: time another build of the library (same signatures)
    alt = ctypes.CDLL(sys.argv[1])
    alt.sdt_lion8_step.argtypes = _lib.SIGNATURES["sdt_lion8_step"]
    alt.sdt_lion8_step.restype = ctypes.c_int
dev = torch.device("cuda", 0)
n = 832 * 1024 * 1024  # ~UNet's quantised + decayed segment
bs = 16
g = torch.Generator(device=dev)
g.manual_seed(0)
p = torch.randn(n, device=dev, generator=g)
gr = torch.randn(n, device=dev, generator=g) * 1e-3
codes = torch.randint(-100, 100, (n,), device=dev, dtype=torch.int8, generator=g)
inv = torch.rand(n // bs, device=dev, generator=g) * 1e3 + 1
ema = p.clone()
wbf = torch.empty(n, dtype=torch.bfloat16, device=dev)
from stable_diffusion_training_amd import params as _params
thr = _params.lion_thresholds(dev)
sq = torch.tensor([float(gr.double().pow(2).sum())], dtype=torch.float64, device=dev)
s = torch.cuda.current_stream().cuda_stream


def step():
    if len(sys.argv) > 1:
        rc = alt.sdt_lion8_step(p.data_ptr(), gr.data_ptr(), codes.data_ptr(), inv.data_ptr(), ema.data_ptr(), wbf.data_ptr(), n, bs,
                                sq.data_ptr(), thr.data_ptr(), 1.0, 1e-6 / 7, 7e-2, 0.9, 0.99, 0.9999, s)
        assert rc == 0
        return
    _lib.call("sdt_lion8_step", p.data_ptr(), gr.data_ptr(), codes.data_ptr(), inv.data_ptr(), ema.data_ptr(), wbf.data_ptr(), n, bs,
              sq.data_ptr(), thr.data_ptr(), 1.0, 1e-6 / 7, 7e-2, 0.9, 0.99, 0.9999, s)


for _ in range(3):
    step()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in ev:
    a.record()
    step()
    b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
byts = n * (4 + 8 + 2 + 8 + 8 / bs)  # grad r, master r+w, codes r+w, ema r+w, scales r+w
print(f"lion8 {n / 1e6:.0f}M params: median {ms[5]:.3f} ms, min {ms[0]:.3f} ms -> {byts / ms[5] / 1e9:.2f} TB/s (median)")
