"""Developer micro-benchmark: 3x3 stride-1 convolution forward per shape of the SD1.5 step (B=4); SDT_CONV_HALO=0/1 selects
the generic gather kernel or the halo-staged kernel (read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stable_diffusion_training_amd import ops, nets, params
dev = torch.device("cuda:0")
if os.environ.get("CONV_MICRO_FEW"):
    FEW = True
else:
    FEW = False
SHAPES = [(4, 512, 512, 128, 128), (4, 256, 256, 256, 256), (4, 128, 128, 512, 512), (4, 64, 64, 512, 512),
          (4, 64, 64, 320, 320), (4, 64, 64, 640, 320), (4, 64, 64, 960, 320), (4, 32, 32, 640, 640), (4, 32, 32, 1280, 640),
          (4, 16, 16, 1280, 1280), (4, 16, 16, 2560, 1280), (4, 8, 8, 1280, 1280), (4, 8, 8, 2560, 1280)]
if FEW:
    SHAPES = [SHAPES[0], SHAPES[2], SHAPES[4]]
def ev(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = 0.0
for B, H, W, Ci, Co in SHAPES:
    spec = [("c/kernel", (3, 3, Ci, Co)), ("c/bias", (Co,))]
    st = params.ParamStore(spec, device=dev, quantise=False, trainable=False)
    st.load(nets.init_params(spec, 0)); st.prepare()
    x = torch.randn(B, H, W, Ci, device=dev).bfloat16()
    with torch.no_grad():
        t = ev(lambda: ops.conv2d(x, st, "c"))
    fl = 2.0 * B * H * W * Ci * Co * 9
    tot += t
    print(f"B{B} {H:3d}x{W:3d} {Ci:4d}->{Co:4d}  {t:8.1f} us  {fl/t/1e6:7.1f} TF", flush=True)
    del st, x
print(f"sum {tot:.0f} us")
