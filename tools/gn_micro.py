"""Developer micro-benchmark: GroupNorm(+SiLU) forward / backward per activation shape of the SD1.5 step (B=4),
device time per call (events over back-to-back launches) and effective HBM GB/s (fwd: 2 reads + 1 write of x;
bwd: 2 reads each of x and dy + 1 write of dx)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stable_diffusion_training_amd import _lib
dev = torch.device("cuda:0")
B, G = 4, 32
SHAPES = [(64, 1280), (64, 2560), (256, 640), (256, 1280), (256, 1920), (256, 2560), (1024, 320), (1024, 640), (1024, 960),
          (1024, 1280), (1024, 1920), (4096, 320), (4096, 640), (4096, 960), (4096, 512), (16384, 256), (16384, 512),
          (65536, 128), (65536, 256), (262144, 128)]
s = torch.cuda.current_stream().cuda_stream
def ev(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tf = tb = 0.0
for HW, C in SHAPES:
    x = torch.randn(B, HW, C, device=dev).bfloat16(); dy = torch.randn_like(x); y = torch.empty_like(x); dx = torch.empty_like(x)
    gamma = torch.randn(C, device=dev); beta = torch.randn(C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    stats = torch.empty(B, G, 2, device=dev); bstats = torch.empty(B, G, 2, device=dev)
    need = _lib.load().sdt_groupnorm_bwd_workspace_bytes(B, HW, C)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    needf = _lib.load().sdt_groupnorm_fwd_workspace_bytes(B, HW, C, G)
    wsf = torch.empty(max(needf, 1), dtype=torch.uint8, device=dev)
    parts = torch.zeros(B, 32, G, 2, device=dev)  # statistics as 32 partial rows per image (what a producer's epilogue leaves)
    f = lambda: _lib.call("sdt_groupnorm_fwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(), B, HW, C, G, 1e-5, 1, None, 0, wsf.data_ptr(), needf, s)
    b = lambda: _lib.call("sdt_groupnorm_bwd", x.data_ptr(), dy.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dx.data_ptr(),
                          dg.data_ptr(), db.data_ptr(), None, B, HW, C, G, 1e-5, 1, ws.data_ptr(), need, s)
    fa = lambda: _lib.call("sdt_groupnorm_fwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(), B, HW, C, G, 1e-5, 1, parts.data_ptr(), 32, None, 0, s)
    f()
    parts[:, 0].copy_(stats)
    t_a = ev(fa)  # apply only (statistics ready: what runs behind a convolution that accumulated them)
    t_f, t_b = ev(f), ev(b)
    nbytes = x.numel() * 2
    tf += t_f; tb += t_b
    print(f"HW={HW:6d} C={C:5d}  {nbytes/1e6:7.1f} MB   apply {t_a:7.1f} us {2*nbytes/t_a/1e3:7.0f} GB/s   fwd {t_f:7.1f} us {3*nbytes/t_f/1e3:7.0f} GB/s   bwd {t_b:7.1f} us {5*nbytes/t_b/1e3:7.0f} GB/s", flush=True)
print(f"sum fwd {tf:.0f} us  bwd {tb:.0f} us")
