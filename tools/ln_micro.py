"""Developer micro-benchmark: LayerNorm forward / backward per activation shape of the SD1.5 / SDXL steps, device time per call
(events over back-to-back launches) and effective GB/s (fwd: read x, write y; bwd: read x, dy, dres, write dx)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stable_diffusion_training_amd import _lib
dev = torch.device("cuda:0")
SHAPES = [(16384, 320), (4096, 640), (1024, 1280), (256, 1280), (308, 768), (2048, 1280), (8192, 640)]
s = torch.cuda.current_stream().cuda_stream
def ev(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, C in SHAPES:
    x = torch.randn(M, C, device=dev).bfloat16(); dy = torch.randn_like(x); y = torch.empty_like(x); dx = torch.empty_like(x); dres = torch.randn_like(x)
    gamma = torch.randn(C, device=dev); beta = torch.randn(C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    mr = torch.empty(M, 2, device=dev)
    need = _lib.load().sdt_layernorm_bwd_workspace_bytes(M, C)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    f = lambda: _lib.call("sdt_layernorm_fwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mr.data_ptr(), M, C, 1e-5, s)
    b = lambda: _lib.call("sdt_layernorm_bwd", x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), mr.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                          dres.data_ptr(), M, C, 0, ws.data_ptr(), need, s)
    b0 = lambda: _lib.call("sdt_layernorm_bwd", x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), mr.data_ptr(), dx.data_ptr(), None, None,
                           dres.data_ptr(), M, C, 0, None, 0, s)
    t_f, t_b, t_b0 = ev(f), ev(b), ev(b0)
    nb = x.numel() * 2
    print(f"M={M:6d} C={C:5d} {nb/1e6:6.1f} MB  fwd {t_f:6.1f} us {2*nb/t_f/1e3:6.0f} GB/s   bwd {t_b:6.1f} us {4*nb/t_b/1e3:6.0f} GB/s   bwd w/o dgamma {t_b0:6.1f} us", flush=True)
