"""Developer probe: how much faster do the Linear weight gradients of one transformer block level run when their launches
overlap (round-robin on 8 streams, a workspace each) than back to back on one stream?  Upper bound for what a grouped launch
of deferred weight gradients could gain."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stable_diffusion_training_amd import ops

dev = torch.device("cuda:0")
BF = torch.bfloat16
# (M, K1, N): SD1.5 batch 4: level 0 (16384 rows, 320), level 1 (4096, 640), level 2 (1024, 1280), text tower (308, 768)
shapes = [(16384, 320, 960), (16384, 320, 320), (16384, 320, 320), (16384, 320, 2560), (16384, 1280, 320), (16384, 320, 320),
          (4096, 640, 1920), (4096, 640, 640), (4096, 640, 640), (4096, 640, 5120), (4096, 2560, 640), (4096, 640, 640),
          (1024, 1280, 3840), (1024, 1280, 1280), (1024, 1280, 10240), (1024, 5120, 1280),
          (308, 768, 2304), (308, 768, 768), (308, 768, 3072), (308, 3072, 768)]
probs = []
for (M, K, N) in shapes:
    probs.append((torch.randn(M, K, device=dev).to(BF), torch.randn(M, N, device=dev).to(BF), torch.zeros(K, N, device=dev), M, K, N))
streams = [torch.cuda.Stream() for _ in range(8)]
ws = {}
orig = ops._tn_workspace


def per_stream_ws(need, device):
    k = torch.cuda.current_stream().cuda_stream
    if k not in ws or ws[k].numel() < need:
        ws[k] = torch.zeros(max(need, 64 << 20), dtype=torch.uint8, device=device)
    return ws[k]


ops._tn_workspace = per_stream_ws


def run(concurrent):
    for i, (x, dy, dw, M, K, N) in enumerate(probs):
        s = streams[i % 8] if concurrent else streams[0]
        with torch.cuda.stream(s):
            ops.gemm_tn(x, dy, dw, M, K, N, K, N, 1, K, N)


for mode in (False, True, False, True):
    run(mode); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    # graph capture removes host launch overhead from the comparison
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g.capture_begin()
        if mode:
            for s in streams:
                s.wait_stream(side)
        run_stream0 = streams[0]
        if not mode:
            streams[0].wait_stream(side)
        run(mode)
        for s in (streams if mode else streams[:1]):
            side.wait_stream(s)
        g.capture_end()
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{'8 streams' if mode else '1 stream '}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us for {len(probs)} weight gradients", flush=True)
