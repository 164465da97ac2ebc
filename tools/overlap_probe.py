"""Developer probe: do two chains of under-filled GEMM launches overlap when they are captured as two branches of one HIP graph?
A = n launches of (M,N,K) on one stream; B = n launches of a second shape on another; times of A, B, A then B, A beside B."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stable_diffusion_training_amd import ops

dev = torch.device("cuda:0")
BF = torch.bfloat16


def mk(M, N, K):
    return (torch.randn(M, K, device=dev).to(BF), torch.randn(N, K, device=dev).to(BF), torch.empty(M, N, device=dev, dtype=BF), M, N, K)


def chain(p, n):
    A, Bt, C, M, N, K = p
    for _ in range(n):
        ops.gemm_nt(A, Bt, C, M, N, K, 1, K, K, 0)


def timed(fn, label):
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        g.capture_begin()
        s0.wait_stream(side); s1.wait_stream(side)
        fn(s0, s1)
        side.wait_stream(s0); side.wait_stream(s1)
        g.capture_end()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{label:40s} {e0.elapsed_time(e1) * 100:9.1f} us", flush=True)


for (sa, sb, n) in (((4096, 640, 640), (4096, 640, 640), 20), ((4096, 640, 640), (16384, 2560, 320), 20), ((1024, 1280, 1280), (1024, 1280, 1280), 20),
                    ((308, 768, 768), (308, 3072, 768), 20), ((16384, 320, 320), (4096, 640, 640), 20)):
    pa, pb = mk(*sa), mk(*sb)
    chain(pa, 1); chain(pb, 1)
    print(sa, sb)

    def onlyA(s0, s1):
        with torch.cuda.stream(s0):
            chain(pa, n)

    def onlyB(s0, s1):
        with torch.cuda.stream(s0):
            chain(pb, n)

    def seq(s0, s1):
        with torch.cuda.stream(s0):
            chain(pa, n); chain(pb, n)

    def par(s0, s1):
        with torch.cuda.stream(s0):
            chain(pa, n)
        with torch.cuda.stream(s1):
            chain(pb, n)

    timed(onlyA, "A"); timed(onlyB, "B"); timed(seq, "A then B (one stream)"); timed(par, "A beside B (two branches)")
