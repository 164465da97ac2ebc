"""Developer probe (needs a -DSDT_NT_DBG build: SDT_LIB=...libsdtrain_hip_dbg.so): do the staging phase and the epilogue phase of a
short-reduction GEMM use the same hardware resource?  The (16384, 2560, 320) GEMM runs as 'staging only' (no MFMAs, no epilogue) and as
'epilogue only' (no staging, no MFMAs); each chain alone, one after the other, and side by side as two branches of one HIP graph."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stable_diffusion_training_amd import _lib, ops

dev = torch.device("cuda:0")
BF = torch.bfloat16
lib = _lib.load()
setdbg = lib.sdt_dbg_set_nt
setdbg.argtypes = [ctypes.c_int]
setdbg.restype = None
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16384, 2560, 320)))
A = torch.randn(M, K, device=dev).to(BF)
W = torch.randn(K, N, device=dev).to(BF)
C1 = torch.empty(M, N, device=dev, dtype=BF)
C2 = torch.empty(M, N, device=dev, dtype=BF)


def chain(bits, C, n=10):
    setdbg(bits)
    for _ in range(n):
        ops.gemm_nt(A, W, C, M, N, K, 1, K, N, 0, b_kmajor=True)
    setdbg(0)


def timed(fn, label):
    s0, s1, side = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
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
    print(f"{label:46s} {e0.elapsed_time(e1) * 10:9.1f} us per launch pair", flush=True)


chain(0, C1, 2)
for a, b, name in ((66, 6, "staging only | epilogue only"), (0, 0, "whole | whole"), (66, 66, "staging | staging"), (6, 6, "epilogue | epilogue")):
    print(name)

    def only_a(s0, s1):
        with torch.cuda.stream(s0):
            chain(a, C1)

    def only_b(s0, s1):
        with torch.cuda.stream(s0):
            chain(b, C2)

    def seq(s0, s1):
        with torch.cuda.stream(s0):
            chain(a, C1); chain(b, C2)

    def par(s0, s1):
        with torch.cuda.stream(s0):
            chain(a, C1)
        with torch.cuda.stream(s1):
            chain(b, C2)

    timed(only_a, "  first alone"); timed(only_b, "  second alone"); timed(seq, "  one after the other"); timed(par, "  side by side")
