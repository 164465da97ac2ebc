"""Developer micro-benchmark: replay grouped weight-gradient launches (sdt_gemm_tn_wgrad_group / sdt_conv_wgrad_group) with the
problem lists of a real step (`SDT_WGRAD_DUMP=1 SDT_GRAPH=0 python bench.py --steps 1 --warmup 0 ... | grep WGRAD_GROUP > file`) and
time every launch with device events.  usage: tn_group_micro.py <dump file> [reps]   (SDT_LIB selects the build)"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stable_diffusion_training_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
G16 = os.environ.get("MICRO_GRAD_BF16", "1") != "0"  # dW as bf16 (what the step's quantised stores use) or float32
lines = [l.split() for l in open(sys.argv[1]) if l.startswith("WGRAD_GROUP")]
seen, groups = set(), []
for l in lines:  # one copy of each distinct launch, with its multiplicity
    key = (l[1], l[2])
    if key in seen:
        for g in groups:
            if g[0] == key:
                g[1] += 1
        continue
    seen.add(key)
    groups.append([key, 1])
stream = torch.cuda.current_stream().cuda_stream
ws = torch.zeros(512 << 20, dtype=torch.uint8, device=dev)
tot = 0.0
for (kind, body), mult in groups:
    keep, probs, flops = [], [], 0.0
    for q in body.split(";"):
        v = [int(t) for t in q.split(",")]
        if kind == "dense":
            M, K1, N, lda, ldb, nseg = v
            a = torch.randn(M, lda, device=dev).to(torch.bfloat16)
            dy = torch.randn(M, ldb, device=dev).to(torch.bfloat16)
            dw = torch.empty(K1 * N, dtype=torch.bfloat16 if G16 else torch.float32, device=dev)
            keep += [a, dy, dw]
            probs.append(_lib.SdtTnProblem(a.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, M, K1, N, K1, N, lda, ldb, nseg if nseg else N, nseg,
                                           K1 * nseg if nseg else 0, None, int(G16)))
            flops += 2.0 * M * K1 * N
        else:
            B, H, W, K1, N, k, stride = v
            a = torch.randn(B * H * W * stride * stride, K1, device=dev).to(torch.bfloat16)
            dy = torch.randn(B * H * W, N, device=dev).to(torch.bfloat16)
            dw = torch.empty(k * k * K1 * N, dtype=torch.bfloat16 if G16 else torch.float32, device=dev)
            keep += [a, dy, dw]
            geom = _lib.SdtConvGeom(B, H * stride, W * stride, H, W, k, k, stride, k // 2, k // 2)
            probs.append(_lib.SdtConvWgradProblem(a.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, geom, K1, N, K1, N, K1, N, None, int(G16)))
            flops += 2.0 * B * H * W * K1 * N * k * k
    if kind == "dense":
        arr = (_lib.SdtTnProblem * len(probs))(*probs)
        fn = lambda: lib.sdt_gemm_tn_wgrad_group(arr, len(probs), ws.data_ptr(), ws.numel(), stream)
    else:
        arr = (_lib.SdtConvWgradProblem * len(probs))(*probs)
        fn = lambda: lib.sdt_conv_wgrad_group(arr, len(probs), ws.data_ptr(), ws.numel(), stream)
    rc = fn()
    assert rc == 0, lib.sdt_last_error().decode()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tot += us * mult
    dims = body.split(";")
    print(f"{kind:5s} x{mult} n={len(probs):2d} {us:8.1f} us {flops / us / 1e6:7.1f} TF  first={dims[0]} last={dims[-1]}", flush=True)
    del keep
print(f"sum over the step's launches: {tot / 1e3:.3f} ms")
