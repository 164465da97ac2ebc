"""Developer micro-benchmark: run a few GEMM / conv shapes back to back (for rocprofv3 --kernel-trace / --pmc)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stable_diffusion_training_amd import ops, params, nets

dev = torch.device("cuda:0")
BF = torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def store(spec):
    st = params.ParamStore(spec, device=dev, quantise=False)
    st.load(nets.init_params(spec, 0))
    st.prepare()
    return st


def timeit(name, fn, flops):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:44s} {dt*1e6:9.1f} us  {flops/dt/1e12:7.1f} TF", flush=True)


cases = {
    "convvae128": ("conv", 4, 512, 512, 128, 128, 3),  # (1048576,128,128,9): 4096 tiles
    "convvae256": ("conv", 4, 256, 256, 256, 256, 3),  # (262144,256,256,9): 2048 tiles
    "conv512": ("conv", 4, 128, 128, 512, 512, 3),     # (65536,512,512,9)
    "conv320": ("conv", 4, 64, 64, 320, 320, 3),       # (16384,320,320,9)
    "conv1280": ("conv", 4, 16, 16, 1280, 1280, 3),    # (1024,1280,1280,9)
    "conv1280s": ("conv", 4, 8, 8, 1280, 1280, 3),     # (256,...) split-K
    "lin1280": ("lin", 1024, 1280, 1280),
    "lin320": ("lin", 16384, 320, 320),
    "lin640": ("lin", 4096, 640, 640),
    "ff320": ("lin", 16384, 320, 2560),
    "qkv320": ("lin", 16384, 320, 960),
    "qkv640": ("lin", 4096, 640, 1920),
    "ff2_1280": ("lin", 1024, 5120, 1280),             # nt (1024,1280,5120): 128-tiles + split-K
    "clip": ("lin", 308, 768, 768),
    "clipff": ("lin", 308, 3072, 768),
    "temb": ("lin", 4, 1280, 1280),
    "temb320": ("lin", 4, 1280, 320),
}
for name, c in cases.items():
    if which != "all" and which != name:
        continue
    if c[0] == "conv":
        _, B, H, W, Ci, Co, k = c
        st = store([("c/kernel", (k, k, Ci, Co)), ("c/bias", (Co,))])
        x = torch.randn(B, H, W, Ci, device=dev).to(BF).requires_grad_(True)
        y = ops.conv2d(x, st, "c")
        dy = torch.randn_like(y)
        fl = 2.0 * B * H * W * Ci * Co * k * k
        with torch.no_grad():
            timeit(name + " fprop", lambda: ops.conv2d(x, st, "c"), fl)
            if os.environ.get("MICRO_GN"):  # the epilogue that also accumulates the next GroupNorm's statistics
                timeit(name + " fprop+gn", lambda: ops.conv2d(x, st, "c", gn_groups=32), fl)
        timeit(name + " fwd+bwd(dgrad+wgrad)", lambda: ops.conv2d(x, st, "c").backward(dy), 3 * fl)
    else:
        _, M, K, N = c
        st = store([("l/kernel", (K, N)), ("l/bias", (N,))])
        x = torch.randn(M, K, device=dev).to(BF).requires_grad_(True)
        dy = torch.randn(M, N, device=dev).to(BF)
        fl = 2.0 * M * K * N
        with torch.no_grad():
            timeit(name + " fwd", lambda: ops.linear(x, st, "l"), fl)
        timeit(name + " fwd+bwd", lambda: ops.linear(x, st, "l").backward(dy), 3 * fl)
        xd = x.detach()
        timeit(name + " wgrad only", lambda: ops.gemm_tn(xd, dy, st.g("l/kernel"), M, K, N, K, N, 1, K, N), fl)
