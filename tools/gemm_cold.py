"""Developer probe: forward Dense GEMMs with WARM operands (back-to-back launches: weights stay in L2 / Infinity Cache) against
COLD ones (a 1 GiB buffer is rewritten between launches, as in the training step where every layer's weights are first read
from HBM): how much of a GEMM's in-step time is the fetch latency of its operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stable_diffusion_training_amd import ops, params, nets
dev = torch.device("cuda:0")
SHAPES = [(16384, 320, 2560), (16384, 2560, 320), (16384, 320, 320), (4096, 640, 5120), (4096, 640, 640), (1024, 1280, 10240),
          (1024, 10240, 1280), (1024, 1280, 1280), (256, 1280, 10240), (308, 768, 3072), (308, 3072, 768)]
if len(sys.argv) > 3:
    SHAPES = [tuple(int(a) for a in sys.argv[1:4])]
junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)
for M, K, N in SHAPES:
    spec = [("l/kernel", (K, N)), ("l/bias", (N,))]
    st = params.ParamStore(spec, device=dev, quantise=False, trainable=False)
    st.load(nets.init_params(spec, 0)); st.prepare()
    x = torch.randn(M, K, device=dev).bfloat16()
    def timed(cold, reps=8):
        ts = []
        for _ in range(reps):
            if cold:
                junk.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            with torch.no_grad():
                ops.linear(x, st, "l")
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]
    timed(False)
    w, c = timed(False), timed(True)
    fl = 2.0 * M * K * N
    print(f"({M},{K},{N})  warm {w:7.1f} us {fl/w/1e6:6.0f} TF   cold {c:7.1f} us {fl/c/1e6:6.0f} TF", flush=True)
