"""Developer tool: algorithmic operand bytes per step of the five largest kernel families beside the bytes the PMC passes counted
(profiles/<round>_pmc_traffic.json: 2 executions of the step per pass), so that the wasted-traffic ratio is on the page
(VERDICT r3, item 9).  Operand bytes = every input read once + every output written once:
  GEMM / conv forward + input gradient (tools/data/sd15_b4_gemm_shapes.txt, the shapes of one step):  2 B x (M x K [x 1: a convolution's
      input is read once, not once per tap] + taps x K x N + M x N)
  weight gradients (tools/data/sd15_b4_wgrad_groups.txt): 2 B x (M x K1 + M x N) + 2 B x taps x K1 x N (bf16 gradient)
  attention: q, k, v, o forward; q, k, v, o, dO read + dq, dk, dv written backward (bf16), from the step's (B, H, Nq, Nk, D) list below
  norms: forward x read + y written (statistics come from the producer's epilogue), backward x, dy read twice + dx written
  optimizer: 22.5 B per parameter (fp32 master r/w, bf16 gradient, int8 codes r/w, scales, fp32 EMA r/w, bf16 mirror) for the quantised
      leaves, 26 B for the others."""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
pmc = json.load(open(f"{root}/profiles/{rnd}_pmc_traffic.json"))["kernels"]
STEPS = 2  # bench.py --steps 1 --warmup 1 under the PMC passes


def counted(*pats):
    return sum(v["total_bytes_per_launch"] * v["launches"] for k, v in pmc.items() if any(p in k for p in pats)) / STEPS


nt = 0.0
for line in open(f"{root}/tools/data/sd15_b4_gemm_shapes.txt"):
    if not line.startswith("nt "):
        continue
    shape, rest = line[3:].split(" calls=")
    M, N, K, taps = [int(t) for t in shape.strip("() ").split(",")[:4]]
    calls = int(rest.split()[0])
    nt += calls * 2.0 * (M * K + taps * K * N + M * N)
tn = 0.0
for line in open(f"{root}/tools/data/sd15_b4_wgrad_groups.txt"):
    t = line.split()
    for q in t[2].split(";"):
        v = [int(x) for x in q.split(",")]
        if t[1] == "dense":
            M, K1, N = v[:3]
            tn += 2.0 * (M * K1 + M * N) + 2.0 * K1 * N
        else:
            B, H, W, K1, N, k, stride = v
            tn += 2.0 * (B * H * W * stride * stride * K1 + B * H * W * N) + 2.0 * k * k * K1 * N
# SD1.5 at batch 4: (layers, B, heads, Nq, Nk, D); self- and cross-attention of the 16 transformer blocks + the text tower's 12 layers
attn_layers = [(5, 4, 8, 4096, 4096, 40), (5, 4, 8, 4096, 77, 40), (5, 4, 8, 1024, 1024, 80), (5, 4, 8, 1024, 77, 80),
               (6, 4, 8, 256, 256, 160), (6, 4, 8, 256, 77, 160), (12, 4, 12, 77, 77, 64)]
attn = sum(L * 2.0 * B * H * D * ((2 * Nq + 2 * Nk) + (3 * Nq + 2 * Nk) + (Nq + 2 * Nk)) for L, B, H, Nq, Nk, D in attn_layers)
# GroupNorm tensors of one step (elements): VAE encoder (frozen: forward only) and UNet (forward + backward); LayerNorm: 3 per block + text tower
vae_gn = 4 * (2 * 512 * 512 * 128 + 1 * 512 * 512 * 128 + 256 * 256 * 128 + 3 * 256 * 256 * 256 + 128 * 128 * 256 + 3 * 128 * 128 * 512 + 5 * 64 * 64 * 512 + 2 * 64 * 64 * 512)
unet_gn = 4 * (61 * 0 + (4 * 4096 * 320 + 4096 * 320 + 3 * 1024 * 640 + 1024 * 320 + 3 * 256 * 1280 + 256 * 640 + 4 * 64 * 1280 + 2 * 64 * 1280
                         + 3 * 64 * 2560 + 3 * 256 * 2560 + 256 * 1920 + 2 * 1024 * 1920 + 1024 * 1280 + 1024 * 960 + 4096 * 960 + 2 * 4096 * 640
                         + 6 * 64 * 1280 + 6 * 256 * 1280 + 6 * 1024 * 640 + 7 * 4096 * 320))
ln = 4 * 3 * (5 * 4096 * 320 + 5 * 1024 * 640 + 6 * 256 * 1280) + 25 * 308 * 768
norms = 2.0 * (2 * vae_gn) + 2.0 * (2 + 5) * unet_gn + 2.0 * (2 + 5) * ln
quant, rest = 859.5e6 + 123.1e6 - 25e6, 25e6
opt = 22.5 * quant + 26.0 * rest
rows = [("sdt_gemm_nt_bf16 (gemm_nt_kernel + conv3x3_halo_kernel)", nt, counted("gemm_nt_kernel", "conv3x3_halo_kernel")),
        ("weight gradients (gemm_tn_* + conv_wgrad3_*)", tn, counted("gemm_tn", "conv_wgrad3")),
        ("attention", attn, counted("attn_")), ("norms (gn_*, ln_*)", norms, counted("gn_", "ln_", "partial_reduce")),
        ("optimizer (lion8 / lion32 / sqnorm / zero)", opt, counted("lion", "sqnorm", "zero_ranges", "sum_f64"))]
print("## Operand bytes beside counted bytes, per step (SD1.5 512x512, batch 4)\n")
print("Operand bytes: every input read once, every output written once (`tools/operand_bytes.py`; GroupNorm tensor list approximate).  Counted: "
      "2 x FETCH_SIZE + WRITE_SIZE of the PMC passes (fabric-side: Infinity-Cache hits are counted, `MI355X_MICROARCH.md`).\n")
print("| family | operand GB / step | counted GB / step | counted / operand |\n|---|---|---|---|")
for name, a, c in rows:
    print(f"| {name} | {a / 1e9:.2f} | {c / 1e9:.2f} | {c / a:.2f} |")
