"""Developer tool: per-step kernel table from a rocprofv3 --kernel-trace --stats run of `bench.py --steps K --warmup W --no-roofline
--no-cpu-baseline`.  The process runs the step W + K + 2 times on the device (two eager set-up steps; capturing executes nothing), so
per-step figures are totals / (K + W + 2); one-off initialisation kernels (parameter upload copies, fills) are listed apart."""
import csv, sys, re
path, steps = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
INIT = ("__amd_rocclr_copyBuffer", "FillFunctor", "__amd_rocclr_fillBuffer", "param_prepare_kernel", "distribution_elementwise", "direct_copy_kernel")
fam = {"gemm fwd+dgrad": ("gemm_nt_kernel", "conv3x3_halo_kernel"), "wgrad": ("gemm_tn_kernel", "conv_wgrad3_kernel", "gemm_tn_group"),
       "attention": ("attn_",), "norms": ("gn_", "ln_", "partial_reduce"), "optimizer": ("lion", "sqnorm", "zero_ranges")}
tot = cnt = 0.0
agg = {k: [0.0, 0.0] for k in fam}
other = [0.0, 0.0]
init = [0.0, 0.0]
table = []
for r in rows:
    n, ms, c = r["Name"], float(r["TotalDurationNs"]) / 1e6, int(r["Calls"])
    if any(t in n for t in INIT):
        init[0] += ms; init[1] += c
        continue
    tot += ms; cnt += c
    table.append((ms / steps, c / steps, float(r["AverageNs"]) / 1e3, n))
    for k, pats in fam.items():
        if any(n.lstrip("void ").startswith(p) for p in pats):
            agg[k][0] += ms; agg[k][1] += c
            break
    else:
        other[0] += ms; other[1] += c
print(f"per step: {tot / steps:.2f} ms of kernels in {cnt / steps:.0f} launches   (init, whole process: {init[0]:.1f} ms in {init[1]:.0f} launches)")
for k, (ms, c) in agg.items():
    print(f"  {k:16s} {ms / steps:7.2f} ms  {c / steps:6.0f} launches")
print(f"  {'other':16s} {other[0] / steps:7.2f} ms  {other[1] / steps:6.0f} launches")
for ms, c, avg, n in sorted(table, reverse=True)[: int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    short = re.sub(r"\(.*", "", n)[:80]
    print(f"{ms:7.3f} ms {c:7.1f}/step {avg:8.1f} us  {short}")
