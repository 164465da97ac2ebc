"""Developer tool: turn the files the GPU runs of tools/refresh_profiles.sh left under gpurun_out/r02/ into the committed
profiles/r02_* artefacts: kernel-time tables (rocprofv3 --kernel-trace --stats), HBM traffic per kernel (separate --pmc
FETCH_SIZE / WRITE_SIZE passes, gfx950 correction: read bytes = 2 x FETCH_SIZE KB), MFMA-busy share per kernel
(--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE), roctx phase times, the bench lines."""
import collections
import csv
import glob
import json
import os
import shutil

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = f"{root}/gpurun_out/r02", f"{root}/profiles"
NSTEP = 16  # bench.py --steps 10 --warmup 2: 3 graph set-up + 2 warm-up + 10 timed + 1 eager instrumented step


def short(n):
    return n.replace("void ", "").split("(")[0]


def find(pattern):
    hits = glob.glob(f"{src}/{pattern}", recursive=True)
    return hits[0] if hits else None


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(path)):
        a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return agg


def bench_line(name):
    p = f"{src}/bench_{name}.json"
    if not os.path.exists(p):
        return None
    b = json.loads(open(p).read().strip().splitlines()[-1])
    json.dump(b, open(f"{dst}/r02_bench_{name}.json", "w"), indent=1)
    return b


L = ["# Round 2 profile summary (MI355X, one GPU)\n"]
bench = bench_line("sd15")
ks = find("kstats/**/k_kernel_stats.csv")
if ks:
    shutil.copy(ks, f"{dst}/r02_bench_sd15_kernel_stats.csv")
    rows = list(csv.DictReader(open(ks)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    L += ["## SD1.5 512x512, batch 4 (BASELINE configs[1]): `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline`\n",
          f"`profiles/r02_bench_sd15_kernel_stats.csv` ({NSTEP} steps in the process: 3 graph set-up + 2 warm-up + 10 timed replays + 1 eager instrumented step;\n"
          f"one-off initialisation copies included).  GPU kernel time {tot / NSTEP / 1e6:.1f} ms per step over {sum(int(r['Calls']) for r in rows) / NSTEP:.0f} launches"
          + (f"; wall {bench['ms_per_step']:.1f} ms/step ({bench['value']:.1f} images/s, `profiles/r02_bench_sd15.json`).\n" if bench else ".\n"),
          "| kernel | launches/step | ms/step | avg us | share |\n|---|---|---|---|---|"]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:34]:
        L.append(f"| `{short(r['Name'])[:62]}` | {int(r['Calls']) / NSTEP:.0f} | {float(r['TotalDurationNs']) / NSTEP / 1e6:.2f} | "
                 f"{float(r['AverageNs']) / 1e3:.1f} | {100 * float(r['TotalDurationNs']) / tot:.1f}% |")
    fam = {"sdt_gemm_nt_bf16 (gemm_nt_kernel + conv3x3_halo_kernel)": ("gemm_nt_kernel", "conv3x3_halo_kernel"),
           "weight gradients (gemm_tn_kernel + conv_wgrad3_kernel)": ("gemm_tn_kernel", "conv_wgrad3_kernel"),
           "attention": ("attn_",), "norms (gn_*, ln_*, reduces)": ("gn_", "ln_", "partial_reduce"),
           "optimizer (lion8 / lion32 / sqnorm / prepare / zero)": ("lion", "sqnorm", "param_prepare", "zero_ranges")}
    L.append("\n| kernel family | launches/step | ms/step |\n|---|---|---|")
    for name, pre in fam.items():
        sel = [r for r in rows if short(r["Name"]).startswith(pre)]
        L.append(f"| {name} | {sum(int(r['Calls']) for r in sel) / NSTEP:.0f} | {sum(float(r['TotalDurationNs']) for r in sel) / NSTEP / 1e6:.2f} |")
    if bench and "roofline" in bench:
        rf = bench["roofline"]
        sel = [r for r in rows if short(r["Name"]).startswith(("gemm_nt_kernel", "conv3x3_halo_kernel"))]
        ms = sum(float(r["TotalDurationNs"]) for r in sel) / NSTEP / 1e6
        L.append(f"\nDominant family by rocprof: {ms:.2f} ms/step for {rf['algorithmic_tflop_per_step']:.2f} algorithmic TFLOP = "
                 f"{rf['algorithmic_tflop_per_step'] / ms * 1e3:.0f} TFLOP/s = {rf['algorithmic_tflop_per_step'] / ms * 1e3 / 2500:.3f} of the 2.5 PFLOP/s dense bf16 peak "
                 f"(bench line, raw HIP events: {rf['achieved']:.0f}; overhead-corrected: {rf.get('achieved_calibrated', 0):.0f}).")
        sel = [r for r in rows if short(r["Name"]).startswith(("gemm_tn_kernel", "conv_wgrad3_kernel"))]
        ms = sum(float(r["TotalDurationNs"]) for r in sel) / NSTEP / 1e6
        L.append(f"Weight-gradient family by rocprof: {ms:.2f} ms/step.\n")

fe, wr = find("pmc_fetch/**/*counter_collection.csv"), find("pmc_write/**/*counter_collection.csv")
if fe and wr:
    f, w = counters(fe), counters(wr)
    traffic = {}
    for k, v in f.items():
        fv, fn = v["FETCH_SIZE"]
        wv, wn = w.get(k, {}).get("WRITE_SIZE", (0.0, 0))
        rd, wb = 2.0 * fv * 1024 / max(fn, 1), wv * 1024 / max(wn, 1)
        traffic[k] = {"launches": fn, "read_bytes_per_launch": rd, "write_bytes_per_launch": wb, "total_bytes_per_launch": rd + wb}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), python3 bench.py --steps 1 --warmup 1 (eager, "
                         "SDT_GRAPH=0); read bytes = 2 x FETCH_SIZE KB per the MI355X guide's gfx950 correction", "kernels": traffic},
              open(f"{dst}/r02_pmc_traffic.json", "w"), indent=1)
    L.append("## HBM-side traffic per launch (PMC, separate passes; `profiles/r02_pmc_traffic.json`)\n")
    L.append("| kernel | launches | read MB (2 x FETCH) | write MB | total MB |\n|---|---|---|---|---|")
    for k, v in sorted(traffic.items(), key=lambda kv: -kv[1]["total_bytes_per_launch"] * kv[1]["launches"])[:16]:
        L.append(f"| `{k[:62]}` | {v['launches']} | {v['read_bytes_per_launch'] / 1e6:.1f} | {v['write_bytes_per_launch'] / 1e6:.1f} | {v['total_bytes_per_launch'] / 1e6:.1f} |")
    opt = [v for k, v in traffic.items() if k.startswith(("lion8_kernel", "lion32_kernel", "sqnorm_kernel", "param_prepare", "zero_ranges"))]
    if opt and bench:
        tot_b = sum(v["total_bytes_per_launch"] * v["launches"] for v in opt) / 2.0  # the pass ran 2 steps (1 warm-up + 1)
        L.append(f"\nOptimizer chain (sqnorm + lion8 + lion32 + prepare + zero): {tot_b / 1e9:.1f} GB per step by the counters.\n")

mf = find("pmc_mfma/**/*counter_collection.csv")
if mf:
    m = counters(mf)
    M = ["# MFMA-busy share per kernel (round 2)\n",
         "`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 1 --warmup 1` (eager, SDT_GRAPH=0).\n"
         "SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs;\n"
         "GRBM_GUI_ACTIVE is the sum over the 8 XCDs of their active cycles (MI355X guide).  busy share = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024):\n"
         "the fraction of SIMD-cycles of the launch in which the matrix pipe worked (1.0 = every SIMD issuing MFMAs back to back).\n",
         "| kernel | launches | MFMA-busy share |\n|---|---|---|"]
    rowsm = []
    for k, v in m.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"][0] > 0:
            busy, act = v["SQ_VALU_MFMA_BUSY_CYCLES"][0], v["GRBM_GUI_ACTIVE"][0]
            rowsm.append((busy / (act / 8 * 1024), v["GRBM_GUI_ACTIVE"][1], k, busy))
    for share, n, k, busy in sorted(rowsm, key=lambda r: -r[3])[:24]:
        M.append(f"| `{k[:62]}` | {n} | {share:.3f} |")
    open(f"{dst}/r02_mfma_busy.md", "w").write("\n".join(M) + "\n")

ph = f"{src}/phase_times.md"
if os.path.exists(ph) and os.path.getsize(ph) > 0:
    L.append("## Phases of an eager step (roctx ranges, `rocprofv3 --marker-trace --kernel-trace`)\n")
    L.append(open(ph).read())

for cfg, title in (("sd21_768", "SD2.1-768 v-prediction, batch 4 (BASELINE configs[3])"), ("sdxl_1024", "SDXL-base 1024x1024, batch 2 (BASELINE configs[4])")):
    b = bench_line(cfg)
    ks = find(f"kstats_{cfg}/**/k_kernel_stats.csv")
    if not b:
        continue
    L.append(f"## {title}: `python bench.py --config {cfg}`\n")
    L.append(f"{b['value']:.2f} images/s, {b['ms_per_step']:.1f} ms/step, {b['config']['step_mfma_frac']:.3f} of the MFMA peak over the whole step "
             f"({b['config']['step_tflop_per_image']} TFLOP/image), HBM high-water {b['config']['peak_hbm_GiB']:.1f} GiB (`profiles/r02_bench_{cfg}.json`).\n")
    if ks:
        shutil.copy(ks, f"{dst}/r02_bench_{cfg}_kernel_stats.csv")
        rows = list(csv.DictReader(open(ks)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        L.append("| kernel | share of GPU time | avg us |\n|---|---|---|")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
            L.append(f"| `{short(r['Name'])[:62]}` | {100 * float(r['TotalDurationNs']) / tot:.1f}% | {float(r['AverageNs']) / 1e3:.1f} |")
        L.append("")
open(f"{dst}/r02_summary.md", "w").write("\n".join(L) + "\n")
print("\n".join(L[:12]))
