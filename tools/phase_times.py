"""Developer tool: per-phase GPU time of eager train_steps from a `rocprofv3 --marker-trace --kernel-trace` run: kernels are
attributed to the roctx range (stable_diffusion_training_amd/trace.py) whose host interval contains their dispatch time."""
import csv
import glob
import sys

d = sys.argv[1]
mk = glob.glob(d + "/**/*marker_api_trace.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not mk or not kt:
    sys.exit("marker / kernel trace csv not found under " + d)
ranges = []
for r in csv.DictReader(open(mk[0])):
    name = r.get("Function") or r.get("Message") or ""
    ranges.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ranges.sort()
agg = {}
# kernels launched eagerly run shortly after their host-side launch; the phases are long (ms), so attribute by the kernel's START
# time shifted into the host range order: walk kernels in time order, assign to the latest range that began before the launch
kernels = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(kt[0])))
import bisect
starts = [a for a, _, _ in ranges]
for ks, ke in kernels:
    i = bisect.bisect_right(starts, ks) - 1
    name = ranges[i][2] if i >= 0 else "(before first range)"
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1
    a[1] += (ke - ks) / 1e6
steps = max(sum(1 for r in ranges if r[2] == "vae_encode"), 1)
print(f"| phase (roctx range) | kernels/step | GPU ms/step |\n|---|---|---|")
for name, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{name}` | {n / steps:.0f} | {ms / steps:.2f} |")
