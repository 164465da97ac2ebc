"""Developer tool: idle time between kernels from a rocprofv3 --kernel-trace CSV (columns Start_Timestamp / End_Timestamp in ns).
Prints, for the last `steps` executions of the step, the busy time, the summed gaps and the largest gaps with the kernels either side."""
import csv
import sys

path, steps = sys.argv[1], int(sys.argv[2])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "")))
rows.sort()
# step boundary: the first kernel of a step is the pixel layout conversion
starts = [i for i, r in enumerate(rows) if "nchw_to_nhwc_kernel" in r[2]]
starts = starts[-steps - 1:]
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    t_end = seg[0][0]
    busy_end = seg[0][0]
    gaps = []
    for s, e, n, q in seg:
        if s > busy_end:
            gaps.append((s - busy_end, prev, n))
        if e > busy_end:
            busy_end, prev = e, n
    span = rows[b][0] - seg[0][0]
    tot_gap = sum(g[0] for g in gaps) + max(rows[b][0] - busy_end, 0)
    big = sorted(gaps, reverse=True)[:6]
    print(f"step span {span / 1e6:.3f} ms  kernels {len(seg)}  idle {tot_gap / 1e6:.3f} ms  (end-of-step gap {max(rows[b][0] - busy_end, 0) / 1e3:.1f} us)")
    for g, p, n in big:
        print(f"    {g / 1e3:8.1f} us  after {p[:50]}  before {n[:50]}")
