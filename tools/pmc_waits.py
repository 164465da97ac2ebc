"""Developer tool: where the waves of each kernel spend their cycles.  Reads a `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES` counter_collection.csv and prints,
per kernel, the share of wave time parked in s_waitcnt / barriers (WAIT_ANY), stalled at issue (WAIT_INST_ANY: MFMA dependency /
pipe), issuing (ACTIVE_INST_ANY), the VALU and LDS shares, LDS bank-conflict cycles per LDS-active cycle, and MFMA-busy cycles per
wave cycle (x 4: the SQ wave counters tick in quad-cycles)."""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        n[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]
print("| kernel | launches | parked (waitcnt / barrier) | issue stall | issuing | VALU | LDS | LDS conflict / LDS active | MFMA busy / wave cycle |\n|---|---|---|---|---|---|---|---|---|")
for k, v in rows:
    w = v["SQ_WAVE_CYCLES"] or 1.0
    lds = v["SQ_ACTIVE_INST_LDS"] or 1.0
    print(f"| `{k}` | {n[k]} | {v['SQ_WAIT_ANY'] / w:.2f} | {v['SQ_WAIT_INST_ANY'] / w:.2f} | {v['SQ_ACTIVE_INST_ANY'] / w:.2f} | {v['SQ_ACTIVE_INST_VALU'] / w:.2f} | "
          f"{v['SQ_ACTIVE_INST_LDS'] / w:.3f} | {v['SQ_LDS_BANK_CONFLICT'] / lds:.2f} | {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * w):.3f} |")
