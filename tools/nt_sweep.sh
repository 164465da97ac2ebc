#!/bin/bash
# developer tool: device-side NT GEMM times for small-M shapes under different split-K thresholds (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "32 8" "8 4" "8 2" "4 2"; do
  set -- $cfg
  for sh in temb temb320 clip clipff; do
    rm -rf gpurun_out/ntsweep
    SDT_NT_SPLIT_MINT=$1 SDT_NT_SPLIT_STEPS=$2 rocprofv3 --kernel-trace --stats -d gpurun_out/ntsweep -o s --output-format csv -- python3 tools/gemm_micro.py $sh 30 > /dev/null 2>&1
    python - "$1" "$2" "$sh" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open('gpurun_out/ntsweep/s_kernel_stats.csv')) if 'gemm_nt' in r['Name']]
for r in rows:
    print(f"minT={sys.argv[1]:>3s} steps={sys.argv[2]:>2s} {sys.argv[3]:8s} {r['Name'][5:34]:30s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:7.1f} us", flush=True)
PY
  done
done
