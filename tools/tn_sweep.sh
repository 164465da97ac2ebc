#!/bin/bash
# developer tool: device-side wgrad kernel times for several split heuristics (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "384 1024" "384 512" "640 512"; do
  set -- $cfg
  for sh in lin320 lin640 lin1280 ff320 conv320 conv1280; do
    rm -rf gpurun_out/tnsweep; echo "run $1 $2 $sh" >> gpurun_out/sweep_progress.log
    SDT_TN_TARGET_WG=$1 SDT_TN_MIN_ROWS=$2 rocprofv3 --kernel-trace --stats -d gpurun_out/tnsweep -o s --output-format csv -- python tools/gemm_micro.py $sh 30 > /dev/null 2>&1
    python - "$1" "$2" "$sh" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open('gpurun_out/tnsweep/s_kernel_stats.csv')) if 'gemm_tn' in r['Name']]
for r in rows:
    print(f"wg={sys.argv[1]:>5s} rows={sys.argv[2]:>5s} {sys.argv[3]:8s} {r['Name'][5:30]:26s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:7.1f} us min={float(r['MinNs'])/1e3:7.1f}")
PY
  done
done
