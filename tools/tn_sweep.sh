#!/bin/bash
# developer tool: device-side wgrad kernel times for several tile / split heuristics (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "1 384 1024" "1 512 512" "1 256 1024" "2 384 1024" "2 256 1024" "2 512 512"; do
  set -- $cfg
  for sh in lin320 lin640 lin1280 qkv320 qkv640 clip; do
    rm -rf gpurun_out/tnsweep; echo "run $1 $2 $3 $sh" >> gpurun_out/sweep_progress.log
    SDT_TN_TM=$1 SDT_TN_TARGET_WG=$2 SDT_TN_MIN_ROWS=$3 rocprofv3 --kernel-trace --stats -d gpurun_out/tnsweep -o s --output-format csv -- python3 tools/gemm_micro.py $sh 30 > /dev/null 2>&1
    python - "$1" "$2" "$3" "$sh" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open('gpurun_out/tnsweep/s_kernel_stats.csv')) if 'gemm_tn' in r['Name']]
for r in rows:
    print(f"tm={sys.argv[1]} wg={sys.argv[2]:>5s} rows={sys.argv[3]:>5s} {sys.argv[4]:8s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:7.1f} us min={float(r['MinNs'])/1e3:7.1f}", flush=True)
PY
  done
done
