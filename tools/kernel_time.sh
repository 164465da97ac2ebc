#!/bin/bash
# developer tool: device time of the GEMM kernels of tools/gemm_micro.py shapes under rocprofv3, for the default build or SDT_LIB
# usage: tools/kernel_time.sh "ff320 lin320" [label]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sh in $1; do
  rm -rf gpurun_out/kt
  rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o s --output-format csv -- python3 tools/gemm_micro.py $sh 30 > /dev/null 2>&1
  python - "$2" "$sh" <<'PY'
import csv, sys
for r in csv.DictReader(open('gpurun_out/kt/s_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('gemm_nt', 'gemm_tn', 'conv3x3', 'conv_wgrad')):
        print(f"{sys.argv[1]:8s} {sys.argv[2]:10s} {r['Name'][5:52]:48s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:7.1f} us", flush=True)
PY
done
rm -rf gpurun_out/kt
